// Specialisations for the remaining tap counts on the reference's configs: the polyphase ConvTranspose1d
// upsamplers (k/stride = 2 for (16,8) and (4,2); 4 for (8,2); 1 for (2,2) is served by conv_k1) and the
// k=13 pre/post convs of firefly-gan-base.yaml.
#include "conv_mfma_impl.h"
namespace fv {
bool launch_conv_misc(const ConvParams& p, int cfg, int batch, hipStream_t s) {
    if (p.dil != 1) return false;
    switch (p.ks) {
        case 2: return launch_cfg<2, 1>(p, cfg, batch, s);
        case 4: return launch_cfg<4, 1>(p, cfg, batch, s);
        case 13: return launch_cfg<13, 1>(p, cfg, batch, s);
        default: return false;
    }
}
}  // namespace fv
