// f16x3 precision mode: the split-fp16 kernel for the polyphase form of the transposed convs (2 and 4 taps per phase).
#include "conv_f16x3_impl.h"
namespace fv {
bool launch_conv_f16x3_misc(const ConvParams& p, int cfg, int batch, hipStream_t s) {
    if (p.dil != 1) return false;
    switch (p.ks) {
        case 2: return launch_f16x3_cfg<2, 1>(p, cfg, batch, s);
        case 4: return launch_f16x3_cfg<4, 1>(p, cfg, batch, s);
        default: return false;
    }
}
}  // namespace fv
