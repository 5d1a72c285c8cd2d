// f16x3 precision mode: specialisations of the split-fp16 conv kernel for kernel size 11.
#include "conv_f16x3_impl.h"
namespace fv {
bool launch_conv_f16x3_k11(const ConvParams& p, int cfg, int batch, hipStream_t s) {
    switch (p.dil) {
        case 1: return launch_f16x3_cfg<11, 1>(p, cfg, batch, s);
        case 3: return launch_f16x3_cfg<11, 3>(p, cfg, batch, s);
        case 5: return launch_f16x3_cfg<11, 5>(p, cfg, batch, s);
        default: return false;
    }
}
}  // namespace fv
