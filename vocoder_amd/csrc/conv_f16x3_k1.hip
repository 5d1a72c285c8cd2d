// f16x3 precision mode: the split-fp16 kernel for pointwise convs (ConvNeXt pwconv1 / pwconv2, stage 1x1 convs).
#include "conv_f16x3_impl.h"
namespace fv {
bool launch_conv_f16x3_k1(const ConvParams& p, int cfg, int batch, hipStream_t s) {
    return p.dil == 1 ? launch_f16x3_cfg<1, 1>(p, cfg, batch, s) : false;
}
}  // namespace fv
