// Winograd F(4,4) conv for kernel size 7: dispatch over the activation in front (the kernels: conv_wino44_k7_{none,silu,any}.hip).
#include "fv_internal.h"
namespace fv {
bool launch_conv_wino44_k7_none(const ConvParams& p, int rows, int batch, hipStream_t s);
bool launch_conv_wino44_k7_silu(const ConvParams& p, int rows, int batch, hipStream_t s);
bool launch_conv_wino44_k7_any(const ConvParams& p, int rows, int batch, hipStream_t s);
bool launch_conv_wino44_k7(const ConvParams& p, int rows, int batch, hipStream_t s) {
    return p.pre_act == FV_ACT_NONE ? launch_conv_wino44_k7_none(p, rows, batch, s)
           : p.pre_act == FV_ACT_SILU ? launch_conv_wino44_k7_silu(p, rows, batch, s) : launch_conv_wino44_k7_any(p, rows, batch, s);
}
}  // namespace fv
