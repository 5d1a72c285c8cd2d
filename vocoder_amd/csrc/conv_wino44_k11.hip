// Winograd F(4,4) conv for kernel size 11: dispatch over the activation in front (the kernels: conv_wino44_k11_{none,silu,any}.hip).
#include "fv_internal.h"
namespace fv {
bool launch_conv_wino44_k11_none(const ConvParams& p, int rows, int batch, hipStream_t s);
bool launch_conv_wino44_k11_silu(const ConvParams& p, int rows, int batch, hipStream_t s);
bool launch_conv_wino44_k11_any(const ConvParams& p, int rows, int batch, hipStream_t s);
bool launch_conv_wino44_k11(const ConvParams& p, int rows, int batch, hipStream_t s) {
    return p.pre_act == FV_ACT_NONE ? launch_conv_wino44_k11_none(p, rows, batch, s)
           : p.pre_act == FV_ACT_SILU ? launch_conv_wino44_k11_silu(p, rows, batch, s) : launch_conv_wino44_k11_any(p, rows, batch, s);
}
}  // namespace fv
