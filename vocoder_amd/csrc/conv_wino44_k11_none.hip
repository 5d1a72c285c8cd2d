// Winograd F(4,4) conv kernels for kernel size 11, activation in front: none (one translation unit per size and PRE: parallel builds).
#include "conv_wino44_impl.h"
namespace fv {
bool launch_conv_wino44_k11_none(const ConvParams& p, int rows, int batch, hipStream_t s) { return launch_wino44_k<11, 0>(p, rows, batch, s); }
}  // namespace fv
