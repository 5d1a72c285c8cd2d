// Winograd F(4,4) conv kernels for kernel size 11 (one translation unit per size: parallel builds).
#include "conv_wino44_impl.h"
namespace fv {
bool launch_conv_wino44_k11(const ConvParams& p, int rows, int batch, hipStream_t s) { return launch_wino44_k<11>(p, rows, batch, s); }
}  // namespace fv
