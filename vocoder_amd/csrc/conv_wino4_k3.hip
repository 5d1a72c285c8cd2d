// Winograd F(4,3) conv kernels for kernel size 3 (one translation unit per size: parallel builds).
#include "conv_wino4_impl.h"
namespace fv {
bool launch_conv_wino4_k3(const ConvParams& p, int rows, int batch, hipStream_t s) { return launch_wino4_k<3>(p, rows, batch, s); }
}  // namespace fv
