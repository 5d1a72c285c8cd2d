// Winograd F(4,3) conv kernels for kernel size 11 (one translation unit per size: parallel builds).
#include "conv_wino4_impl.h"
namespace fv {
bool launch_conv_wino4_k11(const ConvParams& p, int batch, hipStream_t s) { return launch_wino4_k<11>(p, batch, s); }
}  // namespace fv
