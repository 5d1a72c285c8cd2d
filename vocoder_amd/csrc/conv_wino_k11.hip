// Winograd F(2,3) conv kernels for kernel size 11 (one translation unit per size: parallel builds).
#include "conv_wino_impl.h"
namespace fv {
bool launch_conv_wino_k11(const ConvParams& p, int cfg, int batch, hipStream_t s) { return launch_wino_k<11>(p, cfg, batch, s); }
}  // namespace fv
