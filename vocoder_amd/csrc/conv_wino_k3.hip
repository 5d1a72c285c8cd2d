// Winograd F(2,3) conv kernels for kernel size 3 (one translation unit per size: parallel builds).
#include "conv_wino_impl.h"
namespace fv {
bool launch_conv_wino_k3(const ConvParams& p, int cfg, int batch, hipStream_t s) { return launch_wino_k<3>(p, cfg, batch, s); }
}  // namespace fv
