// Winograd F(4,3) conv kernels for kernel size 7 (one translation unit per size: parallel builds).
#include "conv_wino4_impl.h"
namespace fv {
bool have_conv_wino4() { return true; }   // (ABPARTNERS=1 builds; abpartner_stubs.hip says false)
bool launch_conv_wino4_k7(const ConvParams& p, int batch, hipStream_t s) { return launch_wino4_k<7>(p, batch, s); }
}  // namespace fv
