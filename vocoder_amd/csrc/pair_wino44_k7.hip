// Winograd F(4,4) fused (c1, c2) pair kernels, k = 7 (one translation unit per kernel size: parallel builds)
#include "pair_wino44_impl.h"
namespace fv {
bool launch_pair_wino44_k7(const PairParams& p, int C, int dil, int batch, hipStream_t s) { return launch_pair_wino44_k<7>(p, C, dil, batch, s); }
}  // namespace fv
