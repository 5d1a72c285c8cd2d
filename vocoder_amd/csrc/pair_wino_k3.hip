// Winograd fused (c1, c2) pair kernels, k = 3 (one translation unit per kernel size: parallel builds)
#include "pair_wino_impl.h"
namespace fv {
bool launch_pair_wino_k3(const PairParams& p, int C, int dil, int batch, hipStream_t s) { return launch_pair_wino16_k<3>(p, C, dil, batch, s); }
}  // namespace fv
