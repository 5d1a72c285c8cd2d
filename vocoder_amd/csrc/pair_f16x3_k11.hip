// f16x3 precision mode: fused (c1, c2) ResBlock pair for the wide stages, kernel size 11.
#include "pair_f16x3_impl.h"
namespace fv {
bool launch_pair_f16x3_k11(const PairF16Params& p, int C, int dil1, int batch, hipStream_t s) {
    switch (dil1) {
        case 1: return launch_pair_f16x3_cfg<11, 1>(p, C, batch, s);
        case 3: return launch_pair_f16x3_cfg<11, 3>(p, C, batch, s);
        case 5: return launch_pair_f16x3_cfg<11, 5>(p, C, batch, s);
        default: return false;
    }
}
}  // namespace fv
