// `make F16X3=0`: the split-fp16 precision mode's kernels (conv_f16x3_*.hip, pair_f16x3_*.hip, pair16_f16x3.hip) are left out of the build;
// their launchers report "no kernel", so fv_set_precision / fv_conv_set_precision(FV_PRECISION_F16X3) fail loudly instead of linking them in.
#include "fv_internal.h"
#include "pair_f16x3_params.h"
namespace fv {
bool launch_conv_f16x3_k1(const ConvParams&, int, int, hipStream_t) { return false; }
bool launch_conv_f16x3_k3(const ConvParams&, int, int, hipStream_t) { return false; }
bool launch_conv_f16x3_misc(const ConvParams&, int, int, hipStream_t) { return false; }
bool launch_conv_f16x3_k7(const ConvParams&, int, int, hipStream_t) { return false; }
bool launch_conv_f16x3_k11(const ConvParams&, int, int, hipStream_t) { return false; }
bool launch_pair_f16x3_k3(const PairF16Params&, int, int, int, hipStream_t) { return false; }
bool launch_pair_f16x3_k7(const PairF16Params&, int, int, int, hipStream_t) { return false; }
bool launch_pair_f16x3_k11(const PairF16Params&, int, int, int, hipStream_t) { return false; }
bool launch_pair16_f16x3(const PairF16Params&, int, int, int, hipStream_t) { return false; }
int pair16_f16x3_tile(int, int) { return 1; }
}  // namespace fv
