// Default build (`make`, ABPARTNERS=0): the A/B-only kernels are left out of the library — conv_wino4 (Winograd F(4,3), conv_wino4_impl.h; reached with
// FV_WINO44=0 in an ABPARTNERS=1 build).  Their launchers report "no kernel" and conv_layer_create packs no F(4,3) fragments, so FV_WINO44=0 selects the
// F(2,3) per-layer kernel instead.
#include "fv_internal.h"
namespace fv {
bool have_conv_wino4() { return false; }
bool launch_conv_wino4_k7(const ConvParams&, int, hipStream_t) { return false; }
bool launch_conv_wino4_k11(const ConvParams&, int, hipStream_t) { return false; }
}  // namespace fv
