// Latency variant of the Winograd conv (conv_wino_lat_impl.h), k = 7
#include "conv_wino_lat_impl.h"
namespace fv {
bool launch_conv_wino_lat_k7(const ConvParams& p, int nt, int batch, hipStream_t s) { return launch_wino_lat_k<7>(p, nt, batch, s); }
}  // namespace fv
