// Latency variant of the Winograd conv (conv_wino_lat_impl.h), k = 11
#include "conv_wino_lat_impl.h"
namespace fv {
bool launch_conv_wino_lat_k11(const ConvParams& p, int nt, int batch, hipStream_t s) { return launch_wino_lat_k<11>(p, nt, batch, s); }
}  // namespace fv
