// Latency variant of the Winograd F(4,4) conv (conv_wino_lat44_impl.h), k = 11
#include "conv_wino_lat44_impl.h"
namespace fv {
bool launch_conv_wino_lat44_k11(const ConvParams& p, int rows, int batch, hipStream_t s) { return launch_wino_lat44_k<11>(p, rows, batch, s); }
}  // namespace fv
