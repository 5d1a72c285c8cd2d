// Specialisations of the fused conv kernel for kernel size 1 (one translation unit per size: parallel builds).
#include "conv_mfma_impl.h"
namespace fv {
bool launch_conv_k1(const ConvParams& p, int cfg, int batch, hipStream_t s) {
    switch (p.dil) {
        case 1: return launch_cfg<1, 1>(p, cfg, batch, s);
        default: return false;
    }
}
}  // namespace fv
