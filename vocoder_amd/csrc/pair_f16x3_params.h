// Launch parameters of the fused f16x3 (c1, c2) pair kernel, shared by the host dispatch (conv_layer.hip) and the kernel TUs.
#pragma once

#include "fv_internal.h"

namespace fv {

// (NT = 2 / 3 / 4 measured within 2 % of each other)
constexpr int kPairF16NtC32 = 3;   // C = 32: one m-tile, 4 waves x NT n-tiles of intermediate columns
constexpr int kPairF16ColsC32 = 4 * kPairF16NtC32 * 32;

struct PairF16Params {
    const float* x;        // (B, C, T)
    float* y;              // (B, C, T), must not alias x
    const void* w1h;       // c1 / c2 split weight planes (pack_conv_weights_f16x3 layout)
    const void* w2h;
    const float* b1;
    const float* b2;
    float s1, s2;          // 1 / s_w of the two layers
    int T, n_tiles, nch16; // nch16 = C / 16
    int out_mode;
    float out_scale;
};

bool launch_pair_f16x3_k3(const PairF16Params& p, int C, int dil1, int batch, hipStream_t s);
bool launch_pair_f16x3_k7(const PairF16Params& p, int C, int dil1, int batch, hipStream_t s);
bool launch_pair_f16x3_k11(const PairF16Params& p, int C, int dil1, int batch, hipStream_t s);
// C = 16 (pair16_f16x3.hip): w1h / w2h are ConvLayer::d_wph16 planes; T must be even, x / y 8-byte aligned
bool launch_pair16_f16x3(const PairF16Params& p, int ks, int dil1, int batch, hipStream_t s);
int pair16_f16x3_tile(int ks, int dil1);

}  // namespace fv
