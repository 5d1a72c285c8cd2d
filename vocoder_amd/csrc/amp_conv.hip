// BigVGAN AMPBlock conv with its anti-aliased SnakeBeta fused in front (round 3):
//
//     y = conv( Activation1d(SnakeBeta)(x) ) + bias [+ res]          (fish_vocoder/modules/generators/bigvgan.py:235-245:
//                                                                      xt = a1(x); xt = c1(xt); xt = a2(xt); xt = c2(xt); x = xt + x)
//
// for the narrow stages (C = 32 / 64), where the separate activation pass (aa_snake_pk_kernel: one tensor read + one written
// per activation, 22 GB of the 61 GB a BigVGAN-24k B = 64 forward moved in round 2) costs as much as a third of the conv it
// feeds.  One workgroup = one tile of 128 output columns of one clip, all C_in * k resident in LDS like resblock_pair.hip:
//
//   per 8-channel chunk   raw window rows (global -> registers, one chunk ahead) -> LDS  X
//                         2x up-sampling FIR (6 + 6 polyphase taps) + snake on (even, odd) sample pairs -> LDS  E / O
//                         12-tap low-pass, stride 2 -> LDS  A[channel][column]   (0 outside [0, T): the conv's zero padding)
//   then                  the conv as implicit GEMM on fp32 MFMA over the resident A (gemm32_resident), bias, residual, store
//
// The activation arithmetic is aa_snake_tile's (small_kernels.hip), instruction for instruction (replicate padding of both
// FIRs at the sequence ends included), and the MFMA loop adds in the order of the tiled conv kernel (8-channel chunk, tap,
// channel pair), so results are bit-identical to the two-kernel path.  The window of a dilated conv carries a halo of
// (k - 1) d + 12 columns on 128, i.e. 1.1 - 1.5x the activation work of the separate pass — VALU work, against two tensor passes
// through HBM and one launch saved per conv.
#include "pair_common.h"

namespace fv {

typedef float f32x2a __attribute__((ext_vector_type(2)));

template <int KS, int DIL, int C>
struct AmpGeom {
    static constexpr int W1 = 128;                                 // output columns per tile
    static constexpr int PAD = (KS - 1) / 2 * DIL;
    static constexpr int WA_RAW = W1 + (KS - 1) * DIL;             // activated columns the conv reads
    static constexpr int WA = (WA_RAW - 16 + 31) / 32 * 32 + 16;   // row stride == 16 (mod 32)
    static constexpr int NM = WA_RAW + 6;                          // 2x-rate positions per row (3 each side for the low-pass)
    static constexpr int WX = WA_RAW + 12;                         // raw columns per row (6 each side in all)
    static constexpr int RC = 8;                                   // rows per activation chunk
    static constexpr int XS = WX + 1, ES = NM + 2;                 // LDS row strides of the chunk arrays
    static constexpr int NCH = C / 8;
    static constexpr int WM = C / 32, WN = 4 / WM;                 // wave grid: one 32-row m-tile per wave row
    static constexpr int NT = W1 / 32 / WN;                        // n-tiles per wave
    static constexpr int LDS_FLOATS = C * WA + RC * XS + 2 * RC * ES + 3 * C;
    static constexpr int NLD = (RC * WX + 255) / 256;              // raw elements per thread and chunk
};

struct AmpParams {
    const float* x;        // (B, C, T) raw input of the activation
    const float4* w;       // packed conv weights (32x32x2 fragment order)
    const float* bias;
    float* y;              // (B, C, T)
    const float* res;      // residual (may be NULL, may alias y)
    const float* alpha;    // exp(alpha) per channel
    const float* inv_beta; // 1 / (exp(beta) + 1e-9) per channel
    const float* up_taps;  // 12 kaiser-sinc taps
    const float* down_taps;
    int T, n_tiles, batch;
    int out_mode;
    float out_scale;
};

__device__ __forceinline__ f32x2a amp_snake2(f32x2a u, float al_pi, float al_lo, float hb) {   // = snake2() of small_kernels.hip (hardware cosine,
    const f32x2a ph = u * al_pi;                                                                  //   compensated phase): bit-identical to it
    f32x2a r = __builtin_elementwise_fma(u, (f32x2a)(al_pi), -ph);
    r = __builtin_elementwise_fma(u, (f32x2a)(al_lo), r);
    f32x2a c;
    c.x = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(ph.x) + r.x);
    c.y = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(ph.y) + r.y);
    return __builtin_elementwise_fma(c, (f32x2a)(-hb), u + hb);
}

template <int KS, int DIL, int C>
__global__ __launch_bounds__(256, 2) void amp_conv_kernel(const AmpParams p) {
    using G = AmpGeom<KS, DIL, C>;
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                          // [C][WA] activated window
    float* Xc = As + C * G::WA;               // [RC][XS] raw chunk
    float* Ec = Xc + G::RC * G::XS;           // [RC][ES] even 2x-rate samples
    float* Oc = Ec + G::RC * G::ES;           // [RC][ES] odd
    float* prm = Oc + G::RC * G::ES;          // [C] alpha / pi, [C] inv_beta / 2, [C] low part of alpha / pi

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);   // a clip's tiles on one XCD
    if (lid >= p.n_tiles * p.batch) return;
    const int tile = lid % p.n_tiles, b = lid / p.n_tiles;
    const int t0 = tile * G::W1;
    const int ta = t0 - G::PAD;               // global position of activated column 0
    const int T = p.T;
    const float* __restrict__ xb = p.x + (long long)b * C * T;
    // sequence ends inside the window: replicate padding of the up-sampler input and of the 2x-rate signal (alias_free_torch pads
    // with mode="replicate"), zero padding of the conv outside [0, T)
    const bool edge = ta - 6 < 0 || ta + G::WA_RAW + 6 > T;

    if (tid < C) {
        const float al = p.alpha[tid], al_pi = al * 0.318309886183790672f;
        prm[tid] = al_pi;
        prm[C + tid] = 0.5f * p.inv_beta[tid];
        prm[2 * C + tid] = fmaf(al, 0.318309886183790672f, -al_pi) + al * 1.2841276653e-8f;   // (snake_al_lo of small_kernels.hip)
    }
    f32x2a upp[6], dnp[6];   // taps as uniform register pairs (the up-sampler's gain of 2 folded in)
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        upp[q] = f32x2a{2.0f * p.up_taps[2 * q + 1], 2.0f * p.up_taps[2 * q]};
        dnp[q] = f32x2a{p.down_taps[2 * q], p.down_taps[2 * q + 1]};
    }

    // raw rows of chunk c: element e = tid + 256 i of the [RC][WX] chunk window <-> x[c RC + e / WX][clamp(ta - 6 + e % WX)]
    float raw[G::NLD];
    unsigned roff[G::NLD];   // byte offset inside a chunk's rows (the same for every chunk)
#pragma unroll
    for (int i = 0; i < G::NLD; ++i) {
        int e = tid + 256 * i;
        e = e < G::RC * G::WX ? e : G::RC * G::WX - 1;
        const int r = e / G::WX, col = e - r * G::WX;
        int t = ta - 6 + col;
        t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
        roff[i] = (unsigned)(r * T + t) * 4u;
    }
    auto request_chunk = [&](int c) {
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(xb + (long long)c * G::RC * T, (unsigned)(G::RC * T) * 4u);
#pragma unroll
        for (int i = 0; i < G::NLD; ++i) raw[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, roff[i], 0, 0));
    };
    request_chunk(0);

    for (int c = 0; c < C / G::RC; ++c) {
#pragma unroll
        for (int i = 0; i < G::NLD; ++i) {
            const int e = tid + 256 * i;
            if (e < G::RC * G::WX) {
                const int r = e / G::WX;
                Xc[r * G::XS + (e - r * G::WX)] = raw[i];
            }
        }
        __syncthreads();   // X complete (and every thread is past the previous chunk's low-pass: E / O may be overwritten)
        if (c + 1 < C / G::RC) request_chunk(c + 1);
        // 2x up-sampling + snake: position m of a row <-> h = ta - 3 + m, reads X[m .. m + 6]
        for (int it = tid; it < G::RC * G::NM; it += 256) {
            const int r = it / G::NM, m = it - r * G::NM;
            const int h = ta - 3 + m;
            int xi = m + 3;   // column of x[h] in X
            if (edge) {
                const int hc = h < 0 ? 0 : (h > T - 1 ? T - 1 : h);
                xi = hc - ta + 6;
            }
            const float* xr = Xc + r * G::XS;
            f32x2a u = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const f32x2a xp = {xr[xi + 2 - q], xr[xi + 3 - q]};
                u = __builtin_elementwise_fma(upp[q], xp, u);
            }
            f32x2a a = amp_snake2(u, prm[c * G::RC + r], prm[2 * C + c * G::RC + r], prm[C + c * G::RC + r]);
            if (edge) {   // replicate padding of the low-pass input: n < 0 -> a[0], n > 2T - 1 -> a[2T - 1]
                if (h < 0) a.y = a.x;
                if (h > T - 1) a.x = a.y;
            }
            Ec[r * G::ES + m] = a.x;
            Oc[r * G::ES + m] = a.y;
        }
        __syncthreads();
        // low-pass + decimation: activated column i <-> t = ta + i = sum_q dn[2q] odd(m = i + q) + dn[2q + 1] even(m = i + q + 1)
        for (int it = tid; it < G::RC * G::WA_RAW; it += 256) {
            const int r = it / G::WA_RAW, i = it - r * G::WA_RAW;
            f32x2a s2 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const f32x2a ap = {Oc[r * G::ES + i + q], Ec[r * G::ES + i + q + 1]};
                s2 = __builtin_elementwise_fma(dnp[q], ap, s2);
            }
            const int t = ta + i;
            As[(c * G::RC + r) * G::WA + i] = (!edge || (t >= 0 && t < T)) ? s2.x + s2.y : 0.f;
        }
    }
    __syncthreads();

    // ---- conv over the resident activated window ----
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int krow = lane >> 5;
    const int ncol = wn * (NT * 32) + (lane & 31);
    f32x16 acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    gemm32_resident<KS, G::WA, DIL, 1, NT, G::NCH>(p.w + (size_t)wm * (G::NCH * KS * 64), lane, As + krow * G::WA + ncol, acc);

    const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * C * T, (unsigned)(C * T) * 4u);
    const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc((p.res ? p.res : p.y) + (long long)b * C * T, (unsigned)(C * T) * 4u);
    const bool has_res = p.res != nullptr;
    auto off = [&](int r, int jn) -> unsigned {   // byte offset inside this batch item, or 0xFFFFFFFF (masked)
        const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * krow;
        const int t = t0 + ncol + jn * 32;
        return t < T ? (unsigned)(m * T + t) * 4u : 0xFFFFFFFFu;
    };
    float xr[16][NT];
    if (has_res) {   // the whole register tile's residual operands in one round trip
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) xr[r][jn] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, off(r, jn), 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * krow;
        const float bias = p.bias[m];
        float yo[NT];
        if (p.out_mode == OUT_ACCUM) {
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) yo[jn] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, off(r, jn), 0, 0));
        }
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            float v = acc[0][jn][r] + bias;
            if (has_res) v += xr[r][jn];
            if (p.out_mode == OUT_ACCUM) v = (yo[jn] + v) * p.out_scale;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, off(r, jn), 0, 0);
        }
    }
}

template <int KS, int DIL>
static bool launch_amp_c(const AmpParams& p, int C, hipStream_t s) {
    const int grid = (p.batch * p.n_tiles + 7) / 8 * 8;
    if (C == 32) {
        using G = AmpGeom<KS, DIL, 32>;
        const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
        if (!FV_ENSURE_DYN_LDS((amp_conv_kernel<KS, DIL, 32>), lds)) return false;
        hipLaunchKernelGGL((amp_conv_kernel<KS, DIL, 32>), dim3(grid), dim3(256), lds, s, p);
        return true;
    }
    if (C == 64) {
        using G = AmpGeom<KS, DIL, 64>;
        const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
        if (!FV_ENSURE_DYN_LDS((amp_conv_kernel<KS, DIL, 64>), lds)) return false;
        hipLaunchKernelGGL((amp_conv_kernel<KS, DIL, 64>), dim3(grid), dim3(256), lds, s, p);
        return true;
    }
    return false;
}

bool amp_conv_supported(int C, int ks, int dil) {
    return (C == 32 || C == 64) && (ks == 3 || ks == 7 || ks == 11) && (dil == 1 || dil == 3 || dil == 5);
}

bool launch_amp_conv(const ConvLayer& L, const float* x, float* y, const float* res, const float* alpha, const float* inv_beta,
                     const float* up_taps, const float* down_taps, int batch, int t, int out_mode, float out_scale, hipStream_t s) {
    AmpParams p;
    p.x = x;
    p.w = L.d_wp;
    p.bias = L.d_bias;
    p.y = y;
    p.res = res;
    p.alpha = alpha;
    p.inv_beta = inv_beta;
    p.up_taps = up_taps;
    p.down_taps = down_taps;
    p.T = t;
    p.n_tiles = (t + 127) / 128;
    p.batch = batch;
    p.out_mode = out_mode;
    p.out_scale = out_scale;
#define FV_AMP_CASE(K, D) \
    if (L.k == K && L.dil == D) return launch_amp_c<K, D>(p, L.c_in, s);
    FV_AMP_CASE(3, 1) FV_AMP_CASE(3, 3) FV_AMP_CASE(3, 5)
    FV_AMP_CASE(7, 1) FV_AMP_CASE(7, 3) FV_AMP_CASE(7, 5)
    FV_AMP_CASE(11, 1) FV_AMP_CASE(11, 3) FV_AMP_CASE(11, 5)
#undef FV_AMP_CASE
    return false;
}

}  // namespace fv
