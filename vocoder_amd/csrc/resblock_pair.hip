// Fused ResBlock (c1, c2) pair for the narrow stages (C = 16 / 32 at every kernel size) and the k = 3 branch of C = 64 / 128:
//
//     y = x + c2( silu( c1( silu(x) ) ) )          (one iteration of ResBlock1.forward,
//                                                    fish_vocoder/modules/generators/hifigan.py:102-107)
//
// in ONE launch.  Per layer the reference (and the unfused path here) moves 5 tensor passes through HBM for this
// (read x, write xt, read xt, read x, write x'); fused it is 2: the x window is read once (its halo from the L2 the clip's
// neighbouring tiles share), the intermediate silu(c1(.)) never leaves LDS, and the residual comes from an LDS copy of the raw
// tile (round 3; PMC: 181 MB per launch = the algorithmic 2 C T 4 B.  C = 128 re-reads it from HBM: no room next to the window).
//
// Workgroup = 4 wavefronts, one batch item, TT final columns:
//   phase 1  stage A = silu(x[:, t0-HP : t0+W1+HP']) for ALL C channels into LDS (zero outside [0, T))
//   phase 2  c1 as implicit GEMM on fp32 MFMA over W1 = TT + KS-1 columns, epilogue bias + silu (+ zero outside [0, T):
//            c2's zero padding) written to LDS buffer Bf
//   phase 3  c2 as implicit GEMM reading Bf, epilogue bias + residual x (+ MRF accumulate) to HBM
// All K = C*KS is resident, so there are only three barriers per workgroup and every LDS address is an immediate; Bf
// overlays A (written after a barrier once c1 has consumed it).
// C >= 32 use v_mfma_f32_32x32x2_f32 (C >= 64: waves stacked along M, see resblock_pair32_kernel); C = 16 uses
// v_mfma_f32_16x16x4_f32 (no padded rows).
#include "pair_common.h"

namespace fv {

typedef f32x4p f32x4;

constexpr int kPairMinWaves = 2;   // min waves per SIMD the register allocator must leave room for

template <int KS, int DIL, int C>
struct PairGeom {
    static constexpr int W1 = kPairCols / C < 128 ? 128 : kPairCols / C;   // c1 output columns per workgroup (128 at every width since round 3)
    static constexpr int TT = W1 - (KS - 1);            // final output columns per workgroup
    static constexpr int H1 = (KS - 1) / 2 * DIL, H2 = (KS - 1) / 2, HP = H1 + H2;
    static constexpr int WA_RAW = W1 + (KS - 1) * DIL;  // staged x columns
    static constexpr int WB_RAW = W1 + (KS - 1);        // c1 output columns incl. the read-overhang of the last c2 tile
    // row strides == 16 (mod 32): the 16x16x4 B-fragment read puts lanes 0-15 / 16-31 on adjacent channel rows
    static constexpr int WA = (WA_RAW - 16 + 31) / 32 * 32 + 16;
    static constexpr int WB = (WB_RAW - 16 + 31) / 32 * 32 + 16;
    // the intermediate overlays the input window (one extra barrier): half the LDS, up to 8 workgroups per CU (-4 %)
    static constexpr int AB_FLOATS = C * (WA > WB ? WA : WB);
    static constexpr int XS = TT;                       // row stride of the raw centre tile kept for the residual
    static constexpr int LDS_FLOATS = AB_FLOATS + C * XS;
};

// ---------------------------------------------------------------------------------------------------------------
// C = 32 (MT = 1) and C = 64 (MT = 2): 32x32x2 MFMA, each wave owns NT n-tiles of 32 columns and all m-tiles (WM = 1).
// C = 128 (k = 3 only, round 3): WM = 4 — each wave owns ONE m-tile (its own quarter of the weights: no fragment is fetched by
// two waves) and all four n-tiles; the raw tile for the residual does not fit next to the 74 KB window (XRES = false: the
// epilogue reads x from HBM again, L2-warm), two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int DIL, int C, int WM = 1, bool XRES = true>
__global__ __launch_bounds__(256, kPairMinWaves) void resblock_pair32_kernel(const PairParams p) {
    using G = PairGeom<KS, DIL, C>;
    constexpr int WN = 4 / WM;
    constexpr int MT = C / 32 / WM;          // m-tiles per wave
    constexpr int NT = G::W1 / 32 / WN;      // n-tiles per wave
    constexpr int NCH = C / 8;
    constexpr int STEPS64 = NCH * KS * 64;   // float4s of packed weights per m-tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;
    float* Bs = lds;   // overlays As once every wave has finished c1
    float* Xr = lds + G::AB_FLOATS;   // raw x[:, t0 : t0 + TT): the residual operand (XRES)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on it).  Neighbouring tiles of a clip share their
    // halo columns (up to 30 per side on 118 - 246 produced): handing them to the SAME XCD lets its L2 serve those lines once —
    // dealt round-robin, each XCD fetched them from HBM for itself.  Logical id = (b % 8) * (grid / 8) + b / 8 (grid rounded up
    // to a multiple of 8 by the host; surplus workgroups leave).
    const int lid = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (lid >= p.n_tiles * p.batch) return;
    const int tile = lid % p.n_tiles, b = lid / p.n_tiles;
    const int t0 = tile * G::TT;
    const float* __restrict__ xb = p.x + (long long)b * C * p.T;

    // phase 1: A = silu(x) window.  Loads are unconditional on clamped addresses and issued in batches of 8 so that
    // their latencies overlap (a guarded load per element compiles to a branch + vmcnt(0) each).
    if constexpr (XRES) stage_window<C, G::WA_RAW, G::WA, G::HP, G::TT, G::XS>(xb, As, wave, lane, t0, p.T, Xr);
    else stage_window<C, G::WA_RAW, G::WA, G::HP>(xb, As, wave, lane, t0, p.T);
    __syncthreads();

    const int wm = wave / WN, wn = wave % WN;
    const int mrow0 = wm * MT * 32;          // first output row of this wave
    const int ncol = wn * (NT * 32) + (lane & 31);
    const int krow = lane >> 5;

    // phase 2: c1
    {
        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        gemm32_resident<KS, G::WA, DIL, MT, NT, NCH>(p.w1 + (size_t)wm * MT * STEPS64, lane, As + krow * G::WA + ncol, acc);
        __syncthreads();   // every wave is done reading the window before the intermediate overwrites it
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * krow;
                const float bias = p.b1[m];
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) {
                    const int n = ncol + jn * 32;
                    const int pos = t0 - G::H2 + n;
                    const float v = (pos >= 0 && pos < p.T) ? silu_f(acc[i][jn][r] + bias) : 0.f;
                    Bs[m * G::WB + n] = v;
                }
            }
    }
    __syncthreads();

    // phase 3: c2 + residual
    {
        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        gemm32_resident<KS, G::WB, 1, MT, NT, NCH>(p.w2 + (size_t)wm * MT * STEPS64, lane, Bs + krow * G::WB + ncol, acc);
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * C * p.T, (unsigned)(C * p.T) * 4u);
        const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(xb, (unsigned)(C * p.T) * 4u);
        auto off = [&](int i, int r, int jn) -> unsigned {   // byte offset inside this batch item, or 0xFFFFFFFF (masked)
            const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * krow;
            const int n = ncol + jn * 32;
            const int t = t0 + n;
            return (n < G::TT && t < p.T) ? (unsigned)(m * p.T + t) * 4u : 0xFFFFFFFFu;
        };
        float xg[XRES ? 1 : MT][XRES ? 1 : 16][XRES ? 1 : NT];
        if constexpr (!XRES) {   // residual operands of the whole register tile in one round trip
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn)
                        xg[i][r][jn] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, off(i, r, jn), 0, 0));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * krow;
                const float bias = p.b2[m];
                float yo[NT];
                if (p.out_mode == OUT_ACCUM) {
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn) yo[jn] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, off(i, r, jn), 0, 0));
                }
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) {
                    const int n = ncol + jn * 32;
                    // residual from the raw tile in LDS (columns past TT belong to the next tile: masked by off(), any finite address)
                    float res;
                    if constexpr (XRES) res = Xr[m * G::XS + (n < G::TT ? n : 0)];
                    else res = xg[i][r][jn];
                    float v = acc[i][jn][r] + bias + res;
                    if (p.out_mode == OUT_ACCUM) v = (yo[jn] + v) * p.out_scale;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, off(i, r, jn), 0, 0);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C = 16: 16x16x4 MFMA (A: row = lane & 15, k = lane >> 4; C/D: col = lane & 15, row = 4 * (lane >> 4) + reg).
// Weights packed as [tap][lane] float4 = the four channel quads of that tap.
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int DIL>
__global__ __launch_bounds__(256, kPairMinWaves) void resblock_pair16_kernel(const PairParams p) {
    constexpr int C = 16;
    using G = PairGeom<KS, DIL, C>;
    constexpr int NT = G::W1 / 16 / 4;   // n-tiles of 16 columns per wave: 2 (W1 = 128 columns over four waves)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;
    float* Bs = lds;   // overlays As once every wave has finished c1
    float* Xr = lds + G::AB_FLOATS;   // raw x[:, t0 : t0 + TT): the residual operand (see resblock_pair32_kernel)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);   // a clip's tiles on one XCD
    if (lid >= p.n_tiles * p.batch) return;
    const int tile = lid % p.n_tiles, b = lid / p.n_tiles;
    const int t0 = tile * G::TT;
    const float* __restrict__ xb = p.x + (long long)b * C * p.T;

    stage_window<C, G::WA_RAW, G::WA, G::HP, G::TT, G::XS>(xb, As, wave, lane, t0, p.T, Xr);
    __syncthreads();

    const int ncol = wave * (NT * 16) + (lane & 15);
    const int krow = lane >> 4;

    {
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm16_resident<KS, G::WA, DIL, NT>(p.w1, lane, As + krow * G::WA + ncol, acc);
        __syncthreads();   // (see resblock_pair32_kernel)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * krow + r;
            const float bias = p.b1[m];
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) {
                const int n = ncol + jn * 16;
                const int pos = t0 - G::H2 + n;
                Bs[m * G::WB + n] = (pos >= 0 && pos < p.T) ? silu_f(acc[jn][r] + bias) : 0.f;
            }
        }
    }
    __syncthreads();

    {
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm16_resident<KS, G::WB, 1, NT>(p.w2, lane, Bs + krow * G::WB + ncol, acc);
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * C * p.T, (unsigned)(C * p.T) * 4u);
        auto off = [&](int r, int jn) -> unsigned {   // byte offset inside this batch item, 0xFFFFFFFF = masked
            const int n = ncol + jn * 16;
            const int t = t0 + n;
            return (n < G::TT && t < p.T) ? (unsigned)((4 * krow + r) * p.T + t) * 4u : 0xFFFFFFFFu;
        };
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bias = p.b2[4 * krow + r];
            float yo[NT];
            if (p.out_mode == OUT_ACCUM) {
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) yo[jn] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, off(r, jn), 0, 0));
            }
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) {
                const int n = ncol + jn * 16;
                float v = acc[jn][r] + bias + Xr[(4 * krow + r) * G::XS + (n < G::TT ? n : 0)];   // residual from the raw tile in LDS
                if (p.out_mode == OUT_ACCUM) v = (yo[jn] + v) * p.out_scale;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, off(r, jn), 0, 0);
            }
        }
    }
}

template <int KS, int DIL>
static bool launch_pair_c(const PairParams& p, int C, int batch, hipStream_t s) {
    if (C == 16) {
        using G = PairGeom<KS, DIL, 16>;
        PairParams q = p;
        q.n_tiles = (p.T + G::TT - 1) / G::TT;
        const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
        if (!FV_ENSURE_DYN_LDS((resblock_pair16_kernel<KS, DIL>), lds)) return false;
        q.batch = batch;
        hipLaunchKernelGGL((resblock_pair16_kernel<KS, DIL>), dim3((batch * q.n_tiles + 7) / 8 * 8), dim3(256), lds, s, q);
        return true;
    }
    if (C == 32) {
        using G = PairGeom<KS, DIL, 32>;
        PairParams q = p;
        q.n_tiles = (p.T + G::TT - 1) / G::TT;
        const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
        if (!FV_ENSURE_DYN_LDS((resblock_pair32_kernel<KS, DIL, 32>), lds)) return false;
        q.batch = batch;
        hipLaunchKernelGGL((resblock_pair32_kernel<KS, DIL, 32>), dim3((batch * q.n_tiles + 7) / 8 * 8), dim3(256), lds, s, q);
        return true;
    }
    if constexpr (KS == 3) {
        if (C == 128) {
            using G = PairGeom<KS, DIL, 128>;
            PairParams q = p;
            q.n_tiles = (p.T + G::TT - 1) / G::TT;
            q.batch = batch;
            const size_t lds = (size_t)G::AB_FLOATS * sizeof(float);
            if (!FV_ENSURE_DYN_LDS((resblock_pair32_kernel<KS, DIL, 128, 4, false>), lds)) return false;
            hipLaunchKernelGGL((resblock_pair32_kernel<KS, DIL, 128, 4, false>), dim3((batch * q.n_tiles + 7) / 8 * 8), dim3(256), lds, s, q);
            return true;
        }
    }
    if (C == 64) {
        using G = PairGeom<KS, DIL, 64>;
        PairParams q = p;
        q.n_tiles = (p.T + G::TT - 1) / G::TT;
        const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
        // waves along M (one m-tile and two n-tiles per wave: no weight fragment is fetched by two waves): in the B = 32 step 14.94 ->
        // 14.86 ms against the all-m-tiles-per-wave layout of round 2 (stand-alone 5 % slower; profiles/LOG.md R3.10)
        if (!FV_ENSURE_DYN_LDS((resblock_pair32_kernel<KS, DIL, 64, 2, true>), lds)) return false;
        q.batch = batch;
        hipLaunchKernelGGL((resblock_pair32_kernel<KS, DIL, 64, 2, true>), dim3((batch * q.n_tiles + 7) / 8 * 8), dim3(256), lds, s, q);
        return true;
    }
    return false;
}

// C = 64 fuses only at k = 3 (measured per stage: k = 3 -20 %, k = 7 +42 %, k = 11 worse still: with two m-tiles per wave and a
// single n-tile the resident-K kernel re-fetches every weight fragment per wave, which only the short kernel can afford)
// C = 128 at k = 3 (round 3): waves along M; the short kernel's prologue / epilogue share halves, as at C = 64
bool pair_supported(int C, int ks, int dil) {
    if (C != 16 && C != 32 && !((C == 64 || C == 128) && ks == 3)) return false;
    if (ks != 3 && ks != 7 && ks != 11) return false;
    return dil == 1 || dil == 3 || dil == 5;
}

bool launch_resblock_pair(const PairParams& p, int C, int ks, int dil, int batch, hipStream_t s) {
    if (!pair_supported(C, ks, dil)) return false;
#define FV_PAIR_CASE(K, D) \
    if (ks == K && dil == D) return launch_pair_c<K, D>(p, C, batch, s);
    FV_PAIR_CASE(3, 1) FV_PAIR_CASE(3, 3) FV_PAIR_CASE(3, 5)
    FV_PAIR_CASE(7, 1) FV_PAIR_CASE(7, 3) FV_PAIR_CASE(7, 5)
    FV_PAIR_CASE(11, 1) FV_PAIR_CASE(11, 3) FV_PAIR_CASE(11, 5)
#undef FV_PAIR_CASE
    return false;
}

}  // namespace fv
