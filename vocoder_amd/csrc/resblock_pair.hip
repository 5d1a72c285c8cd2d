// Fused ResBlock (c1, c2) pair for the narrow, bandwidth-bound stages (C = 16 / 32 / 64):
//
//     y = x + c2( silu( c1( silu(x) ) ) )          (one iteration of ResBlock1.forward,
//                                                    fish_vocoder/modules/generators/hifigan.py:102-107)
//
// in ONE launch.  Per layer the reference (and the unfused path here) moves 5 tensor passes through HBM for this
// (read x, write xt, read xt, read x, write x'); fused it is ~2-3: the x window is read once (+ halo), the
// intermediate silu(c1(.)) never leaves LDS, and the residual re-read hits L2/MALL.
//
// Workgroup = 4 wavefronts, one batch item, TT final columns:
//   phase 1  stage A = silu(x[:, t0-HP : t0+W1+HP']) for ALL C channels into LDS (zero outside [0, T))
//   phase 2  c1 as implicit GEMM on fp32 MFMA over W1 = TT + KS-1 columns, epilogue bias + silu (+ zero outside [0, T):
//            c2's zero padding) written to LDS buffer Bf
//   phase 3  c2 as implicit GEMM reading Bf, epilogue bias + residual x (+ MRF accumulate) to HBM
// All K = C*KS is resident, so there are only three barriers per workgroup and every LDS address is an immediate; Bf
// overlays A (written after a barrier once c1 has consumed it).
// C = 32 / 64 use v_mfma_f32_32x32x2_f32; C = 16 uses v_mfma_f32_16x16x4_f32 (no padded rows).
#include "conv_mfma_impl.h"

namespace fv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPairMinWaves = 2;   // min waves per SIMD the register allocator must leave room for

template <int KS, int DIL, int C>
struct PairGeom {
    static constexpr int W1 = kPairCols / C < 128 ? 128 : kPairCols / C;   // c1 output columns per workgroup (256 at C=16, 128 at C=32 / 64)
    static constexpr int TT = W1 - (KS - 1);            // final output columns per workgroup
    static constexpr int H1 = (KS - 1) / 2 * DIL, H2 = (KS - 1) / 2, HP = H1 + H2;
    static constexpr int WA_RAW = W1 + (KS - 1) * DIL;  // staged x columns
    static constexpr int WB_RAW = W1 + (KS - 1);        // c1 output columns incl. the read-overhang of the last c2 tile
    // row strides == 16 (mod 32): the 16x16x4 B-fragment read puts lanes 0-15 / 16-31 on adjacent channel rows
    static constexpr int WA = (WA_RAW - 16 + 31) / 32 * 32 + 16;
    static constexpr int WB = (WB_RAW - 16 + 31) / 32 * 32 + 16;
    // the intermediate overlays the input window (one extra barrier): half the LDS, up to 8 workgroups per CU (-4 %)
    static constexpr int LDS_FLOATS = C * (WA > WB ? WA : WB);
};

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// As[r][col] = silu(x[r][t0 - HP + col]) for r < C, col < WA_RAW (0 outside [0, T): silu(0) = 0 is the conv's zero padding).
// Each wave stages C/4 whole rows: per element one buffer load (row descriptor in SGPRs, column offset in a VGPR, out-of-
// range columns come back as 0 from the hardware bounds check), silu, one ds_write — the PMC profile of the first version
// showed the kernel bound by VALU issue (1400 VALU vs 96 MFMA instructions per wave), most of it index / clamp / 64-bit
// address arithmetic around these loads.
template <int C, int WA_RAW, int WA, int HP>
__device__ __forceinline__ void stage_window(const float* __restrict__ xb, float* __restrict__ As, int wave, int lane, int t0,
                                             int T) {
    constexpr int ROWS = C / 4;                  // rows per wave
    constexpr int NI = (WA_RAW + 63) / 64;       // columns per lane
    float v[ROWS][NI];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        const int r = wave * ROWS + rr;
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(xb + (long long)r * T, (unsigned)T * 4u);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            v[rr][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (t0 - HP + lane + 64 * i) * 4, 0, 0));
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int col = lane + 64 * i;
            if (col < WA_RAW) As[(wave * ROWS + rr) * WA + col] = silu_f(v[rr][i]);
        }
}

// acc[i][jn] += sum over (cc, tap j, pair pp) of W-fragment x B-fragment, B element = bsrc[(cc*8 + 2pp)*STRIDE + jn*32 + j*DILX]
// (bsrc already carries the lane's k-half row and column).  One step (= one tap of one 8-channel sub-chunk) deep software
// pipeline on both operands, pinned with sched_barriers (see conv_mfma_impl.h).
template <int KS, int STRIDE, int DILX, int MT, int NT, int NCH>
__device__ __forceinline__ void gemm32_resident(const float4* __restrict__ w, int lane, const float* __restrict__ bsrc,
                                                f32x16 (&acc)[MT][NT]) {
    constexpr int STEPS = NCH * KS;
    // Weight fragments are requested DA steps ahead into a ring with compile-time slots (the step loop is fully unrolled): a step
    // is only 4 MT NT MFMAs (256 cycles at C = 32), less than an L2 round trip, and a single-clip launch has under one workgroup
    // per CU, so one step ahead left every step waiting for its weights (28 us for the k = 11 pair of a single clip).
    constexpr int DA = (MT == 1 ? 4 : 2) < STEPS ? (MT == 1 ? 4 : 2) : STEPS - 1;
    constexpr int RA = DA + 1;
    float4 aq[RA][MT];
    float b_cur[4][NT], b_nxt[4][NT];
    // weights by raw buffer loads: descriptor + constant byte offset in SGPRs, lane * 16 B in one VGPR (no VALU addressing)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    auto load_a1 = [&](int i, int st) __attribute__((always_inline)) {   // [(i*NCH + cc)*KS + j] == i*STEPS + st
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (i * STEPS + st) * 1024, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto load_b = [&](float (&dst)[4][NT], int st) __attribute__((always_inline)) {
        const int cc = st / KS, j = st % KS;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) dst[pp][jn] = bsrc[(cc * 8 + 2 * pp) * STRIDE + jn * 32 + j * DILX];
    };
    static_for<DA>([&](auto d_c) {
        constexpr int d = decltype(d_c)::value;
#pragma unroll
        for (int i = 0; i < MT; ++i) aq[d][i] = load_a1(i, d);
    });
    load_b(b_cur, 0);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        // this step's memory operations (MT weight loads for step st + DA, 4 NT LDS fragment reads for step st + 1) are requested
        // BETWEEN its MFMAs: an in-order wave hides a memory instruction's issue time only under an MFMA that is already
        // executing (conv_mfma_impl.h)
        constexpr int NM = 4 * MT * NT, NLDX = MT + 4 * NT;
        const int cc_n = (st + 1) / KS, j_n = (st + 1) % KS;
        const int SC = st % RA, SN = (st + DA) % RA;   // constants once the loop is unrolled
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int pp = m / (MT * NT), i = (m / NT) % MT, jn = m % NT;
            const float av = pp == 0 ? aq[SC][i].x : pp == 1 ? aq[SC][i].y : pp == 2 ? aq[SC][i].z : aq[SC][i].w;
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_cur[pp][jn], acc[i][jn], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NLDX; ++k) {
                if (k * NM / NLDX == m) {
                    if (k < MT) {
                        if (st + DA < STEPS) aq[SN][k] = load_a1(k, st + DA);
                    } else if (st + 1 < STEPS) {
                        const int pp2 = (k - MT) / NT, jn2 = (k - MT) % NT;
                        b_nxt[pp2][jn2] = bsrc[(cc_n * 8 + 2 * pp2) * STRIDE + jn2 * 32 + j_n * DILX];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < STEPS) {
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) b_cur[pp][jn] = b_nxt[pp][jn];
        }
    }
}

// 16-channel variant on v_mfma_f32_16x16x4_f32: B element = bsrc[(4q)*STRIDE + jn*16 + j*DILX]
template <int KS, int STRIDE, int DILX, int NT>
__device__ __forceinline__ void gemm16_resident(const float4* __restrict__ w, int lane, const float* __restrict__ bsrc,
                                                f32x4 (&acc)[NT]) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    auto load_w = [&](int j) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * 1024, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // all KS taps' weights (one float4 per lane and tap, <= 44 registers) are requested up front: a tap is 4 NT MFMAs of 32 cycles,
    // less than an L2 round trip, and a single-clip launch has nothing else resident to cover it
    float4 aw[KS];
    float b_cur[4][NT], b_nxt[4][NT];
    auto load_b = [&](float (&dst)[4][NT], int j) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) dst[q][jn] = bsrc[(4 * q) * STRIDE + jn * 16 + j * DILX];
    };
    static_for<KS>([&](auto j_c) { aw[decltype(j_c)::value] = load_w(decltype(j_c)::value); });
    load_b(b_cur, 0);
    static_for<KS>([&](auto j_c) __attribute__((always_inline)) {
        constexpr int j = decltype(j_c)::value;
        if (j + 1 < KS) load_b(b_nxt, j + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float av = q == 0 ? aw[j].x : q == 1 ? aw[j].y : q == 2 ? aw[j].z : aw[j].w;
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b_cur[q][jn], acc[jn], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < KS) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) b_cur[q][jn] = b_nxt[q][jn];
        }
    });
}

// ---------------------------------------------------------------------------------------------------------------
// C = 32 (MT = 1) and C = 64 (MT = 2): 32x32x2 MFMA.  Each wave owns NT n-tiles of 32 columns and all m-tiles.
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int DIL, int C>
__global__ __launch_bounds__(256, kPairMinWaves) void resblock_pair32_kernel(const PairParams p) {
    using G = PairGeom<KS, DIL, C>;
    constexpr int MT = C / 32;
    constexpr int NT = G::W1 / 32 / 4;   // n-tiles per wave
    constexpr int NCH = C / 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;
    float* Bs = lds;   // overlays As once every wave has finished c1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x % p.n_tiles, b = blockIdx.x / p.n_tiles;
    const int t0 = tile * G::TT;
    const float* __restrict__ xb = p.x + (long long)b * C * p.T;

    // phase 1: A = silu(x) window.  Loads are unconditional on clamped addresses and issued in batches of 8 so that
    // their latencies overlap (a guarded load per element compiles to a branch + vmcnt(0) each).
    stage_window<C, G::WA_RAW, G::WA, G::HP>(xb, As, wave, lane, t0, p.T);
    __syncthreads();

    const int ncol = wave * (NT * 32) + (lane & 31);
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
        gemm32_resident<KS, G::WA, DIL, MT, NT, NCH>(p.w1, lane, As + krow * G::WA + ncol, acc);
        __syncthreads();   // every wave is done reading the window before the intermediate overwrites it
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * krow;
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
        gemm32_resident<KS, G::WB, 1, MT, NT, NCH>(p.w2, lane, Bs + krow * G::WB + ncol, acc);
        const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(xb, (unsigned)(C * p.T) * 4u);
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * C * p.T, (unsigned)(C * p.T) * 4u);
        // residual operands of the whole register tile first, then combine and store: one HBM round trip instead of one per
        // accumulator row (the loads of row r + 1 could not start before the stores of row r were issued)
        auto off = [&](int i, int r, int jn) -> unsigned {   // byte offset inside this batch item, or 0xFFFFFFFF (masked)
            const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * krow;
            const int n = ncol + jn * 32;
            const int t = t0 + n;
            return (n < G::TT && t < p.T) ? (unsigned)(m * p.T + t) * 4u : 0xFFFFFFFFu;
        };
        float xr[MT][16][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int jn = 0; jn < NT; ++jn)
                    xr[i][r][jn] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, off(i, r, jn), 0, 0));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * krow;
                const float bias = p.b2[m];
                float yo[NT];
                if (p.out_mode == OUT_ACCUM) {
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn) yo[jn] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, off(i, r, jn), 0, 0));
                }
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) {
                    float v = acc[i][jn][r] + bias + xr[i][r][jn];
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
    constexpr int NT = G::W1 / 16 / 4;   // 8 n-tiles of 16 columns per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;
    float* Bs = lds;   // overlays As once every wave has finished c1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x % p.n_tiles, b = blockIdx.x / p.n_tiles;
    const int t0 = tile * G::TT;
    const float* __restrict__ xb = p.x + (long long)b * C * p.T;

    stage_window<C, G::WA_RAW, G::WA, G::HP>(xb, As, wave, lane, t0, p.T);
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
        const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(xb, (unsigned)(C * p.T) * 4u);
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * C * p.T, (unsigned)(C * p.T) * 4u);
        auto off = [&](int r, int jn) -> unsigned {   // byte offset inside this batch item, 0xFFFFFFFF = masked
            const int n = ncol + jn * 16;
            const int t = t0 + n;
            return (n < G::TT && t < p.T) ? (unsigned)((4 * krow + r) * p.T + t) * 4u : 0xFFFFFFFFu;
        };
        float xr[4][NT];   // the whole tile's residual operands in one round trip (see resblock_pair32_kernel)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) xr[r][jn] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, off(r, jn), 0, 0));
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
                float v = acc[jn][r] + bias + xr[r][jn];
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
        hipLaunchKernelGGL((resblock_pair16_kernel<KS, DIL>), dim3(batch * q.n_tiles), dim3(256), lds, s, q);
        return true;
    }
    if (C == 32) {
        using G = PairGeom<KS, DIL, 32>;
        PairParams q = p;
        q.n_tiles = (p.T + G::TT - 1) / G::TT;
        const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
        if (!FV_ENSURE_DYN_LDS((resblock_pair32_kernel<KS, DIL, 32>), lds)) return false;
        hipLaunchKernelGGL((resblock_pair32_kernel<KS, DIL, 32>), dim3(batch * q.n_tiles), dim3(256), lds, s, q);
        return true;
    }
    if (C == 64) {
        using G = PairGeom<KS, DIL, 64>;
        PairParams q = p;
        q.n_tiles = (p.T + G::TT - 1) / G::TT;
        const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
        if (!FV_ENSURE_DYN_LDS((resblock_pair32_kernel<KS, DIL, 64>), lds)) return false;
        hipLaunchKernelGGL((resblock_pair32_kernel<KS, DIL, 64>), dim3(batch * q.n_tiles), dim3(256), lds, s, q);
        return true;
    }
    return false;
}

// C = 64 fuses only at k = 3 (measured per stage: k = 3 -20 %, k = 7 +42 %, k = 11 worse still: with two m-tiles per wave and a
// single n-tile the resident-K kernel re-fetches every weight fragment per wave, which only the short kernel can afford)
bool pair_supported(int C, int ks, int dil) {
    if (C != 16 && C != 32 && !(C == 64 && ks == 3)) return false;
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
