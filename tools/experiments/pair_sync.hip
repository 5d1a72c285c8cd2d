// EXPERIMENT (round 3), NOT part of libfishvoc_hip.so — kept next to its measurement (profiles/LOG.md).  Bit-identical to
// resblock_pair.hip on every shape, but SLOWER at HiFiGAN-V1 B = 32: C = 32: k = 3 0.124 vs 0.110 ms, k = 7 0.214 vs 0.177,
// k = 11 0.301 vs 0.270; C = 64, k = 3: 0.185 vs 0.170 (tools/experiments/probe_pair_sync.py).  With one workgroup per CU every
// latency the tile kernel hides behind its 4 - 8 co-resident workgroups is exposed to all eight waves at once — the first weight
// fragments of each MFMA phase, the bias loads of each epilogue, four barriers per tile — and 11.25 tiles per CU quantise to 12
// rounds; the interference the design avoids (VALU instructions of a partner wave interrupting the MFMA stream) costs the tile
// kernel less than the micro-benchmark's worst case (~6 cycles per instruction by its measured rates, not 12 - 20).
// To rebuild: add to csrc/Makefile, restore PairParams::batch, pair_sync_supported / launch_pair_sync and the FV_PAIR_SYNC knob.
// Fused ResBlock (c1, c2) pair, phase-synchronous persistent form (round 3):
//
//     y = x + c2( silu( c1( silu(x) ) ) )          (fish_vocoder/modules/generators/hifigan.py:102-107)
//
// Why a second pair kernel.  resblock_pair.hip runs one tile per 4-wave workgroup with 4 - 8 workgroups per CU, each in another
// phase: one workgroup's VALU phases (SiLU while staging, the c1 epilogue's bias + SiLU, the final epilogue) share a SIMD with
// another's MFMA phases.  tools/ubench/mfma_mix.hip measured what that costs on gfx950: a VALU instruction issued by a PARTNER
// wave while this wave streams fp32 MFMAs takes 12 - 20 cycles out of the matrix stream (plain, packed and transcendental
// alike: each one interrupts it), against 3.6 cycles (plain) / 10.8 (transcendental) when the SIMD runs VALU work only.  The
// narrow stages spend ~16 VALU instructions per element and pair — for C = 16, k = 3 as many interrupted cycles as MFMA cycles,
// which is the 0.44 - 0.6 of the sustained MFMA rate those kernels measured.
// Here ONE 8-wave workgroup owns a CU and walks a list of tiles; all its waves are in the same phase at any time:
//
//   P1  VALU   next tile's x window (already in registers, requested during the previous MFMA phase) -> SiLU -> LDS  A
//   P2  MFMA   c1 over W1 columns, K = C * k resident in LDS; the loads of the following tile's window are issued first
//   P3  VALU   bias + SiLU (+ zero outside [0, T): c2's padding) -> LDS  B (overlays A)
//   P4  MFMA   c2; the residual operands of the register tile are requested first
//   P5  VALU   + bias + residual (+ MRF accumulate) -> HBM
//
// so VALU instructions cost their stand-alone issue time and every MFMA phase has the matrix pipe to itself (two waves per
// SIMD, one dependent accumulator chain each: a 32x32x2 fp32 MFMA's 64-cycle latency equals its issue time).  HBM latency is
// hidden by the register prefetch one tile ahead, not by occupancy.  Tiles of one clip are neighbours on one XCD (their halos
// share cache lines); the raw x tile is read once for the window and once more (L2-warm) for the residual.
#include "pair_common.h"

namespace fv {

template <int KS, int DIL, int C, int W1>
struct SyncGeom {
    static constexpr int NW = 8;                                   // waves per workgroup (2 per SIMD)
    static constexpr int NWM = C / 32;                             // waves along M: one 32-row m-tile each
    static constexpr int NWN = NW / NWM;
    static constexpr int NT = W1 / 32 / NWN;                       // n-tiles per wave
    static constexpr int NCH = C / 8;
    static constexpr int TT = W1 - (KS - 1);                       // final output columns per tile
    static constexpr int H1 = (KS - 1) / 2 * DIL, H2 = (KS - 1) / 2, HP = H1 + H2;
    static constexpr int WA_RAW = W1 + (KS - 1) * DIL;             // staged x columns
    static constexpr int WB_RAW = W1 + (KS - 1);                   // c1 output columns incl. the read overhang of the last c2 tile
    static constexpr int WA = (WA_RAW - 16 + 31) / 32 * 32 + 16;   // row strides == 16 (mod 32), as in resblock_pair.hip
    static constexpr int WB = (WB_RAW - 16 + 31) / 32 * 32 + 16;
    static constexpr int LDS_FLOATS = C * (WA > WB ? WA : WB);
    static constexpr int ROWS = C / NW;                            // window rows staged per wave
    static constexpr int NI = (WA_RAW + 63) / 64;                  // window columns per lane
    static_assert(NW % NWM == 0 && (W1 / 32) % NWN == 0 && C % NW == 0, "tile does not split over 8 waves");
};

template <int KS, int DIL, int C, int W1>
__global__ __launch_bounds__(512, 2) void pair_sync_kernel(const PairParams p) {
    using G = SyncGeom<KS, DIL, C, W1>;
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;
    float* Bs = lds;   // overlays As once every wave has finished c1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::NWN, wn = wave % G::NWN;
    const int krow = lane >> 5;
    const int ncol = wn * (NT * 32) + (lane & 31);

    // ---- tile list: (item, tile) pairs, neighbours in time consecutive; XCD x owns a contiguous range, its workgroups interleave ----
    const int total = p.n_tiles * p.batch;
    const int nwg = gridDim.x;                        // a multiple of 8 (host)
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8, per_xcd = nwg / 8;
    const int lo = (int)((long long)xcd * total / 8), hi = (int)((long long)(xcd + 1) * total / 8);
    int tile_id = lo + slot;
    if (tile_id >= hi) return;

    // x window of tile `id` -> registers (raw buffer loads: one descriptor per row, columns outside [0, T) come back as 0)
    float win[G::ROWS][G::NI];
    auto request_window = [&](int id) {
        const int b = id / p.n_tiles, t0 = (id - b * p.n_tiles) * G::TT;
        const float* __restrict__ xb = p.x + (long long)b * C * p.T;
#pragma unroll
        for (int rr = 0; rr < G::ROWS; ++rr) {
            const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(xb + (long long)(wave * G::ROWS + rr) * p.T, (unsigned)p.T * 4u);
#pragma unroll
            for (int i = 0; i < G::NI; ++i)
                win[rr][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (t0 - G::HP + lane + 64 * i) * 4, 0, 0));
        }
    };
    request_window(tile_id);

    const float4* __restrict__ w1 = p.w1 + (size_t)wm * (G::NCH * KS * 64);
    const float4* __restrict__ w2 = p.w2 + (size_t)wm * (G::NCH * KS * 64);

    for (;;) {
        const int b = tile_id / p.n_tiles, t0 = (tile_id - b * p.n_tiles) * G::TT;
        const int next_id = tile_id + per_xcd;
        const bool more = next_id < hi;

        // ---- P1: A = silu(x window) ----
#pragma unroll
        for (int rr = 0; rr < G::ROWS; ++rr)
#pragma unroll
            for (int i = 0; i < G::NI; ++i) {
                const int col = lane + 64 * i;
                if (col < G::WA_RAW) As[(wave * G::ROWS + rr) * G::WA + col] = silu_f(win[rr][i]);
            }
        __syncthreads();
        if (more) request_window(next_id);   // travels during both MFMA phases

        // ---- P2: c1 ----
        f32x16 acc[1][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
        gemm32_resident<KS, G::WA, DIL, 1, NT, G::NCH>(w1, lane, As + krow * G::WA + ncol, acc);
        __syncthreads();   // every wave is done reading the window before the intermediate overwrites it

        // ---- P3: B = silu(c1 + bias), zero outside [0, T) ----
        {
            bool inside[NT];
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) {
                const int pos = t0 - G::H2 + ncol + jn * 32;
                inside[jn] = pos >= 0 && pos < p.T;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * krow;
                const float bias = p.b1[m];
#pragma unroll
                for (int jn = 0; jn < NT; ++jn)
                    Bs[m * G::WB + ncol + jn * 32] = inside[jn] ? silu_f(acc[0][jn][r] + bias) : 0.f;
            }
        }
        __syncthreads();

        // ---- P4: c2 (residual / accumulate operands requested first) ----
        const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(p.x + (long long)b * C * p.T, (unsigned)(C * p.T) * 4u);
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * C * p.T, (unsigned)(C * p.T) * 4u);
        unsigned coff[NT];   // byte offset of (row 4 * krow, this lane's column) or masked
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            const int n = ncol + jn * 32, t = t0 + n;
            coff[jn] = (n < G::TT && t < p.T) ? (unsigned)((wm * 32 + 4 * krow) * p.T + t) * 4u : 0xFFFFFFFFu;
        }
        const int rowstep = p.T * 4;   // bytes per row
        float xr[16][NT];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int so = __builtin_amdgcn_readfirstlane(((r & 3) + 8 * (r >> 2)) * rowstep);
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) xr[r][jn] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, coff[jn], so, 0));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
        gemm32_resident<KS, G::WB, 1, 1, NT, G::NCH>(w2, lane, Bs + krow * G::WB + ncol, acc);
        __syncthreads();   // the next tile's P1 overwrites the intermediate

        // ---- P5: + bias + residual -> HBM ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (r & 3) + 8 * (r >> 2);
            const float bias = p.b2[wm * 32 + mr + 4 * krow];
            const int so = __builtin_amdgcn_readfirstlane(mr * rowstep);
            float yo[NT];
            if (p.out_mode == OUT_ACCUM) {   // (single-stream / chain mode only: the branch streams keep their own outputs)
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) yo[jn] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, coff[jn], so, 0));
            }
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) {
                float v = acc[0][jn][r] + bias + xr[r][jn];
                if (p.out_mode == OUT_ACCUM) v = (yo[jn] + v) * p.out_scale;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, coff[jn], so, 0);
            }
        }
        if (!more) break;
        tile_id = next_id;
    }
}

template <int KS, int DIL, int C, int W1>
static bool launch_sync(const PairParams& p, int batch, hipStream_t s) {
    using G = SyncGeom<KS, DIL, C, W1>;
    PairParams q = p;
    q.n_tiles = (p.T + G::TT - 1) / G::TT;
    q.batch = batch;
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
    if (!FV_ENSURE_DYN_LDS((pair_sync_kernel<KS, DIL, C, W1>), lds)) return false;
    const int grid = num_cus() / 8 * 8;   // one workgroup per CU
    hipLaunchKernelGGL((pair_sync_kernel<KS, DIL, C, W1>), dim3(grid), dim3(512), lds, s, q);
    return true;
}

// the persistent kernel needs a few tiles per CU to amortise its pipeline fill; smaller launches keep resblock_pair.hip
bool pair_sync_supported(int C, int ks, int dil, int batch, int t) {
    if (C != 32 && C != 64) return false;
    if (ks != 3 && ks != 7 && ks != 11) return false;
    if (dil != 1 && dil != 3 && dil != 5) return false;
    if (num_cus() < 8) return false;
    const long long tiles = (long long)batch * ((t + 256 - ks) / (256 - (ks - 1)));
    return tiles >= 3LL * (num_cus() / 8 * 8);
}

bool launch_pair_sync(const PairParams& p, int C, int ks, int dil, int batch, hipStream_t s) {
#define FV_SYNC_CASE(K, D)                                                  \
    if (ks == K && dil == D) {                                              \
        if (C == 32) return launch_sync<K, D, 32, 256>(p, batch, s);        \
        if (C == 64) return launch_sync<K, D, 64, 256>(p, batch, s);        \
        return false;                                                       \
    }
    FV_SYNC_CASE(3, 1) FV_SYNC_CASE(3, 3) FV_SYNC_CASE(3, 5)
    FV_SYNC_CASE(7, 1) FV_SYNC_CASE(7, 3) FV_SYNC_CASE(7, 5)
    FV_SYNC_CASE(11, 1) FV_SYNC_CASE(11, 3) FV_SYNC_CASE(11, 5)
#undef FV_SYNC_CASE
    return false;
}

}  // namespace fv
