// f16x3 precision mode, wide stages: one whole ResBlock1 iteration
//
//     y = x + c2( silu( c1( silu(x) ) ) )                (fish_vocoder/modules/generators/hifigan.py:102-107)
//
// in ONE launch for C = 128 / 64 on the split-fp16 matrix path (conv_f16x3_impl.h).  At fp16-MFMA rates the per-layer convs
// cross the roofline ridge: k = 3 is HBM-bound per layer and every tile pays an HBM round trip on either side of an ~8 us
// main loop.  Fused, the intermediate silu(c1(.)) never leaves the CU — c1's epilogue splits it into (hi, lo) fp16 planes
// straight into LDS, in the very layout c2 reads its B fragments from — so the pair moves ~3.5 tensor passes instead of
// ~5.5, c2 needs no staging and no barriers at all, and one prologue / epilogue serves two convs.
//
// Workgroup = 4 (8) waves, C rows (WM m-tiles of 32) x N_H = WN * NT * 32 intermediate columns; each wave owns 32 rows x NT
// n-tiles (C = 128: 4 x 1 waves, NT = 3, N_H = 96; C = 64: 2 x 2 waves, NT = 2, N_H = 128; C = 256: 8 x 1 waves, NT = 3).  Of the N_H columns c2 produces,
// the last KS - 1 would need intermediate columns of the next tile and are dropped: tiles advance by TT = N_H - (KS - 1)
// (8 - 10 % redundant MFMA work at k = 11, 2 % at k = 3).
// LDS (dynamic): x window (two 16-channel chunks in flight, as in the per-layer kernel) + the intermediate planes
// [plane][channel group of 8][column] x 16 B = 4 * C * N_H bytes (49 / 33 KB), the latter overlaying the former once c1 is done.
#pragma once

#include "conv_f16x3_impl.h"
#include "pair_f16x3_params.h"

namespace fv {


template <int KS, int DIL1, int WM, int WN, int NT>
struct PairF16Geom {
    static constexpr int C = WM * 32, KG = C / 8;
    static constexpr int N_H = WN * NT * 32, TT = N_H - (KS - 1);
    static constexpr int W1 = N_H + (KS - 1) * DIL1;
    static constexpr int XS_SLOTS = 2 * 4 * W1;              // two buffers x (2 planes x 2 k-halves x W1)
    static constexpr int HS_SLOTS = 2 * KG * N_H + KS;       // two planes (+ read overhang of the dropped columns)
    // the intermediate planes overlay the x window (written after a barrier once c1 has consumed it)
    static constexpr size_t LDS_BYTES = (size_t)(XS_SLOTS > HS_SLOTS ? XS_SLOTS : HS_SLOTS) * 16;
};

template <int KS, int DIL1, int WM, int WN, int NT>
__global__ __launch_bounds__(WM * WN * 64, 2) void pair_f16x3_kernel(const PairF16Params p) {
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
    constexpr int THREADS = WM * WN * 64;
    constexpr int C = WM * 32;
    constexpr int KG = C / 8;                           // channel groups of 8 (one 16-byte LDS slot per column)
    constexpr int N_H = WN * NT * 32;                   // intermediate columns per workgroup
    constexpr int TT = N_H - (KS - 1);                  // final columns per workgroup
    constexpr int H2 = (KS - 1) / 2, H1 = H2 * DIL1;
    constexpr int W1 = N_H + (KS - 1) * DIL1;           // staged x columns per channel row
    constexpr int ITEMS = 2 * W1;
    constexpr int NE = (ITEMS + THREADS - 1) / THREADS;
    constexpr int PLANE = 2 * W1;
    constexpr int HPLANE = KG * N_H;                    // slots per intermediate plane
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    h8* xs0 = reinterpret_cast<h8*>(lds_raw);           // c1 input window: [buffer][plane][k-half][column]
    h8* hs = xs0;                                       // intermediate: [plane][channel group][column] (+ read overhang); overlays xs

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tile = blockIdx.x % p.n_tiles, b = blockIdx.x / p.n_tiles;
    const int t0 = tile * TT;
    const float* __restrict__ xb = p.x + (long long)b * C * p.T;

    // ------------------------------------------------------------------ phase 1: c1 over N_H columns
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    unsigned st_off[NE];
    const int tbase = t0 - H2 - H1;                     // x position of staged column 0
    const unsigned row_b = (unsigned)p.T * 4u;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        int e = tid + i * THREADS;
        const bool in_tile = e < ITEMS;
        e = in_tile ? e : ITEMS - 1;
        const int h = e / W1;
        const int col = e - h * W1;
        const int t = tbase + col;
        const bool ok = in_tile && t >= 0 && t < p.T;
        st_off[i] = ok ? (unsigned)(8 * h * p.T + t) * 4u : 0xC0000000u;
    }
    float stage[NE][8];
    __amdgpu_buffer_rsrc_t xrs;
    auto chunk_rsrc = [&](int c) {
        xrs = uniform_rsrc(xb + (long long)c * 16 * p.T, (unsigned)((long long)(C - c * 16) * p.T * 4));
    };
    auto load_item = [&](int i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            stage[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, st_off[i] + (unsigned)r * row_b, 0, 0));
    };
    auto store_chunk = [&](h8* dst) {   // silu, split, pack (element-wise on purpose, see conv_f16x3_impl.h)
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * THREADS;
            h8 hi, lo;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float v = stage[i][r];
                v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                const _Float16 vh = (_Float16)v;
                hi[r] = vh;
                lo[r] = (_Float16)((v - (float)vh) * 2048.0f);
            }
            if (e < ITEMS) {
                dst[e] = hi;
                dst[PLANE + e] = lo;
            }
        }
    };
    const int wvoff = lane * 16;
    const int wtile_b = __builtin_amdgcn_readfirstlane(wm * p.nch16 * KS * 2048);   // this wave's m-tile in either layer
    auto load_a = [&](const __amdgpu_buffer_rsrc_t& rs, h8 (&dst)[2], int goff_b) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, wvoff + q * 1024, wtile_b + goff_b, 0);
            dst[q] = __builtin_bit_cast(h8, v);
        }
    };
    constexpr int DA = kF16WeightPrefetch;
    constexpr int RA = DA + 1;
    h8 aq[RA][2];   // (wh, wl); wh * 2^-11 derived in registers
    h8 bq[2][NT][2];                                    // B fragments of the current and the next k-block
    const int nch = p.nch16;
    // nine (NT = 3) MFMAs per k-block, product-major so that consecutive MFMAs never share an accumulator
    auto mfma_block = [&](const h8 (&a)[2], const h8 (&bf)[NT][2]) {
        const h8 a_sc = a[0] * (_Float16)(1.0f / 2048.0f);   // exact (power of two): 4 v_pk_mul_f16
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int jn = 0; jn < NT; ++jn)
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q == 2 ? a_sc : a[q], bf[jn][q == 2 ? 1 : 0], acc[jn], 0, 0, 0);
    };
    {
        const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1h, 0, 0x7fffffff, 0x00020000);
        const int b_lane = (lane >> 5) * W1 + wn * (NT * 32) + (lane & 31);
        auto load_b = [&](h8 (&dst)[NT][2], const h8* xsb, int j) {
#pragma unroll
            for (int q = 1; q >= 0; --q)
#pragma unroll
                for (int jn = NT - 1; jn >= 0; --jn) dst[jn][q] = xsb[q * PLANE + b_lane + jn * 32 + j * DIL1];
        };
        chunk_rsrc(0);
#pragma unroll
        for (int i = 0; i < NE; ++i) load_item(i);
#pragma unroll
        for (int d = 0; d < DA; ++d) load_a(w1rs, aq[d], d * 2048);
        for (int c = 0; c < nch; ++c) {
            h8* xsb = xs0 + (c & 1) * (2 * PLANE);
            store_chunk(xsb);
            __syncthreads();
            if (c + 1 < nch) {
                chunk_rsrc(c + 1);
#pragma unroll
                for (int i = 0; i < NE; ++i) load_item(i);
            }
            const int gchunk_b = __builtin_amdgcn_readfirstlane((c * KS + DA) * 2048);
            load_b(bq[0], xsb, 0);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                load_a(w1rs, aq[(j + DA) % RA], gchunk_b + j * 2048);
                if (j + 1 < KS) load_b(bq[(j + 1) & 1], xsb, j + 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_block(aq[j % RA], bq[j & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KS % RA != 0) {
                h8 t[DA][2];
#pragma unroll
                for (int d = 0; d < DA; ++d)
#pragma unroll
                    for (int q = 0; q < 2; ++q) t[d][q] = aq[(KS + d) % RA][q];
#pragma unroll
                for (int d = 0; d < DA; ++d)
#pragma unroll
                    for (int q = 0; q < 2; ++q) aq[d][q] = t[d][q];
            }
        }
    }
    __syncthreads();   // every wave is done reading the x window: the intermediate planes may overwrite it
    // c1 epilogue: bias, SiLU, zero outside [0, T) (= c2's zero padding), split, into the intermediate planes.
    // Lane (half hh, column n) holds rows 8*rq + 4*hh + (0..3) of its m-tile for rq = 0..3: four consecutive channels of
    // channel group wm*4 + rq -> one 8-byte store per (n-tile, rq, plane).
    {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const int hh = lane >> 5;
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            const int n = wn * (NT * 32) + jn * 32 + (lane & 31);
            const int pos = t0 - H2 + n;
            const bool live = pos >= 0 && pos < p.T;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                h4 vh4, vl4;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int m = wm * 32 + 8 * rq + 4 * hh + rr;
                    float v = fmaf(acc[jn][rq * 4 + rr], p.s1, p.b1[m]);
                    v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                    v = live ? v : 0.f;
                    const _Float16 vh = (_Float16)v;
                    vh4[rr] = vh;
                    vl4[rr] = (_Float16)((v - (float)vh) * 2048.0f);
                }
                h4* slot_hi = reinterpret_cast<h4*>(&hs[(wm * 4 + rq) * N_H + n]) + hh;
                h4* slot_lo = reinterpret_cast<h4*>(&hs[HPLANE + (wm * 4 + rq) * N_H + n]) + hh;
                *slot_hi = vh4;
                *slot_lo = vl4;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ phase 2: c2 straight out of the intermediate planes
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    {
        const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2h, 0, 0x7fffffff, 0x00020000);
        // output column n of this wave needs intermediate columns n + j (tap j): slot (2*kb + hh) * N_H + n + j
        const int h_lane = (lane >> 5) * N_H + wn * (NT * 32) + (lane & 31);
        auto load_b2 = [&](h8 (&dst)[NT][2], int kb, int j) {
            const h8* hk = hs + 2 * kb * N_H + h_lane + j;
#pragma unroll
            for (int q = 1; q >= 0; --q)
#pragma unroll
                for (int jn = NT - 1; jn >= 0; --jn) dst[jn][q] = hk[q * HPLANE + jn * 32];
        };
#pragma unroll
        for (int d = 0; d < DA; ++d) load_a(w2rs, aq[d], d * 2048);
        load_b2(bq[0], 0, 0);
        for (int kb = 0; kb < nch; ++kb) {
            const int gchunk_b = __builtin_amdgcn_readfirstlane((kb * KS + DA) * 2048);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                load_a(w2rs, aq[(j + DA) % RA], gchunk_b + j * 2048);
                // no barriers in this phase: the fragment prefetch runs across channel-group boundaries too
                const int cur = j & 1;   // every channel group starts in bq[0] (see the copy below)
                if (j + 1 < KS) load_b2(bq[cur ^ 1], kb, j + 1);
                else if (kb + 1 < nch) load_b2(bq[cur ^ 1], kb + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_block(aq[j % RA], bq[cur]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KS % 2 != 0) {   // odd tap count: the prefetched first fragment set of the next group sits in bq[1]
#pragma unroll
                for (int jn = 0; jn < NT; ++jn)
#pragma unroll
                    for (int q = 0; q < 2; ++q) bq[0][jn][q] = bq[1][jn][q];
            }
            if (KS % RA != 0) {
                h8 t[DA][2];
#pragma unroll
                for (int d = 0; d < DA; ++d)
#pragma unroll
                    for (int q = 0; q < 2; ++q) t[d][q] = aq[(KS + d) % RA][q];
#pragma unroll
                for (int d = 0; d < DA; ++d)
#pragma unroll
                    for (int q = 0; q < 2; ++q) aq[d][q] = t[d][q];
            }
        }
    }
    // c2 epilogue: bias, residual x (whole tile requested first), MRF accumulate
    {
        const unsigned span = (unsigned)((long long)C * p.T * 4);
        const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(xb, span);
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * C * p.T, span);
        const bool accum = p.out_mode == OUT_ACCUM;
        auto off = [&](int r, int jn) -> unsigned {
            const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int n = wn * (NT * 32) + jn * 32 + (lane & 31);
            const int t = t0 + n;
            return (n < TT && t < p.T) ? (unsigned)(m * p.T + t) * 4u : 0xFFFFFFFFu;
        };
        float rv[NT][16];
#pragma unroll
        for (int jn = 0; jn < NT; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[jn][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, off(r, jn), 0, 0));
        float bias[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = p.b2[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            float yo[16];
            if (accum) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yo[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, off(r, jn), 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fmaf(acc[jn][r], p.s2, bias[r]) + rv[jn][r];
                if (accum) v = (yo[r] + v) * p.out_scale;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, off(r, jn), 0, 0);
            }
        }
    }
}

template <int KS, int DIL1, int WM, int WN, int NT>
inline bool launch_pair_f16x3_one(PairF16Params q, int batch, hipStream_t s) {
    using G = PairF16Geom<KS, DIL1, WM, WN, NT>;
    q.n_tiles = (q.T + G::TT - 1) / G::TT;
    if (!FV_ENSURE_DYN_LDS((pair_f16x3_kernel<KS, DIL1, WM, WN, NT>), G::LDS_BYTES)) return false;
    hipLaunchKernelGGL((pair_f16x3_kernel<KS, DIL1, WM, WN, NT>), dim3(batch * q.n_tiles), dim3(WM * WN * 64), G::LDS_BYTES, s, q);
    return true;
}

// C = 128: 128 rows x 96 intermediate columns; C = 64: 64 rows x 128 columns (49 / 33 KB of LDS, two workgroups per CU);
// C = 256: 256 rows x 96 columns on eight waves (98 KB, one workgroup per CU)
template <int KS, int DIL1>
inline bool launch_pair_f16x3_cfg(const PairF16Params& p, int C, int batch, hipStream_t s) {
    if (C == 128) {
        return launch_pair_f16x3_one<KS, DIL1, 4, 1, 3>(p, batch, s);
    }
    if (C == 64) {
        return launch_pair_f16x3_one<KS, DIL1, 2, 2, 2>(p, batch, s);
    }
    if (C == 256) {   // eight waves (one m-tile each), 98 KB of intermediate planes: one workgroup per CU, two waves per SIMD
        return launch_pair_f16x3_one<KS, DIL1, 8, 1, 3>(p, batch, s);
    }
    if (C == 32) {    // a single m-tile: the four waves split the columns (A fragments reused across NT n-tiles only)
        return launch_pair_f16x3_one<KS, DIL1, 1, 4, kPairF16NtC32>(p, batch, s);
    }
    return false;
}

}  // namespace fv
