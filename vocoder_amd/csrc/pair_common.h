// Device helpers shared by the fused (c1, c2) pair kernels (resblock_pair.hip: one tile per 4-wave workgroup;
// pair_sync.hip: persistent phase-synchronous 8-wave workgroups): SiLU, the window stager and the resident-K MFMA loops.
#pragma once

#include "conv_mfma_impl.h"

namespace fv {

typedef float f32x4p __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// (TT, XS, Xr: optional second destination, see below)
// As[r][col] = silu(x[r][t0 - HP + col]) for r < C, col < WA_RAW (0 outside [0, T): silu(0) = 0 is the conv's zero padding).
// Each wave stages C/4 whole rows: per element one buffer load (row descriptor in SGPRs, column offset in a VGPR, out-of-
// range columns come back as 0 from the hardware bounds check), silu, one ds_write — the PMC profile of the first version
// showed the kernel bound by VALU issue (1400 VALU vs 96 MFMA instructions per wave), most of it index / clamp / 64-bit
// address arithmetic around these loads.
template <int C, int WA_RAW, int WA, int HP, int TT = 0, int XS = 0>
__device__ __forceinline__ void stage_window(const float* __restrict__ xb, float* __restrict__ As, int wave, int lane, int t0,
                                             int T, float* __restrict__ Xr = nullptr) {
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
            // the raw centre columns [HP, HP + TT) — the residual operand of the final epilogue — stay in LDS as well: x is read from
            // HBM once per tile (round 2 fetched it again in the epilogue: 1.6 - 2.7x the algorithmic traffic, profiles/traffic.json)
            if constexpr (TT > 0) {
                if (col >= HP && col < HP + TT) Xr[(wave * ROWS + rr) * XS + col - HP] = v[rr][i];
            }
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
                                                f32x4p (&acc)[NT]) {
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


}  // namespace fv
