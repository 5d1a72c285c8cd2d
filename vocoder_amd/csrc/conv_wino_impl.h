// Dilated "same" Conv1d (k = 3 / 7 / 11: the ResBlock / AMPBlock convs, fish_vocoder/modules/generators/hifigan.py:101-108,
// bigvgan.py:235-245) as an implicit GEMM over Winograd F(2,3) tap groups on the fp32 matrix cores: 16 / 10 / 4 matrix products per
// output pair and (c_out, c_in) instead of 22 / 14 / 6.
//
// Pair lattice.  With dilation D the conv couples samples D apart, so outputs are paired as (t, t + D): pair column
//   n = q D + r  (0 <= r < D)   <->   t0(n) = 2 D q + r,     E[n] = x'[t0(n)],  O[n] = x'[t0(n) + D],    x'[tau] = act(x[tau - pad])
// and x'[t0(n) + 2D] = E[n + D], x'[t0(n) + 3D] = O[n + D]: every dilation looks the same in pair columns.
// Tap groups {0,1,2}, {4,5,6}, {8,9,10} (group g reads pair column n + 2 g D), transformed inputs (four planes, computed once per staged
// window and shared by all groups and output rows):
//   d0 = E[n] - E[n+D]    d1 = O[n] + E[n+D]    d2 = E[n+D] - O[n]    d3 = O[n] - O[n+D]
// transformed weights (host, in double, conv_layer.hip: conv_layer_create):  g0,  (g0+g1+g2)/2,  (g0-g1+g2)/2,  g2
// four accumulator planes   m_p += G_p d_p   and   y[t0] = m0 + m1 + m2,   y[t0 + D] = m1 - m2 - m3.
// The taps between the groups (3, 7) are plain products folded into the same accumulators: w_j O[n + q D] into m0 and
// -w_j E[n + (q+1) D] into m3, q = (j-1)/2 (m0 only reaches the first output of the pair, m3 only the second, negated).
// Accuracy: tools/experiments/winograd_precision.py — the HiFiGAN-V1 waveform deviates from float64 by 1.7e-6 this way, 2.0e-6 with the
// direct fp32 sums (F(2,3) has no large transform constants).
//
// Kernel = conv_mfma_kernel's pipeline (conv_mfma_impl.h) with virtual taps: a chunk of 8 / 16 channels is staged into LDS as six
// planes per channel (d0..d3, E, O), every virtual tap reads one plane at an immediate-offset shifted address and feeds one
// accumulator plane; weights stream from L2 in the same packed fragment order with NV virtual taps per sub-chunk.  The transform
// needs E / O of the neighbouring pair column: each wave stages whole channel rows, writes E / O, and reads the neighbours back
// itself (LDS operations of one wave execute in order) — one workgroup barrier per chunk as before.
#pragma once
#include "conv_mfma_impl.h"

namespace fv {

template <int KS, int DIL, int WM, int WN, int NT>
struct WinoGeom {
    static constexpr int NG = (KS + 1) / 4;            // F(2,3) groups at taps 0, 4, 8
    static constexpr int NS = (KS - 3) / 4;            // single taps 3, 7
    static constexpr int NV = 4 * NG + 2 * NS;         // virtual taps = MFMA k-step groups per 8-channel sub-chunk
    static constexpr int NBP = WN * NT * 32;           // output pairs per workgroup
    static constexpr int WD = NBP + 2 * DIL * (NG - 1);   // columns of a transformed plane (largest group shift: 2 D (NG - 1))
    static constexpr int WR = WD + DIL;                // columns of E / O (the transform reads column n + D)
    static constexpr int ROW = 6 * WR;                 // floats per channel row: six planes of WR columns (d0..d3 use the first WD)
    // 8-channel sub-chunks per LDS chunk (= per workgroup barrier): as many as fit the budget that leaves three workgroups per CU
#ifndef FV_X_WINO_LDS
#define FV_X_WINO_LDS (53 * 1024)
#endif
#ifndef FV_X_WINO_SUBS_MAX
#define FV_X_WINO_SUBS_MAX 4
#endif
    static constexpr int subs_fit(int s) { return (s > 1 && 2 * kChunk * s * ROW * 4 > FV_X_WINO_LDS) ? subs_fit(s / 2) : s; }
    static constexpr int SUBS = subs_fit(FV_X_WINO_SUBS_MAX);
    static constexpr int CH = kChunk * SUBS;
    static constexpr int RPW = CH / 4;                 // channel rows staged by one wave
    static constexpr int NE = (RPW * WR + 63) / 64;    // (E, O) pairs per lane and chunk
    static constexpr int acc_of(int v) { return v < 4 * NG ? v % 4 : ((v - 4 * NG) % 2 == 0 ? 0 : 3); }
    static constexpr int off_of(int v) {               // LDS offset of virtual tap v inside a channel row
        if (v < 4 * NG) return (v % 4) * WR + 2 * DIL * (v / 4);
        const int s = (v - 4 * NG) / 2;
        return (v - 4 * NG) % 2 == 0 ? 5 * WR + (2 * s + 1) * DIL : 4 * WR + (2 * s + 2) * DIL;
    }
};

template <int KS, int DIL, int WM, int WN, int NT>
#ifndef FV_X_WINO_OCC1
#define FV_X_WINO_OCC1 3
#endif
__global__ __launch_bounds__(256, (NT == 1 ? FV_X_WINO_OCC1 : 2)) void conv_wino_kernel(const ConvParams p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    using G = WinoGeom<KS, DIL, WM, WN, NT>;
    constexpr int NV = G::NV, NBP = G::NBP, WD = G::WD, WR = G::WR, ROW = G::ROW, SUBS = G::SUBS, CH = G::CH, RPW = G::RPW, NE = G::NE;
    __shared__ float xs[2][CH * ROW + 8];   // (+ 8: the transform of the last row's unused columns reads D floats past its O plane)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // Workgroup b runs on XCD b % 8: logical id = (b % 8) * (grid / 8) + b / 8 puts neighbouring tiles of a clip — which share the
    // cache lines of their halo columns (a 32-pair tile is two lines wide) — on one XCD, i.e. behind one L2
    int bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (bid >= p.wg_total) return;
    const int n_tile = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int m_blk = bid % p.m_blks;
    const int b = bid / p.m_blks;
    const int n0 = n_tile * NBP;
    const float* __restrict__ xb = p.x + (long long)b * p.x_bstride;

    FV_CV_STAMP(0);
#ifdef FV_X_CONV_TS
    if (p.dbg_ts && threadIdx.x == 0) p.dbg_ts[(long long)blockIdx.x * 16 + 15] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) | ((long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);   // HW_ID, XCC_ID
#endif
    f32x16 acc[4][NT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.f;

    // ---- staging plan: this wave owns channel rows wave * RPW .. + RPW - 1 of every chunk; lane element i = pair column
    // (lane + 64 i) % WR of row (lane + 64 i) / WR.  Byte offsets relative to the chunk's first row, 0xFFFFFFFF outside [0, Tin)
    // (raw buffer loads return 0 there, and for rows past C_in through the descriptor's size) ----
    // Every lane element is staged unconditionally: elements past the wave's last one repeat it (same loads, same values, same LDS
    // addresses), and columns >= WD get transformed values nobody reads — no per-element masks or branches in the staging code.
    unsigned voE[NE], voO[NE];
    int lo[NE];               // LDS float offset of (row, column) inside the chunk buffer
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        int e = lane + 64 * i;
        e = e < RPW * WR ? e : RPW * WR - 1;
        const int rr = e / WR, c = e - rr * WR;
        const int n = n0 + c;
        const int q = n / DIL;
        const int tE = 2 * DIL * q + (n - q * DIL) - p.pad_l, tO = tE + DIL;
        const int row = wave * RPW + rr;
        voE[i] = (tE >= 0 && tE < p.Tin) ? (unsigned)(row * p.Tin + tE) * 4u : 0xFFFFFFFFu;
        voO[i] = (tO >= 0 && tO < p.Tin) ? (unsigned)(row * p.Tin + tO) * 4u : 0xFFFFFFFFu;
        lo[i] = row * ROW + c;
    }
    float sE[NE], sO[NE];
    auto load_chunk = [&](int c) {
        const int cbase = c * CH;
        const long long span = p.x_bstride - (long long)cbase * p.Tin;
        const long long rows = (long long)(p.Cin - cbase) * p.Tin;
        const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(xb + (long long)cbase * p.Tin, (unsigned)((rows < span ? rows : span) * 4));
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            sE[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, voE[i], 0, 0));
            sO[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, voO[i], 0, 0));
        }
    };
    auto store_chunk = [&](float* dst) {
        act_apply_all(sE, p.pre_act, p.slope);   // act(0) == 0 keeps the zero padding
        act_apply_all(sO, p.pre_act, p.slope);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            dst[lo[i] + 4 * WR] = sE[i];
            dst[lo[i] + 5 * WR] = sO[i];
        }
        // the neighbours (column + D of the same row) were written by this wave: its LDS operations execute in order, the fence
        // only keeps the compiler from moving the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float e1[NE], o1[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            e1[i] = dst[lo[i] + 4 * WR + DIL];
            o1[i] = dst[lo[i] + 5 * WR + DIL];
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            dst[lo[i]] = sE[i] - e1[i];
            dst[lo[i] + WR] = sO[i] + e1[i];
            dst[lo[i] + 2 * WR] = e1[i] - sO[i];
            dst[lo[i] + 3 * WR] = sO[i] - o1[i];
        }
    };

    const int mt0 = m_blk * WM + wm;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    const int wbase = __builtin_amdgcn_readfirstlane(mt0 * (p.nchunk * NV * 1024));   // bytes per m-tile: nchunk * NV k-step groups of 1 KiB
    auto load_a = [&](int goff_b) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wbase + goff_b, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    const int b_lane = (lane >> 5) * ROW + wn * (NT * 32) + (lane & 31);

    constexpr int STEPS = SUBS * NV;
#ifndef FV_X_WINO_DA
#define FV_X_WINO_DA 3
#endif
    constexpr int DA = FV_X_WINO_DA;   // weight prefetch distance in virtual taps (4 NT MFMAs each)
    float4 aq[DA + 1];
    float b_cur[4][NT], b_nxt[4][NT];
    const int nch = (p.nchunk_real + SUBS - 1) / SUBS;
    load_chunk(0);
#pragma unroll
    for (int d = 0; d < DA; ++d) aq[d] = load_a(d * 1024);
    for (int c = 0; c < nch; ++c) {
        float* xsb = xs[c & 1];
        store_chunk(xsb);
        __syncthreads();
        if (c < 12) FV_CV_STAMP(1 + c);
        if (c + 1 < nch) load_chunk(c + 1);
        const int gchunk_b = __builtin_amdgcn_readfirstlane((c * STEPS + DA) * 1024);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) b_cur[pp][jn] = xsb[b_lane + 2 * pp * ROW + jn * 32 + G::off_of(0)];
        static_for<STEPS>([&](auto st_c) __attribute__((always_inline)) {
            constexpr int st = decltype(st_c)::value;
            constexpr int A = G::acc_of(st % NV);
            constexpr int NM = 4 * NT, NLDX = 1 + 4 * NT;
            constexpr int sub_n = (st + 1) / NV, off_n = G::off_of((st + 1) % NV);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int pp = m / NT, jn = m % NT;
                const float av = pp == 0 ? aq[0].x : pp == 1 ? aq[0].y : pp == 2 ? aq[0].z : aq[0].w;
                acc[A][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_cur[pp][jn], acc[A][jn], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NLDX; ++k) {
                    if (k * NM / NLDX == m) {
                        if (k == 0) {
                            aq[DA] = load_a(gchunk_b + st * 1024);
                        } else if constexpr (st + 1 < STEPS) {
                            const int pp2 = (k - 1) / NT, jn2 = (k - 1) % NT;
                            b_nxt[pp2][jn2] = xsb[b_lane + (sub_n * kChunk + 2 * pp2) * ROW + jn2 * 32 + off_n];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < DA; ++d) aq[d] = aq[d + 1];
            if constexpr (st + 1 < STEPS) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn) b_cur[pp][jn] = b_nxt[pp][jn];
            }
        });
    }

    FV_CV_STAMP(13);
    // output transform + the shared fused epilogue: n-tile jn of the wave becomes two column sets, t0(n) and t0(n) + D.
    // (An 8-byte store per pair at D = 1 measured 1.75 x the HBM write bytes of the two 4-byte stores — 158 vs 90 MB per launch by
    // WRITE_SIZE — and no time gain: not used.)
#ifndef FV_X_WINO_NO_LEAN_EPI
    // The common case — whole 32-row tiles, bias [+ residual] [+ post-activation], plain store (every ResBlock / AMPBlock conv of the
    // three-stream forward) —
    // without per-element offset registers: a row is the SGPR offset of the buffer instruction, the column the VGPR one (the range check
    // covers the VGPR part, so a masked column stays masked).  All 16 bias and 32 residual operands of the wave's tile are requested
    // before the output transform: one round trip instead of conv_epilogue_cols' two, and the transform runs under it.  Same
    // arithmetic as the general path (fmaf(acc, 1, bias) + residual): bit-identical.
    if (NT == 1 && p.M % 32 == 0 && p.gamma == nullptr && p.out_mode == OUT_SET && p.acc_scale == 1.0f) {
        if (mt0 * 32 >= p.M) return;
        const unsigned span = (unsigned)(p.y_bstride * 4);
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * p.y_bstride, span);
        const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(p.res ? p.res + (long long)b * p.y_bstride : p.y, span);
        const __amdgpu_buffer_rsrc_t brs = uniform_rsrc(p.bias, (unsigned)(p.M * 4));
        const int n = n0 + wn * 32 + (lane & 31);
        const int q = n / DIL;
        const int ta = 2 * DIL * q + (n - q * DIL), tb = ta + DIL;
        const int mrow = 4 * (lane >> 5);                                   // lane part of the row; + mt0 * 32 + (r & 3) + 8 * (r >> 2) in SGPRs
        const unsigned va = ta < p.N ? (unsigned)(mrow * p.N + ta) * 4u : 0xFFFFFFFFu;
        const unsigned vb = tb < p.N ? (unsigned)(mrow * p.N + tb) * 4u : 0xFFFFFFFFu;
        // D = 1: the pair's two outputs are adjacent samples — 8-byte residual loads and stores where every row starts 8-byte aligned (round 5, LOG R5.3)
        const bool pair8 = DIL == 1 && (p.N & 1) == 0 && (p.y_bstride & 1) == 0 &&
                           (((unsigned long long)p.y | (unsigned long long)(p.res ? p.res : p.y)) & 7ull) == 0;
        float bias[16], ra[16], rb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            bias[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, mrow * 4, (mt0 * 32 + (r & 3) + 8 * (r >> 2)) * 4, 0));
        if (p.res && pair8) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rrs, va, (int)((unsigned)(mt0 * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)p.N * 4u), 0);
                ra[r] = __uint_as_float(v.x);
                rb[r] = __uint_as_float(v.y);
            }
        } else if (p.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = (int)((unsigned)(mt0 * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)p.N * 4u);   // (< 4 GiB per item: conv_layer_run)
                ra[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, va, so, 0));
                rb[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, vb, so, 0));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ra[r] = rb[r] = 0.f;
        }
        const f32x16 y0 = (acc[0][0] + acc[1][0]) + acc[2][0];
        const f32x16 y1 = (acc[1][0] - acc[2][0]) - acc[3][0];
        const bool has_res = p.res != nullptr;
        float oa[16], ob[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            oa[r] = fmaf(y0[r], 1.0f, bias[r]);
            ob[r] = fmaf(y1[r], 1.0f, bias[r]);
            if (has_res) {
                oa[r] += ra[r];
                ob[r] += rb[r];
            }
        }
        act_apply_all(oa, p.post_act, p.slope);   // (c1 of a ResBlock pair carries the SiLU in front of c2: hifigan.py:104-106)
        act_apply_all(ob, p.post_act, p.slope);
        if (pair8) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                u32x2 v;
                v.x = __float_as_uint(oa[r]);
                v.y = __float_as_uint(ob[r]);
                __builtin_amdgcn_raw_buffer_store_b64(v, yrs, va, (int)((unsigned)(mt0 * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)p.N * 4u), 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = (int)((unsigned)(mt0 * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)p.N * 4u);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oa[r]), yrs, va, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ob[r]), yrs, vb, so, 0);
            }
        }
#ifdef FV_X_CONV_TS
        __builtin_amdgcn_s_waitcnt(0);
#endif
        FV_CV_STAMP(14);
        return;
    }
#endif
#pragma unroll
    for (int jn = 0; jn < NT; ++jn) {
        f32x16 out[1][2];
        out[0][0] = (acc[0][jn] + acc[1][jn]) + acc[2][jn];
        out[0][1] = (acc[1][jn] - acc[2][jn]) - acc[3][jn];
        const int n = n0 + wn * (NT * 32) + jn * 32 + (lane & 31);
        const int q = n / DIL;
        const int ta = 2 * DIL * q + (n - q * DIL);
        const int coff[2] = {ta, ta + DIL};
        const bool cok[2] = {ta < p.N, ta + DIL < p.N};
        conv_epilogue_cols<1, 2>(p, out, b, mt0, coff, cok, lane);
    }
#ifdef FV_X_CONV_TS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    FV_CV_STAMP(14);
}

template <int KS, int DIL>
inline bool launch_wino_cfg(const ConvParams& p0, int cfg, int batch, hipStream_t s) {
    ConvParams p = p0;
    p.wg_total = batch * p.m_blks * p.n_tiles;
    const int grid = (p.wg_total + 7) / 8 * 8;
    switch (cfg) {
        case WINO_128x32: hipLaunchKernelGGL((conv_wino_kernel<KS, DIL, 4, 1, 1>), dim3(grid), dim3(256), 0, s, p); return true;
        case WINO_64x64: hipLaunchKernelGGL((conv_wino_kernel<KS, DIL, 2, 2, 1>), dim3(grid), dim3(256), 0, s, p); return true;
        case WINO_32x128: hipLaunchKernelGGL((conv_wino_kernel<KS, DIL, 1, 4, 1>), dim3(grid), dim3(256), 0, s, p); return true;
#ifdef FV_X_WINO_NT2
        case WINO_128x64: hipLaunchKernelGGL((conv_wino_kernel<KS, DIL, 4, 1, 2>), dim3(grid), dim3(256), 0, s, p); return true;
        case WINO_64x128: hipLaunchKernelGGL((conv_wino_kernel<KS, DIL, 2, 2, 2>), dim3(grid), dim3(256), 0, s, p); return true;
#endif
        default: return false;
    }
}

template <int KS>
inline bool launch_wino_k(const ConvParams& p, int cfg, int batch, hipStream_t s) {
    switch (p.dil) {
        case 1: return launch_wino_cfg<KS, 1>(p, cfg, batch, s);
        case 3: return launch_wino_cfg<KS, 3>(p, cfg, batch, s);
        case 5: return launch_wino_cfg<KS, 5>(p, cfg, batch, s);
        default: return false;
    }
}

}  // namespace fv
