// Dilated "same" Conv1d (k = 3 / 7 / 11: the ResBlock / AMPBlock convs, fish_vocoder/modules/generators/hifigan.py:101-108,
// bigvgan.py:235-245) as an implicit GEMM over Winograd F(4,3) tap groups on the fp32 matrix cores: 26 / 16 / 6 matrix products per
// FOUR outputs and (c_out, c_in) — 6.5 / 4 / 1.5 per output against F(2,3)'s 8 / 5 / 2 (conv_wino_impl.h) and the direct sum's 11 / 7 / 3.
//
// Quad lattice.  With dilation D outputs are grouped as (t, t + D, t + 2D, t + 3D): quad column
//   n = q D + r  (0 <= r < D)   <->   t0(n) = 4 D q + r,     X_j[n] = x'[t0(n) + j D]  (j = 0..3),    x'[tau] = act(x[tau - pad])
// and x'[t0(n) + (4 + j) D] = X_j[n + D]: a shift by four taps is a shift by D columns for every dilation.
// Tap groups {0,1,2}, {4,5,6}, {8,9,10} (group g reads column n + g D); interpolation points 0, ±1, ±2, ∞ (Lavin & Gray 2016):
//   V0 = 4 x0 - 5 x2 + x4          V1 = (x4 - 4 x2) + (x3 - 4 x1)     V2 = (x4 - 4 x2) - (x3 - 4 x1)
//   V3 = (x4 - x2) + 2 (x3 - x1)   V4 = (x4 - x2) - 2 (x3 - x1)       V5 = 4 x1 - 5 x3 + x5          (x4 = X0[n + D], x5 = X1[n + D])
// transformed weights (host, in double, conv_layer.hip):  g0/4, -(g0+g1+g2)/6, -(g0-g1+g2)/6, g0/24+g1/12+g2/6, g0/24-g1/12+g2/6, g2
// six accumulator planes m_p += U_p V_p and
//   y[t0] = m0+m1+m2+m3+m4    y[t0+D] = (m1-m2) + 2 (m3-m4)    y[t0+2D] = (m1+m2) + 4 (m3+m4)    y[t0+3D] = (m1-m2) + 8 (m3-m4) + m5.
// The taps between the groups (3, 7) are plain products: m0 only reaches y[t0] and m5 only y[t0+3D], so the first and fourth output take
// theirs in those planes; the second and third get one private plane each (S1, S2).  Eight planes per quad column in all.
// Accuracy (tools/experiments/winograd_f43_precision.py): HiFiGAN-V1 waveform 2.0e-6 from float64 (direct fp32 sums 2.0e-6, F(2,3) 1.7e-6),
// BigVGAN 4.5e-6 (3.2e-6 / 3.0e-6): in one dimension the transform constants (≤ 8) cost about 1.5 x a layer's direct-sum error.
//
// Work split: a workgroup = 64 rows x 32 quad columns (128 outputs); wave = (32-row tile wm, plane half h).  Half 0 holds m0, m1, m2, S1 and
// half 1 m5, m4, m3, S2 — 64 accumulator registers each (three waves per SIMD, as conv_wino_kernel) and the same 13 / 8 / 3 matrix products per
// 2 channels, so the halves stay in step.  The LDS row of a channel is laid out so that half 1's operand addresses are half 0's plus a
// constant (V0 V1 V2 | V5 V4 V3 | X3 X2 | X0 X1 with X2, X1 displaced by D columns): one instruction stream, two lane bases.
// The output transform needs both halves: each wave passes two 32 x 32 combinations to its partner through LDS (the chunk buffers are
// free by then) and stores two of the four output phases.
#pragma once
#include "conv_mfma_impl.h"

namespace fv {

template <int KS, int DIL, int WM>
struct Wino4Geom {
    static constexpr int NW = 2 * WM;                  // waves per workgroup: WM 32-row tiles x two plane halves
    static constexpr int NG = (KS + 1) / 4;            // F(4,3) groups at taps 0, 4, 8
    static constexpr int NS = (KS - 3) / 4;            // single taps 3, 7
    static constexpr int NV = 3 * NG + 2 * NS;         // virtual taps of one plane half = MFMA k-step groups per 8-channel sub-chunk
    static constexpr int NBQ = 32;                     // quad columns per workgroup
    static constexpr int WD = NBQ + DIL * (NG - 1);    // columns of a transformed plane (largest group shift: D (NG - 1) >= the single taps' D NS)
    static constexpr int WR = WD + DIL;                // columns of X0..X3 (the transform reads column n + D)
    static constexpr int ROW = 10 * WR + DIL;          // floats per channel row: V0 V1 V2 V5 V4 V3 X3 X2 | X0 at 8 WR, X1 at 9 WR + D
    static constexpr int xoff(int j) { return j == 0 ? 8 * WR : j == 1 ? 9 * WR + DIL : j == 2 ? 7 * WR : 6 * WR; }   // X_j planes
    static constexpr int voff(int pl) { return pl < 3 ? pl * WR : (8 - pl) * WR; }                                      // V_p planes
    static constexpr int HALF_G = 3 * WR, HALF_S = WR + DIL;                  // half 1's lane base - half 0's: group taps / single taps
#ifndef FV_X_WINO4_LDS
#define FV_X_WINO4_LDS (53 * 1024)
#endif
#ifndef FV_X_WINO4_SUBS_MAX
#define FV_X_WINO4_SUBS_MAX 2
#endif
    static constexpr int subs_fit(int s) { return (s > 1 && 2 * kChunk * s * ROW * 4 + 64 > FV_X_WINO4_LDS) ? subs_fit(s / 2) : s; }
    static constexpr int SUBS = subs_fit(FV_X_WINO4_SUBS_MAX);
    static constexpr int CH = kChunk * SUBS;
    static constexpr int RPW = CH / NW;                // channel rows staged by one wave
    static constexpr int NE = (RPW * WR + 63) / 64;    // lattice elements (four samples each) per lane and chunk
    static constexpr int XS_F = 2 * CH * ROW + 8 > NW * 2048 ? 2 * CH * ROW + 8 : NW * 2048;   // (the epilogue's exchange: NW waves x 2 x 16 x 64 floats)
    static constexpr int acc_of(int v) { return v < 3 * NG ? v % 3 : ((v - 3 * NG) % 2 == 0 ? 0 : 3); }
    static constexpr bool single_of(int v) { return v >= 3 * NG; }
    static constexpr int off_of(int v) {               // LDS offset of virtual tap v inside a channel row, half 0 (half 1: + HALF_G / HALF_S)
        if (v < 3 * NG) return (v % 3) * WR + DIL * (v / 3);
        const int s = (v - 3 * NG) / 2;
        return (v - 3 * NG) % 2 == 0 ? 6 * WR + s * DIL : 8 * WR + (s + 1) * DIL;
    }
};

// C64: the 64-channel layers (the narrowest that take this kernel: eight 8-channel blocks, M = 64 = one workgroup row) are their own instances — a constant trip
// count for the chunk loop, and their own rows in rocprofv3's per-kernel statistics (the C = 128 stage's launches have the same grid at B = 32)
template <int KS, int DIL, int WM, bool C64>
#ifndef FV_X_WINO4_OCC
#define FV_X_WINO4_OCC 3
#endif
__global__ __launch_bounds__(128 * WM, (WM == 2 ? FV_X_WINO4_OCC : 4)) void conv_wino4_kernel(const ConvParams p) {
    using G = Wino4Geom<KS, DIL, WM>;
    constexpr int NV = G::NV, NBQ = G::NBQ, WR = G::WR, ROW = G::ROW, SUBS = G::SUBS, CH = G::CH, RPW = G::RPW, NE = G::NE;
    __shared__ float xs[G::XS_F];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, h = wave & 1;
    // workgroup b runs on XCD b % 8: neighbouring tiles of a clip behind one L2 (conv_wino_impl.h)
    int bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (bid >= p.wg_total) return;
    const int n_tile = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int m_blk = bid % p.m_blks;
    const int b = bid / p.m_blks;
    const int n0 = n_tile * NBQ;
    const float* __restrict__ xb = p.x + (long long)b * p.x_bstride;

    FV_CV_STAMP(0);
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // ---- staging plan: this wave owns channel rows wave * RPW .. + RPW - 1 of every chunk; lane element i = quad column
    // (lane + 64 i) % WR of row (lane + 64 i) / WR.  Byte offsets relative to the chunk's first row, 0xFFFFFFFF outside [0, Tin)
    // (raw buffer loads return 0 there, and for rows past C_in through the descriptor's size).  Elements past the wave's last one repeat
    // it; columns >= WD get transformed values nobody reads — no masks or branches in the staging code (conv_wino_impl.h) ----
    unsigned vo[NE][4];
    int lo[NE];               // LDS float offset of (row, column) inside the chunk buffer
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        int e = lane + 64 * i;
        e = e < RPW * WR ? e : RPW * WR - 1;
        const int rr = e / WR, c = e - rr * WR;
        const int n = n0 + c;
        const int q = n / DIL;
        const int t0 = 4 * DIL * q + (n - q * DIL) - p.pad_l;
        const int row = wave * RPW + rr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + j * DIL;
            vo[i][j] = (t >= 0 && t < p.Tin) ? (unsigned)(row * p.Tin + t) * 4u : 0xFFFFFFFFu;
        }
        lo[i] = row * ROW + c;
    }
    float sx[4 * NE];   // [j * NE + i]
    auto load_chunk = [&](int c) {
        const int cbase = c * CH;
        const long long span = p.x_bstride - (long long)cbase * p.Tin;
        const long long rows = (long long)(p.Cin - cbase) * p.Tin;
        const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(xb + (long long)cbase * p.Tin, (unsigned)((rows < span ? rows : span) * 4));
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sx[j * NE + i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, vo[i][j], 0, 0));
    };
    auto store_chunk = [&](float* dst) {
        act_apply_all(sx, p.pre_act, p.slope);   // act(0) == 0 keeps the zero padding
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[lo[i] + G::xoff(j)] = sx[j * NE + i];
        // the neighbours (column + D of the same row) were written by this wave: its LDS operations execute in order, the fence
        // only keeps the compiler from moving the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float x4[NE], x5[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            x4[i] = dst[lo[i] + G::xoff(0) + DIL];
            x5[i] = dst[lo[i] + G::xoff(1) + DIL];
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const float x0 = sx[i], x1 = sx[NE + i], x2 = sx[2 * NE + i], x3 = sx[3 * NE + i];
            const float ea = fmaf(-4.0f, x2, x4[i]), eb = fmaf(-4.0f, x1, x3);
            const float ec = x4[i] - x2, ed = x3 - x1;
            dst[lo[i] + G::voff(0)] = fmaf(4.0f, x0, fmaf(-5.0f, x2, x4[i]));
            dst[lo[i] + G::voff(1)] = ea + eb;
            dst[lo[i] + G::voff(2)] = ea - eb;
            dst[lo[i] + G::voff(3)] = fmaf(2.0f, ed, ec);
            dst[lo[i] + G::voff(4)] = fmaf(-2.0f, ed, ec);
            dst[lo[i] + G::voff(5)] = fmaf(4.0f, x1, fmaf(-5.0f, x3, x5[i]));
        }
    };

    const int mt0 = m_blk * WM + wm;   // 32-row tile
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    const int wbase = __builtin_amdgcn_readfirstlane((mt0 * 2 + h) * (p.nchunk * NV * 1024));   // bytes per (m-tile, half): nchunk * NV fragments of 1 KiB
    auto load_a = [&](int goff_b) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wbase + goff_b, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    const int b_lane_g = (lane >> 5) * ROW + (lane & 31) + h * G::HALF_G;
    const int b_lane_s = (lane >> 5) * ROW + (lane & 31) + h * G::HALF_S;

    constexpr int STEPS = SUBS * NV;
#ifndef FV_X_WINO4_DA
#define FV_X_WINO4_DA 3
#endif
    constexpr int DA = FV_X_WINO4_DA;   // weight prefetch distance in virtual taps (4 MFMAs each)
    float4 aq[DA + 1];
    float b_cur[4], b_nxt[4];
    const int nch = C64 ? 8 / SUBS : (p.nchunk_real + SUBS - 1) / SUBS;
    load_chunk(0);
#pragma unroll
    for (int d = 0; d < DA; ++d) aq[d] = load_a(d * 1024);
    for (int c = 0; c < nch; ++c) {
        float* xsb = xs + (c & 1) * (CH * ROW);
#ifdef FV_X_WINO4_PRIO
        __builtin_amdgcn_s_setprio(FV_X_WINO4_PRIO);
#endif
        store_chunk(xsb);
#ifdef FV_X_WINO4_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();
        if (c < 12) FV_CV_STAMP(1 + c);
        if (c + 1 < nch) load_chunk(c + 1);
        const int gchunk_b = __builtin_amdgcn_readfirstlane((c * STEPS + DA) * 1024);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) b_cur[pp] = xsb[(G::single_of(0) ? b_lane_s : b_lane_g) + 2 * pp * ROW + G::off_of(0)];
        static_for<STEPS>([&](auto st_c) __attribute__((always_inline)) {
            constexpr int st = decltype(st_c)::value;
            constexpr int A = G::acc_of(st % NV);
            constexpr int sub_n = (st + 1) / NV, off_n = G::off_of((st + 1) % NV);
            constexpr bool sgl_n = G::single_of((st + 1) % NV);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float av = m == 0 ? aq[0].x : m == 1 ? aq[0].y : m == 2 ? aq[0].z : aq[0].w;
                acc[A] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_cur[m], acc[A], 0, 0, 0);
                // one weight fragment (four MFMAs ahead of DA virtual taps) and the next virtual tap's four operands, spread over the MFMAs
                if (m == 0) aq[DA] = load_a(gchunk_b + st * 1024);
                if constexpr (st + 1 < STEPS) b_nxt[m] = xsb[(sgl_n ? b_lane_s : b_lane_g) + (sub_n * kChunk + 2 * m) * ROW + off_n];
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < DA; ++d) aq[d] = aq[d + 1];
            if constexpr (st + 1 < STEPS) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) b_cur[pp] = b_nxt[pp];
            }
        });
    }

    FV_CV_STAMP(13);
    // ---- output transform.  Both halves send  slot 0 = acc1 + acc2  and  slot 1 = ± (acc1 - acc2) [x 2 in half 1]  and keep two sums in place:
    //   half 0 (m0 m1 m2 S1):  A = m0 + (m1 + m2) -> y[t0] with the partner's slot 0 (m3 + m4);   B = S1 + (m1 - m2) -> y[t0 + D] with slot 1 (2 (m3 - m4))
    //   half 1 (m5 m4 m3 S2):  A = m5 + 8 (m3 - m4) -> y[t0 + 3D] with the partner's slot 1 (m1 - m2);   B = S2 + 4 (m3 + m4) -> y[t0 + 2D] with slot 0 (m1 + m2)
    // A replaces acc[0], B replaces acc[3]: 96 live registers at the peak ----
    __syncthreads();   // every wave is past its last operand read: the chunk buffers become the exchange area
    {
        float* ex = xs + wave * 2048 + lane;
        if (h == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sm = acc[1][r] + acc[2][r], df = acc[1][r] - acc[2][r];
                ex[r * 64] = sm;
                ex[1024 + r * 64] = df;
                acc[0][r] += sm;
                acc[3][r] += df;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sm = acc[2][r] + acc[1][r], df = acc[2][r] - acc[1][r];
                ex[r * 64] = sm;
                ex[1024 + r * 64] = df + df;
                acc[0][r] = fmaf(8.0f, df, acc[0][r]);
                acc[3][r] = fmaf(4.0f, sm, acc[3][r]);
            }
        }
    }
    __syncthreads();
    const float* pa = xs + (wave ^ 1) * 2048 + h * 1024 + lane;         // partner's slot for A (half 0: slot 0, half 1: slot 1)
    const float* pb = xs + (wave ^ 1) * 2048 + (1 - h) * 1024 + lane;   // ... and for B
    if (mt0 * 32 >= p.M) return;
    const int n = n0 + (lane & 31);
    const int q = n / DIL;
    const int ta = 4 * DIL * q + (n - q * DIL) + 3 * h * DIL, tb = ta + (1 - 2 * h) * DIL;   // half 0: t0, t0 + D; half 1: t0 + 3D, t0 + 2D
    // The common case — whole 32-row tiles, bias [+ residual] [+ post-activation], plain store — without per-element offset registers
    // (conv_wino_impl.h): all bias and residual operands are requested before the partner's planes are read back.
    if (p.M % 32 == 0 && p.gamma == nullptr && p.out_mode == OUT_SET && p.acc_scale == 1.0f) {
        const unsigned span = (unsigned)(p.y_bstride * 4);
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * p.y_bstride, span);
        const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(p.res ? p.res + (long long)b * p.y_bstride : p.y, span);
        const __amdgpu_buffer_rsrc_t brs = uniform_rsrc(p.bias, (unsigned)(p.M * 4));
        const int mrow = 4 * (lane >> 5);                                   // lane part of the row; + mt0 * 32 + (r & 3) + 8 * (r >> 2) in SGPRs
        const unsigned va = ta < p.N ? (unsigned)(mrow * p.N + ta) * 4u : 0xFFFFFFFFu;
        const unsigned vb = tb < p.N ? (unsigned)(mrow * p.N + tb) * 4u : 0xFFFFFFFFu;
        float bias[16], ra[16], rb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            bias[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, mrow * 4, (mt0 * 32 + (r & 3) + 8 * (r >> 2)) * 4, 0));
        if (p.res) {
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
        const bool has_res = p.res != nullptr;
        float oa[16], ob[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float y0 = acc[0][r] + pa[r * 64];
            const float y1 = acc[3][r] + pb[r * 64];
            oa[r] = fmaf(y0, 1.0f, bias[r]);
            ob[r] = fmaf(y1, 1.0f, bias[r]);
            if (has_res) {
                oa[r] += ra[r];
                ob[r] += rb[r];
            }
        }
        act_apply_all(oa, p.post_act, p.slope);   // (c1 of a ResBlock pair carries the SiLU in front of c2: hifigan.py:104-106)
        act_apply_all(ob, p.post_act, p.slope);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int so = (int)((unsigned)(mt0 * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)p.N * 4u);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oa[r]), yrs, va, so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ob[r]), yrs, vb, so, 0);
        }
#ifdef FV_X_CONV_TS
        __builtin_amdgcn_s_waitcnt(0);
#endif
        FV_CV_STAMP(14);
        return;
    }
    f32x16 out[1][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        out[0][0][r] = acc[0][r] + pa[r * 64];
        out[0][1][r] = acc[3][r] + pb[r * 64];
    }
    const int coff[2] = {ta, tb};
    const bool cok[2] = {ta < p.N, tb < p.N};
    conv_epilogue_cols<1, 2>(p, out, b, mt0, coff, cok, lane);
#ifdef FV_X_CONV_TS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    FV_CV_STAMP(14);
}

template <int KS, int WM, bool C64>
inline bool launch_wino4_kw(const ConvParams& p0, int batch, hipStream_t s) {
    ConvParams p = p0;
    p.wg_total = batch * p.m_blks * p.n_tiles;
    const int grid = (p.wg_total + 7) / 8 * 8;
    switch (p.dil) {
        case 1: hipLaunchKernelGGL((conv_wino4_kernel<KS, 1, WM, C64>), dim3(grid), dim3(128 * WM), 0, s, p); return true;
        case 3: hipLaunchKernelGGL((conv_wino4_kernel<KS, 3, WM, C64>), dim3(grid), dim3(128 * WM), 0, s, p); return true;
        case 5: hipLaunchKernelGGL((conv_wino4_kernel<KS, 5, WM, C64>), dim3(grid), dim3(128 * WM), 0, s, p); return true;
        default: return false;
    }
}

// (WM = 4 — eight-wave, 128-row workgroups — measured 10 - 25 % slower than WM = 2 and is not instantiated: LOG R4.14)
template <int KS>
inline bool launch_wino4_k(const ConvParams& p, int batch, hipStream_t s) {
    return (p.Cin == 64 && p.M == 64) ? launch_wino4_kw<KS, 2, true>(p, batch, s) : launch_wino4_kw<KS, 2, false>(p, batch, s);
}

}  // namespace fv
