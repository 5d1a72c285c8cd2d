// Fused ResBlock (c1, c2) pair on Winograd F(4,4) tap groups for the narrow stages (C = 16 / 32, k = 7 / 11; round 5):
//
//     y = x + c2( silu( c1( silu(x) ) ) )          (one iteration of ResBlock1.forward,
//                                                    fish_vocoder/modules/generators/hifigan.py:102-107)
//
// in ONE launch with 20 / 13 matrix products per FOUR outputs and (c_out, c_in) in both convs — pair_wino_impl.h's F(2,3) groups leave
// 32 / 20, the direct sum 44 / 28.  Same data flow as pair_wino16_kernel (window -> LDS once, c1's output never leaves LDS, HBM traffic =
// x once + y once), on conv_wino44_impl.h's quad lattice and transform:
//   quad column n = q D + r <-> u0(n) = 4 D q + r, outputs u0 + j D (j = 0..3); X_j[n] = x'[u0(n) + j D]; tap group g (taps 4g .. 4g + 3)
//   reads the transformed planes at column n + g D; points +-1/2, +-1, +-2, inf (planes 0..6); U(inf) = the group's fourth tap — zero in
//   the last group of both kernel sizes, so the inf plane costs NG - 1 products.
// One workgroup = 4 wavefronts = WM m-tiles (16 rows each: all C rows) x WN n-tiles of 16 quad columns; a wave owns ONE (m-tile, n-tile):
// seven accumulator planes of four registers.  v_mfma_f32_16x16x4_f32; weights straight from L2 in fragment order (host: d_wpq16),
// DA fragments ahead.
//   phase 0   silu(x) window -> LDS as X0..X3 planes of c1's lattice (all C channels); raw centre columns -> LDS (residual operand)
//   per conv  for each chunk of 8 channels: X planes -> seven V planes (LDS -> LDS, 21 FMA-class instructions per lattice element),
//             barrier, MFMA loop over the chunk's NV virtual taps x 2 k-steps, barrier
//   c1 epilogue   output transform (bias rides in the +1 plane: its coefficient is 1 in every output), SiLU -> X0..X3 of c2's lattice
//   c2 epilogue   output transform, + raw x from LDS, 8-byte stores
#pragma once
#include "pair_wino_impl.h"

namespace fv {

template <int KS, int DIL, int C>
struct PQGeom {
    static_assert(C == 16 || C == 32, "one 16-row m-tile per wave");
    static_assert(KS == 7 || KS == 11, "k = 3 stays on F(2,3): 6 products per quad against 8, and measured slower (LOG R4.14)");
    static constexpr int KSZ = KS, DILV = DIL;
    static constexpr int WM = C / 16, WN = 4 / WM;
    static constexpr int NG = (KS + 3) / 4, NSH = NG - 1;
    static constexpr int NV = 7 * NSH + 6;               // virtual taps per channel: (group g < NSH: planes 0..6), (last group: planes 0..5)
    static constexpr int NF = (NV + 1) / 2;              // weight fragments per 8-channel chunk: two taps x two k-steps each
    static constexpr int CH = 8, NCHK = C / CH;
    static constexpr int NBQ = 16 * WN;                  // quad columns per workgroup
    static constexpr int NU = NBQ / DIL * DIL;           // ... of whole 4 D-sample blocks
    static constexpr int W1 = 4 * NU, TT = W1 - (KS - 1);
    static constexpr int H1 = (KS - 1) / 2 * DIL, H2 = (KS - 1) / 2, HP = H1 + H2;
    static constexpr int WD1 = NBQ + DIL * (NG - 1), WR1 = WD1 + DIL;   // V / X plane columns of c1
    static constexpr int WD2 = NBQ + (NG - 1), WR2 = WD2 + 1;           // ... of c2 (dilation 1)
    static constexpr int NQ1 = (WR1 + DIL - 1) / DIL;    // 4 D-sample blocks staged
    static constexpr int NP1 = 4 * DIL * NQ1;            // staged positions per channel row
    static constexpr int PX = (DIL * NQ1 + 1) / 2 * 2;   // X plane stride (>= WR1, WR2)
    static constexpr int SX = 4 * PX;                    // channel row of the X planes: X0 X1 X2 X3
    static constexpr int PV = pw_up(WD1, 32, 16);        // V row stride == 16 (mod 32): the 16x16x4 B read (two channel rows per 32 lanes)
    static constexpr int V_F = 7 * CH * PV;              // [plane][channel][column]
    static constexpr int XS = TT + 2;                    // raw-tile row stride (column TT: dump for the window's halo positions)
    // Round 6 (C = 32): the transform of chunk c + 1 rides inside chunk c's matrix loop into a SECOND V buffer — one workgroup barrier per chunk instead of
    // two, and the transform's vector instructions come from the wave that owns the MFMA stream instead of a phase of their own (conv_wino44_impl.h's in-loop
    // staging, LOG R6.2).  The second buffer takes the raw centre tile's place in LDS (three workgroups per CU: 53.3 KB each): the residual is re-read
    // from global memory — the lines this workgroup staged ~20 us earlier, requested before c2's loop.
#ifndef FV_X_PQ_INLOOP
#define FV_X_PQ_INLOOP 1
#endif
#ifndef FV_X_PQ_INLOOP16
#define FV_X_PQ_INLOOP16 0
#endif
    static constexpr bool INLOOP = FV_X_PQ_INLOOP && (C == 32 || (FV_X_PQ_INLOOP16 && C == 16 && DIL == 1));
    static constexpr bool XRES = !INLOOP;                // raw centre tile in LDS
    static constexpr int NVB = INLOOP ? 2 : 1;           // V buffers
    static constexpr int X_F = C * SX, XR_F = XRES ? C * XS + 16 : 0;
    static constexpr int XR_OFF = X_F + NVB * V_F;
    static constexpr int TRASH = XR_OFF + XR_F;
    static constexpr int LDS_FLOATS = TRASH + 4;
#ifndef FV_X_PQ_DA
#define FV_X_PQ_DA 4
#endif
    static constexpr int DA = FV_X_PQ_DA, RA = DA + 1;   // weight prefetch distance / ring slots (fragments)
    static_assert(PX >= WR1 && PX >= WR2 && PV >= WD1 && PV >= WD2, "plane strides cover both convs");
    static constexpr int g_of(int v) { return v / 7; }
    static constexpr int a_of(int v) { return v % 7; }
};

// Phase 0: silu(x) of the window [t0 - HP, ...) -> X planes of c1's lattice for all C channels (zero outside [0, T): silu(0) == 0 is the
// conv's zero padding), raw centre columns [t0, t0 + TT) -> Xr.  pw_stage_window's scheme: positions loaded in order (coalesced dwords through
// a per-row buffer descriptor), per-lane-slot offsets shared by all rows.
template <class G, int C>
__device__ __forceinline__ void pq_stage_window(const float* __restrict__ xb, int T, int t0, int wave, int lane, float* __restrict__ lds) {
    float* X = lds;
    [[maybe_unused]] float* Xr = lds + G::XR_OFF;
    constexpr int DIL = G::DILV;
    constexpr int NFULL = G::NP1 / 64, TAILW = G::NP1 % 64, RPW = C / 4, NTAIL = (RPW * TAILW + 63) / 64;
    const int ws = t0 - G::HP;
    const int row0 = wave * RPW;
    auto x_of = [&](int pp) {   // position of the window -> offset inside a channel row's X planes
        const int q = pp / (4 * DIL), rem = pp - 4 * DIL * q;
        const int j = rem / DIL;
        return j * G::PX + q * DIL + rem - j * DIL;
    };
    auto xr_of = [&](int pp) {
        const int c = pp - G::HP;
        return (c >= 0 && c < G::TT) ? c : G::TT;
    };
    float v[RPW][NFULL > 0 ? NFULL : 1];
    float vt[NTAIL > 0 ? NTAIL : 1];
    int x_off[NFULL > 0 ? NFULL : 1], xr_off[NFULL > 0 ? NFULL : 1];
    unsigned voff[NFULL > 0 ? NFULL : 1];
#pragma unroll
    for (int i = 0; i < NFULL; ++i) {
        const int pp = lane + 64 * i;
        x_off[i] = row0 * G::SX + x_of(pp);
        xr_off[i] = row0 * G::XS + xr_of(pp);
        voff[i] = (unsigned)(ws + pp) * 4u;   // negative positions wrap past the descriptor's size: the load returns 0
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (long long)(row0 + rr) * T), 0, (unsigned)T * 4u, 0x00020000);
#pragma unroll
        for (int i = 0; i < NFULL; ++i) v[rr][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff[i], 0, 0));
    }
    int x_t[NTAIL > 0 ? NTAIL : 1], xr_t[NTAIL > 0 ? NTAIL : 1];
    if constexpr (NTAIL > 0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (unsigned)(C * T) * 4u, 0x00020000);
#pragma unroll
        for (int j = 0; j < NTAIL; ++j) {
            const int e = lane + 64 * j;
            const bool ok = e < RPW * TAILW;
            const int rr = e / TAILW, pp = NFULL * 64 + e - rr * TAILW;
            const int tpos = ws + pp;
            const bool in = ok && tpos >= 0 && tpos < T;
            x_t[j] = ok ? (row0 + rr) * G::SX + x_of(pp) : G::TRASH;
            xr_t[j] = ok ? (row0 + rr) * G::XS + xr_of(pp) : G::TRASH - G::XR_OFF;   // (Xr-relative)
            vt[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, in ? (unsigned)((row0 + rr) * T + tpos) * 4u : 0xFFFFFFFFu, 0, 0));
        }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int i = 0; i < NFULL; ++i) {
            X[x_off[i] + rr * G::SX] = pw_silu(v[rr][i]);   // silu(0) == 0: the conv's zero padding
            if constexpr (G::XRES) Xr[xr_off[i] + rr * G::XS] = v[rr][i];
        }
    if constexpr (NTAIL > 0) {
#pragma unroll
        for (int j = 0; j < NTAIL; ++j) {
            X[x_t[j]] = pw_silu(vt[j]);
            if constexpr (G::XRES) Xr[xr_t[j]] = vt[j];
        }
    }
}

// X planes of the chunk's 8 channel rows -> seven V planes [plane][channel][column] (conv_wino44_impl.h's transform: symmetric points, the
// even / odd parts shared by +-a).  32 threads per channel row, consecutive columns.
// FULL whole slots of TPR consecutive columns per channel row; the TW columns left over (WD = 64 + D (NG - 1): 1 ... 10 of a 32-column slot) are
// flattened over (row, column) into the first CH * TW threads — the waves past them skip the slot instead of running it all but empty
// (-DFV_X_PQ_TAIL=0: the masked third slot of every wave, A/B builds).
// In pieces (round 6): load() = the 7 (FULL + 1) LDS reads, slot(j) / tail_slot() = one lattice element's 21 FMA-class instructions and seven LDS writes;
// pq_transform() runs them as a phase, the in-loop form (PQGeom::INLOOP) spreads them over the previous chunk's matrix loop.
#ifndef FV_X_PQ_TAIL
#define FV_X_PQ_TAIL 1
#endif
template <class G, int DX, int WD>
struct PQXform {
    static constexpr int TPR = 256 / G::CH;
    static constexpr int FULL = FV_X_PQ_TAIL ? WD / TPR : (WD + TPR - 1) / TPR, TW = FV_X_PQ_TAIL ? WD - FULL * TPR : 0;
    static constexpr int PS = G::CH * G::PV;   // plane stride
    static constexpr int NP = 1 + FULL + (TW > 0 ? 1 : 0);   // pieces: load, FULL slots, the tail slot
    int xo, vo, xt, vt, c0;
    bool tail;
    float a[FULL > 0 ? FULL : 1][7], t[7];
    __device__ __forceinline__ void init(int tid) {
        const int row = tid / TPR;
        c0 = tid % TPR;
        xo = row * G::SX + c0;
        vo = row * G::PV + c0;
        tail = TW > 0 && tid < G::CH * TW;
        const int tr = TW > 0 ? tid / (TW > 0 ? TW : 1) : 0, tc = FULL * TPR + tid - tr * TW;
        xt = tr * G::SX + tc;
        vt = tr * G::PV + tc;
    }
    static __device__ __forceinline__ void load7(const float* __restrict__ xp, float (&a)[7]) {
        a[0] = xp[0];
        a[1] = xp[G::PX];
        a[2] = xp[2 * G::PX];
        a[3] = xp[3 * G::PX];
        a[4] = xp[DX];
        a[5] = xp[G::PX + DX];
        a[6] = xp[2 * G::PX + DX];
    }
    static __device__ __forceinline__ void xform7(const float (&a)[7], float* __restrict__ dp) {
        const float x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3], x4 = a[4], x5 = a[5], x6 = a[6];
        const float eh = fmaf(4.0f, x0, fmaf(-5.0f, x2, x4)), oh = fmaf(4.0f, x1, fmaf(-5.0f, x3, x5));          // a = 1/2
        const float e1 = fmaf(-4.25f, x2, x4) + x0, o1 = fmaf(-4.25f, x3, x5) + x1;                              // a = 1
        const float e2 = fmaf(0.25f, x0, fmaf(-1.25f, x2, x4)), o2 = fmaf(0.25f, x1, fmaf(-1.25f, x3, x5));      // a = 2
        dp[0] = fmaf(0.5f, eh, oh);
        dp[PS] = fmaf(-0.5f, eh, oh);
        dp[2 * PS] = o1 + e1;
        dp[3 * PS] = o1 - e1;
        dp[4 * PS] = fmaf(2.0f, e2, o2);
        dp[5 * PS] = fmaf(-2.0f, e2, o2);
        dp[6 * PS] = fmaf(5.25f, x2 - x4, x6 - x0);
    }
    __device__ __forceinline__ void load(const float* __restrict__ xr) {
#pragma unroll
        for (int j = 0; j < FULL; ++j) load7(xr + xo + TPR * j, a[j]);
        if constexpr (TW > 0) {
            if (tail) load7(xr + xt, t);
        }
    }
    template <int J>
    __device__ __forceinline__ void slot(float* __restrict__ v) {
        if (FV_X_PQ_TAIL || TPR * (J + 1) <= WD || c0 + TPR * J < WD) xform7(a[J], v + vo + TPR * J);
    }
    __device__ __forceinline__ void tail_slot(float* __restrict__ v) {
        if constexpr (TW > 0) {
            if (tail) xform7(t, v + vt);
        }
    }
    // piece I of NP
    template <int I>
    __device__ __forceinline__ void piece(const float* __restrict__ xr, float* __restrict__ v) {
        if constexpr (I == 0) load(xr);
        else if constexpr (I <= FULL) slot<I - 1>(v);
        else if constexpr (I == FULL + 1 && TW > 0) tail_slot(v);
    }
};

template <class G, int DX, int WD>
__device__ __forceinline__ void pq_transform(const float* __restrict__ xr, float* __restrict__ v, int tid) {
    PQXform<G, DX, WD> xf;
    xf.init(tid);
    static_for<PQXform<G, DX, WD>::NP>([&](auto i_c) __attribute__((always_inline)) { xf.template piece<decltype(i_c)::value>(xr, v); });
}

// MFMA loop over one 8-channel chunk: NF weight fragments (taps 2 f, 2 f + 1 x k-steps 0, 1), four 16x16x4 MFMAs each, ordered (tap, ks) =
// (0,0) (1,0) (0,1) (1,1): consecutive instructions never share an accumulator plane.  bl: the lane's base into the V planes (k-quarter
// row and quad column folded in).  CI: chunk index at compile time (ring slots are constants).
struct PQNoHook {
    template <class F>
    __device__ __forceinline__ void operator()(F) const {}
};
// hook(f): called once per fragment behind its second MFMA — the in-loop form's transform pieces of the next chunk
template <class G, int DX, int CI, class Hook = PQNoHook>
__device__ __forceinline__ void pq_gemm_chunk(f32x4w (&acc)[7], const float* __restrict__ bl, const __amdgpu_buffer_rsrc_t wrs, int wvoff, int wsoff,
                                              float4 (&aq)[G::RA], Hook hook = Hook{}) {
    constexpr int NF = G::NF, NV = G::NV, DA = G::DA, RA = G::RA, PS = G::CH * G::PV;
    auto b_off = [](int v, int s) constexpr { return G::a_of(v) * PS + 4 * s * G::PV + G::g_of(v) * DX; };
    float b_cur[4], b_nxt[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int v = h & 1, s = h >> 1;
        b_cur[h] = v < NV ? bl[b_off(v, s)] : 0.f;
    }
    static_for<NF>([&](auto f_c) __attribute__((always_inline)) {
        constexpr int f = decltype(f_c)::value;
        constexpr int slot = (CI * NF + f) % RA;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int tap = h & 1, s = h >> 1;
            const int v = 2 * f + tap;
            if (v < NV) {
                const float4 a4 = aq[slot];
                float apin = tap == 0 ? (s == 0 ? a4.x : a4.y) : (s == 0 ? a4.z : a4.w);
                asm volatile("" : "+v"(apin));   // (pins the MFMA between the memory operations around it: pair_wino_impl.h)
                const int A = G::a_of(v);
                acc[A] = __builtin_amdgcn_mfma_f32_16x16x4f32(apin, b_cur[h], acc[A], 0, 0, 0);
                asm volatile("" : "+v"(acc[A]));
            }
            if (h == 0) {   // the fragment DA ahead (past the conv's last one: zero padding of the packed weights)
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, wsoff + (CI * NF + f + DA) * 1024, 0);
                aq[(CI * NF + f + DA) % RA] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (f + 1 < NF) {
                const int vn = 2 * (f + 1) + tap;
                if (vn < NV) b_nxt[h] = bl[b_off(vn, s)];
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!std::is_same<Hook, PQNoHook>::value) {
                if (h == 1) {
                    hook(f_c);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (f + 1 < NF) {
#pragma unroll
            for (int h = 0; h < 4; ++h) b_cur[h] = b_nxt[h];
        }
    });
}

// y_j = sum over the points a^j m(a) (+ m(inf) for j = 3): planes 0..6 = +1/2, -1/2, +1, -1, +2, -2, inf
__device__ __forceinline__ void pq_output_transform(const f32x4w (&m)[7], f32x4w (&y)[4]) {
    const f32x4w sh = m[0] + m[1], dh = m[0] - m[1], s1 = m[2] + m[3], d1 = m[2] - m[3], s2 = m[4] + m[5], d2 = m[4] - m[5];
    y[0] = (sh + s1) + s2;
    y[1] = (0.5f * dh + d1) + 2.0f * d2;
    y[2] = (0.25f * sh + s1) + 4.0f * s2;
    y[3] = ((0.125f * dh + d1) + 8.0f * d2) + m[6];
}

#ifndef FV_X_PQ_OCC
#define FV_X_PQ_OCC 3
#endif
#ifndef FV_X_PQ_OCC_D1
#define FV_X_PQ_OCC_D1 FV_X_PQ_OCC
#endif
template <int KS, int DIL, int C>
__global__ __launch_bounds__(256, (PQGeom<KS, DIL, C>::INLOOP && DIL == 1 ? FV_X_PQ_OCC_D1 : FV_X_PQ_OCC)) void pair_wino44_kernel(const PairParams p) {
    using G = PQGeom<KS, DIL, C>;
    constexpr int DA = G::DA, NCHK = G::NCHK, NF = G::NF;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* X = lds;
    float* V = lds + G::X_F;
    [[maybe_unused]] float* Xr = lds + G::XR_OFF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN;
    // a clip's neighbouring tiles on one XCD (they share the cache lines of their halo columns): resblock_pair.hip
    const int lid = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (lid >= p.n_tiles * p.batch) return;
    const int tile = lid % p.n_tiles, b = lid / p.n_tiles;
    const int t0 = tile * G::TT;
    const int T = p.T;
    const float* __restrict__ xb = p.x + (long long)b * C * T;

    // weight rings: this wave's m-tile; c1's first fragments are requested before anything else
    const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    const int wbase = __builtin_amdgcn_readfirstlane(wm * (NCHK * NF * 1024));   // bytes per m-tile
    float4 aq[G::RA];
    auto load_w = [&](const __amdgpu_buffer_rsrc_t rs, int f) __attribute__((always_inline)) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, wvoff, wbase + f * 1024, 0);
        return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    };
#pragma unroll
    for (int d = 0; d < DA; ++d) aq[d] = load_w(w1rs, d);

    pq_stage_window<G, C>(xb, T, t0, wave, lane, lds);

    const int krow = lane >> 4;               // C / D layout of 16x16x4: row = 4 (lane >> 4) + reg, column = lane & 15
    const int ncol = 16 * wn + (lane & 15);   // this lane's quad column
    f32x4w acc[7];
    auto init_acc = [&](const float* __restrict__ bias) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 7; ++a) acc[a] = f32x4w{0.f, 0.f, 0.f, 0.f};
        acc[2] = *(const f32x4w*)(bias + 16 * wm + 4 * krow);   // plane +1: coefficient 1 in all four outputs
    };
    init_acc(p.b1);
    const float* bl = V + krow * G::PV + ncol;
    __syncthreads();

    // ---- c1 ----
    // in-loop form: chunk 0 transformed as a phase, chunk c + 1 in pieces behind the MFMAs of chunk c's fragments 0 (the LDS reads), 2, 2 + PSTR, ... (one
    // lattice element each) into the other V buffer; one barrier per chunk.  The last chunk needs none behind it: the X planes were last read before the
    // previous barrier (c1's epilogue overwrites them), and c2's first transform writes the buffer the last chunk does not read.
    auto conv_loop = [&](auto dx_c, auto wd_c, auto ci0_c, const __amdgpu_buffer_rsrc_t wrs, int wsoff) __attribute__((always_inline)) {
        constexpr int DX = decltype(dx_c)::value, WDc = decltype(wd_c)::value, CI0 = decltype(ci0_c)::value;
        using XF = PQXform<G, DX, WDc>;
        constexpr int PSTR = (NF - 2) / (XF::NP - 1) > 0 ? (NF - 2) / (XF::NP - 1) : 1;
        static_assert(2 + (XF::NP - 2) * PSTR <= NF - 1, "every transform piece needs a fragment of its own");
        XF xf;
        xf.init(tid);
        static_for<XF::NP>([&](auto i_c) __attribute__((always_inline)) { xf.template piece<decltype(i_c)::value>(X, V); });
        __syncthreads();
        static_for<NCHK>([&](auto c_c) __attribute__((always_inline)) {
            constexpr int c = decltype(c_c)::value;
            if constexpr (c + 1 < NCHK) {
                const float* xn = X + (c + 1) * G::CH * G::SX;
                float* vn = V + ((c + 1) & 1) * G::V_F;
                auto hook = [&](auto f_c) __attribute__((always_inline)) {
                    constexpr int f = decltype(f_c)::value;
                    if constexpr (f == 0) xf.template piece<0>(xn, vn);
                    else if constexpr (f >= 2 && (f - 2) % PSTR == 0 && (f - 2) / PSTR + 1 < XF::NP) xf.template piece<(f - 2) / PSTR + 1>(xn, vn);
                };
                pq_gemm_chunk<G, DX, CI0 + c>(acc, bl + (c & 1) * G::V_F, wrs, wvoff, wsoff, aq, hook);
                __syncthreads();   // chunk c + 1 is transformed; every wave is past its reads of chunk c
            } else {
                pq_gemm_chunk<G, DX, CI0 + c>(acc, bl + (c & 1) * G::V_F, wrs, wvoff, wsoff, aq);
            }
        });
    };
    if constexpr (G::INLOOP) {
        static_assert(NCHK % 2 == 0, "c2's first transform must land in the buffer c1's last chunk does not read");
        conv_loop(std::integral_constant<int, DIL>{}, std::integral_constant<int, G::WD1>{}, std::integral_constant<int, 0>{}, w1rs, wbase);
    } else {
        static_for<NCHK>([&](auto c_c) __attribute__((always_inline)) {
            constexpr int c = decltype(c_c)::value;
            pq_transform<G, DIL, G::WD1>(X + c * G::CH * G::SX, V, tid);
            __syncthreads();
            pq_gemm_chunk<G, DIL, c>(acc, bl, w1rs, wvoff, wbase, aq);
            __syncthreads();   // the V buffer (next transform) and the X planes (c1 epilogue) are free again
        });
    }
    // c2's first weight fragments travel while the epilogue runs (the ring holds c1's overrun fragments: zeros, never used).  c1 consumed
    // NCHK * NF fragments: c2's fragment f sits in slot (NCHK * NF + f) % RA
    constexpr int S2 = NCHK * NF;
#pragma unroll
    for (int d = 0; d < DA; ++d) aq[(S2 + d) % G::RA] = load_w(w2rs, d);

    // ---- c1 epilogue: silu(c1 + b1) -> X planes of c2's lattice (mid[u], u = position - (t0 - H2): X_{u & 3}[u >> 2]) ----
    {
        f32x4w y[4];
        pq_output_transform(acc, y);
        const int q = ncol / DIL, r = ncol - q * DIL;
        const int u0 = 4 * DIL * q + r;
        const int ts = t0 - G::H2;
        const bool live = ncol < G::NU;       // (columns of a partial 4 D block produce nothing c2 reads)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = u0 + j * DIL;
            const bool in = live && ts + u >= 0 && ts + u < T;   // zero outside [0, T): c2's zero padding
            float* w = X + (16 * wm + 4 * krow) * G::SX + (u & 3) * G::PX + (u >> 2);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float s = pw_silu(y[j][rg]);
                if (live) w[rg * G::SX] = in ? s : 0.f;
            }
        }
    }
    init_acc(p.b2);
    // the last epilogue's addresses; without the raw tile in LDS (in-loop form) the residual is requested now, from the lines phase 0 staged, and arrives
    // during c2's loop
    const int tl = 4 * ncol;
    const int t = t0 + tl;
    const int row0 = 16 * wm + 4 * krow;
    const bool pair8 = (T & 1) == 0 && ((unsigned long long)p.y & 7ull) == 0;
    unsigned va[2], vb[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {              // outputs (0, 1) and (2, 3)
        const int tlh = tl + 2 * hh;
        const bool ok0 = tlh < G::TT && t + 2 * hh < T, ok1 = tlh + 1 < G::TT && t + 2 * hh + 1 < T;
        va[hh] = ok0 ? (unsigned)(row0 * T + t + 2 * hh) * 4u : 0xFFFFFFFFu;
        vb[hh] = ok1 ? (unsigned)(row0 * T + t + 2 * hh + 1) * 4u : 0xFFFFFFFFu;
    }
    [[maybe_unused]] float rx0[2][4], rx1[2][4];
    if constexpr (!G::XRES) {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (unsigned)(C * T) * 4u, 0x00020000);
        if ((T & 1) == 0 && ((unsigned long long)p.x & 7ull) == 0) {   // (an even T: both samples of a half are inside the clip or outside it)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, va[hh], __builtin_amdgcn_readfirstlane(rg * T * 4), 0);
                    rx0[hh][rg] = __uint_as_float(v.x);
                    rx1[hh][rg] = __uint_as_float(v.y);
                }
        } else {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int so = __builtin_amdgcn_readfirstlane(rg * T * 4);
                    rx0[hh][rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, va[hh], so, 0));
                    rx1[hh][rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, vb[hh], so, 0));
                }
        }
    }
    __syncthreads();

    // ---- c2 (dilation 1) ----
    if constexpr (G::INLOOP) {
        conv_loop(std::integral_constant<int, 1>{}, std::integral_constant<int, G::WD2>{}, std::integral_constant<int, NCHK>{}, w2rs, wbase - S2 * 1024);
    } else {
        static_for<NCHK>([&](auto c_c) __attribute__((always_inline)) {
            constexpr int c = decltype(c_c)::value;
            pq_transform<G, 1, G::WD2>(X + c * G::CH * G::SX, V, tid);
            __syncthreads();
            pq_gemm_chunk<G, 1, NCHK + c>(acc, bl, w2rs, wvoff, wbase - S2 * 1024, aq);
            if (c + 1 < NCHK) __syncthreads();
        });
    }

    // ---- c2 epilogue: + raw x (LDS; in-loop form: the registers requested above) -> y; the lane's four outputs are consecutive samples, stored as two
    // 8-byte halves (TT is even, so a half is inside the tile or outside it) ----
    {
        f32x4w y[4];
        pq_output_transform(acc, y);
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long long)b * C * T), 0, (unsigned)(C * T) * 4u, 0x00020000);
        const bool accum = p.out_mode == OUT_ACCUM;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {          // outputs (0, 1) and (2, 3)
            const int tlh = tl + 2 * hh;
            float o0[4], o1[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                if constexpr (G::XRES) {
                    const f32x2w xr = *(const f32x2w*)(Xr + row0 * G::XS + (tlh < G::TT ? tlh : 0) + rg * G::XS);
                    o0[rg] = y[2 * hh][rg] + xr.x;
                    o1[rg] = y[2 * hh + 1][rg] + xr.y;
                } else {
                    o0[rg] = y[2 * hh][rg] + rx0[hh][rg];
                    o1[rg] = y[2 * hh + 1][rg] + rx1[hh][rg];
                }
            }
            if (accum) {
                float a0[4], a1[4];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int so = __builtin_amdgcn_readfirstlane(rg * T * 4);
                    a0[rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, va[hh], so, 0));
                    a1[rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, vb[hh], so, 0));
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    o0[rg] = (a0[rg] + o0[rg]) * p.out_scale;
                    o1[rg] = (a1[rg] + o1[rg]) * p.out_scale;
                }
            }
            if (pair8) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    u32x2 v;
                    v.x = __float_as_uint(o0[rg]);
                    v.y = __float_as_uint(o1[rg]);
                    __builtin_amdgcn_raw_buffer_store_b64(v, yrs, va[hh], __builtin_amdgcn_readfirstlane(rg * T * 4), 0);
                }
            } else {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int so = __builtin_amdgcn_readfirstlane(rg * T * 4);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0[rg]), yrs, va[hh], so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o1[rg]), yrs, vb[hh], so, 0);
                }
            }
        }
    }
}

template <int KS, int DIL, int C>
inline bool launch_pair_wino44_one(const PairParams& p, int batch, hipStream_t s) {
    using G = PQGeom<KS, DIL, C>;
    PairParams q = p;
    q.n_tiles = (p.T + G::TT - 1) / G::TT;
    q.batch = batch;
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
    if (!FV_ENSURE_DYN_LDS((pair_wino44_kernel<KS, DIL, C>), lds)) return false;
    hipLaunchKernelGGL((pair_wino44_kernel<KS, DIL, C>), dim3((batch * q.n_tiles + 7) / 8 * 8), dim3(256), lds, s, q);
    return true;
}

template <int KS>
inline bool launch_pair_wino44_k(const PairParams& p, int C, int dil, int batch, hipStream_t s) {
#define FV_PQ_CASE(D)                                                          \
    if (dil == D) {                                                            \
        if (C == 16) return launch_pair_wino44_one<KS, D, 16>(p, batch, s);    \
        if (C == 32) return launch_pair_wino44_one<KS, D, 32>(p, batch, s);    \
        return false;                                                          \
    }
    FV_PQ_CASE(1) FV_PQ_CASE(3) FV_PQ_CASE(5)
#undef FV_PQ_CASE
    return false;
}

// tile width (final samples per workgroup) of the kernel a launch would take: the profiler's grid figure
constexpr int pair_wino44_tile(int ks, int dil, int C) {
    const int nbq = 16 * (4 / (C / 16));
    return 4 * (nbq / dil * dil) - (ks - 1);
}

}  // namespace fv
