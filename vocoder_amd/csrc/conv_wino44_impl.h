// Dilated "same" Conv1d (k = 7 / 11: the ResBlock / AMPBlock convs, fish_vocoder/modules/generators/hifigan.py:101-108, bigvgan.py:235-245) as an
// implicit GEMM over Winograd F(4,4) tap groups on the fp32 matrix cores: 20 / 13 matrix products per FOUR outputs and (c_out, c_in) against
// F(4,3)'s 26 / 16 (conv_wino4_impl.h), F(2,3)'s 32 / 20 (conv_wino_impl.h) and the direct sum's 44 / 28.
//
// Quad lattice as in conv_wino4_impl.h: column n = q D + r <-> t0(n) = 4 D q + r, X_j[n] = x'[t0(n) + j D]; a shift by four taps = D columns.
// Tap groups {0..3}, {4..7}, {8..11} (the taps past k are zero): FOUR taps per group, seven products per quad and group, no taps left between groups.
// Interpolation points ±1/2, ±1, ±2, ∞ — symmetric, so the input transform splits into even / odd parts (x4 x5 x6 = X0 X1 X2[n + D]):
//   E_a = x4 + c1 x2 + c0 x0,  O_a = x5 + c1 x3 + c0 x1,   V(±a) = O_a ± a E_a     (c0, c1) = (4, -5), (1, -17/4), (1/4, -5/4) for a = 1/2, 1, 2
//   V(∞) = (x6 - x0) + 21/4 (x2 - x4)                                                          21 FMA-class instructions per lattice element
// transformed weights (host, double): U(±a) = f_a (g0 ± a g1 + a² g2 ± a³ g3), f = 32/45, -2/9, 1/45 (the -a rows: -f for a = 1/2, 1 ... see
// conv_layer.hip), U(∞) = g3; outputs y_j = sum_a a^j m(+a) + (-a)^j m(-a)  (+ m(∞) for j = 3).
// U(∞) = g3 is ZERO for the last group of both kernel sizes (taps 11 / 7 do not exist): the ∞ plane costs NG - 1 products, 20 / 13 per quad in all.
// Accuracy (tools/experiments/winograd_f44_precision.py): per layer 1.3 - 1.8e-6 of full scale against 1.3 - 1.5e-6 for F(4,3) — symmetric points
// with |a| <= 2 keep the transform constants <= 8; the set 0, ±1, ±2, 3 would cost 5 x that.
//
// Work split: seven accumulator planes over two waves — half 0: m(1/2) m(-1/2) m(1), half 1: m(-1) m(2) m(-2), and m(∞) SHARED: each half
// accumulates it over two of every four channel pairs (its MFMAs for that plane use k-pairs 2h, 2h + 1 of an 8-channel block); 64 accumulator
// registers and 10 / 6.5 products per two channels for each half.  LDS row of a channel: V(1/2) V(-1/2) V(1) | V(-1) V(2) V(-2) | V(∞) | X0 X1 X2
// (the raw phases are only the transform's scratch).  Output transform: half 0 keeps y[t0], y[t0 + D], half 1 y[t0 + 2D], y[t0 + 3D]; each passes
// its partial sums of the partner's two outputs through the free chunk buffers.
#pragma once
#include "conv_mfma_impl.h"

namespace fv {

template <int KS, int DIL>
struct Wino44Geom {
    static constexpr int NG = (KS + 3) / 4;            // F(4,4) groups at taps 0, 4, 8
    static constexpr int NSH = NG - 1;                 // groups whose fourth tap exists: products of the shared ∞ plane
    static constexpr int NV = 3 * NG + 1;              // weight fragments per 8-channel sub-chunk and half: 3 NG full virtual taps + one shared-plane fragment
    static constexpr int NBQ = 32;                     // quad columns per workgroup
    static constexpr int WD = NBQ + DIL * (NG - 1);    // columns of a transformed plane
    static constexpr int WR = WD + DIL;                // columns of X0..X2 (the transform reads column n + D)
    static constexpr int ROW = 10 * WR;                // floats per channel row
    static constexpr int V_INF = 6 * WR, X_OFF = 7 * WR;
    static constexpr int HALF_G = 3 * WR;              // half 1's own planes
#ifndef FV_X_WINO44_LDS
#define FV_X_WINO44_LDS (52 * 1024)   // three workgroups per CU: 53.8 KB (k = 7, D = 5 with 16-channel chunks) measured as two
#endif
    static constexpr int subs_fit(int s) { return (s > 1 && 2 * kChunk * s * ROW * 4 + 64 > FV_X_WINO44_LDS) ? subs_fit(s / 2) : s; }
#ifndef FV_X_WINO44_SUBS_MAX
#define FV_X_WINO44_SUBS_MAX 2
#endif
    static constexpr int SUBS = subs_fit(FV_X_WINO44_SUBS_MAX);
    static constexpr int CH = kChunk * SUBS;
    static constexpr int RPW = CH / 4;                 // channel rows staged by one wave
    static constexpr int NE = (RPW * WR + 63) / 64;    // lattice elements (four samples each) per lane and chunk
    static constexpr int XS_F = 2 * CH * ROW + 8 > 8192 ? 2 * CH * ROW + 8 : 8192;   // (the epilogue's exchange: 4 waves x 2 x 16 x 64 floats)
    // step v of a sub-chunk: v < 3 NG: group v / 3, own plane v % 3, four MFMAs (channel pairs 0..3) into accumulator v % 3;
    // v == 3 NG: the shared plane: MFMA j -> group j / 2, channel pair 2 h + j % 2, accumulator 3 (2 NSH MFMAs)
    static constexpr bool shared_of(int v) { return v == 3 * NG; }
    static constexpr int n_mfma(int v) { return shared_of(v) ? 2 * NSH : 4; }
    static constexpr int acc_of(int v) { return shared_of(v) ? 3 : v % 3; }
    static constexpr int off_of(int v, int j) {        // LDS float offset of MFMA j's operand relative to the sub-chunk and the lane base
        if (!shared_of(v)) return 2 * j * ROW + (v % 3) * WR + DIL * (v / 3);
        return 2 * (j % 2) * ROW + V_INF + DIL * (j / 2);
    }
};

#ifndef FV_X_WINO44_OCC
#define FV_X_WINO44_OCC 3
#endif
// VAR: 0 = any layer of whole 64-row blocks; 1 = the 64-channel layers (eight 8-channel blocks, one row block: a constant trip count, and their own rows in rocprofv3's
// per-kernel statistics — the C = 128 stage's launches have the same grid at B = 32).  (Layers with an odd number of 32-row tiles — BigVGAN's C = 96 — stay on
// conv_wino_kernel: an instance whose last workgroup idles two waves measured no faster there, and the test for it in the common instance cost 3 %: LOG R4.15)
// MT: 32-row tiles per wave.  1: workgroup = 64 rows, three waves per SIMD.  2: workgroup = 128 rows — the same staging feeds twice the products, each operand read from LDS
// serves two MFMAs — with 128 accumulator registers per wave, two waves per SIMD.
// PRE: the activation in front of the conv at compile time — 0 none (c2 of a pair: its SiLU sits in c1's epilogue), 1 SiLU (c1), 2 p.pre_act at run time
// (leaky-ReLU: RefineGAN).  With the switch gone the staging is one basic block, and the compiler's vmcnt bookkeeping across it stays exact (round 5).
// QR (D = 1 only): the row-split epilogue with 16-byte stores; the host launches it when the layer qualifies (wino44_quad_rows) — an instance of its own: with
// both lean epilogues in one kernel the register allocator spills 40 - 70 values.
// PERS (round 6): a persistent grid — gridDim.x workgroups (two per CU) walk the launch's p.wg_total tiles with stride gridDim.x (tile = the workgroup index of
// the one-tile-per-workgroup form: same (clip, row block, column tile) decomposition, neighbouring tiles of a clip still behind one L2).  What it buys is the
// item boundary: the next tile's address plan is formed and its first activation chunk requested BEFORE the current tile's epilogue, so the ~3 us between a
// workgroup's start and its first staged chunk (dispatch, plan, an HBM round trip) overlap the epilogue's own memory waits instead of following them.
template <int KS, int DIL, int VAR, int MT, int PRE, bool QR = false, bool FLAT = false, bool PERS = false>
__global__ __launch_bounds__(256, (MT == 1 ? FV_X_WINO44_OCC : 2)) void conv_wino44_kernel(const ConvParams p) {
    static_assert(!QR || DIL == 1, "the row-split epilogue needs contiguous quads");
    using G = Wino44Geom<KS, DIL>;
    constexpr bool C64 = VAR == 1;
    constexpr int NV = G::NV, NBQ = G::NBQ, WR = G::WR, ROW = G::ROW, SUBS = G::SUBS, CH = G::CH, RPW = G::RPW, NE = G::NE;
    __shared__ float xs[G::XS_F];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, h = wave & 1;
    // workgroup b runs on XCD b % 8: neighbouring tiles of a clip behind one L2 (conv_wino_impl.h)
    int item = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (item >= p.wg_total) return;
    // the tile this workgroup is on (wave-uniform; PERS: re-formed per tile)
    int m_blk, b, n0;
    const float* __restrict__ xb;
    auto set_item = [&](int it) __attribute__((always_inline)) {
        const int n_tile = it % p.n_tiles;
        it /= p.n_tiles;
        m_blk = it % p.m_blks;
        b = it / p.m_blks;
        n0 = n_tile * NBQ;
        xb = p.x + (long long)b * p.x_bstride;
    };
    set_item(item);

#ifdef FV_X_CONV_TS
    if (p.dbg_ts && threadIdx.x == 0) p.dbg_ts[(long long)blockIdx.x * 16 + 15] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) | ((long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);   // HW_ID, XCC_ID
#endif
#ifdef FV_X_W44_STAGGER   // experiment (LOG R6.x): the first round's second workgroup of every CU starts FV_X_W44_STAGGER x 3.4 us late
    if (blockIdx.x >= 256 && blockIdx.x < 512) {
#pragma unroll
        for (int i = 0; i < FV_X_W44_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    FV_CV_STAMP(0);
    f32x16 acc[MT][4];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][a][r] = 0.f;
    };
    zero_acc();

    // ---- staging plan (conv_wino4_impl.h): this wave owns channel rows wave * RPW .. + RPW - 1 of every chunk; lane element i = quad column
    // (lane + 64 i) % WR of row (lane + 64 i) / WR; byte offsets relative to the chunk's first row, 0xFFFFFFFF outside [0, Tin) ----
    unsigned vo[NE][4];
    int lo[NE];               // LDS float offset of (row, column) inside the chunk buffer
    auto plan = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        int e = lane + 64 * i;
        e = e < RPW * WR ? e : RPW * WR - 1;
        const int rr = e / WR, c = e - rr * WR;
        int n = n0 + c;
        // flattened column axis (p.col_S > 0, b == 0): virtual column -> (clip, quad column); a clip's columns are followed by NG D columns of its own halo —
        // its last outputs' tap groups read those, never the next clip's first columns — see conv_layer.hip
        int xbo = 0;
        bool bok = true;
        if constexpr (FLAT) {
            const int bb = n / p.col_S;
            n -= bb * p.col_S;
            xbo = bb * (int)p.x_bstride;
            bok = bb < p.col_batch;
        }
        const int q = n / DIL;
        const int t0 = 4 * DIL * q + (n - q * DIL) - p.pad_l;
        const int row = wave * RPW + rr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + j * DIL;
            vo[i][j] = (bok && t >= 0 && t < p.Tin) ? (unsigned)(xbo + row * p.Tin + t) * 4u : 0xFFFFFFFFu;
        }
        lo[i] = row * ROW + c;
    }
    };
    plan();
    // Two register sets of raw activations, chunks c and c + 1 (round 5): a chunk's loads are issued at the END of the matrix loop two chunks earlier, not
    // at the start of the previous one.  Loads return in order (one vmcnt counter), so every weight fragment requested after the activation loads waits for
    // them too — HBM / MALL latency against the weights' L2 latency: at the loop's start that put the activation latency minus three steps of matrix work
    // (DA fragments in flight) on every chunk's critical path (-6 % of the launch with the loads removed: tools/ablate_w44.sh, LOG R5.1).  Behind the loop
    // the next weight wait is a whole staging phase + barrier + DA steps away.
    float sx_a[4 * NE], sx_b[4 * NE];   // [j * NE + i]
    // (always issued, never under a branch — the compiler's vmcnt bookkeeping falls back to "wait for everything" behind a conditional load: a chunk past
    // the layer's last one gets an empty descriptor, its loads return 0 without touching memory)
    bool live = true;   // (PERS: false once the walk is past the launch's last tile — the prefetch of "the next tile" gets an empty descriptor)
    auto load_chunk = [&](int c, float (&sx)[4 * NE]) {
        const int cbase = c * CH;
        const long long span = p.x_bstride - (long long)cbase * p.Tin;
        const long long rows = live ? (long long)(p.Cin - cbase) * p.Tin : 0;
        // (flattened columns: the descriptor spans the clips a tile may touch — whole chunks only, host-checked: a part-filled last chunk would read the
        //  next clip's rows where the per-clip form reads zeros)
        const long long lim = FLAT ? (rows > 0 ? (long long)p.col_batch * p.x_bstride - (long long)cbase * p.Tin : 0) : (rows < span ? rows : span);
        const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(xb + (long long)(lim > 0 ? cbase : 0) * p.Tin, lim > 0 ? (unsigned)(lim * 4) : 0u);
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sx[j * NE + i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, vo[i][j], 0, 0));
    };
    // store_chunk: one chunk staged as a PHASE of its own (activation -> raw phases to the scratch columns -> neighbours back -> seven planes).  Since round 6
    // only chunk 0 of a workgroup (and every chunk of the PRE == 2 instances, whose run-time activation switch would put branches into the matrix loop) is
    // staged this way; the others are staged piece by piece INSIDE the previous chunk's matrix loop (stage_A / stage_R / stage_T below).
    auto store_chunk = [&](float* dst, float (&sx)[4 * NE]) {
        if constexpr (PRE != 0) act_apply_all(sx, PRE == 1 ? (int)FV_ACT_SILU : p.pre_act, p.slope);   // act(0) == 0 keeps the zero padding
        float x4[NE], x5[NE], x6[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) dst[lo[i] + G::X_OFF + j * WR] = sx[j * NE + i];
        // the neighbours (column + D of the same row) were written by this wave: its LDS operations execute in order, the fence
        // only keeps the compiler from moving the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            x4[i] = dst[lo[i] + G::X_OFF + DIL];
            x5[i] = dst[lo[i] + G::X_OFF + WR + DIL];
            x6[i] = dst[lo[i] + G::X_OFF + 2 * WR + DIL];
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const float x0 = sx[i], x1 = sx[NE + i], x2 = sx[2 * NE + i], x3 = sx[3 * NE + i];
            const float eh = fmaf(4.0f, x0, fmaf(-5.0f, x2, x4[i])), oh = fmaf(4.0f, x1, fmaf(-5.0f, x3, x5[i]));          // a = 1/2
            const float e1 = fmaf(-4.25f, x2, x4[i]) + x0, o1 = fmaf(-4.25f, x3, x5[i]) + x1;                                // a = 1
            const float e2 = fmaf(0.25f, x0, fmaf(-1.25f, x2, x4[i])), o2 = fmaf(0.25f, x1, fmaf(-1.25f, x3, x5[i]));        // a = 2
            dst[lo[i]] = fmaf(0.5f, eh, oh);
            dst[lo[i] + WR] = fmaf(-0.5f, eh, oh);
            dst[lo[i] + 2 * WR] = o1 + e1;
            dst[lo[i] + 3 * WR] = o1 - e1;
            dst[lo[i] + 4 * WR] = fmaf(2.0f, e2, o2);
            dst[lo[i] + 5 * WR] = fmaf(-2.0f, e2, o2);
            dst[lo[i] + G::V_INF] = fmaf(5.25f, x2 - x4[i], x6[i] - x0);
        }
    };
    // The same staging in three pieces per lattice element, issued as short bursts between the MFMAs of the PREVIOUS chunk's matrix loop (round 6).  Why: a
    // vector instruction costs the matrix pipe ~4 cycles when it comes from the wave that owns the MFMA stream, in a burst, and 12 - 16 when it comes from the
    // partner wave of the SIMD between that wave's MFMAs (tools/ubench/mfma_mix.hip, LOG R3.1) — and a staging PHASE is exactly that: while one workgroup
    // stages, the other workgroup of the CU is in its matrix loop (profiles/r06b_w44_timeline.txt: co-resident workgroups alternate, chunk period 11.0 us
    // for 2 x 4.8 us of MFMAs).  It also takes the staging latency (LDS round trip of the neighbour columns, the barrier) off a lone workgroup's path.
    //   A_i: activation of the element's four samples, raw phases X0..X2 -> scratch columns        (step i)
    //   R_i: neighbours x4 x5 x6 (column + D of the same row, written by this wave: in-order LDS)    (step NE + i)
    //   T_i: transform, seven planes -> the OTHER chunk buffer                                      (step NE + 1 + i)
#ifndef FV_X_W44_INLOOP
#define FV_X_W44_INLOOP 1
#endif
    constexpr bool INLOOP = FV_X_W44_INLOOP && PRE != 2 && !(FLAT && DIL == 3 && PRE == 1);   // (the flattened D = 3 instance behind a SiLU spills 219 registers with the pieces in its loop)
    [[maybe_unused]] float xn[NE][3];
    // FV_X_W44_SURR = N (timing experiment, LOG R6.5; results unchanged): the PRE == 0 instances of layers with one row block (M <= 128) carry, per staged
    // sample, N v_fma_f32 (x * 1 + 0) and two v_cos_f32 in their in-loop staging piece — the vector issue of alias_free_torch's Activation1d(SnakeBeta)
    // (38 FMA-class + 2 transcendental per output sample: small_kernels.hip aa_snake_tile) WITHOUT its LDS traffic, its halo recompute or its barriers: a
    // lower bound of what a consumer-side fusion of that activation would cost these kernels.
#ifndef FV_X_W44_SURR
#define FV_X_W44_SURR 0
#endif
    auto surrogate = [&](float (&sx)[4 * NE], int i) __attribute__((always_inline)) {
        if constexpr (FV_X_W44_SURR > 0 && PRE == 0) {
            if (p.M <= 128) {
                const float one = 1.0f, zero = 0.0f;
#pragma unroll
                for (int q = 0; q < FV_X_W44_SURR; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(sx[j * NE + i]) : "v"(one), "v"(zero));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float c0, c1;
                    asm volatile("v_cos_f32 %0, %1" : "=v"(c0) : "v"(sx[j * NE + i]));
                    asm volatile("v_cos_f32 %0, %1" : "=v"(c1) : "v"(c0));
                }
            }
        }
    };
    auto stage_A = [&](float* dst, float (&sx)[4 * NE], auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        surrogate(sx, i);
        if constexpr (PRE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sx[j * NE + i] = sx[j * NE + i] * __builtin_amdgcn_rcpf(1.0f + __expf(-sx[j * NE + i]));   // silu(0) == 0 keeps the zero padding
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) dst[lo[i] + G::X_OFF + j * WR] = sx[j * NE + i];
    };
    auto stage_R = [&](const float* dst, auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if constexpr (i == 0) {   // (every A piece precedes the first R piece in program order; the fences keep the compiler from reordering across them)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        xn[i][0] = dst[lo[i] + G::X_OFF + DIL];
        xn[i][1] = dst[lo[i] + G::X_OFF + WR + DIL];
        xn[i][2] = dst[lo[i] + G::X_OFF + 2 * WR + DIL];
    };
    auto stage_T = [&](float* dst, const float (&sx)[4 * NE], auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        const float x0 = sx[i], x1 = sx[NE + i], x2 = sx[2 * NE + i], x3 = sx[3 * NE + i], x4 = xn[i][0], x5 = xn[i][1], x6 = xn[i][2];
        const float eh = fmaf(4.0f, x0, fmaf(-5.0f, x2, x4)), oh = fmaf(4.0f, x1, fmaf(-5.0f, x3, x5));          // a = 1/2
        const float e1 = fmaf(-4.25f, x2, x4) + x0, o1 = fmaf(-4.25f, x3, x5) + x1;                              // a = 1
        const float e2 = fmaf(0.25f, x0, fmaf(-1.25f, x2, x4)), o2 = fmaf(0.25f, x1, fmaf(-1.25f, x3, x5));      // a = 2
        dst[lo[i]] = fmaf(0.5f, eh, oh);
        dst[lo[i] + WR] = fmaf(-0.5f, eh, oh);
        dst[lo[i] + 2 * WR] = o1 + e1;
        dst[lo[i] + 3 * WR] = o1 - e1;
        dst[lo[i] + 4 * WR] = fmaf(2.0f, e2, o2);
        dst[lo[i] + 5 * WR] = fmaf(-2.0f, e2, o2);
        dst[lo[i] + G::V_INF] = fmaf(5.25f, x2 - x4, x6 - x0);
    };

    int mt0 = (m_blk * 2 + wm) * MT;   // first 32-row tile of this wave
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    int wbase = __builtin_amdgcn_readfirstlane((mt0 * 2 + h) * (p.nchunk * NV * 1024));   // bytes per (m-tile, half): nchunk * NV fragments of 1 KiB
    const int wtile = __builtin_amdgcn_readfirstlane(2 * p.nchunk * NV * 1024);   // bytes from one 32-row tile's fragments to the next one's (same half)
    auto load_a = [&](int i, int goff_b) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wbase + i * wtile + goff_b, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    const int b_lane_g = (lane >> 5) * ROW + (lane & 31) + h * G::HALF_G;       // own planes
    const int b_lane_s = (lane >> 5) * ROW + (lane & 31) + h * (4 * ROW);       // shared plane: this half's channel pairs 2 h, 2 h + 1

    constexpr int STEPS = SUBS * NV;
#ifndef FV_X_WINO44_DA
#define FV_X_WINO44_DA 3
#endif
    constexpr int DA = FV_X_WINO44_DA;   // weight prefetch distance in fragments
    float4 aq[MT][DA + 1];
    float b_cur[4], b_nxt[4];
    // chunks in pairs (the two register sets alternate at compile time): p.nchunk is a multiple of four 8-channel blocks — the packed weights of the padding are zero
    int nch = C64 ? 8 / SUBS : (p.nchunk / SUBS + 1) / 2 * 2;   // (compile-time counts for C = 128 / 256 too: no difference in the step)
    if constexpr (C64 && INLOOP) asm volatile("" : "+s"(nch));   // (opaque: fully unrolled, the in-loop staging pieces of four chunk pairs cost these instances 24 spilt registers)
    static_assert(!INLOOP || STEPS >= 2 * NE + 1, "the in-loop staging needs 2 NE + 1 matrix steps per chunk");
    // One matrix step.  STG (INLOOP only): the staging pieces of chunk c + 1 that ride in this chunk's steps.
    auto mfma_loop = [&](int c, float* xsb, float* xsn, [[maybe_unused]] float (&sxn)[4 * NE]) __attribute__((always_inline)) {
        const int gchunk_b = __builtin_amdgcn_readfirstlane((c * STEPS + DA) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) b_cur[j] = xsb[b_lane_g + G::off_of(0, j)];
        static_for<STEPS>([&](auto st_c) __attribute__((always_inline)) {
            constexpr int st = decltype(st_c)::value;
            constexpr int v = st % NV;
            constexpr int A = G::acc_of(v), NM = G::n_mfma(v);
            constexpr int sub_n = (st + 1) / NV, v_n = (st + 1) % NV;
            constexpr int NM_n = st + 1 < STEPS ? G::n_mfma(v_n) : 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m < NM) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const float av = m == 0 ? aq[i][0].x : m == 1 ? aq[i][0].y : m == 2 ? aq[i][0].z : aq[i][0].w;
                        acc[i][A] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_cur[m], acc[i][A], 0, 0, 0);
                    }
                }
                // one weight fragment per row tile (DA fragments ahead) and the next step's operands, spread over this step's MFMAs
                if (m < MT) aq[m][DA] = load_a(m, gchunk_b + st * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < NM_n && (j * NM) / 4 == m && m < NM)
                        b_nxt[j] = xsb[(G::shared_of(v_n) ? b_lane_s : b_lane_g) + sub_n * kChunk * ROW + G::off_of(v_n, j)];
                if (m < NM) __builtin_amdgcn_sched_barrier(0);
                if constexpr (INLOOP) {
                    // the next chunk's staging, one burst per step behind the step's second MFMA group (the wave's own MFMAs of this step are in flight)
                    if (m == 1) {
                        if constexpr (st < NE) stage_A(xsn, sxn, std::integral_constant<int, st>{});
                        if constexpr (st >= NE && st < 2 * NE) stage_R(xsn, std::integral_constant<int, st - NE>{});
                        if constexpr (st > NE && st <= 2 * NE) stage_T(xsn, sxn, std::integral_constant<int, st - NE - 1>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int d = 0; d < DA; ++d) aq[i][d] = aq[i][d + 1];
            if constexpr (st + 1 < STEPS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) b_cur[j] = b_nxt[j];
            }
        });
    };
    static_assert(!PERS || INLOOP, "the persistent walk is built on the in-loop staging form");
    // Register sets: sx_t holds chunk 0 (staged as a phase before the first matrix loop; dead afterwards — the accumulators are not live yet; PERS: reloaded with
    // the NEXT tile's chunk 0 before the epilogue), sx_b the odd chunks, sx_a the even ones from chunk 2 on.  Chunk c + 1 is staged inside chunk c's matrix
    // loop; the set it frees is reloaded with chunk c + 3 behind that loop's last weight request (loads return in order: LOG R5.2), two matrix loops before
    // it is needed.
    [[maybe_unused]] float sx_t[4 * NE];
    if constexpr (INLOOP) load_chunk(0, sx_t);
    for (;;) {   // (one pass unless PERS)
    if constexpr (INLOOP) {
#pragma unroll
        for (int d = 0; d < DA; ++d)
#pragma unroll
            for (int i = 0; i < MT; ++i) aq[i][d] = load_a(i, d * 1024);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(1, sx_b);
        load_chunk(2, sx_a);
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(xs, sx_t);
        auto chunk_body = [&](int c, float (&sxn)[4 * NE]) __attribute__((always_inline)) {
            float* xsb = xs + (c & 1) * (CH * ROW);
            float* xsn = xs + ((c + 1) & 1) * (CH * ROW);
            __syncthreads();   // chunk c is staged (by every wave, in the previous loop); every wave is past its reads of the other buffer (chunk c - 1)
            if (c < 12) FV_CV_STAMP(1 + c);
            mfma_loop(c, xsb, xsn, sxn);
            load_chunk(c + 3, sxn);   // (past the layer's last chunk: an empty descriptor, zeros — the piece staged from it lands in the buffer nobody reads)
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int c = 0; c < nch; c += 2) {
            chunk_body(c, sx_b);
            chunk_body(c + 1, sx_a);
        }
    } else {
        // (the same issue order as around the loop's back edge — set a, weight prefetch, set b: where the two paths into the loop header disagree the compiler
        //  assumes the worst of both and waits for set b at the end of the first staging phase)
        load_chunk(0, sx_a);
#pragma unroll
        for (int d = 0; d < DA; ++d)
#pragma unroll
            for (int i = 0; i < MT; ++i) aq[i][d] = load_a(i, d * 1024);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(1, sx_b);
        __builtin_amdgcn_sched_barrier(0);
        auto chunk_body = [&](int c, float (&sx)[4 * NE]) __attribute__((always_inline)) {
            float* xsb = xs + (c & 1) * (CH * ROW);
            store_chunk(xsb, sx);
            __syncthreads();
            if (c < 12) FV_CV_STAMP(1 + c);
            mfma_loop(c, xsb, xsb, sx);
            // the chunk after next, behind every weight request of this loop (see sx_a / sx_b above)
            load_chunk(c + 2, sx);
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int c = 0; c < nch; c += 2) {
            chunk_body(c, sx_a);
            chunk_body(c + 1, sx_b);
        }
    }

    FV_CV_STAMP(13);
    // the tile the epilogue stores; PERS: the walk moves on first — the next tile's plan, weights' base and first activation chunk (an empty descriptor past
    // the last tile) are on their way while this tile's partial sums are exchanged and stored
    const int e_n0 = n0, e_b = b, e_mt0 = mt0;
    if constexpr (PERS) {
        item += (int)gridDim.x;
        live = item < p.wg_total;
        set_item(live ? item : 0);
        plan();
        mt0 = (m_blk * 2 + wm) * MT;
        wbase = __builtin_amdgcn_readfirstlane((mt0 * 2 + h) * (p.nchunk * NV * 1024));
        load_chunk(0, sx_t);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- output transform.  y_j = sum over points a^j m(a) (+ m(∞) for j = 3); each half forms the partial sums of its planes for all four
    // outputs, keeps two of them in place (A in acc 0, B in acc 1) and passes the other two to its partner:
    //   half 0 (m(1/2) m(-1/2) m(1), its part of m(∞)):  s = m(1/2) + m(-1/2), d = m(1/2) - m(-1/2)
    //       keeps  y0: s + m1,  y1: d/2 + m1        sends  y2: s/4 + m1,  y3: d/8 + m1 + m(∞)
    //   half 1 (m(-1) m(2) m(-2), its part of m(∞)):      s = m(2) + m(-2),     d = m(2) - m(-2)
    //       sends  y0: s + m(-1),  y1: 2 d - m(-1)  keeps  y2: 4 s + m(-1),  y3: 8 d - m(-1) + m(∞) ----
    int n = e_n0 + (lane & 31);
    int ybo = 0;                    // flattened columns: the clip's offset in y / the residual (elements); a column past the last clip gets a position past every row's end
    if constexpr (FLAT) {
        const int bb = n / p.col_S;
        n -= bb * p.col_S;
        ybo = bb * (int)p.y_bstride;
        if (bb >= p.col_batch) n = 1 << 26;
    }
    const unsigned yspan = (unsigned)((FLAT ? (long long)p.col_batch * p.y_bstride : p.y_bstride) * 4);
    const int q = n / DIL;
    const int ta = 4 * DIL * q + (n - q * DIL) + 2 * h * DIL, tb = ta + DIL;   // half 0: t0, t0 + D; half 1: t0 + 2D, t0 + 3D
    const float* pa = xs + (wave ^ 1) * 2048 + lane;          // partner's partial sum of this half's first output
    const float* pb = xs + (wave ^ 1) * 2048 + 1024 + lane;   // ... and second
    // the common case — whole 32-row tiles, bias [+ residual] [+ post-activation], plain store — without per-element offset registers (conv_wino_impl.h)
    const bool lean = p.M % 32 == 0 && p.gamma == nullptr && p.out_mode == OUT_SET && p.acc_scale == 1.0f;
    // D = 1 (every c2, a third of the c1 launches): a quad's four outputs are CONTIGUOUS samples, so the halves split a tile's ROWS instead of a quad's
    // outputs — half h keeps accumulator registers 8 h .. 8 h + 7 (rows 16 h .. 16 h + 15 of the 32-row tile) with all four outputs of its quad column and
    // sends the other eight — and every lane moves 16 bytes per row: 8 residual loads + 8 stores per tile, each covering whole 64-byte lines, instead of
    // 32 + 32 single floats on a 16-byte stride.  The stores and residual loads were 11 - 18 % of a c2 launch (tools/ablate_w44.sh mask 128, LOG R5.3).
    // Same sums in the same order as the split by outputs: y_j = (this half's partial) + (the partner's), fp32 addition commutes.
    if constexpr (QR) {
        {
            const unsigned span = yspan;
            const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)e_b * p.y_bstride, span);
            const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(p.res ? p.res + (long long)e_b * p.y_bstride : p.y, span);
            const __amdgpu_buffer_rsrc_t brs = uniform_rsrc(p.bias, (unsigned)(p.M * 4));
            const int mrow = 4 * (lane >> 5);
            const int t4 = 4 * n;                                    // first of the quad's four samples
            const unsigned vq = t4 < p.N ? (unsigned)(ybo + mrow * p.N + t4) * 4u : 0xFFFFFFFFu;
            const bool has_res = p.res != nullptr;
            const float* pq = xs + (wave ^ 1) * 2048 + lane;        // partner's partials of the rows this half keeps: [kept register][output] x 64 lanes
            // Round 6: the bias and residual operands of every row of a tile are requested at once, before the tile's exchange barriers — the epilogue used to
            // request four rows, wait, store, four times per launch: ~9 us of a lone workgroup's life, 19 us beside a partner in its matrix loop
            // (profiles/r06b_w44_timeline.txt), most of it memory latency.  The weight ring and the staging sets are dead by now.
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int mt = e_mt0 + i;
                // (a tile's eight rows: requested before the tile's exchange, consumed behind its two barriers — both tiles at once spilt 32 registers)
                float biasq[1][8];
                u32x4 rqq[1][8];
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int row_s = mt * 32 + (rr & 3) + 8 * (2 * h + (rr >> 2));
                    biasq[0][rr] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, mrow * 4, row_s * 4, 0));
                    if (has_res) rqq[0][rr] = __builtin_amdgcn_raw_buffer_load_b128(rrs, vq, (int)((unsigned)row_s * (unsigned)p.N * 4u), 0);
                }
                __syncthreads();
                // partial sums of all four outputs, in place: register r of accumulator plane j becomes y_j's partial; the registers of the OTHER half's rows go
                // to the exchange area (H at compile time under a wave-uniform branch: no selects, no copies)
                auto phase1 = [&](auto h_c) __attribute__((always_inline)) {
                    constexpr int H = decltype(h_c)::value;
                    float* ex = xs + wave * 2048 + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float y0, y1, y2, y3;
                        if constexpr (H == 0) {
                            const float sm = acc[i][0][r] + acc[i][1][r], df = acc[i][0][r] - acc[i][1][r], m1 = acc[i][2][r];
                            y0 = sm + m1;
                            y1 = fmaf(0.5f, df, m1);
                            y2 = fmaf(0.25f, sm, m1);
                            y3 = fmaf(0.125f, df, m1) + acc[i][3][r];
                        } else {
                            const float sm = acc[i][1][r] + acc[i][2][r], df = acc[i][1][r] - acc[i][2][r], mm = acc[i][0][r];
                            y0 = sm + mm;
                            y1 = fmaf(2.0f, df, -mm);
                            y2 = fmaf(4.0f, sm, mm);
                            y3 = fmaf(8.0f, df, -mm) + acc[i][3][r];
                        }
                        if ((r >> 3) == H) {   // (r is an unrolled loop index: folded at compile time)
                            acc[i][0][r] = y0;
                            acc[i][1][r] = y1;
                            acc[i][2][r] = y2;
                            acc[i][3][r] = y3;
                        } else {
                            ex[((r & 7) * 4 + 0) * 64] = y0;
                            ex[((r & 7) * 4 + 1) * 64] = y1;
                            ex[((r & 7) * 4 + 2) * 64] = y2;
                            ex[((r & 7) * 4 + 3) * 64] = y3;
                        }
                    }
                };
                if (h == 0) phase1(std::integral_constant<int, 0>{}); else phase1(std::integral_constant<int, 1>{});
                __syncthreads();
                // kept register rr <-> r = 8 H + rr <-> row mt * 32 + (r & 3) + 8 * (r >> 2) + mrow; four rows at a time (their operands requested together)
                auto phase2 = [&](auto h_c) __attribute__((always_inline)) {
                    constexpr int H = decltype(h_c)::value;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int rr = 4 * g + k, r = 8 * H + rr;
                            const int row_s = mt * 32 + k + 8 * (2 * H + g);
                            const float bias_k = biasq[0][rr];
                            float o[4];
                            // (the split by outputs forms  own + partner  in half 0 for j = 0, 1 and in half 1 for j = 2, 3: the same two addends)
                            o[0] = fmaf(acc[i][0][r] + pq[(rr * 4 + 0) * 64], 1.0f, bias_k);
                            o[1] = fmaf(acc[i][1][r] + pq[(rr * 4 + 1) * 64], 1.0f, bias_k);
                            o[2] = fmaf(acc[i][2][r] + pq[(rr * 4 + 2) * 64], 1.0f, bias_k);
                            o[3] = fmaf(acc[i][3][r] + pq[(rr * 4 + 3) * 64], 1.0f, bias_k);
                            if (has_res) {
                                o[0] += __uint_as_float(rqq[0][rr].x);
                                o[1] += __uint_as_float(rqq[0][rr].y);
                                o[2] += __uint_as_float(rqq[0][rr].z);
                                o[3] += __uint_as_float(rqq[0][rr].w);
                            }
                            act_apply_all(o, p.post_act, p.slope);
                            u32x4 v;
                            v.x = __float_as_uint(o[0]);
                            v.y = __float_as_uint(o[1]);
                            v.z = __float_as_uint(o[2]);
                            v.w = __float_as_uint(o[3]);
                            __builtin_amdgcn_raw_buffer_store_b128(v, yrs, vq, (int)((unsigned)row_s * (unsigned)p.N * 4u), 0);
                        }
                    }
                };
                if (h == 0) phase2(std::integral_constant<int, 0>{}); else phase2(std::integral_constant<int, 1>{});
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int mt = e_mt0 + i;
        __syncthreads();   // every wave is past its last operand read (i = 0: the chunk buffers become the exchange area) / past the previous tile's exchange
        {
            float* ex = xs + wave * 2048 + lane;
            if (h == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sm = acc[i][0][r] + acc[i][1][r], df = acc[i][0][r] - acc[i][1][r], m1 = acc[i][2][r];
                    ex[r * 64] = fmaf(0.25f, sm, m1);
                    ex[1024 + r * 64] = fmaf(0.125f, df, m1) + acc[i][3][r];
                    acc[i][0][r] = sm + m1;
                    acc[i][1][r] = fmaf(0.5f, df, m1);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sm = acc[i][1][r] + acc[i][2][r], df = acc[i][1][r] - acc[i][2][r], mm = acc[i][0][r];
                    ex[r * 64] = sm + mm;
                    ex[1024 + r * 64] = fmaf(2.0f, df, -mm);
                    acc[i][0][r] = fmaf(4.0f, sm, mm);
                    acc[i][1][r] = fmaf(8.0f, df, -mm) + acc[i][3][r];
                }
            }
        }
        __syncthreads();
        if (lean) {
            // all bias and residual operands are requested before the partner's planes are read back
            const unsigned span = yspan;
            const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)e_b * p.y_bstride, span);
            const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(p.res ? p.res + (long long)e_b * p.y_bstride : p.y, span);
            const __amdgpu_buffer_rsrc_t brs = uniform_rsrc(p.bias, (unsigned)(p.M * 4));
            const int mrow = 4 * (lane >> 5);                                   // lane part of the row; + mt * 32 + (r & 3) + 8 * (r >> 2) in SGPRs
            const unsigned va = ta < p.N ? (unsigned)(ybo + mrow * p.N + ta) * 4u : 0xFFFFFFFFu;
            const unsigned vb = tb < p.N ? (unsigned)(ybo + mrow * p.N + tb) * 4u : 0xFFFFFFFFu;
            float bias[16], ra[16], rb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                bias[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, mrow * 4, (mt * 32 + (r & 3) + 8 * (r >> 2)) * 4, 0));
            if (p.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int so = (int)((unsigned)(mt * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)p.N * 4u);   // (< 4 GiB per item: conv_layer_run)
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
                const float y0 = acc[i][0][r] + pa[r * 64];
                const float y1 = acc[i][1][r] + pb[r * 64];
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
                const int so = (int)((unsigned)(mt * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)p.N * 4u);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oa[r]), yrs, va, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ob[r]), yrs, vb, so, 0);
            }
        } else {
            f32x16 out[1][2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                out[0][0][r] = acc[i][0][r] + pa[r * 64];
                out[0][1][r] = acc[i][1][r] + pb[r * 64];
            }
            const int coff[2] = {ta, tb};
            const bool cok[2] = {ta < p.N, tb < p.N};
            conv_epilogue_cols<1, 2>(p, out, e_b, mt, coff, cok, lane);
        }
    }
    }
#ifdef FV_X_CONV_TS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    FV_CV_STAMP(14);
    if constexpr (!PERS) break;
    else {
        if (!live) break;
        zero_acc();
        __syncthreads();   // every wave is past its reads of the exchange area: the chunk buffers are free for the next tile's chunk 0
    }
    }   // tiles of this workgroup
}

// the row-split epilogue: whole quads per row (N % 4 == 0), 16-byte aligned rows of y and the residual, the lean operand set
inline bool wino44_quad_rows(const ConvParams& p) {
    static const bool off = std::getenv("FV_X_W44_NO_QR") != nullptr;   // A/B runs
    return !off && p.dil == 1 && p.M % 32 == 0 && p.gamma == nullptr && p.out_mode == OUT_SET && p.acc_scale == 1.0f && p.N % 4 == 0 && (p.y_bstride & 3) == 0 &&
           (((unsigned long long)p.y | (unsigned long long)(p.res ? p.res : p.y)) & 15ull) == 0;
}

template <int KS, int VAR, int MT, int PRE>
inline bool launch_wino44_kc(const ConvParams& p0, int batch, hipStream_t s) {
    ConvParams p = p0;
    p.wg_total = batch * p.m_blks * p.n_tiles;
    const int grid = (p.wg_total + 7) / 8 * 8;
    // flattened column axis (p.col_S > 0; conv_layer.hip offers it to wino44_flat_instance() shapes only): instances of their own — the per-clip kernels are
    // at their register limits, two more live values spilled 54 - 184 registers in several of them
    if constexpr (MT == 2 && VAR == 0 && PRE != 2) {
        if (p.col_S > 0) {
            switch (p.dil) {
                case 1:
                    if (!wino44_quad_rows(p)) return false;
                    hipLaunchKernelGGL((conv_wino44_kernel<KS, 1, VAR, MT, PRE, true, true>), dim3(grid), dim3(256), 0, s, p);
                    return true;
                case 3:
                    if constexpr (PRE == 1) { hipLaunchKernelGGL((conv_wino44_kernel<KS, 3, VAR, MT, PRE, false, true>), dim3(grid), dim3(256), 0, s, p); return true; }
                    return false;
                case 5:
                    if constexpr (PRE == 1) { hipLaunchKernelGGL((conv_wino44_kernel<KS, 5, VAR, MT, PRE, false, true>), dim3(grid), dim3(256), 0, s, p); return true; }
                    return false;
                default: return false;
            }
        }
    }
    if (p.col_S > 0) return false;
#ifndef FV_X_W44_PERS
#define FV_X_W44_PERS 0
#endif
    // the persistent walk: launches of more tiles than the chip holds workgroups (the in-loop staging instances; per-clip tiling)
    if constexpr (FV_X_W44_PERS && PRE != 2) {
        static const bool off = std::getenv("FV_X_W44_NO_PERS") != nullptr;   // A/B runs
        const int slots = num_cus() * (MT == 2 ? 2 : FV_X_WINO44_OCC) / 8 * 8;
        if (!off && slots >= 8 && p.wg_total > slots) {
            switch (p.dil) {
                case 1:
                    if (wino44_quad_rows(p)) hipLaunchKernelGGL((conv_wino44_kernel<KS, 1, VAR, MT, PRE, true, false, true>), dim3(slots), dim3(256), 0, s, p);
                    else hipLaunchKernelGGL((conv_wino44_kernel<KS, 1, VAR, MT, PRE, false, false, true>), dim3(slots), dim3(256), 0, s, p);
                    return true;
                case 3: hipLaunchKernelGGL((conv_wino44_kernel<KS, 3, VAR, MT, PRE, false, false, true>), dim3(slots), dim3(256), 0, s, p); return true;
                case 5: hipLaunchKernelGGL((conv_wino44_kernel<KS, 5, VAR, MT, PRE, false, false, true>), dim3(slots), dim3(256), 0, s, p); return true;
                default: return false;
            }
        }
    }
    switch (p.dil) {
        case 1:
            if (wino44_quad_rows(p)) hipLaunchKernelGGL((conv_wino44_kernel<KS, 1, VAR, MT, PRE, true>), dim3(grid), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((conv_wino44_kernel<KS, 1, VAR, MT, PRE>), dim3(grid), dim3(256), 0, s, p);
            return true;
        case 3: hipLaunchKernelGGL((conv_wino44_kernel<KS, 3, VAR, MT, PRE>), dim3(grid), dim3(256), 0, s, p); return true;
        case 5: hipLaunchKernelGGL((conv_wino44_kernel<KS, 5, VAR, MT, PRE>), dim3(grid), dim3(256), 0, s, p); return true;
        default: return false;
    }
}

// one translation unit per (kernel size, PRE): conv_wino44_k{7,11}_{none,silu,any}.hip
template <int KS, int PRE>
inline bool launch_wino44_k(const ConvParams& p, int rows, int batch, hipStream_t s) {
    if (rows == 128) return launch_wino44_kc<KS, 0, 2, PRE>(p, batch, s);
    return (p.Cin == 64 && p.M == 64) ? launch_wino44_kc<KS, 1, 1, PRE>(p, batch, s) : launch_wino44_kc<KS, 0, 1, PRE>(p, batch, s);
}

}  // namespace fv
