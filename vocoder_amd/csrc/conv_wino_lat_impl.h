// Latency variant of the Winograd F(2,3) conv (conv_wino_impl.h) for launches that cannot fill the chip — the single clip of the
// reference's inference call (fish_vocoder/test.py:88-90: one utterance per forward), small batches:
// the dilated "same" Conv1d of a ResBlock / AMPBlock (hifigan.py:101-108), C_in = C_out = C in {64, 128, 256, ...}, k = 3 / 7 / 11.
//
// What bounds such a launch is the length of one workgroup's dependent MFMA chain, not throughput: v_mfma_f32_32x32x2_f32 retires K = 2 per
// 64 cycles, so a 32 x 32 output tile of the C = 256, k = 11 layer is 1408 matrix instructions deep (352 per wave with K split four ways:
// 9.4 us at 2.4 GHz — conv_mfma_splitk_kernel, 23 us per launch in the single-clip forward of round 3).  Here
//   * v_mfma_f32_16x16x4_f32: K = 4 per 32 cycles, 16 x 16 tiles — four times the workgroups, a quarter of the chain per tile;
//   * Winograd tap groups: 16 / 10 / 4 products per output pair instead of 22 / 14 / 6;
//   * the four waves split K by 8-channel blocks (wave w owns blocks w, w + 4, ...) and are INDEPENDENT until the final reduction: each
//     stages, transforms and multiplies its own blocks in a wave-private LDS region (LDS operations of one wave execute in order, so the
//     E / O -> d-plane transform and the reuse of the region need no barrier at all).
// One workgroup = 16 output rows x 16 NT output pairs (pair lattice of conv_wino_impl.h: pair column n = q D + r <-> t = 2 D q + r, t + D).
// Weights: the layer's d_wpwl — Winograd-transformed, one float4 per lane = (virtual taps 2 vp, 2 vp + 1) x (channel quads 0, 1) of an
// 8-channel block, i.e. four 16x16x4 MFMAs on two different accumulator planes (no back-to-back dependent pair).
#pragma once
#include "pair_wino_impl.h"

namespace fv {

template <int KS, int DIL, int NT, int MT = 1>
struct WLGeom {
    static constexpr int NG = (KS + 1) / 4, NS = (KS - 3) / 4, NV = 4 * NG + 2 * NS, NF = NV / 2;   // fragments (tap pairs) per block
    static constexpr int NBP = 16 * NT;                       // output pairs per workgroup
    static constexpr int WD = NBP + 2 * DIL * (NG - 1), WR = WD + DIL;
    static constexpr int NQ = (WR + DIL - 1) / DIL;           // 2 D-sample blocks staged
    static constexpr int NP = 2 * DIL * NQ;                   // staged positions per channel row
    static constexpr int PW = pw_up(DIL * NQ, 32, 16);        // plane row stride == 16 (mod 32): the 16x16x4 B read (two channel rows per 32 lanes)
    static constexpr int PLANE = 8 * PW;                      // one plane of the wave's 8-channel block: [ch][PW]
    // planes d0 d1 d2 d3 E O; k = 3 with the register transform (K3R): E O only — a third of the LDS, so that more of a single clip's concurrent
    // workgroups (three branches: ~5 per CU wanted, the k = 7 / 11 kernels hold 53 - 61 KB each) are resident together
    static constexpr bool K3R = KS == 3 && FV_X_PW_K3_REG != 0;
    static constexpr int EOP = K3R ? 0 : 4;                   // plane index of E (O follows)
    static constexpr int WAVE_F = (K3R ? 2 : 6) * PLANE;
    static constexpr int XCH_F = 4 * 4 * 2 * NT * MT * 64;    // the reduction's exchange buffer
    static constexpr int LDS_F = 4 * WAVE_F > XCH_F ? 4 * WAVE_F : XCH_F;
    static constexpr int NSLOT = (8 * NP + 63) / 64;          // staged elements per lane and block
    static constexpr int TCG = (WD + 15) / 16;                // transform: 16-column groups
    // weight ring: RA = MU NF fragments (MU = blocks per unrolled loop iteration), prefetch distance RA - 1 fragments of 4 MT NT MFMAs each:
    // at least ~900 cycles of matrix time
#ifndef FV_X_LAT_RING
#define FV_X_LAT_RING 7
#endif
    static constexpr int NEED = (FV_X_LAT_RING + MT * NT - 1) / (MT * NT);
    static constexpr int MU = (NEED + 1 + NF - 1) / NF;
    static constexpr int RA = MU * NF;
    static_assert(16 * TCG <= PW, "the transform's column groups stay inside a plane row");
    static_assert(NV % 2 == 0, "tap pairs");
    static constexpr int acc_of(int v) { return v < 4 * NG ? v % 4 : ((v - 4 * NG) % 2 == 0 ? 0 : 3); }
    static constexpr int off_of(int v) {                      // LDS offset of virtual tap v relative to (channel row, pair column)
        if (v < 4 * NG) return (v % 4) * PLANE + 2 * DIL * (v / 4);
        const int s = (v - 4 * NG) / 2;
        return (v - 4 * NG) % 2 == 0 ? 5 * PLANE + (2 * s + 1) * DIL : 4 * PLANE + (2 * s + 2) * DIL;
    }
};

// The workgroup's work as a device function (tools/experiments/conv_wino_lat3.hip runs the same-depth convs of a stage's three ResBlock
// branches through it in one launch).  lds: WLGeom::LDS_F floats.  wg = workgroup index inside the layer's grid.
template <int KS, int DIL, int NT, int MT>
__device__ __forceinline__ void wino_lat_body(const ConvParams& p, float* __restrict__ lds, int wg) {
    using G = WLGeom<KS, DIL, NT, MT>;
    constexpr int NF = G::NF, NSLOT = G::NSLOT, PW = G::PW, PLANE = G::PLANE, MU = G::MU, RA = G::RA;
    constexpr bool K3R = G::K3R;   // the Winograd input transform in registers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* __restrict__ W = lds + wave * G::WAVE_F;           // this wave's planes
    int bid = wg;
    const int n_tile = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int mt = (bid % p.m_blks) * MT;                     // first of the workgroup's MT 16-row tiles
    const int b = bid / p.m_blks;
    const int n0 = n_tile * G::NBP;
    const int Tin = p.Tin;
    const int nblk = p.Cin >> 3;                              // 8-channel blocks; this wave's: wave, wave + 4, ...
    const int nsteps = (nblk - wave + 3) >> 2;

    // ---- staging plan: element e = lane + 64 i of the block's [8 rows][NP positions] -> global byte offset inside the block (0xFFFFFFFF
    // outside [0, Tin): the load returns 0, and act(0) == 0 is the conv's zero padding) and LDS offset in the E / O planes ----
    unsigned voff[NSLOT];
    int loff[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        int e = lane + 64 * i;
        e = e < 8 * G::NP ? e : 8 * G::NP - 1;                // (surplus lanes repeat the last element: same load, same value, same address)
        const int row = e / G::NP, c = e - row * G::NP;       // staged column c <-> pair column n0 + c' of plane E or O
        const int half = c / (G::NP / 2), cc = c - half * (G::NP / 2);   // first half of the row: E plane, second: O plane
        const int n = n0 + cc;                                // (the lattice is global: n = q D + r, whatever n0 is)
        const int q = n / DIL;
        const int t = 2 * DIL * q + (n - q * DIL) + half * DIL - p.pad_l;
        voff[i] = (t >= 0 && t < Tin) ? (unsigned)(row * Tin + t) * 4u : 0xFFFFFFFFu;
        loff[i] = (G::EOP + half) * PLANE + row * PW + cc;
    }
    const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(p.x + (long long)b * p.x_bstride, (unsigned)(p.x_bstride * 4));
    float sv[NSLOT];
    auto load_block = [&](int blk) __attribute__((always_inline)) {
        const int so = __builtin_amdgcn_readfirstlane(blk * 8 * Tin * 4);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) sv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, voff[i], so, 0));
    };

    // ---- weights: fragment (blk, vp) of m-tile mt at ((mt * nblk + blk) * NF + vp) KiB ----
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    auto wbase_of = [&](int blk) { return __builtin_amdgcn_readfirstlane((mt * nblk + blk) * NF * 1024); };   // (m-tile mt + i: + i * mt_stride)
    const int mt_stride = __builtin_amdgcn_readfirstlane(nblk * NF * 1024);
    float4 aq[RA][MT];
    auto load_w = [&](int base, int f) __attribute__((always_inline)) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, base + f * 1024, 0);
        return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    };

    f32x4w acc[4][MT][NT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[a][i][j] = f32x4w{0.f, 0.f, 0.f, 0.f};

    const int krow = lane >> 4, col = lane & 15;
    const float* bl = W + krow * PW + col;                   // B operand: channel row 4 s + krow, pair column col + 16 jn (+ the tap's offset)
    // transform slots: lanes 0-15 / 16-31 / ... = channel rows krow (+ 4), 16 consecutive columns: conflict-free with PW == 16 (mod 32)
    float* tl = W + krow * PW + col;

    FV_CV_STAMP(0);
    if (nsteps > 0) {
        load_block(wave);
        // Weight ring of RA = MU NF fragments, prefetch distance RA - 1: fragment g of the wave's fragment sequence (block s, fragment f:
        // g = s NF + f) sits in slot g % RA, and while it is consumed the load of fragment g + RA - 1 goes into the slot its predecessor
        // just left.  A single clip has ~1.4 workgroups per CU — nothing else hides an L2 / MALL round trip (the weights of a layer are
        // cold when its launch starts), so the ring covers >= 7 fragments (~900 cycles of matrix time) at every kernel size; the block
        // loop is unrolled MU times so that every slot index is a constant.
        const int last_blk = wave + 4 * (nsteps - 1);
        auto wbase_clamped = [&](int blk) { return wbase_of(blk < last_blk ? blk : last_blk); };
        {
            const int wb0 = wbase_of(wave);
            static_for<RA - 1>([&](auto d_c) {
                constexpr int d = decltype(d_c)::value;
                const int base = d / NF == 0 ? wb0 : wbase_clamped(wave + 4 * (d / NF));
#pragma unroll
                for (int i = 0; i < MT; ++i) aq[d][i] = load_w(base + i * mt_stride, d % NF);
            });
        }
        for (int s0 = 0; s0 < nsteps; s0 += MU) {
            static_for<MU>([&](auto u_c) __attribute__((always_inline)) {
                constexpr int u = decltype(u_c)::value;
                const int s = s0 + u;
                if (s < nsteps) {
                    const int blk = wave + 4 * s;
                    // E / O planes of this block (the previous block's matrix instructions have issued their last LDS read: in-order, no barrier)
                    act_apply_all(sv, p.pre_act, p.slope);
#pragma unroll
                    for (int i = 0; i < NSLOT; ++i) W[loff[i]] = sv[i];
                    if (s + 1 < nsteps) load_block(blk + 4);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // d0 = E - E', d1 = O + E', d2 = E' - O, d3 = O - O'   (E' = E[n + D])
                    float dk3[K3R ? 2 : 1][4][NT];   // k = 3: [quad][virtual tap][n-tile], formed in registers (pair_wino_impl.h, pw_gemm_k3)
                    if constexpr (K3R) {
                        // one tap group: every d value is the B operand of exactly one product per m-tile — no d planes, no second pass through LDS
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int jn = 0; jn < NT; ++jn) {
                                const float* e = bl + G::EOP * PLANE + (4 * q) * PW + 16 * jn;
                                const float E = e[0], E1 = e[DIL], O = e[PLANE], O1 = e[PLANE + DIL];
                                dk3[q][0][jn] = E - E1;
                                dk3[q][1][jn] = O + E1;
                                dk3[q][2][jn] = E1 - O;
                                dk3[q][3][jn] = O - O1;
                            }
                    } else {
#pragma unroll
                        for (int g = 0; g < 2 * G::TCG; ++g) {
                            const int ro = (4 * (g & 1)) * PW + 16 * (g >> 1);
                            const float E = tl[4 * PLANE + ro], E1 = tl[4 * PLANE + ro + DIL], O = tl[5 * PLANE + ro], O1 = tl[5 * PLANE + ro + DIL];
                            tl[ro] = E - E1;
                            tl[PLANE + ro] = O + E1;
                            tl[2 * PLANE + ro] = E1 - O;
                            tl[3 * PLANE + ro] = O - O1;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                    if (s == 0) FV_CV_STAMP(1);
                    // byte offsets of the blocks the prefetch reaches (this one + 1 ... MU ahead; past the wave's last block: that block again)
                    int wbs[MU + 1];
#pragma unroll
                    for (int j = 0; j <= MU; ++j) wbs[j] = wbase_clamped(blk + 4 * j);
                    float b_cur[4][NT], b_nxt[4][NT];
#pragma unroll
                    for (int h = 0; h < 4; ++h)   // h = 2 * tap-of-pair + quad
#pragma unroll
                        for (int jn = 0; jn < NT; ++jn) b_cur[h][jn] = K3R ? dk3[h & 1][h >> 1][jn] : bl[G::off_of(h >> 1) + (4 * (h & 1)) * PW + 16 * jn];
                    static_for<NF>([&](auto f_c) __attribute__((always_inline)) {
                        constexpr int f = decltype(f_c)::value;
                        constexpr int A0 = G::acc_of(2 * f), A1 = G::acc_of(2 * f + 1);
                        constexpr int g = u * NF + f;           // position in the ring's period
#pragma unroll
                        for (int m = 0; m < 4 * NT * MT; ++m) {
                            // order: m-tile fastest, then (v0, q0) (v1, q0) (v0, q1) (v1, q1), then n-tile: consecutive MFMAs never share an accumulator
                            const int i = m % MT, h = (m / MT) % 4, jn = m / (4 * MT);
                            const int tap = h & 1, quad = h >> 1;
                            const float4 a4 = aq[g % RA][i];
                            float apin = tap == 0 ? (quad == 0 ? a4.x : a4.y) : (quad == 0 ? a4.z : a4.w);
                            asm volatile("" : "+v"(apin));   // (pins the MFMA between the memory operations around it: pair_wino_impl.h)
                            if (tap == 0) {
                                acc[A0][i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(apin, b_cur[quad][jn], acc[A0][i][jn], 0, 0, 0);
                                asm volatile("" : "+v"(acc[A0][i][jn]));
                            } else {
                                acc[A1][i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(apin, b_cur[2 + quad][jn], acc[A1][i][jn], 0, 0, 0);
                                asm volatile("" : "+v"(acc[A1][i][jn]));
                            }
                            if (m < MT) {
                                constexpr int fn = f + RA - 1;   // fragment to request, counted from this block's first one
                                aq[(g + RA - 1) % RA][m % MT] = load_w(wbs[fn / NF] + (m % MT) * mt_stride, fn % NF);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (f + 1 < NF && i == MT - 1) {
                                if constexpr (K3R) b_nxt[h][jn] = dk3[h & 1][(2 * (f + 1) + (h >> 1)) % 4][jn];
                                else b_nxt[h][jn] = bl[G::off_of(2 * (f + 1) + (h >> 1)) + (4 * (h & 1)) * PW + 16 * jn];
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (f + 1 < NF) {
#pragma unroll
                            for (int h = 0; h < 4; ++h)
#pragma unroll
                                for (int jn = 0; jn < NT; ++jn) b_cur[h][jn] = b_nxt[h][jn];
                        }
                    });
                    if (s < 11) FV_CV_STAMP(2 + s);
                }
            });
        }
    }

    FV_CV_STAMP(13);
    // ---- output transform, reduction over the four waves' K shares, epilogue ----
    // wave w ends up owning accumulator register w of every lane: rows 16 mt + 4 (lane >> 4) + w, pairs col + 16 jn
    __syncthreads();                                        // every wave is done with its planes: they become the exchange buffer
    float* X = lds;                                         // [wave][register][m-tile, n-tile, output of the pair][64 lanes]
    constexpr int NV2 = 2 * NT * MT;                        // values per lane and accumulator register
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            const f32x4w y0 = (acc[0][i][jn] + acc[1][i][jn]) + acc[2][i][jn];
            const f32x4w y1 = (acc[1][i][jn] - acc[2][i][jn]) - acc[3][i][jn];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                X[((wave * 4 + r) * NV2 + (i * NT + jn) * 2) * 64 + lane] = y0[r];
                X[((wave * 4 + r) * NV2 + (i * NT + jn) * 2 + 1) * 64 + lane] = y1[r];
            }
        }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * p.y_bstride, (unsigned)(p.y_bstride * 4));
    const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(p.res ? p.res + (long long)b * p.y_bstride : p.y, (unsigned)(p.y_bstride * 4));
    float val[NV2];
    unsigned off[NV2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = 16 * (mt + i) + 4 * krow + wave;
        const float bias = p.bias[m < p.M ? m : 0];
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            const int n = n0 + 16 * jn + col;
            const int q = n / DIL;
            const int ta = 2 * DIL * q + (n - q * DIL);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = ta + h * DIL;
                const int e = (i * NT + jn) * 2 + h;
                off[e] = (t < p.N && m < p.M) ? (unsigned)(m * p.N + t) * 4u : 0xFFFFFFFFu;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += X[((w * 4 + wave) * NV2 + e) * 64 + lane];   // (fixed order: waves 0, 1, 2, 3)
                val[e] = fmaf(v, 1.0f, bias);
            }
        }
    }
    if (p.res) {
#pragma unroll
        for (int e = 0; e < NV2; ++e) val[e] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, off[e], 0, 0));
    }
    act_apply_all(val, p.post_act, p.slope);
    if (p.out_mode == OUT_ACCUM) {
#pragma unroll
        for (int e = 0; e < NV2; ++e)
            val[e] = (__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, off[e], 0, 0)) + val[e]) * p.out_scale;
    }
#pragma unroll
    for (int e = 0; e < NV2; ++e) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[e]), yrs, off[e], 0, 0);
#ifdef FV_X_CONV_TS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    FV_CV_STAMP(14);
}

template <int KS, int DIL, int NT, int MT>
__global__ __launch_bounds__(256, 2) void conv_wino_lat_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float lds[WLGeom<KS, DIL, NT, MT>::LDS_F];
    wino_lat_body<KS, DIL, NT, MT>(p, lds, (int)blockIdx.x);
}

// tile: 0 = 16 rows x 16 pairs, 1 = 32 x 16, 2 = 32 x 32 (p.m_blks / p.n_tiles count these tiles)
template <int KS, int DIL>
inline bool launch_wino_lat_nt(const ConvParams& p, int tile, int batch, hipStream_t s) {
    const int grid = batch * p.m_blks * p.n_tiles;
    switch (tile) {
        case 0: hipLaunchKernelGGL((conv_wino_lat_kernel<KS, DIL, 1, 1>), dim3(grid), dim3(256), 0, s, p); return true;
        case 1: hipLaunchKernelGGL((conv_wino_lat_kernel<KS, DIL, 1, 2>), dim3(grid), dim3(256), 0, s, p); return true;
        case 2: hipLaunchKernelGGL((conv_wino_lat_kernel<KS, DIL, 2, 2>), dim3(grid), dim3(256), 0, s, p); return true;
        default: return false;
    }
}

template <int KS>
inline bool launch_wino_lat_k(const ConvParams& p, int nt, int batch, hipStream_t s) {
    switch (p.dil) {
        case 1: return launch_wino_lat_nt<KS, 1>(p, nt, batch, s);
        case 3: return launch_wino_lat_nt<KS, 3>(p, nt, batch, s);
        case 5: return launch_wino_lat_nt<KS, 5>(p, nt, batch, s);
        default: return false;
    }
}

}  // namespace fv
