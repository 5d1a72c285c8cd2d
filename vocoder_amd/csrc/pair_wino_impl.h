// Fused ResBlock (c1, c2) pair on Winograd F(2,3) tap groups for the narrow stages (C = 16 / 32):
//
//     y = x + c2( silu( c1( silu(x) ) ) )          (one iteration of ResBlock1.forward,
//                                                    fish_vocoder/modules/generators/hifigan.py:102-107)
//
// in ONE launch, with 16 / 10 / 4 matrix products per output pair and (c_out, c_in) instead of 22 / 14 / 6 (k = 11 / 7 / 3) in BOTH
// convs.  resblock_pair.hip (direct sums) showed these stages bound by instruction issue: the fp32 matrix instruction and the fp32
// vector ALU share the SIMD's datapath (profiles/r04a_pair_pmc.txt: SQ_ACTIVE_INST_VALU + SQ_VALU_MFMA_BUSY_CYCLES ~ 0.94 of the SIMD
// time at C = 16), so this kernel cuts both terms: a third fewer MFMAs, and a staging / epilogue code with no per-element address,
// clamp or mask arithmetic (every LDS / global address is a per-lane base + an immediate; range checks are the buffer descriptors').
//
// Pair lattice (conv_wino_impl.h): with dilation D outputs pair up as (u, u + D); pair column n = q D + r <-> u0(n) = 2 D q + r.
// One workgroup = 4 wavefronts, one batch item, NBP = 64 pair columns (16 per wave): c1 produces the W1 = 2 * (64 / D * D) contiguous
// samples [t0 - H2, t0 - H2 + W1), c2 the TT = W1 - (KS - 1) final samples [t0, t0 + TT) on the D = 1 lattice.
//   phase 0   silu(x) window -> LDS as E / O planes on c1's lattice (all C channels), raw centre columns -> LDS (residual operand)
//   per conv  for each chunk of CH channels:  transform E / O -> d0..d3 planes (LDS -> LDS), barrier, MFMA loop over virtual taps
//             (tap groups read the d planes, the single taps 3 / 7 read E / O directly; v_mfma_f32_16x16x4_f32, weights straight
//             from L2 in fragment order, DA fragments ahead)
//   c1 epilogue   output transform, SiLU (bias sits in the accumulators from the start), -> E / O planes of c2's lattice (overlaying c1's)
//   c2 epilogue   output transform, + raw x from LDS, -> HBM
// HBM traffic = the x window once + y once, as for the direct pair kernel.
#pragma once
#include "conv_mfma_impl.h"

namespace fv {

typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef float f32x2w __attribute__((ext_vector_type(2)));

constexpr int pw_up(int v, int m, int r) { return v + ((r - v % m) % m + m) % m; }   // smallest x >= v with x % m == r

// Weight ring: fragment f of a conv sits in slot f % RA, the load for fragment f + DA is issued while f is consumed.  RA = DA + U divides
// the fragments of a chunk wherever the chunk loop is a real loop (C = 32), so every chunk starts at slot 0 and nothing is moved.
constexpr int pw_da(int ks, int mt) { return mt == 2 ? (ks == 7 ? 4 : 3) : 4; }

// k = 3 is ONE tap group: every transformed value d0..d3 of a (channel, pair column) feeds exactly one matrix product per m-tile, in the wave that owns
// the column — so the narrow k = 3 pairs form their B operands in registers straight from the E / O planes (pw_gemm_k3) instead of passing them
// through a d-plane buffer in LDS: no chunk buffer, no barrier inside a conv, a third of the LDS instructions.  0 = the LDS transform (A/B builds).
#ifndef FV_X_PW_K3_REG
#define FV_X_PW_K3_REG 1
#endif

template <int KS, int DIL, int C, int CH>
struct PWGeom {
    static constexpr bool K3R = KS == 3 && FV_X_PW_K3_REG != 0;
    static_assert(C == 16 || C == 32, "16x16x4 kernel: one or two 16-row m-tiles");
    static_assert(CH == 8 || CH == 16, "chunk = 8 or 16 channels");
    static_assert(C % CH == 0 && CH * (C / 16) >= 16, "a chunk holds whole weight fragments");
    static constexpr int KSZ = KS, DILV = DIL;
    static constexpr int MT = C / 16;                    // 16-row m-tiles per wave (every wave owns all rows)
    static constexpr int U = MT == 2 ? 1 : 2;            // weight fragments (one float4 per lane = 4 MFMAs) per macro-step
    static constexpr int FCH = 16 / MT;                  // channels one weight fragment covers
    static constexpr int KST = FCH / 4;                  // MFMA k-steps (4 channels each) per fragment
    static constexpr int NG = (KS + 1) / 4, NS = (KS - 3) / 4, NV = 4 * NG + 2 * NS;
    static constexpr int NBP = 64;                       // pair columns per workgroup
    static constexpr int NU = NBP / DIL * DIL;           // ... of whole 2 D-sample blocks
    static constexpr int W1 = 2 * NU, TT = W1 - (KS - 1);
    static constexpr int H1 = (KS - 1) / 2 * DIL, H2 = (KS - 1) / 2, HP = H1 + H2;
    static constexpr int WD1 = NBP + 2 * DIL * (NG - 1), WR1 = WD1 + DIL;   // d-plane / E-O plane columns of c1
    static constexpr int WD2 = NBP + 2 * (NG - 1), WR2 = WD2 + 1;           // ... of c2 (dilation 1)
    static constexpr int NQ1 = (WR1 + DIL - 1) / DIL;    // 2 D-sample blocks staged
    static constexpr int NP1 = 2 * DIL * NQ1;            // staged positions per channel row
    static constexpr int PE = pw_up(DIL * NQ1, 16, 8);   // E / O plane stride: row stride 2 PE == 16 (mod 32): the 16x16x4 B-fragment read
    static constexpr int PD = pw_up(WD1, 8, 4);          // d plane stride:     row stride 4 PD == 16 (mod 32)  puts lanes 0-15 / 16-31 on disjoint banks
    static constexpr int SE = 2 * PE, SD = 4 * PD;
    static constexpr int XS = TT + 2;                    // raw-tile row stride (column TT: dump for the window's halo positions)
    static constexpr int EO_F = C * SE, D_F = K3R ? 0 : CH * SD, XR_F = C * XS + 16;
    static constexpr int TRASH = EO_F + D_F + XR_F;      // one float nobody reads
    static constexpr int LDS_FLOATS = TRASH + 4;
    static constexpr int CHN = CH;
    static constexpr int NF4 = CH / FCH * NV;            // weight fragments per chunk
    static constexpr int NMS = NF4 / U;                  // macro-steps per chunk
    static_assert(NF4 % U == 0, "whole macro-steps");
    static constexpr int DA = pw_da(KS, MT), RA = DA + U;   // weight prefetch distance / ring slots (fragments)
    static_assert(C == CH || NF4 % RA == 0, "the ring returns to slot 0 at every chunk boundary");
};

// Virtual tap v of a conv with dilation DX on strides (PE, PD): which accumulator plane it feeds, and where its B operand sits
// relative to (channel row, pair column n): a d plane at a group shift, or the E / O plane for the single taps 3, 7
template <int KS, int DX, int PE, int PD>
struct PWTap {
    static constexpr int NG = (KS + 1) / 4;
    static constexpr bool from_eo(int v) { return v >= 4 * NG; }
    static constexpr int acc_of(int v) { return v < 4 * NG ? v % 4 : ((v - 4 * NG) % 2 == 0 ? 0 : 3); }
    static constexpr int off_of(int v) {
        if (v < 4 * NG) return (v % 4) * PD + 2 * DX * (v / 4);
        const int s = (v - 4 * NG) / 2;
        return (v - 4 * NG) % 2 == 0 ? PE + (2 * s + 1) * DX : (2 * s + 2) * DX;   // + w O[n + (2s+1) D] -> m0,  - w E[n + (2s+2) D] -> m3
    }
};

__device__ __forceinline__ float pw_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// E / O planes (rows eo .. eo + CH - 1) -> d0..d3 planes of the chunk buffer.  TPR threads per channel row, consecutive columns.
template <class G, int DX, int WD>
__device__ __forceinline__ void pw_transform(const float* __restrict__ eo, float* __restrict__ d, int tid) {
    constexpr int CHn = G::CHN;
    constexpr int TPR = 256 / CHn, SLOTS = (WD + TPR - 1) / TPR;
    const int row = tid / TPR, c0 = tid % TPR;
    const float* e = eo + row * G::SE + c0;
    float* dd = d + row * G::SD + c0;
    float E[SLOTS], E1[SLOTS], O[SLOTS], O1[SLOTS];
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        E[j] = e[TPR * j];
        E1[j] = e[TPR * j + DX];
        O[j] = e[G::PE + TPR * j];
        O1[j] = e[G::PE + TPR * j + DX];
    }
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        if (TPR * (j + 1) <= WD || c0 + TPR * j < WD) {
            dd[TPR * j] = E[j] - E1[j];
            dd[G::PD + TPR * j] = O[j] + E1[j];
            dd[2 * G::PD + TPR * j] = E1[j] - O[j];
            dd[3 * G::PD + TPR * j] = O[j] - O1[j];
        }
    }
}

// Phase 0 of both kernel families: silu(x) of the window [t0 - HP, ...) -> E / O planes of c1's lattice for all C channels (zero outside [0, T):
// silu(0) == 0 is the conv's zero padding), and — XRES — the raw centre columns [t0, t0 + TT) -> Xr, the residual operand of the last epilogue.
// Each wave stages C / 4 whole channel rows.  Positions are loaded in order (coalesced dwords through a per-row buffer descriptor whose bounds
// check returns 0 outside the row) and scattered to their plane / pair column; the offsets are per lane slot, the same for every row, so a row
// costs its loads, the SiLUs and the LDS writes — no per-element index arithmetic.  The last NP1 % 64 positions of the wave's rows are
// flattened over (row, position) into whole lanes.
template <class G, int C, bool XRES>
__device__ __forceinline__ void pw_stage_window(const float* __restrict__ xb, int T, int t0, int wave, int lane, float* __restrict__ lds) {
    float* EO = lds;
    float* Xr = lds + G::EO_F + G::D_F;
    constexpr int DIL = G::DILV;
    constexpr int NFULL = G::NP1 / 64, TAILW = G::NP1 % 64, RPW = C / 4, NTAIL = (RPW * TAILW + 63) / 64;
    const int ws = t0 - G::HP;
    const int row0 = wave * RPW;
    auto eo_of = [&](int pp) {   // position of the window -> offset inside a channel row's E / O planes
        const int q = pp / (2 * DIL), rem = pp - 2 * DIL * q;
        const int hi = rem >= DIL ? 1 : 0;
        return hi * G::PE + q * DIL + rem - hi * DIL;
    };
    auto xr_of = [&](int pp) {
        const int c = pp - G::HP;
        return (c >= 0 && c < G::TT) ? c : G::TT;
    };
    float v[RPW][NFULL > 0 ? NFULL : 1];
    float vt[NTAIL > 0 ? NTAIL : 1];
    int eo_off[NFULL > 0 ? NFULL : 1], xr_off[NFULL > 0 ? NFULL : 1];
    unsigned voff[NFULL > 0 ? NFULL : 1];
#pragma unroll
    for (int i = 0; i < NFULL; ++i) {
        const int pp = lane + 64 * i;
        eo_off[i] = row0 * G::SE + eo_of(pp);
        xr_off[i] = row0 * G::XS + xr_of(pp);
        voff[i] = (unsigned)(ws + pp) * 4u;   // negative positions wrap past the descriptor's size: the load returns 0
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (long long)(row0 + rr) * T), 0, (unsigned)T * 4u, 0x00020000);
#pragma unroll
        for (int i = 0; i < NFULL; ++i) v[rr][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff[i], 0, 0));
    }
    // the last TAILW positions of the wave's rows, flattened over (row, position): whole lanes instead of RPW part-filled slots
    int eo_t[NTAIL > 0 ? NTAIL : 1], xr_t[NTAIL > 0 ? NTAIL : 1];
    if constexpr (NTAIL > 0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (unsigned)(C * T) * 4u, 0x00020000);
#pragma unroll
        for (int j = 0; j < NTAIL; ++j) {
            const int e = lane + 64 * j;
            const bool ok = e < RPW * TAILW;
            const int rr = e / TAILW, pp = NFULL * 64 + e - rr * TAILW;
            const int tpos = ws + pp;
            const bool in = ok && tpos >= 0 && tpos < T;
            eo_t[j] = ok ? (row0 + rr) * G::SE + eo_of(pp) : G::TRASH;
            const int xc = xr_of(pp);
            xr_t[j] = ok ? (row0 + rr) * G::XS + xc : G::TRASH - G::EO_F - G::D_F;   // (Xr-relative)
            vt[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, in ? (unsigned)((row0 + rr) * T + tpos) * 4u : 0xFFFFFFFFu, 0, 0));
        }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int i = 0; i < NFULL; ++i) {
            EO[eo_off[i] + rr * G::SE] = pw_silu(v[rr][i]);   // silu(0) == 0: the conv's zero padding
            if constexpr (XRES) Xr[xr_off[i] + rr * G::XS] = v[rr][i];
        }
    if constexpr (NTAIL > 0) {
#pragma unroll
        for (int j = 0; j < NTAIL; ++j) {
            EO[eo_t[j]] = pw_silu(vt[j]);
            if constexpr (XRES) Xr[xr_t[j]] = vt[j];
        }
    }
}

// MFMA loop over one chunk: NF4 weight fragments, four 16x16x4 MFMAs each.  dl / el: the lane's base into the d planes / the chunk's
// E-O rows (k-quarter row and pair column folded in).  aq: weight ring, fragments [f, f + DA) of the conv on entry and on exit.
template <class G, int DX>
__device__ __forceinline__ void pw_gemm_chunk(f32x4w (&acc)[4][G::MT], const float* __restrict__ dl, const float* __restrict__ el,
                                              const __amdgpu_buffer_rsrc_t wrs, int wvoff, int wsoff, float4 (&aq)[G::RA]) {
    constexpr int DA = G::DA, RA = G::RA, U = G::U, KST = G::KST, MT = G::MT, NV = G::NV;
    using TP = PWTap<G::KSZ, DX, G::PE, G::PD>;
    constexpr int NB = U * KST;       // B registers per macro-step
    constexpr int NM = 4 * U;         // MFMAs per macro-step
    float b_cur[NB], b_nxt[NB];
    auto b_addr = [&](int f, int s) __attribute__((always_inline)) -> const float* {   // fragment f of the chunk, k-step s
        const int sb = f / NV, v = f % NV;
        const int rowc = sb * G::FCH + 4 * s;
        return TP::from_eo(v) ? el + rowc * G::SE + TP::off_of(v) : dl + rowc * G::SD + TP::off_of(v);
    };
#pragma unroll
    for (int i = 0; i < NB; ++i) b_cur[i] = *b_addr(i / KST, i % KST);
    static_for<G::NMS>([&](auto ms_c) __attribute__((always_inline)) {
        constexpr int ms = decltype(ms_c)::value;
#ifdef FV_X_PW_SKIP
        if constexpr (ms * 8 >= G::NMS * 5) return;   // timing experiment (wrong results): 5 of every 8 products, as F(4,4) tap groups would leave
#endif
        constexpr int f0 = ms * U;
        constexpr int NLDX = U + NB;   // memory operations of this macro-step: U weight loads (DA fragments ahead), NB LDS reads (next macro-step)
        // each of them is requested BETWEEN two MFMAs: an in-order wave hides a memory instruction's issue time only under a matrix
        // instruction that is already executing (conv_mfma_impl.h)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            // MT == 2: (s0, mt0) (s0, mt1) (s1, mt0) (s1, mt1) of one fragment; MT == 1: the k-steps of two fragments alternate (their
            // accumulator planes differ: a 16x16x4 MFMA has 40 cycles of dependent latency against 32 of issue)
            const int u = MT == 2 ? 0 : m % 2;
            const int s = m / 2;
            const int mt = MT == 2 ? m % 2 : 0;
            const int comp = MT == 2 ? m : m / 2;
            const int A = TP::acc_of((f0 + u) % NV);
            const float4 a4 = u == 0 ? aq[f0 % RA] : aq[(f0 + U - 1) % RA];
            const float av = comp == 0 ? a4.x : comp == 1 ? a4.y : comp == 2 ? a4.z : a4.w;
            // The matrix instruction is a pure value to the instruction selector, which is free to float it past the (ordered) memory
            // operations and scheduling barriers around it — and did: every MFMA of c1 ended up behind all of the chunk's loads, 112
            // operands live.  Two empty volatile asm statements (ordered like the loads) pin it: its A operand passes through one
            // before it, its result through one after it.
            float apin = av;   // (the A value is used by this MFMA alone: no copy; a B value feeds MT of them)
            asm volatile("" : "+v"(apin));
            acc[A][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(apin, b_cur[u * KST + s], acc[A][mt], 0, 0, 0);
            asm volatile("" : "+v"(acc[A][mt]));
#pragma unroll
            for (int k = 0; k < NLDX; ++k) {
                if (k * NM / NLDX == m) {
                    if (k < U) {
                        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, wsoff + (f0 + k) * 1024, 0);
                        const float4 wv = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
                        if (k == 0) aq[(f0 + DA) % RA] = wv; else aq[(f0 + DA + U - 1) % RA] = wv;
                    } else if (ms + 1 < G::NMS) {
                        b_nxt[k - U] = *b_addr(f0 + U + (k - U) / KST, (k - U) % KST);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ms + 1 < G::NMS) {
#pragma unroll
            for (int i = 0; i < NB; ++i) b_cur[i] = b_nxt[i];
        }
    });
}

// k = 3: one conv's whole MFMA loop with the Winograd input transform in registers.  el: the lane's base into the E / O planes (k-quarter row and pair
// column folded in).  w0: the weight fragments of channel block 0 (virtual taps 0..3) on entry; block sb + 1's travel while block sb is multiplied.
// Per block of FCH channels: KST k-steps x (E, E', O, O') -> d0..d3 (d0 = E - E', d1 = O + E', d2 = E' - O, d3 = O - O'; E' = E[n + D]), then
// 4 KST MT MFMAs, virtual tap fastest: consecutive instructions never share an accumulator plane.
template <class G, int DX>
__device__ __forceinline__ void pw_gemm_k3(f32x4w (&acc)[4][G::MT], const float* __restrict__ el, const __amdgpu_buffer_rsrc_t wrs, int wvoff,
                                           float4 (&w0)[4]) {
    static_assert(G::KSZ == 3 && G::NV == 4, "one tap group");
    constexpr int KST = G::KST, MT = G::MT, FCH = G::FCH, NSB = 16 * MT / FCH;
    float raw[2][KST][4];
    float4 wq[2][4];
    auto read_raw = [&](int sb, float (&r)[KST][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KST; ++s) {
            const float* e = el + (sb * FCH + 4 * s) * G::SE;
            r[s][0] = e[0];
            r[s][1] = e[DX];
            r[s][2] = e[G::PE];
            r[s][3] = e[G::PE + DX];
        }
    };
#pragma unroll
    for (int v = 0; v < 4; ++v) wq[0][v] = w0[v];
    read_raw(0, raw[0]);
    static_for<NSB>([&](auto sb_c) __attribute__((always_inline)) {
        constexpr int sb = decltype(sb_c)::value;
        constexpr int cur = sb & 1, nxt = cur ^ 1;
        if constexpr (sb + 1 < NSB) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, ((sb + 1) * 4 + v) * 1024, 0);
                wq[nxt][v] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
            }
            read_raw(sb + 1, raw[nxt]);
        }
        float d[KST][4];
#pragma unroll
        for (int s = 0; s < KST; ++s) {
            const float E = raw[cur][s][0], E1 = raw[cur][s][1], O = raw[cur][s][2], O1 = raw[cur][s][3];
            d[s][0] = E - E1;
            d[s][1] = O + E1;
            d[s][2] = E1 - O;
            d[s][3] = O - O1;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4 * KST * MT; ++m) {
            const int v = m % 4, mt = (m / 4) % MT, ks = m / (4 * MT);
            const int comp = MT == 2 ? 2 * ks + mt : ks;          // (the fragment layouts of conv_layer.hip: C = 16 .[s], C = 32 .[2 s + mt])
            const float4 a4 = wq[cur][v];
            float apin = comp == 0 ? a4.x : comp == 1 ? a4.y : comp == 2 ? a4.z : a4.w;
            asm volatile("" : "+v"(apin));                        // (pins the MFMA in program order: pw_gemm_chunk)
            acc[v][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(apin, d[ks][v], acc[v][mt], 0, 0, 0);
            asm volatile("" : "+v"(acc[v][mt]));
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <int KS, int DIL, int C, int CH>
__global__ __launch_bounds__(256, 4) void pair_wino16_kernel(const PairParams p) {
    using G = PWGeom<KS, DIL, C, CH>;
    constexpr int MT = G::MT, DA = G::DA, NCHK = C / CH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* EO = lds;
    float* Db = lds + G::EO_F;
    float* Xr = lds + G::EO_F + G::D_F;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a clip's neighbouring tiles on one XCD (they share the cache lines of their halo columns): resblock_pair.hip
    const int lid = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (lid >= p.n_tiles * p.batch) return;
    const int tile = lid % p.n_tiles, b = lid / p.n_tiles;
    const int t0 = tile * G::TT;
    const int T = p.T;
    const float* __restrict__ xb = p.x + (long long)b * C * T;

    // weight rings: c1's first fragments are requested before anything else
    const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    float4 aq[G::RA];
    float4 w0[4];                                 // k = 3 (pw_gemm_k3): the fragments of the conv's first channel block
    auto load_w = [&](const __amdgpu_buffer_rsrc_t rs, int f) __attribute__((always_inline)) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, wvoff, f * 1024, 0);
        return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    };
    if constexpr (G::K3R) {
#pragma unroll
        for (int v = 0; v < 4; ++v) w0[v] = load_w(w1rs, v);
    } else {
#pragma unroll
        for (int d = 0; d < DA; ++d) aq[d] = load_w(w1rs, d);
    }

    pw_stage_window<G, C, true>(xb, T, t0, wave, lane, lds);

    // accumulators: the bias rides in m0 (+b) and m3 (-b): y0 = m0 + m1 + m2, y1 = m1 - m2 - m3
    const int krow = lane >> 4;               // C / D layout of 16x16x4: row = 4 (lane >> 4) + reg, column = lane & 15
    const int ncol = 16 * wave + (lane & 15);
    f32x4w acc[4][MT];
    auto init_acc = [&](const float* __restrict__ bias, const float* __restrict__ nbias) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            acc[0][i] = *(const f32x4w*)(bias + 16 * i + 4 * krow);
            acc[3][i] = *(const f32x4w*)(nbias + 16 * i + 4 * krow);   // (-b, negated on the host)
            acc[1][i] = f32x4w{0.f, 0.f, 0.f, 0.f};
            acc[2][i] = f32x4w{0.f, 0.f, 0.f, 0.f};
        }
    };
    init_acc(p.b1, p.b1n);
    const float* dl = Db + krow * G::SD + ncol;
    const float* el = EO + krow * G::SE + ncol;
    __syncthreads();

    // ---- c1 ----
    if constexpr (G::K3R) {
        pw_gemm_k3<G, DIL>(acc, el, w1rs, wvoff, w0);
        __syncthreads();   // every wave has read its E / O columns: the c1 epilogue overwrites the planes
#pragma unroll
        for (int v = 0; v < 4; ++v) w0[v] = load_w(w2rs, v);
    } else {
        for (int c = 0; c < NCHK; ++c) {
            pw_transform<G, DIL, G::WD1>(EO + c * CH * G::SE, Db, tid);
            __syncthreads();
            pw_gemm_chunk<G, DIL>(acc, dl, el + c * CH * G::SE, w1rs, wvoff, __builtin_amdgcn_readfirstlane((c * G::NF4 + DA) * 1024), aq);
            __syncthreads();   // the chunk buffer (next transform) and the E / O planes (c1 epilogue) are free again
        }
        // c2's first weight fragments travel while the epilogue runs (the ring holds c1's overrun fragments: zeros, never used)
#pragma unroll
        for (int d = 0; d < DA; ++d) aq[d] = load_w(w2rs, d);
    }

    // ---- c1 epilogue: silu(c1 + b1) -> E / O planes of c2's lattice (mid[u], u = position - (t0 - H2): E2[u >> 1] / O2[u >> 1]) ----
    {
        const int q = ncol / DIL, r = ncol - q * DIL;
        const int u0 = 2 * DIL * q + r, u1 = u0 + DIL;
        float* w0 = EO + 4 * krow * G::SE + (u0 & 1) * G::PE + (u0 >> 1);
        float* w1 = EO + 4 * krow * G::SE + (u1 & 1) * G::PE + (u1 >> 1);
        const int ts = t0 - G::H2;
        const bool edge = ts < 0 || ts + 2 * G::NBP + DIL > T;   // wave-uniform: only a row's first / last tiles zero positions outside [0, T)
        float k0 = 1.f, k1 = 1.f;
        if (edge) {
            k0 = (ts + u0 >= 0 && ts + u0 < T) ? 1.f : 0.f;
            k1 = (ts + u1 >= 0 && ts + u1 < T) ? 1.f : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const f32x4w y0 = (acc[0][i] + acc[1][i]) + acc[2][i];
            const f32x4w y1 = (acc[1][i] - acc[2][i]) - acc[3][i];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float s0 = pw_silu(y0[rg]), s1 = pw_silu(y1[rg]);
                if (edge) {
                    s0 *= k0;
                    s1 *= k1;
                }
                w0[(16 * i + rg) * G::SE] = s0;
                w1[(16 * i + rg) * G::SE] = s1;
            }
        }
    }
    init_acc(p.b2, p.b2n);
    __syncthreads();

    // ---- c2 (dilation 1) ----
    if constexpr (G::K3R) {
        pw_gemm_k3<G, 1>(acc, el, w2rs, wvoff, w0);
    } else {
        for (int c = 0; c < NCHK; ++c) {
            pw_transform<G, 1, G::WD2>(EO + c * CH * G::SE, Db, tid);
            __syncthreads();
            pw_gemm_chunk<G, 1>(acc, dl, el + c * CH * G::SE, w2rs, wvoff, __builtin_amdgcn_readfirstlane((c * G::NF4 + DA) * 1024), aq);
            if (c + 1 < NCHK) __syncthreads();
        }
    }

    // ---- c2 epilogue: + raw x (LDS) -> y ----
    {
        const int tl = 2 * ncol;                       // first of the lane's two output columns inside the tile
        const int t = t0 + tl;
        const unsigned va = (tl < G::TT && t < T) ? (unsigned)(4 * krow * T + t) * 4u : 0xFFFFFFFFu;
        const unsigned vb = (tl + 1 < G::TT && t + 1 < T) ? (unsigned)(4 * krow * T + t + 1) * 4u : 0xFFFFFFFFu;
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long long)b * C * T), 0, (unsigned)(C * T) * 4u, 0x00020000);
        const float* xl = Xr + 4 * krow * G::XS + (tl < G::TT ? tl : 0);
        const bool accum = p.out_mode == OUT_ACCUM;
        // the lane's two outputs are adjacent samples: one 8-byte store (and accumulate load) per row where every row starts 8-byte aligned — whole 64-byte
        // lines per 16 lanes instead of two passes over every other float (round 5; the 16-byte form of conv_wino44's epilogue, LOG R5.3)
#ifdef FV_X_NO_PAIR8
        const bool pair8 = false;
#else
        const bool pair8 = (T & 1) == 0 && ((unsigned long long)p.y & 7ull) == 0;
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const f32x4w y0 = (acc[0][i] + acc[1][i]) + acc[2][i];
            const f32x4w y1 = (acc[1][i] - acc[2][i]) - acc[3][i];
            float o0[4], o1[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x2w xr = *(const f32x2w*)(xl + (16 * i + rg) * G::XS);
                o0[rg] = y0[rg] + xr.x;
                o1[rg] = y1[rg] + xr.y;
            }
            if (pair8) {   // (TT and t0 are even: a pair is inside the tile and the row, or outside both)
                if (accum) {
                    u32x2 a[4];
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        a[rg] = __builtin_amdgcn_raw_buffer_load_b64(yrs, va, __builtin_amdgcn_readfirstlane((16 * i + rg) * T * 4), 0);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        o0[rg] = (__uint_as_float(a[rg].x) + o0[rg]) * p.out_scale;
                        o1[rg] = (__uint_as_float(a[rg].y) + o1[rg]) * p.out_scale;
                    }
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    u32x2 v;
                    v.x = __float_as_uint(o0[rg]);
                    v.y = __float_as_uint(o1[rg]);
                    __builtin_amdgcn_raw_buffer_store_b64(v, yrs, va, __builtin_amdgcn_readfirstlane((16 * i + rg) * T * 4), 0);
                }
                continue;
            }
            if (accum) {
                float a0[4], a1[4];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int so = __builtin_amdgcn_readfirstlane((16 * i + rg) * T * 4);
                    a0[rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, va, so, 0));
                    a1[rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, vb, so, 0));
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    o0[rg] = (a0[rg] + o0[rg]) * p.out_scale;
                    o1[rg] = (a1[rg] + o1[rg]) * p.out_scale;
                }
            }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int so = __builtin_amdgcn_readfirstlane((16 * i + rg) * T * 4);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0[rg]), yrs, va, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o1[rg]), yrs, vb, so, 0);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Wide stages (C = 64 / 128): v_mfma_f32_32x32x2_f32, waves stacked along M (each wave owns ONE 32-row m-tile — its own share of the
// weights, in conv_wino_impl.h's fragment order: the layers' d_wpw — and NT n-tiles of 32 pair columns); WN = 4 / WM waves along N.
// XRES: the raw tile for the residual stays in LDS (C = 64 at most: at C = 128 it does not fit next to the window); otherwise the last
// epilogue reads x again, requested before c2's matrix loop starts.
// ---------------------------------------------------------------------------------------------------------------------------------------
template <int KS, int DIL, int C, int CH, int NT, bool XRES_, bool DBUF_ = false>
struct PW32Geom {
    static constexpr bool DBUF = DBUF_;                  // two chunk buffers: one workgroup barrier per chunk instead of two
    static_assert(C == 64 || C == 128, "one 32-row m-tile per wave");
    static_assert(CH % 8 == 0 && C % CH == 0, "chunks of whole 8-channel fragments");
    static constexpr bool XRES = XRES_;
    static constexpr int KSZ = KS, DILV = DIL;
    static constexpr int WM = C / 32 > 4 ? 4 : C / 32, WN = 4 / WM;
    static constexpr int NG = (KS + 1) / 4, NS = (KS - 3) / 4, NV = 4 * NG + 2 * NS;
    static constexpr int NBP = WN * NT * 32;
    static constexpr int NU = NBP / DIL * DIL;
    static constexpr int W1 = 2 * NU, TT = W1 - (KS - 1);
    static constexpr int H1 = (KS - 1) / 2 * DIL, H2 = (KS - 1) / 2, HP = H1 + H2;
    static constexpr int WD1 = NBP + 2 * DIL * (NG - 1), WR1 = WD1 + DIL;
    static constexpr int WD2 = NBP + 2 * (NG - 1), WR2 = WD2 + 1;
    static constexpr int NQ1 = (WR1 + DIL - 1) / DIL;
    static constexpr int NP1 = 2 * DIL * NQ1;
    static constexpr int PE = (DIL * NQ1 + 1) / 2 * 2;   // (a 32-lane group of the MFMA reads / the transform covers one row: any stride is conflict-free
    static constexpr int PD = pw_up(WD1, 8, 4);          //  for CH = 8; the 16-lane row segments of CH = 16 want 4 PD == 16 (mod 32))
    static constexpr int SE = 2 * PE, SD = 4 * PD;
    static constexpr int XS = TT + 2;
    static constexpr int DB_F = CH * SD;                 // one chunk buffer
    static constexpr int EO_F = C * SE, D_F = (DBUF ? 2 : 1) * DB_F, XR_F = XRES ? C * XS + 16 : 0;
    static constexpr int TRASH = EO_F + D_F + XR_F;
    static constexpr int LDS_FLOATS = TRASH + 4;
    static constexpr int CHN = CH;
    static constexpr int NF4 = CH / 8 * NV;              // weight fragments (8 channels x one virtual tap = 4 MFMAs per n-tile) per chunk
    static constexpr int DA = 3, RA = 4;                 // weight ring: RA divides the fragments of a chunk (NV * CH / 8)
    static_assert(NF4 % RA == 0, "the ring returns to slot 0 at every chunk boundary");
};

template <class G, int DX>
__device__ __forceinline__ void pw32_gemm_chunk(f32x16 (&acc)[4][1], const float* __restrict__ dl, const float* __restrict__ el,
                                                const __amdgpu_buffer_rsrc_t wrs, int wvoff, int wsoff, float4 (&aq)[G::RA]) {
    constexpr int DA = G::DA, RA = G::RA, NV = G::NV;
    using TP = PWTap<G::KSZ, DX, G::PE, G::PD>;
    float b_cur[4], b_nxt[4];
    auto b_addr = [&](int f, int pp) __attribute__((always_inline)) -> const float* {   // fragment f of the chunk, channel pair pp
        const int sb = f / NV, v = f % NV;
        const int rowc = sb * 8 + 2 * pp;
        return TP::from_eo(v) ? el + rowc * G::SE + TP::off_of(v) : dl + rowc * G::SD + TP::off_of(v);
    };
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) b_cur[pp] = *b_addr(0, pp);
    static_for<G::NF4>([&](auto f_c) __attribute__((always_inline)) {
        constexpr int f = decltype(f_c)::value;
        constexpr int A = TP::acc_of(f % NV);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 a4 = aq[f % RA];
            float apin = m == 0 ? a4.x : m == 1 ? a4.y : m == 2 ? a4.z : a4.w;
            asm volatile("" : "+v"(apin));   // (pins the MFMA between the memory operations around it: pw_gemm_chunk)
            acc[A][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(apin, b_cur[m], acc[A][0], 0, 0, 0);
            asm volatile("" : "+v"(acc[A][0]));
            if (m == 0) {
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, wsoff + f * 1024, 0);
                aq[(f + DA) % RA] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (f + 1 < G::NF4) {
                b_nxt[m] = *b_addr(f + 1, m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (f + 1 < G::NF4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b_cur[i] = b_nxt[i];
        }
    });
}

template <int KS, int DIL, int C, int CH, bool XRES, bool DBUF>
__global__ __launch_bounds__(256, 2) void pair_wino32_kernel(const PairParams p) {
    using G = PW32Geom<KS, DIL, C, CH, 1, XRES, DBUF>;
    constexpr int DA = G::DA, NCHK = C / CH, WN = G::WN;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* EO = lds;
    float* Db = lds + G::EO_F;
    float* Xr = lds + G::EO_F + G::D_F;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lid = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);   // a clip's tiles on one XCD
    if (lid >= p.n_tiles * p.batch) return;
    const int tile = lid % p.n_tiles, b = lid / p.n_tiles;
    const int t0 = tile * G::TT;
    const int T = p.T;
    const float* __restrict__ xb = p.x + (long long)b * C * T;

    // this wave's m-tile of the packed weights: fragments [wm * nfrag, (wm + 1) * nfrag), nfrag = p.n_frag (nchunk * NV)
    const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    const int wbase = __builtin_amdgcn_readfirstlane(wm * p.n_frag * 1024);
    float4 aq[G::RA];
    auto load_w = [&](const __amdgpu_buffer_rsrc_t rs, int f) __attribute__((always_inline)) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, wvoff, wbase + f * 1024, 0);
        return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    };
#pragma unroll
    for (int d = 0; d < DA; ++d) aq[d] = load_w(w1rs, d);

    pw_stage_window<G, C, XRES>(xb, T, t0, wave, lane, lds);

    // C / D layout of 32x32x2: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); the bias rides in m0 (+b) and m3 (-b)
    const int khalf = lane >> 5;
    const int ncol = 32 * wn + (lane & 31);
    const int mrow0 = 32 * wm + 4 * khalf;
    f32x16 acc[4][1];
    auto init_acc = [&](const float* __restrict__ bias, const float* __restrict__ nbias) __attribute__((always_inline)) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const f32x4w bv = *(const f32x4w*)(bias + mrow0 + 8 * rq);
            const f32x4w nv = *(const f32x4w*)(nbias + mrow0 + 8 * rq);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                acc[0][0][4 * rq + rr] = bv[rr];
                acc[3][0][4 * rq + rr] = nv[rr];
                acc[1][0][4 * rq + rr] = 0.f;
                acc[2][0][4 * rq + rr] = 0.f;
            }
        }
    };
    init_acc(p.b1, p.b1n);
    const float* dl = Db + khalf * G::SD + ncol;
    const float* el = EO + khalf * G::SE + ncol;
    __syncthreads();

    // ---- c1 ----
    if constexpr (DBUF) {
        // chunk c + 1 is transformed into the other buffer BEFORE chunk c's matrix loop: one barrier per chunk, and the transform's LDS
        // round trip runs under the matrix instructions that follow it
        pw_transform<G, DIL, G::WD1>(EO, Db, tid);
        __syncthreads();
        for (int c = 0; c < NCHK; ++c) {
            if (c + 1 < NCHK) pw_transform<G, DIL, G::WD1>(EO + (c + 1) * CH * G::SE, Db + ((c + 1) & 1) * G::DB_F, tid);
            pw32_gemm_chunk<G, DIL>(acc, dl + (c & 1) * G::DB_F, el + c * CH * G::SE, w1rs, wvoff,
                                    __builtin_amdgcn_readfirstlane(wbase + (c * G::NF4 + DA) * 1024), aq);
            __syncthreads();
        }
    } else {
        for (int c = 0; c < NCHK; ++c) {
            pw_transform<G, DIL, G::WD1>(EO + c * CH * G::SE, Db, tid);
            __syncthreads();
            pw32_gemm_chunk<G, DIL>(acc, dl, el + c * CH * G::SE, w1rs, wvoff, __builtin_amdgcn_readfirstlane(wbase + (c * G::NF4 + DA) * 1024), aq);
            __syncthreads();
        }
    }
#pragma unroll
    for (int d = 0; d < DA; ++d) aq[d] = load_w(w2rs, d);

    // ---- c1 epilogue: silu(c1 + b1) -> E / O planes of c2's lattice ----
    {
        const int q = ncol / DIL, r = ncol - q * DIL;
        const int u0 = 2 * DIL * q + r, u1 = u0 + DIL;
        float* w0 = EO + mrow0 * G::SE + (u0 & 1) * G::PE + (u0 >> 1);
        float* w1 = EO + mrow0 * G::SE + (u1 & 1) * G::PE + (u1 >> 1);
        const int ts = t0 - G::H2;
        const bool edge = ts < 0 || ts + 2 * G::NBP + DIL > T;
        float k0 = 1.f, k1 = 1.f;
        if (edge) {
            k0 = (ts + u0 >= 0 && ts + u0 < T) ? 1.f : 0.f;
            k1 = (ts + u1 >= 0 && ts + u1 < T) ? 1.f : 0.f;
        }
        const f32x16 y0 = (acc[0][0] + acc[1][0]) + acc[2][0];
        const f32x16 y1 = (acc[1][0] - acc[2][0]) - acc[3][0];
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            float s0 = pw_silu(y0[rg]), s1 = pw_silu(y1[rg]);
            if (edge) {
                s0 *= k0;
                s1 *= k1;
            }
            w0[((rg & 3) + 8 * (rg >> 2)) * G::SE] = s0;
            w1[((rg & 3) + 8 * (rg >> 2)) * G::SE] = s1;
        }
    }
    init_acc(p.b2, p.b2n);

    // the last epilogue's operands that come from HBM (the residual when it is not in LDS, the accumulate operand) are requested now:
    // their latency runs under c2's matrix loop
    const int tl = 2 * ncol;
    const int t = t0 + tl;
    const unsigned va = (tl < G::TT && t < T) ? (unsigned)(mrow0 * T + t) * 4u : 0xFFFFFFFFu;
    const unsigned vb = (tl + 1 < G::TT && t + 1 < T) ? (unsigned)(mrow0 * T + t + 1) * 4u : 0xFFFFFFFFu;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long long)b * C * T), 0, (unsigned)(C * T) * 4u, 0x00020000);
    // the lane's two outputs are adjacent samples: 8-byte residual loads / accumulate loads / stores where every row of x and y starts 8-byte aligned
    // (pair_wino16_kernel's epilogue)
#ifdef FV_X_NO_PAIR8
    const bool pair8 = false;
#else
    const bool pair8 = (T & 1) == 0 && (((unsigned long long)p.y | (unsigned long long)p.x) & 7ull) == 0;
#endif
    float ra[XRES ? 1 : 16], rb[XRES ? 1 : 16];
    if constexpr (!XRES) {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (unsigned)(C * T) * 4u, 0x00020000);
        if (pair8) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, va, __builtin_amdgcn_readfirstlane(((rg & 3) + 8 * (rg >> 2)) * T * 4), 0);
                ra[rg] = __uint_as_float(v.x);
                rb[rg] = __uint_as_float(v.y);
            }
        } else {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int so = __builtin_amdgcn_readfirstlane(((rg & 3) + 8 * (rg >> 2)) * T * 4);
                ra[rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, va, so, 0));
                rb[rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, vb, so, 0));
            }
        }
    }
    __syncthreads();

    // ---- c2 (dilation 1) ----
    if constexpr (DBUF) {
        pw_transform<G, 1, G::WD2>(EO, Db, tid);
        __syncthreads();
        for (int c = 0; c < NCHK; ++c) {
            if (c + 1 < NCHK) pw_transform<G, 1, G::WD2>(EO + (c + 1) * CH * G::SE, Db + ((c + 1) & 1) * G::DB_F, tid);
            pw32_gemm_chunk<G, 1>(acc, dl + (c & 1) * G::DB_F, el + c * CH * G::SE, w2rs, wvoff,
                                  __builtin_amdgcn_readfirstlane(wbase + (c * G::NF4 + DA) * 1024), aq);
            if (c + 1 < NCHK) __syncthreads();
        }
    } else {
        for (int c = 0; c < NCHK; ++c) {
            pw_transform<G, 1, G::WD2>(EO + c * CH * G::SE, Db, tid);
            __syncthreads();
            pw32_gemm_chunk<G, 1>(acc, dl, el + c * CH * G::SE, w2rs, wvoff, __builtin_amdgcn_readfirstlane(wbase + (c * G::NF4 + DA) * 1024), aq);
            if (c + 1 < NCHK) __syncthreads();
        }
    }

    // ---- c2 epilogue: + raw x -> y ----
    {
        const f32x16 y0 = (acc[0][0] + acc[1][0]) + acc[2][0];
        const f32x16 y1 = (acc[1][0] - acc[2][0]) - acc[3][0];
        const float* xl = Xr + mrow0 * G::XS + (tl < G::TT ? tl : 0);
        const bool accum = p.out_mode == OUT_ACCUM;
        float o0[16], o1[16];
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            if constexpr (XRES) {
                const f32x2w xr = *(const f32x2w*)(xl + ((rg & 3) + 8 * (rg >> 2)) * G::XS);
                o0[rg] = y0[rg] + xr.x;
                o1[rg] = y1[rg] + xr.y;
            } else {
                o0[rg] = y0[rg] + ra[rg];
                o1[rg] = y1[rg] + rb[rg];
            }
        }
        if (accum) {
            float a0[16], a1[16];
            if (pair8) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(yrs, va, __builtin_amdgcn_readfirstlane(((rg & 3) + 8 * (rg >> 2)) * T * 4), 0);
                    a0[rg] = __uint_as_float(v.x);
                    a1[rg] = __uint_as_float(v.y);
                }
            } else {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int so = __builtin_amdgcn_readfirstlane(((rg & 3) + 8 * (rg >> 2)) * T * 4);
                    a0[rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, va, so, 0));
                    a1[rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, vb, so, 0));
                }
            }
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                o0[rg] = (a0[rg] + o0[rg]) * p.out_scale;
                o1[rg] = (a1[rg] + o1[rg]) * p.out_scale;
            }
        }
        if (pair8) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                u32x2 v;
                v.x = __float_as_uint(o0[rg]);
                v.y = __float_as_uint(o1[rg]);
                __builtin_amdgcn_raw_buffer_store_b64(v, yrs, va, __builtin_amdgcn_readfirstlane(((rg & 3) + 8 * (rg >> 2)) * T * 4), 0);
            }
        } else {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int so = __builtin_amdgcn_readfirstlane(((rg & 3) + 8 * (rg >> 2)) * T * 4);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0[rg]), yrs, va, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o1[rg]), yrs, vb, so, 0);
            }
        }
    }
}

#ifndef FV_X_PW_CH64
#define FV_X_PW_CH64 8
#endif
#ifndef FV_X_PW_DBUF
#define FV_X_PW_DBUF 0
#endif
#ifndef FV_X_PW_CH128
#define FV_X_PW_CH128 8
#endif
#ifndef FV_X_PW_XRES64
#define FV_X_PW_XRES64 1   // the raw tile stays in LDS at C = 64 (two workgroups per CU; measured equal to three without it, at 1.0 x the traffic)
#endif

template <int KS, int DIL, int C, int CH, bool XRES, bool DBUF>
inline bool launch_pair_wino32_one(const PairParams& p, int batch, hipStream_t s) {
    using G = PW32Geom<KS, DIL, C, CH, 1, XRES, DBUF>;
    PairParams q = p;
    q.n_tiles = (p.T + G::TT - 1) / G::TT;
    q.batch = batch;
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
    if (!FV_ENSURE_DYN_LDS((pair_wino32_kernel<KS, DIL, C, CH, XRES, DBUF>), lds)) return false;
    hipLaunchKernelGGL((pair_wino32_kernel<KS, DIL, C, CH, XRES, DBUF>), dim3((batch * q.n_tiles + 7) / 8 * 8), dim3(256), lds, s, q);
    return true;
}

template <int KS, int DIL, int C, int CH>
inline bool launch_pair_wino16_one(const PairParams& p, int batch, hipStream_t s) {
    using G = PWGeom<KS, DIL, C, CH>;
    PairParams q = p;
    q.n_tiles = (p.T + G::TT - 1) / G::TT;
    q.batch = batch;
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
    if (!FV_ENSURE_DYN_LDS((pair_wino16_kernel<KS, DIL, C, CH>), lds)) return false;
    hipLaunchKernelGGL((pair_wino16_kernel<KS, DIL, C, CH>), dim3((batch * q.n_tiles + 7) / 8 * 8), dim3(256), lds, s, q);
    return true;
}

#ifndef FV_X_PW_CH32
#define FV_X_PW_CH32 8
#endif

template <int KS>
inline bool launch_pair_wino16_k(const PairParams& p, int C, int dil, int batch, hipStream_t s) {
#define FV_PW_CASE(D)                                                                      \
    if (dil == D) {                                                                        \
        if (C == 16) return launch_pair_wino16_one<KS, D, 16, 16>(p, batch, s);            \
        if (C == 32) return launch_pair_wino16_one<KS, D, 32, FV_X_PW_CH32>(p, batch, s);  \
        if constexpr (KS == 3) {                                                           \
            if (C == 64) return launch_pair_wino32_one<KS, D, 64, FV_X_PW_CH64, FV_X_PW_XRES64 != 0, FV_X_PW_DBUF != 0>(p, batch, s);   \
            if (C == 128) return launch_pair_wino32_one<KS, D, 128, FV_X_PW_CH128, false, FV_X_PW_DBUF != 0>(p, batch, s);              \
        }                                                                                  \
    }
    FV_PW_CASE(1) FV_PW_CASE(3) FV_PW_CASE(5)
#undef FV_PW_CASE
    return false;
}

}  // namespace fv
