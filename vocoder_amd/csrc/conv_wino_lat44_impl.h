// Latency variant of the Winograd F(4,4) conv (conv_wino44_impl.h) for launches that cannot fill the chip — the single clip of the reference's
// inference call (fish_vocoder/test.py:88-90: one utterance per forward), small batches: the dilated "same" Conv1d of a ResBlock / AMPBlock
// (hifigan.py:101-108), C_in = C_out = C in {64, 128, 256, ...}, k = 7 / 11 (k = 3 stays on conv_wino_lat_impl.h's F(2,3) groups).
//
// The single-clip forward runs its three ResBlock branches side by side and is bound by the matrix work they issue together, not by one
// kernel's chain (profiles/r05p_b1_timeline.txt: 1.21 ms of kernels in 0.71 ms), so the lever is the same as in the batch kernels: 20 / 13
// products per FOUR outputs on the quad lattice instead of conv_wino_lat's 32 / 20 (direct: 44 / 28).  Structure of conv_wino_lat_kernel:
//   * v_mfma_f32_16x16x4_f32; one workgroup = 16 MT output rows x 16 quad columns (64 outputs; quad column n = q D + r <-> u0 = 4 D q + r,
//     outputs u0 + j D; the lattice is global, whatever the tile's first column);
//   * the four waves split K by 8-channel blocks (wave w owns blocks w, w + 4, ...) and are INDEPENDENT until the final reduction: each stages
//     (four phases X0..X3 of its block's window), transforms (pair_wino44_impl.h's symmetric points: seven V planes) and multiplies its own
//     blocks in a wave-private LDS region — LDS operations of one wave execute in order, so neither step needs a barrier;
//   * weights: the layer's d_wpq16 (F(4,4)-transformed 16x16x4 fragments: virtual tap v = group v / 7, plane v % 7; one float4 per lane = taps
//     2 f, 2 f + 1 x k-steps 0, 1), ring of >= 7 fragments as in conv_wino_lat_kernel;
//   * output transform per wave (linear: before the reduction), reduction over the four waves' K shares through LDS, epilogue (bias, residual,
//     activation, accumulate); at D = 1 a lane's four outputs are consecutive samples: 16-byte residual loads / stores.
#pragma once
#include "pair_wino44_impl.h"

namespace fv {

template <int KS, int DIL, int MT>
struct WL4Geom {
    static_assert(KS == 7 || KS == 11, "k = 3: 6 products per quad against 8 — stays on F(2,3)");
    static constexpr int NG = (KS + 3) / 4, NSH = NG - 1;
    static constexpr int NV = 7 * NSH + 6;                    // virtual taps per channel (the last group has no inf tap)
    static constexpr int NF = (NV + 1) / 2;                   // weight fragments per 8-channel block
    static constexpr int NBQ = 16;                            // quad columns per workgroup
    static constexpr int WD = NBQ + DIL * (NG - 1);           // V plane columns (group g reads column n + g D)
    static constexpr int WR = WD + DIL;                       // X plane columns (the transform reads n and n + D)
    static constexpr int NP = 4 * WR;                         // staged positions per channel row
    static constexpr int PX = (WR + 3) / 8 * 8 + 4;           // X plane stride: >= WR, == 4 (mod 8) ...
    static constexpr int SX = 4 * PX;                         // ... so a channel row X0 X1 X2 X3 is == 16 (mod 32): the transform's reads (two rows per 32 lanes)
    static constexpr int X_F = 8 * SX;
    static constexpr int PV = pw_up(WD, 32, 16);              // V row stride == 16 (mod 32): the 16x16x4 B read
    static constexpr int PS = 8 * PV;                         // V plane [channel][column]
    static constexpr int WAVE_F = X_F + 7 * PS;
    static constexpr int NSLOT = (8 * NP + 63) / 64;          // staged elements per lane and block
    static constexpr int NTR = (8 * WD + 63) / 64;            // transform: slots of 64 lattice elements per block
#ifndef FV_X_LAT44_RING
#define FV_X_LAT44_RING 7
#endif
    static constexpr int NEED = (FV_X_LAT44_RING + MT - 1) / MT;
    static constexpr int MU = (NEED + 1 + NF - 1) / NF;       // blocks per unrolled loop iteration
    static constexpr int RA = MU * NF;                        // ring slots (fragments), prefetch distance RA - 1
    static_assert(PX >= WR && PV >= WD, "plane strides");
    static_assert(4 * 4 * 4 * MT * 64 <= 4 * WAVE_F, "the exchange buffer fits the planes");
    static constexpr int a_of(int v) { return v % 7; }
    static constexpr int off_of(int v, int s) { return (v % 7) * PS + 4 * s * PV + (v / 7) * DIL; }
};

template <int KS, int DIL, int MT>
__global__ __launch_bounds__(256, 2) void conv_wino_lat44_kernel(const ConvParams p) {
    using G = WL4Geom<KS, DIL, MT>;
    constexpr int NF = G::NF, NV = G::NV, NSLOT = G::NSLOT, PX = G::PX, SX = G::SX, PV = G::PV, PS = G::PS, MU = G::MU, RA = G::RA;
    __shared__ __attribute__((aligned(16))) float lds[4 * G::WAVE_F];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* __restrict__ X = lds + wave * G::WAVE_F;           // this wave's X planes [channel][phase][column] ...
    float* __restrict__ V = X + G::X_F;                       // ... and V planes [plane][channel][column]
    int bid = (int)blockIdx.x;
    const int n_tile = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int mt = (bid % p.m_blks) * MT;                     // first of the workgroup's MT 16-row tiles
    const int b = bid / p.m_blks;
    const int n0 = n_tile * G::NBQ;
    const int Tin = p.Tin;
    const int nblk = p.Cin >> 3;                              // 8-channel blocks; this wave's: wave, wave + 4, ...
    const int nsteps = (nblk - wave + 3) >> 2;

    // ---- staging plan: element e = lane + 64 i of the block's [8 rows][NP positions] -> global byte offset inside the block (0xFFFFFFFF outside
    // [0, Tin): the load returns 0, and act(0) == 0 is the conv's zero padding) and LDS offset in the X planes.  D = 1: consecutive elements are
    // consecutive samples (phase = element & 3); D > 1: phase by phase, runs of D consecutive samples ----
    unsigned voff[NSLOT];
    int loff[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        int e = lane + 64 * i;
        e = e < 8 * G::NP ? e : 8 * G::NP - 1;                // (surplus lanes repeat the last element: same load, same value, same address)
        const int row = e / G::NP, c = e - row * G::NP;
        const int j = DIL == 1 ? (c & 3) : c / G::WR, cc = DIL == 1 ? (c >> 2) : c - j * G::WR;
        const int n = n0 + cc;
        const int q = n / DIL;
        const int t = 4 * DIL * q + (n - q * DIL) + j * DIL - p.pad_l;
        voff[i] = (t >= 0 && t < Tin) ? (unsigned)(row * Tin + t) * 4u : 0xFFFFFFFFu;
        loff[i] = row * SX + j * PX + cc;
    }
    const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(p.x + (long long)b * p.x_bstride, (unsigned)(p.x_bstride * 4));
    float sv[NSLOT];
    auto load_block = [&](int blk) __attribute__((always_inline)) {
        const int so = __builtin_amdgcn_readfirstlane(blk * 8 * Tin * 4);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) sv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, voff[i], so, 0));
    };

    // ---- weights: fragment (blk, f) of m-tile mt at ((mt * nblk + blk) * NF + f) KiB ----
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    auto wbase_of = [&](int blk) { return __builtin_amdgcn_readfirstlane((mt * nblk + blk) * NF * 1024); };   // (m-tile mt + i: + i * mt_stride)
    const int mt_stride = __builtin_amdgcn_readfirstlane(nblk * NF * 1024);
    float4 aq[RA][MT];
    auto load_w = [&](int base, int f) __attribute__((always_inline)) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, base + f * 1024, 0);
        return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    };

    f32x4w acc[7][MT];
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[a][i] = f32x4w{0.f, 0.f, 0.f, 0.f};

    const int krow = lane >> 4, col = lane & 15;
    const float* bl = V + krow * PV + col;                   // B operand: channel row 4 s + krow, quad column col (+ the tap's offset)
    // transform slots: lattice element e = lane + 64 g of the block's [8 channel rows][WD columns]
    int txo[G::NTR], tvo[G::NTR];
#pragma unroll
    for (int g = 0; g < G::NTR; ++g) {
        int e = lane + 64 * g;
        e = e < 8 * G::WD ? e : 8 * G::WD - 1;
        const int tr = e / G::WD, tc = e - tr * G::WD;
        txo[g] = tr * SX + tc;
        tvo[g] = tr * PV + tc;
    }

    if (nsteps > 0) {
        load_block(wave);
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
                    // X planes of this block (the previous block's transform has issued its last LDS read: in-order, no barrier)
                    act_apply_all(sv, p.pre_act, p.slope);
#pragma unroll
                    for (int i = 0; i < NSLOT; ++i) X[loff[i]] = sv[i];
                    if (s + 1 < nsteps) load_block(blk + 4);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // X -> seven V planes (the previous block's matrix instructions have issued their last LDS read).  The block's 8 x WD lattice elements are
                    // flattened over the lanes: NTR slots of 64 (WD = 17 ... 26: three or four) instead of two 16-column groups per row half whose second
                    // one holds 1 ... 10 live columns
#pragma unroll
                    for (int g = 0; g < G::NTR; ++g) {
                        if (64 * (g + 1) <= 8 * G::WD || 64 * g + lane < 8 * G::WD) {
                            const float* xp = X + txo[g];
                            float* vp = V + tvo[g];
                            const float x0 = xp[0], x1 = xp[PX], x2 = xp[2 * PX], x3 = xp[3 * PX];
                            const float x4 = xp[DIL], x5 = xp[PX + DIL], x6 = xp[2 * PX + DIL];
                            const float eh = fmaf(4.0f, x0, fmaf(-5.0f, x2, x4)), oh = fmaf(4.0f, x1, fmaf(-5.0f, x3, x5));          // a = 1/2
                            const float e1 = fmaf(-4.25f, x2, x4) + x0, o1 = fmaf(-4.25f, x3, x5) + x1;                              // a = 1
                            const float e2 = fmaf(0.25f, x0, fmaf(-1.25f, x2, x4)), o2 = fmaf(0.25f, x1, fmaf(-1.25f, x3, x5));      // a = 2
                            vp[0] = fmaf(0.5f, eh, oh);
                            vp[PS] = fmaf(-0.5f, eh, oh);
                            vp[2 * PS] = o1 + e1;
                            vp[3 * PS] = o1 - e1;
                            vp[4 * PS] = fmaf(2.0f, e2, o2);
                            vp[5 * PS] = fmaf(-2.0f, e2, o2);
                            vp[6 * PS] = fmaf(5.25f, x2 - x4, x6 - x0);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // byte offsets of the blocks the prefetch reaches (this one + 1 ... MU ahead; past the wave's last block: that block again)
                    int wbs[MU + 1];
#pragma unroll
                    for (int j = 0; j <= MU; ++j) wbs[j] = wbase_clamped(blk + 4 * j);
                    float b_cur[4], b_nxt[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h) b_cur[h] = bl[G::off_of(h & 1, h >> 1)];   // h = tap-of-pair + 2 * k-step
                    static_for<NF>([&](auto f_c) __attribute__((always_inline)) {
                        constexpr int f = decltype(f_c)::value;
                        constexpr int g = u * NF + f;           // position in the ring's period
#pragma unroll
                        for (int m = 0; m < 4 * MT; ++m) {
                            // order: m-tile fastest, then (tap 0, ks 0) (tap 1, ks 0) (tap 0, ks 1) (tap 1, ks 1): consecutive MFMAs never share an accumulator
                            const int i = m % MT, h = m / MT;
                            const int tap = h & 1, ks = h >> 1;
                            const int v = 2 * f + tap;
                            if (v < NV) {
                                const float4 a4 = aq[g % RA][i];
                                float apin = tap == 0 ? (ks == 0 ? a4.x : a4.y) : (ks == 0 ? a4.z : a4.w);
                                asm volatile("" : "+v"(apin));   // (pins the MFMA between the memory operations around it: pair_wino_impl.h)
                                const int A = G::a_of(v);
                                acc[A][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(apin, b_cur[h], acc[A][i], 0, 0, 0);
                                asm volatile("" : "+v"(acc[A][i]));
                            }
                            if (m < MT) {
                                constexpr int fn = f + RA - 1;   // fragment to request, counted from this block's first one
                                aq[(g + RA - 1) % RA][m] = load_w(wbs[fn / NF] + m * mt_stride, fn % NF);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (f + 1 < NF && i == MT - 1) {
                                const int vn = 2 * (f + 1) + tap;
                                if (vn < NV) b_nxt[h] = bl[G::off_of(vn, ks)];
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (f + 1 < NF) {
#pragma unroll
                            for (int h = 0; h < 4; ++h) b_cur[h] = b_nxt[h];
                        }
                    });
                }
            });
        }
    }

    // ---- output transform (per wave: it is linear), reduction over the four waves' K shares, epilogue ----
    // wave w ends up owning accumulator register w of every lane: rows 16 (mt + i) + 4 (lane >> 4) + w, quad column col
    __syncthreads();                                        // every wave is done with its planes: they become the exchange buffer
    float* E = lds;                                         // [wave][register][m-tile, output of the quad][64 lanes]
    constexpr int NV4 = 4 * MT;                             // values per lane and accumulator register
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        f32x4w m7[7], y[4];
#pragma unroll
        for (int a = 0; a < 7; ++a) m7[a] = acc[a][i];
        pq_output_transform(m7, y);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) E[((wave * 4 + r) * NV4 + i * 4 + j) * 64 + lane] = y[j][r];
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * p.y_bstride, (unsigned)(p.y_bstride * 4));
    const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(p.res ? p.res + (long long)b * p.y_bstride : p.y, (unsigned)(p.y_bstride * 4));
    float val[NV4];
    unsigned off[NV4];
    const int n = n0 + col;
    const int q = n / DIL;
    const int ta = 4 * DIL * q + (n - q * DIL);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = 16 * (mt + i) + 4 * krow + wave;
        const float bias = p.bias[m < p.M ? m : 0];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = ta + j * DIL;
            const int e = i * 4 + j;
            off[e] = (t < p.N && m < p.M) ? (unsigned)(m * p.N + t) * 4u : 0xFFFFFFFFu;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += E[((w * 4 + wave) * NV4 + e) * 64 + lane];   // (fixed order: waves 0, 1, 2, 3)
            val[e] = v + bias;
        }
    }
    // D = 1: the lane's four outputs are samples 4 n .. 4 n + 3 of the row — whole 16-byte quads when rows are a multiple of four samples long
    const bool quad16 = DIL == 1 && (p.N & 3) == 0 && (((unsigned long long)p.y | (unsigned long long)(p.res ? p.res : p.y)) & 15ull) == 0 &&
                        (p.y_bstride & 3) == 0;
    if (quad16) {
        if (p.res) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(rrs, off[4 * i], 0, 0);
                val[4 * i] += __uint_as_float(r4.x);
                val[4 * i + 1] += __uint_as_float(r4.y);
                val[4 * i + 2] += __uint_as_float(r4.z);
                val[4 * i + 3] += __uint_as_float(r4.w);
            }
        }
        act_apply_all(val, p.post_act, p.slope);
        if (p.out_mode == OUT_ACCUM) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const u32x4 o4 = __builtin_amdgcn_raw_buffer_load_b128(yrs, off[4 * i], 0, 0);
                val[4 * i] = (__uint_as_float(o4.x) + val[4 * i]) * p.out_scale;
                val[4 * i + 1] = (__uint_as_float(o4.y) + val[4 * i + 1]) * p.out_scale;
                val[4 * i + 2] = (__uint_as_float(o4.z) + val[4 * i + 2]) * p.out_scale;
                val[4 * i + 3] = (__uint_as_float(o4.w) + val[4 * i + 3]) * p.out_scale;
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            u32x4 o;
            o.x = __float_as_uint(val[4 * i]);
            o.y = __float_as_uint(val[4 * i + 1]);
            o.z = __float_as_uint(val[4 * i + 2]);
            o.w = __float_as_uint(val[4 * i + 3]);
            __builtin_amdgcn_raw_buffer_store_b128(o, yrs, off[4 * i], 0, 0);
        }
        return;
    }
    if (p.res) {
#pragma unroll
        for (int e = 0; e < NV4; ++e) val[e] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, off[e], 0, 0));
    }
    act_apply_all(val, p.post_act, p.slope);
    if (p.out_mode == OUT_ACCUM) {
#pragma unroll
        for (int e = 0; e < NV4; ++e)
            val[e] = (__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, off[e], 0, 0)) + val[e]) * p.out_scale;
    }
#pragma unroll
    for (int e = 0; e < NV4; ++e) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[e]), yrs, off[e], 0, 0);
}

// rows: 16 or 32 per workgroup (p.m_blks counts these tiles, p.n_tiles tiles of 16 quad columns)
template <int KS, int DIL>
inline bool launch_wino_lat44_mt(const ConvParams& p, int rows, int batch, hipStream_t s) {
    const int grid = batch * p.m_blks * p.n_tiles;
    if (rows == 16) hipLaunchKernelGGL((conv_wino_lat44_kernel<KS, DIL, 1>), dim3(grid), dim3(256), 0, s, p);
    else if (rows == 32) hipLaunchKernelGGL((conv_wino_lat44_kernel<KS, DIL, 2>), dim3(grid), dim3(256), 0, s, p);
    else return false;
    return true;
}

template <int KS>
inline bool launch_wino_lat44_k(const ConvParams& p, int rows, int batch, hipStream_t s) {
    switch (p.dil) {
        case 1: return launch_wino_lat44_mt<KS, 1>(p, rows, batch, s);
        case 3: return launch_wino_lat44_mt<KS, 3>(p, rows, batch, s);
        case 5: return launch_wino_lat44_mt<KS, 5>(p, rows, batch, s);
        default: return false;
    }
}

}  // namespace fv
