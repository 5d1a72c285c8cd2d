// f16x3 precision mode, last stage (C = 16): one whole ResBlock1 iteration
//
//     y = x + c2( silu( c1( silu(x) ) ) )                (fish_vocoder/modules/generators/hifigan.py:102-107)
//
// in ONE launch on the split-fp16 matrix path.  Sixteen output channels fill only half of the 32 rows of
// v_mfma_f32_32x32x16_f16, so the M axis carries TWO output times per channel instead: row (s, co) of column n is
//
//     out[co][f(n) + s*d]      with f(n) = 2d * (n / d) + n % d        (pairs of times one dilation step apart)
//
// and, because out[co][t + s*d] = sum_j w[co][:, j] . x[:, t + (s + j) * d], both rows read the SAME activation fragment
// x[:, f(n) + jj*d] for the combined tap index jj = s + j in [0, KS]: the A operand of tap block jj holds w[.., jj - s]
// (zero where jj - s falls outside [0, KS)).  One conv therefore costs KS + 1 k-blocks of 16 channels per 64 output
// samples — the rows are KS / (KS + 1) full instead of half empty (a 16x16x32 formulation needs the same MFMA time but
// twice the LDS fragment reads).
//
// For the fragment read "x[:, f(n) + jj*d] for 32 consecutive n" to be 32 consecutive 16-byte slots, the window lives in LDS
// split by the parity of (tau / d): half = (tau / d) & 1, index = (tau / 2d) * d + tau % d, tau = time relative to the
// window start.  Then column n, tap jj  ->  half jj & 1, index n + (jj >> 1) * d.  c2 (dilation 1) reads the intermediate
// from the same kind of layout with d = 1 (even / odd samples), which is where c1's epilogue writes it (bias, SiLU, zero
// outside [0, T), (hi, lo) split) — it never leaves the CU.  C = 16 is a single 16-channel chunk: one staging pass, no
// chunk loop, two barriers per tile.
//
// Workgroup = 4 waves x 2 n-tiles = 256 column pairs = 2d * (256 / d) intermediate samples (510 or 512); tiles advance by
// TT = that - (KS - 1) final samples.  LDS: max(window, intermediate) planes = ~37 KB.
#include "conv_f16x3_impl.h"
#include "pair_f16x3_params.h"

namespace fv {

template <int KS, int DIL1>
struct Pair16Geom {
    static constexpr int NT = 2;
    static constexpr int NPB = 4 * NT * 32;                 // column pairs per workgroup
    static constexpr int G1 = NPB / DIL1;                   // complete groups of d pairs (= 2d samples) in c1's columns
    static constexpr int TH = 2 * DIL1 * G1;                // intermediate samples per workgroup
    static constexpr int TT = TH - (KS - 1);                // final samples per workgroup (even)
    static constexpr int H2 = (KS - 1) / 2, H1 = H2 * DIL1;
    static constexpr int W1 = TH + (KS - 1) * DIL1;         // staged x samples per channel row
    // slots per (plane, channel half, parity half): the window's own extent, and what the fragment reads of the
    // dropped / incomplete columns may touch (index n + (jj >> 1) * d for n < NPB, jj <= KS)
    static constexpr int HW1 = ((W1 + 2 * DIL1 - 1) / (2 * DIL1)) * DIL1 > NPB + (KS / 2) * DIL1
                                   ? ((W1 + 2 * DIL1 - 1) / (2 * DIL1)) * DIL1 : NPB + (KS / 2) * DIL1;
    static constexpr int HW2 = NPB + KS / 2 + 1;
    static constexpr int XS_SLOTS = 8 * HW1, HS_SLOTS = 8 * HW2;   // [plane][channel half][parity half][index]
    static constexpr size_t LDS_BYTES = (size_t)(XS_SLOTS > HS_SLOTS ? XS_SLOTS : HS_SLOTS) * 16;
};

template <int KS, int DIL1>
__global__ __launch_bounds__(256, 2) void pair16_f16x3_kernel(const PairF16Params p) {
    using G = Pair16Geom<KS, DIL1>;
    constexpr int C = 16, NT = G::NT, D = DIL1;
    constexpr int TT = G::TT, W1 = G::W1, HW1 = G::HW1, HW2 = G::HW2;
    constexpr int ITEMS = 2 * W1;                       // (channel half, sample) staging items of 8 channels each
    constexpr int NE = (ITEMS + 255) / 256;
    constexpr int KB = KS + 1;                          // tap blocks per conv
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    h8* xs = reinterpret_cast<h8*>(lds_raw);            // [plane][channel half][parity half][HW1]
    h8* hs = xs;                                        // [plane][channel half][parity half][HW2], overlays xs

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x % p.n_tiles, b = blockIdx.x / p.n_tiles;
    const int t0 = tile * TT;                           // first final sample of the tile
    const int th0 = t0 - G::H2;                         // first intermediate sample
    const int tb = th0 - G::H1;                         // first staged x sample
    const float* __restrict__ xb = p.x + (long long)b * C * p.T;
    const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(xb, (unsigned)((long long)C * p.T * 4));

    // ---- stage silu(x) as (hi, lo) planes in the parity-split layout ----
    float stage[NE][8];
    const unsigned row_b = (unsigned)p.T * 4u;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        int e = tid + i * 256;
        e = e < ITEMS ? e : ITEMS - 1;
        const int hh = e / W1;
        const int t = tb + (e - hh * W1);
        const unsigned off = (t >= 0 && t < p.T) ? (unsigned)(8 * hh * p.T + t) * 4u : 0xC0000000u;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            stage[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, off + (unsigned)r * row_b, 0, 0));
    }
    const int wvoff = lane * 16;
    auto load_a = [&](const __amdgpu_buffer_rsrc_t& rs, h8 (&dst)[2], int blk) {   // tap block -> (wh, wl)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, wvoff + q * 1024, blk * 2048, 0);
            dst[q] = __builtin_bit_cast(h8, v);
        }
    };
    constexpr int DA = kF16WeightPrefetch;
    constexpr int RA = DA + 1;
    h8 aq[RA][2];
    h8 bq[2][NT][2];
    const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1h, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2h, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int d = 0; d < DA; ++d) load_a(w1rs, aq[d], d);
#pragma unroll
    for (int i = 0; i < NE; ++i) {   // silu, split, pack (element-wise on purpose, see conv_f16x3_impl.h)
        const int e = tid + i * 256;
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
            const int hh = e / W1;
            const int tau = e - hh * W1;
            const int g = tau / D, r = tau - g * D;     // constant divisor
            const int slot = (hh * 2 + (g & 1)) * HW1 + (g >> 1) * D + r;
            xs[slot] = hi;
            xs[4 * HW1 + slot] = lo;
        }
    }
    __syncthreads();

    f32x16 acc[NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    };
    auto mfma_block = [&](const h8 (&a)[2], const h8 (&bf)[NT][2]) {
        const h8 a_sc = a[0] * (_Float16)(1.0f / 2048.0f);   // wh * 2^-11, exact
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int jn = 0; jn < NT; ++jn)
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q == 2 ? a_sc : a[q], bf[jn][q == 2 ? 1 : 0], acc[jn], 0, 0, 0);
    };
    // B fragment of column n = ncol0 + jn * 32 + (lane & 31), tap block jj: 8 channels of half (lane >> 5) at
    // parity half jj & 1, index n + (jj >> 1) * dil
    const int ncol0 = wn * (NT * 32) + (lane & 31);
    auto conv_loop = [&](const __amdgpu_buffer_rsrc_t& wrs, const h8* src, int hw, int dil) {
        const int lane_slot = (lane >> 5) * 2 * hw + ncol0;
        auto load_b = [&](h8 (&dst)[NT][2], int jj) {
            const h8* s = src + lane_slot + (jj & 1) * hw + (jj >> 1) * dil;
#pragma unroll
            for (int q = 1; q >= 0; --q)
#pragma unroll
                for (int jn = NT - 1; jn >= 0; --jn) dst[jn][q] = s[q * 4 * hw + jn * 32];
        };
        load_b(bq[0], 0);
#pragma unroll
        for (int jj = 0; jj < KB; ++jj) {
            load_a(wrs, aq[(jj + DA) % RA], jj + DA);   // runs DA blocks past the end: the packed planes are padded
            if (jj + 1 < KB) load_b(bq[(jj + 1) & 1], jj + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(aq[jj % RA], bq[jj & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ------------------------------------------------------------------ c1
    zero_acc();
    conv_loop(w1rs, xs, HW1, D);
    __syncthreads();   // every wave is done with the x window: the intermediate planes may overwrite it
#pragma unroll
    for (int d = 0; d < DA; ++d) load_a(w2rs, aq[d], d);
    {
        // lane (half hh, column n) holds rows 8*rq + 4*hh + rr: s = rq >> 1, channels 8*(rq & 1) + 4*hh + (0..3)
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const int hh = lane >> 5;
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            const int n = wn * (NT * 32) + jn * 32 + (lane & 31);
            const int g = n / D, r = n - g * D;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int s = rq >> 1, cg = rq & 1;
                const int tau = 2 * D * g + r + s * D;          // intermediate sample relative to th0
                const int pos = th0 + tau;
                const bool live = g < G::G1 && pos >= 0 && pos < p.T;
                h4 vh4, vl4;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float v = fmaf(acc[jn][rq * 4 + rr], p.s1, p.b1[8 * cg + 4 * hh + rr]);
                    v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                    v = live ? v : 0.f;
                    const _Float16 vh = (_Float16)v;
                    vh4[rr] = vh;
                    vl4[rr] = (_Float16)((v - (float)vh) * 2048.0f);
                }
                if (g < G::G1) {   // columns of the incomplete last group have no slot
                    const int slot = (cg * 2 + (tau & 1)) * HW2 + (tau >> 1);
                    *(reinterpret_cast<h4*>(&hs[slot]) + hh) = vh4;
                    *(reinterpret_cast<h4*>(&hs[4 * HW2 + slot]) + hh) = vl4;
                }
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ c2 (dilation 1: pairs of adjacent samples)
    zero_acc();
    conv_loop(w2rs, hs, HW2, 1);
    {
        const unsigned span = (unsigned)((long long)C * p.T * 4);
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * C * p.T, span);
        const bool accum = p.out_mode == OUT_ACCUM;
        const int hh = lane >> 5;
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            const int n = wn * (NT * 32) + jn * 32 + (lane & 31);
            const int t = t0 + 2 * n;                   // samples (t, t + 1): rows s = 0 / 1; T and TT are even
            const bool ok = 2 * n < TT && t < p.T;
            u32x2 rv[2][4], yo[2][4];
            unsigned off[2][4];
#pragma unroll
            for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int co = 8 * cg + 4 * hh + rr;
                    off[cg][rr] = ok ? (unsigned)(co * p.T + t) * 4u : 0xFFFFFFF8u;
                    rv[cg][rr] = __builtin_amdgcn_raw_buffer_load_b64(xrs, off[cg][rr], 0, 0);
                }
            if (accum) {
#pragma unroll
                for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) yo[cg][rr] = __builtin_amdgcn_raw_buffer_load_b64(yrs, off[cg][rr], 0, 0);
            }
#pragma unroll
            for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float bias = p.b2[8 * cg + 4 * hh + rr];
                    float v0 = fmaf(acc[jn][cg * 4 + rr], p.s2, bias) + __uint_as_float(rv[cg][rr].x);
                    float v1 = fmaf(acc[jn][(cg + 2) * 4 + rr], p.s2, bias) + __uint_as_float(rv[cg][rr].y);
                    if (accum) {
                        v0 = (__uint_as_float(yo[cg][rr].x) + v0) * p.out_scale;
                        v1 = (__uint_as_float(yo[cg][rr].y) + v1) * p.out_scale;
                    }
                    u32x2 o;
                    o.x = __float_as_uint(v0);
                    o.y = __float_as_uint(v1);
                    __builtin_amdgcn_raw_buffer_store_b64(o, yrs, off[cg][rr], 0, 0);
                }
        }
    }
}

template <int KS, int DIL1>
static bool launch_pair16_one(PairF16Params q, int batch, hipStream_t s) {
    using G = Pair16Geom<KS, DIL1>;
    q.n_tiles = (q.T + G::TT - 1) / G::TT;
    if (!FV_ENSURE_DYN_LDS((pair16_f16x3_kernel<KS, DIL1>), G::LDS_BYTES)) return false;
    hipLaunchKernelGGL((pair16_f16x3_kernel<KS, DIL1>), dim3(batch * q.n_tiles), dim3(256), G::LDS_BYTES, s, q);
    return true;
}

template <int KS>
static bool launch_pair16_k(const PairF16Params& p, int dil1, int batch, hipStream_t s) {
    switch (dil1) {
        case 1: return launch_pair16_one<KS, 1>(p, batch, s);
        case 3: return launch_pair16_one<KS, 3>(p, batch, s);
        case 5: return launch_pair16_one<KS, 5>(p, batch, s);
        default: return false;
    }
}

bool launch_pair16_f16x3(const PairF16Params& p, int ks, int dil1, int batch, hipStream_t s) {
    switch (ks) {
        case 3: return launch_pair16_k<3>(p, dil1, batch, s);
        case 7: return launch_pair16_k<7>(p, dil1, batch, s);
        case 11: return launch_pair16_k<11>(p, dil1, batch, s);
        default: return false;
    }
}

int pair16_f16x3_tile(int ks, int dil1) {   // final samples per workgroup (profiling labels)
    const int th = 2 * dil1 * (256 / dil1);
    return th - (ks - 1);
}

}  // namespace fv
