// Host side of the fused conv layer: weight re-layout into MFMA fragment order, tile selection, dispatch.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "fv_internal.h"
#include "pair_f16x3_params.h"

namespace fv {

void tile_dims(int cfg, int* m_blk, int* n_blk) {
    static const int dims[TILE_COUNT][2] = {{128, 128}, {64, 256}, {32, 512}, {128, 64}, {32, 128}, {64, 128}, {32, 64}, {32, 32}, {256, 64}, {256, 32}, {128, 96}};
    *m_blk = dims[cfg][0];
    *n_blk = dims[cfg][1];
}

// Packed layout (float4 units):  [(m_tile * nchunk + chunk) * ks + tap] * 64 + lane
//   .{x,y,z,w}[pp] = Wc[m_tile*32 + (lane & 31)][chunk*8 + 2*pp + (lane >> 5)][tap]     (0 outside M x Cin)
// i.e. exactly the A fragment of v_mfma_f32_32x32x2_f32 for k-step (tap, channel pair pp): lane l supplies
// row l&31, k-half l>>5 — one coalesced 1 KiB global_load_dwordx4 per wave per tap.
static void pack_conv_weights(const std::vector<float>& wc, int M, int Cin, int ks, int m_pad, int nchunk,
                              std::vector<float>& out) {
    const int mtiles = m_pad / 32;
    // + 8 zero k-steps after the last m-tile: the kernels' weight prefetch runs a few steps past the end
    out.assign(((size_t)mtiles * nchunk * ks + 8) * 64 * 4, 0.f);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int c = 0; c < nchunk; ++c)
            for (int j = 0; j < ks; ++j)
                for (int l = 0; l < 64; ++l) {
                    const int m = mt * 32 + (l & 31);
                    if (m >= M) continue;
                    float* dst = &out[((((size_t)mt * nchunk + c) * ks + j) * 64 + l) * 4];
                    for (int pp = 0; pp < 4; ++pp) {
                        const int ci = c * kChunk + 2 * pp + (l >> 5);
                        if (ci < Cin) dst[pp] = wc[((size_t)m * Cin + ci) * ks + j];
                    }
                }
}

// f16x3 precision mode (conv_f16x3_impl.h).  Packed layout, 16-byte units (8 halfs):
//   [((m_tile * nch16 + chunk) * ks + tap) * 2 + plane] * 64 + lane,   halfs i = 0..7:
//   plane(Wc[m_tile*32 + (lane & 31)][chunk*16 + 8*(lane >> 5) + i][tap] * s_w),   planes: wh, wl = w*s_w - wh
//   (the third operand plane wh * 2^-11 is a packed multiply in registers: one third less weight traffic per MFMA)
// = the A fragment of v_mfma_f32_32x32x16_f16 for k-block (chunk, tap).  s_w: power of two with max|w| * s_w in [2^13, 2^14).
static float pack_conv_weights_f16x3(const std::vector<float>& wc, int M, int Cin, int ks, int m_pad, int nch16,
                                     std::vector<_Float16>& out) {
    float wmax = 0.f;
    for (float v : wc) wmax = std::max(wmax, std::fabs(v));
    int e = 0;
    if (wmax > 0.f && std::isfinite(wmax)) {
        (void)std::frexp(wmax, &e);   // wmax = f * 2^e, f in [0.5, 1)
        e = 14 - e;                   // wmax * 2^e in [2^13, 2^14)
    }
    const float s_w = std::ldexp(1.0f, e);
    const int mtiles = m_pad / 32;
    // + 4 zero k-blocks after the last m-tile: the weight prefetch runs a few blocks past the end
    out.assign(((size_t)mtiles * nch16 * ks + 4) * 2 * 64 * 8, (_Float16)0.f);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int c = 0; c < nch16; ++c)
            for (int j = 0; j < ks; ++j)
                for (int l = 0; l < 64; ++l) {
                    const int m = mt * 32 + (l & 31);
                    if (m >= M) continue;
                    const size_t blk = (((size_t)mt * nch16 + c) * ks + j) * 2;
                    for (int i = 0; i < 8; ++i) {
                        const int ci = c * 16 + 8 * (l >> 5) + i;
                        if (ci >= Cin) continue;
                        const float w = wc[((size_t)m * Cin + ci) * ks + j] * s_w;   // exact: power-of-two scale
                        const _Float16 wh = (_Float16)w;
                        const _Float16 wl = (_Float16)(w - (float)wh);
                        out[((blk + 0) * 64 + l) * 8 + i] = wh;
                        out[((blk + 1) * 64 + l) * 8 + i] = wl;
                    }
                }
    return s_w;
}

// layers the split-fp16 kernels cover: MFMA-bound stride-1 convs with the ResBlock kernel sizes
static bool f16x3_eligible(bool transposed, int c_in, int M, int ks, int dil) {
    // polyphase transposed convs: 1, 2 or 4 taps per phase, M = C_out * stride rows
    if (transposed) return c_in >= 32 && c_in % 16 == 0 && M >= 128 && (ks == 1 || ks == 2 || ks == 4);
    // pointwise convs (ConvNeXt GEMMs): whole 64-channel LDS chunks, 128-row tiles
    if (ks == 1) return dil == 1 && c_in >= 64 && c_in % 64 == 0 && M >= 128;
    return c_in >= 32 && M >= 32 && (ks == 3 || ks == 7 || ks == 11) && (dil == 1 || dil == 3 || dil == 5);
}

// the per-layer split-fp16 kernel tiles 32, 64 or 128 rows (f16x3_eligible() keeps narrower layers out)
static bool f16x3_per_layer_ok(const ConvLayer& L) { return L.M >= 32; }

fv_status conv_layer_create(ConvLayer& L, bool transposed, int c_in, int c_out, int k, int dil, int padding,
                            int stride, const float* host_w, const float* host_bias, bool with_f16x3) {
    if (c_in <= 0 || c_out <= 0 || k <= 0 || dil <= 0 || stride <= 0 || padding < 0) {
        set_error("conv_layer_create: invalid geometry (c_in=%d c_out=%d k=%d dil=%d pad=%d stride=%d)", c_in, c_out,
                  k, dil, padding, stride);
        return FV_ERR_INVALID;
    }
    L.transposed = transposed;
    L.c_in = c_in;
    L.c_out = c_out;
    L.k = k;
    L.dil = transposed ? 1 : dil;
    L.padding = padding;
    L.stride = transposed ? stride : 1;

    std::vector<float> wc, bias;
    if (!transposed) {
        L.M = c_out;
        L.ks = k;
        L.pad_l = padding;
        wc.assign(host_w, host_w + (size_t)c_out * c_in * k);
        bias.assign(c_out, 0.f);
        if (host_bias) std::copy(host_bias, host_bias + c_out, bias.begin());
    } else {
        // polyphase: row (co, r), taps j' = 0..ks'-1 reading x[q + j' - (ks'-1)]; torch weight is (C_in, C_out, k)
        const int u = stride;
        const int ksp = (k + u - 1) / u;
        L.M = c_out * u;
        L.ks = ksp;
        L.pad_l = ksp - 1;
        wc.assign((size_t)L.M * c_in * ksp, 0.f);
        bias.assign(L.M, 0.f);
        for (int co = 0; co < c_out; ++co)
            for (int r = 0; r < u; ++r) {
                const int m = co * u + r;
                if (host_bias) bias[m] = host_bias[co];
                for (int ci = 0; ci < c_in; ++ci)
                    for (int jp = 0; jp < ksp; ++jp) {
                        const int tap = r + (ksp - 1 - jp) * u;
                        if (tap < k) wc[((size_t)m * c_in + ci) * ksp + jp] = host_w[((size_t)ci * c_out + co) * k + tap];
                    }
            }
    }
    L.nchunk_real = (c_in + kChunk - 1) / kChunk;
    L.nchunk = (L.nchunk_real + 3) / 4 * 4;   // whole LDS chunks for every SUBS in {1, 2, 4} (zero weights)
    L.m_pad = (L.M + 127) / 128 * 128;
    std::vector<float> packed;
    pack_conv_weights(wc, L.M, c_in, L.ks, L.m_pad, L.nchunk, packed);
    bias.resize(L.m_pad, 0.f);
    bias.resize(2 * (size_t)L.m_pad);   // second half: -bias (pair_wino_impl.h)
    for (int i = 0; i < L.m_pad; ++i) bias[L.m_pad + i] = -bias[i];
    L.wp_bytes = packed.size() * sizeof(float);
    FV_HIP_CHECK(hipMalloc((void**)&L.d_wp, L.wp_bytes));
    FV_HIP_CHECK(hipMalloc((void**)&L.d_bias, bias.size() * sizeof(float)));
    FV_HIP_CHECK(hipMemcpy(L.d_wp, packed.data(), L.wp_bytes, hipMemcpyHostToDevice));
    FV_HIP_CHECK(hipMemcpy(L.d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!transposed && stride == 1 && (k == 3 || k == 7 || k == 11) && (dil == 1 || dil == 3 || dil == 5) && padding == (k - 1) / 2 * dil &&
        c_in >= 16 && c_in == c_out) {   // the ResBlock / AMPBlock convs (not conv_pre: a launch whose size decides the path per batch)
        // Winograd F(2,3) tap groups (conv_wino_impl.h): groups at taps 0, 4, 8 -> four transformed weights each, the taps between them
        // (3, 7) -> (+w, -w); virtual-tap order = WinoGeom::off_of / acc_of
        // Which transformed copies a layer keeps (every form its shape admits was ~8 x the direct weights at k = 11): the quad-lattice
        // layers (k = 7 / 11 on whole 64-row tiles) keep the F(4,4) fragments and NOT the F(2,3) / F(4,3) ones, which only the A/B knobs FV_WINO44=0 /
        // FV_WINO4=0 reach — a process that sets those knobs BEFORE creating its layers gets the form it asks for (tests/test_gpu_conv.py,
        // tools/ab_wino.sh do); a layer whose form is missing falls back to the direct kernel.  Footprint at k = 11 per (c_out, c_in): direct 11 floats
        // + F(4,4) 10 + latency / pair fragments 16 (was 11 + 16 + 10 + 26 + 16 + 16).
        L.wino = true;
        const bool quad_layer = c_in >= 64 && c_in % 8 == 0 && c_out % 64 == 0 && k >= 7;
        const bool use_f44 = quad_layer && knobs().wino4 && knobs().wino44;
        const bool use_f43 = quad_layer && knobs().wino4 && !knobs().wino44 && have_conv_wino4();
        const int ng = (k + 1) / 4, ns = (k - 3) / 4;
        const int nv = 4 * ng + 2 * ns;
        std::vector<float> ww((size_t)c_out * c_in * nv);
        for (size_t oc = 0; oc < (size_t)c_out * c_in; ++oc) {
            const float* w = &wc[oc * k];
            float* o = &ww[oc * nv];
            for (int g = 0; g < ng; ++g) {
                const double g0 = w[4 * g], g1 = w[4 * g + 1], g2 = w[4 * g + 2];
                o[4 * g + 0] = (float)g0;
                o[4 * g + 1] = (float)((g0 + g1 + g2) * 0.5);
                o[4 * g + 2] = (float)((g0 - g1 + g2) * 0.5);
                o[4 * g + 3] = (float)g2;
            }
            for (int s2 = 0; s2 < ns; ++s2) {
                o[4 * ng + 2 * s2] = w[4 * s2 + 3];
                o[4 * ng + 2 * s2 + 1] = -w[4 * s2 + 3];
            }
        }
        L.nv = nv;
        if (c_in >= 32 && !use_f44 && !use_f43) {   // conv_wino (F(2,3) per layer) and the C = 64 / 128 k = 3 pairs
            std::vector<float> pw;
            pack_conv_weights(ww, L.M, c_in, L.nv, L.m_pad, L.nchunk, pw);
            FV_HIP_CHECK(hipMalloc((void**)&L.d_wpw, pw.size() * sizeof(float)));
            FV_HIP_CHECK(hipMemcpy(L.d_wpw, pw.data(), pw.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (use_f44) {
            // Winograd F(4,4) tap groups (conv_wino44_impl.h): groups of FOUR taps at 0, 4, 8 (taps past k are zero), points ±1/2, ±1, ±2, ∞.  Per (32-row tile,
            // plane half h) and 8-channel block: 3 ng full fragments — group g, own plane i (h = 0: +1/2, -1/2, +1; h = 1: -1, +2, -2), the four channel pairs
            // in .xyzw — then ONE fragment of the shared ∞ plane (U = the group's fourth tap): component j = group j / 2, channel pair 2 h + j % 2
            static const double G44[6][4] = {{16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45}, {-16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45}, {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                                             {2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},   {1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},    {-1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45}};
            const int ng4 = (k + 3) / 4, nv44 = 3 * ng4 + 1, mts = c_out / 32;
            auto tap = [&](int co, int ci, int j) -> double { return (j < k && ci < c_in) ? (double)wc[((size_t)co * c_in + ci) * k + j] : 0.0; };
            std::vector<float> p44(((size_t)mts * 2 * L.nchunk * nv44 + 8) * 64 * 4, 0.f);
            for (int mt = 0; mt < mts; ++mt)
                for (int hh = 0; hh < 2; ++hh)
                    for (int c = 0; c < L.nchunk; ++c)
                        for (int v = 0; v < nv44; ++v)
                            for (int l = 0; l < 64; ++l) {
                                const int co = 32 * mt + (l & 31);
                                float* dst = &p44[(((((size_t)mt * 2 + hh) * L.nchunk + c) * nv44 + v) * 64 + l) * 4];
                                for (int j = 0; j < 4; ++j) {
                                    if (v < 3 * ng4) {
                                        const int g = v / 3, pl = 3 * hh + v % 3, ci = 8 * c + 2 * j + (l >> 5);
                                        double u = 0.0;
                                        for (int t = 0; t < 4; ++t) u += G44[pl][t] * tap(co, ci, 4 * g + t);
                                        dst[j] = (float)u;
                                    } else {
                                        const int g = j / 2, ci = 8 * c + 2 * (2 * hh + j % 2) + (l >> 5);
                                        dst[j] = g < ng4 - 1 ? (float)tap(co, ci, 4 * g + 3) : 0.f;
                                    }
                                }
                            }
            FV_HIP_CHECK(hipMalloc((void**)&L.d_wpw44, p44.size() * sizeof(float)));
            FV_HIP_CHECK(hipMemcpy(L.d_wpw44, p44.data(), p44.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (use_f43) {
            // Winograd F(4,3) tap groups (conv_wino4_impl.h): per (32-row tile, plane half h) nv4 = 3 ng + 2 ns virtual taps = Wino4Geom::off_of / acc_of:
            // group g, i = 0..2 -> transformed weight U_p of taps 4g..4g+2 with p = i (h = 0: m0 m1 m2) or 5 - i (h = 1: m5 m4 m3); then per single tap
            // 4s + 3 two plain copies (h = 0: into m0 and S1, h = 1: into m5 and S2).  Packed as 2 M rows: row (2 mt + h) * 32 + r = (row 32 mt + r, half h)
            const int nv4 = 3 * ng + 2 * ns;
            std::vector<float> w4((size_t)2 * c_out * c_in * nv4);
            for (int co = 0; co < c_out; ++co)
                for (int hh = 0; hh < 2; ++hh)
                    for (int ci = 0; ci < c_in; ++ci) {
                        const float* w = &wc[((size_t)co * c_in + ci) * k];
                        float* o = &w4[((size_t)((co / 32 * 2 + hh) * 32 + co % 32) * c_in + ci) * nv4];
                        for (int g = 0; g < ng; ++g) {
                            const double g0 = w[4 * g], g1 = w[4 * g + 1], g2 = w[4 * g + 2];
                            const double U[6] = {g0 / 4, -(g0 + g1 + g2) / 6, -(g0 - g1 + g2) / 6, g0 / 24 + g1 / 12 + g2 / 6, g0 / 24 - g1 / 12 + g2 / 6, g2};
                            for (int i = 0; i < 3; ++i) o[3 * g + i] = (float)U[hh == 0 ? i : 5 - i];
                        }
                        for (int s2 = 0; s2 < ns; ++s2) o[3 * ng + 2 * s2] = o[3 * ng + 2 * s2 + 1] = w[4 * s2 + 3];
                    }
            L.nv4 = nv4;
            std::vector<float> pw;
            pack_conv_weights(w4, 2 * L.M, c_in, nv4, 2 * L.m_pad, L.nchunk, pw);
            FV_HIP_CHECK(hipMalloc((void**)&L.d_wpw4, pw.size() * sizeof(float)));
            FV_HIP_CHECK(hipMemcpy(L.d_wpw4, pw.data(), pw.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (c_in >= 32 && c_in % 32 == 0) {
            // conv_wino_lat_impl.h: fragment ((mt * nblk + blk) * nv / 2 + vp), one float4 per lane = A operands (row = lane & 15, k = lane >> 4) of
            // four 16x16x4 MFMAs: .{x,y} = virtual tap 2 vp, channel quads 0 / 1 of 8-channel block blk; .{z,w} = virtual tap 2 vp + 1
            const int nblk = c_in / 8, nf = nv / 2, mts = c_out / 16;
            std::vector<float> pl(((size_t)mts * nblk * nf + 8) * 64 * 4, 0.f);
            for (int mt = 0; mt < mts; ++mt)
                for (int blk = 0; blk < nblk; ++blk)
                    for (int vp = 0; vp < nf; ++vp)
                        for (int l = 0; l < 64; ++l)
                            for (int q = 0; q < 4; ++q) {
                                const int co = 16 * mt + (l & 15), ci = 8 * blk + 4 * (q & 1) + (l >> 4), v = 2 * vp + (q >> 1);
                                pl[((((size_t)mt * nblk + blk) * nf + vp) * 64 + l) * 4 + q] = ww[((size_t)co * c_in + ci) * nv + v];
                            }
            FV_HIP_CHECK(hipMalloc((void**)&L.d_wpwl, pl.size() * sizeof(float)));
            FV_HIP_CHECK(hipMemcpy(L.d_wpwl, pl.data(), pl.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if ((c_in == 16 || c_in == 32 || c_in % 32 == 0) && k >= 7) {
            // pair_wino44_impl.h (C = 16 / 32), conv_wino_lat44_impl.h (whole 32-row blocks): F(4,4) tap groups (the G44 transform of conv_wino44 above) as A fragments of v_mfma_f32_16x16x4_f32 (row = lane & 15, k = lane >> 4).
            // Virtual tap v of a channel: group v / 7, plane v % 7 (+1/2 -1/2 +1 -1 +2 -2 inf; the last group has no inf tap).  Fragment (m-tile mt, chunk c of 8
            // channels, f): .{x,y} = tap 2 f, k-steps 0 / 1 (channels 8 c + {0..3}, {4..7}); .{z,w} = tap 2 f + 1; + 8 zero fragments of prefetch overrun
            static const double G44[6][4] = {{16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45}, {-16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45}, {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                                             {2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},   {1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},    {-1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45}};
            const int ng4 = (k + 3) / 4, nvq = 7 * (ng4 - 1) + 6, nfq = (nvq + 1) / 2, nchk = c_in / 8, mts = c_in / 16;
            auto tap = [&](int co, int ci, int j) -> double { return j < k ? (double)wc[((size_t)co * c_in + ci) * k + j] : 0.0; };
            auto utap = [&](int co, int ci, int v) -> float {
                if (v >= nvq) return 0.f;
                const int g = v / 7, pl = v % 7;
                if (pl == 6) return (float)tap(co, ci, 4 * g + 3);
                double u = 0.0;
                for (int t = 0; t < 4; ++t) u += G44[pl][t] * tap(co, ci, 4 * g + t);
                return (float)u;
            };
            std::vector<float> pq(((size_t)mts * nchk * nfq + 8) * 64 * 4, 0.f);
            for (int mt = 0; mt < mts; ++mt)
                for (int c = 0; c < nchk; ++c)
                    for (int f = 0; f < nfq; ++f)
                        for (int l = 0; l < 64; ++l)
                            for (int q = 0; q < 4; ++q) {
                                const int co = 16 * mt + (l & 15), ci = 8 * c + 4 * (q & 1) + (l >> 4), v = 2 * f + (q >> 1);
                                pq[((((size_t)mt * nchk + c) * nfq + f) * 64 + l) * 4 + q] = utap(co, ci, v);
                            }
            FV_HIP_CHECK(hipMalloc((void**)&L.d_wpq16, pq.size() * sizeof(float)));
            FV_HIP_CHECK(hipMemcpy(L.d_wpq16, pq.data(), pq.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (c_in == 16 || c_in == 32) {
            // pair_wino_impl.h: A fragments of v_mfma_f32_16x16x4_f32 (row = lane & 15, k = lane >> 4), one float4 per lane = four MFMAs.
            //   C = 16: fragment v            .{x,y,z,w}[s]       = W'[lane & 15][4 s + (lane >> 4)][v]                      (16 channels)
            //   C = 32: fragment sb * nv + v  .{x,y,z,w}[2 s + mt] = W'[16 mt + (lane & 15)][8 sb + 4 s + (lane >> 4)][v]   (8 channels, both m-tiles)
            // + 8 zero fragments: the kernels' weight ring runs a few fragments past the end
            const int mt_n = c_in / 16, fch = 16 / mt_n, nfr = c_in / fch * nv;
            std::vector<float> p16((size_t)(nfr + 8) * 64 * 4, 0.f);
            for (int sb = 0; sb < c_in / fch; ++sb)
                for (int v = 0; v < nv; ++v)
                    for (int l = 0; l < 64; ++l)
                        for (int q = 0; q < 4; ++q) {
                            const int s = mt_n == 2 ? q / 2 : q, mt = mt_n == 2 ? q % 2 : 0;
                            const int co = 16 * mt + (l & 15), ci = fch * sb + 4 * s + (l >> 4);
                            p16[(((size_t)sb * nv + v) * 64 + l) * 4 + q] = ww[((size_t)co * c_in + ci) * nv + v];
                        }
            FV_HIP_CHECK(hipMalloc((void**)&L.d_wpw16, p16.size() * sizeof(float)));
            FV_HIP_CHECK(hipMemcpy(L.d_wpw16, p16.data(), p16.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (with_f16x3 && f16x3_eligible(transposed, c_in, L.M, L.ks, L.dil)) {
        L.nch16 = (c_in + 15) / 16;
        if (L.ks <= 4) L.nch16 = (L.nch16 + 3) / 4 * 4;   // whole LDS chunks of two / four sub-chunks
        std::vector<_Float16> ph;
        L.w_scale = pack_conv_weights_f16x3(wc, L.M, c_in, L.ks, L.m_pad, L.nch16, ph);
        FV_HIP_CHECK(hipMalloc(&L.d_wph, ph.size() * sizeof(_Float16)));
        FV_HIP_CHECK(hipMemcpy(L.d_wph, ph.data(), ph.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    }
    if (with_f16x3 && !transposed && c_in == 16 && c_out == 16 && (k == 3 || k == 7 || k == 11)) {
        // pair16_f16x3.hip: row (s, co) of tap block jj in [0, k] holds w[co][:, jj - s] (two output samples per channel
        // share one activation fragment); 32x32x16 A-fragment order, (wh, wl) planes, + 4 zero blocks of prefetch overrun
        float wmax = 0.f;
        for (size_t i = 0; i < (size_t)16 * 16 * k; ++i) wmax = std::max(wmax, std::fabs(host_w[i]));
        int e = 0;
        if (wmax > 0.f && std::isfinite(wmax)) {
            (void)std::frexp(wmax, &e);
            e = 14 - e;   // wmax * 2^e in [2^13, 2^14), as in pack_conv_weights_f16x3
        }
        L.w_scale = std::ldexp(1.0f, e);
        std::vector<_Float16> ph((size_t)(k + 1 + 4) * 2 * 64 * 8, (_Float16)0.f);
        for (int jj = 0; jj <= k; ++jj)
            for (int l = 0; l < 64; ++l) {
                const int s = (l & 31) >> 4, co = l & 15, j = jj - s;
                if (j < 0 || j >= k) continue;
                for (int i = 0; i < 8; ++i) {
                    const int ci = 8 * (l >> 5) + i;
                    const float w = host_w[((size_t)co * 16 + ci) * k + j] * L.w_scale;   // exact: power-of-two scale
                    const _Float16 wh = (_Float16)w;
                    ph[(((size_t)jj * 2 + 0) * 64 + l) * 8 + i] = wh;
                    ph[(((size_t)jj * 2 + 1) * 64 + l) * 8 + i] = (_Float16)(w - (float)wh);
                }
            }
        FV_HIP_CHECK(hipMalloc(&L.d_wph16, ph.size() * sizeof(_Float16)));
        FV_HIP_CHECK(hipMemcpy(L.d_wph16, ph.data(), ph.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    }
    if (!transposed && c_in == 16 && c_out == 16) {
        // v_mfma_f32_16x16x4_f32 A fragments: [tap][lane] float4, .q = W[lane & 15][4q + (lane >> 4)][tap]
        std::vector<float> p16((size_t)k * 64 * 4);
        for (int j = 0; j < k; ++j)
            for (int l = 0; l < 64; ++l)
                for (int q = 0; q < 4; ++q)
                    p16[((size_t)j * 64 + l) * 4 + q] = host_w[((size_t)(l & 15) * 16 + 4 * q + (l >> 4)) * k + j];
        FV_HIP_CHECK(hipMalloc((void**)&L.d_wp16, p16.size() * sizeof(float)));
        FV_HIP_CHECK(hipMemcpy(L.d_wp16, p16.data(), p16.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return FV_OK;
}

void conv_layer_destroy(ConvLayer& L) {
    if (L.d_wp) (void)hipFree(L.d_wp);
    if (L.d_bias) (void)hipFree(L.d_bias);
    if (L.d_wp16) (void)hipFree(L.d_wp16);
    if (L.d_wpw) (void)hipFree(L.d_wpw);
    L.d_wpw = nullptr;
    if (L.d_wpw16) (void)hipFree(L.d_wpw16);
    L.d_wpw16 = nullptr;
    if (L.d_wpq16) (void)hipFree(L.d_wpq16);
    L.d_wpq16 = nullptr;
    if (L.d_wpw44) (void)hipFree(L.d_wpw44);
    L.d_wpw44 = nullptr;
    if (L.d_wpw4) (void)hipFree(L.d_wpw4);
    L.d_wpw4 = nullptr;
    if (L.d_wpwl) (void)hipFree(L.d_wpwl);
    L.d_wpwl = nullptr;
    if (L.d_wph) (void)hipFree(L.d_wph);
    if (L.d_wph16) (void)hipFree(L.d_wph16);
    L.d_wph16 = nullptr;
    L.d_wph = nullptr;
    L.d_wp16 = nullptr;
    L.d_wp = nullptr;
    L.d_bias = nullptr;
}

#if defined(FV_X_SPLITK_TS) || defined(FV_X_CONV_TS)
static long long* g_sk_ts = nullptr;
extern "C" __attribute__((visibility("default"))) void fv_debug_set_splitk_timestamps(void* device_buffer) { g_sk_ts = (long long*)device_buffer; }
#endif
// fv_set_conv_algorithm of the running call, with the process-wide FV_WINO knob as the default behind FV_CONV_ALGO_AUTO
static int effective_algo() {
    const int a = cur_algo();
    if (a != FV_CONV_ALGO_AUTO) return a;
    const int w = knobs().wino;
    return w == 0 ? FV_CONV_ALGO_DIRECT : w >= 2 ? FV_CONV_ALGO_WINOGRAD : FV_CONV_ALGO_AUTO;
}

static int wino_lat_tile(const ConvLayer& L, long long np, int batch, int layers);

static int choose_tile(int M, long long N, int batch) {
    int big, small;
    if (M <= 32) {
        big = TILE_32x512;
        small = TILE_32x128;
    } else if (M <= 64) {
        // 64 x 128 tiles also for launches that would fill the chip with 64 x 256 ones (round 3: HiFiGAN step -0.35 %, BigVGAN -0.15 %
        // over five / three interleaved rounds: twice the workgroups, shorter lives, less lock-step)
        big = TILE_64x128;
        small = TILE_64x128;
    } else {
        big = TILE_128x128;
        small = TILE_128x64;
    }
    int mb, nb_big, nb_small;
    tile_dims(big, &mb, &nb_big);
    tile_dims(small, &mb, &nb_small);
    const long long m_blks = (M + mb - 1) / mb;
    const long long tiles_big = (N + nb_big - 1) / nb_big, tiles_small = (N + nb_small - 1) / nb_small;
    const double cols_big = (double)tiles_big * nb_big, cols_small = (double)tiles_small * nb_small * 1.04;
    const long long blocks_big = tiles_big * m_blks * batch;
    // not enough workgroups to fill 256 CUs twice over, or a lot of padded columns -> smaller tile
    // (round 3: < 1024 instead of < 512 workgroups — BigVGAN's C = 256 stage is exactly one round of 768 full tiles, every workgroup in
    //  lock-step: half-width tiles took its B = 64 step from 36.33 to 36.10 ms; HiFiGAN has no launch in that range)
    if (blocks_big < 1024 || cols_small < cols_big) {
        // still fewer workgroups than CUs: 32 x 64 tiles with K split across the four waves (latency variant)
        const long long blocks_small = tiles_small * m_blks * batch;
        static const long long sk_max = std::getenv("FV_SPLITK_MAX") ? std::atoll(std::getenv("FV_SPLITK_MAX")) : 200;   // experiments
        if (blocks_small < sk_max && !cur_invariant()) {   // (batch-invariant mode: the split-K kernels sum K in another order)
            // 32 x 64 tiles: 4x the workgroups of 128 x 64 / 2x those of 32 x 128.  Workgroups are dealt out one per CU and
            // round, and a workgroup's time is its MFMA chain (proportional to the tile width): the busiest CU decides, so
            // take 32 x 32 tiles when their rounds are shorter in total (344 tiles of 32 x 64 = 2 rounds of 2 units against
            // 688 of 32 x 32 = 3 rounds of 1; the narrow tile re-stages the halo and reuses each weight fragment half as often)
            const long long cus = num_cus();
            const long long blocks_64 = ((M + 31) / 32) * ((N + 63) / 64) * batch, blocks_32 = ((M + 31) / 32) * ((N + 31) / 32) * batch;
            const double cost_64 = (double)((blocks_64 + cus - 1) / cus) * 2.0, cost_32 = (double)((blocks_32 + cus - 1) / cus) * 1.1;
            static const char* const force = std::getenv("FV_SPLITK");   // experiments: "32" / "64"
            if (force) return force[0] == '3' ? TILE_SPLITK_32x32 : TILE_SPLITK_32x64;
            return cost_32 < cost_64 ? TILE_SPLITK_32x32 : TILE_SPLITK_32x64;
        }
        return small;
    }
    // Enough workgroups either way: weigh the last, partly filled round of each tiling (workgroups are dealt out in
    // rounds of one per CU; e.g. 516 tiles of 128 x 128 cost three rounds for two rounds' worth of work, 1032 tiles of
    // 128 x 64 cost five half-size rounds).  A half-width tile runs at ~0.9 of the full tile's rate; keep the full tile
    // unless the model sees a clear win.
    {   // (measured on HiFiGAN-V1: B = 12 ... 14 -2 ... -3.5 %, every other batch size from 1 to 64 unchanged)
        const long long cus = num_cus();
        const long long blocks_small = tiles_small * m_blks * batch;
        const double cost_big = (double)((blocks_big + cus - 1) / cus) * nb_big;
        const double cost_small = (double)((blocks_small + cus - 1) / cus) * nb_small / 0.90;
        if (cost_small < 0.97 * cost_big) return small;
    }
    return big;
}

// Configuration of the LDS-free pointwise GEMM (gemm_pw.hip) for this call, or -1 to stay on the general conv kernel.
// The persistent kernel runs CUs x 4 x W waves that share the tile list equally, so what matters is how evenly
// tiles / SIMD divides: busy time of the fullest SIMD vs the average.
static const char* const kGemmPwNames[GEMM_PW_COUNT] = {"64x64 w2", "32x64 w3"};
static int choose_gemm_pw(const ConvLayer& L, const ConvRun& r, long long tout) {
    // 32-bit byte offsets over all batch items
    const long long bytes = 4LL * r.batch * std::max<long long>((long long)L.c_in * r.t_in, (long long)L.c_out * tout);
    if (L.c_in < 64) return -1;   // (>= 8 chunks: the operand ring is primed unconditionally)
    // The two launch-dependent exits.  gemm_pw and the k = 1 conv kernel do NOT form the same sums (measured round 5: equal bits at K = 512, not at
    // K = 2048), so in batch-invariant mode they are refused (-2) instead of silently changing a clip's last bits with the batch it is part of.
    if (bytes >= 0xFFFFFF00LL) return cur_invariant() ? -2 : -1;
    if (num_cus() < 8) return -1;   // a partition below one CU per XCD (or a failed query): the persistent grid would be empty (fixed per device, not per call)
    if (knobs().pw != -2) return knobs().pw;   // experiments (FV_PW): force a configuration, or "old" / -1 for the conv kernel
    const long long simds = (long long)(num_cus() / 8 * 8) * 4;
    const long long n64 = ((long long)tout * r.batch + 63) / 64;
    struct Cand { int cfg, mt, w; double pref; };
    // pref: measured rate of the configuration on a well-balanced large GEMM, relative to the best one (profiles/README.md)
    static const Cand cands[] = {{GEMM_PW_64x64_W2, 2, 2, 1.00}, {GEMM_PW_32x64_W3, 1, 3, 0.94}};
    int best = -1;
    double best_score = 0.0;
    for (const Cand& c : cands) {
        const long long tiles = ((L.M + 32 * c.mt - 1) / (32 * c.mt)) * n64;
        const long long waves = simds * c.w;
        // fullest SIMD: whole rounds give each of its W waves one tile, the last partial round fills first slots first
        const long long left = tiles % waves;
        const long long simd_max = c.w * (tiles / waves) + std::min<long long>(c.w, (left + simds - 1) / simds);
        const double eff = (double)tiles / (double)simds / (double)simd_max;
        const double score = eff * c.pref;
        if (score > best_score) {
            best_score = score;
            best = c.cfg;
        }
    }
    // a sixth of the SIMDs' time used or less (single clips): too few whole tiles — the conv kernel's split-K tiles do better
    // (measured crossover, tools/probe_pointwise.py: B = 32 x 86 frames, 512 -> 128 at 0.17 still wins by a third)
    return (best_score >= 0.15 || cur_invariant()) ? best : -1;   // (batch-invariant mode: by shape alone)
}

// Row groups of the XCD partition (gemm_pw.hip).  An XCD reads 1 / PX of the weights and PX / 8 of the activation columns:
// four row groups when the weights are the operand that does not fit (activations under ten times their size: the C -> 4C
// layers, +2 ... 3 %), one (every XCD owns a column range and streams all weights) when the activations dominate (4C -> C:
// four row groups measured -5 %).  tools/archive/px_probe.sh.
static int choose_gemm_pw_xcd_rows(const ConvLayer& L, const ConvRun& r, long long tout, int cfg) {
    const int mt = cfg == GEMM_PW_32x64_W3 ? 1 : 2;
    const int mtiles = (L.M + 32 * mt - 1) / (32 * mt);
    const double a_bytes = 4.0 * L.c_in * L.c_out, b_bytes = 4.0 * L.c_in * (double)tout * r.batch;
    int px = b_bytes < 10.0 * a_bytes ? 4 : 1;
    if (knobs().pw_px) px = knobs().pw_px;   // experiments (FV_PW_PX)
    if (px != 1 && px != 2 && px != 4 && px != 8) px = 1;
    while (px > 1 && mtiles % px != 0) px /= 2;
    return px;
}

static const char* const kTileNames[TILE_COUNT] = {"128x128", "64x256", "32x512", "128x64", "32x128", "64x128", "splitK32x64", "splitK32x32", "256x64", "256x32", "128x96"};

static const char* const kSplitNames[SPLIT_COUNT] = {"128x128", "64x256", "32x256"};

// f16x3 precision mode: tile choice + dispatch of the split-fp16 kernel (p already describes the layer call)
static fv_status conv_layer_run_f16x3(const ConvLayer& L, const ConvRun& r, ConvParams& p, hipStream_t stream) {
    if (p.x2 || p.x3) {   // these kernels stage one input tensor: a three-operand mean must have been formed by the caller
        set_error("conv_layer_run: the f16x3 kernels take no three-operand input");
        return FV_ERR_INVALID;
    }
    p.wph = L.d_wph;
    p.nch16 = L.nch16;
    p.nch16_real = (L.c_in + 15) / 16;
    p.acc_scale = 1.0f / L.w_scale;
    // wave tile 32 x 128 either way (64 accumulator registers); waves stacked along M when there are >= 128 rows
    const int cfg = L.M <= 32 ? SPLIT_32x256 : L.M <= 64 ? SPLIT_64x256 : SPLIT_128x128;
    const int mb = cfg == SPLIT_32x256 ? 32 : cfg == SPLIT_64x256 ? 64 : 128, nb = cfg == SPLIT_128x128 ? 128 : 256;
    // pointwise convs have no halo: batch and time flatten into one column axis (no per-item partial tiles)
    int launch_batch = r.batch;
    if (!L.transposed && L.ks == 1 && L.pad_l == 0 && r.batch > 1 &&
        (long long)r.batch * std::max<long long>((long long)L.c_in * r.t_in, (long long)L.c_out * p.N) < (1LL << 30)) {
        p.flat = 1;
        p.n_total = p.N * r.batch;
        launch_batch = 1;
    }
    p.m_blks = (L.M + mb - 1) / mb;
    p.n_tiles = ((p.flat ? p.n_total : p.N) + nb - 1) / nb;
    const int prof_idx = prof_begin(stream);
    bool ok = false;
    switch (L.ks) {
        case 1: ok = launch_conv_f16x3_k1(p, cfg, launch_batch, stream); break;
        case 2:
        case 4: ok = launch_conv_f16x3_misc(p, cfg, r.batch, stream); break;
        case 3: ok = launch_conv_f16x3_k3(p, cfg, r.batch, stream); break;
        case 7: ok = launch_conv_f16x3_k7(p, cfg, r.batch, stream); break;
        case 11: ok = launch_conv_f16x3_k11(p, cfg, r.batch, stream); break;
        default: break;
    }
    if (!ok) {
        set_error("conv_layer_run: no f16x3 kernel for (k=%d, dilation=%d)", L.ks, L.dil);
        return FV_ERR_UNSUPPORTED;
    }
    static thread_local char name[96];
    std::snprintf(name, sizeof(name), "conv_f16x3<k=%d d=%d tile=%s>", L.ks, L.dil, kSplitNames[cfg]);
    set_last_kernel(name);
    if (prof_idx >= 0) {
        const long long tout = L.out_len(r.t_in);
        const double macs = (double)L.c_in * L.c_out * L.k * (L.transposed ? (double)r.t_in : (double)tout) * r.batch;
        double elems = (double)L.c_in * r.t_in + (double)L.c_out * tout;
        if (r.res) elems += (double)L.c_out * tout;
        if (r.out_mode == OUT_ACCUM) elems += (double)L.c_out * tout;
        char lbl[160];
        std::snprintf(lbl, sizeof(lbl), "%s cin=%d cout=%d%s grid=%d", name, L.c_in, L.c_out, L.transposed ? " convT" : "",
                      launch_batch * p.m_blks * p.n_tiles);
        prof_end(stream, prof_idx, lbl, 2.0 * macs, elems * r.batch * 4.0 + (double)L.c_in * L.c_out * L.k * 4.0);
    }
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
}

// What every per-layer conv launch records: the kernel's name for fv_last_kernel, and — under fv_profile_begin — one profiler row with the layer's
// ALGORITHMIC work whichever sums the kernel forms (direct-sum MACs; per-layer compulsory bytes: input once, output once, residual / accumulate
// operand once, the folded weights once — DESIGN.md §5).  One copy, so that a change to the accounting reaches every kernel family.
static fv_status finish_conv_launch(const ConvLayer& L, const ConvRun& r, long long tout, hipStream_t stream, int prof_idx, const char* name,
                                    long long grid, const char* tags = "", bool three_inputs = false) {
    if (tags[0]) {
        char full[160];
        std::snprintf(full, sizeof(full), "%s%s", name, tags);
        set_last_kernel(full);
    } else {
        set_last_kernel(name);
    }
    if (prof_idx >= 0) {
        const double macs = (double)L.c_in * L.c_out * L.k * (L.transposed ? (double)r.t_in : (double)tout) * r.batch;
        double elems = (double)L.c_in * r.t_in + (double)L.c_out * tout;
        if (r.res) elems += (double)L.c_out * tout;
        if (r.out_mode == OUT_ACCUM) elems += (double)L.c_out * tout;
        if (three_inputs) elems += 2.0 * L.c_in * r.t_in;   // the other two branch outputs
        char lbl[192];
        std::snprintf(lbl, sizeof(lbl), "%s cin=%d cout=%d%s grid=%lld", name, L.c_in, L.c_out, tags, grid);
        prof_end(stream, prof_idx, lbl, 2.0 * macs, elems * r.batch * 4.0 + (double)L.c_in * L.c_out * L.k * 4.0);
    }
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
}

fv_status conv_layer_run(const ConvLayer& L, const ConvRun& r, hipStream_t stream) {
    if (!L.d_wp) {
        set_error("conv_layer_run: layer not initialised");
        return FV_ERR_STATE;
    }
    if (r.batch <= 0 || r.t_in <= 0) {
        set_error("conv_layer_run: empty input (batch=%d, t_in=%d)", r.batch, r.t_in);
        return FV_ERR_INVALID;
    }
    const long long tout = L.out_len(r.t_in);
    if (tout <= 0) {
        set_error("conv_layer_run: non-positive output length %lld", tout);
        return FV_ERR_INVALID;
    }
    if ((long long)L.c_out * tout >= (1LL << 30) || (long long)L.c_in * r.t_in >= (1LL << 30)) {
        set_error("conv_layer_run: a batch item of %lld elements exceeds the 4 GiB buffer-addressing span",
                  (long long)std::max<long long>((long long)L.c_out * tout, (long long)L.c_in * r.t_in));
        return FV_ERR_UNSUPPORTED;
    }
    // the split-fp16 kernels implement the pre-activations of the MFMA-bound layers only (none / SiLU); ONE decision, used by the
    // three-operand routing below and by the dispatch (those kernels and gemm_pw ignore x2 / x3)
    const bool f16 = L.precision == FV_PRECISION_F16X3 && L.d_wph && f16x3_per_layer_ok(L) &&
                     (r.pre_act == FV_ACT_NONE || r.pre_act == FV_ACT_SILU);
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = r.x;
    if (r.x2 || r.x3) {
        // input = ((x + x2) + x3) / 3: formed by the staging of the exact-fp32 polyphase transposed-conv kernels (SUM3 variants,
        // conv_mfma_impl.h); everything else gets it from a pass of mean_of_three_kernel into the caller's scratch tensor
        if (!r.x2 || !r.x3) {
            set_error("conv_layer_run: x2 and x3 go together");
            return FV_ERR_INVALID;
        }
        static const bool no_fuse = std::getenv("FV_NO_SUM3") != nullptr;   // experiments
        if (L.transposed && conv_sum3_supported(L.ks) && L.dil == 1 && !f16 && !no_fuse) {
            p.x2 = r.x2;
            p.x3 = r.x3;
        } else {
            if (!r.sum_tmp) {
                set_error("conv_layer_run: this layer needs a scratch tensor (sum_tmp) for the three-operand input");
                return FV_ERR_INVALID;
            }
            const fv_status ms = launch_mean_of_three(r.x, r.x2, r.x3, r.sum_tmp, (long long)r.batch * L.c_in * r.t_in, stream);
            if (ms) return ms;
            p.x = r.sum_tmp;
        }
    }
    p.wp = L.d_wp;
    p.bias = L.d_bias;
    p.y = r.y;
    p.res = r.res;
    p.gamma = r.gamma;
    p.Cin = L.c_in;
    p.Tin = r.t_in;
    p.M = L.M;
    p.N = (int)L.gemm_cols(r.t_in);
    p.nchunk = L.nchunk;
    p.nchunk_real = L.nchunk_real;
    p.pad_l = L.pad_l;
    p.ks = L.ks;
    p.dil = L.dil;
    p.pre_act = r.pre_act;
    p.post_act = r.post_act;
    p.slope = r.slope;
    p.out_mode = r.out_mode;
    p.out_scale = r.out_scale;
    p.convt = L.transposed ? 1 : 0;
    p.u = L.stride;
    p.pad_t = L.padding;
    p.Tout = (int)tout;
    p.Cout = L.c_out;
    p.x_bstride = (long long)L.c_in * r.t_in;
    p.y_bstride = (long long)L.c_out * tout;
    p.acc_scale = 1.0f;

    // stride-8 upsamplers, bias only, aligned: 16-byte output quads (conv_mfma_impl.h: conv_epilogue)
    p.vec_store = (L.transposed && L.stride == 8 && !r.res && !r.gamma && r.out_mode == OUT_SET && r.post_act == FV_ACT_NONE &&
                   tout % 4 == 0 && L.padding % 4 == 0 && ((uintptr_t)r.y & 15) == 0 && knobs().vec_store) ? 1 : 0;

    if (f16) return conv_layer_run_f16x3(L, r, p, stream);

    // pointwise convs of the MFMA-bound kind (ConvNeXt's Linear layers): the LDS-free GEMM kernel (gemm_pw.hip)
    if (!L.transposed && L.ks == 1 && L.pad_l == 0 && L.c_in % 8 == 0 && r.pre_act == FV_ACT_NONE && r.out_mode == OUT_SET) {
        const int variant = p.x2 ? -1 : choose_gemm_pw(L, r, tout);   // (gemm_pw stages one input tensor)
        if (variant == -2) {
            set_error("conv_layer_run: batch-invariant mode: a pointwise layer over %d items of %lld elements is past the 32-bit offset span of the "
                      "pointwise GEMM, and the general conv kernel forms other sums — split the batch", r.batch,
                      (long long)std::max<long long>((long long)L.c_in * r.t_in, (long long)L.c_out * tout));
            return FV_ERR_UNSUPPORTED;
        }
        if (variant >= 0) {
            p.flat = 1;
            p.n_total = p.N * r.batch;
            const bool pair = p.N % 2 == 0 && (((uintptr_t)r.x | (uintptr_t)r.y | (uintptr_t)r.res) & 7) == 0;
            p.xcd_rows = choose_gemm_pw_xcd_rows(L, r, tout, variant);
            const int prof_idx = prof_begin(stream);
            const int grid = launch_gemm_pw(p, variant, pair, stream);
            static thread_local char name[96];
            std::snprintf(name, sizeof(name), "gemm_pw<%s%s>", kGemmPwNames[variant], pair ? " pair" : "");
            set_last_kernel(name);
            if (prof_idx >= 0) {
                double elems = (double)L.c_in * r.t_in + (double)L.c_out * tout;
                if (r.res) elems += (double)L.c_out * tout;
                char lbl[160];
                std::snprintf(lbl, sizeof(lbl), "%s cin=%d cout=%d grid=%d", name, L.c_in, L.c_out, grid);
                prof_end(stream, prof_idx, lbl, 2.0 * L.c_in * L.c_out * (double)tout * r.batch,
                         elems * r.batch * 4.0 + (double)L.c_in * L.c_out * 4.0);
            }
            FV_HIP_CHECK(hipGetLastError());
            return FV_OK;
        }
    }

#if defined(FV_X_SPLITK_TS) || defined(FV_X_CONV_TS)
    p.dbg_ts = g_sk_ts;
#endif
    // Winograd tap groups for the dilated ResBlock / AMPBlock convs (layers conv_layer_create marked `wino`): per launch
    //   launches of >= one workgroup per CU (or FV_CONV_ALGO_WINOGRAD, or batch-invariant mode: one algorithm per layer whatever the batch)
    //       k = 7 / 11 on whole 64-row tiles   conv_wino44 — F(4,4) on the quad lattice, 20 / 13 products per four outputs (conv_wino44_impl.h);
    //                                          FV_WINO44=0: conv_wino4, its F(4,3) predecessor (26 / 16; conv_wino4_impl.h)
    //       everything else (k = 3, C = 32, row counts without whole 64-row tiles, FV_WINO4=0)   conv_wino — F(2,3) on the pair lattice (conv_wino_impl.h)
    //   launches below that gate, FV_CONV_ALGO_AUTO only   conv_wino_lat — F(2,3), 16-row tiles, K split over the waves (conv_wino_lat_impl.h)
    // The gate counts the launch's workgroups in the F(2,3) tiling WHICHEVER kernel is launched (a 128-row F(4,4) workgroup covers two of those blocks at
    // C >= 128, one at C = 64): it was swept in these units (tools/sweep_wino_batch.py, GATES=32,...,512; LOG R4.20 — round 3: half a workgroup per CU; with
    // the 128-row workgroups a launch of 128 - 255 blocks leaves CUs empty: gate 256 against 128: B = 6 2.43 / 2.76 ms, B = 8 3.10 / 3.23, B = 2 1.25 / 1.28,
    // equal elsewhere; 512 loses at B = 3 - 4 and 12 - 16).  A single clip (86 blocks at C = 128) stays on the latency kernel either way.
    const int algo = effective_algo();
    if (algo != FV_CONV_ALGO_DIRECT && L.wino && !p.x2 && L.M >= knobs().wino_min_m) {
        static const int wdims[WINO_COUNT][2] = {{128, 32}, {64, 64}, {32, 128}, {128, 64}, {64, 128}};
        // 64 accumulator registers per wave (32 output pairs x 4 planes): three waves per SIMD; the 128 x 64-pair tile (two waves) measured
        // 8 % slower on the headline's C = 128 stage although it fetches each weight fragment half as often
        int wcfg = L.M > 64 ? WINO_128x32 : L.M > 32 ? WINO_64x64 : WINO_32x128;
        const long long np = (long long)L.dil * ((tout + 2 * L.dil - 1) / (2 * L.dil));   // pair columns: whole blocks of 2 D samples
        const long long nq = (long long)L.dil * ((tout + 4 * L.dil - 1) / (4 * L.dil));   // quad columns: whole blocks of 4 D samples
#ifdef FV_X_WINO_NT2
        if (knobs().wino_cfg >= 0 && knobs().wino_cfg < WINO_COUNT) wcfg = knobs().wino_cfg;
#else
        if (knobs().wino_cfg >= 0 && knobs().wino_cfg <= WINO_32x128) wcfg = knobs().wino_cfg;
#endif
        const int mb = wdims[wcfg][0], pairs = wdims[wcfg][1];
        const long long blocks = (long long)r.batch * ((L.M + mb - 1) / mb) * ((np + pairs - 1) / pairs);
        const long long min_blocks = knobs().wino_min_blocks >= 0 ? knobs().wino_min_blocks : num_cus();
        char name[96];
        if (blocks >= min_blocks || algo == FV_CONV_ALGO_WINOGRAD || cur_invariant()) {
            const int form = (knobs().wino4 && knobs().wino44 && L.d_wpw44) ? 44 : (knobs().wino4 && L.d_wpw4) ? 43 : L.d_wpw ? 23 : 0;
            // (form 0: a layer whose Winograd form for the current knobs was not packed at create — the direct kernel below)
            if (form) {
                const int prof_idx = prof_begin(stream);
                bool launched = false;
                if (form == 44) {
                    // two 32-row tiles per wave (128-row workgroups, two waves per SIMD) wherever the layer has whole 128-row blocks: the staging of a window then
                    // feeds twice the products.  Back to back the C = 256 launches are slower that way (384 workgroups on 256 CUs), inside the three-stream step
                    // they are not: 10.95 (64 rows) / 10.77 (128 rows for launches of >= 4 workgroups per CU only) / 10.65 ms (always) interleaved (LOG R4.16)
                    const int rows44 = (L.M % 128 == 0 && knobs().wino44_rows != 64) ? 128 : 64;
                    p.wp = L.d_wpw44;
                    p.m_blks = L.M / rows44;
                    p.n_tiles = (int)((nq + 31) / 32);
                    // Flattened column axis (round 5): a clip whose quad columns are not a whole number of 32-column tiles pads every clip's last tile (T = 688:
                    // 172 columns = 5.4 tiles, 10 % of the launch's products on padding).  The tiles may instead run over ONE axis of all clips' columns, each clip's
                    // nq columns followed by NG D columns of its own halo (the last outputs' tap groups reach n + NG D: with that gap no window ever holds another
                    // clip's columns; the gap columns' outputs lie past T and are not stored).  Every output is the same sum in the same order — tile membership
                    // does not enter it: bit-identical, batch-invariant.  Taken when it needs fewer tiles, for the lean operand set, whole chunks, and tensors whose
                    // clips all sit inside one 4 GiB descriptor.
                    int launch_batch44 = r.batch;
                    // (instances exist for 128-row workgroups with the activation known at compile time: D = 1 with the row-split epilogue, D = 3 / 5 behind a SiLU —
                    //  what the headline's C = 256 stage launches; the launcher declines anything else and the per-clip tiling below takes over)
                    const int n_tiles_clip = p.n_tiles;
                    if (knobs().wino44_flat && r.batch > 1 && rows44 == 128 && !r.gamma && r.out_mode == OUT_SET && p.acc_scale == 1.0f && L.c_in % 32 == 0 &&
                        (r.pre_act == FV_ACT_SILU || (r.pre_act == FV_ACT_NONE && L.dil == 1))) {
                        const long long gap = (long long)((L.ks + 3) / 4) * L.dil, S = nq + gap;
                        const long long tiles_flat = (r.batch * S - gap + 31) / 32;
                        const long long span_x = (long long)r.batch * p.x_bstride * 4, span_y = (long long)r.batch * p.y_bstride * 4;
                        if (tiles_flat < (long long)r.batch * p.n_tiles && span_x < (1LL << 32) && span_y < (1LL << 32)) {
                            p.col_S = (int)S;
                            p.col_batch = r.batch;
                            p.n_tiles = (int)tiles_flat;
                            launch_batch44 = 1;
                        }
                    }
                    launched = L.ks == 7 ? launch_conv_wino44_k7(p, rows44, launch_batch44, stream) : launch_conv_wino44_k11(p, rows44, launch_batch44, stream);
                    if (!launched && p.col_S) {   // (no flattened instance for this operand set — unaligned rows at D = 1: the row-split epilogue does not apply)
                        p.col_S = p.col_batch = 0;
                        p.n_tiles = n_tiles_clip;
                        launch_batch44 = r.batch;
                        launched = L.ks == 7 ? launch_conv_wino44_k7(p, rows44, launch_batch44, stream) : launch_conv_wino44_k11(p, rows44, launch_batch44, stream);
                    }
                    std::snprintf(name, sizeof(name), "conv_wino44<k=%d d=%d tile=%dx32q>", L.ks, L.dil, rows44);
                    if (launched) return finish_conv_launch(L, r, tout, stream, prof_idx, name, (long long)launch_batch44 * p.m_blks * p.n_tiles, p.col_S ? " flat" : "");
                } else if (form == 43) {   // (k = 7 / 11 only — k = 3: 6 products per quad against 8, and F(2,3) measured faster: LOG R4.14)
                    p.wp = L.d_wpw4;
                    p.m_blks = L.M / 64;
                    p.n_tiles = (int)((nq + 31) / 32);
                    launched = L.ks == 7 ? launch_conv_wino4_k7(p, r.batch, stream) : launch_conv_wino4_k11(p, r.batch, stream);
                    std::snprintf(name, sizeof(name), "conv_wino4<k=%d d=%d tile=64x32q>", L.ks, L.dil);
                } else {
                    p.wp = L.d_wpw;
                    p.m_blks = (L.M + mb - 1) / mb;
                    p.n_tiles = (int)((np + pairs - 1) / pairs);
                    launched = L.ks == 3 ? launch_conv_wino_k3(p, wcfg, r.batch, stream)
                               : L.ks == 7 ? launch_conv_wino_k7(p, wcfg, r.batch, stream) : launch_conv_wino_k11(p, wcfg, r.batch, stream);
                    std::snprintf(name, sizeof(name), "conv_wino<k=%d d=%d tile=%dx%dp>", L.ks, L.dil, mb, pairs);
                }
                if (!launched) {
                    set_error("conv_layer_run: no Winograd kernel for (k=%d, dilation=%d)", L.ks, L.dil);
                    return FV_ERR_UNSUPPORTED;
                }
                return finish_conv_launch(L, r, tout, stream, prof_idx, name, (long long)r.batch * p.m_blks * p.n_tiles);
            }
        } else if (knobs().wino_lat && L.d_wpwl && !r.gamma && !cur_invariant() && (r.pre_act == FV_ACT_NONE || r.pre_act == FV_ACT_SILU)) {
            // launches below the gate (single clips, small batches).  Not in batch-invariant mode (another order of the K sum than conv_wino_kernel).
            if (knobs().lat_wino44 && L.d_wpq16 && (L.ks == 7 || L.ks == 11) && L.M % 32 == 0 && L.c_in % 32 == 0) {
                // k = 7 / 11: F(4,4) tap groups — 13 / 20 products per four outputs against conv_wino_lat's 20 / 32.  32-row workgroups (every staged window
                // feeds two m-tiles) while they still leave ~1.3 workgroups per CU
                const long long nt44 = (nq + 15) / 16;
                const long long wgs16 = (long long)r.batch * (L.M / 16) * nt44;
                if (knobs().lat_wino44 < 2 || 2 * wgs16 >= (long long)knobs().lat_wino44 * num_cus()) {
                    // (32 rows everywhere — stage 0 of a single clip: 88 workgroups — measured the same p50: profiles/r05q_ab_lat_wino44_rows.txt)
                    const int rows44 = (wgs16 / 2) * 3 >= 4LL * num_cus() ? 32 : 16;
                    p.wp = L.d_wpq16;
                    p.m_blks = L.M / rows44;
                    p.n_tiles = (int)nt44;
                    const int prof_idx = prof_begin(stream);
                    const bool launched = L.ks == 7 ? launch_conv_wino_lat44_k7(p, rows44, r.batch, stream) : launch_conv_wino_lat44_k11(p, rows44, r.batch, stream);
                    if (!launched) {
                        set_error("conv_layer_run: no Winograd F(4,4) latency kernel for (k=%d, dilation=%d)", L.ks, L.dil);
                        return FV_ERR_UNSUPPORTED;
                    }
                    std::snprintf(name, sizeof(name), "conv_wino_lat44<k=%d d=%d tile=%dx16q>", L.ks, L.dil, rows44);
                    return finish_conv_launch(L, r, tout, stream, prof_idx, name, (long long)r.batch * p.m_blks * p.n_tiles);
                }
            }
            const int tile = wino_lat_tile(L, np, r.batch, 1);
            const int rows = tile == 0 ? 16 : 32, lat_pairs = tile == 2 ? 32 : 16;
            p.wp = L.d_wpwl;
            p.m_blks = (int)(L.M / rows);
            p.n_tiles = (int)((np + lat_pairs - 1) / lat_pairs);
            const int prof_idx = prof_begin(stream);
            const bool launched = L.ks == 3 ? launch_conv_wino_lat_k3(p, tile, r.batch, stream)
                                  : L.ks == 7 ? launch_conv_wino_lat_k7(p, tile, r.batch, stream) : launch_conv_wino_lat_k11(p, tile, r.batch, stream);
            if (!launched) {
                set_error("conv_layer_run: no Winograd latency kernel for (k=%d, dilation=%d)", L.ks, L.dil);
                return FV_ERR_UNSUPPORTED;
            }
            std::snprintf(name, sizeof(name), "conv_wino_lat<k=%d d=%d tile=%dx%dp>", L.ks, L.dil, rows, lat_pairs);
            return finish_conv_launch(L, r, tout, stream, prof_idx, name, (long long)r.batch * p.m_blks * p.n_tiles);
        }
    }
    p.wp = L.d_wp;
    int cfg = choose_tile(L.M, p.N, r.batch);
    // the stage-0 upsampler of a 1 s clip: 87 GEMM columns per item fill two thirds of a 128-column tile — 128 x 96 tiles (four waves
    // along M, three n-tiles each; instantiated for the two-tap polyphase convs only)
    if ((cfg == TILE_128x128 || cfg == TILE_128x64) && L.ks == 2 && L.M >= 128 && p.N > 64 && p.N <= 96) cfg = TILE_128x96;
    // 256 x 64 tiles for the two-tap polyphase upsamplers with whole 256-row blocks (round 6, LOG R6.14): the staged window (three inputs + SiLU behind the
    // stack-mean) serves twice the rows — HiFiGAN-V1's stage-1 upsampler (1024 GEMM rows) 230 -> 217 us, alone on the step's critical path; same sums per
    // output (chunk, tap, channel pair order does not depend on the tile).  FV_X_UPS_TILE256=0: the 128-row tiles (A/B runs)
    static const int ups256 = std::getenv("FV_X_UPS_TILE256") ? std::atoi(std::getenv("FV_X_UPS_TILE256")) : 1;
    if (ups256 && L.transposed && L.ks == 2 && L.M % 256 == 0 && p.N > 96 && (cfg == TILE_128x128 || cfg == TILE_128x64)) cfg = TILE_256x64;
    // the last, HBM-bound upsampler (C -> C / 2 with C / 2 * stride <= 32 rows): 32 x 128 tiles (HiFiGAN step -0.06 ms; for the
    // stride-1 convs of that width — BigVGAN's last stage — the 32 x 512 tile stays: +0.26 ms with the small one)
    if (cfg == TILE_32x512 && L.transposed) cfg = TILE_32x128;
    // pointwise convs have no halo, so batch and time flatten into one GEMM column axis: no per-item partial tiles
    // (Vocos: T = 94 frames per clip would waste 27 % of a 128-column tile)
    int launch_batch = r.batch;
    if (!L.transposed && L.ks == 1 && L.pad_l == 0 && r.batch > 1 &&
        (long long)r.batch * std::max<long long>((long long)L.c_in * r.t_in, (long long)L.c_out * tout) < (1LL << 30)) {
        int cfg_flat = choose_tile(L.M, (long long)p.N * r.batch, 1);
        // wide pointwise layers (ConvNeXt's 4x expansion): 256 x 64 tiles — the same workgroup count and accumulators as
        // 128 x 128, but half the activation columns staged and read per MFMA (staging is what a k = 1 launch pays for:
        // +4 ... 9 % on the Vocos GEMMs); only when 256-row blocks add no padded rows
        if (cfg_flat == TILE_128x128 && L.M >= 512 && ((L.M + 127) / 128) % 2 == 0) cfg_flat = TILE_256x64;
        // ... and 256 x 32 instead of 128 x 64 where the half-width tile was chosen (ConvNeXt's 4x contraction: +7 %)
        if (cfg_flat == TILE_128x64 && L.M >= 512 && ((L.M + 127) / 128) % 2 == 0) cfg_flat = TILE_256x32;
        if (cfg_flat != TILE_SPLITK_32x64 && cfg_flat != TILE_SPLITK_32x32) {
            cfg = cfg_flat;
            // 2: even T and 8-byte aligned rows -> the kernel stages column pairs (conv_mfma_impl.h)
            p.flat = (p.N % 2 == 0 && ((uintptr_t)r.x & 7) == 0) ? 2 : 1;
            p.n_total = p.N * r.batch;
            launch_batch = 1;
        }
    }
    bool ok = false;
    bool specialised = true;
    auto try_launch = [&](int c) {
        int mb, nb;
        tile_dims(c, &mb, &nb);
        p.m_blks = (L.M + mb - 1) / mb;
        p.n_tiles = ((p.flat ? p.n_total : p.N) + nb - 1) / nb;
        switch (L.ks) {
            case 1: return launch_conv_k1(p, c, launch_batch, stream);
            case 3: return launch_conv_k3(p, c, r.batch, stream);
            case 7: return launch_conv_k7(p, c, r.batch, stream);
            case 11: return launch_conv_k11(p, c, r.batch, stream);
            default: return launch_conv_misc(p, c, r.batch, stream);
        }
    };
    const int prof_idx = prof_begin(stream);
    ok = try_launch(cfg);
    if (!ok) {
        if (p.x2) {   // (cannot happen for the tap counts conv_sum3_supported() accepts: they all have specialised kernels)
            set_error("conv_layer_run: no specialised kernel for (k=%d, dilation=%d) with a three-operand input", L.ks, L.dil);
            return FV_ERR_UNSUPPORTED;
        }
        specialised = false;
        p.flat = 0;
        launch_batch = r.batch;
        cfg = TILE_64x128;
        int mb, nb;
        tile_dims(cfg, &mb, &nb);
        p.m_blks = (L.M + mb - 1) / mb;
        p.n_tiles = (p.N + nb - 1) / nb;
        size_t lds = 0;
        ok = launch_conv_generic(p, cfg, r.batch, stream, &lds);
        if (!ok) {
            set_error("conv_layer_run: (k=%d, dilation=%d) needs %zu B of LDS staging, above the 64 KiB generic limit",
                      L.ks, L.dil, lds);
            return FV_ERR_UNSUPPORTED;
        }
    }
    char name[96];
    std::snprintf(name, sizeof(name), "conv_mfma<%s k=%d d=%d tile=%s>", specialised ? "spec" : "generic", L.ks, L.dil,
                  kTileNames[cfg]);
    // (split-K launches whose B operands come straight from global memory — conv_mfma_splitk_direct_kernel — carry " direct": the rule of launch_cfg())
    const bool direct = specialised && (cfg == TILE_SPLITK_32x64 || cfg == TILE_SPLITK_32x32) && knobs().splitk_direct &&
                        splitk_direct_shape(L.ks, L.dil, cfg == TILE_SPLITK_32x64 ? 2 : 1, p.x2 != nullptr);
    char tags[32];
    std::snprintf(tags, sizeof(tags), "%s%s%s", L.transposed ? " convT" : "", p.x2 ? " sum3" : "", direct ? " direct" : "");
    return finish_conv_launch(L, r, tout, stream, prof_idx, name, (long long)launch_batch * p.m_blks * p.n_tiles, tags, p.x2 != nullptr);
}

// Tile of the Winograd latency kernel for `layers` equal layers launched together: the largest one that still leaves ~2 workgroups per CU
static int wino_lat_tile(const ConvLayer& L, long long np, int batch, int layers) {
    const long long mts = L.M / 16, nts = (np + 15) / 16;
    int tile = 0;
    if (L.M % 32 == 0) {
        if ((long long)layers * batch * (mts / 2) * ((nts + 1) / 2) >= 2LL * num_cus()) tile = 2;
        else if ((long long)layers * batch * (mts / 2) * nts >= 2LL * num_cus()) tile = 1;
    }
    if (knobs().wino_lat >= 10) tile = knobs().wino_lat - 10;   // experiments: FV_WINO_LAT=10 / 11 / 12 force a tile
    if (L.M % 32 != 0) tile = 0;
    return tile;
}

// f16x3 precision mode: wide (C = 128 / 64) SiLU pairs on the fused split-fp16 kernel (pair_f16x3_impl.h)
bool pair_f16x3_supported(const ConvLayer& c1, const ConvLayer& c2) {
    const int C = c1.c_in;
    return c1.precision == FV_PRECISION_F16X3 && c2.precision == FV_PRECISION_F16X3 && c1.d_wph && c2.d_wph &&
           !c1.transposed && !c2.transposed && (C == 256 || C == 128 || C == 64 || C == 32) && c1.c_out == C && c2.c_in == C && c2.c_out == C &&
           c1.k == c2.k && (c1.k == 3 || c1.k == 7 || c1.k == 11) && (c1.dil == 1 || c1.dil == 3 || c1.dil == 5) && c2.dil == 1 &&
           c1.padding == (c1.k - 1) / 2 * c1.dil && c2.padding == (c2.k - 1) / 2 && getenv("FV_NO_F16X3_PAIRS") == nullptr;
}

static fv_status conv_pair_run_f16x3(const ConvLayer& c1, const ConvLayer& c2, const float* x, float* y, int batch, int t,
                                     int out_mode, float out_scale, hipStream_t stream) {
    if (x == y) {
        set_error("conv_pair_run: output must not alias the input (halo reads)");
        return FV_ERR_INVALID;
    }
    const int C = c1.c_in;
    if ((long long)C * t >= (1LL << 30)) {
        set_error("conv_pair_run: a batch item of %lld elements exceeds the 4 GiB buffer-addressing span", (long long)C * t);
        return FV_ERR_UNSUPPORTED;
    }
    PairF16Params p;
    std::memset(&p, 0, sizeof(p));
    p.x = x;
    p.y = y;
    p.w1h = c1.d_wph;
    p.w2h = c2.d_wph;
    p.b1 = c1.d_bias;
    p.b2 = c2.d_bias;
    p.s1 = 1.0f / c1.w_scale;
    p.s2 = 1.0f / c2.w_scale;
    p.T = t;
    p.nch16 = C / 16;
    p.out_mode = out_mode;
    p.out_scale = out_scale;
    const int prof_idx = prof_begin(stream);
    bool ok = false;
    switch (c1.k) {
        case 3: ok = launch_pair_f16x3_k3(p, C, c1.dil, batch, stream); break;
        case 7: ok = launch_pair_f16x3_k7(p, C, c1.dil, batch, stream); break;
        case 11: ok = launch_pair_f16x3_k11(p, C, c1.dil, batch, stream); break;
        default: break;
    }
    if (!ok) {
        if (dynamic_lds_refused()) return FV_ERR_HIP;
        set_error("conv_pair_run: no f16x3 pair kernel for (C=%d k=%d d=%d)", C, c1.k, c1.dil);
        return FV_ERR_UNSUPPORTED;
    }
    static thread_local char name[96];
    std::snprintf(name, sizeof(name), "pair_f16x3<k=%d d=%d C=%d>", c1.k, c1.dil, C);
    set_last_kernel(name);
    if (prof_idx >= 0) {
        const int tt = (C == 32 ? kPairF16ColsC32 : C == 64 ? 128 : 96) - (c1.k - 1);
        char lbl[128];
        std::snprintf(lbl, sizeof(lbl), "%s grid=%d", name, batch * ((t + tt - 1) / tt));
        const double macs = 2.0 * C * C * c1.k * (double)t * batch;
        const double elems = (out_mode == OUT_ACCUM ? 4.0 : 3.0) * C * (double)t * batch;   // x, residual, y (+ accumulate)
        prof_end(stream, prof_idx, lbl, 2.0 * macs, elems * 4.0 + 2.0 * C * C * c1.k * 4.0);
    }
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
}

// C = 16 in f16x3 mode (pair16_f16x3.hip): needs an even T and 8-byte aligned tensors (paired loads / stores); anything else
// keeps the exact-fp32 pair kernel
static bool pair16_f16x3_usable(const ConvLayer& c1, const ConvLayer& c2, const float* x, const float* y, int t) {
    return c1.precision == FV_PRECISION_F16X3 && c2.precision == FV_PRECISION_F16X3 && c1.d_wph16 && c2.d_wph16 &&
           c1.k == c2.k && (c1.dil == 1 || c1.dil == 3 || c1.dil == 5) && c2.dil == 1 &&
           c1.padding == (c1.k - 1) / 2 * c1.dil && c2.padding == (c2.k - 1) / 2 && t % 2 == 0 &&
           (((uintptr_t)x | (uintptr_t)y) & 7) == 0 && x != y && (long long)16 * t < (1LL << 30) &&
           getenv("FV_NO_F16X3_PAIRS") == nullptr;
}

static fv_status conv_pair16_run_f16x3(const ConvLayer& c1, const ConvLayer& c2, const float* x, float* y, int batch, int t,
                                       int out_mode, float out_scale, hipStream_t stream) {
    PairF16Params p;
    std::memset(&p, 0, sizeof(p));
    p.x = x;
    p.y = y;
    p.w1h = c1.d_wph16;
    p.w2h = c2.d_wph16;
    p.b1 = c1.d_bias;
    p.b2 = c2.d_bias;
    p.s1 = 1.0f / c1.w_scale;
    p.s2 = 1.0f / c2.w_scale;
    p.T = t;
    p.nch16 = 1;
    p.out_mode = out_mode;
    p.out_scale = out_scale;
    const int prof_idx = prof_begin(stream);
    if (!launch_pair16_f16x3(p, c1.k, c1.dil, batch, stream)) {
        if (dynamic_lds_refused()) return FV_ERR_HIP;   // error text already names the refused attribute
        set_error("conv_pair_run: no f16x3 pair kernel for (C=16 k=%d d=%d)", c1.k, c1.dil);
        return FV_ERR_UNSUPPORTED;
    }
    static thread_local char name[96];
    std::snprintf(name, sizeof(name), "pair_f16x3<k=%d d=%d C=16>", c1.k, c1.dil);
    set_last_kernel(name);
    if (prof_idx >= 0) {
        const int tt = pair16_f16x3_tile(c1.k, c1.dil);
        char lbl[128];
        std::snprintf(lbl, sizeof(lbl), "%s grid=%d", name, batch * ((t + tt - 1) / tt));
        const double macs = 2.0 * 16 * 16 * c1.k * (double)t * batch;
        const double elems = (out_mode == OUT_ACCUM ? 4.0 : 3.0) * 16 * (double)t * batch;
        prof_end(stream, prof_idx, lbl, 2.0 * macs, elems * 4.0 + 2.0 * 16 * 16 * c1.k * 4.0);
    }
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
}

bool pair_wino_supported(int C, int ks, int dil) {
    if (dil != 1 && dil != 3 && dil != 5) return false;
    if (C == 16 || C == 32) return ks == 3 || ks == 7 || ks == 11;
    return (C == 64 || C == 128) && ks == 3;   // (wider pairs fuse at k = 3 only: resblock_pair.hip)
}

fv_status conv_pair_run(const ConvLayer& c1, const ConvLayer& c2, const float* x, float* y, int batch, int t, int out_mode,
                        float out_scale, hipStream_t stream) {
    const int C = c1.c_in;
    if (pair_f16x3_supported(c1, c2)) return conv_pair_run_f16x3(c1, c2, x, y, batch, t, out_mode, out_scale, stream);
    if (pair16_f16x3_usable(c1, c2, x, y, t)) return conv_pair16_run_f16x3(c1, c2, x, y, batch, t, out_mode, out_scale, stream);
    const bool shape_ok = !c1.transposed && !c2.transposed && c1.c_out == C && c2.c_in == C && c2.c_out == C &&
                          c1.k == c2.k && c2.dil == 1 && c1.padding == (c1.k - 1) / 2 * c1.dil && c2.padding == (c2.k - 1) / 2;
    if (!shape_ok || !pair_supported(C, c1.k, c1.dil)) {
        set_error("conv_pair_run: unsupported pair (C=%d k=%d d=%d)", C, c1.k, c1.dil);
        return FV_ERR_UNSUPPORTED;
    }
    if (x == y) {
        set_error("conv_pair_run: output must not alias the input (halo reads)");
        return FV_ERR_INVALID;
    }
    // the narrow stages at k = 7 / 11: F(4,4) tap groups in both convs (pair_wino44_impl.h; round 5) — 20 / 13 products per four outputs against F(2,3)'s 32 / 20
    if (knobs().pair_wino && knobs().pair_wino44 && effective_algo() != FV_CONV_ALGO_DIRECT && c1.d_wpq16 && c2.d_wpq16 && (C == 16 || C == 32) &&
        (c1.k == 7 || c1.k == 11)) {
        PairParams p;
        std::memset(&p, 0, sizeof(p));
        p.x = x;
        p.y = y;
        p.w1 = c1.d_wpq16;
        p.w2 = c2.d_wpq16;
        p.b1 = c1.d_bias;
        p.b2 = c2.d_bias;
        p.T = t;
        p.out_mode = out_mode;
        p.out_scale = out_scale;
        const int prof_idx = prof_begin(stream);
        const bool ok = c1.k == 7 ? launch_pair_wino44_k7(p, C, c1.dil, batch, stream) : launch_pair_wino44_k11(p, C, c1.dil, batch, stream);
        if (!ok) {
            if (dynamic_lds_refused()) return FV_ERR_HIP;
            set_error("conv_pair_run: no F(4,4) pair kernel for (C=%d k=%d d=%d)", C, c1.k, c1.dil);
            return FV_ERR_UNSUPPORTED;
        }
        char name[96];
        std::snprintf(name, sizeof(name), "pair_wino44<k=%d d=%d C=%d>", c1.k, c1.dil, C);
        set_last_kernel(name);
        if (prof_idx >= 0) {
            const int nbq = 16 * (4 / (C / 16));
            const int tt = 4 * (nbq / c1.dil * c1.dil) - (c1.k - 1);   // PQGeom::TT
            char lbl[128];
            std::snprintf(lbl, sizeof(lbl), "%s grid=%d", name, batch * ((t + tt - 1) / tt));
            const double macs = 2.0 * C * C * c1.k * (double)t * batch;   // ALGORITHMIC (direct-sum) MACs of the two convs
            const double elems = (out_mode == OUT_ACCUM ? 3.0 : 2.0) * C * (double)t * batch;
            prof_end(stream, prof_idx, lbl, 2.0 * macs, elems * 4.0 + 2.0 * C * C * c1.k * 4.0);
        }
        FV_HIP_CHECK(hipGetLastError());
        return FV_OK;
    }
    const bool wide = C >= 64;   // 32x32x2 kernels on the layers' d_wpw; C <= 32: 16x16x4 kernels on d_wpw16
    if (knobs().pair_wino && effective_algo() != FV_CONV_ALGO_DIRECT && (wide ? (c1.d_wpw && c2.d_wpw) : (c1.d_wpw16 && c2.d_wpw16)) &&
        pair_wino_supported(C, c1.k, c1.dil)) {
        // Winograd F(2,3) tap groups in both convs (pair_wino_impl.h)
        PairParams p;
        std::memset(&p, 0, sizeof(p));
        p.x = x;
        p.y = y;
        p.w1 = wide ? c1.d_wpw : c1.d_wpw16;
        p.w2 = wide ? c2.d_wpw : c2.d_wpw16;
        p.n_frag = c1.nchunk * c1.nv;
        p.b1 = c1.d_bias;
        p.b2 = c2.d_bias;
        p.b1n = c1.d_bias + c1.m_pad;
        p.b2n = c2.d_bias + c2.m_pad;
        p.T = t;
        p.out_mode = out_mode;
        p.out_scale = out_scale;
        const int prof_idx = prof_begin(stream);
        const bool ok = c1.k == 3 ? launch_pair_wino_k3(p, C, c1.dil, batch, stream)
                        : c1.k == 7 ? launch_pair_wino_k7(p, C, c1.dil, batch, stream) : launch_pair_wino_k11(p, C, c1.dil, batch, stream);
        if (!ok) {
            if (dynamic_lds_refused()) return FV_ERR_HIP;
            set_error("conv_pair_run: no Winograd pair kernel for (C=%d k=%d d=%d)", C, c1.k, c1.dil);
            return FV_ERR_UNSUPPORTED;
        }
        static thread_local char name[96];
        std::snprintf(name, sizeof(name), "pair_wino<k=%d d=%d C=%d>", c1.k, c1.dil, C);
        set_last_kernel(name);
        if (prof_idx >= 0) {
            const int nbp = C == 128 ? 32 : 64;
            const int tt = 2 * (nbp / c1.dil * c1.dil) - (c1.k - 1);   // PWGeom::TT / PW32Geom::TT
            char lbl[128];
            std::snprintf(lbl, sizeof(lbl), "%s grid=%d", name, batch * ((t + tt - 1) / tt));
            const double macs = 2.0 * C * C * c1.k * (double)t * batch;   // ALGORITHMIC (direct-sum) MACs of the two convs
            const double elems = (out_mode == OUT_ACCUM ? 3.0 : 2.0) * C * (double)t * batch;
            prof_end(stream, prof_idx, lbl, 2.0 * macs, elems * 4.0 + 2.0 * C * C * c1.k * 4.0);
        }
        FV_HIP_CHECK(hipGetLastError());
        return FV_OK;
    }
    PairParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x;
    p.y = y;
    p.w1 = C == 16 ? c1.d_wp16 : c1.d_wp;
    p.w2 = C == 16 ? c2.d_wp16 : c2.d_wp;
    p.b1 = c1.d_bias;
    p.b2 = c2.d_bias;
    p.T = t;
    p.out_mode = out_mode;
    p.out_scale = out_scale;
    const int prof_idx = prof_begin(stream);
    if (!launch_resblock_pair(p, C, c1.k, c1.dil, batch, stream)) {
        if (dynamic_lds_refused()) return FV_ERR_HIP;
        set_error("conv_pair_run: no kernel for (C=%d k=%d d=%d)", C, c1.k, c1.dil);
        return FV_ERR_UNSUPPORTED;
    }
    static thread_local char name[96];
    std::snprintf(name, sizeof(name), "resblock_pair<k=%d d=%d C=%d>", c1.k, c1.dil, C);
    set_last_kernel(name);
    if (prof_idx >= 0) {
        const int tt = std::max(128, kPairCols / C) - (c1.k - 1);   // PairGeom::TT
        char lbl[128];
        std::snprintf(lbl, sizeof(lbl), "%s grid=%d", name, batch * ((t + tt - 1) / tt));
        // algorithmic work: the two convs' MACs; bytes: x once + output once (+ accumulate operand)
        const double macs = 2.0 * C * C * c1.k * (double)t * batch;
        const double elems = (out_mode == OUT_ACCUM ? 3.0 : 2.0) * C * (double)t * batch;
        prof_end(stream, prof_idx, lbl, 2.0 * macs, elems * 4.0 + 2.0 * C * C * c1.k * 4.0);
    }
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
}

}  // namespace fv
