// Exact-fp32 MFMA GEMM for the pointwise (k = 1) convs of the ConvNeXt trunk — the one MFMA-bound segment north_star names:
//   Linear(C -> 4C) -> GELU and Linear(4C -> C) -> * gamma -> + input   (fish_vocoder/modules/encoders/convnext.py:130-141),
//   the stage 1x1 convs (convnext.py:177-182) and the ISTFT head's projection (vocos.py:26,57).
//
//   Y[b][m][t] = post( (bias[m] + sum_k W[m][k] * X[b][k][t]) * gamma[m] + res[b][m][t] )
//   GEMM view: M = C_out, N = batch * T (batch and time flattened: a pointwise conv has no halo), K = C_in.
//
// Why its own kernel: the general conv kernel (conv_mfma_impl.h) stages an activation window through LDS because every tap
// re-reads it; at k = 1 there is nothing to re-read, and stage -> barrier -> ds_read was 15 - 25 % of a launch.  Here there is
// NO LDS AND NO BARRIER: v_mfma_f32_32x32x2_f32 needs one A and one B dword per lane per 64-cycle instruction, little enough
// to come straight from L1 / L2 into registers:
//   B (activations): X is (B, K, T) with T fastest, and the MFMA B fragment is "lane l holds column l & 31 of row l >> 5",
//     i.e. 32 consecutive floats of two consecutive K rows — already a coalesced access to X.  With an even T a lane loads
//     the column PAIR (2c, 2c + 1) with one 8-byte buffer load and feeds two n-tiles (tile e holds columns 2c + e; the
//     epilogue stores the pairs back as 8-byte stores), so a k-step costs NT / 2 vector-memory instructions per wave.
//   A (weights): the packed fragment-order layout of conv_layer.hip, one 16-byte buffer load per m-tile per 4 k-steps.
//   Both operands run PD chunks (of 8 channels = 4 k-steps) ahead in a static register ring (the K loop is unrolled by
//   the ring length, so there are no register moves); addresses live in SGPRs (descriptor base + scalar offset) plus one
//   constant VGPR per load stream: the main loop has no VALU instruction at all — on gfx950 fp32 VALU work and fp32 MFMA
//   share issue, so every VALU instruction removed is matrix time.
//   Waves never synchronise; 1 - 3 of them per SIMD (persistent launch, see gemm_pw_persist_kernel).
#include "conv_mfma_impl.h"

namespace fv {

// Persistent form of the same GEMM.  What the one-tile-per-wave kernel above loses on the short-K layers (512 -> 2048: the
// whole launch is 1.5 rounds of 4 waves per SIMD) is lock-step and quantisation: every wave of a round sits in its prologue
// (first loads), main loop and epilogue (GELU + a burst of stores) at the same time, and the last round is partly empty.
// Here a launch has exactly as many waves as the chip holds (CUs x 4 SIMDs x W); the output is cut into tiles of MT x 32
// rows x 64 columns, ordered rows-fastest inside a 64-column block, and dealt out round by round (wave g takes tiles g,
// g + waves, ...; the blocks of one XCD are neighbours in that order, so a round's column blocks are read once per XCD and
// the weights stay in every L2).  The host picks (MT, W) per layer so that the fullest SIMD is close to the average
// (conv_layer.hip).
// A wave requests the first chunks of its NEXT tile before it runs the epilogue of the current one (XPF), and spreads the
// loads of the chunk PD ahead between the MFMAs of the current chunk.
template <int MT, int PD, bool PAIR, int W, bool XPF, bool STAGGER, int EB>
__global__ __launch_bounds__(256, W) void gemm_pw_persist_kernel(const ConvParams p, long long* ts) {
    constexpr int NT = 2;
    constexpr int NL = PAIR ? 1 : 2;
    constexpr int R = PD + 1;
    constexpr int NM = 4 * MT * NT, NLD = 4 * NL + MT;   // MFMAs / loads per chunk
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // tile list
    const int mtiles = (p.M + 32 * MT - 1) / (32 * MT);
    const int n64 = (p.n_total + 63) / 64;
    // Wave order.  The dispatcher deals blocks round-robin: block b runs on XCD b % 8, and blocks b, b + CUs, b + 2 CUs ... share
    // a CU, one wave per SIMD each (observed, tools/probe_pw_timeline.py; only speed depends on it).  The 8 XCDs split the
    // output PX x PY (PX row groups of m-tiles, PY column ranges): an XCD's L2 then has to hold only 1 / PX of the weights next
    // to the activation columns streaming through it.  Inside an XCD waves are ordered slot-major (all first blocks of its CUs,
    // then all second blocks ...) and wave w takes the XCD's tiles w, w + waves, ...: a round hands neighbouring tiles to
    // neighbouring waves, and a partial last round fills the first slot of every SIMD before any second one.
    const int nb = gridDim.x;           // W x CUs, CUs a multiple of 8 (host)
    const int ncu = nb / W;
    const int bslot = blockIdx.x / ncu, bcu = blockIdx.x - bslot * ncu;
    const int xcd = bcu % 8;
    const int px = p.xcd_rows, py = 8 / px;            // host: px in {1, 2, 4, 8}, mtiles % px == 0
    const int xr = xcd % px, xc = xcd / px;
    const int mtl = mtiles / px;                        // m-tiles of this XCD's row group
    const int cb0 = (int)((long long)xc * n64 / py), cb1 = (int)((long long)(xc + 1) * n64 / py);
    const long long gw = (long long)(bslot * (ncu / 8) + bcu / 8) * 4 + wave, nw = (long long)nb * 4 / 8;   // wave index / waves of this XCD
    int u = __builtin_amdgcn_readfirstlane((int)gw);
    const int u1 = __builtin_amdgcn_readfirstlane((cb1 - cb0) * mtl);
    const int ustep = __builtin_amdgcn_readfirstlane((int)nw);
    if (u >= u1) return;
    const long long t_start = ts ? (long long)wall_clock64() : 0;
    const int u_first = u;

    const unsigned x_bytes = (unsigned)((long long)(p.n_total / p.N) * p.x_bstride * 4);
    const unsigned y_bytes = (unsigned)((long long)(p.n_total / p.N) * p.y_bstride * 4);
    const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(p.x, x_bytes);
    const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y, y_bytes);
    const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(p.res ? p.res : p.y, y_bytes);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = uniform_rsrc(p.bias, (unsigned)((p.M + 127) / 128 * 128 * 4));   // zero padded to 128 rows (conv_layer.hip)
    const __amdgpu_buffer_rsrc_t grs = uniform_rsrc(p.gamma ? p.gamma : p.bias, (unsigned)(p.M * 4));
    const int wvoff = lane * 16;
    const int row2_b = __builtin_amdgcn_readfirstlane(2 * p.Tin * 4);
    const int chunk_b = __builtin_amdgcn_readfirstlane(8 * p.Tin * 4);
    const int wtile_b = __builtin_amdgcn_readfirstlane(p.nchunk * 1024);   // bytes of packed weights per 32-row tile
    const int nch = p.nchunk_real;
    const bool has_res = p.res != nullptr;
    const bool has_gamma = p.gamma != nullptr;

    // per-tile state: first 32-row tile, first column, the B stream's per-lane offset, the A stream's scalar base
    unsigned voffB[NL];
    int m32 = 0, ncol0 = 0, wbase = 0;
    auto setup = [&](int uu) {
        const int blk = uu / mtl;
        m32 = (xr * mtl + (uu - blk * mtl)) * MT;
        ncol0 = (cb0 + blk) * 64;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int n = PAIR ? ncol0 + 2 * (lane & 31) : ncol0 + q * 32 + (lane & 31);
            const int bb = n / p.N;
            const int t = n - bb * p.N;
            voffB[q] = n < p.n_total ? (unsigned)(bb * (int)p.x_bstride + (lane >> 5) * p.Tin + t) * 4u : 0xFFFFFFF0u;
        }
        wbase = __builtin_amdgcn_readfirstlane(m32 * wtile_b);
    };

    float a[R][MT][4];
    float b[R][4][NT];
    f32x16 acc[MT][NT];
    // Load number K of chunk c into ring slot SLOT.  Every ring index is a constant expression (static_for): an index that only
    // becomes constant after unrolling can leave the array in scratch memory.
    auto issue_one = [&](auto slot_c, auto k_c, int c) {
        constexpr int SLOT = decltype(slot_c)::value, K = decltype(k_c)::value;
        if constexpr (K < 4 * NL) {
            constexpr int pp = K / NL, q = K % NL;
            const int so = c * chunk_b + pp * row2_b;
            if constexpr (PAIR) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, voffB[0], so, 0);
                b[SLOT][pp][0] = __uint_as_float(v.x);
                b[SLOT][pp][1] = __uint_as_float(v.y);
            } else {
                b[SLOT][pp][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, voffB[q], so, 0));
            }
        } else {
            constexpr int i = K - 4 * NL;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wbase + i * wtile_b + c * 1024, 0);
            a[SLOT][i][0] = __uint_as_float(v.x);
            a[SLOT][i][1] = __uint_as_float(v.y);
            a[SLOT][i][2] = __uint_as_float(v.z);
            a[SLOT][i][3] = __uint_as_float(v.w);
        }
    };
    // The MFMAs of ring slot RS with (LOADS) the loads of chunk c spread between them: a vector-memory instruction takes tens
    // of cycles to issue, and an in-order wave hides that only under an MFMA that is already executing.
    auto step = [&](auto rs_c, auto loads_c, auto slot_c, int c) {
        constexpr int RS = decltype(rs_c)::value;
        constexpr bool LOADS = decltype(loads_c)::value;
        static_for<NM>([&](auto m_c) {
            constexpr int m = decltype(m_c)::value;
            constexpr int pp = m / (MT * NT), i = (m / NT) % MT, jn = m % NT;
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[RS][i][pp], b[RS][pp][jn], acc[i][jn], 0, 0, 0);
            if constexpr (LOADS) {
                static_for<NLD>([&](auto k_c) {
                    constexpr int k = decltype(k_c)::value;
                    if constexpr (k * NM / NLD == m) {
                        issue_one(slot_c, k_c, c);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto prime = [&]() {   // chunks 0 .. PD-1 (host guarantees nch >= PD)
        static_for<PD>([&](auto d_c) { static_for<NLD>([&](auto k_c) { issue_one(d_c, k_c, decltype(d_c)::value); }); });
    };
    auto mainloop = [&]() {
        static_for<MT>([&](auto i_c) {
            static_for<NT>([&](auto j_c) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[decltype(i_c)::value][decltype(j_c)::value][r] = 0.f;
            });
        });
        // whole ring revolutions with every request inside K: a branch-free body (exact s_waitcnt counts) ...
        int c0 = 0;
        for (; c0 + R + PD <= nch; c0 += R) {
            static_for<R>([&](auto r_c) {
                constexpr int r = decltype(r_c)::value;
                step(r_c, std::true_type{}, std::integral_constant<int, (r + PD) % R>{}, c0 + r + PD);
            });
        }
        // ... then the last PD .. PD + R - 1 chunks: requests only while they stay inside K
        static_for<R - 1 + PD>([&](auto st_c) {
            constexpr int st = decltype(st_c)::value;
            const int c = c0 + st;
            if (c < nch) {
                if (c + PD < nch) step(std::integral_constant<int, st % R>{}, std::true_type{}, std::integral_constant<int, (st + PD) % R>{}, c + PD);
                else step(std::integral_constant<int, st % R>{}, std::false_type{}, std::integral_constant<int, 0>{}, 0);
            }
        });
    };
    auto epilogue = [&](int e_m32, int e_ncol0) {
        int coff[NL];
        bool cok[NL];
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int n = PAIR ? e_ncol0 + 2 * (lane & 31) : e_ncol0 + q * 32 + (lane & 31);
            const int bb = n / p.N;
            coff[q] = bb * (int)p.y_bstride + (n - bb * p.N);
            cok[q] = n < p.n_total;
        }
        static_for<MT>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            static_for<16 / EB>([&](auto hb_c) {
                constexpr int hb = decltype(hb_c)::value;
                // everything these EB rows (acc registers EB hb .. EB hb + EB - 1) read — bias, layer scale, residual — is requested
                // up front: one memory latency per half m-tile, not one per row group
                const int mrow = (e_m32 + i) * 32 + 4 * (lane >> 5) + 2 * EB * hb;   // acc register r <-> row (r & 3) + 8 (r >> 2)
                float gm[EB], bs[EB];
                float rv[EB][NT];
#pragma unroll
                for (int rq = 0; rq < EB / 4; ++rq) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)((mrow + 8 * rq) * 4), 0, 0);
                    bs[4 * rq + 0] = __uint_as_float(v.x);
                    bs[4 * rq + 1] = __uint_as_float(v.y);
                    bs[4 * rq + 2] = __uint_as_float(v.z);
                    bs[4 * rq + 3] = __uint_as_float(v.w);
                }
                if (has_gamma) {
#pragma unroll
                    for (int rq = 0; rq < EB / 4; ++rq) {
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(grs, (unsigned)((mrow + 8 * rq) * 4), 0, 0);
                        gm[4 * rq + 0] = __uint_as_float(v.x);
                        gm[4 * rq + 1] = __uint_as_float(v.y);
                        gm[4 * rq + 2] = __uint_as_float(v.z);
                        gm[4 * rq + 3] = __uint_as_float(v.w);
                    }
                }
                unsigned off[EB][NL];
#pragma unroll
                for (int r = 0; r < EB; ++r) {
                    const int m = mrow + (r & 3) + 8 * (r >> 2);
#pragma unroll
                    for (int q = 0; q < NL; ++q) off[r][q] = (m < p.M && cok[q]) ? (unsigned)(m * p.N + coff[q]) * 4u : 0xFFFFFFF0u;
                }
                if (has_res) {
#pragma unroll
                    for (int r = 0; r < EB; ++r) {
                        if constexpr (PAIR) {
                            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rrs, off[r][0], 0, 0);
                            rv[r][0] = __uint_as_float(v.x);
                            rv[r][1] = __uint_as_float(v.y);
                        } else {
                            rv[r][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, off[r][0], 0, 0));
                            rv[r][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, off[r][1], 0, 0));
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < EB; ++r) {
                    float val[NT];
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn) {
                        float v = acc[i][jn][EB * hb + r] + bs[r];
                        if (has_gamma) v *= gm[r];
                        if (has_res) v += rv[r][jn];
                        val[jn] = v;
                    }
                    if (p.post_act == FV_ACT_GELU) {
#pragma unroll
                        for (int jn = 0; jn < NT; ++jn) val[jn] = gelu_fast(val[jn]);
                    } else if (p.post_act != FV_ACT_NONE) {
                        act_apply_all(val, p.post_act, p.slope);
                    }
                    if constexpr (PAIR) {
                        u32x2 v;
                        v.x = __float_as_uint(val[0]);
                        v.y = __float_as_uint(val[1]);
                        __builtin_amdgcn_raw_buffer_store_b64(v, yrs, off[r][0], 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[0]), yrs, off[r][0], 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[1]), yrs, off[r][1], 0, 0);
                    }
                }
            });
        });
    };

    if (STAGGER && W > 1) {
        // The W waves of a SIMD start together and own equal work: left alone they reach every epilogue (VALU, stores, the
        // latency of the bias / residual loads) at the same moment and the matrix pipe idles through it.  Wave slot s therefore
        // waits s / W of one tile's MFMA time before it starts — its partners have the pipe to themselves meanwhile (one wave
        // with four independent accumulators keeps it busy) — and from then on the epilogues of a SIMD's waves alternate.
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID.wave_id: slot on its SIMD
        const long long wait = (long long)(slot % W) * ((long long)nch * NM * 64) / W;
        const long long t0 = (long long)__builtin_amdgcn_s_memtime();
        while ((long long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    setup(u);
    prime();
    for (;;) {
        mainloop();
        const int e_m32 = m32, e_ncol0 = ncol0;
        u += ustep;
        const bool more = u < u1;
        if (XPF && more) {   // the next tile's first chunks travel while this tile's epilogue runs
            setup(u);
            prime();
        }
        epilogue(e_m32, e_ncol0);
        if (!more) break;
        if (!XPF) {
            setup(u);
            prime();
        }
    }
    if (ts && lane == 0) {
        const long long gi = ((long long)xcd * nw + gw) * 4;
        ts[gi + 0] = t_start;
        ts[gi + 1] = (long long)wall_clock64();
        ts[gi + 2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) | ((long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);   // HW_ID, XCC_ID
        ts[gi + 3] = ((long long)u_first << 32) | (unsigned)((u1 - u_first + ustep - 1) / ustep + u_first);
    }
}

// Debug aid (tools/probe_pw_timeline.py): when set, every wave of the persistent kernel records {start, end} of its life in
// 100 MHz wall-clock ticks plus its hardware id and tile range: 4 x int64 per wave.
static long long* g_pw_ts = nullptr;
extern "C" __attribute__((visibility("default"))) void fv_debug_set_pw_timestamps(void* device_buffer) { g_pw_ts = (long long*)device_buffer; }

template <int MT, int PD, int W, bool XPF, bool STAGGER = false, int EB = 8>
static int launch_pw_persist(const ConvParams& p, bool pair, hipStream_t s) {
    const int grid = num_cus() / 8 * 8 * W;   // one workgroup = one wave per SIMD; W of them fill a CU
    if (pair)
        hipLaunchKernelGGL((gemm_pw_persist_kernel<MT, PD, true, W, XPF, STAGGER, EB>), dim3(grid), dim3(256), 0, s, p, g_pw_ts);
    else
        hipLaunchKernelGGL((gemm_pw_persist_kernel<MT, PD, false, W, XPF, STAGGER, EB>), dim3(grid), dim3(256), 0, s, p, g_pw_ts);
    return grid;
}

// Kernel configurations (fv_internal.h GemmPwCfg): tile rows / waves per SIMD.
int launch_gemm_pw(const ConvParams& p, int cfg, bool pair, hipStream_t s) {
    switch (cfg) {
        case GEMM_PW_64x64_W2: return launch_pw_persist<2, 4, 2, true, true>(p, pair, s);   // 64 x 64 tiles, 4 chunks ahead, 2 waves / SIMD
        case GEMM_PW_32x64_W3: return launch_pw_persist<1, 3, 3, true, true>(p, pair, s);   // 32 x 64 tiles, 3 chunks ahead, 3 waves / SIMD
        // (64 x 64 tiles at 3 waves / SIMD do not fit 168 registers even with 4-row epilogue batches: 109 spills)
        default: return 0;
    }
}

}  // namespace fv
