// EXPERIMENT (round 3), NOT part of libfishvoc_hip.so — kept for the record next to its measurement (profiles/LOG.md):
// on the HiFiGAN-V1 B = 32 shapes this kernel ran the C = 128 / 64, k = 7 / 11 layers 2 % faster than the tiled LDS kernel
// (127 vs 125 TFLOP/s at k = 11, C = 128), k = 3 layers 5 - 10 % faster and the C = 256, T = 688 stage 15 % slower (too few
// 64 x 64 tiles per persistent wave); both kernels sit at ~0.9 of the 140 TFLOP/s the chip sustains at the 2.14 GHz it clocks
// under fp32 MFMA load (tools/ubench/mfma_mix.hip), so the activated-copy stores this design needs from its producers would
// cost more than it gains.  To build it again: add it to csrc/Makefile and restore the ConvParams::y2 / ConvRun::x_preactivated /
// FV_DIRECT plumbing (see tools/experiments/probe_direct.py for the A/B harness).  Ragged lengths were not yet bit-identical.
// LDS-free, barrier-free exact-fp32 MFMA convolution for the MFMA-bound ResBlock layers (C >= 64, k in {3, 7, 11}, dilation in
// {1, 3, 5}; fish_vocoder/modules/generators/hifigan.py:101-108 / bigvgan.py:235-245):
//
//   y[b][m][t] = post( bias[m] + sum_{ci, j} W[m][ci][j] * x[b][ci][t + j*dil - pad] ) [+ res[b][m][t]]      ( [, y2 = silu(y)] )
//
// The tiled kernel (conv_mfma_impl.h) stages an activation window through LDS so that every tap re-reads it from there; that
// costs two barriers per chunk, a chip-wide lock-step of the prologue / epilogue bursts and tile quantisation: 0.77 of the
// fp32-MFMA peak on its best shape (profiles/r02d_conv_phase_timeline.txt).  The pointwise GEMM (gemm_pw.hip) showed what the
// matrix pipe needs instead — v_mfma_f32_32x32x2_f32 wants ONE A and ONE B dword per lane per 64-cycle instruction, little
// enough to come straight from L1 / L2 — and reached 0.85 - 0.90 with independent waves.  This kernel is that design with
// taps: the B fragment of tap j is the same coalesced row segment shifted by j * dil elements, i.e. the same per-lane address
// with another immediate offset.  Every x element is fetched k times, but from L1 (the k fetches of a chunk are consecutive
// k-steps of the same wave): 12 B / clk / CU against the 64 B / clk the vector L1 delivers.
//
// What it needs from its caller (conv_layer.hip / engine.hip): an input that already carries its pre-activation — applying SiLU
// per fetched operand would cost k times the VALU work, and fp32 VALU instructions are matrix time on gfx950 — so the producing
// layer's epilogue stores the activated copy next to the raw tensor (ConvParams::y2: one more HBM write on layers whose
// arithmetic intensity is 90 - 360 flop/B); and tiles whose windows cross the item's ends take the slower masked path below.
//
// Structure = gemm_pw_persist_kernel: a launch has exactly CUs x 4 x W waves, each wave owns whole 64 x 64 output tiles
// (2 x 2 accumulators), operands run PD k-step groups (one tap of one 8-channel chunk = 4 k-steps = 16 MFMAs) ahead in a static
// register ring, the loads of a group are issued between the MFMAs of an earlier one, the next tile's first groups are
// requested before the current tile's epilogue, and the W waves of a SIMD start staggered.
#include "conv_mfma_impl.h"

namespace fv {

__device__ __forceinline__ float silu_d(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

template <int KS, int DIL, int PD, int W>
__global__ __launch_bounds__(256, W) void conv_direct_kernel(const ConvParams p) {
    constexpr int MT = 2, NT = 2;
    constexpr int R = PD + 1;                       // ring slots
    constexpr int NM = 4 * MT * NT;                 // MFMAs per group
    constexpr int NLDI = 4 + MT;                    // loads per interior group: 4 column-pair loads + MT weight loads
    constexpr int NLDE = 8 + MT;                    // edge tiles: the two columns of a pair are masked separately
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- tile list: column blocks (item, 64-column block) x m-tiles of 64 rows, rows fastest; XCD x owns a contiguous range of
    //      column blocks (neighbouring blocks share halo lines and both m-tiles of a block read the same activations) ----
    const int mtiles = (p.M + 63) / 64;
    const int n64 = (p.N + 63) / 64;
    const int nblk = n64 * p.n_tiles;               // n_tiles = batch items here (host)
    const int nb = gridDim.x;                       // W x CUs, CUs a multiple of 8
    const int ncu = nb / W;
    const int bslot = blockIdx.x / ncu, bcu = blockIdx.x - bslot * ncu;
    const int xcd = bcu % 8;
    const int cb0 = (int)((long long)xcd * nblk / 8), cb1 = (int)((long long)(xcd + 1) * nblk / 8);
    const long long gw = (long long)(bslot * (ncu / 8) + bcu / 8) * 4 + wave, nw = (long long)nb * 4 / 8;
    int u = __builtin_amdgcn_readfirstlane((int)gw);
    const int u1 = __builtin_amdgcn_readfirstlane((cb1 - cb0) * mtiles);
    const int ustep = __builtin_amdgcn_readfirstlane((int)nw);
    if (u >= u1) return;

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = uniform_rsrc(p.bias, (unsigned)((p.M + 127) / 128 * 128 * 4));
    const int wvoff = lane * 16;
    const int T = p.Tin;                             // == p.N: stride-1 'same' conv
    const int row2_b = __builtin_amdgcn_readfirstlane(2 * T * 4);
    const int chunk_b = __builtin_amdgcn_readfirstlane(8 * T * 4);
    const int wtile_b = __builtin_amdgcn_readfirstlane(p.nchunk * KS * 1024);   // bytes of packed weights per 32-row tile
    const int nch = p.nchunk_real;                   // host: a multiple of R
    const unsigned item_bytes = (unsigned)((long long)p.Cin * T * 4);
    const unsigned out_bytes = (unsigned)((long long)p.M * T * 4);

    // per-tile state
    int m32 = 0, ncol0 = 0, item = 0, wbase = 0;
    bool edge = false;
    unsigned voffP;                                  // interior: byte offset of (row lane >> 5, column ncol0 - pad + 2 (lane & 31))
    __amdgpu_buffer_rsrc_t xrs;
    auto setup = [&](int uu) {
        const int blk = cb0 + uu / mtiles;
        m32 = (uu % mtiles) * MT;
        item = blk / n64;
        ncol0 = (blk - item * n64) * 64;
        const int first = ncol0 - p.pad_l, last = ncol0 + 63 + (KS - 1) * DIL - p.pad_l;
        edge = first < 0 || last >= T || ncol0 + 64 > T;
        voffP = (unsigned)(((lane >> 5) * T + first + 2 * (lane & 31)) * 4);
        xrs = uniform_rsrc(p.x + (long long)item * p.x_bstride, item_bytes);
        wbase = __builtin_amdgcn_readfirstlane(m32 * wtile_b);
    };

    float a[R][MT][4];
    float b[R][4][NT];
    f32x16 acc[MT][NT];
    // load number K of group (chunk c, tap J) into ring slot SLOT; every index a constant expression (register-resident ring)
    auto issue_int = [&](auto slot_c, auto k_c, auto j_c, int c) {
        constexpr int SLOT = decltype(slot_c)::value, K = decltype(k_c)::value, J = decltype(j_c)::value;
        if constexpr (K < 4) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, voffP + (unsigned)(J * DIL * 4), c * chunk_b + K * row2_b, 0);
            b[SLOT][K][0] = __uint_as_float(v.x);
            b[SLOT][K][1] = __uint_as_float(v.y);
        } else {
            constexpr int i = K - 4;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wbase + i * wtile_b + (c * KS + J) * 1024, 0);
            a[SLOT][i][0] = __uint_as_float(v.x);
            a[SLOT][i][1] = __uint_as_float(v.y);
            a[SLOT][i][2] = __uint_as_float(v.z);
            a[SLOT][i][3] = __uint_as_float(v.w);
        }
    };
    // edge tiles: every column is tested against [0, T) (zero padding; a row's neighbours are other channels, so the hardware
    // bounds check of the descriptor cannot do it)
    auto issue_edge = [&](auto slot_c, auto k_c, auto j_c, int c) {
        constexpr int SLOT = decltype(slot_c)::value, K = decltype(k_c)::value, J = decltype(j_c)::value;
        if constexpr (K < 8) {
            constexpr int pp = K / 2, e = K % 2;
            const int t = ncol0 - p.pad_l + 2 * (lane & 31) + e + J * DIL;
            const unsigned off = (t >= 0 && t < T) ? (unsigned)(((lane >> 5) * T + t) * 4) : 0xFFFFFFF0u;
            b[SLOT][pp][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, off, c * chunk_b + pp * row2_b, 0));
        } else {
            issue_int(slot_c, std::integral_constant<int, K - 4>{}, j_c, c);
        }
    };
    auto issue_group = [&](auto edge_c, auto slot_c, auto j_c, int c) {   // all loads of one group at once (priming)
        constexpr bool E = decltype(edge_c)::value;
        static_for<(E ? NLDE : NLDI)>([&](auto k_c) {
            if constexpr (E) issue_edge(slot_c, k_c, j_c, c);
            else issue_int(slot_c, k_c, j_c, c);
        });
    };
    // MFMAs of ring slot RS, with the loads of group (chunk cl, tap JL) into slot SL spread between them
    auto step = [&](auto edge_c, auto rs_c, auto loads_c, auto sl_c, auto jl_c, int cl) {
        constexpr bool E = decltype(edge_c)::value;
        constexpr int RS = decltype(rs_c)::value;
        constexpr bool LOADS = decltype(loads_c)::value;
        constexpr int NLD = E ? NLDE : NLDI;
        static_for<NM>([&](auto m_c) {
            constexpr int m = decltype(m_c)::value;
            constexpr int pp = m / (MT * NT), i = (m / NT) % MT, jn = m % NT;
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[RS][i][pp], b[RS][pp][jn], acc[i][jn], 0, 0, 0);
            if constexpr (LOADS) {
                static_for<NLD>([&](auto k_c) {
                    constexpr int k = decltype(k_c)::value;
                    if constexpr (k * NM / NLD == m) {
                        if constexpr (E) issue_edge(sl_c, k_c, jl_c, cl);
                        else issue_int(sl_c, k_c, jl_c, cl);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    // groups are numbered g = c * KS + j; group g lives in ring slot g % R.  The chunk loop is unrolled by R so that slot and tap are
    // constant expressions: body position q = cc * KS + j (cc < R) <-> slot q % R (R * KS is a multiple of R).
    auto prime = [&](auto edge_c) {
        static_for<PD>([&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            issue_group(edge_c, std::integral_constant<int, g % R>{}, std::integral_constant<int, g % KS>{}, g / KS);
        });
    };
    auto mainloop = [&](auto edge_c) {
        static_for<MT>([&](auto i_c) {
            static_for<NT>([&](auto j_c) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[decltype(i_c)::value][decltype(j_c)::value][r] = 0.f;
            });
        });
        for (int c0 = 0; c0 < nch; c0 += R) {
            static_for<R * KS>([&](auto q_c) {
                constexpr int q = decltype(q_c)::value;
                constexpr int ql = q + PD;                       // body position of the group requested now
                constexpr int ccl = ql / KS, jl = ql % KS;       // (ccl may be R: first chunk of the next revolution)
                // past the tile's last group the requests re-read the last chunk (L1 hits, results never used: the next prime()
                // overwrites those ring slots) — a branch-free body keeps exact s_waitcnt counts and the ring in registers
                int cl = c0 + ccl;
                if constexpr (ql >= R * KS) cl = cl < nch ? cl : nch - 1;
                step(edge_c, std::integral_constant<int, q % R>{}, std::true_type{}, std::integral_constant<int, ql % R>{},
                     std::integral_constant<int, jl>{}, cl);
            });
        }
    };

    auto epilogue = [&](int e_m32, int e_ncol0, int e_item) {
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)e_item * p.y_bstride, out_bytes);
        const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc((p.res ? p.res : p.y) + (long long)e_item * p.y_bstride, out_bytes);
        const __amdgpu_buffer_rsrc_t y2rs = uniform_rsrc((p.y2 ? p.y2 : p.y) + (long long)e_item * p.y_bstride, out_bytes);
        const bool has_res = p.res != nullptr, has_y2 = p.y2 != nullptr;
        const int n = e_ncol0 + 2 * (lane & 31);                 // this lane's column pair (n, n + 1)
        const bool ok0 = n < T, ok1 = n + 1 < T;
        constexpr int EB = 8;
        static_for<MT>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            static_for<16 / EB>([&](auto hb_c) {
                constexpr int hb = decltype(hb_c)::value;
                const int mrow = (e_m32 + i) * 32 + 4 * (lane >> 5) + 2 * EB * hb;   // acc register r <-> row (r & 3) + 8 (r >> 2)
                float bs[EB], rv[EB][NT];
#pragma unroll
                for (int rq = 0; rq < EB / 4; ++rq) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)((mrow + 8 * rq) * 4), 0, 0);
                    bs[4 * rq + 0] = __uint_as_float(v.x);
                    bs[4 * rq + 1] = __uint_as_float(v.y);
                    bs[4 * rq + 2] = __uint_as_float(v.z);
                    bs[4 * rq + 3] = __uint_as_float(v.w);
                }
                unsigned off[EB];
#pragma unroll
                for (int r = 0; r < EB; ++r) {
                    const int m = mrow + (r & 3) + 8 * (r >> 2);
                    off[r] = (m < p.M && ok0) ? (unsigned)(m * T + n) * 4u : 0xFFFFFFF0u;
                }
                if (has_res) {
#pragma unroll
                    for (int r = 0; r < EB; ++r) {
                        if (ok1) {
                            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rrs, off[r], 0, 0);
                            rv[r][0] = __uint_as_float(v.x);
                            rv[r][1] = __uint_as_float(v.y);
                        } else {
                            rv[r][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, off[r], 0, 0));
                            rv[r][1] = 0.f;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < EB; ++r) {
                    float val[NT];
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn) {
                        float v = acc[i][jn][EB * hb + r] + bs[r];
                        if (has_res) v += rv[r][jn];
                        val[jn] = v;
                    }
                    if (p.post_act == FV_ACT_SILU) {
                        val[0] = silu_d(val[0]);
                        val[1] = silu_d(val[1]);
                    } else if (p.post_act != FV_ACT_NONE) {
                        act_apply_all(val, p.post_act, p.slope);
                    }
                    if (ok1) {
                        u32x2 v;
                        v.x = __float_as_uint(val[0]);
                        v.y = __float_as_uint(val[1]);
                        __builtin_amdgcn_raw_buffer_store_b64(v, yrs, off[r], 0, 0);
                        if (has_y2) {
                            v.x = __float_as_uint(silu_d(val[0]));
                            v.y = __float_as_uint(silu_d(val[1]));
                            __builtin_amdgcn_raw_buffer_store_b64(v, y2rs, off[r], 0, 0);
                        }
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[0]), yrs, off[r], 0, 0);
                        if (has_y2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(silu_d(val[0])), y2rs, off[r], 0, 0);
                    }
                }
            });
        });
    };

    if (W > 1) {   // staggered start of the W waves of a SIMD (gemm_pw.hip): afterwards their epilogues alternate
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID.wave_id
        const long long wait = (long long)(slot % W) * ((long long)nch * KS * NM * 64) / W;
        const long long t0 = (long long)__builtin_amdgcn_s_memtime();
        while ((long long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    setup(u);
    if (edge) prime(std::true_type{});
    else prime(std::false_type{});
    for (;;) {
        if (edge) mainloop(std::true_type{});
        else mainloop(std::false_type{});
        const int e_m32 = m32, e_ncol0 = ncol0, e_item = item;
        u += ustep;
        const bool more = u < u1;
        if (more) {   // the next tile's first groups travel while this tile's epilogue runs
            setup(u);
            if (edge) prime(std::true_type{});
            else prime(std::false_type{});
        }
        epilogue(e_m32, e_ncol0, e_item);
        if (!more) break;
    }
}

template <int KS, int DIL>
static int launch_direct(const ConvParams& p, hipStream_t s) {
    constexpr int W = 2, PD = 3;
    const int grid = num_cus() / 8 * 8 * W;
    hipLaunchKernelGGL((conv_direct_kernel<KS, DIL, PD, W>), dim3(grid), dim3(256), 0, s, p);
    return grid;
}

bool conv_direct_supported(int c_in, int c_out, int ks, int dil) {
    return c_in % 32 == 0 && c_in >= 64 && c_out % 64 == 0 && (ks == 3 || ks == 7 || ks == 11) && (dil == 1 || dil == 3 || dil == 5);
}

// p: a stride-1 'same' conv (Tin == N), n_tiles = batch items, input already activated.  Returns the workgroup count (0: no kernel).
int launch_conv_direct(const ConvParams& p, hipStream_t s) {
#define FV_DIRECT_CASE(K, D) \
    if (p.ks == K && p.dil == D) return launch_direct<K, D>(p, s);
    FV_DIRECT_CASE(3, 1) FV_DIRECT_CASE(3, 3) FV_DIRECT_CASE(3, 5)
    FV_DIRECT_CASE(7, 1) FV_DIRECT_CASE(7, 3) FV_DIRECT_CASE(7, 5)
    FV_DIRECT_CASE(11, 1) FV_DIRECT_CASE(11, 3) FV_DIRECT_CASE(11, 5)
#undef FV_DIRECT_CASE
    return 0;
}

}  // namespace fv
