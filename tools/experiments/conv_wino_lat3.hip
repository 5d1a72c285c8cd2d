// EXPERIMENT (round 4, measured, NOT kept): the convs at the same depth of a stage's three ResBlock branches (k = 11 / 7 / 3, same width,
// dilation and shapes) as ONE launch of the Winograd latency kernel on the caller's stream instead of three chains on three streams.
// Motivation: tools/latency_timeline.py on a single clip under rocprofv3 shows the sibling chains of a replayed graph starting 20 - 110 us
// apart in stages 0 / 1 (profiles/r04n_b1_timeline.txt).  Result (profiles/LOG.md R4.7): as one grid the three layers take the SUM of
// their stand-alone times wherever one layer alone fills the chip (stage 1 of a single clip: 6 x 42 us against 203 us on three streams);
// restricted to the C = 256 stage (352 workgroups per layer) the traced stage shrank 187 -> 144 us, but the untraced p50 did not move
// (0.851 / 0.856 ms with, 0.839 / 0.856 without, interleaved on one box) — the stagger is largely an artefact of tracing.
// Kept here with its host side and engine hook-up as they were wired in (conv_layer.hip / engine.hip / fv_internal.h).
#if 0
// ---- conv_wino_lat_params.h ----
// Launch parameters of conv_wino_lat3_kernel (conv_wino_lat_impl.h), shared with the host dispatch (conv_layer.hip)
#pragma once
#include "fv_internal.h"

namespace fv {

// One launch for the convs at the same depth of the three ResBlock branches of a stage (hifigan.py:117-133: k = 11, 7, 3 on the same
// input shape and dilation)
struct ConvParams3 {
    ConvParams p[3];   // k = 11, 7, 3 (equal grids: same C, T, dilation and tile)
    int per_layer;     // workgroups per layer
};
bool launch_conv_wino_lat3(const ConvParams3& t, int dil, int tile, hipStream_t s);

}  // namespace fv

// ---- conv_wino_lat_impl.h ----
// One launch for the convs at the same depth of the three ResBlock branches of a stage (hifigan.py:117-133: k = 11, 7, 3 on the same
// input shape and dilation).  On separate streams a replayed graph starts the sibling chains ~20 - 110 us apart (tools/latency_timeline.py on
// a single clip: the k = 7 chain of stage 0 began 54 us, the k = 3 chain 110 us after the k = 11 chain) and every stage pays a fork and a
// join across hardware queues; as one grid the three layers start together, longest first, on one stream.
template <int DIL, int NT, int MT>
__global__ __launch_bounds__(256, 2) void conv_wino_lat3_kernel(const ConvParams3 t) {
    constexpr int F11 = WLGeom<11, DIL, NT, MT>::WAVE_F, F7 = WLGeom<7, DIL, NT, MT>::WAVE_F, F3 = WLGeom<3, DIL, NT, MT>::WAVE_F;
    __shared__ __attribute__((aligned(16))) float lds[4 * (F11 > F7 ? (F11 > F3 ? F11 : F3) : (F7 > F3 ? F7 : F3))];
    const int which = __builtin_amdgcn_readfirstlane((int)blockIdx.x / t.per_layer);
    const int wg = (int)blockIdx.x - which * t.per_layer;
    if (which == 0) wino_lat_body<11, DIL, NT, MT>(t.p[0], lds, wg);
    else if (which == 1) wino_lat_body<7, DIL, NT, MT>(t.p[1], lds, wg);
    else wino_lat_body<3, DIL, NT, MT>(t.p[2], lds, wg);
}

template <int DIL>
inline bool launch_wino_lat3_d(const ConvParams3& t, int tile, int batch, hipStream_t s) {
    const int grid = 3 * t.per_layer;
    (void)batch;
    switch (tile) {
        case 0: hipLaunchKernelGGL((conv_wino_lat3_kernel<DIL, 1, 1>), dim3(grid), dim3(256), 0, s, t); return true;
        case 1: hipLaunchKernelGGL((conv_wino_lat3_kernel<DIL, 1, 2>), dim3(grid), dim3(256), 0, s, t); return true;
        case 2: hipLaunchKernelGGL((conv_wino_lat3_kernel<DIL, 2, 2>), dim3(grid), dim3(256), 0, s, t); return true;
        default: return false;
    }
}


// ---- conv_wino_lat3.hip ----
// The same-depth convs of a stage's three ResBlock branches in one launch (conv_wino_lat_impl.h: conv_wino_lat3_kernel)
#include "conv_wino_lat_impl.h"
namespace fv {
bool launch_conv_wino_lat3(const ConvParams3& t, int dil, int tile, hipStream_t s) {
    switch (dil) {
        case 1: return launch_wino_lat3_d<1>(t, tile, 0, s);
        case 3: return launch_wino_lat3_d<3>(t, tile, 0, s);
        case 5: return launch_wino_lat3_d<5>(t, tile, 0, s);
        default: return false;
    }
}
}  // namespace fv

// ---- conv_layer.hip ----
// Would conv_layer_run send this layer call to the Winograd latency kernel?  (the launch-size gate of conv_wino_kernel, the algorithm switch)
static bool wino_lat_applies(const ConvLayer& L, int batch, int t_in, int pre_act) {
    if (L.transposed || !L.d_wpwl || L.precision != FV_PRECISION_F32) return false;
    if (effective_algo() != FV_CONV_ALGO_AUTO || cur_invariant() || !knobs().wino_lat) return false;
    if (pre_act != FV_ACT_NONE && pre_act != FV_ACT_SILU) return false;
    if (L.M < knobs().wino_min_m) return false;
    const long long tout = L.out_len(t_in);
    if (tout <= 0 || (long long)L.c_out * tout >= (1LL << 30)) return false;
    if (L.d_wpw) {   // conv_wino_kernel's gate (conv_layer_run)
        static const int wdims[3][2] = {{128, 32}, {64, 64}, {32, 128}};
        const int wcfg = L.M > 64 ? 0 : L.M > 32 ? 1 : 2;
        const long long np = (long long)L.dil * ((tout + 2 * L.dil - 1) / (2 * L.dil));
        const long long blocks = (long long)batch * ((L.M + wdims[wcfg][0] - 1) / wdims[wcfg][0]) * ((np + wdims[wcfg][1] - 1) / wdims[wcfg][1]);
        const long long min_blocks = knobs().wino_min_blocks >= 0 ? knobs().wino_min_blocks : num_cus() / 2;
        if (blocks >= min_blocks) return false;
    }
    return true;
}

bool conv_trio_eligible(const ConvLayer* const L[3], int batch, int t_in, int pre_act) {
    static const bool off = std::getenv("FV_NO_TRIO") != nullptr;   // experiments
    if (off) return false;
    int seen = 0;
    for (int j = 0; j < 3; ++j) {
        const ConvLayer& l = *L[j];
        if (l.c_in != L[0]->c_in || l.c_out != l.c_in || l.dil != L[0]->dil || l.padding != (l.k - 1) / 2 * l.dil) return false;
        if (l.k != 3 && l.k != 7 && l.k != 11) return false;
        seen |= l.k == 11 ? 1 : l.k == 7 ? 2 : 4;
        if (!wino_lat_applies(l, batch, t_in, pre_act)) return false;
    }
    if (seen != 7) return false;
    // Only where one layer's launch cannot fill the chip by itself (< 2 workgroups per CU on 16 x 16 tiles: the C = 256 stage of a single
    // clip).  Larger launches are throughput-bound already: as one grid they took the SUM of the three layers' times (stage 1 of a single
    // clip: 6 x 42 us against 203 us on three streams, profiles/r04p_b1_timeline_trio.txt), the branch streams overlap their ramps and tails.
    const long long tout = L[0]->out_len(t_in);
    const long long np = (long long)L[0]->dil * ((tout + 2 * L[0]->dil - 1) / (2 * L[0]->dil));
    return (long long)batch * (L[0]->M / 16) * ((np + 15) / 16) < 2LL * num_cus();
}

fv_status conv_layer_run_trio(const ConvLayer* const L[3], const ConvRun r[3], hipStream_t stream) {
    ConvParams3 t;
    std::memset(&t, 0, sizeof(t));
    const ConvLayer& L0 = *L[0];
    const long long tout = L0.out_len(r[0].t_in);
    const long long np = (long long)L0.dil * ((tout + 2 * L0.dil - 1) / (2 * L0.dil));
    const int tile = wino_lat_tile(L0, np, r[0].batch, 3);
    const int rows = tile == 0 ? 16 : 32, pairs = tile == 2 ? 32 : 16;
    double macs = 0, elems = 0;
    for (int j = 0; j < 3; ++j) {
        const ConvLayer& l = *L[j];
        const ConvRun& q = r[j];
        if (q.batch != r[0].batch || q.t_in != r[0].t_in || q.x2 || q.gamma) {
            set_error("conv_layer_run_trio: the three calls must share batch / length and take one input tensor");
            return FV_ERR_INVALID;
        }
        ConvParams& p = t.p[l.k == 11 ? 0 : l.k == 7 ? 1 : 2];   // longest first inside the grid
        p.x = q.x;
        p.wp = l.d_wpwl;
        p.bias = l.d_bias;
        p.y = q.y;
        p.res = q.res;
        p.Cin = l.c_in;
        p.Tin = q.t_in;
        p.M = l.M;
        p.N = (int)tout;
        p.pad_l = l.pad_l;
        p.ks = l.ks;
        p.dil = l.dil;
        p.pre_act = q.pre_act;
        p.post_act = q.post_act;
        p.slope = q.slope;
        p.out_mode = q.out_mode;
        p.out_scale = q.out_scale;
        p.Tout = (int)tout;
        p.Cout = l.c_out;
        p.x_bstride = (long long)l.c_in * q.t_in;
        p.y_bstride = (long long)l.c_out * tout;
        p.acc_scale = 1.0f;
        p.m_blks = l.M / rows;
        p.n_tiles = (int)((np + pairs - 1) / pairs);
        macs += (double)l.c_in * l.c_out * l.k * (double)tout * q.batch;
        elems += ((double)l.c_in * q.t_in + (double)l.c_out * tout * (1 + (q.res ? 1 : 0) + (q.out_mode == OUT_ACCUM ? 1 : 0))) * q.batch;
    }
    t.per_layer = r[0].batch * t.p[0].m_blks * t.p[0].n_tiles;
    const int prof_idx = prof_begin(stream);
    if (!launch_conv_wino_lat3(t, L0.dil, tile, stream)) {
        set_error("conv_layer_run_trio: no kernel for dilation %d", L0.dil);
        return FV_ERR_UNSUPPORTED;
    }
    static thread_local char name[96];
    std::snprintf(name, sizeof(name), "conv_wino_lat3<k=11+7+3 d=%d tile=%dx%dp>", L0.dil, rows, pairs);
    set_last_kernel(name);
    if (prof_idx >= 0) {
        char lbl[160];
        std::snprintf(lbl, sizeof(lbl), "%s c=%d grid=%d", name, L0.c_in, 3 * t.per_layer);
        prof_end(stream, prof_idx, lbl, 2.0 * macs, elems * 4.0 + 21.0 * L0.c_in * L0.c_out * 4.0);
    }
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
}


// ---- engine.hip (fv_engine::run_upsampler, before the branch loop of a stage) ----
        // Launches too small to fill the chip (single clips, small batches): the three branches' convs of one depth go out as ONE launch on
        // the caller's stream (conv_wino_lat3_kernel) instead of three chains on three streams — a replayed graph started the sibling chains
        // 20 - 110 us apart and paid a fork and a join across hardware queues per stage (profiles/r04n_b1_timeline.txt).  Same buffers and
        // the same per-layer arithmetic as the branch-stream path: XB(j) keeps branch j's output for the mean.
        bool stage_trio = false;
        if (tree && !ups.bigvgan && !dbg_here) {
            const ConvLayer* l1[3] = {&stg->branches[0]->c1[0], &stg->branches[1]->c1[0], &stg->branches[2]->c1[0]};
            bool ok = ch >= 64;   // (narrower stages run the fused pair kernels)
            ok = ok && conv_trio_eligible(l1, B, t, FV_ACT_SILU);
            for (int n = 0; n < FV_MAX_DILATIONS && ok; ++n) {
                const ConvLayer* a[3] = {&stg->branches[0]->c1[n], &stg->branches[1]->c1[n], &stg->branches[2]->c1[n]};
                const ConvLayer* c[3] = {&stg->branches[0]->c2[n], &stg->branches[1]->c2[n], &stg->branches[2]->c2[n]};
                ok = conv_trio_eligible(a, B, t, FV_ACT_SILU) && conv_trio_eligible(c, B, t, FV_ACT_NONE);
            }
            stage_trio = ok;
        }
        if (stage_trio) {
            for (int n = 0; n < FV_MAX_DILATIONS; ++n) {
                const bool last = n == FV_MAX_DILATIONS - 1;
                const ConvLayer* a[3];
                const ConvLayer* c[3];
                ConvRun r1[3], r2[3];
                for (int j = 0; j < 3; ++j) {
                    ResBranch& br = *stg->branches[j];
                    a[j] = &br.c1[n];
                    c[j] = &br.c2[n];
                    const float* src = n == 0 ? S : XB(j);
                    // xt = silu(c1(silu(x)))   (the second activation rides in c1's epilogue)
                    r1[j].batch = B;
                    r1[j].t_in = t;
                    r1[j].x = src;
                    r1[j].y = XT(j);
                    r1[j].pre_act = FV_ACT_SILU;
                    r1[j].post_act = FV_ACT_SILU;
                    // x = c2(xt) + x   (the last pair leaves the branch output in XB(j): the tree mean / the next upsampler's staging reads it)
                    r2[j].batch = B;
                    r2[j].t_in = t;
                    r2[j].x = XT(j);
                    r2[j].res = src;
                    r2[j].y = XB(j);
                    (void)last;
                }
                if ((st = conv_layer_run_trio(a, r1, s))) return st;
                if ((st = conv_layer_run_trio(c, r2, s))) return st;
            }
        }

#endif
