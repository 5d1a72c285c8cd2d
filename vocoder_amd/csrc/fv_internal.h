// Internal declarations shared by the host engine and the HIP kernels of libfishvoc_hip.so.
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts, v_mfma_f32_32x32x2_f32, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "fishvoc.h"

namespace fv {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);
int num_cus();   // compute units of the current device (cached)

// Experiment knobs, read from the environment once per process (fv_reload_env() re-reads them): the launch path never calls getenv
struct Knobs {
    int pw = -2;          // FV_PW: forced gemm_pw configuration, -1 = "old" (the conv kernel), -2 = unset
    int pw_px = 0;        // FV_PW_PX: forced XCD row groups, 0 = unset
    bool dwln_ng8 = false, dwln_rr = false, old_dwln = false;   // FV_DWLN_NG8 / FV_DWLN_RR / FV_OLD_DWLN
    int wino = 1;         // FV_WINO: 0 = direct sums only, 1 = Winograd F(2,3) tap groups for the dilated k = 3 / 7 / 11 convs of launches that
                          // fill the chip (conv_wino_impl.h), 2 = for every eligible launch (tests)
    int wino_min_m = 32;  // FV_WINO_MIN_M: narrowest layer (output rows) that takes it
    int wino_cfg = -1;    // FV_WINO_CFG: forced tile (WinoCfg 0 ... 2), -1 = by shape
    int wino_min_blocks = -1;   // FV_WINO_MIN_BLOCKS: fewest workgroups of a launch that takes it, -1 = default
    int vec_store = 1;    // FV_VEC_STORE: 0 = the stride-8 upsamplers store single floats instead of 16-byte output quads (A/B runs, tests)
    // (FV_WINO4 / FV_WINO44 are also read when a layer is CREATED: it packs only the Winograd form they select — conv_layer.hip)
    int pair_wino44 = 1;  // FV_PAIR_WINO44: 1 = the fused narrow pairs at k = 7 / 11 on F(4,4) tap groups (pair_wino44_impl.h), 0 = F(2,3) (pair_wino_impl.h)
    int wino44 = 1;       // FV_WINO44: 1 = F(4,4) tap groups (conv_wino44_impl.h) for k = 7 / 11 where FV_WINO4 would take F(4,3), 0 = F(4,3) there
    int wino44_rows = 0;  // FV_WINO44_ROWS: 64 = one 32-row tile per wave (64-row workgroups) everywhere; otherwise two (128 rows) where the layer has whole 128-row blocks
    int wino4 = 1;        // FV_WINO4: 1 = the quad-lattice kernels (conv_wino44_impl.h / conv_wino4_impl.h) for k = 7 / 11 where the Winograd path is taken and the layer has whole 64-row blocks, 0 = F(2,3) everywhere
    int lat_wino44 = 1;   // FV_LAT_WINO44: 1 = launches below the Winograd gate at k = 7 / 11 run F(4,4) tap groups (conv_wino_lat44_impl.h), 0 = F(2,3) (conv_wino_lat_impl.h);
                          //   n >= 2: ... only launches of >= n / 2 workgroups per CU in its 16-row tiling (experiments)
    int splitk_direct = 1;   // FV_SPLITK_DIRECT: 1 = the few-tap split-K launches (conv_pre, the upsamplers of a single clip) load their B operands straight from global memory (conv_mfma_splitk_direct_kernel), 0 = staged through LDS
    int wino44_flat = 1;  // FV_WINO44_FLAT: 1 = conv_wino44 tiles a flattened (clip, quad column) axis where that needs fewer 32-column tiles than tiling every clip on its own (T = 688: 5.4 tiles per clip), 0 = per clip
    int wino_lat = 1;     // FV_WINO_LAT: 0 = launches below the Winograd gate run the direct split-K kernels, 1 = the Winograd latency kernel
    int pair_wino = 1;    // FV_PAIR_WINO: 0 = the fused (c1, c2) pairs run direct sums (resblock_pair.hip), 1 = Winograd tap groups where a kernel exists
};
const Knobs& knobs();

// Conv algorithm selection of the call being enqueued (fv_set_conv_algorithm / fv_set_batch_invariant): thread-local, set by the C-ABI entry
// points for the duration of the call — conv_layer_run / conv_pair_run and the engine's fusion decisions read it
struct AlgoScope {
    AlgoScope(int algo, bool invariant);
    ~AlgoScope();
    int prev_algo;
    bool prev_inv;
};
int cur_algo();          // fv_conv_algo
bool cur_invariant();    // kernel choices from the layer shape alone

// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}).  Register-resident operand
// rings need every index to be a constant expression in the source (an index that only becomes constant after loop unrolling
// can leave the array in scratch memory: the optimiser promotes arrays to registers before it unrolls).
template <class F, int... Is>
__device__ __host__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __host__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Optional per-launch timing (fv_profile_begin / fv_profile_end): hipEvents recorded on the launch stream around every
// kernel of a forward, aggregated by label together with the launch's ALGORITHMIC flops and bytes.
struct ProfRec {
    std::string label;
    double flops = 0, bytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};
struct Profiler {
    std::vector<ProfRec> recs;
};
Profiler* current_profiler();              // thread-local; nullptr when profiling is off
int prof_begin(hipStream_t s);             // returns record index or -1
void prof_end(hipStream_t s, int idx, const char* label, double flops, double bytes);

#define FV_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            ::fv::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return FV_ERR_HIP;                                                                   \
        }                                                                                        \
    } while (0)

// Opt a kernel in to more than 64 KiB of dynamic LDS (gfx950: 160 KiB per CU).  The attribute is PER DEVICE, so the
// "already done" state is a bit per device ordinal (one mask per call site), set with an atomic OR: a process that drives
// several GPUs, or several host threads, gets every (kernel, device) pair opted in exactly once.  Returns false and sets
// the error string when the runtime refuses.
bool ensure_dynamic_lds(const void* kernel, int bytes, unsigned long long* done_mask);
bool dynamic_lds_refused();
#define FV_ENSURE_DYN_LDS(kernel, bytes)                                                   \
    ([&]() -> bool {                                                                       \
        static unsigned long long _fv_mask = 0;                                            \
        return ::fv::ensure_dynamic_lds((const void*)(kernel), (int)(bytes), &_fv_mask);   \
    }())

// ---------------------------------------------------------------------------------------------
// Fused conv layer ("implicit GEMM on fp32 MFMA").
//
//   out[b][m][n] = post( bias[m] + sum_{ci<Cin} sum_{j<ks} W[m][ci][j] * pre(x[b][ci][n + j*dil - pad_l]) )
//
// Conv1d:           m = c_out, n = t, pad_l = padding.
// ConvTranspose1d:  polyphase form.  m = (c_out, r) with r = output phase in [0, stride), n = q, ks' = ceil(k/stride),
//                   dil = 1, pad_l = ks'-1, and the result lands at t = q*stride + r - padding (scatter store).
// ---------------------------------------------------------------------------------------------
constexpr int kChunk = 8;  // input channels staged per LDS chunk (= 4 MFMA k-steps of 2 per tap)

enum OutMode : int { OUT_SET = 0, OUT_ACCUM = 1 };  // OUT_ACCUM: y = (y_old + v) * out_scale  (MRF stack-mean)

struct ConvParams {
    const float* x;      // (B, Cin, Tin)
    const float* x2;     // SUM3 kernels: the input is ((x + x2) + x3) / 3 — the stack-mean of three ResBlock branches formed while
    const float* x3;     //   staging (same layout as x); NULL otherwise
    const float4* wp;    // packed weights, see pack_conv_weights()
    const float* bias;   // (m_pad) zero padded, never NULL
    float* y;
    const float* res;    // residual, same indexing as y (may be NULL, may alias y)
    const float* gamma;  // per-row scale applied before the residual (ConvNeXt layer-scale), may be NULL
    int Cin, Tin;
    int M, N;            // valid GEMM rows / columns per batch item
    int nchunk;          // 8-channel sub-chunks in the packed weights (ceil(Cin / 8) rounded up to a multiple of 4)
    int nchunk_real;     // ceil(Cin / 8): sub-chunks that actually hold channels
    int n_tiles, m_blks; // grid = B * m_blks * n_tiles
    int pad_l;
    int ks, dil;         // runtime copies (used by the generic variant)
    int pre_act, post_act;
    float slope;
    int out_mode;
    float out_scale;
    int convt;           // scatter store for the transposed conv
    int u, pad_t, Tout, Cout;
    long long x_bstride, y_bstride;
    int flat;            // pointwise convs (ks == 1): GEMM columns run over the flattened (batch, time) axis
    int n_total;         // flat: batch * N
    float acc_scale;     // accumulator scale applied before the bias (1 for the fp32 kernels, 1/s_w for f16x3)
    int xcd_rows;        // gemm_pw.hip: row groups the 8 XCDs split the m-tiles into (1, 2, 4 or 8)
    int wg_total;        // conv_wino_impl.h: workgroups of the launch (the grid is rounded up to a multiple of the 8 XCDs)
    int vec_store;       // conv_epilogue: stride-8 polyphase transposed conv whose output quads can go out as 16-byte stores
    int col_S, col_batch;   // conv_wino44_impl.h: > 0 = the launch's quad columns run over a flattened (clip, column) axis with col_S columns per clip, col_batch clips (grid batch 1)
#if defined(FV_X_SPLITK_TS) || defined(FV_X_CONV_TS)
    long long* dbg_ts;   // experiment: per-workgroup phase time stamps (split-K kernel / tiled conv kernel)
#endif
    // f16x3 precision mode (conv_f16x3_impl.h)
    const void* wph;     // split fp16 weight planes, see pack_conv_weights_f16x3()
    int nch16;           // 16-channel chunks in the packed planes
    int nch16_real;      // ceil(Cin / 16)
};

struct ConvLayer {
    // logical description
    bool transposed = false;
    int c_in = 0, c_out = 0, k = 0, dil = 1, padding = 0, stride = 1;
    // GEMM view
    int M = 0, ks = 0, pad_l = 0, nchunk = 0, nchunk_real = 0, m_pad = 0;
    float4* d_wp = nullptr;
    bool wino = false;         // a ResBlock / AMPBlock conv (k in {3, 7, 11}, dilation in {1, 3, 5}, 'same', C -> C): Winograd forms below, as its shape admits
    float4* d_wpw = nullptr;   // Winograd-transformed weights in the same fragment order, nv virtual taps (conv_wino_impl.h); optional
    int nv = 0;
    float4* d_wp16 = nullptr;  // 16x16x4-fragment layout, only for 16 -> 16 channel Conv1d (fused pair kernel)
    float4* d_wpw44 = nullptr; // Winograd F(4,4)-transformed weights (conv_wino44_impl.h): (32-row tile, plane half) x chunk x (3 ng + 1) fragments; optional
    float4* d_wpw4 = nullptr;  // Winograd F(4,3)-transformed weights (conv_wino4_impl.h): (32-row tile, plane half) x chunk x nv4 fragments; optional
    int nv4 = 0;
    float4* d_wpwl = nullptr;  // Winograd-transformed weights of the latency kernel: 16-row tiles, 8-channel blocks, tap pairs (conv_wino_lat_impl.h); optional
    float4* d_wpq16 = nullptr; // Winograd F(4,4)-transformed weights in 16x16x4 fragment order, C -> C, k in {7, 11}: C in {16, 32} for pair_wino44_impl.h, whole 32-row blocks for conv_wino_lat44_impl.h; optional
    float4* d_wpw16 = nullptr; // Winograd-transformed weights in 16x16x4 fragment order, C -> C with C in {16, 32} (pair_wino_impl.h); optional
    void* d_wph16 = nullptr;   // f16x3 mode, 16 -> 16 channel Conv1d: (wh, wl) planes of the two-samples-per-row layout (pair16_f16x3.hip)
    void* d_wph = nullptr;     // f16x3 mode: (wh, wl, wh * 2^-11) fp16 planes in 32x32x16 fragment order (optional)
    float w_scale = 1.f;       // s_w: power of two folded into those planes
    int nch16 = 0;
    int precision = FV_PRECISION_F32;   // FV_PRECISION_F16X3: conv_layer_run uses the split-fp16 kernel (needs d_wph)
    int algo = FV_CONV_ALGO_AUTO;       // fv_conv_set_algorithm (single layers; an engine's layers follow the engine's setting)
    float* d_bias = nullptr;   // m_pad values, followed by their negatives (m_pad more)
    size_t wp_bytes = 0;

    int64_t out_len(int t_in) const {
        return transposed ? (int64_t)(t_in - 1) * stride - 2 * padding + k
                          : (int64_t)t_in + 2 * padding - (int64_t)dil * (k - 1);
    }
    // number of GEMM columns per item
    int64_t gemm_cols(int t_in) const {
        if (!transposed) return out_len(t_in);
        const int64_t tout = out_len(t_in);
        return (tout + padding + stride - 1) / stride;  // q in [0, ceil((Tout+pad)/u))
    }
};

// Builds the device-side layer from an already-folded torch-layout weight (host).  bias may be NULL.
// with_f16x3: also pack the split fp16 planes when the layer is eligible for the f16x3 kernels.
fv_status conv_layer_create(ConvLayer& L, bool transposed, int c_in, int c_out, int k, int dil, int padding,
                            int stride, const float* host_w, const float* host_bias, bool with_f16x3 = false);
void conv_layer_destroy(ConvLayer& L);

// do the specialised kernels of this tap count form the three-operand input mean themselves (ConvParams::x2 / x3)?  The tap counts
// of the polyphase transposed convs (the upsamplers): conv_mfma_impl.h instantiates SUM3 variants for exactly these
constexpr bool conv_sum3_supported(int ks) { return ks == 1 || ks == 2 || ks == 4; }

struct ConvRun {
    const float* x = nullptr;
    // the input is ((x + x2) + x3) / 3 when x2 / x3 are given (stack-mean of three branch outputs, formed by the conv's staging
    // where the kernel supports it, otherwise by mean_of_three_kernel into sum_tmp first)
    const float* x2 = nullptr;
    const float* x3 = nullptr;
    float* sum_tmp = nullptr;
    float* y = nullptr;
    const float* res = nullptr;
    const float* gamma = nullptr;
    int batch = 0, t_in = 0;
    int pre_act = FV_ACT_NONE, post_act = FV_ACT_NONE;
    float slope = 0.f;
    int out_mode = OUT_SET;
    float out_scale = 1.f;
};
fv_status conv_layer_run(const ConvLayer& L, const ConvRun& r, hipStream_t stream);

// Per-(kernel size) translation units (parallel builds): return false if (ks, dil) has no specialisation.
bool launch_conv_k1(const ConvParams& p, int cfg, int batch, hipStream_t s);
bool launch_conv_k3(const ConvParams& p, int cfg, int batch, hipStream_t s);
bool launch_conv_k7(const ConvParams& p, int cfg, int batch, hipStream_t s);
bool launch_conv_k11(const ConvParams& p, int cfg, int batch, hipStream_t s);
bool launch_conv_misc(const ConvParams& p, int cfg, int batch, hipStream_t s);   // k=2, 4, 5, 13 ...
bool launch_conv_generic(const ConvParams& p, int cfg, int batch, hipStream_t s, size_t* lds_bytes);
// f16x3 precision mode: tiles of the split-fp16 kernel (rows x columns per workgroup)
enum SplitCfg : int { SPLIT_128x128 = 0, SPLIT_64x256 = 1, SPLIT_32x256 = 2, SPLIT_COUNT };
bool launch_conv_f16x3_k1(const ConvParams& p, int cfg, int batch, hipStream_t s);
bool launch_conv_f16x3_k3(const ConvParams& p, int cfg, int batch, hipStream_t s);
bool launch_conv_f16x3_misc(const ConvParams& p, int cfg, int batch, hipStream_t s);   // k = 2, 4 (polyphase transposed convs)
bool launch_conv_f16x3_k7(const ConvParams& p, int cfg, int batch, hipStream_t s);
bool launch_conv_f16x3_k11(const ConvParams& p, int cfg, int batch, hipStream_t s);

// Pointwise (k = 1) convs as a persistent fp32-MFMA GEMM without LDS (gemm_pw.hip).  p as for a flat conv launch
// (n_total = batch * N); pair: even T and 8-byte aligned tensors.  Returns the workgroup count (0: unknown configuration).
enum GemmPwCfg : int { GEMM_PW_64x64_W2 = 0, GEMM_PW_32x64_W3 = 1, GEMM_PW_COUNT };
int launch_gemm_pw(const ConvParams& p, int cfg, bool pair, hipStream_t s);

#ifndef FV_X_PAIRCOLS
#define FV_X_PAIRCOLS 2048   // round 3: 128-column tiles at C = 16 too (4096: 256 columns) — same B = 32 step, single-clip p50 -4 %
#endif
constexpr int kPairCols = FV_X_PAIRCOLS;   // c1 columns per workgroup x channels (LDS budget of the fused pair kernel)

// Fused ResBlock (c1, c2) pair for narrow stages: y = x + c2(silu(c1(silu(x))))  (resblock_pair.hip)
struct PairParams {
    const float* x;       // (B, C, T)
    const float4* w1;     // c1 packed weights (32x32x2 layout for C >= 32, 16x16x4 layout for C == 16)
    const float* b1;
    const float4* w2;
    const float* b2;
    const float* b1n;     // -b1 / -b2 (pair_wino_impl.h: the bias rides in two accumulator planes, one of them negated)
    const float* b2n;
    float* y;             // (B, C, T); must NOT alias x (neighbouring workgroups read x's halo)
    int T, n_tiles;
    int out_mode;
    float out_scale;
    int batch;            // items in the launch (the kernels map workgroups to (item, tile) themselves)
    int n_frag;           // pair_wino32_kernel: weight fragments per 32-row m-tile of w1 / w2 (nchunk * nv)
};
bool pair_supported(int C, int ks, int dil);

bool pair_f16x3_supported(const ConvLayer& c1, const ConvLayer& c2);   // wide SiLU pairs of the f16x3 precision mode
bool launch_resblock_pair(const PairParams& p, int C, int ks, int dil, int batch, hipStream_t s);
// Winograd F(2,3) form of the same pair (pair_wino_impl.h): p.w1 / p.w2 = the layers' d_wpw16; C in {16, 32}, dil in {1, 3, 5}
bool pair_wino_supported(int C, int ks, int dil);
bool launch_pair_wino_k3(const PairParams& p, int C, int dil, int batch, hipStream_t s);
bool launch_pair_wino_k7(const PairParams& p, int C, int dil, int batch, hipStream_t s);
bool launch_pair_wino_k11(const PairParams& p, int C, int dil, int batch, hipStream_t s);
fv_status conv_pair_run(const ConvLayer& c1, const ConvLayer& c2, const float* x, float* y, int batch, int t, int out_mode,
                        float out_scale, hipStream_t stream);

// BigVGAN AMPBlock conv with its anti-aliased SnakeBeta fused in front (amp_conv.hip): y = conv(Activation1d(x)) + bias [+ res]
// for C_in = C_out in {32, 64}, k in {3, 7, 11}, 'same' padding; alpha / inv_beta / taps as for launch_aa_snake
bool amp_conv_supported(int C, int ks, int dil);
bool launch_amp_conv(const ConvLayer& L, const float* x, float* y, const float* res, const float* alpha, const float* inv_beta,
                     const float* up_taps, const float* down_taps, int batch, int t, int out_mode, float out_scale, hipStream_t s);

// Tile configurations (block = 4 waves): rows = WM*MT*32, cols = WN*NT*32.
// conv_mfma_splitk_direct_kernel (conv_mfma_impl.h): registers of one ring slot — 4 KS NT operands (x 3: SUM3) + KS weight float4s — decide the prefetch
// distance in chunks; 0 = the rings would spill and the launch stays on the LDS-staged split-K kernel.  Few-tap convs only (conv_pre, the polyphase
// upsamplers): every other single-clip conv is on the Winograd latency kernels.  KS = 4 — the k = 4, stride-2 upsampler at C = 128: four chunks, 2.7
// workgroups per CU — measured 21.1 us this way against 18.4 staged through LDS: with few chunks and several waves per SIMD the per-operand loads cost more
// issue than the barriers they replace.
constexpr int splitk_direct_pf(int ks, int nt, bool sum3) {
    const int r = 4 * ks * nt * (sum3 ? 3 : 1) + 4 * ks;
    return r <= 32 ? 3 : r <= 48 ? 2 : r <= 56 ? 1 : 0;
}
constexpr bool splitk_direct_shape(int ks, int dil, int nt, bool sum3) {
    return (ks == 1 || ks == 2 || ks == 7) && dil == 1 && splitk_direct_pf(ks, nt, sum3) > 0;
}

enum TileCfg : int { TILE_128x128 = 0, TILE_64x256 = 1, TILE_32x512 = 2, TILE_128x64 = 3, TILE_32x128 = 4, TILE_64x128 = 5, TILE_SPLITK_32x64 = 6, TILE_SPLITK_32x32 = 7, TILE_256x64 = 8, TILE_256x32 = 9, TILE_128x96 = 10, TILE_COUNT };
void tile_dims(int cfg, int* m_blk, int* n_blk);
// conv_wino_impl.h: Winograd F(2,3) variant of the dilated "same" convs; tiles = output rows x output PAIRS per workgroup
// (two-n-tile variants — 128 x 64 and 64 x 128 pairs, 128 accumulator registers, two waves per SIMD — measured 8 % slower: LOG R3.16;
//  -DFV_X_WINO_NT2 builds them back in as configurations 3 and 4)
enum WinoCfg : int { WINO_128x32 = 0, WINO_64x64 = 1, WINO_32x128 = 2, WINO_128x64 = 3, WINO_64x128 = 4, WINO_COUNT };
// p.wp = the layer's d_wpw, p.n_tiles in pair columns
bool launch_conv_wino_k3(const ConvParams& p, int cfg, int batch, hipStream_t s);
bool launch_conv_wino_k7(const ConvParams& p, int cfg, int batch, hipStream_t s);
bool launch_conv_wino_k11(const ConvParams& p, int cfg, int batch, hipStream_t s);
// conv_wino4_impl.h: F(4,3) tap groups (k = 7 / 11), 64 rows x 32 quad columns per workgroup; p.wp = the layer's d_wpw4, p.m_blks = M / 64, p.n_tiles over quad columns
bool have_conv_wino4();   // false in the shipped library (make ABPARTNERS=1 builds the F(4,3) kernels: conv_wino4_k{7,11}.hip / abpartner_stubs.hip)
bool launch_conv_wino4_k7(const ConvParams& p, int batch, hipStream_t s);
bool launch_conv_wino4_k11(const ConvParams& p, int batch, hipStream_t s);
// conv_wino44_impl.h: F(4,4) tap groups, 64 rows x 32 quad columns per workgroup; p.wp = the layer's d_wpw44, p.m_blks = M / 64, p.n_tiles over quad columns
bool launch_pair_wino44_k7(const PairParams& p, int C, int dil, int batch, hipStream_t s);
bool launch_pair_wino44_k11(const PairParams& p, int C, int dil, int batch, hipStream_t s);
bool launch_conv_wino44_k7(const ConvParams& p, int rows, int batch, hipStream_t s);
bool launch_conv_wino44_k11(const ConvParams& p, int rows, int batch, hipStream_t s);
// conv_wino_lat_impl.h: latency variant (16 rows x 16 nt pairs per workgroup, K split over the four waves); p.wp = the layer's d_wpwl,
// p.m_blks = C / 16, p.n_tiles in units of 16 nt pair columns
bool launch_conv_wino_lat_k3(const ConvParams& p, int nt, int batch, hipStream_t s);
bool launch_conv_wino_lat_k7(const ConvParams& p, int nt, int batch, hipStream_t s);
bool launch_conv_wino_lat_k11(const ConvParams& p, int nt, int batch, hipStream_t s);
// conv_wino_lat44_impl.h: the same on F(4,4) tap groups (16 or 32 rows x 16 quad columns per workgroup); p.wp = the layer's d_wpq16, p.m_blks = C / rows,
// p.n_tiles in units of 16 quad columns
bool launch_conv_wino_lat44_k7(const ConvParams& p, int rows, int batch, hipStream_t s);
bool launch_conv_wino_lat44_k11(const ConvParams& p, int rows, int batch, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Small fused kernels (elementwise / narrow-output / reduction)
// ---------------------------------------------------------------------------------------------
// y[b][co][t] = post( bias[co] + sum_{ci,j} w[co][ci][j] * pre(x[b][ci][t + j - pad]) ), c_out <= 4 (conv_post).
// x2 / x3 (optional): the input is ((x + x2) + x3) / 3, formed while staging — only where conv_narrow_sum3_ok() says so
fv_status launch_conv_narrow(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int T,
                             int Cout, int k, int pad, int pre_act, int post_act, float slope, hipStream_t s,
                             const float* x2 = nullptr, const float* x3 = nullptr);
bool conv_narrow_sum3_ok(int B, int Cin, int T, int Cout, int k, int pad);

// Anti-aliased SnakeBeta: y = down2(snake(up2(x))) with 12-tap kaiser-sinc filters (alias_free_torch Activation1d).
// alpha_eff/inv_beta are per-channel, already exp()'d / inverted on the host.
// x2 / x3 (optional): the input is ((x + x2) + x3) / 3 (three branch outputs), formed while loading.
fv_status launch_aa_snake(const float* x, float* y, const float* alpha_eff, const float* inv_beta, const float* up_taps,
                          const float* down_taps, int B, int C, int T, hipStream_t s, const float* x2 = nullptr,
                          const float* x3 = nullptr);

// Depthwise conv (k taps, zero pad) + LayerNorm over channels, fused: y = LN_c(dwconv(x)) * w + b.
// With dw_w == NULL: plain channels-first LayerNorm.
fv_status launch_dwconv_ln(const float* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                           float* y, int B, int C, int T, int k, float eps, hipStream_t s);

// x[b][c][t] += bias[c] + sum_j w[c][j] * template[b][t*stride + j - pad]  (use_template branch, hifigan.py:233-234)
fv_status launch_noise_conv_add(const float* tmpl, const float* w, const float* bias, float* x, int B, int C, int T, int Ta,
                                int k, int stride, int pad, hipStream_t s);

// Log-mel front-end glue (small_kernels.hip)
// yp[b][r][tp] = wave[b][reflect(tp*hop + r - pad_l)] (0 past the padded length): polyphase layout so that the STFT becomes
// a stride-1 conv with hop channels and n_fft/hop taps
fv_status launch_polyphase_reflect(const float* wave, float* yp, int B, int L, int hop, int TP, int pad_l, int pad_r, hipStream_t s);
// mag[b][k][t] = sqrt(re^2 + im^2 + 1e-6) from spec rows [0,nb) = Re, [nb,2nb) = Im
fv_status launch_magnitude(const float* spec, float* mag, int B, int nb, int T, hipStream_t s);

// ISTFT head glue: h (B, 2*n_fft, T) rows [0,nb) = log-mag, [n_fft, n_fft+nb) = phase -> spec (B, 2*nbp, T):
// long clips as a batch of equal-length time tiles (engine.hip run_model); hop: frames -> samples of the tensor being moved
fv_status launch_gather_tiles(const float* x, float* tiles, int B, int C, int T, int n, int L, int stride, int hop, hipStream_t s);
fv_status launch_scatter_tiles(const float* tiles, float* y, int B, int C, int T, int n, int L, int stride, int halo, int hop, hipStream_t s);
// rows [0,nb) = Re, [nbp, nbp+nb) = Im, zero padded to nbp = round_up(nb, 8)
fv_status launch_istft_spec(const float* h, float* spec, int B, int n_fft, int T, int nb, int nbp, hipStream_t s);
// frames (B, n_fft, T) (already windowed by the synthesis basis) -> overlap-add, crop, divide by the envelope.
fv_status launch_istft_ola(const float* frames, const float* inv_env, float* y, int B, int n_fft, int T, int hop, int pad, long long out_len,
                           hipStream_t s);


// RefineGAN glue (small_kernels.hip): linear interpolation (nn.Upsample(mode="linear")) with optional leaky_relu, written into
// a channel slice of a wider tensor; channel-slice copy; AdaIN with caller-supplied noise
fv_status launch_leaky_interp(const float* x, float* y, int B, int C, int Lin, int Lout, float scale, int leaky, float slope,
                              int ctot, int coff, hipStream_t s);
fv_status launch_copy_channels(const float* x, float* y, int B, int C, int T, int ctot, int coff, hipStream_t s);
fv_status launch_mean_of_three(const float* a, const float* b, const float* c, float* y, long long n, hipStream_t s);   // ((a + b) + c) / 3
fv_status launch_adain(const float* x, const float* noise, const float* w, float* y, int B, int C, int T, float slope,
                       int accumulate, float scale, hipStream_t s);

}  // namespace fv
