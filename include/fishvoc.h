/*
 * fishvoc.h — C ABI of libfishvoc_hip.so, the MI355X (gfx950) vocoder inference engine.
 *
 * The reference (fishaudio/vocoder, /root/reference) has NO FFI: its generator hot path is
 * `nn.Module.forward` built by Hydra `_target_` (fish_vocoder/test.py:31,89 ->
 * fish_vocoder/models/gan.py:282-288 -> fish_vocoder/modules/generators/).  This header is
 * therefore the boundary a maintainer would bind from Python (ctypes) to replace that forward:
 * each entry point names the reference interface it stands in for.  INTEGRATION.md shows the
 * reference-side stub.
 *
 * Conventions
 *   - plain C types only; tensors are contiguous fp32, layout (B, C, T) with T fastest — the
 *     reference's own layout (hifigan.py:226, forward(x: (B, num_mels, T_mel))).
 *   - `const float* host_*` pointers are HOST memory and are copied during the call (the caller keeps
 *     ownership); `d_*` pointers are DEVICE memory borrowed for the duration of the asynchronous
 *     work enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *   - every function returns FV_OK (0) or a negative fv_status; fv_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI.
 *   - one engine per device/model; fv_forward on one engine is re-entrant only with distinct
 *     workspaces and streams.
 */
#ifndef FISHVOC_H
#define FISHVOC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FV_ABI_VERSION 5

/* every entry point below is exported with default visibility (the library is built -fvisibility=hidden) */
#if defined(__GNUC__)
#define FV_API __attribute__((visibility("default")))
#else
#define FV_API
#endif

typedef int fv_status;
enum {
    FV_OK = 0,
    FV_ERR_INVALID = -1,      /* bad argument / config (e.g. prod(upsample_rates) != hop_length) */
    FV_ERR_UNSUPPORTED = -2,  /* valid in the reference but out of scope here (use_template=True) */
    FV_ERR_MISSING_WEIGHT = -3,
    FV_ERR_SHAPE = -4,
    FV_ERR_HIP = -5,          /* a HIP runtime call failed */
    FV_ERR_STATE = -6         /* e.g. fv_forward before fv_finalize */
};

/* Model families == the reference generator classes reachable from configs/model/generator/ YAMLs */
typedef enum fv_model_kind {
    FV_MODEL_HIFIGAN = 1,  /* fish_vocoder.modules.generators.hifigan.HiFiGANGenerator (hifigan.py:136) */
    FV_MODEL_BIGVGAN = 2,  /* fish_vocoder.modules.generators.bigvgan.BigVGANGenerator (bigvgan.py:255) */
    FV_MODEL_VOCOS = 3,    /* UnifyGenerator(ConvNeXtEncoder, ISTFTHead) (unify.py:5, convnext.py:146, vocos.py:6) */
    FV_MODEL_FIREFLY = 4,  /* UnifyGenerator(ConvNeXtEncoder, HiFiGANGenerator) (firefly-gan-base.yaml) */
    FV_MODEL_CONVNEXT = 5, /* ConvNeXtEncoder alone: (B, C_in, T) -> (B, dims[-1], T) */
    FV_MODEL_ISTFT_HEAD = 6, /* ISTFTHead alone: (B, dim, T) -> (B, 1, T*hop) (vocos.py:43-69) */
    FV_MODEL_LOGMEL = 7,     /* LogMelSpectrogram: (B, 1, L) wave -> (B, n_mels, frames) (data/transforms/spectrogram.py:59-104);
                                the step right before the generator in test.py:71 */
    FV_MODEL_REFINEGAN = 8   /* fish_vocoder.modules.generators.refinegan.RefineGANGenerator (refinegan.py:182): forward(mel, template)
                                with AdaIN noise supplied by the caller (fv_forward_refinegan) */
} fv_model_kind;

#define FV_MAX_STAGES 8
#define FV_MAX_KERNELS 8
#define FV_MAX_DILATIONS 3 /* ResBlock1 / AMPBlock hard-code three (c1, c2) pairs (hifigan.py:29-93) */

/* Activation selector for the fused conv kernels (the reference "HiFiGAN" uses SiLU). */
typedef enum fv_act {
    FV_ACT_NONE = 0,
    FV_ACT_SILU = 1,       /* F.silu (hifigan.py:103,105,230) */
    FV_ACT_LEAKY_RELU = 2, /* slope in fv_conv_desc.act_slope (refinegan.py:89; kept for completeness) */
    FV_ACT_GELU = 3,       /* nn.GELU() exact erf (convnext.py:114) */
    FV_ACT_TANH = 4,       /* torch.tanh (hifigan.py:247) */
    FV_ACT_LOG_CLAMP = 5   /* log(clamp(x, min=1e-5)) (spectrogram.py:93-94) */
} fv_act;

/* Arithmetic of the MFMA-bound conv layers.  The reference computes in fp32; FV_PRECISION_F32 reproduces that
 * with fp32 matrix instructions (direct sums, or Winograd F(2,3) tap groups for the dilated ResBlock convs of launches that fill the
 * chip: fp32 products and sums throughout, as close to a float64 forward as the direct sums) and is the default.  FV_PRECISION_F16X3 is opt-in: each fp32 operand is split in two
 * fp16 planes and three fp16 products accumulate in fp32 (error ~2^-22 per product instead of 2^-24; activations must
 * stay below 65504 in magnitude) — same API, same parity tolerance, several times the throughput. */
typedef enum fv_precision {
    FV_PRECISION_F32 = 0,
    FV_PRECISION_F16X3 = 1
} fv_precision;

/* fv_upsampler_config.post_activation values beside the fv_act ones (ABI 5) */
#define FV_POST_ACT_DEFAULT 0     /* the reference default: nn.SiLU (hifigan.py:150) */
#define FV_POST_ACT_IDENTITY (-1) /* nn.Identity */

/* HiFiGANGenerator / BigVGANGenerator ctor kwargs (hifigan.py:137-151, bigvgan.py:256-270). */
typedef struct fv_upsampler_config {
    int32_t hop_length;
    int32_t num_upsamples;
    int32_t upsample_rates[FV_MAX_STAGES];
    int32_t upsample_kernel_sizes[FV_MAX_STAGES];
    int32_t num_kernels;
    int32_t resblock_kernel_sizes[FV_MAX_KERNELS];
    int32_t resblock_dilation_sizes[FV_MAX_KERNELS][FV_MAX_DILATIONS];
    int32_t num_mels;
    int32_t upsample_initial_channel;
    int32_t use_template; /* 1: pitch-template noise_convs branch (the ctor default; every shipped YAML sets false) */
    int32_t pre_conv_kernel_size;
    int32_t post_conv_kernel_size;
    /* `post_activation()` in front of conv_post (hifigan.py:150,213,245 — any nn.Module factory upstream; HiFiGAN only, BigVGAN has its own
     * activation_post).  ABI 5: **0 = FV_POST_ACT_DEFAULT = the reference default, SiLU** (partial(nn.SiLU, inplace=True)), so that a
     * zero-initialised struct builds the reference's generator; FV_ACT_SILU says the same explicitly, FV_ACT_LEAKY_RELU with
     * post_activation_slope covers nn.LeakyReLU(slope) of classic HiFi-GAN checkpoints and nn.ReLU (slope 0), FV_ACT_GELU / FV_ACT_TANH
     * their exact forms, and nn.Identity is FV_POST_ACT_IDENTITY (-1; NOT FV_ACT_NONE, whose value 0 is the default here).
     * Anything else -> FV_ERR_UNSUPPORTED. */
    int32_t post_activation;
    float post_activation_slope;
} fv_upsampler_config;

/* ConvNeXtEncoder ctor kwargs (convnext.py:147-155); drop_path is identity in eval and not represented. */
typedef struct fv_convnext_config {
    int32_t input_channels;
    int32_t num_stages;
    int32_t depths[FV_MAX_STAGES];
    int32_t dims[FV_MAX_STAGES];
    int32_t kernel_size;
} fv_convnext_config;

/* ISTFTHead ctor kwargs (vocos.py:19-26).  padding: FV_ISTFT_SAME ("same", every shipped vocos YAML: output length T * hop) or
 * FV_ISTFT_CENTER ("center": vocos 0.0.2 falls back to torch.istft(center=True) — frames overlap-added at t * hop, n_fft / 2 samples
 * trimmed from both ends, output length (T - 1) * hop; torch raises when the window envelope has a zero there, here FV_ERR_INVALID
 * at fv_finalize). */
enum { FV_ISTFT_SAME = 0, FV_ISTFT_CENTER = 1 };
typedef struct fv_istft_head_config {
    int32_t dim;
    int32_t n_fft;
    int32_t hop_length;
    int32_t win_length;
    int32_t padding;
} fv_istft_head_config;

/* LogMelSpectrogram ctor kwargs (spectrogram.py:60-70); center must be 0 (the reference default), win_length == n_fft,
 * hop_length <= n_fft (hop need not divide n_fft: resolution/24000_2048_3072.yaml).  n_mels == 0 selects LinearSpectrogram
 * alone (spectrogram.py:25-56, mode "pow2_sqrt"): the output is the (B, n_fft/2+1, frames) magnitude, no filterbank, no log. */
typedef struct fv_logmel_config {
    int32_t sample_rate;
    int32_t n_fft;
    int32_t win_length;
    int32_t hop_length;
    int32_t n_mels;
    float f_min;
    float f_max; /* <= 0: sample_rate // 2 */
} fv_logmel_config;

/* RefineGANGenerator ctor kwargs (refinegan.py:183-193); prod(downsample_rates) == prod(upsample_rates) == hop_length. */
typedef struct fv_refinegan_config {
    int32_t hop_length;
    int32_t num_stages; /* len(downsample_rates) == len(upsample_rates) */
    int32_t downsample_rates[FV_MAX_STAGES];
    int32_t upsample_rates[FV_MAX_STAGES];
    int32_t num_mels;
    int32_t start_channels;
    float leaky_relu_slope;
} fv_refinegan_config;

typedef struct fv_config {
    int32_t abi_version; /* = FV_ABI_VERSION */
    int32_t model;       /* fv_model_kind */
    fv_upsampler_config ups;   /* HIFIGAN, BIGVGAN, FIREFLY(head) */
    fv_convnext_config backbone; /* VOCOS, FIREFLY, CONVNEXT */
    fv_istft_head_config head;   /* VOCOS */
    fv_logmel_config mel;        /* LOGMEL */
    fv_refinegan_config refine;  /* REFINEGAN */
} fv_config;

typedef struct fv_engine fv_engine;

/* -------- engine lifecycle: replaces `instantiate(cfg.model)` + `load_state_dict` (test.py:31-38) -------- */

/* Validates the config exactly as the reference ctor does (assert prod(upsample_rates) == hop_length,
 * hifigan.py:154-156 -> FV_ERR_INVALID) and allocates an engine on the current HIP device. */
FV_API fv_status fv_create(const fv_config* cfg, fv_engine** out);

/* Feed one tensor of the reference state dict, by its reference name (keys as produced by the reference
 * modules, e.g. "conv_pre.parametrizations.weight.original0", "ups.0.bias",
 * "resblocks.1.blocks.2.convs2.0.parametrizations.weight.original1"; legacy "weight_g"/"weight_v" and plain
 * "weight" are accepted too).  Unknown names -> FV_ERR_INVALID; wrong shape -> FV_ERR_SHAPE.  Data is copied. */
FV_API fv_status fv_load_weight(fv_engine* e, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);

/* Fold weight-norm (w = g * v / ||v||, dim 0 — hifigan.py:31; C_in for ConvTranspose1d), re-lay-out weights
 * into MFMA fragment order and upload.  FV_ERR_MISSING_WEIGHT names the first absent tensor (strict load,
 * like test.py:37). */
FV_API fv_status fv_finalize(fv_engine* e);

FV_API void fv_destroy(fv_engine* e);

/* Selects the arithmetic of the conv layers (fv_precision).  Call between fv_create and fv_finalize (the weight planes
 * are packed at finalize; later -> FV_ERR_STATE).  No reference counterpart: torch runs this path in fp32 only. */
FV_API fv_status fv_set_precision(fv_engine* e, int32_t precision);

/* How the fp32 conv layers form their sums (FV_PRECISION_F32 only; no reference counterpart: torch picks its own convolution algorithm per
 * call).  Every choice computes in fp32 and stays well inside the parity bar (whole forwards differ by <= 2e-5 of full scale between them);
 * what changes is the LAST BITS of a clip's output and the speed:
 *   FV_CONV_ALGO_AUTO      per launch, whatever is fastest.  The dilated k = 3 / 7 / 11 ResBlock / AMPBlock convs: launches that fill the chip (>= one
 *                          workgroup per CU: depends on batch size, clip length and the device's CU count) run the throughput Winograd kernels — F(4,4) tap
 *                          groups on the quad lattice for k = 7 / 11 on layers of whole 64-row tiles, F(2,3) on the pair lattice otherwise; launches below
 *                          that gate (single clips, small batches) run the Winograd LATENCY kernels (F(4,4) for k = 7 / 11, F(2,3) for k = 3; K split over the waves), direct split-K sums
 *                          where it has no instance; the fused (c1, c2) pairs of the narrow stages use Winograd whenever a kernel exists.  Default.
 *   FV_CONV_ALGO_DIRECT    direct sums everywhere.
 *   FV_CONV_ALGO_WINOGRAD  the throughput Winograd kernels wherever one exists, whatever the launch size (never the latency kernel: single clips are
 *                          slower than under AUTO).
 * fv_set_batch_invariant(e, 1) additionally makes every choice BETWEEN DIFFERENT SUMS a function of the layer shape alone (C, k, dilation) — never of the
 * batch size or the clip length: no split-K / latency kernels, the fused pairs and the pointwise GEMM wherever their
 * shape allows, and (under FV_CONV_ALGO_AUTO) Winograd wherever a kernel exists.  A clip's output is then bit-identical whichever other clips
 * share its batch — e.g. a 37-clip batch sharded 5/5/5/5/5/4/4/4 over 8 GPUs equals the single-GPU batch bit for bit (tests pin HiFiGAN, BigVGAN,
 * Vocos, Firefly and RefineGAN; a pointwise-GEMM launch past the 32-bit offset span — > 4 GiB per layer: the default engine moves it to the k = 1 conv
 * kernel, which forms other sums — is refused with FV_ERR_UNSUPPORTED in this mode: split the batch) — at the price of
 * single-clip latency (the launch-size gates exist because small launches are faster on the direct / split-K kernels).  Without it, batches
 * that differ in size may differ in the last bits (<= 2e-5 of full scale; tests/test_gpu_models.py pins the bound).
 * Both may be called at any time; captured graphs are dropped. */
typedef enum fv_conv_algo {
    FV_CONV_ALGO_AUTO = 0,
    FV_CONV_ALGO_DIRECT = 1,
    FV_CONV_ALGO_WINOGRAD = 2
} fv_conv_algo;
FV_API fv_status fv_set_conv_algorithm(fv_engine* e, int32_t algo);
FV_API fv_status fv_set_batch_invariant(fv_engine* e, int32_t enable);

/* hipGraph replay of the static launch sequence (on by default; a call with the same buffers / shape / stream as an
 * earlier one replays a captured graph; calls on the null stream are captured on a stream the engine owns and replayed on the null stream).
 * enable = 0 makes every fv_forward* enqueue its kernels eagerly — what a server
 * that never sees the same (buffers, batch, frames) twice gets.  May be called at any time.  No reference counterpart. */
FV_API fv_status fv_set_graph_replay(fv_engine* e, int32_t enable);
/* 1 while repeated calls are replayed, 0 when every call is enqueued eagerly — by fv_set_graph_replay(e, 0), by FV_NO_GRAPH=1 /
 * FV_DEBUG_STOP in the environment at fv_create, or because stream capture failed once in this context.  A binding that routes
 * eager null-stream calls through a side stream of its own (vocoder_amd/engine.py) asks this instead of mirroring the flag. */
FV_API int32_t fv_get_graph_replay(const fv_engine* e);

/* Kernel-selection knobs for experiments (FV_PW, FV_PW_PX, FV_DWLN_NG8, FV_DWLN_RR, FV_OLD_DWLN; FV_WINO = 0 / 1 / 2 — the process-wide default
 * of fv_set_conv_algorithm: direct / auto / Winograd —, FV_WINO_MIN_M, FV_WINO_CFG, FV_WINO_MIN_BLOCKS, FV_PAIR_WINO: these change which sums are
 * formed, i.e. the last bits of the output) are read from the environment
 * ONCE per process (no getenv on the launch path); a harness that changes them afterwards calls this to re-read them.
 * Not for production use; not thread-safe against concurrent forwards.  No reference counterpart. */
FV_API void fv_reload_env(void);

/* -------- forward: replaces `self.generator(input_spec)` (gan.py:286) -------- */

/* Output length per clip for T_in input frames (T_mel * hop_length for the generators). */
FV_API int64_t fv_output_length(const fv_engine* e, int32_t t_in);
/* Output channels (1 for the generators, dims[-1] for FV_MODEL_CONVNEXT). */
FV_API int32_t fv_output_channels(const fv_engine* e);
/* Input channels (num_mels / input_channels). */
FV_API int32_t fv_input_channels(const fv_engine* e);

/* Bytes of device scratch fv_forward needs for a (batch, t_in) call. */
FV_API size_t fv_workspace_bytes(const fv_engine* e, int32_t batch, int32_t t_in);

/* d_in: (batch, C_in, t_in) fp32; d_out: (batch, C_out, fv_output_length) fp32; d_workspace: >= fv_workspace_bytes,
 * 256-byte aligned.  Asynchronous on `stream`. */
FV_API fv_status fv_forward(fv_engine* e, const float* d_in, float* d_out, int32_t batch, int32_t t_in, void* d_workspace,
                     size_t workspace_bytes, void* stream);

/* Same, for generators built with use_template=1: d_template is the pitch template (batch, 1, fv_output_length) that the
 * reference adds through strided `noise_convs` after every up-sampling stage (hifigan.py:192-204,233-234;
 * forward(x, template), hifigan.py:226).  fv_forward == fv_forward_template with d_template = NULL. */
FV_API fv_status fv_forward_template(fv_engine* e, const float* d_in, const float* d_template, float* d_out, int32_t batch,
                                     int32_t t_in, void* d_workspace, size_t workspace_bytes, void* stream);

/* RefineGANGenerator.forward(mel, template) (refinegan.py:287-323).  d_mel: (batch, num_mels, t_in); d_template: (batch, 1,
 * t_in * hop_length); d_out likewise.  The reference's AdaIN layers draw torch.randn_like on every call (refinegan.py:125);
 * here the caller supplies those standard-normal samples in d_noise — fv_refinegan_noise_elems floats, laid out as the
 * concatenation, in order of use (up-sampling stage, branch k = 3 / 7 / 11, first / second AdaIN), of (batch, C_stage, T_stage)
 * tensors — so that a run is reproducible and parity with the reference can be pinned. */
FV_API int64_t fv_refinegan_noise_elems(const fv_engine* e, int32_t batch, int32_t t_in);
FV_API fv_status fv_forward_refinegan(fv_engine* e, const float* d_mel, const float* d_template, const float* d_noise,
                                      float* d_out, int32_t batch, int32_t t_in, void* d_workspace, size_t workspace_bytes,
                                      void* stream);

/* -------- single fused conv layer (the hot kernel on its own; used by the parity tests and the roofline
 *          bench).  Stands in for one weight-normed nn.Conv1d / nn.ConvTranspose1d call plus the elementwise
 *          ops the reference runs around it (F.silu before, residual add after: hifigan.py:101-108). -------- */
typedef struct fv_conv_desc {
    int32_t transposed; /* 0: Conv1d weight (C_out, C_in, k); 1: ConvTranspose1d weight (C_in, C_out, k) */
    int32_t c_in, c_out, kernel_size;
    int32_t dilation; /* Conv1d only */
    int32_t padding;  /* Conv1d: zero padding each side; ConvTranspose1d: `padding` argument */
    int32_t stride;   /* ConvTranspose1d only (Conv1d stride is always 1 on this path) */
    int32_t pre_act;  /* fv_act applied to the input (fused into the LDS staging) */
    int32_t post_act; /* fv_act applied after bias (+ residual) */
    float act_slope;
} fv_conv_desc;

typedef struct fv_conv fv_conv;

/* host_weight: already-folded weight in the torch layout named by `transposed`; host_bias may be NULL. */
FV_API fv_status fv_conv_create(const fv_conv_desc* desc, const float* host_weight, const float* host_bias, fv_conv** out);
FV_API int64_t fv_conv_output_length(const fv_conv* c, int32_t t_in);
/* y = post_act(conv(pre_act(x)) + bias [+ d_residual]); d_residual may be NULL or alias d_y. */
FV_API fv_status fv_conv_forward(fv_conv* c, const float* d_x, float* d_y, const float* d_residual, int32_t batch,
                          int32_t t_in, void* stream);
/* One ResBlock1 iteration in a single launch: y = x + c2(silu(c1(silu(x)))) (hifigan.py:102-107).  c1 is the dilated
 * conv, c2 the dilation-1 conv; both C -> C with the same odd kernel size and 'same' padding; exact-fp32 kernels for C in {16, 32},
 * k in {3, 7, 11} and (C, k) = (64, 3); split-fp16 kernels for C in {16, 32, 64, 128, 256} when both layers are set to FV_PRECISION_F16X3 (C = 16: even t and
 * 8-byte aligned tensors, otherwise the fp32 kernel runs);
 * dilation in {1, 3, 5} (else FV_ERR_UNSUPPORTED).  d_y must not alias d_x. */
FV_API fv_status fv_conv_pair_forward(fv_conv* c1, fv_conv* c2, const float* d_x, float* d_y, int32_t batch, int32_t t,
                                      void* stream);
/* fv_precision for the following fv_conv_forward calls on this layer; layers the split-fp16 kernels do not cover
 * (transposed, C_in < 32, kernel size not in {3, 7, 11}) keep running in fp32. */
FV_API fv_status fv_conv_set_precision(fv_conv* c, int32_t precision);
/* fv_conv_algo for the following fv_conv_forward / fv_conv_pair_forward calls on this layer (for a pair: c1's setting). */
FV_API fv_status fv_conv_set_algorithm(fv_conv* c, int32_t algo);
FV_API void fv_conv_destroy(fv_conv* c);

/* -------- per-launch timing (measurement aid; bench.py's roofline leg) --------
 * Between fv_profile_begin and fv_profile_end every kernel that fv_forward enqueues is bracketed by hipEvents recorded
 * on the launch stream.  fv_profile_end synchronises them and writes a JSON array
 *   [{"kernel": label, "launches": n, "total_ms": .., "avg_ms": .., "flops_per_launch": .., "bytes_per_launch": ..}]
 * (algorithmic flops / per-layer compulsory bytes) into json_buf (truncated to cap); *needed
 * receives the full size.  The reference has no counterpart (its only timer is test.py:88-90). */
FV_API fv_status fv_profile_begin(fv_engine* e);
FV_API fv_status fv_profile_end(fv_engine* e, char* json_buf, size_t cap, size_t* needed);

/* -------- misc -------- */
FV_API const char* fv_last_error(void);
FV_API int32_t fv_abi_version(void);
/* Name of the device kernel variant the last fv_conv_forward on this thread dispatched (diagnostics). */
FV_API const char* fv_last_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* FISHVOC_H */
