/*
 * fv_oracle.c — CPU restatement of the fish_vocoder generator forward path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle (and the "port" CPU
 * baseline timed by bench.py).  Nothing in vocoder_amd/ (the product) may
 * import, link or call it; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Every routine restates, in plain C over contiguous fp32 (B, C, T) tensors,
 * one primitive the reference delegates to torch, citing the reference call
 * site it stands in for (paths relative to /root/reference):
 *
 *   fvo_weight_norm          torch weight_norm(dim=0) on every conv
 *                            (fish_vocoder/modules/generators/hifigan.py:31-57,158,178,214)
 *   fvo_conv1d               nn.Conv1d incl. dilation / groups (hifigan.py:31-93,158-166,214-222;
 *                            fish_vocoder/modules/encoders/convnext.py:103-109,165-171,180)
 *   fvo_conv_transpose1d     nn.ConvTranspose1d (hifigan.py:177-187)
 *   fvo_silu / fvo_tanh      F.silu / torch.tanh (hifigan.py:103,105,230,245,247)
 *   fvo_leaky_relu           (refinegan only; kept for the activation template)
 *   fvo_snake                Snake / SnakeBeta (fish_vocoder/modules/generators/bigvgan.py:60-71,121-135)
 *   fvo_upsample_fir / fvo_downsample_fir / fvo_kaiser_sinc_filter
 *                            alias_free_torch==0.0.6 UpSample1d / DownSample1d /
 *                            kaiser_sinc_filter1d (third-party, NOT in /root/reference;
 *                            call sites bigvgan.py:9,226-233,335-337) — restated from the
 *                            package's published algorithm: PARITY UNPINNED for this piece.
 *   fvo_layernorm_cf         convnext.LayerNorm, both data formats reduce over C
 *                            (convnext.py:66-74)
 *   fvo_gelu                 nn.GELU() exact erf form (convnext.py:114)
 *   fvo_scale_residual       gamma * x + input (convnext.py:134-141)
 *   fvo_istft_head_post      exp / clip / cos / sin of ISTFTHead (fish_vocoder/modules/generators/vocos.py:57-67)
 *   fvo_istft_same / _crop   vocos==0.0.2 spectral_ops.ISTFT(padding="same" / "center") (third-party, NOT in
 *                            /root/reference; call sites vocos.py:3,33-38,69) — restated from the
 *                            package's published algorithm: PARITY UNPINNED for this piece.
 *
 * Arithmetic is fp32 with fp32 accumulation (like the reference's MKLDNN path);
 * transcendental helpers use libm float functions.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define FVO_API __attribute__((visibility("default")))

FVO_API int fvo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

FVO_API void fvo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* w[i, :] = g[i] * v[i, :] / ||v[i, :]||_2   (torch._weight_norm(v, g, dim=0)).
 * n0 = size of dim 0 (C_out for Conv1d, C_in for ConvTranspose1d), inner = product of the rest. */
FVO_API void fvo_weight_norm(const float* g, const float* v, float* w, int64_t n0, int64_t inner) {
    for (int64_t i = 0; i < n0; ++i) {
        const float* vi = v + i * inner;
        double s = 0.0;
        for (int64_t j = 0; j < inner; ++j) s += (double)vi[j] * (double)vi[j];
        float scale = g[i] / (float)sqrt(s);
        for (int64_t j = 0; j < inner; ++j) w[i * inner + j] = vi[j] * scale;
    }
}

/* ---- register-blocked direct convolution (groups == 1) ---------------------------------------------------------
 * Every output element is still  bias + the products in (ci ascending, tap ascending) order, one fused multiply-add
 * each — the summation order of the plain loops further down, which this replaces for speed only (the plain form
 * streamed one output row per (ci, tap) pass and was memory-bound: ~20 GFLOP/s on one core, and it stopped scaling
 * past a quarter of a 256-thread host).  Here a block of COB output channels x TVB*8 time steps stays in vector
 * registers across the whole (ci, tap) loop; x is read from a zero-padded copy of the item so that the inner loop has
 * no bounds tests (adding w * 0 leaves a sum unchanged). */
typedef float v8f __attribute__((vector_size(32)));
#define COB 4
#define TVB 3
#define TBLK (TVB * 8)
#define TSB 21   /* time blocks per L2-resident super-block (504 samples) */

static inline v8f ld8(const float* p) {
    v8f v;
    memcpy(&v, p, sizeof(v));
    return v;
}

/* xp: (Cin, Tp) zero-padded rows, xp[ci][t + j*dil] is the input of output t and tap j.  w: (Cout, Cin, k). */
static void conv_block(const float* xp, int64_t Tp, const float* w, const float* bias, float* y, int Cin, int Cout,
                       int Tout, int k, int dil, int co0, int t0) {
    const int nco = Cout - co0 < COB ? Cout - co0 : COB;
    const int nt = Tout - t0 < TBLK ? Tout - t0 : TBLK;
    v8f acc[COB][TVB];
    for (int c = 0; c < COB; ++c) {
        const float bv = (c < nco && bias) ? bias[co0 + c] : 0.0f;
        for (int v = 0; v < TVB; ++v) acc[c][v] = (v8f){bv, bv, bv, bv, bv, bv, bv, bv};
    }
    const float* wr[COB];
    for (int c = 0; c < COB; ++c) wr[c] = w + (int64_t)(co0 + (c < nco ? c : 0)) * Cin * k;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* xr = xp + (int64_t)ci * Tp + t0;
        for (int j = 0; j < k; ++j) {
            const float* xo = xr + (int64_t)j * dil;
            const v8f x0 = ld8(xo), x1 = ld8(xo + 8), x2 = ld8(xo + 16);
            for (int c = 0; c < COB; ++c) {
                const float wv = wr[c][(int64_t)ci * k + j];
                const v8f wb = (v8f){wv, wv, wv, wv, wv, wv, wv, wv};
                acc[c][0] += wb * x0;
                acc[c][1] += wb * x1;
                acc[c][2] += wb * x2;
            }
        }
    }
    for (int c = 0; c < nco; ++c) {
        float tmp[TBLK];
        memcpy(tmp, &acc[c][0], sizeof(tmp));
        memcpy(y + (int64_t)(co0 + c) * Tout + t0, tmp, (size_t)nt * sizeof(float));
    }
}

/* y[b, co, t] = bias[co] + sum_{ci in group} sum_j w[co, ci, j] * x[b, g*cpg+ci, t*1 + j*dil - pad]
 * stride fixed to 1 on the generator path except the anti-alias down-sampler, which has its own routine.
 * w: (Cout, Cin/groups, k).  T_out = T + 2*pad - dil*(k-1). */
FVO_API void fvo_conv1d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int T,
                        int Cout, int k, int dil, int pad, int groups) {
    const int Tout = T + 2 * pad - dil * (k - 1);
    const int cin_g = Cin / groups, cout_g = Cout / groups;
    if (groups == 1 && Tout > 0) {
        /* padded rows: `pad` zeros in front, then x, then zeros up to the last element a (partial) time block may touch */
        const int n_tb = (Tout + TBLK - 1) / TBLK;
        const int64_t Tp = (int64_t)n_tb * TBLK + (int64_t)dil * (k - 1) + 8;
        float* xp = (float*)calloc((size_t)B * Cin * Tp, sizeof(float));
        if (xp) {
#pragma omp parallel for collapse(2) schedule(static)
            for (int b = 0; b < B; ++b)
                for (int ci = 0; ci < Cin; ++ci) {
                    /* x[t] sits at padded index t + pad; negative pads (never used on this path) would crop instead */
                    const int lo = pad < 0 ? -pad : 0;
                    if (T > lo)
                        memcpy(xp + ((int64_t)b * Cin + ci) * Tp + (pad > 0 ? pad : 0), x + ((int64_t)b * Cin + ci) * T + lo,
                               (size_t)(T - lo) * sizeof(float));
                }
            /* loop order: a super-block of TSB time blocks (all Cin rows of it: <= 0.5 MB at Cin = 256) stays in the core's L2 while
             * every output-channel block sweeps over it, so x is streamed from memory once per conv, not Cout / COB times */
            const int n_cb = (Cout + COB - 1) / COB;
            const int n_tsb = (n_tb + TSB - 1) / TSB;
#pragma omp parallel for collapse(3) schedule(static)
            for (int b = 0; b < B; ++b)
                for (int tsb = 0; tsb < n_tsb; ++tsb)
                    for (int cb = 0; cb < n_cb; ++cb) {
                        const int tb1 = (tsb + 1) * TSB < n_tb ? (tsb + 1) * TSB : n_tb;
                        for (int tb = tsb * TSB; tb < tb1; ++tb)
                            conv_block(xp + (int64_t)b * Cin * Tp, Tp, w, bias, y + (int64_t)b * Cout * Tout, Cin, Cout, Tout, k, dil,
                                       cb * COB, tb * TBLK);
                    }
            free(xp);
            return;
        }
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int co = 0; co < Cout; ++co) {
            float* yr = y + ((int64_t)b * Cout + co) * Tout;
            const float bv = bias ? bias[co] : 0.0f;
            for (int t = 0; t < Tout; ++t) yr[t] = bv;
            const int g = co / cout_g;
            /* time-blocked so the output block stays in L1 while all (ci, tap) pairs stream over it */
            for (int tb = 0; tb < Tout; tb += 2048) {
                const int te = tb + 2048 < Tout ? tb + 2048 : Tout;
                for (int ci = 0; ci < cin_g; ++ci) {
                    const float* xr = x + ((int64_t)b * Cin + g * cin_g + ci) * T;
                    const float* wr = w + ((int64_t)co * cin_g + ci) * k;
                    for (int j = 0; j < k; ++j) {
                        const float wv = wr[j];
                        const int off = j * dil - pad; /* x index = t + off */
                        int t0 = off < 0 ? -off : 0;
                        int t1 = T - off < Tout ? T - off : Tout;
                        if (t0 < tb) t0 = tb;
                        if (t1 > te) t1 = te;
                        float* restrict yo = yr;
                        const float* restrict xo = xr + off;
#pragma omp simd
                        for (int t = t0; t < t1; ++t) yo[t] += wv * xo[t];
                    }
                }
            }
        }
    }
}

/* y[b, co, i*stride - pad + j] += x[b, ci, i] * w[ci, co, j];  w: (Cin, Cout, k)
 * T_out = (Tin-1)*stride - 2*pad + k   (output_padding 0, dilation 1, groups 1)
 * Evaluated as `stride` polyphase stride-1 convolutions on the blocked kernel above: output n = q*stride + r - pad
 * (phase r = (n + pad) mod stride) collects taps j = r, r + stride, ... from inputs i = q - (j - r)/stride, i.e. a
 * conv over q with kp = ceil((k - r)/stride) taps.  Per output element the products are still added in (ci ascending,
 * j ascending) order — the order of the scatter loops this replaces. */
FVO_API void fvo_conv_transpose1d(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                                  int Tin, int Cout, int k, int stride, int pad) {
    const int Tout = (Tin - 1) * stride - 2 * pad + k;
    if (Tout <= 0) return;
    for (int r = 0; r < stride && r < k; ++r) {
        const int kp = (k - r + stride - 1) / stride;            /* taps of this phase: j = r + m*stride, m < kp */
        /* outputs of the phase: n = q*stride + r - pad in [0, Tout)  ->  q in [q_lo, q_hi) */
        const int q_lo = pad - r > 0 ? (pad - r + stride - 1) / stride : 0;
        const int q_hi = (Tout - 1 - r + pad) / stride + 1;
        const int nq = q_hi - q_lo;
        if (nq <= 0) continue;
        /* phase weights as a Conv1d kernel over q: u[q] = sum_ci sum_m wp[co][ci][m'] * x[ci][q + m' - (kp-1)], where
         * m' = kp-1-m reverses the taps — but that would reverse the order of the additions, so instead the INPUT is
         * reversed: with xr[ci][s] = x[ci][Tin-1-s], u is produced for reversed q and tap m ascending stays ascending. */
        float* wp = (float*)malloc((size_t)Cout * Cin * kp * sizeof(float));
        float* xr = (float*)malloc((size_t)B * Cin * Tin * sizeof(float));
        const int Tc = Tin + kp - 1;                               /* conv output length with padding kp-1 on both sides */
        float* u = (float*)malloc((size_t)B * Cout * Tc * sizeof(float));
        if (!wp || !xr || !u) {
            /* out of memory for the phase buffers: the plain gather form of the same sums (ci ascending, tap ascending, one
             * fused multiply-add per product — the whole file is compiled with -ffp-contract=fast, see the Makefile) */
            free(wp);
            free(xr);
            free(u);
            for (int b = 0; b < B; ++b)
                for (int co = 0; co < Cout; ++co)
                    for (int q = q_lo; q < q_hi; ++q) {
                        float acc = bias ? bias[co] : 0.0f;
                        for (int ci = 0; ci < Cin; ++ci)
                            for (int m = 0; m < kp; ++m) {
                                const int i = q - m;
                                if (i >= 0 && i < Tin)
                                    acc = __builtin_fmaf(w[((int64_t)ci * Cout + co) * k + r + m * stride], x[((int64_t)b * Cin + ci) * Tin + i], acc);
                            }
                        y[((int64_t)b * Cout + co) * Tout + q * stride + r - pad] = acc;
                    }
            continue;
        }
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci)
                for (int m = 0; m < kp; ++m) wp[((int64_t)co * Cin + ci) * kp + m] = w[((int64_t)ci * Cout + co) * k + r + m * stride];
#pragma omp parallel for schedule(static)
        for (int64_t row = 0; row < (int64_t)B * Cin; ++row)
            for (int s = 0; s < Tin; ++s) xr[row * Tin + s] = x[row * Tin + (Tin - 1 - s)];
        /* uc[s'] = bias + sum_ci sum_m wp[m] * xr_padded[s' + m]  with xr_padded[p] = xr[p - (kp-1)]:
         * xr index = s' + m - (kp-1) = Tin-1-i  ->  i = Tin-1 - s' - m + kp-1;  we need i = q - m  ->  s' = Tin-1 + kp-1 - q */
        fvo_conv1d(xr, wp, bias, u, B, Cin, Tin, Cout, kp, 1, kp - 1, 1);
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int co = 0; co < Cout; ++co) {
                const float* ur = u + ((int64_t)b * Cout + co) * Tc;
                float* yr = y + ((int64_t)b * Cout + co) * Tout;
                for (int q = q_lo; q < q_hi; ++q) yr[q * stride + r - pad] = ur[Tin - 1 + kp - 1 - q];
            }
        free(wp);
        free(xr);
        free(u);
    }
    /* phases r >= k (stride > k) receive no tap: bias only */
    for (int r = k; r < stride; ++r)
        for (int b = 0; b < B; ++b)
            for (int co = 0; co < Cout; ++co)
                for (int n = 0; n < Tout; ++n)
                    if ((n + pad) % stride == r) y[((int64_t)b * Cout + co) * Tout + n] = bias ? bias[co] : 0.0f;
}

FVO_API void fvo_silu(const float* x, float* y, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = x[i] / (1.0f + expf(-x[i]));
}

FVO_API void fvo_tanh(const float* x, float* y, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = tanhf(x[i]);
}

FVO_API void fvo_leaky_relu(const float* x, float* y, int64_t n, float slope) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = x[i] >= 0.0f ? x[i] : slope * x[i];
}

FVO_API void fvo_gelu(const float* x, float* y, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = 0.5f * x[i] * (1.0f + erff(x[i] * 0.70710678118654752440f));
}

/* Snake / SnakeBeta: y = x + 1/(beta + 1e-9) * sin(alpha*x)^2, per-channel alpha, beta
 * (beta == alpha pointer for plain Snake); logscale -> exp(param) first (bigvgan.py:128-133). */
FVO_API void fvo_snake(const float* x, const float* alpha, const float* beta, float* y, int B, int C, int T,
                       int logscale) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int c = 0; c < C; ++c) {
            float a = alpha[c], bt = beta[c];
            if (logscale) {
                a = expf(a);
                bt = expf(bt);
            }
            const float inv = 1.0f / (bt + 0.000000001f);
            const float* xr = x + ((int64_t)b * C + c) * T;
            float* yr = y + ((int64_t)b * C + c) * T;
            for (int t = 0; t < T; ++t) {
                const float s = sinf(xr[t] * a);
                yr[t] = xr[t] + inv * (s * s);
            }
        }
    }
}

/* ---- alias_free_torch 0.0.6 (restated; parity unpinned) -------------------------------------- */

static double bessel_i0(double x) {
    /* power series, converges fast for the beta range used here (< 20) */
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-20 * sum) break;
    }
    return sum;
}

/* kaiser_sinc_filter1d(cutoff, half_width, kernel_size) -> taps (normalised to sum 1).
 * torch.kaiser_window(periodic=False): w[n] = I0(beta*sqrt(1-((n-N/2')/(N/2'))^2))/I0(beta), N' = ks-1. */
FVO_API void fvo_kaiser_sinc_filter(double cutoff, double half_width, int ks, float* taps) {
    const int even = (ks % 2 == 0);
    const int half = ks / 2;
    const double delta_f = 4.0 * half_width;
    const double A = 2.285 * (half - 1) * M_PI * delta_f + 7.95;
    double beta;
    if (A > 50.0)
        beta = 0.1102 * (A - 8.7);
    else if (A >= 21.0)
        beta = 0.5842 * pow(A - 21.0, 0.4) + 0.07886 * (A - 21.0);
    else
        beta = 0.0;
    double* f = (double*)malloc(sizeof(double) * ks);
    double sum = 0.0;
    for (int n = 0; n < ks; ++n) {
        const double r = (ks > 1) ? (2.0 * n / (double)(ks - 1) - 1.0) : 0.0;
        double arg = 1.0 - r * r;
        if (arg < 0) arg = 0;
        const double win = bessel_i0(beta * sqrt(arg)) / bessel_i0(beta);
        const double tm = even ? ((double)(n - half) + 0.5) : (double)(n - half);
        const double xs = 2.0 * cutoff * tm;
        const double sinc = (xs == 0.0) ? 1.0 : sin(M_PI * xs) / (M_PI * xs);
        f[n] = (cutoff == 0.0) ? 0.0 : 2.0 * cutoff * win * sinc;
        sum += f[n];
    }
    for (int n = 0; n < ks; ++n) taps[n] = (float)((cutoff == 0.0) ? 0.0 : f[n] / sum);
    free(f);
}

/* UpSample1d(ratio, ks): replicate-pad(pad,pad), pad = ks/ratio - 1; y = ratio * conv_transpose1d(x, taps,
 * stride=ratio) (depthwise); crop pad_left = pad*ratio + (ks-ratio)/2, pad_right = pad*ratio + (ks-ratio+1)/2.
 * Output length = T*ratio. */
FVO_API void fvo_upsample_fir(const float* x, const float* taps, float* y, int B, int C, int T, int ratio, int ks) {
    const int pad = ks / ratio - 1;
    const int pad_left = pad * ratio + (ks - ratio) / 2;
    const int Tout = T * ratio;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)B * C; ++r) {
        const float* xr = x + r * T;
        float* yr = y + r * Tout;
        for (int n = 0; n < Tout; ++n) {
            /* full transposed-conv index m = n + pad_left = i*ratio + j, i over padded input [0, T+2pad) */
            const int m = n + pad_left;
            float acc = 0.0f;
            for (int j = m % ratio; j < ks; j += ratio) {
                const int i = (m - j) / ratio; /* padded index */
                if (i < 0 || i >= T + 2 * pad) continue;
                int src = i - pad;
                if (src < 0) src = 0;
                if (src > T - 1) src = T - 1;
                acc += taps[j] * xr[src];
            }
            yr[n] = (float)ratio * acc;
        }
    }
}

/* DownSample1d(ratio, ks) = LowPassFilter1d(stride=ratio): replicate-pad(ks/2 - even, ks/2) then depthwise
 * conv1d stride=ratio.  Output length = floor((T + pl + pr - ks)/ratio) + 1. */
FVO_API void fvo_downsample_fir(const float* x, const float* taps, float* y, int B, int C, int T, int ratio, int ks) {
    const int even = (ks % 2 == 0);
    const int pl = ks / 2 - even, pr = ks / 2;
    const int Tout = (T + pl + pr - ks) / ratio + 1;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)B * C; ++r) {
        const float* xr = x + r * T;
        float* yr = y + r * Tout;
        for (int n = 0; n < Tout; ++n) {
            float acc = 0.0f;
            for (int j = 0; j < ks; ++j) {
                int src = n * ratio + j - pl;
                if (src < 0) src = 0;
                if (src > T - 1) src = T - 1;
                acc += taps[j] * xr[src];
            }
            yr[n] = acc;
        }
    }
}

/* ---- ConvNeXt pieces -------------------------------------------------------------------------- */

/* LayerNorm over the channel dim of a (B, C, T) tensor: u = mean_c, s = mean_c (x-u)^2 (biased),
 * y = (x-u)/sqrt(s+eps) * w[c] + b[c]  (convnext.py:66-74; the channels_last F.layer_norm branch is the
 * same arithmetic on the permuted tensor). */
FVO_API void fvo_layernorm_cf(const float* x, const float* w, const float* bvec, float* y, int B, int C, int T,
                              float eps) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int t = 0; t < T; ++t) {
            const float* xb = x + (int64_t)b * C * T + t;
            float* yb = y + (int64_t)b * C * T + t;
            float u = 0.0f;
            for (int c = 0; c < C; ++c) u += xb[(int64_t)c * T];
            u /= (float)C;
            float s = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float d = xb[(int64_t)c * T] - u;
                s += d * d;
            }
            s /= (float)C;
            const float inv = 1.0f / sqrtf(s + eps);
            for (int c = 0; c < C; ++c) yb[(int64_t)c * T] = (xb[(int64_t)c * T] - u) * inv * w[c] + bvec[c];
        }
    }
}

/* y = res + gamma[c] * x  (gamma may be NULL -> 1) */
FVO_API void fvo_scale_residual(const float* x, const float* gamma, const float* res, float* y, int B, int C,
                                int T) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const int64_t o = ((int64_t)b * C + c) * T;
            const float gm = gamma ? gamma[c] : 1.0f;
            for (int t = 0; t < T; ++t) y[o + t] = res[o + t] + gm * x[o + t];
        }
}

/* ---- Vocos ISTFT head ------------------------------------------------------------------------- */

/* h: (B, 2*n_fft, T) output of the 1x1 conv.  mag = min(exp(h[:, :n_fft]), 100); p = h[:, n_fft:];
 * re = mag*cos p, im = mag*sin p  -> (B, n_fft, T) each (vocos.py:57-67). */
FVO_API void fvo_istft_head_post(const float* h, float* re, float* im, int B, int n_fft, int T) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < n_fft; ++k) {
            const float* mg = h + ((int64_t)b * 2 * n_fft + k) * T;
            const float* ph = h + ((int64_t)b * 2 * n_fft + n_fft + k) * T;
            float* r = re + ((int64_t)b * n_fft + k) * T;
            float* i = im + ((int64_t)b * n_fft + k) * T;
            for (int t = 0; t < T; ++t) {
                float m = expf(mg[t]);
                if (m > 100.0f) m = 100.0f;
                r[t] = m * cosf(ph[t]);
                i[t] = m * sinf(ph[t]);
            }
        }
}

/* ISTFT(padding="same"): frames = irfft(S[:, :n_fft/2+1], n_fft) * hann(win) ; overlap-add with hop ;
 * crop pad=(win-hop)/2 both ends ; divide by the overlap-added window^2 envelope.  re/im: (B, NB, T) where
 * only the first n_fft/2+1 bins are read (torch.fft.irfft trims).  y: (B, T*hop).  Requires win == n_fft.
 * irfft is evaluated as a direct real DFT in double (the oracle favours obviousness over speed). */
/* pad = samples trimmed from both ends of the overlap-add: (win - hop) / 2 for padding="same" (vocos ISTFT.forward), n_fft / 2 for
 * padding="center" (the package falls back to torch.istft(center=True): same frames, same window-envelope division, other trim) */
FVO_API void fvo_istft_crop(const float* re, const float* im, float* y, int B, int NB, int T, int n_fft, int hop, int win,
                            int pad) {
    const int nb = n_fft / 2 + 1;
    const int full = (T - 1) * hop + win;
    const int Tout = full - 2 * pad;
    double* window = (double*)malloc(sizeof(double) * win);
    for (int n = 0; n < win; ++n) window[n] = (double)(float)(0.5 - 0.5 * cos(2.0 * M_PI * n / (double)win));
    double* env = (double*)calloc(full, sizeof(double));
    for (int t = 0; t < T; ++t)
        for (int n = 0; n < win; ++n) env[t * hop + n] += window[n] * window[n];
    /* twiddles */
    double* ctab = (double*)malloc(sizeof(double) * n_fft);
    double* stab = (double*)malloc(sizeof(double) * n_fft);
    for (int n = 0; n < n_fft; ++n) {
        ctab[n] = cos(2.0 * M_PI * n / (double)n_fft);
        stab[n] = sin(2.0 * M_PI * n / (double)n_fft);
    }
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        double* acc = (double*)calloc(full, sizeof(double));
        double* fr = (double*)malloc(sizeof(double) * n_fft);
        for (int t = 0; t < T; ++t) {
            for (int n = 0; n < n_fft; ++n) {
                /* irfft: x[n] = (1/N) [Re X0 + (-1)^n Re X_{N/2} + 2 sum_{k=1}^{N/2-1} (Re Xk cos - Im Xk sin)] */
                double s = (double)re[((int64_t)b * NB + 0) * T + t];
                s += ((n & 1) ? -1.0 : 1.0) * (double)re[((int64_t)b * NB + (nb - 1)) * T + t];
                for (int k = 1; k < nb - 1; ++k) {
                    const int idx = (int)(((int64_t)k * n) % n_fft);
                    s += 2.0 * ((double)re[((int64_t)b * NB + k) * T + t] * ctab[idx] -
                                (double)im[((int64_t)b * NB + k) * T + t] * stab[idx]);
                }
                fr[n] = s / (double)n_fft;
            }
            for (int n = 0; n < win; ++n) acc[t * hop + n] += fr[n] * window[n];
        }
        for (int n = 0; n < Tout; ++n) y[(int64_t)b * Tout + n] = (float)(acc[n + pad] / env[n + pad]);
        free(acc);
        free(fr);
    }
    free(window);
    free(env);
    free(ctab);
    free(stab);
}

FVO_API void fvo_istft_same(const float* re, const float* im, float* y, int B, int NB, int T, int n_fft, int hop,
                            int win) {
    fvo_istft_crop(re, im, y, B, NB, T, n_fft, hop, win, (win - hop) / 2);
}
