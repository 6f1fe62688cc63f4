/*
 * whisper_oracle.c — CPU restatement of the whisper.axera hot path.
 * TEST INFRASTRUCTURE ONLY (see whisper_oracle.h): never linked into the product library.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Parity status: front-end pinned against oracle/_ref (the reference's own librosa.h build);
 * encoder/decoder pinned against transformers-generated goldens (tests/golden), because the
 * arithmetic itself lives in the un-vendored openai-whisper==20240930 package.
 */
#include "whisper_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ---------------------------------------------------------------- bf16 helpers */
float orc_bf16_round(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return x; /* NaN stays NaN */
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  memcpy(&x, &u, 4);
  return x;
}
void orc_bf16_round_array(float *x, long n) {
  for (long i = 0; i < n; ++i) x[i] = orc_bf16_round(x[i]);
}
/* IEEE binary16 round-to-nearest-even (F16C); overflow goes to inf exactly as the device's v_cvt_f16_f32 does */
float orc_f16_round(float x) { return _cvtsh_ss(_cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT)); }
/* policy 1: the engine's bfloat16 build, policy 2: its IEEE-half build (same storage points, other 16-bit type) */
static inline float round16(const orc_policy *p, float x) {
  return p->bf16_policy == 2 ? orc_f16_round(x) : orc_bf16_round(x);
}
static void maybe_round(const orc_policy *p, float *x, long n) {
  if (!p || !p->bf16_policy) return;
  if (p->bf16_policy == 2) {
    for (long i = 0; i < n; ++i) x[i] = orc_f16_round(x[i]);
  } else {
    orc_bf16_round_array(x, n);
  }
}

/* ---------------------------------------------------------------- front-end */

/* librosa.h:102-144 — Slaney scale, Slaney norm, fmin=0, fmax=8000 (ints), fp32 maths. */
void orc_mel_filterbank(int n_mels, float *out) {
  const int sr = 16000, n_fft = ORC_N_FFT, n_f = ORC_N_BINS;
  const int fmin = 0, fmax = 8000;
  const float f_min = 0.f, f_sp = 200.f / 3.f, min_log_hz = 1000.f;
  const float min_log_mel = (min_log_hz - f_min) / f_sp;
  const float logstep = logf(6.4f) / 27.f;
  float fft_freqs[ORC_N_BINS];
  for (int i = 0; i < n_f; ++i) fft_freqs[i] = ((float)i * sr) / n_fft;
  /* hz_to_mel takes an int hz (librosa.h:112) */
  float min_mel = (fmin - f_min) / f_sp;
  if (fmin >= min_log_hz) min_mel = min_log_mel + logf(fmin / min_log_hz) / logstep;
  float max_mel = (fmax - f_min) / f_sp;
  if (fmax >= min_log_hz) max_mel = min_log_mel + logf(fmax / min_log_hz) / logstep;
  int nm2 = n_mels + 2;
  float *mel_f = (float *)malloc(sizeof(float) * nm2);
  for (int i = 0; i < nm2; ++i) {
    /* Eigen LinSpaced(size, low, high): low + i*(high-low)/(size-1), last == high */
    float mel = (i == nm2 - 1) ? max_mel : min_mel + (float)i * ((max_mel - min_mel) / (float)(nm2 - 1));
    mel_f[i] = (mel > min_log_mel) ? expf((mel - min_log_mel) * logstep) * min_log_hz
                                   : mel * f_sp + f_min;
  }
  for (int m = 0; m < n_mels; ++m) {
    float fd0 = mel_f[m + 1] - mel_f[m];
    float fd1 = mel_f[m + 2] - mel_f[m + 1];
    float enorm = (float)(2.0 / (double)(mel_f[m + 2] - mel_f[m]));
    for (int k = 0; k < n_f; ++k) {
      float lower = -(mel_f[m] - fft_freqs[k]) / fd0;
      float upper = (mel_f[m + 2] - fft_freqs[k]) / fd1;
      float w = lower < upper ? lower : upper;
      if (w < 0.f) w = 0.f;
      out[m * n_f + k] = w * enorm;
    }
  }
  free(mel_f);
}

/* librosa.h:46-100,146-155 (pad/stft/spectrogram/mel GEMM) + Whisper.cpp:151-184. */
int orc_log_mel(const float *pcm, int n_samples, int n_mels, float *out, float *mmax_out) {
  const int n_fft = ORC_N_FFT, hop = ORC_HOP, n_f = ORC_N_BINS, pad = n_fft / 2;
  const int n_pad = n_samples + 2 * pad;
  const int n_frames = 1 + (n_pad - n_fft) / hop; /* librosa.h:87 */
  float *xp = (float *)malloc(sizeof(float) * (size_t)n_pad);
  /* reflect pad, librosa.h:50-57: left x[left-i]; right x[size-2-i+left] for i in [left, left+right) */
  memcpy(xp + pad, pcm, sizeof(float) * (size_t)n_samples);
  for (int i = 0; i < pad; ++i) xp[i] = pcm[pad - i];
  for (int i = pad; i < 2 * pad; ++i) xp[i + n_samples] = pcm[n_samples - 2 - i + pad];

  float window[ORC_N_FFT];
  for (int n = 0; n < n_fft; ++n) /* librosa.h:81 periodic Hann */
    window[n] = 0.5f * (1.f - cosf((float)n * 2.f * (float)M_PI / (float)n_fft));
  double *ctab = (double *)malloc(sizeof(double) * n_fft * 2);
  for (int i = 0; i < n_fft; ++i) {
    ctab[2 * i] = cos(2.0 * M_PI * i / n_fft);
    ctab[2 * i + 1] = sin(2.0 * M_PI * i / n_fft);
  }
  float *basis = (float *)malloc(sizeof(float) * (size_t)n_mels * n_f);
  orc_mel_filterbank(n_mels, basis);

  float *mel = (float *)malloc(sizeof(float) * (size_t)n_mels * n_frames); /* [n_mels][n_frames] */
#pragma omp parallel
  {
    float frame[ORC_N_FFT];
    float power[ORC_N_BINS];
#pragma omp for schedule(static)
    for (int f = 0; f < n_frames; ++f) {
      const float *seg = xp + (size_t)f * hop;
      for (int n = 0; n < n_fft; ++n) frame[n] = window[n] * seg[n]; /* librosa.h:92 */
      for (int k = 0; k < n_f; ++k) { /* 400-point DFT, bins 0..200 (librosa.h:93-95) */
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int n = 0; n < n_fft; ++n) {
          re += (double)frame[n] * ctab[2 * idx];
          im -= (double)frame[n] * ctab[2 * idx + 1];
          idx += k;
          if (idx >= n_fft) idx -= n_fft;
        }
        power[k] = (float)(re * re + im * im); /* librosa.h:98-100 with power = 2 */
      }
      for (int m = 0; m < n_mels; ++m) { /* librosa.h:153 */
        double acc = 0.0;
        const float *b = basis + (size_t)m * n_f;
        for (int k = 0; k < n_f; ++k) acc += (double)b[k] * (double)power[k];
        mel[(size_t)m * n_frames + f] = (float)acc;
      }
    }
  }
  /* Whisper.cpp:157-167: log10(max(.,1e-10)), global max over ALL frames (incl. frame 3000) */
  float mmax = -3.402823466e38f;
  for (size_t i = 0; i < (size_t)n_mels * n_frames; ++i) {
    float v = log10f(mel[i] > 1e-10f ? mel[i] : 1e-10f);
    mel[i] = v;
    if (v > mmax) mmax = v;
  }
  /* Whisper.cpp:169-181: max(., mmax-8) then (.+4)/4 in double, rows resized to 3000 (zero fill) */
  const float floor_v = (float)(mmax - 8.0);
  const int n_out = ORC_N_FRAMES_OUT;
  for (int m = 0; m < n_mels; ++m) {
    for (int n = 0; n < n_out; ++n) {
      float v = 0.f;
      if (n < n_frames) {
        float x = mel[(size_t)m * n_frames + n];
        v = (float)(((double)(x > floor_v ? x : floor_v) + 4.0) / 4.0);
      }
      out[(size_t)m * n_out + n] = v;
    }
  }
  if (mmax_out) *mmax_out = mmax;
  free(mel); free(basis); free(ctab); free(xp);
  return n_frames;
}

/* librosa.filters.mel(sr=16000, n_fft=400, n_mels) as upstream's assets/mel_filters.npz was generated (comment in
 * openai-whisper 20240930 whisper/audio.py mel_filters()): Slaney scale + Slaney norm like librosa.h:102-144, but the
 * ramps are evaluated in float64, narrowed to float32, then scaled by the float64 norm and narrowed again. */
static void mel_filterbank_librosa_f64(int n_mels, float *out) {
  const int n_f = ORC_N_BINS;
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
  const double max_mel = min_log_mel + log(8000.0 / min_log_hz) / logstep;
  double *mel_f = (double *)malloc(sizeof(double) * (size_t)(n_mels + 2));
  for (int i = 0; i < n_mels + 2; ++i) {
    const double mel = max_mel * (double)i / (double)(n_mels + 1);
    mel_f[i] = mel >= min_log_mel ? min_log_hz * exp(logstep * (mel - min_log_mel)) : f_sp * mel;
  }
  for (int m = 0; m < n_mels; ++m) {
    const double fd0 = mel_f[m + 1] - mel_f[m], fd1 = mel_f[m + 2] - mel_f[m + 1];
    const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
    for (int k = 0; k < n_f; ++k) {
      const double freq = (double)k * 16000.0 / (2.0 * (n_f - 1));
      const double lower = -(mel_f[m] - freq) / fd0, upper = (mel_f[m + 2] - freq) / fd1;
      const double w = lower < upper ? lower : upper;
      const float w32 = (float)(w > 0.0 ? w : 0.0);
      out[(size_t)m * n_f + k] = (float)((double)w32 * enorm);
    }
  }
  free(mel_f);
}

/* The fp32 ONNX / PyTorch lineage's front-end (SURVEY A.1 column 3; call sites model_convert/generate_data.py:162-176
 * and export_onnx.py:557; arithmetic = upstream openai-whisper 20240930 whisper/audio.py pad_or_trim +
 * log_mel_spectrogram, third-party and absent from the reference tree): audio zero-padded / trimmed to 30 s BEFORE
 * the STFT, torch.stft(400, 160, periodic Hann, center, reflect), the LAST frame dropped, |X|^2, mel_filters.npz,
 * log10(clamp 1e-10), max over the 3000 kept frames, max(., gmax - 8), (. + 4) / 4. No zero fill: the padded region
 * holds the clamp floor. out: [n_mels][3000]. Pinned by tests/test_oracle_frontend.py against torch.stft. */
void orc_log_mel_openai(const float *pcm, int n_samples, int n_mels, float *out, float *mmax_out) {
  const int n_fft = ORC_N_FFT, hop = ORC_HOP, n_f = ORC_N_BINS, pad = n_fft / 2, N = 480000, n_frames = ORC_N_FRAMES_OUT;
  float *xp = (float *)calloc((size_t)N + 2 * pad, sizeof(float));
  memcpy(xp + pad, pcm, sizeof(float) * (size_t)(n_samples < N ? n_samples : N));
  for (int i = 0; i < pad; ++i) xp[i] = xp[2 * pad - i];                       /* reflect: x[pad - i] of the padded clip */
  for (int i = 0; i < pad; ++i) xp[pad + N + i] = xp[pad + N - 2 - i];
  float window[ORC_N_FFT];
  for (int n = 0; n < n_fft; ++n) window[n] = (float)(0.5 * (1.0 - cos(2.0 * M_PI * n / n_fft)));  /* torch.hann_window */
  double *ctab = (double *)malloc(sizeof(double) * n_fft * 2);
  for (int i = 0; i < n_fft; ++i) { ctab[2 * i] = cos(2.0 * M_PI * i / n_fft); ctab[2 * i + 1] = sin(2.0 * M_PI * i / n_fft); }
  float *basis = (float *)malloc(sizeof(float) * (size_t)n_mels * n_f);
  mel_filterbank_librosa_f64(n_mels, basis);
  float *mel = (float *)malloc(sizeof(float) * (size_t)n_mels * n_frames);
#pragma omp parallel
  {
    float frame[ORC_N_FFT];
    float power[ORC_N_BINS];
#pragma omp for schedule(static)
    for (int f = 0; f < n_frames; ++f) {  /* frame 3000 of torch.stft's 3001 is dropped (stft[..., :-1]) */
      const float *seg = xp + (size_t)f * hop;
      for (int n = 0; n < n_fft; ++n) frame[n] = window[n] * seg[n];
      for (int k = 0; k < n_f; ++k) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int n = 0; n < n_fft; ++n) {
          re += (double)frame[n] * ctab[2 * idx];
          im -= (double)frame[n] * ctab[2 * idx + 1];
          idx += k;
          if (idx >= n_fft) idx -= n_fft;
        }
        power[k] = (float)(re * re + im * im);
      }
      for (int m = 0; m < n_mels; ++m) {
        double acc = 0.0;
        const float *b = basis + (size_t)m * n_f;
        for (int k = 0; k < n_f; ++k) acc += (double)b[k] * (double)power[k];
        mel[(size_t)m * n_frames + f] = (float)acc;
      }
    }
  }
  float mmax = -3.402823466e38f;
  for (size_t i = 0; i < (size_t)n_mels * n_frames; ++i) {
    const float v = log10f(mel[i] > 1e-10f ? mel[i] : 1e-10f);
    mel[i] = v;
    if (v > mmax) mmax = v;
  }
  const float floor_v = mmax - 8.0f;
  for (size_t i = 0; i < (size_t)n_mels * n_frames; ++i) out[i] = ((mel[i] > floor_v ? mel[i] : floor_v) + 4.0f) / 4.0f;
  if (mmax_out) *mmax_out = mmax;
  free(mel); free(basis); free(ctab); free(xp);
}

/* upstream whisper/model.py sinusoids() [openai-whisper 20240930]. */
void orc_sinusoids(int length, int channels, float *out) {
  int half = channels / 2;
  float inc = logf(10000.f) / (float)(half - 1);
  for (int t = 0; t < length; ++t)
    for (int c = 0; c < half; ++c) {
      float inv = expf(-inc * (float)c);
      float st = (float)t * inv;
      out[(size_t)t * channels + c] = sinf(st);
      out[(size_t)t * channels + half + c] = cosf(st);
    }
}

/* ---------------------------------------------------------------- dense primitives */

/* C[M,N] (ldc) = A[M,K] (lda) * Bt[K,N] (+ bias[N]); Bt row-major [K][N]. */
static void gemm_nn(int M, int N, int K, const float *A, int lda, const float *Bt, int ldb,
                    const float *bias, float *C, int ldc) {
  enum { RB = 4, CB = 32 };
#pragma omp parallel for schedule(static)
  for (int i0 = 0; i0 < M; i0 += RB) {
    int rb = M - i0 < RB ? M - i0 : RB;
    for (int j0 = 0; j0 < N; j0 += CB) {
      int cb = N - j0 < CB ? N - j0 : CB;
      float acc[RB][CB];
      for (int r = 0; r < RB; ++r)
        for (int c = 0; c < CB; ++c) acc[r][c] = 0.f;
      if (rb == RB && cb == CB) {
        for (int k = 0; k < K; ++k) {
          const float *b = Bt + (size_t)k * ldb + j0;
          float a0 = A[(size_t)(i0 + 0) * lda + k], a1 = A[(size_t)(i0 + 1) * lda + k];
          float a2 = A[(size_t)(i0 + 2) * lda + k], a3 = A[(size_t)(i0 + 3) * lda + k];
          for (int c = 0; c < CB; ++c) {
            float bv = b[c];
            acc[0][c] += a0 * bv; acc[1][c] += a1 * bv;
            acc[2][c] += a2 * bv; acc[3][c] += a3 * bv;
          }
        }
      } else {
        for (int k = 0; k < K; ++k) {
          const float *b = Bt + (size_t)k * ldb + j0;
          for (int r = 0; r < rb; ++r) {
            float a = A[(size_t)(i0 + r) * lda + k];
            for (int c = 0; c < cb; ++c) acc[r][c] += a * b[c];
          }
        }
      }
      for (int r = 0; r < rb; ++r)
        for (int c = 0; c < cb; ++c)
          C[(size_t)(i0 + r) * ldc + j0 + c] = acc[r][c] + (bias ? bias[j0 + c] : 0.f);
    }
  }
}

/* transposed-weight cache: nn.Linear weights arrive as W[N][K]; gemm_nn wants [K][N]. */
typedef struct { const float *src; int n, k; float *t; } tcache_ent;
static tcache_ent g_tc[1024];
static int g_ntc = 0;
static const float *transposed(const float *W, int N, int K) {
  for (int i = 0; i < g_ntc; ++i)
    if (g_tc[i].src == W && g_tc[i].n == N && g_tc[i].k == K) return g_tc[i].t;
  float *t = (float *)malloc(sizeof(float) * (size_t)N * K);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) t[(size_t)k * N + n] = W[(size_t)n * K + k];
  if (g_ntc < 1024) { g_tc[g_ntc].src = W; g_tc[g_ntc].n = N; g_tc[g_ntc].k = K; g_tc[g_ntc].t = t; ++g_ntc; }
  return t;
}
void orc_clear_cache(void) {
  for (int i = 0; i < g_ntc; ++i) free(g_tc[i].t);
  g_ntc = 0;
}

/* y[M,N] = x[M,K] W[N,K]^T + b  (nn.Linear) */
static void linear(int M, int N, int K, const float *x, const float *W, const float *b, float *y) {
  if (M >= 8) {
    gemm_nn(M, N, K, x, K, transposed(W, N, K), N, b, y, N);
  } else {
    for (int i = 0; i < M; ++i) {
#pragma omp parallel for schedule(static)
      for (int n = 0; n < N; ++n) {
        const float *w = W + (size_t)n * K;
        const float *xi = x + (size_t)i * K;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int k = 0;
        for (; k + 8 <= K; k += 8)
          for (int u = 0; u < 8; ++u) acc[u] += xi[k + u] * w[k + u];
        float s = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
        for (; k < K; ++k) s += xi[k] * w[k];
        y[(size_t)i * N + n] = s + (b ? b[n] : 0.f);
      }
    }
  }
}

/* nn.LayerNorm, eps 1e-5, biased variance [upstream]. */
static void layer_norm(int M, int D, const float *x, const float *g, const float *b, float *y) {
#pragma omp parallel for schedule(static) if (M > 16)
  for (int i = 0; i < M; ++i) {
    const float *xi = x + (size_t)i * D;
    double s = 0.0;
    for (int d = 0; d < D; ++d) s += xi[d];
    double mean = s / D, v = 0.0;
    for (int d = 0; d < D; ++d) { double t = xi[d] - mean; v += t * t; }
    float rstd = (float)(1.0 / sqrt(v / D + 1e-5));
    float fm = (float)mean;
    for (int d = 0; d < D; ++d) y[(size_t)i * D + d] = (xi[d] - fm) * rstd * g[d] + b[d];
  }
}

/* exact-erf GELU (nn.GELU() default / F.gelu) */
static inline float gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

/* ---------------------------------------------------------------- encoder */

/* Conv1d(k=3, padding=1, stride s) on x[Cin][Tin] -> y[Tout][Cout] (time-major out), + GELU.
 * export_onnx.py:158-159 (F.gelu(self.conv1(x)), F.gelu(self.conv2(x))). Implemented as an
 * im2col GEMM: row t = [x[:, s*t-1], x[:, s*t], x[:, s*t+1]] flattened as [k][c]. */
static void conv1d_k3_gelu_tm(const float *x_tm /* [Tin][Cin] time-major */, int Tin, int Cin,
                              const float *W /* [Cout][Cin][3] */, const float *b, int Cout,
                              int stride, float *y_tm /* [Tout][Cout] */) {
  int Tout = (Tin + 2 - 3) / stride + 1;
  int K = 3 * Cin;
  float *col = (float *)calloc((size_t)Tout * K, sizeof(float));
  for (int t = 0; t < Tout; ++t)
    for (int kk = 0; kk < 3; ++kk) {
      int ti = stride * t + kk - 1;
      if (ti < 0 || ti >= Tin) continue;
      memcpy(col + (size_t)t * K + (size_t)kk * Cin, x_tm + (size_t)ti * Cin, sizeof(float) * Cin);
    }
  /* Wt[k*Cin + c][n] = W[n][c][k] */
  float *Wt = (float *)malloc(sizeof(float) * (size_t)K * Cout);
  for (int n = 0; n < Cout; ++n)
    for (int c = 0; c < Cin; ++c)
      for (int kk = 0; kk < 3; ++kk)
        Wt[((size_t)kk * Cin + c) * Cout + n] = W[((size_t)n * Cin + c) * 3 + kk];
  gemm_nn(Tout, Cout, K, col, K, Wt, Cout, b, y_tm, Cout);
  for (size_t i = 0; i < (size_t)Tout * Cout; ++i) y_tm[i] = gelu(y_tm[i]);
  free(Wt); free(col);
}

/* upstream qkv_attention without mask (encoder self-attn), SDPA disabled (export_onnx.py:714):
 * softmax_fp32((q*s)(k*s)^T) v, s = head_dim^-0.25. q,k,v: [T][d]; out: [T][d]. */
static void attention_full(const orc_policy *p, int T, int d, int H, const float *q, const float *k,
                           const float *v, float *out) {
  int hd = d / H;
  float scale = powf((float)hd, -0.25f);
  float *qs = (float *)malloc(sizeof(float) * (size_t)T * hd);
  float *kst = (float *)malloc(sizeof(float) * (size_t)T * hd); /* [hd][T] */
  float *vh = (float *)malloc(sizeof(float) * (size_t)T * hd);
  float *S = (float *)malloc(sizeof(float) * (size_t)T * T);
  float *oh = (float *)malloc(sizeof(float) * (size_t)T * hd);
  for (int h = 0; h < H; ++h) {
    for (int t = 0; t < T; ++t)
      for (int c = 0; c < hd; ++c) {
        qs[(size_t)t * hd + c] = q[(size_t)t * d + h * hd + c] * scale;
        kst[(size_t)c * T + t] = k[(size_t)t * d + h * hd + c] * scale;
        vh[(size_t)t * hd + c] = v[(size_t)t * d + h * hd + c];
      }
    gemm_nn(T, T, hd, qs, hd, kst, T, NULL, S, T);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < T; ++i) {
      float *s = S + (size_t)i * T;
      float mx = s[0];
      for (int j = 1; j < T; ++j) if (s[j] > mx) mx = s[j];
      double sum = 0.0;
      for (int j = 0; j < T; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
      float inv = (float)(1.0 / sum);
      if (p && p->bf16_policy) {
        /* engine: P is narrowed to bf16 for the PV MFMA, the row sum stays fp32 */
        for (int j = 0; j < T; ++j) s[j] = round16(p, s[j]) * inv;
      } else {
        for (int j = 0; j < T; ++j) s[j] *= inv;
      }
    }
    gemm_nn(T, hd, T, S, T, vh, hd, NULL, oh, hd);
    for (int t = 0; t < T; ++t)
      memcpy(out + (size_t)t * d + h * hd, oh + (size_t)t * hd, sizeof(float) * hd);
  }
  free(oh); free(S); free(vh); free(kst); free(qs);
}

/* export_onnx.py:153-184 (modified_audio_encoder_forward) + :193-213 (cross K/V heads). */
void orc_encoder(const orc_model *m, const orc_policy *p, const float *mel, float *cross_k,
                 float *cross_v) {
  const int d = m->n_audio_state, T0 = ORC_N_FRAMES_OUT, T = m->n_audio_ctx, H = m->n_audio_head;
  const int nm = m->n_mels;
  /* mel [n_mels][3000] -> time-major */
  float *mel_tm = (float *)malloc(sizeof(float) * (size_t)T0 * nm);
  for (int c = 0; c < nm; ++c)
    for (int t = 0; t < T0; ++t) mel_tm[(size_t)t * nm + c] = mel[(size_t)c * T0 + t];
  maybe_round(p, mel_tm, (long)T0 * nm);
  float *h1 = (float *)malloc(sizeof(float) * (size_t)T0 * d);
  conv1d_k3_gelu_tm(mel_tm, T0, nm, m->conv1_w, m->conv1_b, d, 1, h1);
  maybe_round(p, h1, (long)T0 * d);
  float *x = (float *)malloc(sizeof(float) * (size_t)T * d);
  conv1d_k3_gelu_tm(h1, T0, d, m->conv2_w, m->conv2_b, d, 2, x);
  /* x.permute(0,2,1) + positional_embedding[:T]  (export_onnx.py:160-176) */
  for (size_t i = 0; i < (size_t)T * d; ++i) x[i] += m->enc_pos[i];
  free(h1); free(mel_tm);

  float *ln = (float *)malloc(sizeof(float) * (size_t)T * d);
  float *q = (float *)malloc(sizeof(float) * (size_t)T * d);
  float *k = (float *)malloc(sizeof(float) * (size_t)T * d);
  float *v = (float *)malloc(sizeof(float) * (size_t)T * d);
  float *a = (float *)malloc(sizeof(float) * (size_t)T * d);
  float *t1 = (float *)malloc(sizeof(float) * (size_t)T * d);
  float *hid = (float *)malloc(sizeof(float) * (size_t)T * 4 * d);
  for (int l = 0; l < m->n_audio_layer; ++l) {
    const orc_block *b = &m->enc[l];
    /* x = x + attn(attn_ln(x))  [upstream ResidualAttentionBlock.forward] */
    layer_norm(T, d, x, b->attn_ln_w, b->attn_ln_b, ln);
    maybe_round(p, ln, (long)T * d);
    linear(T, d, d, ln, b->q_w, b->q_b, q);
    linear(T, d, d, ln, b->k_w, NULL, k); /* key: bias=False [upstream] */
    linear(T, d, d, ln, b->v_w, b->v_b, v);
    maybe_round(p, q, (long)T * d); maybe_round(p, k, (long)T * d); maybe_round(p, v, (long)T * d);
    attention_full(p, T, d, H, q, k, v, a);
    maybe_round(p, a, (long)T * d);
    linear(T, d, d, a, b->o_w, b->o_b, t1);
    for (size_t i = 0; i < (size_t)T * d; ++i) x[i] += t1[i];
    /* x = x + mlp(mlp_ln(x)) */
    layer_norm(T, d, x, b->mlp_ln_w, b->mlp_ln_b, ln);
    maybe_round(p, ln, (long)T * d);
    linear(T, 4 * d, d, ln, b->fc1_w, b->fc1_b, hid);
    for (size_t i = 0; i < (size_t)T * 4 * d; ++i) hid[i] = gelu(hid[i]);
    maybe_round(p, hid, (long)T * 4 * d);
    linear(T, d, 4 * d, hid, b->fc2_w, b->fc2_b, t1);
    for (size_t i = 0; i < (size_t)T * d; ++i) x[i] += t1[i];
  }
  layer_norm(T, d, x, m->ln_post_w, m->ln_post_b, ln); /* export_onnx.py:183 */
  maybe_round(p, ln, (long)T * d);
  const int dt = m->n_text_state;
  for (int l = 0; l < m->n_text_layer; ++l) { /* export_onnx.py:205-210 */
    const orc_block *b = &m->dec[l];
    linear(T, dt, d, ln, b->ck_w, NULL, cross_k + (size_t)l * T * dt);
    linear(T, dt, d, ln, b->cv_w, b->cv_b, cross_v + (size_t)l * T * dt);
  }
  maybe_round(p, cross_k, (long)m->n_text_layer * T * dt);
  maybe_round(p, cross_v, (long)m->n_text_layer * T * dt);
  free(hid); free(t1); free(a); free(v); free(k); free(q); free(ln); free(x);
}

/* ---------------------------------------------------------------- decoder */

int orc_argmax(const float *x, int n) { /* Whisper.cpp:42-45, std::max_element: first max wins */
  int best = 0;
  for (int i = 1; i < n; ++i) if (x[i] > x[best]) best = i;
  return best;
}

/* single-query attention over n_keys rows of K,V [n][d] (per head), fp32 softmax.
 * Used for cross-attn (export_onnx.py:221-230, 1500 keys, no mask) and for self-attn, where
 * export_onnx.py:103-147 attends over the 448-row cache with rows >= offset filled with
 * -60000 plus a separate column for the current token: with exp underflow that equals
 * attention over rows 0..offset-1 of the cache and the current k1/v1; here the
 * current token has already been appended at row `offset`, so n_keys = offset + 1. */
static void attend_one(int n_keys, int d, int H, const float *q, const float *K, const float *V,
                       float *out) {
  int hd = d / H;
  float scale = powf((float)hd, -0.25f);
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; ++h) {
    float *s = (float *)malloc(sizeof(float) * (size_t)n_keys);
    float qs[256];
    for (int c = 0; c < hd; ++c) qs[c] = q[h * hd + c] * scale;
    float mx = -3.402823466e38f;
    for (int j = 0; j < n_keys; ++j) {
      const float *kj = K + (size_t)j * d + h * hd;
      float acc = 0.f;
      for (int c = 0; c < hd; ++c) acc += qs[c] * (kj[c] * scale);
      s[j] = acc;
      if (acc > mx) mx = acc;
    }
    double sum = 0.0;
    for (int j = 0; j < n_keys; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
    float inv = (float)(1.0 / sum);
    float o[256];
    for (int c = 0; c < hd; ++c) o[c] = 0.f;
    for (int j = 0; j < n_keys; ++j) {
      const float *vj = V + (size_t)j * d + h * hd;
      float w = s[j] * inv;
      for (int c = 0; c < hd; ++c) o[c] += w * vj[c];
    }
    for (int c = 0; c < hd; ++c) out[h * hd + c] = o[c];
    free(s);
  }
}

/* modified_self_qkv_attention as written (export_onnx.py:103-147), one head at a time: qk over the whole n_ctx-row
 * cache (k_cache * scale, q * scale), masked_fill_(mask, -60000) with mask[j] = (j >= offset) (causal_mask_1d,
 * export_onnx.py:59-68 / Whisper.cpp:253-258), qk1 against the current token's k1, softmax over the n_ctx + 1
 * concatenated values in fp32, out = w @ v_cache + w1 @ v1. The cache rows >= offset are whatever the host left
 * there (zeros after the reset of Whisper.cpp:204-205); they are multiplied by weights that underflow to 0. */
static void attend_self_literal(int n_ctx, int offset, int d, int H, const float *q, const float *Kc, const float *Vc,
                                const float *k1, const float *v1, float *out) {
  int hd = d / H;
  float scale = powf((float)hd, -0.25f);
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; ++h) {
    float *s = (float *)malloc(sizeof(float) * (size_t)(n_ctx + 1));
    float qs[256];
    for (int c = 0; c < hd; ++c) qs[c] = q[h * hd + c] * scale;
    for (int j = 0; j <= n_ctx; ++j) {
      const float *kj = j < n_ctx ? Kc + (size_t)j * d + h * hd : k1 + h * hd;
      float acc = 0.f;
      for (int c = 0; c < hd; ++c) acc += qs[c] * (kj[c] * scale);
      s[j] = (j < n_ctx && j >= offset) ? -60000.f : acc; /* masked_fill_ (:130) */
    }
    float mx = -3.402823466e38f;
    for (int j = 0; j <= n_ctx; ++j) if (s[j] > mx) mx = s[j];
    double sum = 0.0;
    for (int j = 0; j <= n_ctx; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
    float inv = (float)(1.0 / sum);
    float o[256];
    for (int c = 0; c < hd; ++c) o[c] = 0.f;
    for (int j = 0; j <= n_ctx; ++j) { /* w @ v_cache, then + w1 @ v1 (:139-141) */
      const float *vj = j < n_ctx ? Vc + (size_t)j * d + h * hd : v1 + h * hd;
      float w = s[j] * inv;
      for (int c = 0; c < hd; ++c) o[c] += w * vj[c];
    }
    for (int c = 0; c < hd; ++c) out[h * hd + c] = o[c];
    free(s);
  }
}

/* export_onnx.py:312-387 for one token + the cache append of Whisper.cpp:328-342. */
void orc_decoder_step(const orc_model *m, const orc_policy *p, int token, int offset,
                      const float *cross_k, const float *cross_v, float *self_k, float *self_v,
                      float *logits) {
  const int d = m->n_text_state, H = m->n_text_head, Ta = m->n_audio_ctx, Tc = m->n_text_ctx;
  float *x = (float *)malloc(sizeof(float) * d * 8);
  float *ln = x + d, *q = x + 2 * d, *a = x + 3 * d, *t1 = x + 4 * d;
  float *hid = (float *)malloc(sizeof(float) * 4 * d);
  /* token_embedding(tokens) + positional_embedding[offset]  (export_onnx.py:334-336) */
  for (int c = 0; c < d; ++c) x[c] = m->tok_emb[(size_t)token * d + c] + m->dec_pos[(size_t)offset * d + c];
  for (int l = 0; l < m->n_text_layer; ++l) {
    const orc_block *b = &m->dec[l];
    float *Kc = self_k + (size_t)l * Tc * d, *Vc = self_v + (size_t)l * Tc * d;
    /* self-attn (export_onnx.py:238-261, 103-147) */
    layer_norm(1, d, x, b->attn_ln_w, b->attn_ln_b, ln);
    linear(1, d, d, ln, b->q_w, b->q_b, q);
    if (p && p->literal_mask) {
      /* the graph as exported: k, v of this token stay outside the cache while it is attended over (export_onnx.py:
       * 245-261), the host appends them afterwards (Whisper.cpp:328-342) */
      float *k1 = (float *)malloc(sizeof(float) * 2 * d), *v1 = k1 + d;
      linear(1, d, d, ln, b->k_w, NULL, k1);
      linear(1, d, d, ln, b->v_w, b->v_b, v1);
      maybe_round(p, k1, d);
      maybe_round(p, v1, d);
      attend_self_literal(Tc, offset, d, H, q, Kc, Vc, k1, v1, a);
      memcpy(Kc + (size_t)offset * d, k1, sizeof(float) * d);
      memcpy(Vc + (size_t)offset * d, v1, sizeof(float) * d);
      free(k1);
    } else {
      linear(1, d, d, ln, b->k_w, NULL, Kc + (size_t)offset * d);
      linear(1, d, d, ln, b->v_w, b->v_b, Vc + (size_t)offset * d);
      maybe_round(p, Kc + (size_t)offset * d, d); /* engine keeps the self-KV cache in bf16 */
      maybe_round(p, Vc + (size_t)offset * d, d);
      attend_one(offset + 1, d, H, q, Kc, Vc, a);
    }
    linear(1, d, d, a, b->o_w, b->o_b, t1);
    for (int c = 0; c < d; ++c) x[c] += t1[c];
    /* cross-attn (export_onnx.py:221-230, 292-295) */
    layer_norm(1, d, x, b->cross_ln_w, b->cross_ln_b, ln);
    linear(1, d, d, ln, b->cq_w, b->cq_b, q);
    attend_one(Ta, d, H, q, cross_k + (size_t)l * Ta * d, cross_v + (size_t)l * Ta * d, a);
    linear(1, d, d, a, b->co_w, b->co_b, t1);
    for (int c = 0; c < d; ++c) x[c] += t1[c];
    /* mlp (export_onnx.py:298) */
    layer_norm(1, d, x, b->mlp_ln_w, b->mlp_ln_b, ln);
    linear(1, 4 * d, d, ln, b->fc1_w, b->fc1_b, hid);
    for (int c = 0; c < 4 * d; ++c) hid[c] = gelu(hid[c]);
    linear(1, d, 4 * d, hid, b->fc2_w, b->fc2_b, t1);
    for (int c = 0; c < d; ++c) x[c] += t1[c];
  }
  if (logits) {
    layer_norm(1, d, x, m->dec_ln_w, m->dec_ln_b, ln); /* export_onnx.py:364 */
    linear(1, m->n_vocab, d, ln, m->tok_emb, NULL, logits); /* tied embedding, :378-385 */
  }
  free(hid); free(x);
}

/* Whisper.cpp:207-222 (and generate_data.py:193-246). */
int orc_greedy(const orc_model *m, const orc_policy *p, const float *cross_k, const float *cross_v,
               const int sot_seq[4], int eot, int max_new, const int *forced, int n_forced,
               int *out_tokens, float *step_logits) {
  const int d = m->n_text_state, Tc = m->n_text_ctx, L = m->n_text_layer;
  float *self_k = (float *)calloc((size_t)L * Tc * d, sizeof(float)); /* Whisper.cpp:204-205 */
  float *self_v = (float *)calloc((size_t)L * Tc * d, sizeof(float));
  float *logits = (float *)malloc(sizeof(float) * (size_t)m->n_vocab);
  int offset = 0, idx = 0, n = 0, nlog = 0;
  for (int i = 0; i < 4; ++i) { /* Whisper.cpp:214-217 */
    orc_decoder_step(m, p, sot_seq[i], offset, cross_k, cross_v, self_k, self_v, i == 3 ? logits : NULL);
    offset++;
  }
  idx = orc_argmax(logits, m->n_vocab);
  if (step_logits) memcpy(step_logits + (size_t)(nlog++) * m->n_vocab, logits, sizeof(float) * m->n_vocab);
  /* teacher-forced runs ignore eot and run exactly n_forced steps */
  while ((forced ? n < n_forced : idx != eot) && offset < Tc && n < max_new) { /* Whisper.cpp:219-222 */
    out_tokens[n] = idx;
    int feed = (forced && n < n_forced) ? forced[n] : idx;
    n++;
    orc_decoder_step(m, p, feed, offset++, cross_k, cross_v, self_k, self_v, logits);
    idx = orc_argmax(logits, m->n_vocab);
    if (step_logits) memcpy(step_logits + (size_t)(nlog++) * m->n_vocab, logits, sizeof(float) * m->n_vocab);
  }
  free(logits); free(self_v); free(self_k);
  return n;
}

int orc_transcribe(const orc_model *m, const orc_policy *p, const float *pcm, int n_samples,
                   const int sot_seq[4], int eot, int max_new, int *out_tokens) {
  float *mel = (float *)malloc(sizeof(float) * (size_t)m->n_mels * ORC_N_FRAMES_OUT);
  orc_log_mel(pcm, n_samples, m->n_mels, mel, NULL);
  size_t ckv = (size_t)m->n_text_layer * m->n_audio_ctx * m->n_text_state;
  float *ck = (float *)malloc(sizeof(float) * ckv), *cv = (float *)malloc(sizeof(float) * ckv);
  orc_encoder(m, p, mel, ck, cv);
  int n = orc_greedy(m, p, ck, cv, sot_seq, eot, max_new, NULL, 0, out_tokens, NULL);
  free(cv); free(ck); free(mel);
  return n;
}
