/*
 * whisper_oracle.h — CPU restatement of the whisper.axera hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This library is the parity checker for the HIP engine. It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * What it restates (all paths relative to the reference tree):
 *   - log-mel front-end: cpp/src/librosa/librosa.h:46-155 + cpp/src/Whisper.cpp:151-184
 *   - encoder graph:     model_convert/export_onnx.py:153-213 (+ upstream openai-whisper
 *                        20240930 whisper/model.py for Linear/Conv1d/LayerNorm/GELU wiring)
 *   - decoder step:      model_convert/export_onnx.py:103-150, 216-387
 *   - greedy loop:       cpp/src/Whisper.cpp:186-222, 290-346
 *
 * Pinning: the front-end is pinned against the reference's own C++ (oracle/_ref, compiled
 * from /root/reference) on demo.wav and seeded clips; the encoder/decoder arithmetic lives
 * in the un-vendored third-party package openai-whisper==20240930 (model_convert/
 * requirements.txt:1), so it is pinned against goldens generated here with an independent
 * implementation of the same architecture (transformers.WhisperForConditionalGeneration)
 * on seeded weights: tests/golden/make_model_goldens.py. The reference holds no golden
 * vectors of its own for the model arithmetic.
 */
#ifndef WHISPER_ORACLE_H_
#define WHISPER_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_N_FFT 400
#define ORC_HOP 160
#define ORC_N_BINS 201
#define ORC_N_FRAMES_OUT 3000
#define ORC_N_AUDIO_CTX 1500

/* Rounding policy: where the GPU engine narrows to bf16, the oracle can narrow at the
 * same points so that the only remaining differences are fp32 summation order and
 * libm-vs-device transcendental ulps. 0 = pure fp32 everywhere (the reference's ONNX fp32
 * lineage), 1 = mirror the engine's bf16 storage points, 2 = the same storage points in IEEE
 * half (the engine's fp16 build: BASELINE configs[3] "Whisper-turbo fp16"). */
typedef struct {
  int bf16_policy;
  /* 1: decoder self-attention exactly as the exported graph states it (export_onnx.py:124-137): scores against ALL
   * 448 cache rows, rows >= offset filled with -60000 (the int mask of Whisper.cpp:253-258), a separate score
   * column for the current token's k1, one fp32 softmax over the 449 values, w @ v_cache + w1 @ v1, and the cache
   * append done by the host afterwards (Whisper.cpp:328-342). 0: the equivalent causal form (keys 0..offset, the
   * current token appended first) that the engine's kernels implement. tests/test_oracle_model.py holds the two
   * against each other bit for bit. */
  int literal_mask;
} orc_policy;

typedef struct {
  /* attention */
  const float *attn_ln_w, *attn_ln_b;
  const float *q_w, *q_b, *k_w, *v_w, *v_b, *o_w, *o_b;
  /* cross attention (decoder only; NULL in encoder blocks) */
  const float *cross_ln_w, *cross_ln_b;
  const float *cq_w, *cq_b, *ck_w, *cv_w, *cv_b, *co_w, *co_b;
  /* mlp */
  const float *mlp_ln_w, *mlp_ln_b;
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} orc_block;

typedef struct {
  int n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
  const float *conv1_w, *conv1_b; /* [d, n_mels, 3] */
  const float *conv2_w, *conv2_b; /* [d, d, 3] */
  const float *enc_pos;           /* [n_audio_ctx, d] */
  const float *ln_post_w, *ln_post_b;
  const float *tok_emb;           /* [n_vocab, d] */
  const float *dec_pos;           /* [n_text_ctx, d] */
  const float *dec_ln_w, *dec_ln_b;
  orc_block *enc;                 /* [n_audio_layer] */
  orc_block *dec;                 /* [n_text_layer] */
} orc_model;

/* Slaney mel filterbank exactly as librosa.h:102-144 builds it (fp32 arithmetic).
 * out: [n_mels, 201] row-major. */
void orc_mel_filterbank(int n_mels, float *out);

/* Whisper::preprocess (Whisper.cpp:151-184): reflect pad, periodic Hann, 400-pt DFT,
 * |X|^2, mel, log10/clamp/scale, zero-pad or truncate to 3000 frames.
 * out: [n_mels, 3000] row-major. Returns the number of real frames (before truncation),
 * writes the global max (log10 power) to *mmax if non-NULL. */
int orc_log_mel(const float *pcm, int n_samples, int n_mels, float *out, float *mmax);

/* feature_mode "openai" (SURVEY A.1 column 3; model_convert/generate_data.py:162-176): pad/trim to 30 s, drop the
 * last STFT frame, clamp floor instead of zeros in the padded region. out: [n_mels, 3000]. */
void orc_log_mel_openai(const float *pcm, int n_samples, int n_mels, float *out, float *mmax);

/* upstream sinusoids(length, channels) — encoder positional embedding. */
void orc_sinusoids(int length, int channels, float *out);

/* Encoder + cross-KV projection (export_onnx.py:193-213).
 * mel: [n_mels, 3000]; cross_k, cross_v: [n_text_layer, n_audio_ctx, d] each. */
void orc_encoder(const orc_model *m, const orc_policy *p, const float *mel,
                 float *cross_k, float *cross_v);

/* One decoder step (export_onnx.py:312-387) INCLUDING the host-side cache append of
 * Whisper.cpp:328-342. self_k/self_v: [n_text_layer, n_text_ctx, d], rows < offset valid.
 * logits: [n_vocab] or NULL to skip the vocabulary projection. */
void orc_decoder_step(const orc_model *m, const orc_policy *p, int token, int offset,
                      const float *cross_k, const float *cross_v,
                      float *self_k, float *self_v, float *logits);

/* first-max-wins argmax (Whisper.cpp:42-45). */
int orc_argmax(const float *x, int n);

/* Greedy loop of Whisper::run (Whisper.cpp:207-222): 4 SOT steps then until eot or ctx.
 * If forced != NULL, generated token i is replaced by forced[i] for feeding back
 * (teacher forcing; n_forced entries) while the argmax ids are still recorded.
 * out_tokens: capacity n_text_ctx. step_logits (optional): [max_steps, n_vocab] logits of
 * every step from the 4th SOT step on. Returns the number of generated ids (eot excluded). */
int orc_greedy(const orc_model *m, const orc_policy *p, const float *cross_k,
               const float *cross_v, const int sot_seq[4], int eot, int max_new,
               const int *forced, int n_forced, int *out_tokens, float *step_logits);

/* Full path: PCM -> ids. Returns number of ids. */
int orc_transcribe(const orc_model *m, const orc_policy *p, const float *pcm, int n_samples,
                   const int sot_seq[4], int eot, int max_new, int *out_tokens);

/* bf16 round-to-nearest-even of an fp32 value (returned as fp32). */
float orc_bf16_round(float x);
float orc_f16_round(float x);
void orc_bf16_round_array(float *x, long n);

void orc_set_threads(int n);
void orc_clear_cache(void);

#ifdef __cplusplus
}
#endif
#endif
