// common.hpp — shared device helpers and the launch interface of the HIP kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// ------------------------------------------------------------------ the 16-bit storage / MFMA operand type
// Every kernel file and the engine are compiled TWICE (Makefile): -DAXW_F16=0 -> bfloat16 (the default; what
// BASELINE configs[1], [2], [4] name), -DAXW_F16=1 -> IEEE half (configs[3]: "Whisper-turbo fp16"; also the dtype
// OpenAI's and HF's checkpoints are published in, so their weights load without a rounding step). Everything that
// depends on the type lives in the inline namespace axw::bf / axw::hf — same source, two sets of symbols — and the C ABI
// (api.cpp) picks one per model from the dtype of its weights file. `h16` is that type: weights, MFMA operands,
// K/V caches and the (hi, lo) activation pairs of the batched decoder. Accumulation, LayerNorm, softmax, GELU, the
// residual stream and the logits are fp32 in both builds.
#ifndef AXW_F16
#define AXW_F16 0
#endif
#if AXW_F16
#define AXW_NS hf
#else
#define AXW_NS bf
#endif

namespace axw {
inline namespace AXW_NS {

#if AXW_F16
typedef _Float16 h16;
constexpr const char* kDtypeName = "fp16";
#else
typedef __bf16 h16;
constexpr const char* kDtypeName = "bf16";
#endif
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // 16-byte staging register (HIP's uint4 struct can end up in scratch)
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kNFFT = 400;
constexpr int kHop = 160;
constexpr int kBins = 201;
constexpr int kFramesOut = 3000;   // Whisper.cpp:172 resize(3000)
constexpr int kHeadDim = 64;       // all Whisper sizes
constexpr int kKeyBlk = 64;        // keys per block of the decode K layout

// ------------------------------------------------------------------ device helpers
#ifdef __HIPCC__
// the two h16 values packed in one dword (element 2i in the low half), widened to fp32: one shift / one mask for
// bfloat16; v_cvt_f32_f16 (folded into v_fma_mix_f32 where the value feeds an FMA) for half
#if AXW_F16
typedef _Float16 axw_half2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float h16lo(unsigned u) { return (float)__builtin_bit_cast(axw_half2, u)[0]; }
__device__ __forceinline__ float h16hi(unsigned u) { return (float)__builtin_bit_cast(axw_half2, u)[1]; }
#define AXW_MFMA_32x32x16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define AXW_MFMA_16x16x32(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
#else
__device__ __forceinline__ float h16lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float h16hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
#define AXW_MFMA_32x32x16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#define AXW_MFMA_16x16x32(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
#endif
// fp32 accumulate of a 2-element dot product of packed h16 pairs (v_dot2c_f32_bf16 / v_dot2c_f32_f16), and the split of
// two fp32 values into packed (hi, lo) h16 pairs: x = hi + lo to 16 (bfloat16) / 22 (half) significant bits
#if AXW_F16
__device__ __forceinline__ float h16dot2(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(axw_half2, a), __builtin_bit_cast(axw_half2, b), c, false);
}
#else
typedef __bf16 axw_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float h16dot2(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(axw_bf2, a), __builtin_bit_cast(axw_bf2, b), c, false);
}
#endif
__device__ __forceinline__ void h16split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const h16 h0 = (h16)x0, h1 = (h16)x1;
  const h16 l0 = (h16)(x0 - (float)h0), l1 = (h16)(x1 - (float)h1);
  hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
  lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
}
// sum / maximum over the wave in every lane (all 64 lanes active): DPP butterflies inside a 16-lane row, then
// v_permlane{16,32}_swap across rows — ~8 short vector instructions; the __shfl_xor form compiles to six dependent
// ds_bpermute round trips (~700 cycles of latency per call, which bounded the K/V block loop of decode_attention_kernel)
#define AXW_DPP_F(CTRL, X) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xf, 0xf, true))
__device__ __forceinline__ float wave_sum(float v) {
  v += AXW_DPP_F(0xB1, v);   // quad_perm [1,0,3,2]
  v += AXW_DPP_F(0x4E, v);   // quad_perm [2,3,0,1]
  v += AXW_DPP_F(0x141, v);  // row_half_mirror
  v += AXW_DPP_F(0x140, v);  // row_mirror
  {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, AXW_DPP_F(0xB1, v));   // quad_perm [1,0,3,2]
  v = fmaxf(v, AXW_DPP_F(0x4E, v));   // quad_perm [2,3,0,1]
  v = fmaxf(v, AXW_DPP_F(0x141, v));  // row_half_mirror
  v = fmaxf(v, AXW_DPP_F(0x140, v));  // row_mirror
#undef AXW_DPP_F
  {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  return v;
}
// (value, index) maximum over the wave in every lane, lower index on equal values; DPP butterflies inside a 16-lane row,
// v_permlane{16,32}_swap across rows (the __shfl_xor form: twelve dependent ds_bpermute round trips)
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
  auto take = [&](float ov, int oi) { if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; } };
#define AXW_DPP_I(CTRL, X) __builtin_amdgcn_update_dpp(0, X, CTRL, 0xf, 0xf, true)
#define AXW_ARGMAX_STEP(CTRL) { const float ov = __int_as_float(AXW_DPP_I(CTRL, __float_as_int(v))); const int oi = AXW_DPP_I(CTRL, idx); take(ov, oi); }
  AXW_ARGMAX_STEP(0xB1)   // quad_perm [1,0,3,2]
  AXW_ARGMAX_STEP(0x4E)   // quad_perm [2,3,0,1]
  AXW_ARGMAX_STEP(0x141)  // row_half_mirror
  AXW_ARGMAX_STEP(0x140)  // row_mirror
#undef AXW_ARGMAX_STEP
#undef AXW_DPP_I
  {
    auto rv = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    auto ri = __builtin_amdgcn_permlane16_swap((unsigned)idx, (unsigned)idx, false, false);
    v = __uint_as_float(rv[0]); idx = (int)ri[0];
    take(__uint_as_float(rv[1]), (int)ri[1]);
  }
  {
    auto rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    auto ri = __builtin_amdgcn_permlane32_swap((unsigned)idx, (unsigned)idx, false, false);
    v = __uint_as_float(rv[0]); idx = (int)ri[0];
    take(__uint_as_float(rv[1]), (int)ri[1]);
  }
}

// exact-erf GELU (nn.GELU default; export_onnx.py:158-159 F.gelu)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
// The same GELU with erf from Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, i.e. far below one h16 ulp of the
// result): 1 rcp + 1 exp + 6 FMA instead of libm erff's ~60 instructions. Used where the result is narrowed to h16.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(fmaf(0.3275911f, z, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = 1.f - poly * __expf(-z * z);
  return 0.5f * x * (1.f + copysignf(e, x));
}
#endif

// Two values at once with packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32) and the hardware reciprocal (1 ulp; the
// correctly rounded __frcp_rn above expands to a division sequence of ~10 instructions): 2 rcp + 2 exp + ~12 packed ops
// per PAIR. In the mlp.0 epilogue of the encoder the GELU arithmetic is comparable to the tile's MFMA time.
#ifdef __HIPCC__
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf_fast2(f32x2_t x) {
  // 0.5 x (1 + erf(x / sqrt 2)) = 0.5 x + |x| (0.5 - 0.5 poly(t) exp(-x^2 / 2)),  t = 1 / (1 + p |x| / sqrt 2): erf is odd, so
  // the sign needs no copysign; 1/sqrt 2, the 0.5 and log2(e) are folded into the constants (17 instructions per pair)
  const f32x2_t ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2_t den = ax * (0.3275911f * 0.70710678118654752440f) + 1.f;
  const f32x2_t t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2_t poly = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
  poly = poly * t + (0.5f * 1.421413741f);
  poly = poly * t + (0.5f * -0.284496736f);
  poly = poly * t + (0.5f * 0.254829592f);
  poly = poly * t;
  const f32x2_t u = ax * 0.84932180028801904272f;  // sqrt(0.5 log2 e): exp(-x^2 / 2) = exp2(-u^2)
  const f32x2_t mu2 = -u * u;
  const f32x2_t ex = {__builtin_amdgcn_exp2f(mu2[0]), __builtin_amdgcn_exp2f(mu2[1])};
  const f32x2_t e = 0.5f - poly * ex;
  return ax * e + x * 0.5f;
}
#endif

// ------------------------------------------------------------------ GEMM (encoder)
// C[M,N] = A[M,K] (h16, row stride lda, rows may overlap) * W[N,K]^T (h16) with a fused epilogue.
enum GemmEpilogue : int {
  EPI_BIAS_BF16 = 0,       // C h16 = acc + bias
  EPI_BIAS_GELU_BF16 = 1,  // C h16 = gelu(acc + bias)
  EPI_GELU_POS_F32 = 2,    // C f32  = gelu(acc + bias) + aux[m][n]      (conv2 + positional embedding)
  EPI_RESID_F32 = 3,       // C f32 += acc + bias                        (out-proj / FFN2 residual)
  EPI_QKV = 4,             // n<d: Q h16 [m][d]; d<=n<2d: K h16 [m][d]; else V^T h16 [h][64][Tp]
  EPI_CROSS_KV = 5,        // decode layouts: K blocked [l][b][h][blk][8][64][8], V [l][b][h][Tp][64]
  EPI_PARTIAL_F32 = 6,     // split-K: part[split][batch*M + m][n] f32 = acc over this split's k range (no bias); the
                           // LayerNorm that follows folds the partials (+ bias) into the residual stream, in fixed order
};

struct GemmParams {
  const h16* A; long lda; long a_batch_stride;
  const h16* W;                     // [N][K]
  const float* bias;                 // [N] fp32
  void* C; long ldc; long c_batch_stride;
  const float* aux;                  // EPI_GELU_POS_F32: pos [M][N]
  void* C2; void* C3;                // EPI_QKV: K, V^T ; EPI_CROSS_KV: C=K blocked, C2=V
  long c2_batch_stride, c3_batch_stride;
  int M, N, K, batch;
  int d_model;                       // EPI_QKV / EPI_CROSS_KV
  int t_pad;                         // padded key count (multiple of 64)
  int n_batch_total;                 // EPI_CROSS_KV: slot count B in [l][B][h]...
  const int* kv_slot_map;            // EPI_CROSS_KV: clip b of this launch -> slot kv_slot_map[b] (device), nullptr: slot b
  int n_layer;                       // EPI_CROSS_KV: decoder layers (weight rows: all K, then all V)
  int n_begin;                       // set by launch_gemm: first output column of this launch
  int n_tiles;                       // set by launch_gemm: 128-column tiles of this launch
  int epilogue;
  int qkv_part;                      // EPI_QKV: 0 both launches, 1 only Q,K columns, 2 only V columns (run side by side on two streams)
  int ksplit;                        // EPI_PARTIAL_F32: K slices (grid.y), K / 64 divisible by it
  float* part; long part_stride;     // EPI_PARTIAL_F32: partial sums, one [batch*M][N] slab per slice
};
void launch_gemm(const GemmParams& p, hipStream_t s);

// fp32 [rows][d] -> h16 [rows][d] LayerNorm (eps 1e-5)
// n_part > 0: first x[row] += bias + part[0][row] + ... + part[n_part-1][row] (split-K partials of the GEMM before, slab
// stride part_stride), written back to x
void launch_layernorm_bf16(float* x, const float* g, const float* b, h16* y, long rows, int d, hipStream_t s,
                           const float* part = nullptr, int n_part = 0, long part_stride = 0, const float* part_bias = nullptr);

// encoder self-attention (non-causal, T keys), Q/K h16 [B][T][d], V^T h16 [B][H][64][Tp] (frames of every 16-group in
// the order [0-3, 8-11, 4-7, 12-15], as EPI_QKV writes them) -> O h16 [B][T][d]. rescale_thr: how far (log2 units) a
// tile's row maximum may exceed the running one before the accumulators are rescaled; 0 = on every increase.
void launch_encoder_attention(const h16* q, const h16* k, const h16* vt, h16* o, int batch, int T, int t_pad,
                              int d_model, int n_head, hipStream_t s, float rescale_thr = 8.f);

// ------------------------------------------------------------------ front-end
struct FrontendParams {
  const float* pcm;        // device [batch][stride]
  int stride;
  const int* n_samples;    // device [batch]
  int batch, n_mels;
  const float* twiddle;    // device [400][2] cos,sin
  const float* window;     // device [400]
  const float* mel_basis;  // device [n_mels][201]
  float* power;            // device scratch [batch][n_frames_max][208]
  float* logmel;           // device scratch [batch][n_mels][3008]
  unsigned* gmax;          // device [batch] (float bits, ordered-int encoded)
  float* mel_ref;          // device [batch][n_mels][3000] f32 (reference layout) or nullptr
  h16* mel_tm;            // device [batch][mel_rows][n_mels] h16 time-major, row 0 = left pad, or nullptr
  int mel_rows;
  int max_frames;          // frames computed per clip (<= 3001)
  const float* overflow;   // samples beyond `stride` of clips longer than the staging row (device, packed), or nullptr:
  const long long* over_off;  //   sample j >= stride of clip b is overflow[over_off[b] + j - stride] (device [batch])
  int openai;              // 1: the fp32 ONNX lineage's front-end (SURVEY A.1 column 3, generate_data.py:162-176): clip
                           // zero-padded / trimmed to 30 s before the STFT, last frame dropped, no zero fill
};
void launch_frontend(const FrontendParams& p, hipStream_t s);
void launch_mel_to_tm(const float* mel_ref, h16* mel_tm, int batch, int n_mels, int mel_rows, hipStream_t s);

// ------------------------------------------------------------------ decoder
struct DecState {          // device-resident loop state, one per engine
  int step;                // decoder steps run since the last reset (bookkeeping; no kernel derives a position from it)
  int n_done;
  int pad0, pad1;
};
// Every utterance slot has its OWN offset (`off[b]`: the position fed to this step = number of cached self-attention keys
// = the `offset` input of the reference's decoder graph, export_onnx.py:312-336 / Whisper.cpp:207-222, which handles one
// utterance at a time and so has one): slots of one batch may be at different positions, a finished slot stops
// advancing, and a new clip can be admitted into it while the others decode on (Engine::stream_*). Kernel parameter
// structs carry `off` with the same clip origin as their other per-clip pointers.

struct DecLayerW {
  const float *attn_ln_w, *attn_ln_b, *cross_ln_w, *cross_ln_b, *mlp_ln_w, *mlp_ln_b;
  const h16 *w_qkv, *w_o, *w_cq, *w_co, *w_fc1, *w_fc2;
  const float *b_qkv, *b_o, *b_cq, *b_co, *b_fc1, *b_fc2;
};

// Layout of the decoder-layer weight arenas (engine.cpp load_weights): layer l's h16 matrices start at
// w_base + l * w_stride(d), in units of d*d elements; its fp32 vectors at f_base + l * f_stride(d), in units of d.
struct DecArena {
  enum : int { W_QKV = 0, W_O = 3, W_CQ = 4, W_CO = 5, W_FC1 = 6, W_FC2 = 10, W_UNITS = 14 };
  enum : int { F_ATTN_LN_W = 0, F_ATTN_LN_B = 1, F_B_QKV = 2, F_B_O = 5, F_CROSS_LN_W = 6, F_CROSS_LN_B = 7, F_B_CQ = 8, F_B_CO = 9,
               F_MLP_LN_W = 10, F_MLP_LN_B = 11, F_B_FC1 = 12, F_B_FC2 = 16, F_UNITS = 17 };
  static constexpr __host__ __device__ long w_stride(int d) { return (long)W_UNITS * d * d; }
  static constexpr __host__ __device__ long f_stride(int d) { return (long)F_UNITS * d; }
};

enum GemvPrologue : int { PRO_PLAIN = 0, PRO_LAYERNORM = 1, PRO_ATTN_COMBINE = 2 };
enum GemvEpilogue : int {
  GEPI_STORE = 0,      // out[b][n] = y + bias
  GEPI_GELU = 1,       // out[b][n] = gelu(y + bias)
  GEPI_RESID = 2,      // out[b][n] += y + bias
  GEPI_QKV_CACHE = 3,  // n<d: q; then self-K (blocked) / self-V cache rows at `step`
  GEPI_LOGITS = 4,     // per-WG argmax partials (+ optional full logits dump)
  GEPI_PARTIAL = 5,    // batched path only: split-K partial sums [ksplit][part_batch][N], folded by the consumer
};

struct GemvParams {
  const h16* W; const float* bias; int N, K, batch;
  // prologue
  int prologue;
  const float* in;            // PRO_PLAIN: [B][K]; PRO_LAYERNORM: x [B][K]
  const float* ln_w; const float* ln_b;
  const float* part; int n_split; int n_head;   // PRO_ATTN_COMBINE: partials [B][H][n_split][66]
  // epilogue
  int epilogue;
  float* out;                 // [B][N]
  h16* k_cache; h16* v_cache; long kv_batch_stride; int d_model; int n_ctx_pad;  // GEPI_QKV_CACHE (this layer)
  const DecState* state;
  const int* off;                  // per-clip offsets [batch] (GEPI_QKV_CACHE: cache row; GEPI_LOGITS: skip while every clip is below skip_before_step)
  float* amax_val; int* amax_idx; int amax_stride;  // GEPI_LOGITS: partials [clip][amax_stride], one per workgroup
  float* logits_dump;              // optional logits row of this step for clip b at logits_dump + b*logits_dump_stride
  long logits_dump_stride;
  int skip_before_step;            // GEPI_LOGITS: do nothing while state->step < this (SOT steps)
};
void launch_gemv(const GemvParams& p, hipStream_t s);
int gemv_grid(const GemvParams& p);  // number of workgroups launch_gemv will use

// x[b] = tok_emb[tok[b]] + pos[off[b]]
void launch_embed(const h16* tok_emb, const float* pos, const int* tok, const int* off, float* x, int batch, int d,
                  hipStream_t s);

// single-query attention over blocked K / row-major V, writes split partials [B][H][n_split][66]
struct DecAttnParams {
  const float* q;             // [B][d]
  const h16* k; const h16* v; long kv_batch_stride;   // this layer, slot 0
  float* part; int n_split;
  int batch, n_head, d_model;
  int n_keys;                 // fixed key count (cross) or -1: off[b] + 1 (self)
  int cap_blocks;             // allocated 64-key blocks per (slot, head): 24 cross, 7 self
  const DecState* state;
  const int* off;             // per-clip offsets [B]
  const int* done;            // device [B], never null: clips whose flag is set are skipped (greedy loop past their eot); all zero where nobody stops
  int done_late;              // few clips (a launch is one dependent chain): the flag is looked at BEHIND the first requests, not before them
  h16* out_hi; h16* out_lo; int nbs; // normalised output as a fragment-major h16 pair instead of partials; with n_split > 1
                                     // the splits of a (clip, head) meet through mpart / mcnt and the last one to arrive writes it
  float* mpart; unsigned* mcnt;      // [B][H][n_split][66] / [B][H] (zero between launches), same clip origin as q / out
  // fused query projection (batched cross-attention): q = Wq[head rows] . LayerNorm(x[b]) + bq computed by the
  // (clip, head) workgroup itself while its first K/V block is in flight; wq == nullptr: q is read from `q`
  const float* x; const float* ln_w; const float* ln_b; const h16* wq; const float* bq;
  // folded query (decode_gemm.hip "QUERY FOLD"; tq != nullptr): q[j] = rstd (tq[b][j] - mean fold_s[j]) + fold_c[j], mean / rstd of
  // the clip's residual row from its stat_part[b][d_model / 16][2] block statistics
  const float* tq; const float* stat_part; const float* fold_s; const float* fold_c;
  // measurement only (Engine::bench "attn_stamp"): [workgroups][2] = {begin, end} of every workgroup of this launch in
  // 100 MHz wall-clock ticks, written by the kernel itself (a separate template instantiation: the production kernel
  // carries no stamp code)
  unsigned long long* stamp;
};
void launch_decode_attention(const DecAttnParams& p, hipStream_t s);

// ---- batched decode (3..64 clips per launch): activations as h16 (hi, lo) pairs, MFMA GEMM (decode_gemm.hip)
struct DecGemmParams {
  const h16* W;                           // fragment-major packed weights
  const float* bias; int N, K, batch;
  const h16* a_hi; const h16* a_lo;      // fragment-major h16 pair (decode_gemm.hip), this launch's first clip block
  int nbs;                                 // allocated clip blocks (stride of the activation layout)
  int epilogue;                            // GemvEpilogue
  int rt;                                  // weight-row tiles per wave: 1 (16 rows/WG) or 4 (64 rows/WG, vocabulary)
  float* out;                              // fp32 [batch][N] (STORE / RESID / q of QKV_CACHE)
  h16* out_hi; h16* out_lo;              // GEPI_GELU: h16 pair [batch][N]
  h16* k_cache; h16* v_cache; long kv_batch_stride; int d_model; int n_ctx_pad;
  const DecState* state;
  const int* off;                          // per-clip offsets [batch]
  float* amax_val; int* amax_idx; int amax_stride;
  float* logits_dump; long logits_dump_stride;
  int skip_before_step;
  int ksplit; int part_batch;              // GEPI_PARTIAL: K slices (grid.y) and the clip stride of the partial buffer
};
void launch_decode_gemm(const DecGemmParams& p, hipStream_t s);

// Clip-block form of the batched linear layer: one workgroup = 16 clips x 16*rt weight rows over the whole K, so the
// activations of a workgroup are 16 rows instead of 64, LayerNorm of the residual stream can be the prologue (no
// separate preparation launch, no activation round trip through memory) and the residual add can be the epilogue
// (every output element has exactly one owner: no split-K partials to fold).
struct DecCGemmParams {
  const h16* W;                           // fragment-major packed weights
  const float* bias; int N, K, batch;
  const float* x; const float* ln_w; const float* ln_b;   // ln_w != nullptr: input = LayerNorm(x [batch][K] fp32)
  const h16* a_hi; const h16* a_lo;      // else: fragment-major h16 pair
  int nbs;                                 // allocated clip blocks (stride of the pair layouts)
  int epilogue;                            // GEPI_STORE / GEPI_GELU / GEPI_RESID / GEPI_QKV_CACHE
  int rt;                                  // 1 or 2
  float* out;                              // fp32 [batch][N] (STORE, RESID: out += y; q of QKV_CACHE)
  h16* out_hi; h16* out_lo;              // GEPI_GELU: h16 pair
  h16* k_cache; h16* v_cache; long kv_batch_stride; int d_model; int n_ctx_pad;
  const DecState* state;
  const int* off;                          // per-clip offsets [batch]
  // query fold of the clip-block step (decode_gemm.hip "QUERY FOLD"); fold_row0 == 0: none
  int fold_row0;                           // rows >= fold_row0 are fold rows (QKV launch: 3 d_model; o launch: d_model)
  const float* ln_w2;                      // LayerNorm-prologue launch: the fold rows' input is ln_w2 . x (no statistics, no shift)
  const h16* W_lo;                         // pair-input launch: lo halves of the fold rows' weights (fragment-major, row block 0 = fold_row0)
  float* out2;                             // the fold rows' output / residual target [batch][N - fold_row0]
  float* stat_part;                        // GEPI_RESID: [batch][fold_row0 / 16][2] = (sum, squares about the block mean) of the new rows
  // measurement only (Engine::bench "attn_stamp"): [workgroups][2] = {begin, time at stamp_point} of every workgroup, 100 MHz ticks;
  // a separate template instantiation (the production kernel carries no stamp code)
  unsigned long long* stamp; int stamp_point;
};
void launch_decode_cgemm(const DecCGemmParams& p, hipStream_t s);
int decode_gemm_grid(int N, int rt);
bool decode_logits_resident_ok(int K, int batch);  // rt == 0 (one workgroup per CU, activations in registers) supports this shape
void launch_act_prep(float* x, const float* g, const float* be, h16* hi, h16* lo, int batch, int K, bool do_ln, int nbs,
                     const float* part, int n_part, int part_batch, const float* part_bias, hipStream_t s);
void launch_pack_weight_frag(const h16* w, h16* wp, int N, int K, hipStream_t s);
void launch_pack_weight_frag_split(const float* w, h16* hi, h16* lo, int N, int K, hipStream_t s);  // fp32 -> (hi, lo) h16 pair

struct AdvanceParams {
  const float* amax_val; const int* amax_idx; int n_part; int amax_stride;
  DecState* state; int* off; int* tok; int* done; int* n_out; int* out_ids; int batch;
  int* done_host;             // optional host-mapped [batch]: set (behind a system-scope fence) when a clip finishes, so that the
                              // host sees it without a copy or a synchronisation (Engine::stream_step)
  int n_ctx, eot, max_new, n_vocab;
  const int* max_new_clip;    // optional device [B]: per-clip id budget (a ragged batch), capped by max_new
  const int* sot;             // device [4]
  const int* forced; int n_forced;   // teacher forcing (device [B][n_forced]) or nullptr
  int* argmax_dump;           // optional [B][n_forced+1]
  const h16* tok_emb; const float* pos; float* x; int d_model;  // fused embedding of the next step
};
void launch_advance(const AdvanceParams& p, hipStream_t s);

// ---- persistent batch-1 decode (decode_persistent.hip): the whole greedy loop of one clip in ONE launch
typedef unsigned long long u64;
struct PersistParams {
  const h16* wl; const float* fl;   // decoder-layer weight arenas (DecArena layout)
  const float* qf;                  // one-clip launch, d_model <= 768: the query-fold arena (decode_persistent.hip), nullptr = unfolded
  const h16* tok_emb; const float* pos; const float* ln_w; const float* ln_b;
  const h16* cross_k; const h16* cross_v; long cross_layer_stride;  // this clip's slot, layer 0
  int n_layer, n_vocab, n_ctx, n_audio_ctx;
  int eot, max_new, total_steps;
  const int* sot;               // device [4]
  const int* forced; int n_forced; float* logits_dump; int* argmax_dump;   // teacher forcing (tests)
  u64* gran; unsigned* err;     // granule area + error word, zeroed before every launch
  int gran_bytes;               // size of the granule area (buffer-resource bound of the 16-byte polls)
  int* out_ids; int* n_out; DecState* state;
  int fault;                    // test hook (AX_WHISPER_PERSIST_FAULT=1): workgroup 0 leaves at once, so the launch must give up
  long long* prof;              // optional [grid][64]: per-phase 100 MHz tick sums + one layer's absolute timeline (AX_WHISPER_PERSIST_PROF), else nullptr
  // Two clips in one launch (n_clip == 2; greedy decode only): every phase runs for clip 0 and then for clip 1 with the same
  // weight rows in registers, so that one clip's hand-off is in flight while the other clip's rows are computed. Clip 0 keeps
  // the LDS-resident K/V of the one-clip launch; clip 1 reads its K/V from global memory: its cross K/V in the next slot
  // (cross_clip_stride elements on), its self-attention cache in self_k1 / self_v1 ([n_layer * n_head][8 blocks][4096], the
  // per-owner layouts of the LDS cache). Granules, ids and counts of clip 1: gran + gran_clip_u64, out_ids1, n_out1.
  int n_clip;
  int prof_clip;  // two-clip timeline: whose absolute stamps are kept (decode_persistent2.hip)
  long cross_clip_stride;
  h16* self_k1; h16* self_v1;
  long gran_clip_u64;
  int* out_ids1; int* n_out1; int max_new1;
  // three clips (n_clip == 3): clip 2's ids at out_ids1 + n_ctx, its count at n_out1 + 1, its self cache self_clip_stride elements on
  int max_new2; long self_clip_stride;
};
bool decode_persistent_supported(int d_model, int n_head, int n_layer, int n_cu);
int decode_persistent_grid(int d_model, int n_cu);
size_t decode_persistent_gran_bytes(int d_model, int grid);   // granule area; the error word sits in its last 8 bytes
size_t qfold_floats(int d_model, int n_layer);                // floats of the query-fold arena
void launch_qfold_build(const h16* wl, const float* fl, float* qf, int d_model, int n_layer, hipStream_t s);
hipError_t launch_decode_persistent(const PersistParams& p, int d_model, int grid, hipStream_t s);
hipError_t launch_decode_persistent2(const PersistParams& p, int d_model, int grid, hipStream_t s);  // n_clip == 2 (decode_persistent2.hip)
int decode_persistent_max_clips(int d_model, int n_head, int n_layer, int grid);  // clips per persistent launch: 1, 2 or 3  // shapes whose every linear layer is ONE pass of rows per workgroup

// weight preparation (device): raw file dtype -> h16 / fp32, with the layout changes the kernels want
void launch_convert_to_h16(const void* src, int src_dtype /*0 f32,1 bf16,2 f16*/, h16* dst, long n, hipStream_t s);
void launch_convert_to_f32(const void* src, int src_dtype, float* dst, long n, hipStream_t s);
// conv weight [Cout][Cin][3] -> [Cout][Kpad] with k-major taps: dst[n][k*Cin + c]
void launch_conv_weight_pack(const void* src, int src_dtype, h16* dst, int cout, int cin, int kpad, hipStream_t s);

}  // inline namespace AXW_NS
}  // namespace axw
