// decode_gemm.hip — the decoder step's nn.Linear layers for 3..64 clips per launch on the matrix cores.
//
// y[b][n] = sum_k W[n][k] a[b][k] (+ bias[n]): W h16 [N][K] streamed from HBM exactly once per step for the
// whole batch (277.8 MB/step for small), activations fp32-equivalent: every activation is carried
// as a h16 pair (hi, lo) with hi + lo == x to 16 mantissa bits, and both terms are multiplied on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation. The step stays HBM-bound (two MFMAs per 1 KiB of weights
// is far below the matrix rate), so the second term is free and keeps batched decoding numerically in
// line with the fp32-FMA GEMV used for 1..4 clips (decode_gemv.hip).
//
// Mapping (wave64): C[n][b] = W[n][:] . a[b][:], A operand = 16 weight rows, B operand = 16 clips.
//   workgroup = 8 waves = 16*RT weight rows; the 8 waves split K (k-steps round-robin), then reduce through LDS;
//   a wave keeps RT x NB accumulator tiles (NB = ceil(batch/16) <= 4) so one activation fragment feeds RT MFMAs;
//   weights: 16-byte loads, 4 lanes cover 64 contiguous bytes of a row per k-step, k-steps unrolled so the
//   whole 128-byte line is requested back to back; activations come from L2 (h16 pairs written by the
//   producer: act_prep_kernel / attention / the GELU epilogue).
// Three kernels: decode_cgemm_kernel (clip blocks of 16, LayerNorm prologue / residual epilogue: the default sequence
// for d_model <= 1024), decode_logits_kernel (vocabulary projection with register-resident activations) and
// decode_gemm_kernel + act_prep_kernel (64 clips per workgroup, split-K partials: d_model > 1024 and the fallbacks).
// Epilogues mirror decode_gemv.hip: bias, GELU (writes the h16 pair), residual add, q + KV-cache append,
// vocabulary argmax partials (first max wins, Whisper.cpp:42-45).
#include "common.hpp"
#include <algorithm>

namespace axw {
inline namespace AXW_NS {

__device__ __forceinline__ void split_bf16(float x, h16& hi, h16& lo) {
  hi = (h16)x;
  lo = (h16)(x - (float)hi);
}

// Fragment-major layouts: both MFMA operands are stored in the order the 16x16x32 instruction consumes them, so
// every operand load of a wave is ONE contiguous 1 KiB access (lane*16 bytes):
//   activations  ap[((ks * nbs + c) * 64 + lane) * 8 + j] = a[clip = c*16 + (lane&15)][k = ks*32 + (lane>>4)*8 + j]
//   weights      wp[((nb * KS + ks) * 64 + lane) * 8 + j] = W[row = nb*16 + (lane&15)][k = ks*32 + (lane>>4)*8 + j]
__device__ __forceinline__ long frag_index(int row, int k, int row_blocks_stride /* nbs or unused */, int ks_count, bool weight) {
  const int ks = k >> 5, q = (k >> 3) & 3, j = k & 7, blk = row >> 4, r = row & 15;
  const long tile = weight ? ((long)blk * ks_count + ks) : ((long)ks * row_blocks_stride + blk);
  return (tile * 64 + q * 16 + r) * 8 + j;
}
__device__ __forceinline__ void store_pair_frag(float x, h16* hi, h16* lo, int clip, int k, int nbs) {
  h16 h, l;
  split_bf16(x, h, l);
  const long i = frag_index(clip, k, nbs, 0, false);
  hi[i] = h; lo[i] = l;
}

// Residual fold + LayerNorm of the residual stream -> fragment-major h16 (hi, lo) rows; one workgroup per clip.
// x[b] += bias + sum of the previous GEMM's split-K partials (fixed order: deterministic), then LayerNorm; every
// global load is issued up front and the row is touched once (values stay in registers between the two reductions).
__global__ __launch_bounds__(256) void act_prep_kernel(float* x, const float* __restrict__ g, const float* __restrict__ be,
                                                       h16* __restrict__ hi, h16* __restrict__ lo, int K, int do_ln, int nbs,
                                                       const float* __restrict__ part, int n_part, int part_batch,
                                                       const float* __restrict__ part_bias) {
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* xr = x + (long)b * K;
  constexpr int MAXE = 8;  // K <= 2048
  float v[MAXE], gg[MAXE], bb[MAXE], pb[MAXE], ps[4][MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int c = tid + 256 * e;
    const bool on = c < K;
    v[e] = on ? xr[c] : 0.f;
    gg[e] = (on && do_ln) ? g[c] : 1.f;
    bb[e] = (on && do_ln) ? be[c] : 0.f;
    pb[e] = (on && n_part > 0) ? part_bias[c] : 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) ps[s][e] = (on && s < n_part) ? part[((long)s * part_batch + b) * K + c] : 0.f;
  }
  float s1 = 0.f;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    if (n_part > 0) {
      v[e] += pb[e];
#pragma unroll
      for (int s = 0; s < 4; ++s) v[e] += ps[s][e];
      const int c = tid + 256 * e;
      if (c < K) xr[c] = v[e];
    }
    s1 += v[e];
  }
  if (!do_ln) {
#pragma unroll
    for (int e = 0; e < MAXE; ++e) { const int c = tid + 256 * e; if (c < K) store_pair_frag(v[e], hi, lo, b, c, nbs); }
    return;
  }
  s1 = wave_sum(s1);
  if (lane == 0) red[wave] = s1;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / K;
  float s2 = 0.f;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) { const int c = tid + 256 * e; const float t = c < K ? v[e] - mean : 0.f; s2 += t * t; }
  s2 = wave_sum(s2);
  if (lane == 0) red[4 + wave] = s2;
  __syncthreads();
  const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / K + 1e-5f);
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int c = tid + 256 * e;
    if (c < K) store_pair_frag((v[e] - mean) * rstd * gg[e] + bb[e], hi, lo, b, c, nbs);
  }
}

void launch_act_prep(float* x, const float* g, const float* be, h16* hi, h16* lo, int batch, int K, bool do_ln, int nbs,
                     const float* part, int n_part, int part_batch, const float* part_bias, hipStream_t s) {
  hipLaunchKernelGGL(act_prep_kernel, dim3(batch), dim3(256), 0, s, x, g, be, hi, lo, K, do_ln ? 1 : 0, nbs, part, n_part,
                     part_batch, part_bias);
}

// row-major h16 [N][K] -> fragment-major (rows padded to a multiple of 16 with zeros)
__global__ void pack_weight_frag_kernel(const h16* __restrict__ w, h16* __restrict__ wp, int N, int K) {
  const int KS = K / 32;
  const long total = (long)((N + 15) / 16) * KS * 512;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long tile = i >> 9;
    const int ks = (int)(tile % KS), nb = (int)(tile / KS);
    const int row = nb * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + j;
    wp[i] = row < N ? w[(long)row * K + k] : (h16)0.f;
  }
}
void launch_pack_weight_frag(const h16* w, h16* wp, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(pack_weight_frag_kernel, dim3(2048), dim3(256), 0, s, w, wp, N, K);
}

// row-major fp32 [N][K] -> TWO fragment-major h16 arrays, hi = h16(m) and lo = h16(m - hi) (the query fold's product matrix M)
__global__ void pack_weight_frag_split_kernel(const float* __restrict__ w, h16* __restrict__ hi, h16* __restrict__ lo, int N, int K) {
  const int KS = K / 32;
  const long total = (long)((N + 15) / 16) * KS * 512;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long tile = i >> 9;
    const int ks = (int)(tile % KS), nb = (int)(tile / KS);
    const int row = nb * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + j;
    h16 h = (h16)0.f, l = (h16)0.f;
    if (row < N) split_bf16(w[(long)row * K + k], h, l);
    hi[i] = h;
    lo[i] = l;
  }
}
void launch_pack_weight_frag_split(const float* w, h16* hi, h16* lo, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(pack_weight_frag_split_kernel, dim3(2048), dim3(256), 0, s, w, hi, lo, N, K);
}

template <int RT, int NB>
__global__ __launch_bounds__(512) void decode_gemm_kernel(DecGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [8 waves][RT][NB][16 clips][16 rows]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;  // 8 waves
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * (16 * RT);
  const int KS_all = p.K / 32;     // k-steps in total
  const int KS = KS_all / (int)gridDim.y;  // k-steps of this workgroup's K slice (blockIdx.y); wave w takes w, w+8, ...
  const int ks_base = blockIdx.y * KS;
  const int nb0 = blockIdx.x * RT; // first 16-row block of this workgroup
  const int n_rb = (p.N + 15) / 16;

  const h16* wrow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) wrow[t] = p.W + ((long)min(nb0 + t, n_rb - 1) * KS_all + ks_base) * 512 + lane * 8;
  const h16* ahi = p.a_hi + ks_base * ((long)p.nbs * 512) + lane * 8;
  const h16* alo = p.a_lo + ks_base * ((long)p.nbs * 512) + lane * 8;
  const long a_step = (long)p.nbs * 512;  // elements between consecutive k-steps of the activations

  // The bias of this thread's outputs is requested first: every output of a thread has the same weight row
  // (512 % (16 * RT) == 0), and a load issued in the epilogue, behind the reduction barrier, would add one more
  // memory round trip to a launch that is a single dependent chain.
  static_assert(512 % (16 * RT) == 0, "row of a thread's outputs");
  float bias_t = 0.f;
  if (p.bias && p.epilogue != GEPI_PARTIAL && n0 + tid % (16 * RT) < p.N) bias_t = p.bias[n0 + tid % (16 * RT)];

  f32x4 acc[RT][NB];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][c][e] = 0.f;

  // Three register sets of operand fragments: with 8 waves splitting K, Whisper-small's K = 768 is 3 k-steps
  // per wave, i.e. every operand byte of the workgroup is requested before the first MFMA (one memory round
  // trip); longer K rotates the sets. Every load is one contiguous 1 KiB wave access.
  struct Frag { h16x8 w[RT], h[NB], l[NB]; };
  auto load = [&](Frag& f, int ks) {
#pragma unroll
    for (int t = 0; t < RT; ++t) f.w[t] = *reinterpret_cast<const h16x8*>(wrow[t] + (long)ks * 512);
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      f.h[c] = *reinterpret_cast<const h16x8*>(ahi + ks * a_step + c * 512);
      f.l[c] = *reinterpret_cast<const h16x8*>(alo + ks * a_step + c * 512);
    }
  };
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        acc[t][c] = AXW_MFMA_16x16x32(f.w[t], f.h[c], acc[t][c]);
        acc[t][c] = AXW_MFMA_16x16x32(f.w[t], f.l[c], acc[t][c]);
      }
  };
  Frag f0, f1, f2;
  if (wave < KS) load(f0, wave);
  if (wave + 8 < KS) load(f1, wave + 8);
  if (wave + 16 < KS) load(f2, wave + 16);

  // the clips' offsets are read after the operand loads are in flight. SOT steps: logits are discarded
  // (Whisper.cpp:214-217) — the launch has nothing to do while every clip is still below skip_before_step
  if (p.epilogue == GEPI_LOGITS) {
    const int o = lane < p.batch ? p.off[lane] : -1;
    if (__ballot(o >= p.skip_before_step) == 0) return;
  }

  for (int ks = wave; ks < KS; ks += 24) {
    mma(f0);
    if (ks + 24 < KS) load(f0, ks + 24);
    if (ks + 8 < KS) {
      mma(f1);
      if (ks + 32 < KS) load(f1, ks + 32);
    }
    if (ks + 16 < KS) {
      mma(f2);
      if (ks + 40 < KS) load(f2, ks + 40);
    }
  }

  // split-K reduction across the 8 waves through LDS: red[wave][t][c][clip r][row 4q+e], rows fastest, so a
  // lane's 4 accumulator values are one 16-byte LDS store and the epilogue reads consecutive rows.
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int c = 0; c < NB; ++c)
      *reinterpret_cast<f32x4*>(red + (((wave * RT + t) * NB + c) * 16 + r) * 16 + 4 * q) = acc[t][c];
  __syncthreads();

  constexpr int OUT = RT * NB * 256;  // outputs of this workgroup
  for (int o = tid; o < OUT; o += 512) {
    // o -> (clip, row) with the weight row fastest: consecutive threads write consecutive n of one clip
    // (out[b][n], the V cache rows and the fragment-major pair are all contiguous in n)
    const int nl = o % (16 * RT), bl = o / (16 * RT);
    const int t = nl >> 4, nn = nl & 15, c = bl >> 4, bb = bl & 15;
    const int n = n0 + nl, b = bl;
    float y = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) y += red[(((w * RT + t) * NB + c) * 16 + bb) * 16 + nn];
    if (n >= p.N || b >= p.batch) continue;
    if (p.epilogue == GEPI_PARTIAL) {  // split-K slice: plain partial sums, folded (with the bias) by the consumer
      p.out[((long)blockIdx.y * p.part_batch + b) * p.N + n] = y;
      continue;
    }
    y += bias_t;
    switch (p.epilogue) {
      case GEPI_STORE: p.out[(long)b * p.N + n] = y; break;
      case GEPI_GELU: store_pair_frag(gelu_erf(y), p.out_hi, p.out_lo, b, n, p.nbs); break;
      case GEPI_RESID: p.out[(long)b * p.N + n] += y; break;
      case GEPI_QKV_CACHE: {
        const int d = p.d_model;
        if (n < d) {
          p.out[(long)b * d + n] = y;
        } else {
          const int step = p.off[b];  // this clip's cache row
          const int cc = (n < 2 * d) ? n - d : n - 2 * d;
          const int head = cc >> 6, dd = cc & 63;
          const long base = (long)b * p.kv_batch_stride + (long)head * p.n_ctx_pad * 64;
          if (n < 2 * d) p.k_cache[base + (long)(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (h16)y;
          else p.v_cache[base + (long)step * 64 + dd] = (h16)y;
        }
        break;
      }
      case GEPI_LOGITS:
        if (p.logits_dump) p.logits_dump[(long)b * p.logits_dump_stride + n] = y;
        red[8 * RT * NB * 256 + nl * (NB * 16) + b] = y;  // [16*RT rows][NB*16 clips] for the argmax below
        break;
    }
  }
  if (p.epilogue == GEPI_LOGITS) {  // per-clip argmax over this workgroup's rows, first max wins
    __syncthreads();
    if (tid < NB * 16 && tid < p.batch) {
      float best_v = -INFINITY;
      int best_i = 0x7fffffff;
      for (int i = 0; i < 16 * RT; ++i) {
        const int n = n0 + i;
        if (n >= p.N) break;
        const float y = red[8 * RT * NB * 256 + i * (NB * 16) + tid];
        if (y > best_v) { best_v = y; best_i = n; }
      }
      p.amax_val[(long)tid * p.amax_stride + blockIdx.x] = best_v;
      p.amax_idx[(long)tid * p.amax_stride + blockIdx.x] = best_i;
    }
  }
}

// v[l] + v[l^16] + v[l^32] + v[l^48] in every lane (the same pairing and order as two __shfl_xor steps, bit-identical,
// without their two dependent ds_bpermute round trips)
__device__ __forceinline__ float sum_lanes_16_32(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// sum over the 16 lanes of a DPP row (equal lane >> 4), in every lane of the row
__device__ __forceinline__ float row16_sum_f(float v) {
#define AXW_DPP_ROW(CTRL, X) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xf, 0xf, true))
  v += AXW_DPP_ROW(0xB1, v);   // quad_perm [1,0,3,2]
  v += AXW_DPP_ROW(0x4E, v);   // quad_perm [2,3,0,1]
  v += AXW_DPP_ROW(0x141, v);  // row_half_mirror
  v += AXW_DPP_ROW(0x140, v);  // row_mirror
#undef AXW_DPP_ROW
  return v;
}

// ---------------------------------------------------------------------------- clip-block GEMM (DecCGemmParams)
// grid = (weight-row blocks of 16*RT, clip blocks of 16); 8 waves split K (k-steps w, w+8, ...), reduce through LDS;
// at most one output per thread (RT <= 2), so the bias and the residual value are requested before anything else.
// CH > 0: LayerNorm prologue, CH = k-steps per wave held in registers (K = 256*CH at most); CH == 0: h16-pair input.
// STAMP (measurement builds only, Engine::bench "attn_stamp"): thread 0 of every workgroup records its start and the time it reaches
// point p.stamp_point: 1 every request of the prologue is out | 2 LayerNorm statistics done (CH > 0) | 3 MFMAs issued | 4 the eight
// waves' partial sums are in LDS (behind the barrier) | 0 end
// QUERY FOLD (round 6; p.fold_row0 > 0): the cross-attention query of the clip-block step is folded through the self-attention output
// projection as in the one-clip launch (decode_persistent.hip header): cq = r (T - mu s) + c with T = A0 + M a + d,
// A0 = W_cq (g . x0), M = W_cq diag(g) W_o. Rows >= fold_row0 of a launch are those extra rows:
//   QKV launch (CH > 0), rows [3d, 4d): W_cq against g_cross . x0 (no statistics, no shift) -> out2 = A0
//   o launch (FOLD), rows [d, 2d): M as an (hi, lo) h16 pair against the attention vector, added onto out2 with bias d -> out2 = T;
//   its ordinary rows also leave (sum, centred squares) of the new residual rows per 16-row block in stat_part.
// decode_attention_kernel<2> then builds its head's query from 64 T values, the 48 partial statistics and two constant vectors:
// no LayerNorm barriers, no 98 KB of W_cq through every (clip, head) workgroup, 50 registers fewer.
template <int RT, int CH, bool STAMP = false, bool FOLD = false>
__global__ __launch_bounds__(512) void decode_cgemm_kernel(DecCGemmParams p) {
  __shared__ __attribute__((aligned(16))) float red[8 * RT * 256];  // [wave][t][clip][row]
  __shared__ float stat[2][8][16];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: k-step selection is wave-uniform control flow
  unsigned long long* my_stamp = nullptr;
  if constexpr (STAMP) {
    my_stamp = p.stamp + 2 * (blockIdx.y * gridDim.x + blockIdx.x);
    if (tid == 0) my_stamp[0] = (unsigned long long)wall_clock64();
  }
  auto stamp_at = [&](int k) { if constexpr (STAMP) { if (tid == 0 && p.stamp_point == k) my_stamp[1] = (unsigned long long)wall_clock64(); } };
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * (16 * RT), cb = blockIdx.y;
  const int KS = p.K / 32;
  const int nb0 = blockIdx.x * RT, n_rb = (p.N + 15) / 16;
  const int fr0 = p.fold_row0 > 0 ? p.fold_row0 : 0x7fffffff;  // first query-fold row (a multiple of 16 RT)
  const bool fw = n0 >= fr0;                                   // workgroup-uniform: this workgroup's rows are fold rows
  const int nx = min(p.N, fr0);                                // row length of the primary output

  // this thread's output: weight row n0 + tid % (16 RT), clip cb*16 + tid / (16 RT)
  const int o_n = n0 + tid % (16 * RT), o_b = cb * 16 + tid / (16 * RT);
  const bool has_out = tid < RT * 256 && o_n < p.N && o_b < p.batch;
  float bias_t = 0.f, old_t = 0.f;
  if (has_out && p.bias) bias_t = p.bias[o_n];
  if (has_out && p.epilogue == GEPI_RESID) old_t = fw ? p.out2[(long)o_b * (p.N - fr0) + (o_n - fr0)] : p.out[(long)o_b * nx + o_n];
  const int step = (has_out && p.epilogue == GEPI_QKV_CACHE) ? p.off[o_b] : 0;  // this thread's clip: its own cache row

  const h16* wrow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) wrow[t] = p.W + (long)min(nb0 + t, n_rb - 1) * KS * 512 + lane * 8;

  f32x4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;

  if constexpr (CH > 0) {
    // weights of this wave's k-steps: all requested before the LayerNorm arithmetic
    h16x8 w[CH][RT];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ks = min(wave + 8 * c, KS - 1);
#pragma unroll
      for (int t = 0; t < RT; ++t) w[c][t] = *reinterpret_cast<const h16x8*>(wrow[t] + (long)ks * 512);
    }
    // lane (r, q) holds x[clip r][k = ks*32 + 8q .. +8] of every k-step of this wave: exactly its MFMA B fragment
    const float* xr = p.x + (long)min(cb * 16 + r, p.batch - 1) * p.K;
    f32x4 v[CH][2];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k0 = min(wave + 8 * c, KS - 1) * 32 + q * 8;
#pragma unroll
      for (int u = 0; u < 2; ++u) v[c][u] = *reinterpret_cast<const f32x4*>(xr + k0 + 4 * u);
    }
    // The LayerNorm gain and bias go through LDS: the 16 clip lanes of a k-group need the SAME eight values, and as per-lane 16-byte
    // global loads they were half of this prologue's vector-memory instructions (12 of 24 per wave) — the prologue is bound by how
    // fast a CU issues them (the launch's last workgroup had its requests out after 1.3 us at 144 workgroups, 3.4 us at 288:
    // profiles/r05_step_timeline_b4.txt). Two dword loads per thread instead, parked behind the first statistics barrier.
    __shared__ __attribute__((aligned(16))) float s_ln[2][CH * 256];
    float lnw_t[(CH * 256 + 511) / 512], lnb_t[(CH * 256 + 511) / 512];
#pragma unroll
    for (int i = 0; i < (CH * 256 + 511) / 512; ++i) {
      const int kk = tid + 512 * i;
      lnw_t[i] = kk < p.K ? (fw ? p.ln_w2[kk] : p.ln_w[kk]) : 0.f;  // fold rows: g_cross . x, unnormalised, unshifted
      lnb_t[i] = (kk < p.K && !fw) ? p.ln_b[kk] : 0.f;
    }
    stamp_at(1);
    // statistics per clip: lanes (r, 0..3) of 8 waves hold one row between them. ONE pass and one barrier (round 6): sum and sum of
    // squares together, var = E[x^2] - mean^2 in fp32 (a residual row's mean is small against its spread — outlier channels of +-500 among
    // 768 values give E[x^2] / var < 4 — so the subtraction costs two bits, not the result; the two-pass form behind two barriers held
    // every wave 0.3 us longer in front of its MFMAs, 24 launches per step and branch)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (wave + 8 * c < KS) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          s1 += (v[c][u][0] + v[c][u][1]) + (v[c][u][2] + v[c][u][3]);
          s2 += (v[c][u][0] * v[c][u][0] + v[c][u][1] * v[c][u][1]) + (v[c][u][2] * v[c][u][2] + v[c][u][3] * v[c][u][3]);
        }
      }
    s1 = sum_lanes_16_32(s1);
    s2 = sum_lanes_16_32(s2);
    if (q == 0) { stat[0][wave][r] = s1; stat[1][wave][r] = s2; }
#pragma unroll
    for (int i = 0; i < (CH * 256 + 511) / 512; ++i) {
      const int kk = tid + 512 * i;
      if (kk < CH * 256) { s_ln[0][kk] = lnw_t[i]; s_ln[1][kk] = lnb_t[i]; }
    }
    __syncthreads();
    float mean = 0.f, ex2 = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 8; ++w2) { mean += stat[0][w2][r]; ex2 += stat[1][w2][r]; }
    mean = fw ? 0.f : mean / (float)p.K;
    const float var = fmaxf(ex2 / (float)p.K - mean * mean, 0.f);
    const float rstd = fw ? 1.f : rsqrtf(var + 1e-5f);
    stamp_at(2);
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (wave + 8 * c < KS) {
        h16x8 hi, lo;
        const int k0 = (wave + 8 * c) * 32 + q * 8;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 gg = *reinterpret_cast<const f32x4*>(s_ln[0] + k0 + 4 * u), bb = *reinterpret_cast<const f32x4*>(s_ln[1] + k0 + 4 * u);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = (v[c][u][e] - mean) * rstd * gg[e] + bb[e];
            h16 hh, ll;
            split_bf16(y, hh, ll);
            hi[4 * u + e] = hh; lo[4 * u + e] = ll;
          }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          acc[t] = AXW_MFMA_16x16x32(w[c][t], hi, acc[t]);
          acc[t] = AXW_MFMA_16x16x32(w[c][t], lo, acc[t]);
        }
      }
  } else {
    // Activation fragments as buffer loads: lane (r, q) holds clip cb*16 + r, and the lanes of a clip the batch does not have get
    // an out-of-range offset — their loads return zeros and move NO bytes. A clip block is 2 x 1 KB per k-step whatever it
    // holds: at 4 clips mlp.2 (K = 3072) pulled 196 KB of activations per workgroup through its CU for 49 KB of content, and
    // was the slowest GEMM launch of the few-clip step (6.0 us; profiles/r05_step_timeline_b4.txt).
    const long a_step = (long)p.nbs * 512;
    const unsigned a_bytes = (unsigned)((long)KS * a_step * 2 - (long)cb * 1024);
    const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a_hi + (long)cb * 512), 0, (int)a_bytes, 0x27000);
    const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a_lo + (long)cb * 512), 0, (int)a_bytes, 0x27000);
    const int a_off = cb * 16 + r < p.batch ? lane * 16 : 0x7ffffff0;
    struct Frag { h16x8 w[RT], wl[FOLD ? RT : 1], h, l; };
    const h16* wlrow[RT];  // FOLD: the lo halves of the fold rows' weights (their own array, row block 0 = row fold_row0)
#pragma unroll
    for (int t = 0; t < RT; ++t) wlrow[t] = (FOLD && fw) ? p.W_lo + (long)(nb0 + t - fr0 / 16) * KS * 512 + lane * 8 : nullptr;
    auto load = [&](Frag& f, int ks) {
#pragma unroll
      for (int t = 0; t < RT; ++t) f.w[t] = *reinterpret_cast<const h16x8*>(wrow[t] + (long)ks * 512);
      if constexpr (FOLD) {
        if (fw) {
#pragma unroll
          for (int t = 0; t < RT; ++t) f.wl[t] = *reinterpret_cast<const h16x8*>(wlrow[t] + (long)ks * 512);
        }
      }
      const int so = (int)(ks * a_step * 2);  // wave-uniform: the scalar offset of the k-step
      const u32x4 uh = __builtin_amdgcn_raw_buffer_load_b128(rs_hi, a_off, so, 0);
      const u32x4 ul = __builtin_amdgcn_raw_buffer_load_b128(rs_lo, a_off, so, 0);
      f.h = __builtin_bit_cast(h16x8, uh);
      f.l = __builtin_bit_cast(h16x8, ul);
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        acc[t] = AXW_MFMA_16x16x32(f.w[t], f.h, acc[t]);
        acc[t] = AXW_MFMA_16x16x32(f.w[t], f.l, acc[t]);
        if constexpr (FOLD) {
          if (fw) {  // M = M_hi + M_lo: the product matrix keeps 16 significant bits
            acc[t] = AXW_MFMA_16x16x32(f.wl[t], f.h, acc[t]);
            acc[t] = AXW_MFMA_16x16x32(f.wl[t], f.l, acc[t]);
          }
        }
      }
    };
    // eight k-steps in flight per wave (96 VGPRs at RT = 1): K = 5120 is 20 k-steps per wave, i.e. 3 dependent round
    // trips instead of 5 with four. (Twelve — all of mlp.2's K = 3072 in one round trip — measured the same at 4 clips, 6.04 us per
    // launch: bytes through the CU, not round trips; and 160 VGPRs leave one workgroup per CU where 114 leave two: 256 clips -2.7 %.)
    constexpr int DEPTH = FOLD ? 4 : 8;  // FOLD: K = d_model <= 1024 is at most 4 k-steps per wave
    Frag f[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
      if (wave + 8 * i < KS) load(f[i], wave + 8 * i);
    stamp_at(1);
    for (int ks = wave; ks < KS; ks += 8 * DEPTH) {
#pragma unroll
      for (int i = 0; i < DEPTH; ++i)
        if (ks + 8 * i < KS) {
          mma(f[i]);
          if (ks + 8 * (i + DEPTH) < KS) load(f[i], ks + 8 * (i + DEPTH));
        }
    }
  }

  // split-K reduction across the 8 waves: red[wave][t][clip r][row 4q + e]
  stamp_at(3);
#pragma unroll
  for (int t = 0; t < RT; ++t) *reinterpret_cast<f32x4*>(red + ((wave * RT + t) * 16 + r) * 16 + 4 * q) = acc[t];
  __syncthreads();
  stamp_at(4);
  if (!has_out) { stamp_at(0); return; }
  const int nl = tid % (16 * RT), bl = tid / (16 * RT);
  const int t = nl >> 4, nn = nl & 15;
  float y = bias_t;
#pragma unroll
  for (int w2 = 0; w2 < 8; ++w2) y += red[((w2 * RT + t) * 16 + bl) * 16 + nn];
  const int n = o_n, b = o_b;
  switch (p.epilogue) {
    case GEPI_STORE: p.out[(long)b * p.N + n] = y; break;
    case GEPI_RESID: {
      const float xn = old_t + y;
      if (fw) { p.out2[(long)b * (p.N - fr0) + (n - fr0)] = xn; break; }
      p.out[(long)b * nx + n] = xn;
      if (p.stat_part) {  // the 16 rows of this block for one clip sit in one DPP row (RT == 1: launch_decode_cgemm checks)
        const float s1 = row16_sum_f(xn);
        const float dm = xn - s1 * (1.f / 16.f);
        const float q2 = row16_sum_f(dm * dm);
        if (nn == 0) *reinterpret_cast<f32x2_t*>(p.stat_part + ((long)b * (nx >> 4) + blockIdx.x) * 2) = f32x2_t{s1, q2};
      }
      break;
    }
    case GEPI_GELU: store_pair_frag(gelu_erf(y), p.out_hi, p.out_lo, b, n, p.nbs); break;
    case GEPI_QKV_CACHE: {
      const int d = p.d_model;
      if (n < d) {
        p.out[(long)b * d + n] = y;
      } else if (n >= 3 * d) {  // query-fold rows: A0
        p.out2[(long)b * d + (n - 3 * d)] = y;
      } else {
        const int cc = (n < 2 * d) ? n - d : n - 2 * d;
        const int head = cc >> 6, dd = cc & 63;
        const long base = (long)b * p.kv_batch_stride + (long)head * p.n_ctx_pad * 64;
        if (n < 2 * d) p.k_cache[base + (long)(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (h16)y;
        else p.v_cache[base + (long)step * 64 + dd] = (h16)y;
      }
      break;
    }
  }
  stamp_at(0);
}

template <int RT>
static void launch_cg(const DecCGemmParams& p, hipStream_t s) {
  const dim3 grid((p.N + 16 * RT - 1) / (16 * RT), (p.batch + 15) / 16);
  const int ch = p.ln_w ? (p.K / 32 + 7) / 8 : 0;
  const bool fold_pair = ch == 0 && p.fold_row0 > 0;  // the o launch with its M rows
  bool ch_has_stamp = true;
  if (p.stamp) {  // measurement builds: the shapes of d_model 768
    if (fold_pair) hipLaunchKernelGGL((decode_cgemm_kernel<RT, 0, true, true>), grid, dim3(512), 0, s, p);
    else if (ch == 0) hipLaunchKernelGGL((decode_cgemm_kernel<RT, 0, true>), grid, dim3(512), 0, s, p);
    else if (ch == 3) hipLaunchKernelGGL((decode_cgemm_kernel<RT, 3, true>), grid, dim3(512), 0, s, p);
    else ch_has_stamp = false;  // no measurement build for this K: the production kernel runs, unstamped (a measurement switch never ends the process)
    if (ch_has_stamp) return;
  }
  if (fold_pair) { hipLaunchKernelGGL((decode_cgemm_kernel<RT, 0, false, true>), grid, dim3(512), 0, s, p); return; }
  switch (ch) {
    case 0: hipLaunchKernelGGL((decode_cgemm_kernel<RT, 0>), grid, dim3(512), 0, s, p); break;
    case 1: case 2: hipLaunchKernelGGL((decode_cgemm_kernel<RT, 2>), grid, dim3(512), 0, s, p); break;
    case 3: hipLaunchKernelGGL((decode_cgemm_kernel<RT, 3>), grid, dim3(512), 0, s, p); break;
    case 4: hipLaunchKernelGGL((decode_cgemm_kernel<RT, 4>), grid, dim3(512), 0, s, p); break;
    case 5: hipLaunchKernelGGL((decode_cgemm_kernel<RT, 5>), grid, dim3(512), 0, s, p); break;
    default: fprintf(stderr, "[ax_whisper] launch_decode_cgemm: LayerNorm prologue supports K <= 1280 (K=%d)\n", p.K); abort();
  }
}
void launch_decode_cgemm(const DecCGemmParams& p, hipStream_t s) {
  if (p.K % 128 != 0) { fprintf(stderr, "[ax_whisper] launch_decode_cgemm: unsupported K=%d\n", p.K); abort(); }
  if (p.fold_row0 > 0 && (p.fold_row0 % (16 * (p.rt == 2 ? 2 : 1)) != 0 || !p.out2 || (p.ln_w ? !p.ln_w2 : (!p.W_lo || p.K > 1024)) || (p.stat_part && p.rt == 2))) {
    fprintf(stderr, "[ax_whisper] launch_decode_cgemm: bad query-fold launch (fold_row0=%d, rt=%d)\n", p.fold_row0, p.rt);
    abort();
  }
  if (p.rt == 2) launch_cg<2>(p, s);
  else launch_cg<1>(p, s);
}

// ---------------------------------------------------------------------------- vocabulary projection (rt == 0)
// The tied-embedding logits of up to 64 clips: 80 MB of weights against 196 KB of activations. One workgroup per CU
// stays for its share of the 16-row weight blocks: every wave keeps the activation fragments of ITS k-steps (all clip
// blocks, hi and lo) in registers for the whole launch, so an iteration streams nothing but 16 rows of weights
// (requested one iteration ahead), and a thread keeps the running argmax of its (clip, row lane) over all iterations
// (rows ascend, strict >, so the first maximum wins as in Whisper.cpp:42-45). The launch-per-row-block form
// (decode_gemm_kernel<2, 4>: 1621 workgroups, one resident per CU, each re-reading all activations and paying its own
// load -> MFMA -> reduce -> epilogue chain) took 60 us per step at 64 clips.
template <int NB, int CH>
__global__ __launch_bounds__(512) void decode_logits_kernel(DecGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [2][8 waves][NB][16 clips][16 rows]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: k-step selection is wave-uniform
  const int r = lane & 15, q = lane >> 4;
  const int KS = p.K / 32, n_rb = (p.N + 15) / 16, G = gridDim.x;

  h16x8 ah[CH][NB], al[CH][NB];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const long ks = min(wave + 8 * c, KS - 1);
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      ah[c][cb] = *reinterpret_cast<const h16x8*>(p.a_hi + (ks * p.nbs + cb) * 512 + lane * 8);
      al[c][cb] = *reinterpret_cast<const h16x8*>(p.a_lo + (ks * p.nbs + cb) * 512 + lane * 8);
    }
  }
  auto loadw = [&](h16x8 (&w)[CH], int rb) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
      w[c] = *reinterpret_cast<const h16x8*>(p.W + ((long)min(rb, n_rb - 1) * KS + min(wave + 8 * c, KS - 1)) * 512 + lane * 8);
  };
  // three register sets: the blocks of the next TWO iterations are in flight while one is multiplied (with one block
  // ahead an iteration was one memory round trip: 25 us for 80 MB at 64 clips). loadw clamps the block index, so the
  // requests are unconditional (a conditional request joins old and new registers at the loop's end, which hipcc
  // resolves with a full drain); the tail re-reads the last block (L2 hits).
  h16x8 w0[CH], w1[CH], w2[CH];
  loadw(w0, blockIdx.x);
  {  // SOT steps: logits are discarded (Whisper.cpp:214-217): nothing to do while every clip is below skip_before_step
    const int o = lane < p.batch ? p.off[lane] : -1;
    if (__ballot(o >= p.skip_before_step) == 0) return;
  }

  constexpr int NU = (NB * 256 + 511) / 512;  // outputs per thread and iteration
  float bv[NU];
  int bi[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) { bv[u] = -INFINITY; bi[u] = 0x7fffffff; }

  auto body = [&](const h16x8 (&w)[CH], int rb, int it) {
    f32x4 acc[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[cb][e] = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (wave + 8 * c < KS) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
          acc[cb] = AXW_MFMA_16x16x32(w[c], ah[c][cb], acc[cb]);
          acc[cb] = AXW_MFMA_16x16x32(w[c], al[c][cb], acc[cb]);
        }
      }
    float* rd = red + (it & 1) * (8 * NB * 256);  // two buffers: a wave may park iteration i+1 while others still sum i
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) *reinterpret_cast<f32x4*>(rd + ((wave * NB + cb) * 16 + r) * 16 + 4 * q) = acc[cb];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int o = tid + 512 * u;
      if (o < NB * 256) {
        const int cb = o >> 8, bl = (o >> 4) & 15, nn = o & 15;
        float y = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 8; ++w2) y += rd[((w2 * NB + cb) * 16 + bl) * 16 + nn];
        const int n = rb * 16 + nn, b = cb * 16 + bl;
        if (n < p.N && b < p.batch) {
          if (p.logits_dump) p.logits_dump[(long)b * p.logits_dump_stride + n] = y;
          if (y > bv[u]) { bv[u] = y; bi[u] = n; }
        }
      }
    }
  };
  int it = 0;
  loadw(w1, blockIdx.x + G);
  for (int rb = blockIdx.x; rb < n_rb; rb += 3 * G) {
    loadw(w2, rb + 2 * G);
    body(w0, rb, it++);
    if (rb + G >= n_rb) break;
    loadw(w0, rb + 3 * G);
    body(w1, rb + G, it++);
    if (rb + 2 * G >= n_rb) break;
    loadw(w1, rb + 4 * G);
    body(w2, rb + 2 * G, it++);
  }
  // the 16 lanes that share a clip hold its candidates of different row lanes: lowest index wins ties
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    {  // DPP butterflies inside the 16-lane row (the __shfl_xor form: eight dependent ds_bpermute round trips at the launch's end)
      auto take = [&](float ov, int oi) { if (ov > bv[u] || (ov == bv[u] && oi < bi[u])) { bv[u] = ov; bi[u] = oi; } };
#define AXW_ROW_STEP(CTRL) { const float ov = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bv[u]), CTRL, 0xf, 0xf, true)); \
                             const int oi = __builtin_amdgcn_update_dpp(0, bi[u], CTRL, 0xf, 0xf, true); take(ov, oi); }
      AXW_ROW_STEP(0xB1)   // quad_perm [1,0,3,2]
      AXW_ROW_STEP(0x4E)   // quad_perm [2,3,0,1]
      AXW_ROW_STEP(0x141)  // row_half_mirror
      AXW_ROW_STEP(0x140)  // row_mirror
#undef AXW_ROW_STEP
    }
    const int o = tid + 512 * u, b = o >> 4;
    if (o < NB * 256 && (o & 15) == 0 && b < p.batch) {
      p.amax_val[(long)b * p.amax_stride + blockIdx.x] = bv[u];
      p.amax_idx[(long)b * p.amax_stride + blockIdx.x] = bi[u];
    }
  }
}

template <int CH>
static void launch_logits(const DecGemmParams& p, hipStream_t s) {
  const int nb = (p.batch + 15) / 16;
  const dim3 grid(decode_gemm_grid(p.N, 0));
  const size_t lds = (size_t)2 * 8 * nb * 256 * 4;
  if (nb == 1) hipLaunchKernelGGL((decode_logits_kernel<1, CH>), grid, dim3(512), lds, s, p);
  if constexpr (CH * 2 <= 15) { if (nb == 2) hipLaunchKernelGGL((decode_logits_kernel<2, CH>), grid, dim3(512), lds, s, p); }
  if constexpr (CH * 3 <= 15) { if (nb == 3) hipLaunchKernelGGL((decode_logits_kernel<3, CH>), grid, dim3(512), lds, s, p); }
  if constexpr (CH * 4 <= 15) { if (nb >= 4) hipLaunchKernelGGL((decode_logits_kernel<4, CH>), grid, dim3(512), lds, s, p); }
}
// register-resident activations: k-steps per wave x clip blocks x (hi, lo) x 4 VGPRs (at most 15 x 8 = 120)
bool decode_logits_resident_ok(int K, int batch) {
  const int ch = (K / 32 + 7) / 8, nb = (std::min(batch, 64) + 15) / 16;
  return K % 128 == 0 && ch <= 5 && ch * nb <= 15;
}

int decode_gemm_grid(int N, int rt) {
  if (rt == 0) return std::min((N + 15) / 16, 256); return (N + 16 * rt - 1) / (16 * rt); }

template <int RT>
static void launch_nb(const DecGemmParams& p, hipStream_t s) {
  const int nb = (p.batch + 15) / 16;
  const dim3 grid(decode_gemm_grid(p.N, RT), p.epilogue == GEPI_PARTIAL ? p.ksplit : 1);
  const size_t lds = (size_t)(8 * RT * nb * 256 + (p.epilogue == GEPI_LOGITS ? 16 * RT * nb * 16 : 0)) * 4;
  switch (nb) {
    case 1: hipLaunchKernelGGL((decode_gemm_kernel<RT, 1>), grid, dim3(512), lds, s, p); break;
    case 2: hipLaunchKernelGGL((decode_gemm_kernel<RT, 2>), grid, dim3(512), lds, s, p); break;
    case 3: hipLaunchKernelGGL((decode_gemm_kernel<RT, 3>), grid, dim3(512), lds, s, p); break;
    case 4: hipLaunchKernelGGL((decode_gemm_kernel<RT, 4>), grid, dim3(512), lds, s, p); break;
    default: fprintf(stderr, "[ax_whisper] launch_decode_gemm: batch %d > 64 per launch\n", p.batch); abort();
  }
}

// p.batch <= 64 per launch (the engine tiles larger batches); K % 128 == 0.
void launch_decode_gemm(const DecGemmParams& p, hipStream_t s) {
  if (p.K % 128 != 0) { fprintf(stderr, "[ax_whisper] launch_decode_gemm: unsupported K=%d\n", p.K); abort(); }
  const int rt = p.rt;
  if (rt == 0) {  // vocabulary projection with register-resident activations
    if (p.epilogue != GEPI_LOGITS || !decode_logits_resident_ok(p.K, p.batch) || p.batch > 64) { fprintf(stderr, "[ax_whisper] launch_decode_gemm: rt 0 is the vocabulary projection with register-resident activations (K %d, batch %d unsupported)\n", p.K, p.batch); abort(); }
    switch ((p.K / 32 + 7) / 8) {
      case 1: launch_logits<1>(p, s); break;
      case 2: launch_logits<2>(p, s); break;
      case 3: launch_logits<3>(p, s); break;
      case 4: launch_logits<4>(p, s); break;
      default: launch_logits<5>(p, s); break;
    }
    return;
  }
  if (rt == 4) launch_nb<4>(p, s);
  else if (rt == 2) launch_nb<2>(p, s);
  else launch_nb<1>(p, s);
}

}  // inline namespace AXW_NS
}  // namespace axw
