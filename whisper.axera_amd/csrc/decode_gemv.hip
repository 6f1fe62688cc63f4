// decode_gemv.hip — weight-streaming GEMV of the decoder step for 1..4 clips, and the loop-advance kernel.
//
// y[b][n] = sum_k W[n][k] * a[b][k] (+ bias[n]) with W h16 [N][K] read exactly once, activations fp32.
// This is where the decoder's nn.Linear layers run at small batch (export_onnx.py:238-261 q/k/v/out,
// :221-230 cross query/out, :298 mlp, :378-385 tied-embedding logits); decode is HBM-bound, the FLOPs
// are free, so the products stay fp32 FMA (h16 weight x fp32 activation is exact in fp32).
//
// Latency structure (batch 1 is a chain of ~100 dependent launches per token, each a few microseconds):
//   1. every lane issues the 16-byte weight loads of its first row BEFORE the prologue, so the HBM round
//      trip overlaps the activation work;
//   2. prologue builds the activation rows in LDS with all 256 threads: LayerNorm of the residual stream
//      (single read, shifted one-pass variance, one block reduction), or the merge of the attention split
//      partials, or a plain copy;
//   3. LPR lanes share a weight row (LPR*16 contiguous bytes per load instruction, full cache lines), each
//      lane owns CH chunks; the next row's loads are issued before the current row is reduced;
//   4. fused epilogues: bias, GELU, residual add, q + self-KV cache append (blocked K / row-major V),
//      vocabulary argmax partials (first max wins, Whisper.cpp:42-45).
#include "common.hpp"

namespace axw {
inline namespace AXW_NS {

constexpr int kPartStride = 66;  // m, l, o[64]

template <int BT>
__device__ __forceinline__ void prologue_layernorm(const GemvParams& p, float* act, float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K;
  constexpr int MAXE = 8;  // K <= 2048
  float v[BT][MAXE], g[MAXE], be[MAXE];
  float s1[BT], s2[BT], shift[BT];
  // every global load of the prologue is issued here, before the first reduction
#pragma unroll
  for (int b = 0; b < BT; ++b) shift[b] = b < p.batch ? p.in[(long)b * K] : 0.f;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int c = tid + 256 * e;
    g[e] = c < K ? p.ln_w[c] : 0.f;
    be[e] = c < K ? p.ln_b[c] : 0.f;
#pragma unroll
    for (int b = 0; b < BT; ++b) v[b][e] = (c < K && b < p.batch) ? p.in[(long)b * K + c] : shift[b];
  }
#pragma unroll
  for (int b = 0; b < BT; ++b) {  // shifted one-pass variance: no cancellation for data near `shift`
    s1[b] = 0.f; s2[b] = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) { const float t = v[b][e] - shift[b]; s1[b] += t; s2[b] += t * t; }
    s1[b] = wave_sum(s1[b]); s2[b] = wave_sum(s2[b]);
  }
  if (lane == 0) {
#pragma unroll
    for (int b = 0; b < BT; ++b) { red[(wave * BT + b) * 2] = s1[b]; red[(wave * BT + b) * 2 + 1] = s2[b]; }
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < BT; ++b) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { t1 += red[(w * BT + b) * 2]; t2 += red[(w * BT + b) * 2 + 1]; }
    const float dm = t1 / K;  // mean - shift
    const float var = fmaxf(t2 / K - dm * dm, 0.f);
    const float mean = shift[b] + dm, rstd = rsqrtf(var + 1e-5f);
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int c = tid + 256 * e;
      if (c < K) act[b * K + c] = b < p.batch ? (v[b][e] - mean) * rstd * g[e] + be[e] : 0.f;
    }
  }
}

template <int BT>
__device__ __forceinline__ void prologue_attn_combine(const GemvParams& p, float* act) {
  const int K = p.K;
  constexpr int MAXS = 8;  // n_split <= 8
  for (int i = threadIdx.x; i < BT * K; i += 256) {
    const int b = i / K, c = i - b * K;
    float v = 0.f;
    if (b < p.batch) {
      const float* pp = p.part + ((long)b * p.n_head + (c >> 6)) * p.n_split * kPartStride;
      float ms[MAXS], ls[MAXS], os[MAXS];
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {  // independent loads, one round trip
        const bool on = s < p.n_split;
        ms[s] = on ? pp[s * kPartStride] : -INFINITY;
        ls[s] = on ? pp[s * kPartStride + 1] : 0.f;
        os[s] = on ? pp[s * kPartStride + 2 + (c & 63)] : 0.f;
      }
      float m = ms[0];
#pragma unroll
      for (int s = 1; s < MAXS; ++s) m = fmaxf(m, ms[s]);
      float l = 0.f, o = 0.f;
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {
        const float w = __expf(ms[s] - m);
        l += w * ls[s];
        o += w * os[s];
      }
      v = o / l;
    }
    act[i] = v;
  }
}

template <int LPR, int BT, int CH>
__global__ __launch_bounds__(256) void gemv_kernel(GemvParams p, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* act = reinterpret_cast<float*>(smem);  // [BT][K]
  __shared__ float s_red[4 * BT * 2];
  __shared__ float s_val[4 * BT];
  __shared__ int s_idx[4 * BT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K;

  constexpr int RP = 256 / LPR;  // rows per pass
  const int j = tid % LPR, rsub = tid / LPR;
  const int row_begin = blockIdx.x * rows_per_wg;
  const int row_end = min(p.N, row_begin + rows_per_wg);

  // ---- 1. first row's weights: issued before anything that depends on the activations
  uint4 wnext[CH];
  {
    const int n = min(row_begin + rsub, row_end - 1);
    const h16* wrow = p.W + (long)n * K;
#pragma unroll
    for (int i = 0; i < CH; ++i) wnext[i] = *reinterpret_cast<const uint4*>(wrow + (j + LPR * i) * 8);
  }

  // The clips' offsets live in device memory (written by the previous step's advance kernel); they are read
  // only AFTER the weight loads are in flight so their round trip does not delay them.
  int stepb[BT];
  bool any_logits = false;
#pragma unroll
  for (int b = 0; b < BT; ++b) {
    stepb[b] = ((p.epilogue == GEPI_LOGITS || p.epilogue == GEPI_QKV_CACHE) && b < p.batch) ? p.off[b] : 0;
    any_logits |= b < p.batch && stepb[b] >= p.skip_before_step;
  }
  if (p.epilogue == GEPI_LOGITS && !any_logits) return;  // SOT steps: logits are discarded (Whisper.cpp:214-217)

  // epilogue operands of the first pass (bias, residual) ride along with the weight loads
  float bias0 = 0.f, resid0[BT];
  {
    const int n = row_begin + rsub;
    const bool on = j == 0 && n < row_end;
    bias0 = (on && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
    for (int b = 0; b < BT; ++b) resid0[b] = (on && p.epilogue == GEPI_RESID && b < p.batch) ? p.out[(long)b * p.N + n] : 0.f;
  }

  // ---- 2. prologue: activation rows -> LDS
  if (p.prologue == PRO_LAYERNORM) {
    prologue_layernorm<BT>(p, act, s_red);
  } else if (p.prologue == PRO_ATTN_COMBINE) {
    prologue_attn_combine<BT>(p, act);
  } else {
    const int n4 = BT * K / 4, lim = p.batch * K / 4;
    for (int i = tid; i < n4; i += 256)
      reinterpret_cast<float4*>(act)[i] = i < lim ? reinterpret_cast<const float4*>(p.in)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  float best_v[BT];
  int best_i[BT];
#pragma unroll
  for (int b = 0; b < BT; ++b) { best_v[b] = -INFINITY; best_i[b] = 0x7fffffff; }

  // ---- 3. rows
  for (int row0 = row_begin; row0 < row_end; row0 += RP) {
    const int n = row0 + rsub;
    const bool valid = n < row_end;
    uint4 w[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) w[i] = wnext[i];
    if (row0 + RP < row_end) {  // next pass in flight while this one is reduced
      const int nn = min(row0 + RP + rsub, row_end - 1);
      const h16* wrow = p.W + (long)nn * K;
#pragma unroll
      for (int i = 0; i < CH; ++i) wnext[i] = *reinterpret_cast<const uint4*>(wrow + (j + LPR * i) * 8);
    }
    float acc[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[b] = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c8 = (j + LPR * i) * 8;
      const unsigned uw[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        const float4 a0 = *reinterpret_cast<const float4*>(act + b * K + c8);
        const float4 a1 = *reinterpret_cast<const float4*>(act + b * K + c8 + 4);
        acc[b] = fmaf(h16lo(uw[0]), a0.x, acc[b]);
        acc[b] = fmaf(h16hi(uw[0]), a0.y, acc[b]);
        acc[b] = fmaf(h16lo(uw[1]), a0.z, acc[b]);
        acc[b] = fmaf(h16hi(uw[1]), a0.w, acc[b]);
        acc[b] = fmaf(h16lo(uw[2]), a1.x, acc[b]);
        acc[b] = fmaf(h16hi(uw[2]), a1.y, acc[b]);
        acc[b] = fmaf(h16lo(uw[3]), a1.z, acc[b]);
        acc[b] = fmaf(h16hi(uw[3]), a1.w, acc[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < BT; ++b)
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) acc[b] += __shfl_xor(acc[b], o, 64);

    if (j == 0 && valid) {
      const bool first = row0 == row_begin;
      const float bias = first ? bias0 : (p.bias ? p.bias[n] : 0.f);
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        if (b >= p.batch) break;
        const float y = acc[b] + bias;
        switch (p.epilogue) {
          case GEPI_STORE: p.out[(long)b * p.N + n] = y; break;
          case GEPI_GELU: p.out[(long)b * p.N + n] = gelu_erf(y); break;
          case GEPI_RESID: p.out[(long)b * p.N + n] = (first ? resid0[b] : p.out[(long)b * p.N + n]) + y; break;
          case GEPI_QKV_CACHE: {
            const int d = p.d_model;
            if (n < d) {
              p.out[(long)b * d + n] = y;
            } else {
              const int c = (n < 2 * d) ? n - d : n - 2 * d;
              const int head = c >> 6, dd = c & 63;
              const long base = (long)b * p.kv_batch_stride + (long)head * p.n_ctx_pad * 64;
              const int step = stepb[b];
              if (n < 2 * d)  // blocked K: [blk][dd/8][key%64][8]
                p.k_cache[base + (long)(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (h16)y;
              else            // row-major V: [key][64]
                p.v_cache[base + (long)step * 64 + dd] = (h16)y;
            }
            break;
          }
          case GEPI_LOGITS:
            if (p.logits_dump) p.logits_dump[(long)b * p.logits_dump_stride + n] = y;
            if (y > best_v[b] || (y == best_v[b] && n < best_i[b])) { best_v[b] = y; best_i[b] = n; }
            break;
        }
      }
    }
  }

  if (p.epilogue == GEPI_LOGITS) {  // workgroup argmax, first max wins (Whisper.cpp:42-45)
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      float v = best_v[b];
      int ix = best_i[b];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(ix, o, 64);
        if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
      }
      if (lane == 0) { s_val[wave * BT + b] = v; s_idx[wave * BT + b] = ix; }
    }
    __syncthreads();
    if (tid < BT && tid < p.batch) {
      float v = s_val[tid];
      int ix = s_idx[tid];
      for (int w = 1; w < 4; ++w) {
        const float ov = s_val[w * BT + tid];
        const int oi = s_idx[w * BT + tid];
        if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
      }
      p.amax_val[(long)tid * p.amax_stride + blockIdx.x] = v;
      p.amax_idx[(long)tid * p.amax_stride + blockIdx.x] = ix;
    }
  }
}

// ------------------------------------------------------------------------------- single-clip GEMV
// Batch 1 is a chain of ~100 dependent launches, so what matters is the length of each launch's critical path.
// With 8*LPR*CH == K the LPR lanes that share a weight row hold, between them, the WHOLE activation vector in
// exactly the chunks they multiply: the prologue needs no LDS, no barrier and no block reduction — LayerNorm
// statistics are an LPR-lane shuffle reduction done redundantly by every row group, and the activations stay in
// registers across all rows of the workgroup. Chain: [all global loads issued] -> shuffles -> FMAs -> shuffles -> store.
template <int LPR, int CH, int PRO>
__global__ __launch_bounds__(256) void gemv1_kernel(GemvParams p, int rows_per_wg) {
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K;
  constexpr int RP = 256 / LPR;
  const int j = tid % LPR, rsub = tid / LPR;
  const int row_begin = blockIdx.x * rows_per_wg;
  const int row_end = min(p.N, row_begin + rows_per_wg);

  uint4 wnext[CH];
  {
    const int n = min(row_begin + rsub, row_end - 1);
    const h16* wrow = p.W + (long)n * K;
#pragma unroll
    for (int i = 0; i < CH; ++i) wnext[i] = *reinterpret_cast<const uint4*>(wrow + (j + LPR * i) * 8);
  }
  float bias0 = 0.f, resid0 = 0.f;
  {
    const int n = row_begin + rsub;
    const bool on = j == 0 && n < row_end;
    bias0 = (on && p.bias) ? p.bias[n] : 0.f;
    resid0 = (on && p.epilogue == GEPI_RESID) ? p.out[n] : 0.f;
  }

  // ---- activations of this lane's chunks, in registers
  float a[CH][8];
  if constexpr (PRO == PRO_LAYERNORM) {
    float g[CH][8], be[CH][8];
    const float shift = p.in[0];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c8 = (j + LPR * i) * 8;
      const float4 x0 = *reinterpret_cast<const float4*>(p.in + c8), x1 = *reinterpret_cast<const float4*>(p.in + c8 + 4);
      const float4 g0 = *reinterpret_cast<const float4*>(p.ln_w + c8), g1 = *reinterpret_cast<const float4*>(p.ln_w + c8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.ln_b + c8), b1 = *reinterpret_cast<const float4*>(p.ln_b + c8 + 4);
      a[i][0] = x0.x; a[i][1] = x0.y; a[i][2] = x0.z; a[i][3] = x0.w; a[i][4] = x1.x; a[i][5] = x1.y; a[i][6] = x1.z; a[i][7] = x1.w;
      g[i][0] = g0.x; g[i][1] = g0.y; g[i][2] = g0.z; g[i][3] = g0.w; g[i][4] = g1.x; g[i][5] = g1.y; g[i][6] = g1.z; g[i][7] = g1.w;
      be[i][0] = b0.x; be[i][1] = b0.y; be[i][2] = b0.z; be[i][3] = b0.w; be[i][4] = b1.x; be[i][5] = b1.y; be[i][6] = b1.z; be[i][7] = b1.w;
    }
    float s1 = 0.f, s2 = 0.f;  // shifted one-pass variance
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = a[i][e] - shift; s1 += t; s2 += t * t; }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const float dm = s1 / K, var = fmaxf(s2 / K - dm * dm, 0.f);
    const float mean = shift + dm, rstd = rsqrtf(var + 1e-5f);
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[i][e] = (a[i][e] - mean) * rstd * g[i][e] + be[i][e];
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c8 = (j + LPR * i) * 8;
      const float4 x0 = *reinterpret_cast<const float4*>(p.in + c8), x1 = *reinterpret_cast<const float4*>(p.in + c8 + 4);
      a[i][0] = x0.x; a[i][1] = x0.y; a[i][2] = x0.z; a[i][3] = x0.w; a[i][4] = x1.x; a[i][5] = x1.y; a[i][6] = x1.z; a[i][7] = x1.w;
    }
  }

  const int step = (p.epilogue == GEPI_LOGITS || p.epilogue == GEPI_QKV_CACHE) ? p.off[0] : 0;  // the one clip of this launch
  if (p.epilogue == GEPI_LOGITS && step < p.skip_before_step) return;  // SOT steps: logits are discarded (Whisper.cpp:214-217)

  float best_v = -INFINITY;
  int best_i = 0x7fffffff;
  for (int row0 = row_begin; row0 < row_end; row0 += RP) {
    const int n = row0 + rsub;
    const bool valid = n < row_end;
    uint4 w[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) w[i] = wnext[i];
    if (row0 + RP < row_end) {
      const int nn = min(row0 + RP + rsub, row_end - 1);
      const h16* wrow = p.W + (long)nn * K;
#pragma unroll
      for (int i = 0; i < CH; ++i) wnext[i] = *reinterpret_cast<const uint4*>(wrow + (j + LPR * i) * 8);
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const unsigned uw[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = fmaf(h16lo(uw[e]), a[i][2 * e], acc);
        acc = fmaf(h16hi(uw[e]), a[i][2 * e + 1], acc);
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (j == 0 && valid) {
      const bool first = row0 == row_begin;
      const float y = acc + (first ? bias0 : (p.bias ? p.bias[n] : 0.f));
      switch (p.epilogue) {
        case GEPI_STORE: p.out[n] = y; break;
        case GEPI_GELU: p.out[n] = gelu_erf(y); break;
        case GEPI_RESID: p.out[n] = (first ? resid0 : p.out[n]) + y; break;
        case GEPI_QKV_CACHE: {
          const int d = p.d_model;
          if (n < d) {
            p.out[n] = y;
          } else {
            const int c = (n < 2 * d) ? n - d : n - 2 * d;
            const int head = c >> 6, dd = c & 63;
            const long base = (long)head * p.n_ctx_pad * 64;
            if (n < 2 * d) p.k_cache[base + (long)(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (h16)y;
            else p.v_cache[base + (long)step * 64 + dd] = (h16)y;
          }
          break;
        }
        case GEPI_LOGITS:
          if (p.logits_dump) p.logits_dump[n] = y;
          if (y > best_v || (y == best_v && n < best_i)) { best_v = y; best_i = n; }
          break;
      }
    }
  }
  if (p.epilogue == GEPI_LOGITS) {  // workgroup argmax, first max wins (Whisper.cpp:42-45)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best_v, o, 64);
      const int oi = __shfl_xor(best_i, o, 64);
      if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
    }
    if (lane == 0) { s_val[wave] = best_v; s_idx[wave] = best_i; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (s_val[w] > best_v || (s_val[w] == best_v && s_idx[w] < best_i)) { best_v = s_val[w]; best_i = s_idx[w]; }
      p.amax_val[blockIdx.x] = best_v;
      p.amax_idx[blockIdx.x] = best_i;
    }
  }
}

static int pick_lpr(int K) {
  // widest lane group whose per-lane chunk count stays small
  if (K % 512 == 0 && K / 512 <= 10 && K >= 1024) return 64;
  if (K % 256 == 0 && K / 256 <= 6) return 32;
  if (K % 128 == 0 && K / 128 <= 10) return 16;
  return 0;
}

static int rows_per_wg_for(int N, int lpr) {
  const int rp = 256 / lpr;
  const int target_wgs = 2048;  // small N: one pass per workgroup; vocabulary: a few pipelined passes
  int rows = (N + target_wgs - 1) / target_wgs;
  rows = ((rows + rp - 1) / rp) * rp;
  return rows < rp ? rp : rows;
}

int gemv_grid(const GemvParams& p) {
  const int lpr = pick_lpr(p.K);
  if (!lpr) return 0;
  const int rpw = rows_per_wg_for(p.N, lpr);
  return (p.N + rpw - 1) / rpw;
}

template <int LPR>
static bool launch_gemv1_ch(const GemvParams& p, int rpw, int grid, hipStream_t s) {
  const int ch = p.K / (8 * LPR);
#define AXW_GEMV1_CASE(C)                                                                                       \
  case C:                                                                                                      \
    if (p.prologue == PRO_LAYERNORM) hipLaunchKernelGGL((gemv1_kernel<LPR, C, PRO_LAYERNORM>), dim3(grid), dim3(256), 0, s, p, rpw); \
    else hipLaunchKernelGGL((gemv1_kernel<LPR, C, PRO_PLAIN>), dim3(grid), dim3(256), 0, s, p, rpw);            \
    return true;
  switch (ch) {
    AXW_GEMV1_CASE(1) AXW_GEMV1_CASE(2) AXW_GEMV1_CASE(3) AXW_GEMV1_CASE(4) AXW_GEMV1_CASE(5) AXW_GEMV1_CASE(6) AXW_GEMV1_CASE(8)
    AXW_GEMV1_CASE(10)
    default: return false;
  }
#undef AXW_GEMV1_CASE
}

template <int LPR, int BT>
static bool launch_gemv_ch(const GemvParams& p, int rpw, int grid, hipStream_t s) {
  const int ch = p.K / (8 * LPR);
  const size_t lds = (size_t)BT * p.K * 4;
#define AXW_GEMV_CASE(C) \
  case C: hipLaunchKernelGGL((gemv_kernel<LPR, BT, C>), dim3(grid), dim3(256), lds, s, p, rpw); return true;
  switch (ch) {
    AXW_GEMV_CASE(1) AXW_GEMV_CASE(2) AXW_GEMV_CASE(3) AXW_GEMV_CASE(4) AXW_GEMV_CASE(5) AXW_GEMV_CASE(6) AXW_GEMV_CASE(8)
    AXW_GEMV_CASE(10)
    default: return false;
  }
#undef AXW_GEMV_CASE
}

// Handles p.batch <= 4 per launch; the engine tiles larger batches. K must be one of the supported
// widths (multiples of 128 with <= 10 chunks per lane): every Whisper size is.
void launch_gemv(const GemvParams& p, hipStream_t s) {
  const int lpr = pick_lpr(p.K);
  bool ok = lpr != 0;
  if (ok) {
    const int rpw = rows_per_wg_for(p.N, lpr);
    const int grid = (p.N + rpw - 1) / rpw;
    const bool one = p.batch == 1 && p.prologue != PRO_ATTN_COMBINE;  // register-resident single-clip kernel
    const bool one_lds = p.batch == 1;
    switch (lpr) {
      case 64: ok = one ? launch_gemv1_ch<64>(p, rpw, grid, s) : (one_lds ? launch_gemv_ch<64, 1>(p, rpw, grid, s) : launch_gemv_ch<64, 4>(p, rpw, grid, s)); break;
      case 32: ok = one ? launch_gemv1_ch<32>(p, rpw, grid, s) : (one_lds ? launch_gemv_ch<32, 1>(p, rpw, grid, s) : launch_gemv_ch<32, 4>(p, rpw, grid, s)); break;
      default: ok = one ? launch_gemv1_ch<16>(p, rpw, grid, s) : (one_lds ? launch_gemv_ch<16, 1>(p, rpw, grid, s) : launch_gemv_ch<16, 4>(p, rpw, grid, s)); break;
    }
  }
  if (!ok) {
    fprintf(stderr, "[ax_whisper] launch_gemv: unsupported K=%d\n", p.K);
    abort();  // fail loudly: there is no fallback path
  }
}

// ------------------------------------------------------------------------------- advance
// Whisper.cpp:207-222: steps 0..2 feed the next SOT token and drop the logits; from step 3 on the
// argmax is either the stop condition (eot / context full) or the next recorded + fed token.
// One wave per clip merges the per-workgroup argmax partials (first max wins); 16 clips per workgroup. Every clip
// advances its OWN offset (the reference decodes one utterance at a time and stops it at its own eot,
// Whisper.cpp:219-222): a finished clip stays where it is.
// One wave per clip. The kernel is one dependent chain per decoder step (step counter -> argmax partials -> loop state ->
// embedding row of the chosen token), so everything that does not depend on the previous link is requested early: the
// loop state and the next position's embedding beside the partials, the step ticket right behind the step counter, and
// the embedding row in 8-byte pieces that are all in flight before the first store (as `for (c = lane; c < d; c += 64)`
// the row was 12 load -> add -> store round trips, 12.7 us per step at turbo dims).
__global__ __launch_bounds__(1024) void advance_kernel(AdvanceParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 16 + wave;
  // the engine-wide step counter is bookkeeping only (steps run since the reset): whichever workgroup draws the last
  // ticket bumps it. Positions come from the clips' own offsets.
  if (threadIdx.x == 0) {
    const int ticket = atomicAdd(&p.state->pad0, 1);
    if (ticket == (int)gridDim.x - 1) {
      p.state->pad0 = 0;
      p.state->step = p.state->step + 1;
    }
  }
  if (b >= p.batch) return;
  const int s = p.off[b];  // this clip's offset: the position that was fed in this step
  constexpr int MAXJ = 8;  // d_model <= 2048: lane l owns elements 4l + 256j
  const int d = p.d_model;
  // x of every slot is re-seeded every step, finished slots included (their rows of the linear layers keep running and
  // must stay finite): a slot at the end of the context re-reads the last position row
  const bool room = s + 1 < p.n_ctx;
  const int pos_row = room ? s + 1 : p.n_ctx - 1;
  float4 pv[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
    if (4 * lane + 256 * j < d) pv[j] = *reinterpret_cast<const float4*>(p.pos + (long)pos_row * d + 4 * lane + 256 * j);
  // loop state of this clip: requested beside the partials, used after the reduction
  const int tok_old = p.tok[b];
  const bool greedy = !p.forced;
  const int done_b = greedy ? p.done[b] : 0;
  int tok = tok_old;
  bool advance = !done_b;  // a finished (or idle) slot keeps its offset: its cache row and position stay in bounds for good
  if (done_b) {
    // nothing: the slot waits for Engine::stream_admit or the end of the batch
  } else if (s < 3) {
    tok = p.sot[s + 1];
    if (lane == 0) p.tok[b] = tok;
  } else {
    const int n_out = greedy ? p.n_out[b] : 0;
    const int max_new = greedy ? (p.max_new_clip ? min(p.max_new_clip[b], p.max_new) : p.max_new) : 0;
    float v = -INFINITY;
    int idx = 0x7fffffff;
#pragma unroll 4
    for (int i = lane; i < p.n_part; i += 64) {
      const float ov = p.amax_val[(long)b * p.amax_stride + i];
      const int oi = p.amax_idx[(long)b * p.amax_stride + i];
      if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    wave_argmax(v, idx);
    // no logit compared greater than -inf (all NaN / -inf: non-finite audio): std::max_element returns index 0
    // (Whisper.cpp:42-45); never let the "no candidate" index reach the embedding lookup below
    if ((unsigned)idx >= (unsigned)p.n_vocab) idx = 0;
    const int gi = s - 3;
    if (p.forced) {
      if (gi < p.n_forced) tok = p.forced[(long)b * p.n_forced + gi];
    } else {
      if (idx == p.eot || s + 1 >= p.n_ctx || n_out >= max_new) {  // Whisper.cpp:219-222: this utterance ends HERE
        if (lane == 0) {
          p.done[b] = 1;
          atomicAdd(&p.state->n_done, 1);
          if (p.done_host) {  // ids and count of this clip are final: make them visible before the host can see the flag
            __threadfence_system();
            __hip_atomic_store(p.done_host + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        advance = false;
      } else {
        if (lane == 0) { p.out_ids[(long)b * p.n_ctx + n_out] = idx; p.n_out[b] = n_out + 1; }
        tok = idx;
      }
    }
    if (lane == 0) {
      if (p.argmax_dump && gi <= p.n_forced) p.argmax_dump[(long)b * (p.n_forced + 1) + gi] = idx;
      p.tok[b] = tok;
    }
  }
  if (lane == 0 && advance && room) p.off[b] = s + 1;
  // fused embedding of the NEXT step: x = tok_emb[token] + pos[offset + 1]   (export_onnx.py:334-336)
  {
    h16x4 ev[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
      if (4 * lane + 256 * j < d) ev[j] = *reinterpret_cast<const h16x4*>(p.tok_emb + (long)tok * d + 4 * lane + 256 * j);
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
      if (4 * lane + 256 * j < d) {
        float4 o;
        o.x = (float)ev[j][0] + pv[j].x; o.y = (float)ev[j][1] + pv[j].y; o.z = (float)ev[j][2] + pv[j].z; o.w = (float)ev[j][3] + pv[j].w;
        *reinterpret_cast<float4*>(p.x + (long)b * d + 4 * lane + 256 * j) = o;
      }
  }
}

void launch_advance(const AdvanceParams& p, hipStream_t s) {
  if (p.d_model > 2048 || p.d_model % 4 != 0) { fprintf(stderr, "[ax_whisper] launch_advance: d_model %d unsupported\n", p.d_model); abort(); }
  hipLaunchKernelGGL(advance_kernel, dim3((p.batch + 15) / 16), dim3(1024), 0, s, p);
}

}  // inline namespace AXW_NS
}  // namespace axw
