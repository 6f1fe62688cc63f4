// decoder.hip — one autoregressive decoder step on gfx950 (K9-K14 of SURVEY §8a).
//
// Replaces Whisper::run_decoder (cpp/src/Whisper.cpp:290-346) = the decoder NPU blob
// (TextDecoderTensorCache.forward, model_convert/export_onnx.py:312-387) plus the host-side
// cache scatter (:328-342), argmax (:42-45) and loop bookkeeping (:207-222).
//
// All loop state (step counter, fed token, done flags, output ids) lives in device memory, so a
// step is a fixed sequence of launches that is captured ONCE into a hipGraph and replayed; the
// host never reads anything back inside the loop except a done counter every few steps.
//
// Kernels (decode is HBM-bound: weights are bf16 and read exactly once per step, K/V caches bf16):
//   embed_kernel            x = tok_emb[token] + pos[step]                       (export_onnx.py:334-336)
//   gemv_kernel<LPR,BT,CH>  y = W a (+bias) for B <= 4 rows at a time, fp32 activations x bf16
//                           weights with fp32 FMA; LPR lanes share one weight row (16-byte loads,
//                           all CH loads of a lane issued before the first use); fused prologues
//                           (LayerNorm of the residual stream / merge of the attention split
//                           partials) and epilogues (bias, GELU, residual add, q + KV-cache
//                           append, vocabulary argmax partials)
//   decode_attention_kernel single-query attention over a BLOCKED K layout
//                           [blk][d/8][64 keys][8] (lane = key: the q.k dot product is lane-local,
//                           every load is a contiguous 1 KiB wave access) and row-major V;
//                           used for self-attention over the incremental cache (keys 0..step,
//                           export_onnx.py:103-147: the -60000 mask + separate current-token column
//                           equal causal attention over 0..step) and for cross-attention over the
//                           1500 encoder keys (export_onnx.py:221-230)
//   advance_kernel          argmax merge (first max wins), SOT forcing, eot/ctx stop, token feed
#include "common.hpp"

namespace axw {

constexpr int kPartStride = 66;  // m, l, o[64]

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short u) { return __uint_as_float((unsigned)u << 16); }

// ------------------------------------------------------------------------------- embed
__global__ __launch_bounds__(256) void embed_kernel(const bf16* __restrict__ tok_emb, const float* __restrict__ pos,
                                                    const int* __restrict__ tok, const DecState* __restrict__ st, float* __restrict__ x,
                                                    int d) {
  const int b = blockIdx.x;
  const int t = tok[b], step = st->step;
  for (int c = threadIdx.x; c < d; c += 256) x[(long)b * d + c] = (float)tok_emb[(long)t * d + c] + pos[(long)step * d + c];
}

void launch_embed(const bf16* tok_emb, const float* pos, const int* tok, const DecState* st, float* x, int batch, int d,
                  hipStream_t s) {
  hipLaunchKernelGGL(embed_kernel, dim3(batch), dim3(256), 0, s, tok_emb, pos, tok, st, x, d);
}

// ------------------------------------------------------------------------------- GEMV
template <int LPR, int BT, int CH>
__global__ __launch_bounds__(256) void gemv_kernel(GemvParams p, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* act = reinterpret_cast<float*>(smem);  // [BT][K]
  __shared__ float s_val[4 * BT];
  __shared__ int s_idx[4 * BT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K;
  const int step = p.state ? p.state->step : 0;
  if (p.epilogue == GEPI_LOGITS && step < p.skip_before_step) return;  // SOT steps: logits are discarded (Whisper.cpp:214-217)

  // ---- prologue: build the activation rows in LDS
  if (p.prologue == PRO_PLAIN) {
    for (int i = tid; i < BT * K; i += 256) {
      int b = i / K;
      act[i] = b < p.batch ? p.in[i] : 0.f;
    }
  } else if (p.prologue == PRO_LAYERNORM) {
    for (int b = wave; b < BT; b += 4) {
      if (b >= p.batch) {
        for (int c = lane; c < K; c += 64) act[b * K + c] = 0.f;
        continue;
      }
      const float* xr = p.in + (long)b * K;
      float s = 0.f;
      for (int c = lane; c < K; c += 64) s += xr[c];
      const float mean = wave_sum(s) / K;
      float q = 0.f;
      for (int c = lane; c < K; c += 64) { float t = xr[c] - mean; q += t * t; }
      const float rstd = rsqrtf(wave_sum(q) / K + 1e-5f);
      for (int c = lane; c < K; c += 64) act[b * K + c] = (xr[c] - mean) * rstd * p.ln_w[c] + p.ln_b[c];
    }
  } else {  // PRO_ATTN_COMBINE: merge the split partials of decode_attention_kernel
    for (int i = tid; i < BT * K; i += 256) {
      int b = i / K, c = i - b * K;
      float v = 0.f;
      if (b < p.batch) {
        const float* pp = p.part + ((long)b * p.n_head + (c >> 6)) * p.n_split * kPartStride;
        float m = -INFINITY;
        for (int s = 0; s < p.n_split; ++s) m = fmaxf(m, pp[s * kPartStride]);
        float l = 0.f, o = 0.f;
        for (int s = 0; s < p.n_split; ++s) {
          float w = __expf(pp[s * kPartStride] - m);
          l += w * pp[s * kPartStride + 1];
          o += w * pp[s * kPartStride + 2 + (c & 63)];
        }
        v = o / l;
      }
      act[i] = v;
    }
  }
  __syncthreads();

  constexpr int RP = 256 / LPR;  // rows per pass
  const int j = tid % LPR, rsub = tid / LPR;
  const int row_begin = blockIdx.x * rows_per_wg;
  const int row_end = min(p.N, row_begin + rows_per_wg);
  const int nch = CH > 0 ? CH : K / (8 * LPR);

  float best_v[BT];
  int best_i[BT];
#pragma unroll
  for (int b = 0; b < BT; ++b) { best_v[b] = -INFINITY; best_i[b] = 0x7fffffff; }

  for (int row0 = row_begin; row0 < row_end; row0 += RP) {
    const int n = row0 + rsub;
    const bool valid = n < row_end;
    const bf16* wrow = p.W + (long)(valid ? n : row_begin) * K;
    float acc[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[b] = 0.f;
    if constexpr (CH > 0) {
      uint4 w[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) w[i] = *reinterpret_cast<const uint4*>(wrow + (j + LPR * i) * 8);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c8 = (j + LPR * i) * 8;
        const unsigned uw[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
        for (int b = 0; b < BT; ++b) {
          const float4 a0 = *reinterpret_cast<const float4*>(act + b * K + c8);
          const float4 a1 = *reinterpret_cast<const float4*>(act + b * K + c8 + 4);
          acc[b] = fmaf(__uint_as_float(uw[0] << 16), a0.x, acc[b]);
          acc[b] = fmaf(__uint_as_float(uw[0] & 0xffff0000u), a0.y, acc[b]);
          acc[b] = fmaf(__uint_as_float(uw[1] << 16), a0.z, acc[b]);
          acc[b] = fmaf(__uint_as_float(uw[1] & 0xffff0000u), a0.w, acc[b]);
          acc[b] = fmaf(__uint_as_float(uw[2] << 16), a1.x, acc[b]);
          acc[b] = fmaf(__uint_as_float(uw[2] & 0xffff0000u), a1.y, acc[b]);
          acc[b] = fmaf(__uint_as_float(uw[3] << 16), a1.z, acc[b]);
          acc[b] = fmaf(__uint_as_float(uw[3] & 0xffff0000u), a1.w, acc[b]);
        }
      }
    } else {
      for (int i = 0; i < nch; ++i) {
        const int c8 = (j + LPR * i) * 8;
        const uint4 w = *reinterpret_cast<const uint4*>(wrow + c8);
        const unsigned uw[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int b = 0; b < BT; ++b) {
          const float* a = act + b * K + c8;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[b] = fmaf(__uint_as_float(uw[e] << 16), a[2 * e], acc[b]);
            acc[b] = fmaf(__uint_as_float(uw[e] & 0xffff0000u), a[2 * e + 1], acc[b]);
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < BT; ++b)
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) acc[b] += __shfl_xor(acc[b], o, 64);

    if (j == 0 && valid) {
      const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        if (b >= p.batch) break;
        const float y = acc[b] + bias;
        switch (p.epilogue) {
          case GEPI_STORE: p.out[(long)b * p.N + n] = y; break;
          case GEPI_GELU: p.out[(long)b * p.N + n] = gelu_erf(y); break;
          case GEPI_RESID: p.out[(long)b * p.N + n] += y; break;
          case GEPI_QKV_CACHE: {
            const int d = p.d_model;
            if (n < d) {
              p.out[(long)b * d + n] = y;
            } else {
              const int c = (n < 2 * d) ? n - d : n - 2 * d;
              const int head = c >> 6, dd = c & 63;
              const long base = (long)b * p.kv_batch_stride + (long)head * p.n_ctx_pad * 64;
              if (n < 2 * d)  // blocked K: [blk][dd/8][key%64][8]
                p.k_cache[base + (long)(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (bf16)y;
              else            // row-major V: [key][64]
                p.v_cache[base + (long)step * 64 + dd] = (bf16)y;
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
        float ov = __shfl_xor(v, o, 64);
        int oi = __shfl_xor(ix, o, 64);
        if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
      }
      if (lane == 0) { s_val[wave * BT + b] = v; s_idx[wave * BT + b] = ix; }
    }
    __syncthreads();
    if (tid < BT && tid < p.batch) {
      float v = s_val[tid];
      int ix = s_idx[tid];
      for (int w = 1; w < 4; ++w) {
        float ov = s_val[w * BT + tid];
        int oi = s_idx[w * BT + tid];
        if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
      }
      p.amax_val[(long)blockIdx.x * p.amax_stride + tid] = v;
      p.amax_idx[(long)blockIdx.x * p.amax_stride + tid] = ix;
    }
  }
}

static int pick_lpr(int K) {
  // widest lane group whose per-lane chunk count stays small (more rows in flight per wave otherwise)
  if (K % 512 == 0 && K / 512 <= 10 && K >= 1024) return 64;
  if (K % 256 == 0 && K / 256 <= 6) return 32;
  if (K % 128 == 0) return 16;
  return 0;
}

static int rows_per_wg_for(int N, int lpr) {
  const int rp = 256 / lpr;
  // aim for >= ~2 workgroups per CU on small N, cap the grid on large N
  int target_wgs = 512;
  int rows = (N + target_wgs - 1) / target_wgs;
  rows = ((rows + rp - 1) / rp) * rp;
  return rows < rp ? rp : rows;
}

int gemv_grid(const GemvParams& p) {
  int lpr = pick_lpr(p.K);
  int rpw = rows_per_wg_for(p.N, lpr);
  return (p.N + rpw - 1) / rpw;
}

template <int LPR, int BT>
static void launch_gemv_ch(const GemvParams& p, int rpw, int grid, hipStream_t s) {
  const int ch = p.K / (8 * LPR);
  const size_t lds = (size_t)BT * p.K * 4;
#define AXW_GEMV_CASE(C) \
  case C: hipLaunchKernelGGL((gemv_kernel<LPR, BT, C>), dim3(grid), dim3(256), lds, s, p, rpw); break;
  switch (ch) {
    AXW_GEMV_CASE(1) AXW_GEMV_CASE(2) AXW_GEMV_CASE(3) AXW_GEMV_CASE(4) AXW_GEMV_CASE(5) AXW_GEMV_CASE(6) AXW_GEMV_CASE(10)
    default: hipLaunchKernelGGL((gemv_kernel<LPR, BT, 0>), dim3(grid), dim3(256), lds, s, p, rpw); break;
  }
#undef AXW_GEMV_CASE
}

// Handles p.batch <= 4 per launch; the engine tiles larger batches.
void launch_gemv(const GemvParams& p, hipStream_t s) {
  const int lpr = pick_lpr(p.K);
  const int rpw = rows_per_wg_for(p.N, lpr);
  const int grid = (p.N + rpw - 1) / rpw;
  const bool one = p.batch == 1;
  switch (lpr) {
    case 64: one ? launch_gemv_ch<64, 1>(p, rpw, grid, s) : launch_gemv_ch<64, 4>(p, rpw, grid, s); break;
    case 32: one ? launch_gemv_ch<32, 1>(p, rpw, grid, s) : launch_gemv_ch<32, 4>(p, rpw, grid, s); break;
    default: one ? launch_gemv_ch<16, 1>(p, rpw, grid, s) : launch_gemv_ch<16, 4>(p, rpw, grid, s); break;
  }
}

// ------------------------------------------------------------------------------- decode attention
__global__ __launch_bounds__(256) void decode_attention_kernel(DecAttnParams p, int cap_blocks) {
  __shared__ float s_part[4][kPartStride];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int n_keys = p.n_keys >= 0 ? p.n_keys : p.state->step + 1;
  const int n_blocks = (n_keys + 63) >> 6;
  const int bps = (cap_blocks + p.n_split - 1) / p.n_split;
  const int blk_begin = split * bps, blk_end = min(n_blocks, blk_begin + bps);

  const float* __restrict__ qp = p.q + (long)b * p.d_model + head * 64;  // wave-uniform: scalar loads
  const bf16* kb = p.k + (long)b * p.kv_batch_stride + (long)head * cap_blocks * 4096;
  const bf16* vb = p.v + (long)b * p.kv_batch_stride + (long)head * cap_blocks * 4096;

  float m_w = -INFINITY, l_lane = 0.f;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;

  for (int blk = blk_begin + wave; blk < blk_end; blk += 4) {
    uint4 kr[8], vr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kr[i] = *reinterpret_cast<const uint4*>(kb + (long)blk * 4096 + i * 512 + lane * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) vr[i] = *reinterpret_cast<const uint4*>(vb + ((long)blk * 64 + 8 * i + (lane >> 3)) * 64 + (lane & 7) * 8);
    // lane = key blk*64 + lane: dot(q, k) over 64 dims, q from scalar registers
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned u[4] = {kr[i].x, kr[i].y, kr[i].z, kr[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc = fmaf(qp[i * 8 + 2 * e], __uint_as_float(u[e] << 16), sc);
        sc = fmaf(qp[i * 8 + 2 * e + 1], __uint_as_float(u[e] & 0xffff0000u), sc);
      }
    }
    sc *= 0.125f;  // (64^-0.25)^2, export_onnx.py:116,124-126
    if (blk * 64 + lane >= n_keys) sc = -INFINITY;
    const float m_new = fmaxf(m_w, wave_max(sc));
    const float alpha = __expf(m_w - m_new);
    const float pk = __expf(sc - m_new);
    m_w = m_new;
    l_lane = l_lane * alpha + pk;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float w = __shfl(pk, 8 * i + (lane >> 3), 64);
      const unsigned u[4] = {vr[i].x, vr[i].y, vr[i].z, vr[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = fmaf(w, __uint_as_float(u[e] << 16), o[2 * e]);
        o[2 * e + 1] = fmaf(w, __uint_as_float(u[e] & 0xffff0000u), o[2 * e + 1]);
      }
    }
  }
  // wave partial: sum o over the 8 key sub-rows (lanes with equal lane&7), sum l over the wave
  const float l_w = wave_sum(l_lane);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    o[e] += __shfl_xor(o[e], 8, 64);
    o[e] += __shfl_xor(o[e], 16, 64);
    o[e] += __shfl_xor(o[e], 32, 64);
  }
  if (lane == 0) { s_part[wave][0] = m_w; s_part[wave][1] = l_w; }
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s_part[wave][2 + lane * 8 + e] = o[e];
  }
  __syncthreads();
  if (tid < 64) {
    float m = fmaxf(fmaxf(s_part[0][0], s_part[1][0]), fmaxf(s_part[2][0], s_part[3][0]));
    float l = 0.f, ov = 0.f;
    if (m > -INFINITY) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float f = __expf(s_part[w][0] - m);
        l += f * s_part[w][1];
        ov += f * s_part[w][2 + tid];
      }
    }
    float* out = p.part + (((long)b * p.n_head + head) * p.n_split + split) * kPartStride;
    if (tid == 0) { out[0] = m; out[1] = l; }
    out[2 + tid] = ov;
  }
}

void launch_decode_attention(const DecAttnParams& p, hipStream_t s) {
  hipLaunchKernelGGL(decode_attention_kernel, dim3(p.n_split, p.n_head, p.batch), dim3(256), 0, s, p, p.cap_blocks);
}

// ------------------------------------------------------------------------------- advance
// Whisper.cpp:207-222: steps 0..2 feed the next SOT token and drop the logits; from step 3 on the
// argmax is either the stop condition (eot / context full) or the next recorded + fed token.
__global__ __launch_bounds__(256) void advance_kernel(AdvanceParams p) {
  const int s = p.state->step;
  for (int b = threadIdx.x; b < p.batch; b += 256) {
    if (s < 3) {
      p.tok[b] = p.sot[s + 1];
      continue;
    }
    float v = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = 0; i < p.n_part; ++i) {
      float ov = p.amax_val[(long)i * p.amax_stride + b];
      int oi = p.amax_idx[(long)i * p.amax_stride + b];
      if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int gi = s - 3;
    if (p.argmax_dump && gi <= p.n_forced) p.argmax_dump[(long)b * (p.n_forced + 1) + gi] = idx;
    if (p.forced) {
      if (gi < p.n_forced) p.tok[b] = p.forced[(long)b * p.n_forced + gi];
    } else if (!p.done[b]) {
      if (idx == p.eot || s + 1 >= p.n_ctx || p.n_out[b] >= p.max_new) {
        p.done[b] = 1;
        atomicAdd(&p.state->n_done, 1);
      } else {
        p.out_ids[(long)b * p.n_ctx + p.n_out[b]] = idx;
        p.n_out[b] += 1;
        p.tok[b] = idx;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) p.state->step = s + 1;
}

void launch_advance(const AdvanceParams& p, hipStream_t s) { hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(256), 0, s, p); }

// ------------------------------------------------------------------------------- weight preparation
__device__ __forceinline__ float load_as_f32(const void* src, int dt, long i) {
  if (dt == 0) return reinterpret_cast<const float*>(src)[i];
  if (dt == 1) return bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(src)[i]);
  return (float)reinterpret_cast<const _Float16*>(src)[i];
}
__global__ void convert_to_bf16_kernel(const void* src, int dt, bf16* dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = (bf16)load_as_f32(src, dt, i);
}
__global__ void convert_to_f32_kernel(const void* src, int dt, float* dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = load_as_f32(src, dt, i);
}
// Conv1d weight [Cout][Cin][3] -> GEMM weight [Cout][kpad], column k*Cin + c (zero beyond 3*Cin)
__global__ void conv_weight_pack_kernel(const void* src, int dt, bf16* dst, int cout, int cin, int kpad) {
  const long total = (long)cout * kpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int n = (int)(i / kpad), col = (int)(i - (long)n * kpad);
    float v = 0.f;
    if (col < 3 * cin) {
      int k = col / cin, c = col - k * cin;
      v = load_as_f32(src, dt, ((long)n * cin + c) * 3 + k);
    }
    dst[i] = (bf16)v;
  }
}
static int grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }
void launch_convert_to_bf16(const void* src, int dt, bf16* dst, long n, hipStream_t s) {
  hipLaunchKernelGGL(convert_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dt, dst, n);
}
void launch_convert_to_f32(const void* src, int dt, float* dst, long n, hipStream_t s) {
  hipLaunchKernelGGL(convert_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dt, dst, n);
}
void launch_conv_weight_pack(const void* src, int dt, bf16* dst, int cout, int cin, int kpad, hipStream_t s) {
  hipLaunchKernelGGL(conv_weight_pack_kernel, dim3(grid_for((long)cout * kpad)), dim3(256), 0, s, src, dt, dst, cout, cin, kpad);
}

}  // namespace axw
