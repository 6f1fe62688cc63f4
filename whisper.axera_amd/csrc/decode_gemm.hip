// decode_gemm.hip — the decoder step's nn.Linear layers for 5..64 clips per launch on the matrix cores.
//
// y[b][n] = sum_k W[n][k] a[b][k] (+ bias[n]): W bf16 [N][K] streamed from HBM exactly once per step for the
// whole batch (277.8 MB/step for small), activations fp32-equivalent: every activation is carried
// as a bf16 pair (hi, lo) with hi + lo == x to 16 mantissa bits, and both terms are multiplied on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation. The step stays HBM-bound (two MFMAs per 1 KiB of weights
// is far below the matrix rate), so the second term is free and keeps batched decoding numerically in
// line with the fp32-FMA GEMV used for 1..4 clips (decode_gemv.hip).
//
// Mapping (wave64): C[n][b] = W[n][:] . a[b][:], A operand = 16 weight rows, B operand = 16 clips.
//   workgroup = 4 waves = 16*RT weight rows; the 4 waves split K (each wave K/4), then reduce through LDS;
//   a wave keeps RT x NB accumulator tiles (NB = ceil(batch/16) <= 4) so one activation fragment feeds RT MFMAs;
//   weights: 16-byte loads, 4 lanes cover 64 contiguous bytes of a row per k-step, k-steps unrolled so the
//   whole 128-byte line is requested back to back; activations come from L2 (bf16 pairs written by the
//   producer: act_prep_kernel / attention / the GELU epilogue).
// Epilogues mirror decode_gemv.hip: bias, GELU (writes the bf16 pair), residual add, q + KV-cache append,
// vocabulary argmax partials (first max wins, Whisper.cpp:42-45).
#include "common.hpp"

namespace axw {

__device__ __forceinline__ void split_bf16(float x, bf16& hi, bf16& lo) {
  hi = (bf16)x;
  lo = (bf16)(x - (float)hi);
}

// LayerNorm (or plain copy) of the residual stream -> bf16 (hi, lo) rows; one workgroup per clip.
__global__ __launch_bounds__(256) void act_prep_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ be, bf16* __restrict__ hi, bf16* __restrict__ lo,
                                                       int K, int do_ln) {
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (long)b * K;
  constexpr int MAXE = 8;  // K <= 2048 for LayerNorm rows
  if (!do_ln) {
    for (int c = tid; c < K; c += 256) split_bf16(xr[c], hi[(long)b * K + c], lo[(long)b * K + c]);
    return;
  }
  float v[MAXE], gg[MAXE], bb[MAXE];
  const float shift = xr[0];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int c = tid + 256 * e;
    v[e] = c < K ? xr[c] : shift;
    gg[e] = c < K ? g[c] : 0.f;
    bb[e] = c < K ? be[c] : 0.f;
    const float t = v[e] - shift;
    s1 += t; s2 += t * t;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
  __syncthreads();
  const float t1 = (red[0] + red[2]) + (red[4] + red[6]), t2 = (red[1] + red[3]) + (red[5] + red[7]);
  const float dm = t1 / K, var = fmaxf(t2 / K - dm * dm, 0.f);
  const float mean = shift + dm, rstd = rsqrtf(var + 1e-5f);
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int c = tid + 256 * e;
    if (c < K) split_bf16((v[e] - mean) * rstd * gg[e] + bb[e], hi[(long)b * K + c], lo[(long)b * K + c]);
  }
}

void launch_act_prep(const float* x, const float* g, const float* be, bf16* hi, bf16* lo, int batch, int K, bool do_ln,
                     hipStream_t s) {
  hipLaunchKernelGGL(act_prep_kernel, dim3(batch), dim3(256), 0, s, x, g, be, hi, lo, K, do_ln ? 1 : 0);
}

template <int RT, int NB>
__global__ __launch_bounds__(256) void decode_gemm_kernel(DecGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [4 waves][RT][NB][16 n][16 b]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int K = p.K;
  const int n0 = blockIdx.x * (16 * RT);
  const int kw = K / 4;            // this wave's K range
  const int k_begin = wave * kw;
  const int nks = kw / 32;

  const bf16* wrow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) wrow[t] = p.W + (long)min(n0 + t * 16 + r, p.N - 1) * K + k_begin + 8 * q;
  const bf16* ahi[NB];
  const bf16* alo[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const int b = min(c * 16 + r, p.batch - 1);  // clips beyond the batch: duplicate loads, results discarded
    ahi[c] = p.a_hi + (long)b * K + k_begin + 8 * q;
    alo[c] = p.a_lo + (long)b * K + k_begin + 8 * q;
  }

  f32x4 acc[RT][NB];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][c][e] = 0.f;

  const int step = (p.epilogue == GEPI_LOGITS || p.epilogue == GEPI_QKV_CACHE) ? p.state->step : 0;
  if (p.epilogue == GEPI_LOGITS && step < p.skip_before_step) return;

#pragma unroll 2
  for (int ks = 0; ks < nks; ++ks) {
    bf16x8 wf[RT], hf[NB], lf[NB];
#pragma unroll
    for (int t = 0; t < RT; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(wrow[t] + ks * 32);
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      hf[c] = *reinterpret_cast<const bf16x8*>(ahi[c] + ks * 32);
      lf[c] = *reinterpret_cast<const bf16x8*>(alo[c] + ks * 32);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], hf[c], acc[t][c], 0, 0, 0);
        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], lf[c], acc[t][c], 0, 0, 0);
      }
  }

  // split-K reduction across the 4 waves: acc[t][c][e] = C[n = t*16 + 4q + e][b = c*16 + r]
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(((wave * RT + t) * NB + c) * 16 + 4 * q + e) * 16 + r] = acc[t][c][e];
  __syncthreads();

  constexpr int OUT = RT * NB * 256;  // outputs of this workgroup
  float best_v = -INFINITY;
  int best_i = 0x7fffffff;
  for (int o = tid; o < OUT; o += 256) {
    // o -> (t, c, nn, bb) with bb fastest: consecutive threads = consecutive clips of one weight row
    const int bb = o & 15, nn = (o >> 4) & 15, c = (o >> 8) % NB, t = (o >> 8) / NB;
    const int n = n0 + t * 16 + nn, b = c * 16 + bb;
    float y = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) y += red[(((w * RT + t) * NB + c) * 16 + nn) * 16 + bb];
    if (n >= p.N || b >= p.batch) continue;
    y += p.bias ? p.bias[n] : 0.f;
    switch (p.epilogue) {
      case GEPI_STORE: p.out[(long)b * p.N + n] = y; break;
      case GEPI_GELU: split_bf16(gelu_erf(y), p.out_hi[(long)b * p.N + n], p.out_lo[(long)b * p.N + n]); break;
      case GEPI_RESID: p.out[(long)b * p.N + n] += y; break;
      case GEPI_QKV_CACHE: {
        const int d = p.d_model;
        if (n < d) {
          p.out[(long)b * d + n] = y;
        } else {
          const int cc = (n < 2 * d) ? n - d : n - 2 * d;
          const int head = cc >> 6, dd = cc & 63;
          const long base = (long)b * p.kv_batch_stride + (long)head * p.n_ctx_pad * 64;
          if (n < 2 * d) p.k_cache[base + (long)(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (bf16)y;
          else p.v_cache[base + (long)step * 64 + dd] = (bf16)y;
        }
        break;
      }
      case GEPI_LOGITS:
        if (p.logits_dump) p.logits_dump[(long)b * p.logits_dump_stride + n] = y;
        red[4 * RT * NB * 256 + (t * 16 + nn) * (NB * 16) + b] = y;  // [16*RT rows][NB*16 clips] for the argmax below
        break;
    }
  }
  if (p.epilogue == GEPI_LOGITS) {  // per-clip argmax over this workgroup's rows, first max wins
    __syncthreads();
    if (tid < NB * 16 && tid < p.batch) {
      for (int i = 0; i < 16 * RT; ++i) {
        const int n = n0 + i;
        if (n >= p.N) break;
        const float y = red[4 * RT * NB * 256 + i * (NB * 16) + tid];
        if (y > best_v) { best_v = y; best_i = n; }
      }
      p.amax_val[(long)blockIdx.x * p.amax_stride + tid] = best_v;
      p.amax_idx[(long)blockIdx.x * p.amax_stride + tid] = best_i;
    }
  }
}

int decode_gemm_grid(int N, int rt) { return (N + 16 * rt - 1) / (16 * rt); }

template <int RT>
static void launch_nb(const DecGemmParams& p, hipStream_t s) {
  const int nb = (p.batch + 15) / 16;
  const int grid = decode_gemm_grid(p.N, RT);
  const size_t lds = (size_t)(4 * RT * nb * 256 + (p.epilogue == GEPI_LOGITS ? 16 * RT * nb * 16 : 0)) * 4;
  switch (nb) {
    case 1: hipLaunchKernelGGL((decode_gemm_kernel<RT, 1>), dim3(grid), dim3(256), lds, s, p); break;
    case 2: hipLaunchKernelGGL((decode_gemm_kernel<RT, 2>), dim3(grid), dim3(256), lds, s, p); break;
    case 3: hipLaunchKernelGGL((decode_gemm_kernel<RT, 3>), dim3(grid), dim3(256), lds, s, p); break;
    case 4: hipLaunchKernelGGL((decode_gemm_kernel<RT, 4>), dim3(grid), dim3(256), lds, s, p); break;
    default: fprintf(stderr, "[ax_whisper] launch_decode_gemm: batch %d > 64 per launch\n", p.batch); abort();
  }
}

// p.batch <= 64 per launch (the engine tiles larger batches); K % 128 == 0.
void launch_decode_gemm(const DecGemmParams& p, hipStream_t s) {
  if (p.K % 128 != 0) { fprintf(stderr, "[ax_whisper] launch_decode_gemm: unsupported K=%d\n", p.K); abort(); }
  if (p.rt == 4) launch_nb<4>(p, s);
  else launch_nb<1>(p, s);
}

}  // namespace axw
