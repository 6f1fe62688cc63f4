// decode_tail.hip — the second half of a decoder layer for 16..64 clips as ONE launch (gfx950).
//
// Replaces four dependent launches of the clip-block sequence (decode_gemm.hip, decode_cgemm_kernel):
//   cross-attention output projection (+ residual)  ->  LayerNorm + mlp.0 + GELU  ->  mlp.2 (+ residual)
//   ->  LayerNorm + Q,K,V projection of the NEXT layer (+ self-K/V cache append)
// i.e. export_onnx.py:292-299 of layer l and :285-289 / :245-247 of layer l + 1 (Whisper.cpp:290-346 runs them inside the
// decoder blob). The batched decoder step was the sum of its bandwidth-bound attention launches and a latency-bound
// chain of 60 small GEMM launches (5 per layer); three of the five seams of a layer are now INSIDE a launch.
//
// Why this can be cheaper than launches (and what the round-2 attempt with grid-wide barriers lacked):
//   * clips never mix: a clip block (16 clips) is served by its own CLUSTER of workgroups (all resident, one per CU),
//     and a phase waits only for the producers of its own clip block — a counter per (clip block, phase), no grid barrier;
//   * a workgroup requests the weight fragments of its NEXT phase before it waits for the hand-off (they do not depend
//     on it): what is left behind the wait is the activation read of one clip block (49 KB from L2) and the arithmetic;
//   * every work item of a phase has its own resident workgroup (no rounds);
//   * the residual value of an output element stays in the register of the thread that owns it from the cross-attention
//     projection to mlp.2 (same workgroup, same thread).
// Hand-off form (MI355X_MICROARCH.md, hand-offs without an acquire, first row): payload stored write-through (sc1), the
// storing waves drain vmcnt, a workgroup barrier, ONE lane adds to the phase's counter (agent scope); the consumer's
// lane 0 polls the counter with sc1 loads, a workgroup barrier, then every wave reads the payload with sc1 loads.
// Arithmetic, operand layouts and summation order are decode_cgemm_kernel's (k-steps round-robin over 8 waves, hi then lo
// MFMA per k-step, waves folded 0..7 onto the bias): results are bit-identical to the launch-per-layer sequence at
// d = 512 (tests/test_gpu_batched.py; at d = 128 the compiler contracts the two kernels' LayerNorm arithmetic differently: 2e-5).
// Every wait is bounded in time; a workgroup that gives up sets an error word that the host checks (engine.cpp).
#include "common.hpp"
#include <algorithm>

namespace axw {
inline namespace AXW_NS {

namespace {

typedef __attribute__((address_space(1))) unsigned tail_gu32;

constexpr int kTailRT = 3;                 // weight-row tiles per workgroup in the two LayerNorm phases
constexpr long long kTailSpinTicks = 5000000;  // 50 ms at 100 MHz

__device__ __forceinline__ void tail_split(float x, h16& hi, h16& lo) {
  hi = (h16)x;
  lo = (h16)(x - (float)hi);
}
__device__ __forceinline__ float tail_sum_16_32(float v) {  // v[l] + v[l^16] + v[l^32] + v[l^48], as decode_gemm.hip
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// write-through / L1-bypassing 16-byte accesses (aux 16 = sc1)
__device__ __forceinline__ f32x4 ld_sc1_f32x4(__amdgpu_buffer_rsrc_t rs, long byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ h16x8 ld_sc1_h16x8(__amdgpu_buffer_rsrc_t rs, long byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
  return __builtin_bit_cast(h16x8, v);
}
__device__ __forceinline__ void st_sc1_b128(__amdgpu_buffer_rsrc_t rs, long byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_off, 0, 16);
}
__device__ __forceinline__ void st_sc1_f32(float* p, float v) {
  __hip_atomic_store((__attribute__((address_space(1))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

// sync words (unsigned), per launch group (graph branch): [2] error; per clip block cb at [8 + 8*cb]: the three phase counters,
// the cluster's epoch and its exit ticket — per cluster, so that launches with different clip counts stay consistent
template <int CH>
__global__ __launch_bounds__(512) void decode_tail_kernel(DecTailParams p) {
  __shared__ __attribute__((aligned(16))) float red[8 * kTailRT * 256];  // [wave][t][clip][row]
  __shared__ float stat[2][8][16];
  __shared__ __attribute__((aligned(16))) h16 stg[2][16][16 * kTailRT];  // GELU outputs (hi, lo) [clip][row] for 16-byte stores
  __shared__ int s_fail;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int d = p.d, KS = d / 32, KS4 = d / 8;
  const int W = p.wgs_per_cluster;
  const int cb = blockIdx.x / W, j = blockIdx.x % W;  // clip block of this cluster, workgroup index in the cluster
  const int nA = d / 16, nBb = d / 4, nDb = 3 * d / 16;  // weight-row blocks of the d-, 4d- and 3d-row layers
  const int nB = (nBb + kTailRT - 1) / kTailRT, nD = (nDb + kTailRT - 1) / kTailRT;  // producers of the two LayerNorm phases
  unsigned* const sync = p.sync;
  unsigned* const cnt = sync + 8 + 8 * cb;
  const unsigned seq1 = cnt[3] + 1u;  // this launch's epoch of the cluster (the word is bumped by its last workgroup to leave)
  if (tid == 0) s_fail = 0;

  const __amdgpu_buffer_rsrc_t RX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7fffffff, 0x27000);
  const __amdgpu_buffer_rsrc_t RH = __builtin_amdgcn_make_buffer_rsrc((void*)p.hid_hi, 0, 0x7fffffff, 0x27000);
  const __amdgpu_buffer_rsrc_t RL = __builtin_amdgcn_make_buffer_rsrc((void*)p.hid_lo, 0, 0x7fffffff, 0x27000);

  // lane 0 of the workgroup waits until `n_prod` producers of this launch have added to `c`; everybody learns the outcome
  auto wait_for = [&](unsigned* c, int n_prod) -> bool {
    if (tid == 0) {
      const unsigned target = (unsigned)n_prod * seq1;
      long long t0 = 0;
      for (int spins = 0;; ++spins) {
        const unsigned v = __hip_atomic_load((tail_gu32*)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255) == 255) {
          if (__hip_atomic_load((tail_gu32*)(sync + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { s_fail = 1; break; }
          const long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > kTailSpinTicks) {
            __hip_atomic_store((tail_gu32*)(sync + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_fail = 1;
            break;
          }
        }
      }
    }
    __syncthreads();
    return s_fail == 0;
  };
  // after the payload stores of this workgroup: drain, barrier, one add
  auto signal = [&](unsigned* c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add((tail_gu32*)c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // this thread's output element in the two d-row phases (cross-attention projection, mlp.2): row block j, RT = 1
  const bool in_d = j < nA;
  const int o_n = j * 16 + (tid & 15), o_b = cb * 16 + (tid >> 4);
  const bool own_d = in_d && tid < 256 && o_b < p.batch;
  float xa = 0.f;  // the residual stream's value at (o_b, o_n): lives here from phase A to phase C

  bool ok = true;

  // ============================================================ A: x += W_co . att + b_co   (export_onnx.py:292-295)
  if (in_d) {
    const float bias_t = own_d ? p.b_co[o_n] : 0.f;
    const float old_t = own_d ? p.x[(long)o_b * d + o_n] : 0.f;
    const h16* wrow = p.w_co + (long)j * KS * 512 + lane * 8;
    const h16* ahi = p.att_hi + (long)cb * 512 + lane * 8;
    const h16* alo = p.att_lo + (long)cb * 512 + lane * 8;
    const long a_step = (long)p.nbs * 512;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    h16x8 w[CH], ah[CH], al[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ks = min(wave + 8 * c, KS - 1);
      w[c] = *reinterpret_cast<const h16x8*>(wrow + (long)ks * 512);
      ah[c] = *reinterpret_cast<const h16x8*>(ahi + ks * a_step);
      al[c] = *reinterpret_cast<const h16x8*>(alo + ks * a_step);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (wave + 8 * c < KS) {
        acc = AXW_MFMA_16x16x32(w[c], ah[c], acc);
        acc = AXW_MFMA_16x16x32(w[c], al[c], acc);
      }
    *reinterpret_cast<f32x4*>(red + (wave * 16 + r) * 16 + 4 * q) = acc;
    __syncthreads();
    if (own_d) {
      float y = bias_t;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2) y += red[(w2 * 16 + (tid >> 4)) * 16 + (tid & 15)];
      xa = old_t + y;
      st_sc1_f32(p.x + (long)o_b * d + o_n, xa);
    }
    signal(cnt + 0);
  }

  // ============================================================ B: hid = gelu(W_fc1 . LN(x) + b_fc1)   (export_onnx.py:298)
  // D below has the same shape (LayerNorm prologue, kTailRT row tiles): one body, two epilogues
  auto ln_phase = [&](const h16* Wp, const float* bias, const float* lnw, const float* lnb, int n_blocks, int N, unsigned* wait_c,
                      int wait_n, bool is_qkv) {
    const int nb0 = j * kTailRT;
    const bool active = nb0 < n_blocks;
    h16x8 w[CH][kTailRT];
    f32x4 gg[CH][2], bb[CH][2];
    if (active) {  // everything that does not depend on the hand-off: requested before the wait
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int ks = min(wave + 8 * c, KS - 1);
#pragma unroll
        for (int t = 0; t < kTailRT; ++t)
          w[c][t] = *reinterpret_cast<const h16x8*>(Wp + ((long)min(nb0 + t, n_blocks - 1) * KS + ks) * 512 + lane * 8);
        const int k0 = ks * 32 + q * 8;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          gg[c][u] = *reinterpret_cast<const f32x4*>(lnw + k0 + 4 * u);
          bb[c][u] = *reinterpret_cast<const f32x4*>(lnb + k0 + 4 * u);
        }
      }
    }
    if (!active) return true;  // (workgroup-uniform) no rows of this layer here: nothing to wait for
    if (wait_c && !wait_for(wait_c, wait_n)) return false;
    // x of this clip block: lane (r, q) holds x[clip r][k = ks*32 + 8q .. +8] of every k-step of its wave
    const long xrow = (long)min(cb * 16 + r, p.batch - 1) * d;
    f32x4 v[CH][2];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k0 = min(wave + 8 * c, KS - 1) * 32 + q * 8;
#pragma unroll
      for (int u = 0; u < 2; ++u) v[c][u] = ld_sc1_f32x4(RX, (xrow + k0 + 4 * u) * 4);
    }
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (wave + 8 * c < KS) {
#pragma unroll
        for (int u = 0; u < 2; ++u) s1 += (v[c][u][0] + v[c][u][1]) + (v[c][u][2] + v[c][u][3]);
      }
    s1 = tail_sum_16_32(s1);
    if (q == 0) stat[0][wave][r] = s1;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 8; ++w2) mean += stat[0][w2][r];
    mean /= (float)d;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (wave + 8 * c < KS) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float t2 = v[c][u][e] - mean; s2 += t2 * t2; }
      }
    s2 = tail_sum_16_32(s2);
    if (q == 0) stat[1][wave][r] = s2;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 8; ++w2) var += stat[1][w2][r];
    const float rstd = rsqrtf(var / (float)d + 1e-5f);
    f32x4 acc[kTailRT];
#pragma unroll
    for (int t = 0; t < kTailRT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (wave + 8 * c < KS) {
        h16x8 hi, lo;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = (v[c][u][e] - mean) * rstd * gg[c][u][e] + bb[c][u][e];
            h16 hh, ll;
            tail_split(y, hh, ll);
            hi[4 * u + e] = hh; lo[4 * u + e] = ll;
          }
#pragma unroll
        for (int t = 0; t < kTailRT; ++t) {
          acc[t] = AXW_MFMA_16x16x32(w[c][t], hi, acc[t]);
          acc[t] = AXW_MFMA_16x16x32(w[c][t], lo, acc[t]);
        }
      }
#pragma unroll
    for (int t = 0; t < kTailRT; ++t) *reinterpret_cast<f32x4*>(red + ((wave * kTailRT + t) * 16 + r) * 16 + 4 * q) = acc[t];
    __syncthreads();
    const int n0 = nb0 * 16;
    for (int o = tid; o < kTailRT * 256; o += 512) {
      const int nl = o % (16 * kTailRT), bl = o / (16 * kTailRT);
      const int t = nl >> 4, nn = nl & 15;
      const int n = n0 + nl, b = cb * 16 + bl;
      float y = (n < N && bias) ? bias[n] : 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2) y += red[((w2 * kTailRT + t) * 16 + bl) * 16 + nn];
      if (!is_qkv) {  // GELU -> (hi, lo) pair, staged for 16-byte stores
        h16 hh, ll;
        tail_split(gelu_erf(y), hh, ll);
        stg[0][bl][nl] = hh;
        stg[1][bl][nl] = ll;
      } else if (n < N && b < p.batch) {  // q, or this clip's self-K/V cache row (Whisper.cpp:328-342)
        if (n < d) {
          p.q_out[(long)b * d + n] = y;
        } else {
          const int step = p.off[b];
          const int cc = (n < 2 * d) ? n - d : n - 2 * d;
          const int head = cc >> 6, dd = cc & 63;
          const long base = (long)b * p.kv_batch_stride + (long)head * p.n_ctx_pad * 64;
          if (n < 2 * d) p.k_cache[base + (long)(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (h16)y;
          else p.v_cache[base + (long)step * 64 + dd] = (h16)y;
        }
      }
    }
    if (!is_qkv) {
      __syncthreads();
      // 8 consecutive rows of one clip are 16 contiguous bytes of the fragment-major pair layout (decode_gemm.hip)
      constexpr int GROUPS = 2 * kTailRT;  // 8-row groups per clip
      if (tid < 2 * 16 * GROUPS) {
        const int which = tid / (16 * GROUPS), rem = tid % (16 * GROUPS);
        const int bl = rem / GROUPS, g = rem % GROUPS;
        const int n = n0 + 8 * g;
        if (n < N) {
          const long idx = ((((long)(n >> 5) * p.nbs + cb) * 64) + ((n >> 3) & 3) * 16 + bl) * 8;
          const u32x4 val = *reinterpret_cast<const u32x4*>(&stg[which][bl][8 * g]);
          st_sc1_b128(which ? RL : RH, idx * 2, val);
        }
      }
    }
    return true;
  };

  ok = ln_phase(p.w_fc1, p.b_fc1, p.ln2_w, p.ln2_b, nBb, 4 * d, cnt + 0, nA, false);
  if (ok && j < nB) signal(cnt + 1);

  // ============================================================ C: x += W_fc2 . hid + b_fc2
  if (ok && in_d) {
    const float bias_t = own_d ? p.b_fc2[o_n] : 0.f;
    const h16* wrow = p.w_fc2 + (long)j * KS4 * 512 + lane * 8;
    const long a_step = (long)p.nbs * 512;
    constexpr int DEPTH = 4 * CH;  // k-steps per wave: K = 4d
    h16x8 w[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) w[i] = *reinterpret_cast<const h16x8*>(wrow + (long)min(wave + 8 * i, KS4 - 1) * 512);
    ok = wait_for(cnt + 1, nB);
    if (ok) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      constexpr int HALF = (DEPTH + 1) / 2;
#pragma unroll
      for (int h0 = 0; h0 < DEPTH; h0 += HALF) {
        h16x8 ah[HALF], al[HALF];
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
          const long idx = ((long)min(wave + 8 * (h0 + i), KS4 - 1) * a_step + (long)cb * 512 + lane * 8) * 2;
          ah[i] = ld_sc1_h16x8(RH, idx);
          al[i] = ld_sc1_h16x8(RL, idx);
        }
#pragma unroll
        for (int i = 0; i < HALF; ++i)
          if (h0 + i < DEPTH && wave + 8 * (h0 + i) < KS4) {
            acc = AXW_MFMA_16x16x32(w[h0 + i], ah[i], acc);
            acc = AXW_MFMA_16x16x32(w[h0 + i], al[i], acc);
          }
      }
      *reinterpret_cast<f32x4*>(red + (wave * 16 + r) * 16 + 4 * q) = acc;
      __syncthreads();
      if (own_d) {
        float y = bias_t;
#pragma unroll
        for (int w2 = 0; w2 < 8; ++w2) y += red[(w2 * 16 + (tid >> 4)) * 16 + (tid & 15)];
        xa = xa + y;
        st_sc1_f32(p.x + (long)o_b * d + o_n, xa);
      }
      signal(cnt + 2);  // (also in the last layer, which has no phase D: every counter advances once per launch)
    }
  }

  // ============================================================ D: q, k, v of the NEXT layer = W_qkv . LN(x) + b   (export_onnx.py:245-247)
  if (ok && p.w_qkv) ok = ln_phase(p.w_qkv, p.b_qkv, p.ln1_w, p.ln1_b, nDb, 3 * d, cnt + 2, nA, true);

  // the last workgroup of the cluster to leave opens the cluster's next epoch
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add((tail_gu32*)(cnt + 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == (unsigned)W - 1u) {
      __hip_atomic_store((tail_gu32*)(cnt + 4), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((tail_gu32*)(cnt + 3), seq1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int decode_tail_cluster(int d) { return std::max(d / 16, (d / 4 + kTailRT - 1) / kTailRT); }
bool decode_tail_supported(int d) { return d % 128 == 0 && d >= 128 && d <= 1024; }

void launch_decode_tail(const DecTailParams& p, hipStream_t s) {
  if (!decode_tail_supported(p.d) || p.wgs_per_cluster != decode_tail_cluster(p.d)) {
    fprintf(stderr, "[ax_whisper] launch_decode_tail: d_model %d unsupported\n", p.d);
    abort();
  }
  const dim3 grid(p.wgs_per_cluster * ((p.batch + 15) / 16));
  switch ((p.d / 32 + 7) / 8) {
    case 1: hipLaunchKernelGGL((decode_tail_kernel<1>), grid, dim3(512), 0, s, p); break;
    case 2: hipLaunchKernelGGL((decode_tail_kernel<2>), grid, dim3(512), 0, s, p); break;
    case 3: hipLaunchKernelGGL((decode_tail_kernel<3>), grid, dim3(512), 0, s, p); break;
    default: hipLaunchKernelGGL((decode_tail_kernel<4>), grid, dim3(512), 0, s, p); break;
  }
}

}  // inline namespace AXW_NS
}  // namespace axw
