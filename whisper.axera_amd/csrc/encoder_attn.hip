// encoder_attn.hip — fused non-causal self-attention of the audio encoder (K7 of SURVEY §8a).
//
// Replaces upstream qkv_attention with SDPA disabled (export_onnx.py:714-717) inside the
// encoder blob: softmax_fp32((q*s)(k*s)^T) v with s = 64^-0.25, T = 1500 keys, no mask (the
// zero-padded mel frames ARE attended to, as in the reference; only the 36 rows that pad
// 1500 -> 1536 are masked).
//
// Structure (wave64, flash-style, scores never leave registers):
//   workgroup = 4 waves = 128 query rows of one (clip, head); wave = 32 query rows.
//   S^T = K Q^T  (A = K tile rows from LDS, B = Q rows held in registers for the whole kernel):
//     the 32x32 accumulator then has the QUERY on the lane and the keys in registers, so the
//     row max / row sum are per-lane reductions plus one lane^32 exchange;
//   the exponentiated accumulator registers, narrowed pairwise to h16, ARE the B operand of
//   O^T = V^T P^T with no lane movement (cdna guide §3, accumulator-as-operand); the A operand
//   wants the keys of a 16-group in the matching order {0-3, 8-11 | 4-7, 12-15}, which is the order the QKV GEMM's
//   swapped-operand epilogue stores the frames of V^T in (gemm.hip): one 16-byte read per fragment.
//   The running maximum is only raised when some row's tile maximum exceeds it by more than `rescale_thr` (log2 units):
//   until then exp2 of a score may reach 2^thr instead of 1, which fp32 sums and h16 probabilities absorb, and the 32
//   accumulator multiplies per tile drop out of every tile but the first few (cdna guide T13).
// LDS: two buffers of K tile [64 keys][64 d] + V^T tile [64 d][64 keys], h16, XOR-swizzled 16-byte chunks; one barrier per tile.
// Work per (clip, layer): 4*T^2*64*H flops = 6.9 GFLOP (small); HBM: Q,K,V read once per
// 128-row block (K,V stay in ONE L2 across the 12 blocks of a head: XCD-aware block order), O written once.
#include "common.hpp"
#include <type_traits>

namespace axw {
inline namespace AXW_NS {

__device__ __forceinline__ int swz128(int row, int chunk) { return row * 128 + 16 * (chunk ^ ((row >> 1) & 7)); }

__global__ __launch_bounds__(256, 2) void encoder_attention_kernel(const h16* __restrict__ Q, const h16* __restrict__ K,
                                                                const h16* __restrict__ VT, h16* __restrict__ O, int T,
                                                                int t_pad, int d_model, int n_head, float rescale_thr) {
  __shared__ __attribute__((aligned(16))) char KVs[2][2][64 * 128];  // [buffer][K | V^T]: tile t+1 is stored while tile t is read

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware order (1-D grid): workgroups are dealt round-robin over the 8 XCDs, each with its own L2, so every XCD
  // gets a CONTIGUOUS chunk of the (clip, head, query block) space with the query block fastest: the 12 query blocks
  // of one (clip, head) then read its K and V^T through ONE L2 (dealt in launch order they sat on 8 different XCDs and
  // K/V were fetched 4.6 times: 2.7 GB per launch at 64 clips against 0.59 GB of distinct bytes).
  int qblk, head, b;
  {
    const int nq = (T + 127) / 128;
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    qblk = wg % nq;
    const int rest = wg / nq;
    head = rest % n_head;
    b = rest / n_head;
  }
  const int q_row = qblk * 128 + wave * 32 + r;
  const int q_ld = min(q_row, T - 1);

  const h16* Qb = Q + ((long)b * T) * d_model + head * 64;
  const h16* Kb = K + ((long)b * T) * d_model + head * 64;
  const h16* Vb = VT + ((long)b * n_head + head) * 64 * t_pad;

  h16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const h16x8*>(Qb + (long)q_ld * d_model + 16 * s + 8 * h);

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ld_row = tid >> 3, ld_c = tid & 7;
  const int nkt = t_pad / 64;
  u32x4 rk[2], rv[2];
  // K / V^T tiles through buffer loads: scalar resource of this (clip, head), one constant 32-bit offset per lane and
  // row, the tile as one add (K) or a scalar offset (V^T) — next to no vector address arithmetic in a kernel bound by vector issue.
  // K's resource ends behind row T - 1, so the rows of the last tile at or beyond T read as zeros (they are masked).
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (unsigned)(((long)T * d_model - head * 64) * 2), 0x27000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (unsigned)(64 * t_pad * 2), 0x27000);
  int offk[2], offv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    offk[i] = ((ld_row + 32 * i) * d_model + ld_c * 8) * 2;
    offv[i] = ((ld_row + 32 * i) * t_pad + ld_c * 8) * 2;
  }
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, offk[i] + kt * 64 * d_model * 2, 0, 0);  // in the lane offset: the range check ignores the scalar one
      rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsV, offv[i], kt * 64 * 2, 0);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(KVs[buf][0] + swz128(ld_row + 32 * i, ld_c)) = rk[i];
      *reinterpret_cast<u32x4*>(KVs[buf][1] + swz128(ld_row + 32 * i, ld_c)) = rv[i];
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();

  const float sc = 0.125f * 1.44269504088896340736f;  // (64^-0.25)^2 * log2(e)

  // One 64-key tile out of LDS: scores, online softmax, P.V. TAIL (a compile-time tag) masks the keys >= T: only the
  // last tile has any, and as a run-time test the compiler evaluated the 32 compare/select pairs on every tile.
  // The raw MFMA scores stay unscaled: exp2(s*sc - m*sc) is one FMA per element into v_exp, the running maximum is
  // kept in raw-score units (sc > 0).
  auto process_tile = [&](int kt, auto tail_tag, const char* Ks, const char* Vs) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    // ---- S^T = K Q^T : sacc[kb][e] = score(key = kt*64 + kb*32 + (e&3) + 8*(e>>2) + 4h, query = lane r)
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        h16x8 kf = *reinterpret_cast<const h16x8*>(Ks + swz128(kb * 32 + r, 2 * s + h));
        sacc[kb] = AXW_MFMA_32x32x16(kf, qf[s], sacc[kb]);
      }
    }
    // ---- online softmax (fp32, base-2)
    float mt = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if constexpr (TAIL) {
          const int key = kt * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= T) sacc[kb][e] = -INFINITY;
        }
        mt = fmaxf(mt, sacc[kb][e]);
      }
    {
      auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt = fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
    }
    // raise the running maximum (raw-score units; every tile has at least one unmasked key) only if some row needs it
    if (__builtin_amdgcn_ballot_w64((mt - m_run) * sc > rescale_thr) != 0) {
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
    }
    const float m_sc = m_run * sc;
    float ls = 0.f;
    h16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kb][e], sc, -m_sc));
        ls += pv;
        pf[kb][e >> 3][e & 7] = (h16)pv;
      }
    l_run += ls;

    // ---- O^T += V^T P^T : A element j of lane half h = V^T[d][key = kb*32 + 16*s2 + 8*(j>>2) + 4h + (j&3)] = the
    // 8 h16 at positions 8h .. 8h+7 of that 16-group in the stored frame order
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const h16x8 vf = *reinterpret_cast<const h16x8*>(Vs + swz128(db * 32 + r, 4 * kb + 2 * s2 + h));
          oacc[db] = AXW_MFMA_32x32x16(vf, pf[kb][s2], oacc[db]);
        }
  };
  // Peeled so the staging registers are assigned unconditionally inside the loop (a conditional prefetch makes
  // hipcc keep them in scratch memory and serialise every load).
  // (two tiles per iteration: the buffer of a tile is a compile-time constant, its LDS addresses immediates)
  int kt = 0;
  for (; kt + 2 < nkt; kt += 2) {
    load_tile(kt + 1);
    process_tile(kt, std::false_type{}, KVs[0][0], KVs[0][1]);
    store_tile(1);  // the buffer tile kt-1 was read from: every wave is past the barrier that ended iteration kt-1
    __syncthreads();
    load_tile(kt + 2);
    process_tile(kt + 1, std::false_type{}, KVs[1][0], KVs[1][1]);
    store_tile(0);
    __syncthreads();
  }
  if (kt + 1 < nkt) {  // an even tile count: one more full tile in buffer 0 before the last one
    load_tile(kt + 1);
    process_tile(kt, std::false_type{}, KVs[0][0], KVs[0][1]);
    store_tile(1);
    __syncthreads();
    ++kt;
  }
  process_tile(nkt - 1, std::true_type{}, KVs[kt & 1][0], KVs[kt & 1][1]);  // t_pad - T < 64: only the last tile holds padded keys

  const float l = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l;
  if (q_row < T) {
    h16* orow = O + ((long)b * T + q_row) * d_model + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h16x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = (h16)(oacc[db][4 * g + e] * inv);
        *reinterpret_cast<h16x4*>(orow + db * 32 + 8 * g + 4 * h) = pk;
      }
  }
}

void launch_encoder_attention(const h16* q, const h16* k, const h16* vt, h16* o, int batch, int T, int t_pad, int d_model,
                              int n_head, hipStream_t s, float rescale_thr) {
  dim3 grid(((T + 127) / 128) * n_head * batch);
  hipLaunchKernelGGL(encoder_attention_kernel, grid, dim3(256), 0, s, q, k, vt, o, T, t_pad, d_model, n_head, rescale_thr);
}

}  // inline namespace AXW_NS
}  // namespace axw
