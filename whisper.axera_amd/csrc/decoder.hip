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
// Kernels (decode is HBM-bound: weights are h16 and read exactly once per step, K/V caches h16):
//   embed_kernel            x = tok_emb[token] + pos[step]                       (export_onnx.py:334-336)
//   gemv_kernel<LPR,BT,CH>  y = W a (+bias) for B <= 4 rows at a time, fp32 activations x h16
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
inline namespace AXW_NS {

constexpr int kPartStride = 66;  // m, l, o[64]  (decode_gemv.hip merges these partials)
constexpr int kAttnSplitMax = 6;  // workgroups per (clip, head) whose partials a launch folds itself (Engine::kCrossSplitMax)

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short u) { return __uint_as_float((unsigned)u << 16); }

// ------------------------------------------------------------------------------- embed
__global__ __launch_bounds__(256) void embed_kernel(const h16* __restrict__ tok_emb, const float* __restrict__ pos,
                                                    const int* __restrict__ tok, const int* __restrict__ off, float* __restrict__ x,
                                                    int d) {
  const int b = blockIdx.x;
  const int t = tok[b], step = off[b];
  for (int c = threadIdx.x; c < d; c += 256) x[(long)b * d + c] = (float)tok_emb[(long)t * d + c] + pos[(long)step * d + c];
}

void launch_embed(const h16* tok_emb, const float* pos, const int* tok, const int* off, float* x, int batch, int d,
                  hipStream_t s) {
  hipLaunchKernelGGL(embed_kernel, dim3(batch), dim3(256), 0, s, tok_emb, pos, tok, off, x, d);
}

// ------------------------------------------------------------------------------- decode attention
// FUSE_Q: the workgroup computes its own 64 query values, q = Wq[head*64 .. +64][:] . LayerNorm(x[b]) + bq (fp32 FMA on
// h16 weights, export_onnx.py:221-230), instead of reading them from a preceding GEMM launch: the 98 KB of weight rows
// come from L2 (the 64 clips of a head share them) while the first K/V block is already in flight, and a decoder
// layer loses one dependent launch.
// K/V blocks are read once per decoder step and a step streams 4 GB of them at 64 clips: the loads are non-temporal
// (do not keep the lines), so that the 198 MB of layer weights the GEMMs in between re-read every step stay in the
// Infinity Cache. Measured at 64 clips: decode 552 -> 522 ms, the attention launches 5.1 -> 5.5 TB/s.
// base: wave-uniform h16 pointer of this (clip, head)'s K or V, off: element offset of this lane. A buffer load (scalar
// resource + 32-bit lane offset + immediate) instead of a 64-bit global address per load: the eight loads of a block
// share one offset register (64 clips: step 1.153 -> 1.136 ms; the cache-policy bits nt / nt+sc1 / sc0+nt+sc1 measure alike).
__device__ __forceinline__ uint4 ld_kv(const h16* base, long off) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x27000);
  const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 2), 0, 2);  // aux 2 = nt
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// STAMP (measurement builds of the kernel only): thread 0 of every workgroup records its start and end time, so that a decoder step's attention launches can be placed on one time axis while two graph
// branches run them side by side (a profiler serialises the branches; hipEvents see only whole replays).
// NCHL > 0 (FUSE_Q at d_model 384 / 512 / 768 / 1024): d_model = 32 * NCHL, the lane's
// NCHL weight chunks of the query projection are ALL requested up front, unconditionally — one memory round trip and straight-line
// code instead of one round trip per 8 chunks behind per-chunk bounds checks (which hipcc turns into a branch per load).
// QM (query mode): 0 = q is read from p.q; 1 = FUSE_Q above; 2 = FOLDED (decode_gemm.hip "QUERY FOLD"): the head's 64 query values are
// r (T - mu s) + c from the clip's T row (written by the o launch), the 48 block statistics of its residual row and two constant
// vectors — one round trip of small loads behind the first K/V requests, one barrier, no weights.
template <int QM, bool STAMP = false, int NCHL = 0>
__global__ __launch_bounds__(256) void decode_attention_kernel(DecAttnParams p, int cap_blocks, int stamp_point) {
  constexpr bool FUSE_Q = QM == 1;
  __shared__ float s_part[4][kPartStride];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the block loop below is wave-uniform control flow
  const int split = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  // (one {begin, end} pair per WORKGROUP, plain stores; the host takes the minimum and the maximum: 768 atomics on one
  // address per launch cost ~6 us per launch and moved the branches apart)
  unsigned long long* my_stamp = nullptr;
  if constexpr (STAMP) {
    my_stamp = p.stamp + 2 * ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    if (tid == 0) my_stamp[0] = (unsigned long long)wall_clock64();
  }
  auto stamp_end = [&] { if constexpr (STAMP) { if (tid == 0 && stamp_point == 0) my_stamp[1] = (unsigned long long)wall_clock64(); } };
  // AX_WHISPER_ATTN_STAMP_POINT=k (measurement builds): the "end" stamp is taken at point k of the workgroup's path instead
  auto stamp_at = [&](int k) { if constexpr (STAMP) { if (tid == 0 && stamp_point == k) my_stamp[1] = (unsigned long long)wall_clock64(); } };
  // A clip that has reached its eot keeps its slot in the batch but streams no K/V any more (the reference stops each
  // utterance at its own eot, Whisper.cpp:219-222). Its stale attention output feeds linear layers whose results
  // nobody reads: rows of different clips never mix, and advance_kernel re-seeds x[b] from the embedding every step.
  // The flag is REQUESTED here (an unconditional scalar load: the pointer is never null) and looked at behind the first K/V — and, with
  // FUSE_Q, activation and weight — requests: a launch of the few-clip step is one dependent chain, and a finished clip's extra
  // requests are one block of its own, allocated, K/V.
  // At batch (done_late == 0) it is looked at first: a finished clip of a ragged batch then costs nothing at all (64 clips with
  // budgets of 60-150 ids: 334 clips/s, against 323 with the late check everywhere).
  const int clip_done = p.done[b];
  if (!p.done_late && clip_done) { stamp_end(); return; }
  const int bps = (cap_blocks + p.n_split - 1) / p.n_split;
  const int blk_begin = split * bps, blk_cap_end = min(cap_blocks, blk_begin + bps);

  const float* __restrict__ qp = p.q + (long)b * p.d_model + head * 64;  // wave-uniform: scalar loads
  const h16* kb = p.k + (long)b * p.kv_batch_stride + (long)head * cap_blocks * 4096;
  const h16* vb = p.v + (long)b * p.kv_batch_stride + (long)head * cap_blocks * 4096;

  // The first block's K/V loads go out before the step counter is even known: every block below
  // cap_blocks is allocated (and zero-initialised), keys beyond n_keys are masked afterwards.
  uint4 kn[8], vn[8];
  int blk = blk_begin + wave;
  // keys at or beyond `limit` are not fetched (they are masked anyway): the 36 rows that pad 1500 keys to 24 blocks of
  // 64 (2.3 % of the cross-attention bytes) and, on average, half a block of the self-attention cache per wave.
  // Lanes whose key is at or beyond `limit` read the block's last valid key instead (the same cache lines as their
  // neighbours: no extra bytes, and no per-lane predicate around the loads, which hipcc turns into one branch per load).
  // A block that is loaded at all has at least one key below the limit.
  auto load_k = [&](int bk, int limit) {
    const int lk = max(0, min(lane, limit - 1 - bk * 64));
#pragma unroll
    for (int i = 0; i < 8; ++i) kn[i] = ld_kv(kb, (long)bk * 4096 + i * 512 + lk * 8);
  };
  auto load_v = [&](int bk, int limit) {
    const int last = max(bk * 64, limit - 1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      vn[i] = ld_kv(vb, (long)min(bk * 64 + 8 * i + (lane >> 3), last) * 64 + (lane & 7) * 8);
  };
  // cross-attention knows its key count; self-attention does not yet (the step counter is a load): its first block is
  // fetched whole (every block below cap_blocks is allocated and zero-initialised)
  if (blk < blk_cap_end) {
    load_k(blk, p.n_keys >= 0 ? p.n_keys : 0x7fffffff);
    load_v(blk, p.n_keys >= 0 ? p.n_keys : 0x7fffffff);
  }
  float qv[64];
  stamp_at(1);
  if constexpr (FUSE_Q) {
    __shared__ __attribute__((aligned(16))) float s_act[1024];
    __shared__ float s_q[64];
    __shared__ float s_red[8];
    const int d = p.d_model;  // <= 1024, multiple of 32
    const float* xr = p.x + (long)b * d;
    float xv[4], gg[4], bb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = tid + 256 * e;
      xv[e] = c < d ? xr[c] : 0.f;
      gg[e] = c < d ? p.ln_w[c] : 0.f;
      bb[e] = c < d ? p.ln_b[c] : 0.f;
    }
    // the first weight chunks of this lane's row: 4 lanes share a row, lane j takes the 16-byte chunks j, j+4, ...
    const int qrow = tid >> 2, qj = tid & 3;
    const h16* wr = p.wq + (long)(head * 64 + qrow) * d;
    const int nch = d >> 3;  // 16-byte chunks per row
    constexpr int NWC = NCHL > 0 ? NCHL : 8;
    u32x4 wc[NWC];
#pragma unroll
    for (int i = 0; i < NWC; ++i) {
      const int c = qj + 4 * i;
      if constexpr (NCHL > 0) wc[i] = *reinterpret_cast<const u32x4*>(wr + c * 8);
      else wc[i] = c < nch ? *reinterpret_cast<const u32x4*>(wr + c * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    const float bq = p.bq[head * 64 + qrow];
    if (clip_done) { stamp_end(); return; }  // (workgroup-uniform, in front of the first barrier)
    // statistics in one pass behind one barrier (round 6, as in the clip-block GEMM prologue: decode_gemm.hip): var = E[x^2] - mean^2
    float s1 = (xv[0] + xv[1]) + (xv[2] + xv[3]);
    float s2 = (xv[0] * xv[0] + xv[1] * xv[1]) + (xv[2] * xv[2] + xv[3] * xv[3]);  // (elements beyond d were loaded as 0)
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) { s_red[wave] = s1; s_red[4 + wave] = s2; }
    __syncthreads();
    const float mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / d;
    const float rstd = rsqrtf(fmaxf(((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])) / d - mean * mean, 0.f) + 1e-5f);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int c = tid + 256 * e; if (c < d) s_act[c] = (xv[e] - mean) * rstd * gg[e] + bb[e]; }
    __syncthreads();
    stamp_at(2);
    float a0 = 0.f, a1 = 0.f;
    if constexpr (NCHL > 0) {
      float b0 = 0.f, b1 = 0.f;  // four independent chains
#pragma unroll
      for (int i = 0; i < NCHL; ++i) {
        const int c = qj + 4 * i;
        const float4 y0 = *reinterpret_cast<const float4*>(s_act + c * 8), y1 = *reinterpret_cast<const float4*>(s_act + c * 8 + 4);
        a0 = fmaf(h16lo(wc[i][0]), y0.x, a0);
        a1 = fmaf(h16hi(wc[i][0]), y0.y, a1);
        b0 = fmaf(h16lo(wc[i][1]), y0.z, b0);
        b1 = fmaf(h16hi(wc[i][1]), y0.w, b1);
        a0 = fmaf(h16lo(wc[i][2]), y1.x, a0);
        a1 = fmaf(h16hi(wc[i][2]), y1.y, a1);
        b0 = fmaf(h16lo(wc[i][3]), y1.z, b0);
        b1 = fmaf(h16hi(wc[i][3]), y1.w, b1);
      }
      a0 += b0; a1 += b1;
    }
    for (int c0 = 0; c0 < nch && NCHL == 0; c0 += 32) {  // 8 chunks of this lane per pass
      u32x4 cur[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) cur[i] = wc[i];
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int c = c0 + 32 + qj + 4 * i; wc[i] = c < nch ? *reinterpret_cast<const u32x4*>(wr + c * 8) : u32x4{0u, 0u, 0u, 0u}; }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + qj + 4 * i;
        if (c < nch) {
          const float4 y0 = *reinterpret_cast<const float4*>(s_act + c * 8), y1 = *reinterpret_cast<const float4*>(s_act + c * 8 + 4);
          a0 = fmaf(h16lo(cur[i][0]), y0.x, a0);
          a1 = fmaf(h16hi(cur[i][0]), y0.y, a1);
          a0 = fmaf(h16lo(cur[i][1]), y0.z, a0);
          a1 = fmaf(h16hi(cur[i][1]), y0.w, a1);
          a0 = fmaf(h16lo(cur[i][2]), y1.x, a0);
          a1 = fmaf(h16hi(cur[i][2]), y1.y, a1);
          a0 = fmaf(h16lo(cur[i][3]), y1.z, a0);
          a1 = fmaf(h16hi(cur[i][3]), y1.w, a1);
        }
      }
    }
    float acc = a0 + a1;
    acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
    acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
    if (qj == 0) s_q[qrow] = acc + bq;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 64; ++c) qv[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_q[c])));
  } else if constexpr (QM == 2) {
    __shared__ float s_q[64];
    const int d = p.d_model, nst = d >> 4;  // nst <= 64 (d_model <= 1024)
    float tq = 0.f, sj = 0.f, cj = 0.f;
    f32x2_t sp = {0.f, 0.f};
    if (wave == 0) {  // the first wave builds the query: T, s, c of its lane's dimension and one block statistic each
      tq = p.tq[(long)b * d + head * 64 + lane];
      sj = p.fold_s[head * 64 + lane];
      cj = p.fold_c[head * 64 + lane];
      if (lane < nst) sp = *reinterpret_cast<const f32x2_t*>(p.stat_part + ((long)b * nst + lane) * 2);
    }
    if (clip_done) { stamp_end(); return; }  // (workgroup-uniform, in front of the barrier)
    if (wave == 0) {
      // Chan et al.: the row's mean from the block sums, its centred squares from the blocks' own + 16 (block mean - mean)^2
      const float mean = wave_sum(sp[0]) / (float)d;
      const float dm = sp[0] * (1.f / 16.f) - mean;
      const float m2 = wave_sum(lane < nst ? sp[1] + 16.f * dm * dm : 0.f);
      const float rstd = rsqrtf(m2 / (float)d + 1e-5f);
      s_q[lane] = rstd * (tq - mean * sj) + cj;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 64; ++c) qv[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_q[c])));
  } else {
#pragma unroll
    for (int c = 0; c < 64; ++c) qv[c] = qp[c];
  }

  stamp_at(3);
  const int n_keys = p.n_keys >= 0 ? p.n_keys : p.off[b] + 1;  // self-attention: this clip's own position
  if (QM == 0 && clip_done) { stamp_end(); return; }  // (behind the query and position requests: one round trip for all three)
  const int blk_end = min((n_keys + 63) >> 6, blk_cap_end);

  float m_w = -INFINITY, l_lane = 0.f;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;

  // One register set per operand (K 32 + V 32 VGPRs instead of current + next of both): the next block's K is requested
  // as soon as the scores of this one are formed and lands under the softmax and the P.V arithmetic, the next V right
  // after P.V. The requests sit behind a wave-uniform `if (more)`, and hipcc joins the loaded registers with the old
  // ones at the end of the iteration (`s_waitcnt vmcnt(0)` + copies), so V's latency is exposed once per block. The
  // peeled form (unconditional requests, counted waits, loads crossing the iteration: 366 instead of 524 instructions
  // per block) measured the same per attention launch (14.7 us) but a SLOWER decoder step at 64 clips (1.157 against
  // 1.140 ms, twice in one call): the launch-latency-bound GEMM chain of the other graph branch shares the memory
  // queues with these loads, and the deeper this kernel keeps them, the longer each dependent GEMM launch takes.
  auto one_block = [&](int cur, bool prefetch) {
    // lane = key cur*64 + lane: dot(q, k) over 64 dims
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned u[4] = {kn[i].x, kn[i].y, kn[i].z, kn[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc = fmaf(qv[i * 8 + 2 * e], h16lo(u[e]), sc);
        sc = fmaf(qv[i * 8 + 2 * e + 1], h16hi(u[e]), sc);
      }
    }
    if (prefetch) load_k(cur + 4, n_keys);
    sc *= 0.125f;  // (64^-0.25)^2, export_onnx.py:116,124-126
    if (cur * 64 + lane >= n_keys) sc = -INFINITY;
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
      const unsigned u[4] = {vn[i].x, vn[i].y, vn[i].z, vn[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = fmaf(w, h16lo(u[e]), o[2 * e]);
        o[2 * e + 1] = fmaf(w, h16hi(u[e]), o[2 * e + 1]);
      }
    }
    if (prefetch) load_v(cur + 4, n_keys);
  };
  for (; blk < blk_end; blk += 4) one_block(blk, blk + 4 < blk_end);
  stamp_at(4);
  // wave partial: sum o over the 8 key sub-rows (lanes with equal lane&7), sum l over the wave
  const float l_w = wave_sum(l_lane);
#pragma unroll
  for (int e = 0; e < 8; ++e) {  // lanes l, l^8, l^16, l^32 ...: a DPP rotation inside the 16-lane row, then the row swaps
    o[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o[e]), 0x128, 0xf, 0xf, true));  // row_ror:8
    {
      auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[e]), __float_as_uint(o[e]), false, false);
      o[e] = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    }
    {
      auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[e]), __float_as_uint(o[e]), false, false);
      o[e] = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    }
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
    stamp_at(5);
    if (p.out_hi && p.n_split > 1) {
      // Splits of one (clip, head) in the batched path at FEW clips (3 clips x 12 heads are 36 workgroups for 256 CUs,
      // each streaming its 24 key blocks one after the other). Every split publishes (m, l, o[64]) with write-through
      // stores, drains them, and draws a ticket; whoever draws the last one folds all of them IN SPLIT ORDER (the
      // result does not depend on who arrives last) and writes the output.
      float* base = p.mpart + ((long)b * p.n_head + head) * p.n_split * kPartStride;
      float* mine = base + split * kPartStride;
      __hip_atomic_store(mine + 2 + tid, ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid == 0) {
        __hip_atomic_store(mine, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp_at(6);
      // The hand-off is the first row of MI355X_MICROARCH.md's table of hand-offs that need no acquire: every payload
      // store is sc1 (write-through), the storing wave drains vmcnt, ONE lane adds to an agent-scope counter, and the
      // workgroup whose add came last — told by the value its add returned — reads the others' records with sc1 loads
      // after that add has returned (the loads sit behind a branch on the returned value; the signal fences pin the
      // compiler's ordering on both sides). An acq_rel RMW instead (buffer_wbl2 + buffer_inv per workgroup, what the C++
      // memory model would ask for) was measured at turbo dims, 16 clips, 2 splits: decode 151.5 -> 197.0 ms (round 3).
      unsigned ticket = 0;
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      if (tid == 0) {
        ticket = __hip_atomic_fetch_add(p.mcnt + b * p.n_head + head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      stamp_at(7);
      if (ticket != (unsigned)p.n_split - 1u) { stamp_end(); return; }  // (the whole wave; the other waves are past their last use of LDS)
      // every other split's record is requested before the first one is used: ONE memory round trip for the fold, not one per
      // split (five dependent sc1 round trips at six splits were a quarter of this launch at four clips)
      float ms[kAttnSplitMax], ls[kAttnSplitMax], os[kAttnSplitMax];
#pragma unroll
      for (int s2 = 0; s2 < kAttnSplitMax; ++s2) {
        ms[s2] = m; ls[s2] = l; os[s2] = ov;
        if (s2 < p.n_split && s2 != split) {
          const float* other = base + s2 * kPartStride;
          ms[s2] = __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ls[s2] = __hip_atomic_load(other + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          os[s2] = __hip_atomic_load(other + 2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      float M = -INFINITY, Ls = 0.f, O = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < kAttnSplitMax; ++s2) {
        if (s2 < p.n_split) {
          const float m2 = ms[s2], l2 = ls[s2], o2 = os[s2];
          const float mn = fmaxf(M, m2);
          const float f1 = M > -INFINITY ? __expf(M - mn) : 0.f, f2 = m2 > -INFINITY ? __expf(m2 - mn) : 0.f;
          Ls = f1 * Ls + f2 * l2;
          O = f1 * O + f2 * o2;
          M = mn;
        }
      }
      stamp_at(8);
      if (tid == 0) __hip_atomic_store(p.mcnt + b * p.n_head + head, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
      const float y = O / Ls;
      const h16 yh = (h16)y;
      const int k = head * 64 + tid;
      const long i = ((((long)(k >> 5) * p.nbs + (b >> 4)) * 64) + ((k >> 3) & 3) * 16 + (b & 15)) * 8 + (k & 7);
      p.out_hi[i] = yh;
      p.out_lo[i] = (h16)(y - (float)yh);
    } else if (p.out_hi) {  // single split: this IS the attention output (batched decode path)
      const float y = ov / l;
      const h16 yh = (h16)y;
      const int k = head * 64 + tid;  // fragment-major pair (layout: decode_gemm.hip)
      const long i = ((((long)(k >> 5) * p.nbs + (b >> 4)) * 64) + ((k >> 3) & 3) * 16 + (b & 15)) * 8 + (k & 7);
      p.out_hi[i] = yh;
      p.out_lo[i] = (h16)(y - (float)yh);
    } else {
      float* out = p.part + (((long)b * p.n_head + head) * p.n_split + split) * kPartStride;
      if (tid == 0) { out[0] = m; out[1] = l; }
      out[2 + tid] = ov;
    }
    stamp_end();  // thread 0 sits in the wave that writes the output: the last thing a workgroup does
  }
}

void launch_decode_attention(const DecAttnParams& p, hipStream_t s) {
  static const int stamp_point = [] { const char* e = getenv("AX_WHISPER_ATTN_STAMP_POINT"); return e ? atoi(e) : 0; }();
  if (p.tq) {  // folded query
    if (p.d_model > 1024 || p.d_model % 64 != 0 || !p.stat_part || !p.fold_s || !p.fold_c || p.n_split > kAttnSplitMax ||
        (p.n_split != 1 && !(p.out_hi && p.mpart && p.mcnt))) { fprintf(stderr, "[ax_whisper] folded query: d_model %d, n_split %d unsupported\n", p.d_model, p.n_split); abort(); }
    const dim3 grid(p.n_split, p.n_head, p.batch);
    if (p.stamp) hipLaunchKernelGGL((decode_attention_kernel<2, true>), grid, dim3(256), 0, s, p, p.cap_blocks, stamp_point);
    else hipLaunchKernelGGL((decode_attention_kernel<2>), grid, dim3(256), 0, s, p, p.cap_blocks, stamp_point);
  } else if (p.wq) {
    if (p.d_model > 1024 || p.d_model % 32 != 0 || p.n_split > kAttnSplitMax || (p.n_split != 1 && !(p.out_hi && p.mpart && p.mcnt))) { fprintf(stderr, "[ax_whisper] fused query projection: d_model %d, n_split %d unsupported\n", p.d_model, p.n_split); abort(); }
    static const bool qall = [] { const char* e = getenv("AX_WHISPER_ATTN_QALL"); return !(e && e[0] == '0'); }();  // A/B
    const dim3 grid(p.n_split, p.n_head, p.batch);
    // the projection's weight chunks all up front, at every clip count: 14 more registers leave the same two workgroups per CU, and
    // the launch is shorter at few clips (4 clips 11.3 -> 10.5 us) and at batch (64 clips: decode 510.8 -> 500.1 ms per call, 16: -1.5 %)
    const bool few = qall;
    if (p.stamp) {
      if (few && p.d_model == 768) hipLaunchKernelGGL((decode_attention_kernel<1, true, 24>), grid, dim3(256), 0, s, p, p.cap_blocks, stamp_point);
      else hipLaunchKernelGGL((decode_attention_kernel<1, true>), grid, dim3(256), 0, s, p, p.cap_blocks, stamp_point);
    } else if (few && p.d_model == 384) hipLaunchKernelGGL((decode_attention_kernel<1, false, 12>), grid, dim3(256), 0, s, p, p.cap_blocks, stamp_point);
    else if (few && p.d_model == 512) hipLaunchKernelGGL((decode_attention_kernel<1, false, 16>), grid, dim3(256), 0, s, p, p.cap_blocks, stamp_point);
    else if (few && p.d_model == 768) hipLaunchKernelGGL((decode_attention_kernel<1, false, 24>), grid, dim3(256), 0, s, p, p.cap_blocks, stamp_point);
    else if (few && p.d_model == 1024) hipLaunchKernelGGL((decode_attention_kernel<1, false, 32>), grid, dim3(256), 0, s, p, p.cap_blocks, stamp_point);
    else hipLaunchKernelGGL((decode_attention_kernel<1>), grid, dim3(256), 0, s, p, p.cap_blocks, stamp_point);
  } else {
    if (p.stamp) hipLaunchKernelGGL((decode_attention_kernel<0, true>), dim3(p.n_split, p.n_head, p.batch), dim3(256), 0, s, p, p.cap_blocks, stamp_point);
    else hipLaunchKernelGGL((decode_attention_kernel<0>), dim3(p.n_split, p.n_head, p.batch), dim3(256), 0, s, p, p.cap_blocks, stamp_point);
  }
}

// ------------------------------------------------------------------------------- weight preparation
__device__ __forceinline__ float load_as_f32(const void* src, int dt, long i) {
  if (dt == 0) return reinterpret_cast<const float*>(src)[i];
  if (dt == 1) return bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(src)[i]);
  return (float)reinterpret_cast<const _Float16*>(src)[i];
}
__global__ void convert_to_bf16_kernel(const void* src, int dt, h16* dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = (h16)load_as_f32(src, dt, i);
}
__global__ void convert_to_f32_kernel(const void* src, int dt, float* dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = load_as_f32(src, dt, i);
}
// Conv1d weight [Cout][Cin][3] -> GEMM weight [Cout][kpad], column k*Cin + c (zero beyond 3*Cin)
__global__ void conv_weight_pack_kernel(const void* src, int dt, h16* dst, int cout, int cin, int kpad) {
  const long total = (long)cout * kpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int n = (int)(i / kpad), col = (int)(i - (long)n * kpad);
    float v = 0.f;
    if (col < 3 * cin) {
      int k = col / cin, c = col - k * cin;
      v = load_as_f32(src, dt, ((long)n * cin + c) * 3 + k);
    }
    dst[i] = (h16)v;
  }
}
static int grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }
void launch_convert_to_h16(const void* src, int dt, h16* dst, long n, hipStream_t s) {
  hipLaunchKernelGGL(convert_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dt, dst, n);
}
void launch_convert_to_f32(const void* src, int dt, float* dst, long n, hipStream_t s) {
  hipLaunchKernelGGL(convert_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dt, dst, n);
}
void launch_conv_weight_pack(const void* src, int dt, h16* dst, int cout, int cin, int kpad, hipStream_t s) {
  hipLaunchKernelGGL(conv_weight_pack_kernel, dim3(grid_for((long)cout * kpad)), dim3(256), 0, s, src, dt, dst, cout, cin, kpad);
}

}  // inline namespace AXW_NS
}  // namespace axw
