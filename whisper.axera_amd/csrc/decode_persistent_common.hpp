// decode_persistent_common.hpp — device helpers shared by the persistent decode launches (decode_persistent.hip: one clip
// per launch; decode_persistent2.hip: two clips per launch): lane-group reductions, {tag, value} granules and their polls,
// weight-row sets with the one-instruction publish, the 64-key attention block, the partial merge.
#pragma once
#include "common.hpp"

namespace axw {
inline namespace AXW_NS {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

constexpr int PT = 1024;           // threads per workgroup, one workgroup per CU
constexpr int NPW = 8, NCW = 8;    // poller waves, compute waves
constexpr int PL = NPW * 64;       // poller lanes
constexpr int CT = NCW * 64;       // compute threads
// A lane gives up on a hand-off by TIME: a real wait is microseconds, so the first kSpinFree polls (about a
// millisecond) never read the clock; after that the 100 MHz wall clock is sampled every 256 polls and the lane gives up
// kSpinTicks later (50 ms). A launch whose workgroups cannot all be resident (CUs taken by another process, a CU-masked
// stream, a partitioned device) therefore costs a request ~50 ms, not a second, before the engine falls back.
// Cache policy of the three streams of a decoder step (Whisper-small: 198 MB of layer weights, 80 MB of vocabulary
// rows, 55 MB of cross K/V against 256 MB of Infinity Cache + 32 MB of L2). The layer weights are the latency-critical
// loads (requested one hand-off ahead of their use) and are re-read every step: default policy, so that they are served
// from the Infinity Cache. The vocabulary rows are a once-per-step bandwidth-bound stream and the cross K/V tiles are
// requested a whole layer ahead: both non-temporal, so that they do not evict the layer weights. Measured (decode of
// one clip, A/B/A/B inside one GPU call): on one box all-default 121.0 / vocabulary nt 118.1 / both 118.1 ms, on
// another all-default 121.2 / vocabulary nt 121.2 / cross K/V nt 118.9 / both 119.1 ms — which of the two streams
// matters differs between boxes (allocation placement), both together are within 0.3 ms of the better everywhere.
// Every weight row nt: 129.1 ms (the layer weights do live in the cache between steps).
#ifndef AXW_VOCAB_NT
#define AXW_VOCAB_NT 1
#endif
#ifndef AXW_KV_NT_LDS
#define AXW_KV_NT_LDS 1
#endif
constexpr bool kVocabNT = AXW_VOCAB_NT != 0;
constexpr int kKvAux = AXW_KV_NT_LDS ? 2 : 0;  // aux bits of global_load_lds: 2 = nt
constexpr int kSpinFree = 1024;
constexpr long long kSpinTicks = 5000000;
constexpr int kPS = 66;            // attention partial record in LDS: m, l, o[64]
constexpr int kRec = 80;           // cross-attention partial record as granules: o[64] (four full lines), m, l; 5-line stride
constexpr int kCrossSplit = 3;     // cross-attention key ranges per head (8 blocks of 64 keys each = 8 compute waves)
constexpr int kKvBytes = 2 * NCW * 8192;  // LDS K/V region: K [8 blk][8][64][8] h16 + V [512 keys][64] h16

// ---------------------------------------------------------------------------------------- lane-group reductions
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over aligned groups of LPR lanes (16, 32 or 64), result in every lane of the group; every lane of the wave
// must be active. DPP butterflies inside a 16-lane row, v_permlane{16,32}_swap across rows.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  if constexpr (LPR >= 32) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if constexpr (LPR >= 64) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}
__device__ __forceinline__ float wsum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wmax(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
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

// ---------------------------------------------------------------------------------------- granules
__device__ __forceinline__ void gput(u64* g, unsigned tag, float v) {
  __hip_atomic_store((gu64*)g, ((u64)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gput_u(u64* g, unsigned tag, unsigned v) {
  __hip_atomic_store((gu64*)g, ((u64)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gget(const u64* g) {
  return __hip_atomic_load((gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned eget(const unsigned* e) {
  return __hip_atomic_load((gu32*)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Pair polls: 16-byte sc1 loads of two adjacent granules (profiles/microbench/publish_shape.cpp: a hand-off gathered
// with 16-byte polls completes 0.25 us earlier than with 8-byte ones). pidx(k) = granule index of the k-th pair of this
// lane (even), < 0: none. v[2k], v[2k+1] = the two values. Returns true on give-up.
template <int NP, typename IDX>
__device__ __forceinline__ bool gather2(__amdgpu_buffer_rsrc_t rs, unsigned tag, unsigned (&v)[2 * NP], const unsigned* err, const int* ctl, IDX pidx) {
  bool ok[NP];
  int ix[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) { ix[k] = pidx(k); ok[k] = ix[k] < 0; v[2 * k] = 0u; v[2 * k + 1] = 0u; }
  long long t_start = 0;
  for (int spins = 0;; ++spins) {
    bool all = true;
    u32x4 x[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) if (!ok[k]) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ix[k] * 8, 0, 16);  // aux 16 = sc1
#pragma unroll
    for (int k = 0; k < NP; ++k)
      if (!ok[k]) {
        if (x[k][1] == tag && x[k][3] == tag) { v[2 * k] = x[k][0]; v[2 * k + 1] = x[k][2]; ok[k] = true; } else all = false;
      }
    if (all) return false;
    if ((spins & 63) == 63 && *(volatile const int*)ctl) return true;  // a wave of this workgroup gave up
    if ((spins & 255) == 255 && spins >= kSpinFree) {
      if (eget(err)) return true;                                      // another workgroup gave up: leave as well
      const long long now = wall_clock64();
      if (t_start == 0) t_start = now;
      else if (now - t_start > kSpinTicks) return true;
    }
  }
}

// Lane `tid` collects granules idx(k) for k < MAXG (idx < 0: none) of epoch `tag`; returns true on give-up.
template <int MAXG, typename IDX>
__device__ __forceinline__ bool gather(const u64* buf, unsigned tag, unsigned (&v)[MAXG], const unsigned* err, const int* ctl, IDX idx) {
  bool ok[MAXG];
  int ix[MAXG];
#pragma unroll
  for (int k = 0; k < MAXG; ++k) { ix[k] = idx(k); ok[k] = ix[k] < 0; v[k] = 0u; }
  long long t_start = 0;
  for (int spins = 0;; ++spins) {
    bool all = true;
    u64 x[MAXG];
#pragma unroll
    for (int k = 0; k < MAXG; ++k) if (!ok[k]) x[k] = gget(buf + ix[k]);  // independent loads, one round trip
#pragma unroll
    for (int k = 0; k < MAXG; ++k)
      if (!ok[k]) {
        if ((unsigned)(x[k] >> 32) == tag) { v[k] = (unsigned)x[k]; ok[k] = true; } else all = false;
      }
    if (all) return false;
    if ((spins & 63) == 63 && *(volatile const int*)ctl) return true;  // a wave of this workgroup gave up
    if ((spins & 255) == 255 && spins >= kSpinFree) {
      if (eget(err)) return true;                                      // another workgroup gave up: leave as well
      const long long now = wall_clock64();
      if (t_start == 0) t_start = now;
      else if (now - t_start > kSpinTicks) return true;
    }
  }
}


// ---------------------------------------------------------------------------------------- weight rows
template <int LPR, int CH, bool NT = false>
__device__ __forceinline__ void rows_load(u32x4 (&w)[CH], const h16* W, int K, int row, int tid) {
  const int j = tid % LPR;
  const h16* wr = W + (long)row * K;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    if constexpr (NT) w[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr + (j + LPR * i) * 8));
    else w[i] = *reinterpret_cast<const u32x4*>(wr + (j + LPR * i) * 8);
  }
}
#define AXW_FMA8(ACC0, ACC1, U, X0, X1)                       \
  ACC0 = fmaf(h16lo(U[0]), X0.x, ACC0);       \
  ACC1 = fmaf(h16hi(U[0]), X0.y, ACC1); \
  ACC0 = fmaf(h16lo(U[1]), X0.z, ACC0);       \
  ACC1 = fmaf(h16hi(U[1]), X0.w, ACC1); \
  ACC0 = fmaf(h16lo(U[2]), X1.x, ACC0);       \
  ACC1 = fmaf(h16hi(U[2]), X1.y, ACC1); \
  ACC0 = fmaf(h16lo(U[3]), X1.z, ACC0);       \
  ACC1 = fmaf(h16hi(U[3]), X1.w, ACC1);
// dot product of one weight row (registers) with the activation vector in LDS; LPR lanes share the row
template <int LPR, int CH>
__device__ __forceinline__ float rows_dot(const u32x4 (&w)[CH], const float* act, int tid) {
  const int j = tid % LPR;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const float4 x0 = *reinterpret_cast<const float4*>(act + (j + LPR * i) * 8);
    const float4 x1 = *reinterpret_cast<const float4*>(act + (j + LPR * i) * 8 + 4);
    AXW_FMA8(a0, a1, w[i], x0, x1)
  }
  return group_sum<LPR>(a0 + a1);
}
template <int LPR, int CH>
__device__ __forceinline__ float rows_dot_reg(const u32x4 (&w)[CH], const float4 (&a)[CH][2]) {
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) { AXW_FMA8(a0, a1, w[i], a[i][0], a[i][1]) }
  return group_sum<LPR>(a0 + a1);
}

// ---------------------------------------------------------------------------------------- workgroup barrier
// s_barrier with only the LDS counter drained. HIP's __syncthreads() also drains vmcnt, which would make every
// barrier wait for the weight rows that were just requested for the NEXT phase.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------- row sets
// The rows of one linear layer that this workgroup computes: slot = ctid / LPR handles row r0 + slot (+ k * slots),
// ctid = thread index among the compute waves. prefetch() issues the loads of the first pass one phase ahead.
template <int LPR, int CH>
struct RowSet {
  u32x4 w[CH];
  float bias;
  float sfac;  // run_ln only: s_j = sum_i W_ji g_i of this slot's row (the LayerNorm folded into the rows)
  int r0, r1;
  // first < 0: rows dealt evenly over all workgroups; else one full pass (SLOTS rows) per producer, producers =
  // workgroups first, first + 1, ... (the others get no rows)
  __device__ __forceinline__ void prefetch(const h16* W, const float* b, int K, int N, int wg, int P, int ctid, int first = -1) {
    if (first < 0) {
      r0 = (int)((unsigned)wg * (unsigned)N / (unsigned)P);  // wg * N < 2^31 (256 workgroups x 51866 rows)
      r1 = (int)((unsigned)(wg + 1) * (unsigned)N / (unsigned)P);
    } else {
      constexpr int SLOTS = CT / LPR;
      int pidx = wg - first;
      if (pidx < 0) pidx += P;
      r0 = pidx * SLOTS < N ? pidx * SLOTS : 0;
      r1 = pidx * SLOTS < N ? (r0 + SLOTS < N ? r0 + SLOTS : N) : 0;
    }
    const int slot = ctid / LPR, j = ctid % LPR;
    const int row = r0 + slot;
    rows_load<LPR, CH>(w, W, K, row < r1 ? row : r0, ctid);
    bias = (b && row < r1 && j == 0) ? b[row] : 0.f;
  }
  // The same for rows that carry their LayerNorm (decode_persistent.hip, round 5): bias = c_j = W beta + b, sfac = s_j = W g
  __device__ __forceinline__ void prefetch_ln(const h16* W, const float* sv, const float* cv, int K, int N, int wg, int P, int ctid, int first) {
    prefetch(W, cv, K, N, wg, P, ctid, first);
    const int slot = ctid / LPR, j = ctid % LPR, row = r0 + slot;
    sfac = (row < r1 && j == 0) ? sv[row] : 0.f;
  }
  // act holds g . x (not normalised): res = rstd (W (g . x) - mean s) + c. One pass, or two where the rows are dealt evenly.
  __device__ __forceinline__ void run_ln(const h16* W, const float* sv, const float* cv, int K, const float* act, int ctid, float (&res)[2], float mean, float rstd) {
    constexpr int SLOTS = CT / LPR;
    const int slot = ctid / LPR, j = ctid % LPR;
    res[0] = rstd * (rows_dot<LPR, CH>(w, act, ctid) - mean * sfac) + bias;
    res[1] = 0.f;
    const int row1 = r0 + slot + SLOTS;
    if (row1 < r1) {
      rows_load<LPR, CH>(w, W, K, row1, ctid);
      const float c1 = j == 0 ? cv[row1] : 0.f, s1 = j == 0 ? sv[row1] : 0.f;
      int ctid2 = ctid;
      asm volatile("" : "+v"(ctid2));
      res[1] = rstd * (rows_dot<LPR, CH>(w, act, ctid2) - mean * s1) + c1;
    }
  }
  // Computes this slot's rows (at most two passes: every supported shape has <= 2 * slots rows per workgroup) into
  // res[]. The caller requests the NEXT phase's rows before it publishes: a write-through store in front of a load
  // holds the load back for about a microsecond.
  __device__ __forceinline__ void run(const h16* W, const float* b, int K, const float* act, int ctid, float (&res)[2]) {
    constexpr int SLOTS = CT / LPR;
    const int slot = ctid / LPR, j = ctid % LPR;
    res[0] = rows_dot<LPR, CH>(w, act, ctid) + bias;
    res[1] = 0.f;
    const int row1 = r0 + slot + SLOTS;
    if (row1 < r1) {
      rows_load<LPR, CH>(w, W, K, row1, ctid);
      const float b1 = (b && j == 0) ? b[row1] : 0.f;
      int ctid2 = ctid;
      asm volatile("" : "+v"(ctid2));  // re-read the activations from LDS: keeping them live across both passes spills
      res[1] = rows_dot<LPR, CH>(w, act, ctid2) + b1;
    }
  }
  // Publishes this workgroup's rows (contiguous granules r0..r1-1 of `buf`) with ONE store instruction: the slot
  // leaders drop f(result) into pk[] (LDS), every compute wave bumps an LDS counter, and the wave that arrives last
  // stores all rows. Several waves each storing a few granules of the same 128-byte lines cost the hand-off 1.7 us
  // (profiles/microbench/publish_shape.cpp: 48 producers x 16 rows, 3.4 -> 1.75 us per phase).
  template <typename F>
  __device__ __forceinline__ void publish(int ctid, const float (&res)[2], float* pk, int* cnt, u64* buf, unsigned tag, F f) const {
    constexpr int SLOTS = CT / LPR;
    const int slot = ctid / LPR, j = ctid % LPR, lane = ctid & 63;
    if (j == 0) {
      if (r0 + slot < r1) pk[slot] = f(res[0]);
      if (r0 + slot + SLOTS < r1) pk[slot + SLOTS] = f(res[1]);
    }
    __builtin_amdgcn_wave_barrier();
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if ((old + 1) % NCW == 0 && lane < r1 - r0) gput(buf + r0 + lane, tag, pk[lane]);
  }
};

// sum over the lanes that share (lane & 7): lane bits 3, 4, 5
__device__ __forceinline__ float sum_hi3(float v) {
  v += dpp_mov<0x128>(v);  // row_ror:8
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

// One 16-byte piece of a K/V block that lives in GLOBAL memory (the two-clip launch keeps clip 1's self-attention cache there,
// written by this workgroup's pollers with plain stores): a buffer load with sc0, so it is served by L2 and never by a line
// that L1 still holds from an earlier step. base: the block (wave-uniform), off: this lane's h16 offset inside it.
__device__ __forceinline__ u32x4 kv_global16(const h16* base, int off) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 8192, 0x27000);  // one 64-key block
  return __builtin_amdgcn_raw_buffer_load_b128(rs, off * 2, 0, 1);  // aux 1 = sc0
}
// One wave, one block of 64 keys in LDS: kblk = [8 (d/8)][64 keys][8] h16 (lane = key for the scores). Writes the
// softmax partial (m, l, o[64]) to part[0..66). qp: the query as packed h16 pairs, [32] hi then [32] lo (q = hi + lo).
// pw: 64 dwords of wave-private LDS scratch.
// The block is VALU-bound (two compute waves share a SIMD; skipping its arithmetic altogether shortens the decode of
// one clip by 12.6 %), so it is written for instruction count: the scores are v_dot2c dot products of the packed K
// dwords with the packed query (2 instructions per 2 dims instead of 4), and for the self-attention cache, whose LDS
// layout is this kernel's own, V is kept TRANSPOSED (VT: vblk = [8 (key/8)][64 dims][8 keys]) so that lane = dim
// accumulates o[dim] with dot2 over key pairs against the packed probabilities — no unpacking, no cross-lane sums.
// Cross-attention V tiles arrive by LDS-DMA in the HBM layout [64 keys][64 dims] and keep the lane = (key row, dim
// chunk) form.
template <bool VT>
__device__ __forceinline__ void attn_block(const h16* kblk, const h16* vblk, const unsigned* qp, bool valid, float* pw, float* part, int lane) {
#ifdef AXW_ATTN_SKIP  // timing-only build (wrong results): bounds what any speed-up of this block's arithmetic can buy
  if (lane == 0) { part[0] = 0.f; part[1] = 1.f; }
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[2 + lane * 8 + e] = 0.f;
  }
  return;
#endif
  float sc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int i = 0; i < 8; ++i) {
    const u32x4 kq = *reinterpret_cast<const u32x4*>(kblk + i * 512 + lane * 8);
    const u32x4 qh = *reinterpret_cast<const u32x4*>(qp + i * 4), ql = *reinterpret_cast<const u32x4*>(qp + 32 + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sc[e] = h16dot2(kq[e], qh[e], sc[e]);
      sc[e] = h16dot2(kq[e], ql[e], sc[e]);
    }
  }
  float s = ((sc[0] + sc[1]) + (sc[2] + sc[3])) * 0.125f;  // (64^-0.25)^2, export_onnx.py:116,124-126
  if (!valid) s = -INFINITY;
  const float m = wmax(s);  // -inf only for a block without a single valid key
  const float pk = m > -INFINITY ? __expf(s - m) : 0.f;
  const float lsum = wsum(pk);
  if constexpr (VT) {
    // probabilities as packed (hi, lo) h16 in wave-private LDS: key k -> half-word k of ph (dwords 0..31) / pl (32..63)
    const h16 ph = (h16)pk, pl = (h16)(pk - (float)ph);
    reinterpret_cast<h16*>(pw)[lane] = ph;
    reinterpret_cast<h16*>(pw + 32)[lane] = pl;
    __builtin_amdgcn_wave_barrier();
    float o0 = 0.f, o1 = 0.f;
    const unsigned* pwu = reinterpret_cast<const unsigned*>(pw);
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {  // keys 8i..8i+7 of dim `lane`
      const u32x4 vv = *reinterpret_cast<const u32x4*>(vblk + i * 512 + lane * 8);
      const u32x4 h4 = *reinterpret_cast<const u32x4*>(pwu + i * 4), l4 = *reinterpret_cast<const u32x4*>(pwu + 32 + i * 4);
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        o0 = h16dot2(vv[e], h4[e], o0);
        o0 = h16dot2(vv[e], l4[e], o0);
        o1 = h16dot2(vv[e + 1], h4[e + 1], o1);
        o1 = h16dot2(vv[e + 1], l4[e + 1], o1);
      }
    }
    if (lane == 0) { part[0] = m; part[1] = lsum; }
    part[2 + lane] = o0 + o1;
  } else {
    pw[(lane & 7) * 8 + (lane >> 3)] = pk;  // key k = 8i + r -> pw[r*8 + i]
    __builtin_amdgcn_wave_barrier();
    const float4 p0 = *reinterpret_cast<const float4*>(pw + (lane >> 3) * 8), p1 = *reinterpret_cast<const float4*>(pw + (lane >> 3) * 8 + 4);
    const float pr[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // V row of key 8i + (lane>>3), dims (lane&7)*8 .. +8
      const u32x4 vv = *reinterpret_cast<const u32x4*>(vblk + (8 * i + (lane >> 3)) * 64 + (lane & 7) * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = fmaf(pr[i], h16lo(vv[e]), o[2 * e]);
        o[2 * e + 1] = fmaf(pr[i], h16hi(vv[e]), o[2 * e + 1]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = sum_hi3(o[e]);
    if (lane == 0) { part[0] = m; part[1] = lsum; }
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) part[2 + lane * 8 + e] = o[e];
    }
  }
}

// attn_block<true> on a block whose 16 pieces are already in registers (kr[i]: dims 8i..8i+7 of key `lane`; vr[i]: keys
// 8i..8i+7 of dim `lane`): the same operations in the same order, so a block gives the same bits from either home.
__device__ __forceinline__ void attn_block_regs(const u32x4 (&kr)[8], const u32x4 (&vr)[8], const unsigned* qp, bool valid, float* pw, float* part, int lane) {
  float sc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32x4 qh = *reinterpret_cast<const u32x4*>(qp + i * 4), ql = *reinterpret_cast<const u32x4*>(qp + 32 + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sc[e] = h16dot2(kr[i][e], qh[e], sc[e]);
      sc[e] = h16dot2(kr[i][e], ql[e], sc[e]);
    }
    if (i & 1) __builtin_amdgcn_sched_barrier(0);  // the block's 64 registers leave room for two pieces of the query at a time
  }
  float s = ((sc[0] + sc[1]) + (sc[2] + sc[3])) * 0.125f;
  if (!valid) s = -INFINITY;
  const float m = wmax(s);
  const float pk = m > -INFINITY ? __expf(s - m) : 0.f;
  const float lsum = wsum(pk);
  const h16 ph = (h16)pk, pl = (h16)(pk - (float)ph);
  reinterpret_cast<h16*>(pw)[lane] = ph;
  reinterpret_cast<h16*>(pw + 32)[lane] = pl;
  __builtin_amdgcn_wave_barrier();
  float o0 = 0.f, o1 = 0.f;
  const unsigned* pwu = reinterpret_cast<const unsigned*>(pw);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32x4 h4 = *reinterpret_cast<const u32x4*>(pwu + i * 4), l4 = *reinterpret_cast<const u32x4*>(pwu + 32 + i * 4);
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      o0 = h16dot2(vr[i][e], h4[e], o0);
      o0 = h16dot2(vr[i][e], l4[e], o0);
      o1 = h16dot2(vr[i][e + 1], h4[e + 1], o1);
      o1 = h16dot2(vr[i][e + 1], l4[e + 1], o1);
    }
    if (i & 1) __builtin_amdgcn_sched_barrier(0);
  }
  if (lane == 0) { part[0] = m; part[1] = lsum; }
  part[2 + lane] = o0 + o1;
}

// merge nb (<= NCW) wave partials (m, l, o[64]) in LDS: returns (l, o[c]) rescaled to the common maximum *m_out.
// Unrolled over all NCW records with blocks >= nb masked: every LDS read goes out at once and the exponentials are
// independent (as a loop over a run-time nb this was nb dependent read -> exp -> FMA round trips on the one wave
// that every consumer of the attention output waits for).
__device__ __forceinline__ void merge_partials(const float* wpart, int nb, int c, float* m_out, float* l_out, float* o_out) {
  float mb[NCW], lb[NCW], ob[NCW];
#pragma unroll
  for (int b = 0; b < NCW; ++b) {
    const float mv = wpart[b * kPS], lv = wpart[b * kPS + 1], ov = wpart[b * kPS + 2 + c];
    const bool on = b < nb;
    mb[b] = on ? mv : -INFINITY;
    lb[b] = on ? lv : 0.f;
    ob[b] = on ? ov : 0.f;
  }
  float m = mb[0];
#pragma unroll
  for (int b = 1; b < NCW; ++b) m = fmaxf(m, mb[b]);
  float lt = 0.f, ov = 0.f;
#pragma unroll
  for (int b = 0; b < NCW; ++b) {
    const float f = mb[b] > -INFINITY ? __expf(mb[b] - m) : 0.f;
    lt = fmaf(f, lb[b], lt);
    ov = fmaf(f, ob[b], ov);
  }
  *m_out = m; *l_out = lt; *o_out = ov;
}


// ---------------------------------------------------------------------------------------- the query fold's pieces (decode_persistent.hip)
// fp32 weight rows of M: LPR lanes share a row, 2 * CH chunks of 4 floats per lane (element layout of rows_dot).
template <int LPR, int CH>
struct RowSetF32 {
  u32x4 w[2 * CH];
  float bias;
  __device__ __forceinline__ void prefetch(const float* W, const float* b, int K, int row, bool on, int ctid) {
    const int j = ctid % LPR;
    const float* wr = W + (long)row * K;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      w[2 * i] = *reinterpret_cast<const u32x4*>(wr + (j + LPR * i) * 8);
      w[2 * i + 1] = *reinterpret_cast<const u32x4*>(wr + (j + LPR * i) * 8 + 4);
    }
    bias = (on && j == 0) ? b[row] : 0.f;
  }
  __device__ __forceinline__ float run(const float* act, int ctid) const {
    const int j = ctid % LPR;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const float4 x0 = *reinterpret_cast<const float4*>(act + (j + LPR * i) * 8);
      const float4 x1 = *reinterpret_cast<const float4*>(act + (j + LPR * i) * 8 + 4);
      a0 = fmaf(__uint_as_float(w[2 * i][0]), x0.x, a0); a1 = fmaf(__uint_as_float(w[2 * i][1]), x0.y, a1);
      a0 = fmaf(__uint_as_float(w[2 * i][2]), x0.z, a0); a1 = fmaf(__uint_as_float(w[2 * i][3]), x0.w, a1);
      a0 = fmaf(__uint_as_float(w[2 * i + 1][0]), x1.x, a0); a1 = fmaf(__uint_as_float(w[2 * i + 1][1]), x1.y, a1);
      a0 = fmaf(__uint_as_float(w[2 * i + 1][2]), x1.z, a0); a1 = fmaf(__uint_as_float(w[2 * i + 1][3]), x1.w, a1);
    }
    return group_sum<LPR>(a0 + a1) + bias;
  }
};

// per-layer block of the fold arena (floats): M [D][D], then d, s, c [D] each, then the LayerNorm fold of the two LayerNorm-fed
// row phases (below): s_qkv, c_qkv [3D], s_fc1, c_fc1 [4D]
__host__ __device__ constexpr long qfold_stride(int d) { return (long)d * d + 17L * d; }
constexpr int QF_SQKV = 3, QF_CQKV = 6, QF_SFC1 = 9, QF_CFC1 = 13;  // vector offsets behind M, in units of D

// One cross-attention unit (a clip's head, one third of the 1536 padded keys) on the eight compute waves, behind the barrier that
// handed over the query: every wave runs its 64-key block from the LDS tiles (export_onnx.py:221-230: fp32 softmax, no mask but
// the padding), the wave that arrives last merges the eight partials and publishes the record — o[64] as four full lines, then
// (m, l). out: the record's granules; cnt: the LDS arrival counter.
__device__ __forceinline__ void cross_unit_block(const h16* sK, const h16* sV, const unsigned* qs, float* pscr, float* wpart, int* cnt, u64* out,
                                                 unsigned tag, int ca_split, int n_audio_ctx, int cw, int lane) {
  // this wave's own K/V tiles have landed. The builtin, not inline asm: behind an asm that may touch the counters the compiler
  // drains vmcnt at every following join (measured: +18 ms on Whisper-small for one such asm in a cold path)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  asm volatile("" ::: "memory");
  const int key = (ca_split * NCW + cw) * 64 + lane;
  attn_block<false>(sK + cw * 4096, sV + cw * 4096, qs, key < n_audio_ctx, pscr + cw * 64, wpart + cw * kPS, lane);
  __builtin_amdgcn_wave_barrier();
  int old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  old = __builtin_amdgcn_readfirstlane(old);
  if ((old + 1) % NCW == 0) {
    float m, lt, ov;
    merge_partials(wpart, NCW, lane, &m, &lt, &ov);
    gput(out + lane, tag, ov);
    if (lane < 2) gput(out + 64 + lane, tag, lane == 0 ? m : lt);
  }
}

// The row producers' merge of a head's kCrossSplit partial records (pbuf: [H][kCrossSplit][kPS] = o[64], m, l each) into the
// attention vector element i of this lane (softmax over the whole key range: rescale to the common maximum, then normalise).
__device__ __forceinline__ float merge_cross_records(const float* pbuf, int i) {
  const float* pp = pbuf + (i >> 6) * kCrossSplit * kPS;
  float m = pp[64];
#pragma unroll
  for (int sp = 1; sp < kCrossSplit; ++sp) m = fmaxf(m, pp[sp * kPS + 64]);
  float lt = 0.f, ov = 0.f;
#pragma unroll
  for (int sp = 0; sp < kCrossSplit; ++sp) {
    const float ms = pp[sp * kPS + 64];
    const float f = ms > -INFINITY ? __expf(ms - m) : 0.f;
    lt += f * pp[sp * kPS + 65];
    ov += f * pp[sp * kPS + (i & 63)];
  }
  return ov / lt;
}

// A cross-attention unit's side of the query fold: wave 0 gathers the head's 64 T values (lanes 0-31, pairs at granule base_cq)
// and the row producers' statistics (lanes 32.., up to two producers each, one 16-granule line per producer at base_stat),
// derives mu / r of x1 = x0 + y1 (shift = the mean of x0, the same bits in every workgroup) and leaves the query
// cq_j = r (T_j - mu s_j) + c_j as packed (hi, lo) h16 pairs in qs. vec: the layer's fold vectors behind M (d, s, c at 0, D, 2D).
// Every wave of the pollers calls it (the gather's give-up logic is workgroup-wide); returns true on give-up.
template <int D, int NP_D>
__device__ __forceinline__ bool qfold_unit_query(__amdgpu_buffer_rsrc_t GR, unsigned tag, int tid, int base_cq, int base_stat, const float* vec,
                                                 int head, float shift, unsigned* qs, const unsigned* err, const int* ctl) {
  constexpr int SL = (NP_D + 1) / 2;  // lanes that hold statistics
  static_assert(32 + SL <= 64, "statistics lanes");
  float sj[2] = {0.f, 0.f}, cj[2] = {0.f, 0.f};
  if (tid < 32) {
    const float2 s2 = *reinterpret_cast<const float2*>(vec + D + head * 64 + 2 * tid);
    const float2 c2 = *reinterpret_cast<const float2*>(vec + 2 * D + head * 64 + 2 * tid);
    sj[0] = s2.x; sj[1] = s2.y; cj[0] = c2.x; cj[1] = c2.y;
  }
  unsigned v[4];
  const bool fail = gather2<2>(GR, tag, v, err, ctl, [&](int k2) {
    if (tid < 32) return k2 == 0 ? base_cq + head * 64 + 2 * tid : -1;
    const int pi = (tid - 32) + k2 * SL;
    return (tid < 32 + SL && pi < NP_D) ? base_stat + 16 * pi : -1;
  });
  if (tid < 64) {
    const bool st = tid >= 32;
    const float t1 = wsum(st ? __uint_as_float(v[0]) + __uint_as_float(v[2]) : 0.f);
    const float t2 = wsum(st ? __uint_as_float(v[1]) + __uint_as_float(v[3]) : 0.f);
    const float dm = t1 / D, var = fmaxf(t2 / D - dm * dm, 0.f);
    const float mu = shift + dm, r = rsqrtf(var + 1e-5f);
    if (tid < 32) {
      const float q0 = r * (__uint_as_float(v[0]) - mu * sj[0]) + cj[0], q1 = r * (__uint_as_float(v[1]) - mu * sj[1]) + cj[1];
      unsigned hi, lo;
      h16split2(q0, q1, hi, lo);
      qs[tid] = hi;
      qs[32 + tid] = lo;
    }
  }
  return fail;
}

// A row producer's side: the slot leaders have left y1 in pk[slot], T in pk[32 + slot] and x1 - shift in pscr[slot]; the compute
// wave that arrives last (LDS counter cnt) stores T, the two sums and y1 — three lines, one store instruction each, the ones
// the units wait for first. Gc: the clip's granule area; o_cq / o_stat / o_y1: buffer offsets; r0: first row, nrows <= 32.
__device__ __forceinline__ void qfold_publish(int lane, const float* pk, const float* pscr, int* cnt, u64* Gc, int o_cq, int o_stat, int o_y1,
                                              int r0, int nrows, int producer, unsigned tag) {
  __builtin_amdgcn_wave_barrier();
  int old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  old = __builtin_amdgcn_readfirstlane(old);
  if ((old + 1) % NCW == 0) {
    const bool on = lane < nrows;
    const float tv = on ? pscr[lane] : 0.f;
    const float s1 = wsum(tv), s2 = wsum(tv * tv);
    if (on) gput(Gc + o_cq + r0 + lane, tag, pk[32 + lane]);
    if (lane < 2) gput(Gc + o_stat + 16 * producer + lane, tag, lane == 0 ? s1 : s2);
    if (on) gput(Gc + o_y1 + r0 + lane, tag, pk[lane]);
  }
}

}  // inline namespace AXW_NS
}  // namespace axw
