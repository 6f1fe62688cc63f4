// decode_persistent.hip — the whole greedy decode loop of ONE clip as a single persistent launch (gfx950).
//
// Replaces, for batch 1, the per-token sequence Whisper::run_decoder (cpp/src/Whisper.cpp:290-346) + argmax (:42-45)
// + the host loop (:207-222) = the decoder graph TextDecoderTensorCache.forward (model_convert/export_onnx.py:312-387)
// with its attention (:103-150, :221-230). Same arithmetic as the launch-per-phase path (decode_gemv.hip,
// decoder.hip): bf16 weights x fp32 activations with fp32 FMA, fp32 LayerNorm/softmax, bf16 self/cross K/V.
//
// Why: at batch 1 a decoder step is a chain of ~100 dependent phases that each move only a few MB, so the chain is
// bound by per-phase latency, not by HBM. As separate graph nodes a phase costs ~4.4 us (kernel
// boundary + wave start + activation round trip + drain). Here one workgroup per CU stays resident for the whole
// utterance and the phases hand their outputs over INSIDE the launch:
//   * every output element travels as one 8-byte {tag, value} granule written by ONE sc1 (write-through) store and
//     polled with sc1 loads — the data is the flag, no fence, no separate barrier
//     (cdna_hip_programming.md §6 Guideline 16 form R2; tags = step * n_layer + layer + 1, never 0, buffers zeroed
//     before every launch);
//   * each workgroup issues the weight loads of its rows BEFORE it starts polling, so the HBM round trip of the
//     weights hides behind the hand-off; cross-attention K/V tiles (constant during the utterance) are prefetched
//     into registers two phases ahead;
//   * the self-attention K/V cache of one (layer, head) lives in the LDS of the workgroup that owns that head for
//     the whole utterance (448 keys x 64 x 2 x bf16 = 112 KB): it never touches HBM;
//   * every workgroup keeps its own copy of the residual stream, so a LayerNorm needs no extra hand-off;
//   * the token feedback (argmax merge, SOT forcing, eot / context stop, embedding of the next token) is computed
//     redundantly by every workgroup from the gathered argmax partials: the loop never returns to the host.
// Every spin is bounded: a workgroup that waits too long sets an error word and leaves; the others see the word
// (or time out themselves) and leave too, so the grid always drains. The host then throws.
#include "common.hpp"

namespace axw {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int PT = 512;            // threads per workgroup (8 waves, one workgroup per CU: 256 VGPRs per lane)
constexpr int PW = PT / 64;
constexpr int kSpinMax = 1 << 21;  // polls before a lane gives up (~1 s; a real wait is microseconds)
constexpr int kPS = 66;            // attention partial record: m, l, o[64]
constexpr int kCrossSplit = 3;     // cross-attention key ranges per head (8 blocks of 64 keys each = 8 waves)
constexpr int kSelfBlocks = 7;     // 448 / 64

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

// Lane `tid` collects granules idx(k) for k < MAXG (idx < 0: none) of epoch `tag`; returns true on give-up.
template <int MAXG, typename IDX>
__device__ __forceinline__ bool gather(const u64* buf, unsigned tag, unsigned (&v)[MAXG], const unsigned* err, IDX idx) {
  bool ok[MAXG];
  int ix[MAXG];
#pragma unroll
  for (int k = 0; k < MAXG; ++k) { ix[k] = idx(k); ok[k] = ix[k] < 0; v[k] = 0u; }
  for (int spins = 0; spins < kSpinMax; ++spins) {
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
    if ((spins & 1023) == 1023 && eget(err)) return true;  // somebody gave up: leave as well
  }
  return true;
}


// ---------------------------------------------------------------------------------------- weight rows
template <int LPR, int CH>
__device__ __forceinline__ void rows_load(u32x4 (&w)[CH], const bf16* W, int K, int row) {
  const int j = threadIdx.x % LPR;
  const bf16* wr = W + (long)row * K;
#pragma unroll
  for (int i = 0; i < CH; ++i) w[i] = *reinterpret_cast<const u32x4*>(wr + (j + LPR * i) * 8);
}
#define AXW_FMA8(ACC0, ACC1, U, X0, X1)                       \
  ACC0 = fmaf(__uint_as_float(U[0] << 16), X0.x, ACC0);       \
  ACC1 = fmaf(__uint_as_float(U[0] & 0xffff0000u), X0.y, ACC1); \
  ACC0 = fmaf(__uint_as_float(U[1] << 16), X0.z, ACC0);       \
  ACC1 = fmaf(__uint_as_float(U[1] & 0xffff0000u), X0.w, ACC1); \
  ACC0 = fmaf(__uint_as_float(U[2] << 16), X1.x, ACC0);       \
  ACC1 = fmaf(__uint_as_float(U[2] & 0xffff0000u), X1.y, ACC1); \
  ACC0 = fmaf(__uint_as_float(U[3] << 16), X1.z, ACC0);       \
  ACC1 = fmaf(__uint_as_float(U[3] & 0xffff0000u), X1.w, ACC1);
// dot product of one weight row (registers) with the activation vector in LDS; LPR lanes share the row
template <int LPR, int CH>
__device__ __forceinline__ float rows_dot(const u32x4 (&w)[CH], const float* act) {
  const int j = threadIdx.x % LPR;
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

// ---------------------------------------------------------------------------------------- the kernel
// d_model = 8*LD*CD (rows with K = d: LD lanes x CD 16-byte chunks), 4*d_model = 8*LF*CF.
template <int LD, int CD, int LF, int CF>
__global__ __launch_bounds__(PT) void decode_persistent_kernel(PersistParams p) {
  constexpr int D = 8 * LD * CD, F = 8 * LF * CF, H = D / 64;
  static_assert(F == 4 * D, "mlp width");
  constexpr int GD = (D + PT - 1) / PT, GF = (F + PT - 1) / PT, NPART = H * kCrossSplit * kPS, GP = (NPART + PT - 1) / PT;
  constexpr int SD = PT / LD, SF = PT / LF;  // row slots per pass
  // granule buffers (u64 units)
  constexpr int O_QKV = 0, O_ATT = 3 * D, O_Y1 = 4 * D, O_CQ = 5 * D, O_PART = 6 * D, O_Y2 = 10 * D, O_HID = 11 * D, O_Y3 = 15 * D,
                O_AMAX = 16 * D;
  static_assert(NPART <= 4 * D && NPART <= 3 * D + D / 8, "partial buffer");
  static_assert(kCrossSplit * PW == 24, "cross-attention key blocks");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sK = reinterpret_cast<bf16*>(smem);                  // [7 blk][8][64 keys][8]  (blocked, lane = key)
  bf16* sV = sK + kSelfBlocks * 4096;                        // [448 keys][64]
  float* xres = reinterpret_cast<float*>(sV + kSelfBlocks * 4096);  // [D] residual stream (own copy)
  float* act = xres + D;                                     // [F + D/8] input vector of the current rows phase
  float* wpart = act + F + D / 8;                            // [PW][kPS] per-wave attention partials
  float* red = wpart + PW * kPS;                             // [2*PW] LayerNorm partial sums
  float* qs = red + 2 * PW;                                  // [64] query of the attention phase
  float* am_v = qs + 64;                                     // [64] argmax scratch
  int* am_i = reinterpret_cast<int*>(am_v + 64);             // [64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = gridDim.x, wg = blockIdx.x;
  const int L = p.n_layer;
  u64* const G = p.gran;

  // self-attention ownership: unit (l, h) -> workgroup P-1-(l*H+h); cross: unit ((l*H+h)*2+s) -> workgroup unit % P
  const int sa_unit = P - 1 - wg;
  const int sa_layer = sa_unit < L * H ? sa_unit / H : -1, sa_head = sa_unit % H;

  // zero the LDS K/V cache: masked keys must hold finite values
  for (int i = tid; i < kSelfBlocks * 4096 * 2 / 8; i += PT) reinterpret_cast<u32x4*>(sK)[i] = u32x4{0u, 0u, 0u, 0u};

  int tok = p.sot[0];
  int n_out = 0, n_done = 0, steps_run = 0;
  float shift = 0.f;  // LayerNorm variance shift (previous mean): sums stay small without a second pass

#define AXW_GIVE_UP(CODE)                                                                    \
  {                                                                                          \
    if (tid == 0) __hip_atomic_store((gu32*)p.err, (unsigned)(CODE) | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    return;                                                                                  \
  }

  // LayerNorm of (xres += y) into act[0..D): y[k] belongs to element tid + k*PT. Two workgroup barriers.
  auto ln_stage = [&](const unsigned (&y)[GD], bool add, const float* g, const float* be, bool fail) -> bool {
    float xs[GD], gg[GD], bb[GD];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < GD; ++k) {
      const int i = tid + k * PT;
      if (i < D) {
        gg[k] = g[i]; bb[k] = be[i];
        xs[k] = xres[i] + (add ? __uint_as_float(y[k]) : 0.f);
        xres[i] = xs[k];
        const float t = xs[k] - shift;
        s1 += t; s2 += t * t;
      }
    }
    s1 = wsum(s1); s2 = wsum(s2);
    if (lane == 0) { red[2 * wave] = s1; red[2 * wave + 1] = s2; }
    if (__syncthreads_or(fail)) return true;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < PW; ++w) { t1 += red[2 * w]; t2 += red[2 * w + 1]; }
    const float dm = t1 / D, var = fmaxf(t2 / D - dm * dm, 0.f);
    const float mean = shift + dm, rstd = rsqrtf(var + 1e-5f);
#pragma unroll
    for (int k = 0; k < GD; ++k) {
      const int i = tid + k * PT;
      if (i < D) act[i] = (xs[k] - mean) * rstd * gg[k] + bb[k];
    }
    shift = mean;
    __syncthreads();
    return false;
  };

  for (int step = 0; step < p.total_steps; ++step) {
    // x = token_embedding[tok] + positional_embedding[step]   (export_onnx.py:334-336)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GD; ++k) {
      const int i = tid + k * PT;
      if (i < D) xres[i] = (float)p.tok_emb[(long)tok * D + i] + p.pos[(long)step * D + i];
    }
    // (made visible by the first barrier of ln_stage)

    for (int l = 0; l < L; ++l) {
      const DecLayerW& w = p.layers[l];
      const unsigned tag = (unsigned)(step * L + l + 1);

      // ================================================================= QKV rows (export_onnx.py:245-247)
      {
        constexpr int N = 3 * D;
        const int r0 = (int)((long)wg * N / P), r1 = (int)((long)(wg + 1) * N / P);
        const int slot = tid / LD, j = tid % LD;
        u32x4 wr[CD];
        int row = r0 + slot;
        rows_load<LD, CD>(wr, w.w_qkv, D, row < r1 ? row : r0);
        float bias = (row < r1 && j == 0) ? w.b_qkv[row] : 0.f;
        unsigned y[GD];
        bool fail = false;
        if (l > 0) fail = gather<GD>(G + O_Y3, tag - 1, y, p.err, [&](int k) { const int i = tid + k * PT; return i < D ? i : -1; });
        if (ln_stage(y, l > 0, w.attn_ln_w, w.attn_ln_b, fail)) AXW_GIVE_UP(0x100 + l)
        for (; row < r1; row += SD) {
          if (row != r0 + slot) { rows_load<LD, CD>(wr, w.w_qkv, D, row); bias = j == 0 ? w.b_qkv[row] : 0.f; }
          const float acc = rows_dot<LD, CD>(wr, act);
          if (j == 0) gput(G + O_QKV + row, tag, acc + bias);
        }
      }

      // ================================================================= self-attention of one head (export_onnx.py:103-147)
      // keys 0..step: the -60000 mask + the separate current-token column of the reference equal causal attention.
      if (l == sa_layer) {
        const int h = sa_head;
        unsigned v[1];
        const bool fail = gather<1>(G + O_QKV, tag, v, p.err, [&](int) { return tid < 192 ? (tid >> 6) * D + h * 64 + (tid & 63) : -1; });
        if (tid < 64) qs[tid] = __uint_as_float(v[0]);
        else if (tid < 128) {  // K row `step`, blocked [blk][d/8][key%64][8]
          const int dd = tid - 64;
          sK[(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (bf16)__uint_as_float(v[0]);
        } else if (tid < 192) {
          sV[step * 64 + (tid - 128)] = (bf16)__uint_as_float(v[0]);
        }
        if (__syncthreads_or(fail)) AXW_GIVE_UP(0x200 + l)
        const int nblk = (step >> 6) + 1;
        if (wave < nblk) {
          float sc0 = 0.f, sc1 = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const u32x4 kk = *reinterpret_cast<const u32x4*>(sK + (wave * 8 + i) * 512 + lane * 8);
            const float4 q0 = *reinterpret_cast<const float4*>(qs + i * 8), q1 = *reinterpret_cast<const float4*>(qs + i * 8 + 4);
            AXW_FMA8(sc0, sc1, kk, q0, q1)
          }
          float sc = (sc0 + sc1) * 0.125f;  // (64^-0.25)^2, export_onnx.py:116,124-126
          if (wave * 64 + lane > step) sc = -INFINITY;
          const float m = wmax(sc);
          const float pk = __expf(sc - m);
          const float lsum = wsum(pk);
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float wgt = __shfl(pk, 8 * i + (lane >> 3), 64);
            const u32x4 vv = *reinterpret_cast<const u32x4*>(sV + (wave * 64 + 8 * i + (lane >> 3)) * 64 + (lane & 7) * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[2 * e] = fmaf(wgt, __uint_as_float(vv[e] << 16), o[2 * e]);
              o[2 * e + 1] = fmaf(wgt, __uint_as_float(vv[e] & 0xffff0000u), o[2 * e + 1]);
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o[e] += __shfl_xor(o[e], 8, 64);
            o[e] += __shfl_xor(o[e], 16, 64);
            o[e] += __shfl_xor(o[e], 32, 64);
          }
          if (lane == 0) { wpart[wave * kPS] = m; wpart[wave * kPS + 1] = lsum; }
          if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) wpart[wave * kPS + 2 + lane * 8 + e] = o[e];
          }
        }
        __syncthreads();
        if (tid < 64) {
          float m = -INFINITY;
          for (int b = 0; b < nblk; ++b) m = fmaxf(m, wpart[b * kPS]);
          float lt = 0.f, ov = 0.f;
          for (int b = 0; b < nblk; ++b) {
            const float f = __expf(wpart[b * kPS] - m);
            lt += f * wpart[b * kPS + 1];
            ov += f * wpart[b * kPS + 2 + tid];
          }
          gput(G + O_ATT + h * 64 + tid, tag, ov / lt);
        }
      }

      // ---- cross-attention unit of this workgroup in this layer (K/V are constant: prefetch now, use two phases later)
      int ca_head = -1, ca_split = 0;
      {
        const int base = kCrossSplit * l * H;
        const int kk = wg >= base ? 0 : (base - wg + P - 1) / P;
        const int u = wg + kk * P;
        if (u >= base && u < base + kCrossSplit * H) { ca_head = (u - base) / kCrossSplit; ca_split = (u - base) % kCrossSplit; }
      }
      u32x4 ckr[8], cvr[8];
      if (ca_head >= 0) {
        const int kb = ca_split * PW + wave;  // 64-key block of this wave (24 blocks = t_pad 1536)
        const bf16* kbp = p.cross_k + (long)l * p.cross_layer_stride + (long)ca_head * 24 * 4096 + (long)kb * 4096;
        const bf16* vbp = p.cross_v + (long)l * p.cross_layer_stride + (long)ca_head * 24 * 4096 + (long)kb * 4096;
#pragma unroll
        for (int i = 0; i < 8; ++i) ckr[i] = *reinterpret_cast<const u32x4*>(kbp + i * 512 + lane * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) cvr[i] = *reinterpret_cast<const u32x4*>(vbp + (8 * i + (lane >> 3)) * 64 + (lane & 7) * 8);
      }

      // ================================================================= attention output projection
      {
        const int r0 = (int)((long)wg * D / P), r1 = (int)((long)(wg + 1) * D / P);
        const int slot = tid / LD, j = tid % LD;
        u32x4 wr[CD];
        int row = r0 + slot;
        rows_load<LD, CD>(wr, w.w_o, D, row < r1 ? row : r0);
        float bias = (row < r1 && j == 0) ? w.b_o[row] : 0.f;
        unsigned y[GD];
        const bool fail = gather<GD>(G + O_ATT, tag, y, p.err, [&](int k) { const int i = tid + k * PT; return i < D ? i : -1; });
#pragma unroll
        for (int k = 0; k < GD; ++k) { const int i = tid + k * PT; if (i < D) act[i] = __uint_as_float(y[k]); }
        if (__syncthreads_or(fail)) AXW_GIVE_UP(0x300 + l)
        for (; row < r1; row += SD) {
          if (row != r0 + slot) { rows_load<LD, CD>(wr, w.w_o, D, row); bias = j == 0 ? w.b_o[row] : 0.f; }
          const float acc = rows_dot<LD, CD>(wr, act);
          if (j == 0) gput(G + O_Y1 + row, tag, acc + bias);
        }
      }

      // ================================================================= cross-attention query (export_onnx.py:221-230)
      {
        const int r0 = (int)((long)wg * D / P), r1 = (int)((long)(wg + 1) * D / P);
        const int slot = tid / LD, j = tid % LD;
        u32x4 wr[CD];
        int row = r0 + slot;
        rows_load<LD, CD>(wr, w.w_cq, D, row < r1 ? row : r0);
        float bias = (row < r1 && j == 0) ? w.b_cq[row] : 0.f;
        unsigned y[GD];
        const bool fail = gather<GD>(G + O_Y1, tag, y, p.err, [&](int k) { const int i = tid + k * PT; return i < D ? i : -1; });
        if (ln_stage(y, true, w.cross_ln_w, w.cross_ln_b, fail)) AXW_GIVE_UP(0x400 + l)
        for (; row < r1; row += SD) {
          if (row != r0 + slot) { rows_load<LD, CD>(wr, w.w_cq, D, row); bias = j == 0 ? w.b_cq[row] : 0.f; }
          const float acc = rows_dot<LD, CD>(wr, act);
          if (j == 0) gput(G + O_CQ + row, tag, acc + bias);
        }
      }

      // ================================================================= cross-attention over one third of the 1536 padded keys
      if (ca_head >= 0) {
        unsigned v[1];
        const bool fail = gather<1>(G + O_CQ, tag, v, p.err, [&](int) { return tid < 64 ? ca_head * 64 + tid : -1; });
        if (tid < 64) qs[tid] = __uint_as_float(v[0]);
        if (__syncthreads_or(fail)) AXW_GIVE_UP(0x500 + l)
        float sc0 = 0.f, sc1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 q0 = *reinterpret_cast<const float4*>(qs + i * 8), q1 = *reinterpret_cast<const float4*>(qs + i * 8 + 4);
          AXW_FMA8(sc0, sc1, ckr[i], q0, q1)
        }
        float sc = (sc0 + sc1) * 0.125f;
        const int key = (ca_split * PW + wave) * 64 + lane;
        if (key >= p.n_audio_ctx) sc = -INFINITY;
        const float m = wmax(sc);  // may be -inf for a fully padded block
        const float pk = m > -INFINITY ? __expf(sc - m) : 0.f;
        const float lsum = wsum(pk);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float wgt = __shfl(pk, 8 * i + (lane >> 3), 64);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[2 * e] = fmaf(wgt, __uint_as_float(cvr[i][e] << 16), o[2 * e]);
            o[2 * e + 1] = fmaf(wgt, __uint_as_float(cvr[i][e] & 0xffff0000u), o[2 * e + 1]);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] += __shfl_xor(o[e], 8, 64);
          o[e] += __shfl_xor(o[e], 16, 64);
          o[e] += __shfl_xor(o[e], 32, 64);
        }
        if (lane == 0) { wpart[wave * kPS] = m; wpart[wave * kPS + 1] = lsum; }
        if (lane < 8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) wpart[wave * kPS + 2 + lane * 8 + e] = o[e];
        }
        __syncthreads();
        if (tid < 64) {
          float m2 = -INFINITY;
#pragma unroll
          for (int b = 0; b < PW; ++b) m2 = fmaxf(m2, wpart[b * kPS]);
          float lt = 0.f, ov = 0.f;
#pragma unroll
          for (int b = 0; b < PW; ++b) {
            const float mb = wpart[b * kPS];
            const float f = mb > -INFINITY ? __expf(mb - m2) : 0.f;
            lt += f * wpart[b * kPS + 1];
            ov += f * wpart[b * kPS + 2 + tid];
          }
          u64* out = G + O_PART + (ca_head * kCrossSplit + ca_split) * kPS;
          if (tid == 0) { gput(out, tag, m2); gput(out + 1, tag, lt); }
          gput(out + 2 + tid, tag, ov);
        }
      }

      // ================================================================= cross-attention output projection
      {
        const int r0 = (int)((long)wg * D / P), r1 = (int)((long)(wg + 1) * D / P);
        const int slot = tid / LD, j = tid % LD;
        u32x4 wr[CD];
        int row = r0 + slot;
        rows_load<LD, CD>(wr, w.w_co, D, row < r1 ? row : r0);
        float bias = (row < r1 && j == 0) ? w.b_co[row] : 0.f;
        unsigned y[GP];
        const bool fail = gather<GP>(G + O_PART, tag, y, p.err, [&](int k) { const int i = tid + k * PT; return i < NPART ? i : -1; });
        float* pbuf = act + D;  // [H][kCrossSplit][66]  (NPART <= 3.1 D floats behind act[0..D))
#pragma unroll
        for (int k = 0; k < GP; ++k) { const int i = tid + k * PT; if (i < NPART) pbuf[i] = __uint_as_float(y[k]); }
        if (__syncthreads_or(fail)) AXW_GIVE_UP(0x600 + l)
#pragma unroll
        for (int k = 0; k < GD; ++k) {
          const int i = tid + k * PT;
          if (i < D) {
            const float* pp = pbuf + (i >> 6) * kCrossSplit * kPS;
            float m = pp[0];
#pragma unroll
            for (int sp = 1; sp < kCrossSplit; ++sp) m = fmaxf(m, pp[sp * kPS]);
            float lt = 0.f, ov = 0.f;
#pragma unroll
            for (int sp = 0; sp < kCrossSplit; ++sp) {
              const float ms = pp[sp * kPS];
              const float f = ms > -INFINITY ? __expf(ms - m) : 0.f;
              lt += f * pp[sp * kPS + 1];
              ov += f * pp[sp * kPS + 2 + (i & 63)];
            }
            act[i] = ov / lt;
          }
        }
        __syncthreads();
        for (; row < r1; row += SD) {
          if (row != r0 + slot) { rows_load<LD, CD>(wr, w.w_co, D, row); bias = j == 0 ? w.b_co[row] : 0.f; }
          const float acc = rows_dot<LD, CD>(wr, act);
          if (j == 0) gput(G + O_Y2 + row, tag, acc + bias);
        }
      }

      // ================================================================= mlp.0 + GELU (export_onnx.py:298)
      {
        const int r0 = (int)((long)wg * F / P), r1 = (int)((long)(wg + 1) * F / P);
        const int slot = tid / LD, j = tid % LD;
        u32x4 wr[CD];
        int row = r0 + slot;
        rows_load<LD, CD>(wr, w.w_fc1, D, row < r1 ? row : r0);
        float bias = (row < r1 && j == 0) ? w.b_fc1[row] : 0.f;
        unsigned y[GD];
        const bool fail = gather<GD>(G + O_Y2, tag, y, p.err, [&](int k) { const int i = tid + k * PT; return i < D ? i : -1; });
        if (ln_stage(y, true, w.mlp_ln_w, w.mlp_ln_b, fail)) AXW_GIVE_UP(0x700 + l)
        for (; row < r1; row += SD) {
          if (row != r0 + slot) { rows_load<LD, CD>(wr, w.w_fc1, D, row); bias = j == 0 ? w.b_fc1[row] : 0.f; }
          const float acc = rows_dot<LD, CD>(wr, act);
          if (j == 0) gput(G + O_HID + row, tag, gelu_erf(acc + bias));
        }
      }

      // ================================================================= mlp.2
      {
        const int r0 = (int)((long)wg * D / P), r1 = (int)((long)(wg + 1) * D / P);
        const int slot = tid / LF, j = tid % LF;
        u32x4 wr[CF];
        int row = r0 + slot;
        rows_load<LF, CF>(wr, w.w_fc2, F, row < r1 ? row : r0);
        float bias = (row < r1 && j == 0) ? w.b_fc2[row] : 0.f;
        unsigned y[GF];
        const bool fail = gather<GF>(G + O_HID, tag, y, p.err, [&](int k) { const int i = tid + k * PT; return i < F ? i : -1; });
#pragma unroll
        for (int k = 0; k < GF; ++k) { const int i = tid + k * PT; if (i < F) act[i] = __uint_as_float(y[k]); }
        if (__syncthreads_or(fail)) AXW_GIVE_UP(0x800 + l)
        for (; row < r1; row += SF) {
          if (row != r0 + slot) { rows_load<LF, CF>(wr, w.w_fc2, F, row); bias = j == 0 ? w.b_fc2[row] : 0.f; }
          const float acc = rows_dot<LF, CF>(wr, act);
          if (j == 0) gput(G + O_Y3 + row, tag, acc + bias);
        }
        __syncthreads();  // act is rewritten by the next phase's stage
      }
    }  // layers

    steps_run = step + 1;
    // ===================================================================== token feedback (Whisper.cpp:207-222)
    if (step < 3) {  // SOT steps: feed the next forced token, logits are discarded (Whisper.cpp:214-217)
      tok = p.sot[step + 1];
      continue;
    }
    // logits = token_embedding . ln(x)   (tied embedding, export_onnx.py:364-385) + argmax (first max wins, Whisper.cpp:42-45)
    int best_idx;
    {
      const int N = p.n_vocab;
      const int r0 = (int)((long)wg * N / P), r1 = (int)((long)(wg + 1) * N / P);
      const int slot = tid / LD, j = tid % LD;
      const unsigned tag = (unsigned)(step * L + L);  // y3 of the last layer
      u32x4 wn[CD];
      int row = r0 + slot;
      rows_load<LD, CD>(wn, p.tok_emb, D, row < r1 ? row : r0);
      unsigned y[GD];
      const bool fail = gather<GD>(G + O_Y3, tag, y, p.err, [&](int k) { const int i = tid + k * PT; return i < D ? i : -1; });
      if (ln_stage(y, true, p.ln_w, p.ln_b, fail)) AXW_GIVE_UP(0x900)
      float4 a[CD][2];
#pragma unroll
      for (int i = 0; i < CD; ++i) {
        a[i][0] = *reinterpret_cast<const float4*>(act + (j + LD * i) * 8);
        a[i][1] = *reinterpret_cast<const float4*>(act + (j + LD * i) * 8 + 4);
      }
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      float* dump = p.logits_dump ? p.logits_dump + (long)(step - 3) * N : nullptr;
      for (; row < r1; row += SD) {
        u32x4 wr[CD];
#pragma unroll
        for (int i = 0; i < CD; ++i) wr[i] = wn[i];
        const int nrow = row + SD;
        rows_load<LD, CD>(wn, p.tok_emb, D, nrow < r1 ? nrow : r0);  // next pass in flight
        const float acc = rows_dot_reg<LD, CD>(wr, a);
        if (j == 0) {
          if (dump) dump[row] = acc;
          if (acc > bv) { bv = acc; bi = row; }
        }
      }
      // workgroup argmax: lanes with j == 0 hold candidates; lower index wins ties
      if (j != 0) { bv = -INFINITY; bi = 0x7fffffff; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (lane == 0) { am_v[wave] = bv; am_i[wave] = bi; }
      __syncthreads();
      if (tid == 0) {
        for (int w2 = 1; w2 < PW; ++w2)
          if (am_v[w2] > bv || (am_v[w2] == bv && am_i[w2] < bi)) { bv = am_v[w2]; bi = am_i[w2]; }
        gput(G + O_AMAX + 2 * wg, (unsigned)(step + 1), bv);
        gput_u(G + O_AMAX + 2 * wg + 1, (unsigned)(step + 1), (unsigned)bi);
      }
      // every workgroup merges all partials itself
      unsigned v[1];
      const bool fail2 = gather<1>(G + O_AMAX, (unsigned)(step + 1), v, p.err, [&](int) { return tid < 2 * P ? tid : -1; });
      // lanes: even tid = value of workgroup tid/2, odd tid = index
      float cv = (tid < 2 * P && !(tid & 1)) ? __uint_as_float(v[0]) : -INFINITY;
      int ci = (int)__shfl_down(v[0], 1, 64);
      if (tid >= 2 * P || (tid & 1)) { cv = -INFINITY; ci = 0x7fffffff; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(cv, o, 64);
        const int oi = __shfl_xor(ci, o, 64);
        if (ov > cv || (ov == cv && oi < ci)) { cv = ov; ci = oi; }
      }
      __syncthreads();  // am_v/am_i of the partial stage are free again
      if (lane == 0) { am_v[wave] = cv; am_i[wave] = ci; }
      if (__syncthreads_or(fail2)) AXW_GIVE_UP(0xA00)
      cv = am_v[0]; ci = am_i[0];
      for (int w2 = 1; w2 < PW; ++w2)
        if (am_v[w2] > cv || (am_v[w2] == cv && am_i[w2] < ci)) { cv = am_v[w2]; ci = am_i[w2]; }
      best_idx = ci;
    }
    const int gi = step - 3;
    if (p.forced) {
      if (wg == 0 && tid == 0 && p.argmax_dump && gi <= p.n_forced) p.argmax_dump[gi] = best_idx;
      if (gi < p.n_forced) tok = p.forced[gi];
    } else {
      if (best_idx == p.eot || step + 1 >= p.n_ctx || n_out >= p.max_new) { n_done = 1; break; }
      if (wg == 0 && tid == 0) p.out_ids[n_out] = best_idx;
      ++n_out;
      tok = best_idx;
    }
  }

  if (wg == 0 && tid == 0) {
    p.n_out[0] = n_out;
    p.state->step = steps_run;
    p.state->n_done = n_done;
  }
#undef AXW_GIVE_UP
}

// ---------------------------------------------------------------------------------------- host side
bool decode_persistent_supported(int d_model, int n_head, int n_layer, int n_cu) {
  if (n_head * 64 != d_model) return false;
  const int P = n_cu < d_model ? n_cu : d_model;
  if (n_layer * n_head > P) return false;  // one (layer, head) self-attention cache per workgroup
  switch (d_model) { case 128: case 256: case 384: case 512: case 768: case 1280: return true; default: return false; }
}
int decode_persistent_grid(int d_model, int n_cu) { return n_cu < d_model ? n_cu : d_model; }
size_t decode_persistent_gran_bytes(int d_model, int grid) { return ((size_t)16 * d_model + 2 * (size_t)grid + 64) * 8; }

static size_t persist_lds_bytes(int d) { return (size_t)kSelfBlocks * 4096 * 2 * 2 + ((size_t)5 * d + d / 8) * 4 + (PW * kPS + 2 * PW + 64 + 128) * 4; }

template <int LD, int CD, int LF, int CF>
static hipError_t launch_one(const PersistParams& p, int grid, hipStream_t s) {
  const size_t lds = persist_lds_bytes(8 * LD * CD);
  auto kfn = decode_persistent_kernel<LD, CD, LF, CF>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(PT), lds, s, p);
  return hipGetLastError();
}

hipError_t launch_decode_persistent(const PersistParams& p, int d_model, int grid, hipStream_t s) {
  switch (d_model) {
    case 128: return launch_one<16, 1, 32, 2>(p, grid, s);
    case 256: return launch_one<32, 1, 64, 2>(p, grid, s);
    case 384: return launch_one<16, 3, 64, 3>(p, grid, s);
    case 512: return launch_one<32, 2, 64, 4>(p, grid, s);
    case 768: return launch_one<32, 3, 64, 6>(p, grid, s);
    case 1280: return launch_one<32, 5, 64, 10>(p, grid, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace axw
