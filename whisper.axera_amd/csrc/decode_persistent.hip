// decode_persistent.hip — the whole greedy decode loop of ONE clip as a single persistent launch (gfx950).
//
// Replaces, for batch 1, the per-token sequence Whisper::run_decoder (cpp/src/Whisper.cpp:290-346) + argmax (:42-45)
// + the host loop (:207-222) = the decoder graph TextDecoderTensorCache.forward (model_convert/export_onnx.py:312-387)
// with its attention (:103-150, :221-230). Same arithmetic as the launch-per-phase path (decode_gemv.hip,
// decoder.hip): h16 weights x fp32 activations with fp32 FMA, fp32 LayerNorm/softmax, h16 self/cross K/V.
//
// Why: at batch 1 a decoder step is a chain of ~100 dependent phases that each move only a few MB, so the chain is
// bound by per-phase latency, not by HBM. As separate graph nodes a phase costs ~4.4 us (kernel
// boundary + wave start + activation round trip + drain). Here one 1024-thread workgroup per CU stays resident for
// the whole utterance and the phases hand their outputs over INSIDE the launch:
//   * every output element travels as one 8-byte {tag, value} granule; a producer assembles its rows in LDS and ONE
//     wave stores them with ONE sc1 (write-through) instruction (a full 128-byte line), consumers poll pairs of
//     granules with 16-byte sc1 loads — the data is the flag, no fence, no separate barrier
//     (cdna_hip_programming.md §6 Guideline 16 form R2; tags = step * n_layer + layer + 1, never 0, buffers zeroed
//     before every launch);
//   * the waves of a workgroup have fixed ROLES: waves 0-7 ("pollers") gather granules, keep the residual stream in
//     registers and do LayerNorm; waves 8-15 ("compute") own the weight rows, the attention blocks and every
//     bulk load. Vector-memory results return to a wave in issue order, so a wave that polls must have no long load
//     in flight — with the roles split, a poll never queues behind a weight or K/V load, and the compute waves request
//     the rows of the NEXT phase right after publishing the current one, a whole hand-off ahead of their use;
//   * the self-attention K/V cache of one (layer, head) lives in the LDS of the workgroup that owns that head for
//     the whole utterance (448 keys x 64 x 2 x h16 = 112 KB): it never touches HBM. Workgroups that own no head
//     use the same LDS region to stage cross-attention K/V tiles by LDS-DMA one layer before their use;
//   * every workgroup keeps its own copy of the residual stream, so a LayerNorm needs no extra hand-off;
//   * the token feedback (argmax merge, SOT forcing, eot / context stop, embedding of the next token) is computed
//     redundantly by every workgroup from the gathered argmax partials: the loop never returns to the host.
// Every spin is bounded: a lane that waits too long raises a flag, its workgroup sets an error word and leaves; the
// others see the word (or time out themselves) and leave too, so the grid always drains. The host then falls back.
#include "decode_persistent_common.hpp"

namespace axw {
inline namespace AXW_NS {

// ---------------------------------------------------------------------------------------- query fold (round 5)
// A layer of the launch was eight dependent all-to-all hand-offs; three of them carried the self-attention output to the
// cross-attention units:   att -> [hop] -> out-projection rows -> y1 -> [hop] -> LayerNorm(x0 + y1) -> query rows -> [hop]
// The query is linear in everything but the LayerNorm statistics (export_onnx.py:221-230, 238-261):
//   cq_j = r (A0_j + (M a)_j + d_j - mu s_j) + c_j,    x1 = x0 + W_o a + b_o,  mu / r = mean / rstd of x1
//   A0 = W_cq (g . x0)   — needs only the layer's input: computed by the row producers WHILE self-attention runs
//   M  = W_cq diag(g) W_o (fp32, built once at load), d = W_cq (g . b_o), s = W_cq g, c = W_cq beta + b_cq
// so the 16-row producers of y1 also produce T = A0 + M a + d for the same 16 rows and the two sums of their x1 slice, and
// a cross-attention unit finishes its head's query from 64 T granules + 2 x (number of producers) statistics: the middle
// hop, the LayerNorm stage (two workgroup barriers in every workgroup) and the query-row phase are gone; y1 is added to
// the residual copies together with y2 (same order of additions: the stream is bit-identical to the unfolded launch).
// LayerNorm folded into the rows it feeds: W LN(x) + b = r (W (g . x) - mu s) + c with s = W g, c = W beta + b, so the pollers
// hand the compute waves g . x and the two sums behind ONE workgroup barrier instead of statistics, barrier, normalised
// vector, barrier; every compute wave derives mu and r from the eight partial sums itself.
__global__ void qfold_vec_kernel(const h16* __restrict__ wl, const float* __restrict__ fl, float* __restrict__ qf, int D) {
  const int l = blockIdx.y;
  const long DD = (long)D * D;
  const h16* WL = wl + (long)l * DecArena::w_stride(D);
  const float* F = fl + (long)l * DecArena::f_stride(D);
  float* out = qf + (long)l * qfold_stride(D) + DD;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < 3 * D) {
    const h16* w = WL + DecArena::W_QKV * DD + (long)j * D;
    const float *g = F + DecArena::F_ATTN_LN_W * D, *be = F + DecArena::F_ATTN_LN_B * D;
    double ss = 0.0, cc = 0.0;
    for (int i = 0; i < D; ++i) { const double wv = (double)(float)w[i]; ss += wv * (double)g[i]; cc += wv * (double)be[i]; }
    out[QF_SQKV * D + j] = (float)ss;
    out[QF_CQKV * D + j] = (float)(cc + (double)F[DecArena::F_B_QKV * D + j]);
  }
  if (j < 4 * D) {
    const h16* w = WL + DecArena::W_FC1 * DD + (long)j * D;
    const float *g = F + DecArena::F_MLP_LN_W * D, *be = F + DecArena::F_MLP_LN_B * D;
    double ss = 0.0, cc = 0.0;
    for (int i = 0; i < D; ++i) { const double wv = (double)(float)w[i]; ss += wv * (double)g[i]; cc += wv * (double)be[i]; }
    out[QF_SFC1 * D + j] = (float)ss;
    out[QF_CFC1 * D + j] = (float)(cc + (double)F[DecArena::F_B_FC1 * D + j]);
  }
}


// M = W_cq diag(g) W_o in double, one thread per element; d, s, c by the first D threads of block row 0
__global__ void qfold_build_kernel(const h16* __restrict__ wl, const float* __restrict__ fl, float* __restrict__ qf, int D) {
  const int l = blockIdx.z;
  const long DD = (long)D * D;
  const h16* wq = wl + (long)l * DecArena::w_stride(D) + DecArena::W_CQ * DD;
  const h16* wo = wl + (long)l * DecArena::w_stride(D) + DecArena::W_O * DD;
  const float* F = fl + (long)l * DecArena::f_stride(D);
  const float *g = F + DecArena::F_CROSS_LN_W * D, *be = F + DecArena::F_CROSS_LN_B * D, *bo = F + DecArena::F_B_O * D, *bq = F + DecArena::F_B_CQ * D;
  float* out = qf + (long)l * qfold_stride(D);
  const int k = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (k >= D) return;
  double acc = 0.0;
  for (int i = 0; i < D; ++i) acc += (double)(float)wq[(long)j * D + i] * (double)g[i] * (double)(float)wo[(long)i * D + k];
  out[(long)j * D + k] = (float)acc;
  if (k == 0) {
    double dd = 0.0, ss = 0.0, cc = 0.0;
    for (int i = 0; i < D; ++i) {
      const double w = (double)(float)wq[(long)j * D + i];
      dd += w * (double)g[i] * (double)bo[i];
      ss += w * (double)g[i];
      cc += w * (double)be[i];
    }
    out[DD + j] = (float)dd;
    out[DD + D + j] = (float)ss;
    out[DD + 2 * D + j] = (float)(cc + (double)bq[j]);
  }
}
void launch_qfold_build(const h16* wl, const float* fl, float* qf, int d_model, int n_layer, hipStream_t s) {
  qfold_build_kernel<<<dim3((d_model + 127) / 128, d_model, n_layer), 128, 0, s>>>(wl, fl, qf, d_model);
  qfold_vec_kernel<<<dim3((4 * d_model + 127) / 128, n_layer), 128, 0, s>>>(wl, fl, qf, d_model);
}
size_t qfold_floats(int d_model, int n_layer) { return (size_t)n_layer * (size_t)qfold_stride(d_model); }

// ---------------------------------------------------------------------------------------- the kernel
// d_model = 8*LD*CD (rows with K = d: LD lanes x CD 16-byte chunks), 4*d_model = 8*LF*CF.
// QF: the query fold above (d_model <= 768: the fp32 rows of M cost 2 * CD more registers per lane)
template <int LD, int CD, int LF, int CF, bool PROF, bool QF>
__global__ __launch_bounds__(PT) void decode_persistent_kernel(PersistParams p) {
  constexpr int D = 8 * LD * CD, F = 8 * LF * CF, H = D / 64;
  static_assert(F == 4 * D, "mlp width");
  // a poller lane owns PAIRS of adjacent vector elements: pair tid + j*PL (j < GPD) = elements 2*pair, 2*pair + 1
  constexpr int GPD = (D / 2 + PL - 1) / PL, GD = 2 * GPD, NPART = H * kCrossSplit * kPS;
  constexpr int NPP = NPART / 2, NPP1 = (NPP + 1) / 2, GP1 = (NPP1 + PL - 1) / PL, GP2 = (NPP - NPP1 + CT - 1) / CT;  // partial-record pairs, split between the roles
  static_assert(NPART % 2 == 0 && kPS % 2 == 0 && kRec % 2 == 0 && D % 2 == 0, "pair polls need even layouts");
  constexpr int NU = kCrossSplit * H;  // cross-attention units per layer
  // granule buffers (u64 units)
  constexpr int O_QKV = 0, O_ATT = 3 * D, O_Y1 = 4 * D, O_CQ = 5 * D, O_PART = 6 * D, O_Y2 = 10 * D, O_HID = 11 * D, O_Y3 = 15 * D,
                O_AMAX = 16 * D, O_STAT = 16 * D + 512;  // statistics of producer p: granules O_STAT + 16 p, + 1: a line of its own (packed, 8
                                                        // producers' partial-line stores per line: units gather 0.4 us later, 108.7 -> 111.1 ms)
  constexpr int XG = D, X0R = 2 * D, A0S = 3 * D;  // QF: act[XG..) = g_cross . x0, act[X0R..) = x0 (written with the QKV LayerNorm, read by
                                                  // the row producers), act[A0S + slot] = A0 of the slot's row (free until the partial records)
  static_assert(NPART <= 3 * D + D / 8 && NU * kRec <= 4 * D, "partial buffer");
  static_assert(kCrossSplit * NCW == 24, "cross-attention key blocks");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  h16* sK = reinterpret_cast<h16*>(smem);                  // [8 blk][8][64 keys][8]  (blocked, lane = key)
  h16* sV = sK + NCW * 4096;                                // cross tiles: [512 keys][64]; self-attention cache: per block [8 (key/8)][64 dims][8 keys]
  float* act = reinterpret_cast<float*>(smem + kKvBytes);    // [F + D/8] input vector of the current rows phase
  float* wpart = act + F + D / 8;                            // [NCW][kPS] per-wave attention partials
  float* red = wpart + NCW * kPS;                            // [2*NPW] LayerNorm partial sums, [2*NPW] = the stage's mean (QF: the statistics' shift)
  unsigned* qs = reinterpret_cast<unsigned*>(red + 2 * NPW + 4);  // [64] query of the attention phase as packed h16 pairs: [32] hi, [32] lo
  float* am_v = reinterpret_cast<float*>(qs) + 64;           // [16] argmax scratch
  int* am_i = reinterpret_cast<int*>(am_v + 16);             // [16]
  int* ctl = am_i + 16;                                      // [16]: 0 give-up flag, 1 argmax of the step
  float* pk = reinterpret_cast<float*>(ctl + 16);            // [64] this workgroup's rows of the phase, assembled for the one-instruction publish
  float* pscr = pk + 64;                                     // [NCW][64] probability transpose scratch
  long long* prof_acc = reinterpret_cast<long long*>(pscr + NCW * 64);  // [64] per-phase time sums + one layer's absolute timeline (profiling runs only)

  // tid is re-derived behind an opaque asm at the top of every layer: without it the compiler hoists every
  // per-thread address of every phase out of the step loop and keeps >100 registers of loop invariants alive
  int tid = threadIdx.x;
  const bool poller = tid < PL;  // wave-uniform
  const int P = gridDim.x, wg = blockIdx.x;
  const int L = p.n_layer;
  u64* const G = p.gran;
  const __amdgpu_buffer_rsrc_t GR = __builtin_amdgcn_make_buffer_rsrc((void*)p.gran, 0, p.gran_bytes, 0x27000);

  // self-attention ownership: unit (l, h) -> workgroup P-1-(l*H+h). The other NS workgroups take the cross-attention
  // units: unit u of layer l -> workgroup (l*NU + u) % NS.
  const int sa_unit = P - 1 - wg;
  const int sa_layer = sa_unit < L * H ? sa_unit / H : -1, sa_head = sa_unit % H;
  const int NS = P - L * H;
  // producers of the d-row phases (one pass of CT/LD resp. CT/LF rows each, workgroups 0..): only they consume the
  // attention outputs / cross-attention partials / mlp hidden vector; every other workgroup skips those three phases
  // altogether (no polls, no barriers): a hand-off is the faster the fewer workgroups poll it (-5 % decode time)
  constexpr int NP_D = (D + CT / LD - 1) / (CT / LD), NP_F2 = (D + CT / LF - 1) / (CT / LF);
  const int rwg = (wg - NS + P) % P;  // row roles by a rotated workgroup index: rwg 0 = the first self-attention owner
  const bool in_o = rwg < NP_D, in_f2 = rwg < NP_F2;

  if (p.fault && wg == 0) return;  // test hook: a workgroup that never publishes; everybody else must give up and drain
  for (int i = tid; i < kKvBytes / 16; i += PT) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};  // masked keys must be finite
  if (tid < 16) ctl[tid] = 0;
  if (PROF && tid < 64) prof_acc[tid] = 0;
  __syncthreads();

  long long t_last = PROF ? wall_clock64() : 0;
  // pollers stamp slots 0..15 (thread 0), compute waves 16..31 (thread PL)
#define AXW_TL(IDX) \
  if (PROF && tl_on && (tid == 0 || tid == PL)) prof_acc[32 + (IDX)] = wall_clock64();
#define AXW_STAMP(IDX) \
  if (PROF && (tid == 0 || tid == PL)) { const long long t_now = wall_clock64(); prof_acc[IDX] += t_now - t_last; t_last = t_now; }
  // first barrier of a phase: everybody learns whether a poller gave up
#define AXW_BARRIER_CHECK(CODE)                                                                                          \
  {                                                                                                                      \
    wg_barrier();                                                                                                        \
    if (ctl[0]) {                                                                                                        \
      if (tid == 0) __hip_atomic_store((gu32*)p.err, (unsigned)(CODE) | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
      return;                                                                                                            \
    }                                                                                                                    \
  }

  // Launch parameters that are read once per STEP or less (token feedback, teacher forcing, dumps, results) are not
  // kept in scalar registers for the whole launch: they are re-read from the kernel-argument segment at their use,
  // through a pointer the compiler cannot see through (so it can neither hoist the loads out of the step loop nor
  // keep their results live). The d_model-768 instantiation was spilling 185 scalar registers into vector lanes.
  const __attribute__((address_space(4))) PersistParams* kargs =
      (const __attribute__((address_space(4))) PersistParams*)__builtin_amdgcn_kernarg_segment_ptr();
#define AXW_COLD(FIELD) ([&] { auto* kp_ = kargs; asm volatile("" : "+s"(kp_)); return kp_->FIELD; }())
  int tok = AXW_COLD(sot)[0];
  int n_out = 0, n_done = 0, steps_run = 0;

  // Cross-attention unit of this workgroup in the t-th layer of the LAUNCH (t = step * L + l), or -1. The units of
  // consecutive layers take consecutive ranges of NU workgroups modulo NS, counted over the whole launch and not per
  // step: 2 * NU <= NS then keeps the two units of any workgroup at least two layers apart across the step boundary
  // as well. (Counted per step, the last layer's range wrapped onto the first layer's of the next step whenever
  // L * NU > NS — large-v3-turbo: 4 x 60 units on 176 workgroups — and a workgroup staged the next step's K tiles over
  // the ones its last-layer unit had not used yet: logits off by 4e-2 at every step of that model.)
  auto ca_unit_of = [&](int t) -> int {
    if (wg >= NS) return -1;
    int r = (wg - (int)(((long)t * NU) % NS)) % NS;
    if (r < 0) r += NS;
    return r < NU ? r : -1;
  };

  if (poller) {
    // ======================================================================================= pollers
    float x[GD];        // residual stream, element tid + k*PL
    float lg[GD], lb[GD];
    float g2[GD];       // QF, row producers: the cross-attention LayerNorm's gain of the NEXT layer to run (A0 = W_cq (g . x0)),
                        // requested a stage ahead like lg / lb: a polling wave must have no load in flight
    float shift = 0.f;  // LayerNorm variance shift (previous mean): sums stay small without a second pass
    auto el = [&](int k) { return 2 * (tid + (k >> 1) * PL) + (k & 1); };  // vector element of register slot k
    auto g2_prefetch = [&](int layer) {
#pragma unroll
      for (int k = 0; k < GD; ++k) { const int i = el(k); g2[k] = (QF && in_o && i < D) ? p.fl[(long)layer * DecArena::f_stride(D) + DecArena::F_CROSS_LN_W * D + i] : 0.f; }
    };
    g2_prefetch(0);
    auto ln_prefetch = [&](const float* g, const float* be) {
#pragma unroll
      for (int k = 0; k < GD; ++k) {
        const int i = el(k);
        lg[k] = i < D ? g[i] : 0.f;
        lb[k] = i < D ? be[i] : 0.f;
      }
    };
    // the pairs of a d-wide vector that starts at granule `base`
#define AXW_PAIRS_D(BASE) [&](int j) { const int pr = tid + j * PL; return 2 * pr < D ? (BASE) + 2 * pr : -1; }

    // x += y, LayerNorm into act[0..D): two workgroup barriers
    // QF: the stage of a LayerNorm that is folded into its rows (QKV, mlp.0): g . x and the two sums, ONE barrier. The caller has
    // added the gathered update to x. Writing act before the barrier is safe: the update that was just gathered (or, in the first
    // layer, the barrier at the end of the previous step) lies behind every read of act by this workgroup's compute waves.
#define AXW_RAW_STAGE(FAIL, CODE, XGW)                                                      \
  {                                                                                          \
    float s1 = 0.f, s2 = 0.f;                                                                \
    _Pragma("unroll") for (int k = 0; k < GD; ++k) {                                         \
      const int i = el(k);                                                                   \
      if (i < D) {                                                                           \
        const float t = x[k] - shift;                                                        \
        s1 += t; s2 += t * t;                                                                \
        act[i] = x[k] * lg[k];                                                               \
        if (XGW) { act[XG + i] = x[k] * g2[k]; act[X0R + i] = x[k]; }                        \
      }                                                                                      \
    }                                                                                        \
    s1 = wsum(s1); s2 = wsum(s2);                                                            \
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = s1; red[2 * (tid >> 6) + 1] = s2; }         \
    if (tid == 0) red[2 * NPW + 1] = shift;                                                  \
    if (FAIL) ctl[0] = 1;                                                                    \
    AXW_BARRIER_CHECK(CODE)                                                                  \
    float t1 = 0.f;                                                                          \
    _Pragma("unroll") for (int w2 = 0; w2 < NPW; ++w2) t1 += red[2 * w2];                    \
    shift += t1 / D;                                                                         \
    if (tid == 0) red[2 * NPW] = shift;                                                      \
  }
#define AXW_LN_STAGE(Y, ADD, FAIL, CODE) AXW_LN_STAGE_X(Y, ADD, FAIL, CODE, false)
    // XGW (QF, row producers, QKV stage): also leave g_cross . x and x itself in act[XG..), act[X0R..) — behind the stage's
    // first barrier, when the compute waves have left the previous phase (mlp.2 reads all of act)
#define AXW_LN_STAGE_X(Y, ADD, FAIL, CODE, XGW)                                             \
  {                                                                                          \
    float s1 = 0.f, s2 = 0.f;                                                                \
    _Pragma("unroll") for (int k = 0; k < GD; ++k) {                                         \
      if (el(k) < D) {                                                                       \
        if (ADD) x[k] += __uint_as_float(Y[k]);                                              \
        const float t = x[k] - shift;                                                        \
        s1 += t; s2 += t * t;                                                                \
      }                                                                                      \
    }                                                                                        \
    s1 = wsum(s1); s2 = wsum(s2);                                                            \
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = s1; red[2 * (tid >> 6) + 1] = s2; }         \
    if (FAIL) ctl[0] = 1;                                                                    \
    AXW_BARRIER_CHECK(CODE)                                                                  \
    float t1 = 0.f, t2 = 0.f;                                                                \
    _Pragma("unroll") for (int w2 = 0; w2 < NPW; ++w2) { t1 += red[2 * w2]; t2 += red[2 * w2 + 1]; } \
    const float dm = t1 / D, var = fmaxf(t2 / D - dm * dm, 0.f);                             \
    const float mean = shift + dm, rstd = rsqrtf(var + 1e-5f);                               \
    if (QF && tid == 0) red[2 * NPW] = mean;                                                 \
    _Pragma("unroll") for (int k = 0; k < GD; ++k) {                                         \
      const int i = el(k);                                                                   \
      if (i < D) act[i] = (x[k] - mean) * rstd * lg[k] + lb[k];                              \
      if (QF && (XGW) && i < D) { act[XG + i] = x[k] * g2[k]; act[X0R + i] = x[k]; }         \
    }                                                                                        \
    shift = mean;                                                                            \
    wg_barrier();                                                                            \
  }

    for (int step = 0; step < p.total_steps; ++step) {
      // x = token_embedding[tok] + positional_embedding[step]   (export_onnx.py:334-336)
#pragma unroll
      for (int k = 0; k < GD; ++k) {
        const int i = el(k);
        x[k] = i < D ? (float)AXW_COLD(tok_emb)[(long)tok * D + i] + AXW_COLD(pos)[(long)step * D + i] : 0.f;
      }
      ln_prefetch(p.fl + DecArena::F_ATTN_LN_W * D, p.fl + DecArena::F_ATTN_LN_B * D);

      for (int l = 0; l < L; ++l) {
        asm volatile("" : "+v"(tid));
        const bool tl_on = step == p.total_steps / 2 && l == L / 2;
        const float* FL = p.fl + (long)l * DecArena::f_stride(D);
        const unsigned tag = (unsigned)(step * L + l + 1);
        // ---- QKV
        {
          unsigned y[GD];
          bool fail = false;
          if (l > 0) fail = gather2<GPD>(GR, tag - 1, y, p.err, ctl, AXW_PAIRS_D(O_Y3));
          AXW_STAMP(0)
          AXW_TL(0)
          if constexpr (QF) {
            if (l > 0) {
#pragma unroll
              for (int k = 0; k < GD; ++k) if (el(k) < D) x[k] += __uint_as_float(y[k]);
            }
            AXW_RAW_STAGE(fail, 0x100 + l, in_o)
            ln_prefetch(FL + DecArena::F_MLP_LN_W * D, FL + DecArena::F_MLP_LN_B * D);
          } else {
            AXW_LN_STAGE_X(y, l > 0, fail, 0x100 + l, in_o)
            ln_prefetch(FL + DecArena::F_CROSS_LN_W * D, FL + DecArena::F_CROSS_LN_B * D);
          }
          AXW_STAMP(1)
          AXW_TL(1)
        }
        // ---- self-attention owner: collect q, k, v of the head; append k, v to the LDS cache
        if (l == sa_layer) {
          unsigned v[2];  // lanes 0-31: q, 32-63: k, 64-95: v of the head, two adjacent dims each
          const bool fail = gather2<1>(GR, tag, v, p.err, ctl, [&](int) { return tid < 96 ? O_QKV + (tid >> 5) * D + sa_head * 64 + 2 * (tid & 31) : -1; });
          if (tid < 32) {  // the query: dims 2 tid, 2 tid + 1 as one packed (hi, lo) pair
            unsigned hi, lo;
            h16split2(__uint_as_float(v[0]), __uint_as_float(v[1]), hi, lo);
            qs[tid] = hi;
            qs[32 + tid] = lo;
          } else if (tid < 96) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int dd = 2 * (tid & 31) + e;
              const float val = __uint_as_float(v[e]);
              if (tid < 64)  // K row `step`, blocked [blk][d/8][key%64][8]
                sK[(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (h16)val;
              else           // V row `step`, TRANSPOSED per block: [blk][key%64 / 8][dim][8 keys]
                sV[(step >> 6) * 4096 + ((step >> 3) & 7) * 512 + dd * 8 + (step & 7)] = (h16)val;
            }
          }
          if (fail) ctl[0] = 1;
          AXW_STAMP(2)
          AXW_BARRIER_CHECK(0x200 + l)
          AXW_STAMP(3)
        }
        // ---- attention output projection
        if (in_o) {
          unsigned y[GD];
          const bool fail = gather2<GPD>(GR, tag, y, p.err, ctl, AXW_PAIRS_D(O_ATT));
#pragma unroll
          for (int k = 0; k < GD; ++k) { const int i = el(k); if (i < D) act[i] = __uint_as_float(y[k]); }
          if (fail) ctl[0] = 1;
          AXW_STAMP(4)
          AXW_TL(2)
          AXW_BARRIER_CHECK(0x300 + l)
        }
        // ---- cross-attention query
        if constexpr (!QF) {
          unsigned y[GD];
          const bool fail = gather2<GPD>(GR, tag, y, p.err, ctl, AXW_PAIRS_D(O_Y1));
          AXW_STAMP(5)
          AXW_TL(3)
          AXW_LN_STAGE(y, true, fail, 0x400 + l)
          ln_prefetch(FL + DecArena::F_MLP_LN_W * D, FL + DecArena::F_MLP_LN_B * D);
          AXW_STAMP(6)
          AXW_TL(4)
        }
        // ---- cross-attention unit: collect the head's query
        const int cu = ca_unit_of(step * L + l);
        if (cu >= 0) {
          const int ca_head = cu / kCrossSplit;
          if constexpr (QF) {
            const bool fail = qfold_unit_query<D, NP_D>(GR, tag, tid, O_CQ, O_STAT, AXW_COLD(qf) + (long)l * qfold_stride(D) + (long)D * D,
                                                        ca_head, shift, qs, p.err, ctl);
            if (fail) ctl[0] = 1;
            AXW_TL(3)
          } else {
            unsigned v[2];
            const bool fail = gather2<1>(GR, tag, v, p.err, ctl, [&](int) { return tid < 32 ? O_CQ + ca_head * 64 + 2 * tid : -1; });
            if (tid < 32) {
              unsigned hi, lo;
              h16split2(__uint_as_float(v[0]), __uint_as_float(v[1]), hi, lo);
              qs[tid] = hi;
              qs[32 + tid] = lo;
            }
            if (fail) ctl[0] = 1;
          }
          AXW_STAMP(7)
          AXW_BARRIER_CHECK(0x500 + l)
          AXW_STAMP(8)
        }
        // ---- cross-attention output projection: merge the partials of every head
        if (in_o) {
          // the pollers collect the first NP1 granules of the partial records, the (idle) compute waves the rest
          unsigned y[2 * GP1];  // pair pi of the records: record pi / 33, granules 2 * (pi % 33), +1
          const bool fail = gather2<GP1>(GR, tag, y, p.err, ctl, [&](int j) { const int pi = tid + j * PL; return pi < NPP1 ? O_PART + (pi / (kPS / 2)) * kRec + 2 * (pi % (kPS / 2)) : -1; });
          float* pbuf = act + D;  // [H][kCrossSplit][66]
#pragma unroll
          for (int j = 0; j < GP1; ++j) {
            const int pi = tid + j * PL;
            if (pi < NPP1) { pbuf[2 * pi] = __uint_as_float(y[2 * j]); pbuf[2 * pi + 1] = __uint_as_float(y[2 * j + 1]); }
          }
          if (fail) ctl[0] = 1;
          AXW_STAMP(9)
          AXW_TL(5)
          AXW_BARRIER_CHECK(0x600 + l)
#pragma unroll
          for (int k = 0; k < GD; ++k) {
            const int i = tid + k * PL;
            if (i < D) {
              act[i] = merge_cross_records(pbuf, i);
            }
          }
          wg_barrier();
          AXW_STAMP(10)
          AXW_TL(6)
        }
        // ---- mlp.0
        if constexpr (QF) {  // x += y1, then += y2: the unfolded launch's order of additions
          unsigned y[2 * GD];
          const bool fail = gather2<2 * GPD>(GR, tag, y, p.err, ctl, [&](int j) { const int pr = tid + (j % GPD) * PL; return 2 * pr < D ? (j < GPD ? O_Y1 : O_Y2) + 2 * pr : -1; });
#pragma unroll
          for (int k = 0; k < GD; ++k) if (el(k) < D) x[k] = (x[k] + __uint_as_float(y[k])) + __uint_as_float(y[GD + k]);
          AXW_STAMP(11)
          AXW_TL(7)
          AXW_RAW_STAGE(fail, 0x700 + l, false)
          if (l + 1 < L) ln_prefetch(FL + DecArena::f_stride(D) + DecArena::F_ATTN_LN_W * D, FL + DecArena::f_stride(D) + DecArena::F_ATTN_LN_B * D);
          else ln_prefetch(AXW_COLD(ln_w), AXW_COLD(ln_b));
          g2_prefetch(l + 1 < L ? l + 1 : 0);
          AXW_STAMP(12)
          AXW_TL(8)
        } else {
          unsigned y[GD];
          const bool fail = gather2<GPD>(GR, tag, y, p.err, ctl, AXW_PAIRS_D(O_Y2));
          AXW_STAMP(11)
          AXW_TL(7)
          AXW_LN_STAGE(y, true, fail, 0x700 + l)
          if (l + 1 < L) ln_prefetch(FL + DecArena::f_stride(D) + DecArena::F_ATTN_LN_W * D, FL + DecArena::f_stride(D) + DecArena::F_ATTN_LN_B * D);
          else ln_prefetch(AXW_COLD(ln_w), AXW_COLD(ln_b));
          AXW_STAMP(12)
          AXW_TL(8)
        }
        // ---- mlp.2: the 4d-wide hidden vector is the largest hand-off; the pollers collect its first half, the compute
        //      waves (idle until it is complete, their rows already in registers) the second half
        if (in_f2) {
          constexpr int GH = (F / 4 + PL - 1) / PL;  // pairs per lane of the first half
          unsigned y[2 * GH];
          const bool fail = gather2<GH>(GR, tag, y, p.err, ctl, [&](int j) { const int pi = tid + j * PL; return pi < F / 4 ? O_HID + 2 * pi : -1; });
#pragma unroll
          for (int j = 0; j < GH; ++j) {
            const int pi = tid + j * PL;
            if (pi < F / 4) { act[2 * pi] = __uint_as_float(y[2 * j]); act[2 * pi + 1] = __uint_as_float(y[2 * j + 1]); }
          }
          if (fail) ctl[0] = 1;
          AXW_STAMP(13)
          AXW_TL(9)
          AXW_BARRIER_CHECK(0x800 + l)
        }
      }  // layers

      steps_run = step + 1;
      asm volatile("" : "+v"(tid));
      if (step < 3) {  // SOT steps: feed the next forced token, logits are discarded (Whisper.cpp:214-217)
        tok = AXW_COLD(sot)[step + 1];
        if constexpr (QF) wg_barrier();  // the next step's first stage writes act before its barrier: nobody may still be reading it
        continue;
      }
      // ---- final LayerNorm for the vocabulary projection, then merge the argmax partials of every workgroup
      {
        unsigned y[GD];
        const bool fail = gather2<GPD>(GR, (unsigned)(step * L + L), y, p.err, ctl, AXW_PAIRS_D(O_Y3));
        AXW_LN_STAGE(y, true, fail, 0x900)
        AXW_STAMP(14)
        wg_barrier();  // B3: the compute waves have their workgroup argmax
        unsigned v[2];  // {value, row index} of workgroup tid: one pair
        const bool fail2 = gather2<1>(GR, (unsigned)(step + 1), v, p.err, ctl, [&](int) { return tid < P ? O_AMAX + 2 * tid : -1; });
        // first max wins (Whisper.cpp:42-45)
        float cv = tid < P ? __uint_as_float(v[0]) : -INFINITY;
        int ci = tid < P ? (int)v[1] : 0x7fffffff;
        wave_argmax(cv, ci);  // DPP + row swaps: the __shfl_xor form is twelve dependent ds_bpermute round trips on the token's path
        if ((tid & 63) == 0) { am_v[tid >> 6] = cv; am_i[tid >> 6] = ci; }
        if (fail2) ctl[0] = 1;
        AXW_STAMP(15)
        AXW_BARRIER_CHECK(0xA00)  // B4
      }
      float cv = am_v[0];
      int best_idx = am_i[0];
      for (int w2 = 1; w2 < NPW; ++w2)
        if (am_v[w2] > cv || (am_v[w2] == cv && am_i[w2] < best_idx)) { cv = am_v[w2]; best_idx = am_i[w2]; }
      // no logit compared greater than -inf (all NaN / -inf: non-finite audio): std::max_element returns index 0
      // (Whisper.cpp:42-45); never let the "no candidate" index reach the embedding lookup
      if ((unsigned)best_idx >= (unsigned)AXW_COLD(n_vocab)) best_idx = 0;
      wg_barrier();  // B5: am_v/am_i are free again
      const int gi = step - 3;
      if (AXW_COLD(forced)) {
        if (wg == 0 && tid == 0 && AXW_COLD(argmax_dump) && gi <= AXW_COLD(n_forced)) AXW_COLD(argmax_dump)[gi] = best_idx;
        if (gi < AXW_COLD(n_forced)) tok = AXW_COLD(forced)[gi];
      } else {
        if (best_idx == AXW_COLD(eot) || step + 1 >= AXW_COLD(n_ctx) || n_out >= AXW_COLD(max_new)) { n_done = 1; break; }
        if (wg == 0 && tid == 0) AXW_COLD(out_ids)[n_out] = best_idx;
        ++n_out;
        tok = best_idx;
      }
    }
#undef AXW_LN_STAGE
#undef AXW_LN_STAGE_X
#undef AXW_RAW_STAGE
#undef AXW_PAIRS_D
  } else {
    // ======================================================================================= compute waves
    // The pollers spin; without a priority the arbiter gives their loops the same share of the issue slots as the
    // waves that do the work (measured: decode 178 -> 158 ms for Whisper-small with this one instruction).
    __builtin_amdgcn_s_setprio(3);
    int ctid = tid - PL;
    // Which workgroups produce the rows of a phase. A hand-off is faster the fewer workgroups publish into it
    // (12 producers: 0.9 us, 256: 2-2.7 us), so a layer's rows go to as few producers as one full pass each allows
    // (16 rows for K = d). Every workgroup must still publish in at least one all-to-all phase of every layer — that
    // is what bounds how far any consumer can lag behind a producer that reuses a buffer one layer later — so the
    // 4d-wide layer (first workgroups) and the 3d-wide one (last workgroups) are packed only if together they cover
    // the grid; otherwise the 4d-wide layer keeps the even deal over all workgroups.
    constexpr int SLD = CT / LD, NP_Q = (3 * D + SLD - 1) / SLD, NP_F = (F + SLD - 1) / SLD;
    const int pk_qkv = NP_Q <= P ? P - NP_Q : -1;
    // The 4d-wide layer's producers start BEHIND the NP_D workgroups that produce the three d-wide layers (those are
    // the busiest: their mlp.0 rows could only be requested after their cross-attention-output publish, ~2 us before
    // use, and arrived late — every consumer of the hidden vector waited for them: 1.2 us of skew per layer).
    const int pk_f = (NP_D + NP_F <= P && NP_Q <= P && NP_D + NP_F + NP_Q >= P) ? NP_D : ((NP_F <= P && NP_Q <= P && NP_F + NP_Q >= P) ? 0 : -1);
    const bool is_fc1 = pk_f < 0 || (rwg >= pk_f && rwg < pk_f + NP_F);
    // its mlp.2 rows can be requested a phase earlier (no mlp.0 rows in the way); not for wide models: 10 chunks per lane
    // held across the mlp.0 phase do not fit the register budget (the d=1280 instantiation went to scratch)
    constexpr bool kEarlyFc2 = CF <= 6;
    const bool early_fc2 = kEarlyFc2 && in_f2 && !is_fc1;
    const int pk_d = 0;  // d rows in passes of CT/LD (or CT/LF) rows: never more producers than workgroups (P <= d)
    // two register sets for the d-wide layers are enough: a phase computes from one while the next phase's rows land
    // in the other (qkv A, o B, cq A, co B, mlp.0 A, [mlp.2 F], next qkv / vocabulary A)
    RowSet<LD, CD> ra, rb;
    RowSet<LF, CF> rs_fc2;
    // QKV rows of layer `ly`: with their LayerNorm folded in (QF) or with their plain bias
    auto qkv_prefetch = [&](int ly) {
      const h16* w = p.wl + (long)ly * DecArena::w_stride(D);
      if constexpr (QF) {
        const float* v = AXW_COLD(qf) + (long)ly * qfold_stride(D) + (long)D * D;
        ra.prefetch_ln(w, v + QF_SQKV * D, v + QF_CQKV * D, D, 3 * D, rwg, P, ctid, pk_qkv);
      } else {
        ra.prefetch(w, p.fl + (long)ly * DecArena::f_stride(D) + DecArena::F_B_QKV * D, D, 3 * D, rwg, P, ctid, pk_qkv);
      }
    };
    // mean and rstd of the stage the pollers have just closed: the eight partial sums + the shift they are relative to
    auto stage_stats = [&](float& mean, float& rstd) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NPW; ++w2) { t1 += red[2 * w2]; t2 += red[2 * w2 + 1]; }
      const float dm = t1 / D, var = fmaxf(t2 / D - dm * dm, 0.f);
      mean = red[2 * NPW + 1] + dm;
      rstd = rsqrtf(var + 1e-5f);
    };
    qkv_prefetch(0);
    {  // the first layer's cross-attention unit has no previous layer to hide behind
      const int cu0 = ca_unit_of(0);
      if (cu0 >= 0) {
        const int lane = ctid & 63, cw = __builtin_amdgcn_readfirstlane(ctid >> 6);
        const long off = (long)(cu0 / kCrossSplit) * 24 * 4096 + (long)((cu0 % kCrossSplit) * NCW + cw) * 4096;
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_global_load_lds((gptr_t)(p.cross_k + off + i * 512 + lane * 8), (lds_ptr_t)(sK + cw * 4096 + i * 512), 16, 0, kKvAux);
          __builtin_amdgcn_global_load_lds((gptr_t)(p.cross_v + off + i * 512 + lane * 8), (lds_ptr_t)(sV + cw * 4096 + i * 512), 16, 0, kKvAux);
        }
      }
    }

    for (int step = 0; step < p.total_steps; ++step) {
      for (int l = 0; l < L; ++l) {
        asm volatile("" : "+v"(ctid));
        const bool tl_on = step == p.total_steps / 2 && l == L / 2;
        const int lane = ctid & 63, cw = __builtin_amdgcn_readfirstlane(ctid >> 6);
        constexpr long DD = (long)D * D;
        const h16* WL = p.wl + (long)l * DecArena::w_stride(D);
        const float* FL = p.fl + (long)l * DecArena::f_stride(D);
        const h16 *w_qkv = WL + DecArena::W_QKV * DD, *w_o = WL + DecArena::W_O * DD, *w_cq = WL + DecArena::W_CQ * DD,
                   *w_co = WL + DecArena::W_CO * DD, *w_fc1 = WL + DecArena::W_FC1 * DD, *w_fc2 = WL + DecArena::W_FC2 * DD;
        const float *b_qkv = FL + DecArena::F_B_QKV * D, *b_o = FL + DecArena::F_B_O * D, *b_cq = FL + DecArena::F_B_CQ * D,
                    *b_co = FL + DecArena::F_B_CO * D, *b_fc1 = FL + DecArena::F_B_FC1 * D, *b_fc2 = FL + DecArena::F_B_FC2 * D;
        const unsigned tag = (unsigned)(step * L + l + 1);
        const int cu = ca_unit_of(step * L + l);
        // Cross K/V tiles are constant during the utterance: the unit this workgroup runs in the NEXT layer is staged
        // into LDS now (LDS-DMA, 16 x 1 KiB per wave), a few instructions after each publish of this layer, so that no
        // publish waits behind a burst of DMA requests. Units of one workgroup are at least two layers apart (ca_unit_of).
        const int ln = l + 1 < L ? l + 1 : 0;
        const int cun = ca_unit_of(step * L + l + 1);
        auto kv_piece = [&](int i0, int i1) {
          if (cun < 0) return;
          const int kb = (cun % kCrossSplit) * NCW + cw;  // 64-key block of this wave (24 blocks = t_pad 1536)
          const long off = (long)ln * p.cross_layer_stride + (long)(cun / kCrossSplit) * 24 * 4096 + (long)kb * 4096;
          for (int i = i0; i < i1; ++i) {
            const h16* src = (i < 8 ? p.cross_k : p.cross_v) + off + (i & 7) * 512 + lane * 8;
            h16* dst = (i < 8 ? sK : sV) + cw * 4096 + (i & 7) * 512;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)dst, 16, 0, kKvAux);
          }
        };
        // ---- QKV rows (export_onnx.py:245-247)
        AXW_BARRIER_CHECK(0x100 + l)
        float res[2];
        if constexpr (QF) {
          AXW_STAMP(16)
          AXW_TL(18)
          float mean, rstd;
          stage_stats(mean, rstd);
          const float* v = AXW_COLD(qf) + (long)l * qfold_stride(D) + (long)D * D;
          ra.run_ln(w_qkv, v + QF_SQKV * D, v + QF_CQKV * D, D, act, ctid, res, mean, rstd);
        } else {
          wg_barrier();
          AXW_STAMP(16)
          AXW_TL(18)
          ra.run(w_qkv, b_qkv, D, act, ctid, res);
        }
        ra.publish(ctid, res, pk, ctl + 2, G + O_QKV, tag, [](float v) { return v; });
        if constexpr (QF) {
          // row producers: A0 = W_cq (g . x0) of this slot's row NOW — the self-attention owners' q, k, v are still travelling
          // (x0 has been in LDS since the QKV stage) — then W_o's rows and the rows of M into the registers W_cq's rows leave.
          // W_cq's rows are requested first: results return in issue order, and an owner's publish of its attention vector
          // waits for every load issued before it
          if (in_o) {
            ra.prefetch(w_cq, nullptr, D, D, rwg, P, ctid, pk_d);
            float ra0[2];
            ra.run(w_cq, nullptr, D, act + XG, ctid, ra0);
            if (ctid % LD == 0) act[A0S + ctid / LD] = ra0[0];  // parked in LDS: the attention blocks of an owner need the registers
          }
        }
        rb.prefetch(w_o, b_o, D, D, rwg, P, ctid, pk_d);  // (unconditional: a set assigned on one path only stays live around the step loop)
        kv_piece(0, 2);
        AXW_STAMP(17)
        AXW_TL(10)
        // ---- self-attention of one head over keys 0..step (export_onnx.py:103-147: the -60000 mask + the separate
        //      current-token column of the reference equal causal attention)
        if (l == sa_layer) {
          AXW_BARRIER_CHECK(0x200 + l)
          const int nblk = (step >> 6) + 1;
          if (cw < nblk)
            attn_block<true>(sK + cw * 4096, sV + cw * 4096, qs, cw * 64 + lane <= step, pscr + cw * 64, wpart + cw * kPS, lane);
          // no second workgroup barrier: the compute wave that arrives last merges the block partials and publishes
          __builtin_amdgcn_wave_barrier();
          int old = 0;
          if (lane == 0) old = __hip_atomic_fetch_add(ctl + 3, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
          old = __builtin_amdgcn_readfirstlane(old);
          if ((old + 1) % NCW == 0) {
            float m, lt, ov;
            merge_partials(wpart, nblk, lane, &m, &lt, &ov);
            gput(G + O_ATT + sa_head * 64 + lane, tag, ov / lt);
          }
          AXW_STAMP(18)
          AXW_TL(11)
        }
        // ---- attention output projection
        if constexpr (QF) {
          if (in_o) {
            // the rows of M: behind the self-attention section (24 registers that the attention blocks need), and behind an
            // owner's publish of its attention vector (a store waits for every load issued before it)
            const float* qfl = AXW_COLD(qf) + (long)l * qfold_stride(D);
            const int slot = ctid / LD, row = rb.r0 + slot < rb.r1 ? rb.r0 + slot : rb.r0;
            RowSetF32<LD, CD> rm;  // this producer's rows of M = W_cq diag(g) W_o (scoped: a set that lived across the step loop was spilled)
            rm.prefetch(qfl, qfl + (long)D * D, D, row, rb.r0 + slot < rb.r1, ctid);
            AXW_BARRIER_CHECK(0x300 + l)
            AXW_STAMP(19)
            AXW_TL(19)
            // y1 rows as always, T = A0 + M a + d for the same rows, and the two sums of this slice of x1 = x0 + y1:
            // three lines, one store instruction each, all by the compute wave that arrives last
            rb.run(w_o, b_o, D, act, ctid, res);
            const float tq = rm.run(act, ctid);
            const int j = ctid % LD, nrows = rb.r1 - rb.r0;
            if (j == 0 && slot < nrows) {
              pk[slot] = res[0];
              pk[32 + slot] = tq + act[A0S + slot];
              pscr[slot] = (act[X0R + rb.r0 + slot] + res[0]) - red[2 * NPW];
            }
            qfold_publish(lane, pk, pscr, ctl + 2, G, O_CQ, O_STAT, O_Y1, rb.r0, nrows, rwg, tag);
            AXW_TL(13)
          }
          rb.prefetch(w_co, b_co, D, D, rwg, P, ctid, pk_d);
          kv_piece(2, 8);
          AXW_STAMP(20)
          AXW_TL(12)
        } else {
          if (in_o) {
            AXW_BARRIER_CHECK(0x300 + l)
            AXW_STAMP(19)
            AXW_TL(19)
            rb.run(w_o, b_o, D, act, ctid, res);
            rb.publish(ctid, res, pk, ctl + 2, G + O_Y1, tag, [](float v) { return v; });
          }
          ra.prefetch(w_cq, b_cq, D, D, rwg, P, ctid, pk_d);
          kv_piece(2, 5);
          AXW_STAMP(20)
          AXW_TL(12)
          // ---- cross-attention query (export_onnx.py:221-230)
          AXW_BARRIER_CHECK(0x400 + l)
          wg_barrier();
          AXW_STAMP(21)
          AXW_TL(20)
          ra.run(w_cq, b_cq, D, act, ctid, res);
          ra.publish(ctid, res, pk, ctl + 2, G + O_CQ, tag, [](float v) { return v; });
          rb.prefetch(w_co, b_co, D, D, rwg, P, ctid, pk_d);
          kv_piece(5, 8);
          AXW_STAMP(22)
          AXW_TL(13)
        }
        // ---- cross-attention over one third of the 1536 padded keys
        if (cu >= 0) {
          const int ca_head = cu / kCrossSplit, ca_split = cu % kCrossSplit;
          AXW_BARRIER_CHECK(0x500 + l)
          cross_unit_block(sK, sV, qs, pscr, wpart, ctl + 3, G + O_PART + (ca_head * kCrossSplit + ca_split) * kRec, tag, ca_split, p.n_audio_ctx, cw, lane);
          AXW_STAMP(23)
          AXW_TL(14)
        }
        // ---- cross-attention output projection
        if (in_o) {
          {
            unsigned y[2 * GP2];
            const bool fail = gather2<GP2>(GR, tag, y, p.err, ctl, [&](int j) { const int pi = NPP1 + ctid + j * CT; return pi < NPP ? O_PART + (pi / (kPS / 2)) * kRec + 2 * (pi % (kPS / 2)) : -1; });
            float* pbuf = act + D;
#pragma unroll
            for (int j = 0; j < GP2; ++j) {
              const int pi = NPP1 + ctid + j * CT;
              if (pi < NPP) { pbuf[2 * pi] = __uint_as_float(y[2 * j]); pbuf[2 * pi + 1] = __uint_as_float(y[2 * j + 1]); }
            }
            if (fail) ctl[0] = 1;
          }
          AXW_BARRIER_CHECK(0x600 + l)
          wg_barrier();
          AXW_STAMP(24)
          AXW_TL(21)
          rb.run(w_co, b_co, D, act, ctid, res);
          rb.publish(ctid, res, pk, ctl + 2, G + O_Y2, tag, [](float v) { return v; });
        }
        if constexpr (QF) {
          const float* v = AXW_COLD(qf) + (long)l * qfold_stride(D) + (long)D * D;
          ra.prefetch_ln(w_fc1, v + QF_SFC1 * D, v + QF_CFC1 * D, D, F, rwg, P, ctid, pk_f);
        } else {
          ra.prefetch(w_fc1, b_fc1, D, F, rwg, P, ctid, pk_f);
        }
        if constexpr (kEarlyFc2) {
          if (early_fc2) rs_fc2.prefetch(w_fc2, b_fc2, F, D, rwg, P, ctid, pk_d);
        }
        kv_piece(8, 11);
        AXW_STAMP(25)
        AXW_TL(15)
        // ---- mlp.0 + GELU (export_onnx.py:298)
        AXW_BARRIER_CHECK(0x700 + l)
        if constexpr (QF) {
          AXW_STAMP(26)
          AXW_TL(22)
          float mean, rstd;
          stage_stats(mean, rstd);
          const float* v = AXW_COLD(qf) + (long)l * qfold_stride(D) + (long)D * D;
          ra.run_ln(w_fc1, v + QF_SFC1 * D, v + QF_CFC1 * D, D, act, ctid, res, mean, rstd);
        } else {
          wg_barrier();
          AXW_STAMP(26)
          AXW_TL(22)
          ra.run(w_fc1, b_fc1, D, act, ctid, res);
        }
        ra.publish(ctid, res, pk, ctl + 2, G + O_HID, tag, [](float v) { return gelu_erf(v); });
        if (!early_fc2) rs_fc2.prefetch(w_fc2, b_fc2, F, D, rwg, P, ctid, pk_d);
        kv_piece(11, 13);
        AXW_STAMP(27)
        AXW_TL(16)
        // ---- mlp.2
        if (in_f2) {
          {
            constexpr int GH = (F / 4 + CT - 1) / CT;
            unsigned y[2 * GH];
            const bool fail = gather2<GH>(GR, tag, y, p.err, ctl, [&](int j) { const int pi = ctid + j * CT; return pi < F / 4 ? O_HID + F / 2 + 2 * pi : -1; });
#pragma unroll
            for (int j = 0; j < GH; ++j) {
              const int pi = ctid + j * CT;
              if (pi < F / 4) { act[F / 2 + 2 * pi] = __uint_as_float(y[2 * j]); act[F / 2 + 2 * pi + 1] = __uint_as_float(y[2 * j + 1]); }
            }
            if (fail) ctl[0] = 1;
          }
          AXW_BARRIER_CHECK(0x800 + l)
          AXW_STAMP(28)
          AXW_TL(23)
          rs_fc2.run(w_fc2, b_fc2, F, act, ctid, res);
          rs_fc2.publish(ctid, res, pk, ctl + 2, G + O_Y3, tag, [](float v) { return v; });
          AXW_STAMP(31)
        }
        // next consumer of the residual stream: the next layer's QKV rows, the vocabulary projection, or the next step
        if (l + 1 < L) qkv_prefetch(l + 1);
        else if (step >= 3) ra.prefetch(AXW_COLD(tok_emb), nullptr, D, AXW_COLD(n_vocab), rwg, P, ctid);
        else qkv_prefetch(0);
        kv_piece(13, 16);
        AXW_STAMP(29)
        AXW_TL(17)
      }  // layers

      steps_run = step + 1;
      asm volatile("" : "+v"(ctid));
      if (step < 3) {
        tok = AXW_COLD(sot)[step + 1];
        if constexpr (QF) wg_barrier();
        continue;
      }
      // ---- logits = token_embedding . ln(x)  (tied embedding, export_onnx.py:364-385) + argmax (first max wins, Whisper.cpp:42-45)
      {
        constexpr int SD = CT / LD;
        const int lane = ctid & 63, cw = __builtin_amdgcn_readfirstlane(ctid >> 6);
        const int N = AXW_COLD(n_vocab);
        const int slot = ctid / LD, j = ctid % LD;
        AXW_BARRIER_CHECK(0x900)
        wg_barrier();
        // activations of this lane's chunks stay in registers over all passes (wide models: re-read from LDS per pass,
        // the registers are needed for the rows in flight)
        constexpr bool kActInRegs = CD <= 3;
        float4 a[kActInRegs ? CD : 1][2];
        if constexpr (kActInRegs) {
#pragma unroll
          for (int i = 0; i < CD; ++i) {
            a[i][0] = *reinterpret_cast<const float4*>(act + (j + LD * i) * 8);
            a[i][1] = *reinterpret_cast<const float4*>(act + (j + LD * i) * 8 + 4);
          }
        }
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        float* dump = AXW_COLD(logits_dump) ? AXW_COLD(logits_dump) + (long)(step - 3) * N : nullptr;
        const int r0 = ra.r0, r1 = ra.r1;
        auto consume = [&](const u32x4 (&wr)[CD], int row) {
          float acc;
          if constexpr (kActInRegs) acc = rows_dot_reg<LD, CD>(wr, a);
          else acc = rows_dot<LD, CD>(wr, act, ctid);
          if (j == 0) {
            if (dump) dump[row] = acc;
            if (acc > bv) { bv = acc; bi = row; }
          }
        };
        if constexpr (CD <= 3) {  // two passes ahead: ra.w holds pass 0, wn pass 1
          u32x4 wn[CD];
          {
            const int nrow = r0 + slot + SD;
            rows_load<LD, CD, kVocabNT>(wn, AXW_COLD(tok_emb), D, nrow < r1 ? nrow : r0, ctid);
          }
          for (int row = r0 + slot; row < r1; row += SD) {
            u32x4 wr[CD];
#pragma unroll
            for (int i = 0; i < CD; ++i) { wr[i] = ra.w[i]; ra.w[i] = wn[i]; }
            const int nrow = row + 2 * SD;
            rows_load<LD, CD, kVocabNT>(wn, AXW_COLD(tok_emb), D, nrow < r1 ? nrow : r0, ctid);
            consume(wr, row);
          }
        } else {  // wide rows: one pass ahead (register budget)
          for (int row = r0 + slot; row < r1; row += SD) {
            u32x4 wr[CD];
#pragma unroll
            for (int i = 0; i < CD; ++i) wr[i] = ra.w[i];
            const int nrow = row + SD;
            rows_load<LD, CD, kVocabNT>(ra.w, AXW_COLD(tok_emb), D, nrow < r1 ? nrow : r0, ctid);
            consume(wr, row);
          }
        }
        // the next step's first rows: requested before the token is even known
        qkv_prefetch(0);
        // workgroup argmax: lanes with j == 0 hold candidates; the lower index wins ties
        if (j != 0) { bv = -INFINITY; bi = 0x7fffffff; }
        wave_argmax(bv, bi);
        if (lane == 0) { am_v[8 + cw] = bv; am_i[8 + cw] = bi; }
        // compute waves only: named exchange through LDS, then wave 0 publishes. The pollers sit at B3 meanwhile.
        wg_barrier();  // B3 (all waves)
        if (ctid == 0) {
          for (int w2 = 1; w2 < NCW; ++w2)
            if (am_v[8 + w2] > bv || (am_v[8 + w2] == bv && am_i[8 + w2] < bi)) { bv = am_v[8 + w2]; bi = am_i[8 + w2]; }
          am_v[8] = bv; am_i[8] = bi;
        }
        __builtin_amdgcn_wave_barrier();
        if (ctid < 2)  // value and index of this workgroup's best row in ONE store instruction
          gput_u(G + O_AMAX + 2 * wg + ctid, (unsigned)(step + 1), ctid == 0 ? __float_as_uint(am_v[8]) : (unsigned)am_i[8]);
        AXW_STAMP(30)
        AXW_BARRIER_CHECK(0xA00)  // B4
      }
      float cv = am_v[0];
      int best_idx = am_i[0];
      for (int w2 = 1; w2 < NPW; ++w2)
        if (am_v[w2] > cv || (am_v[w2] == cv && am_i[w2] < best_idx)) { cv = am_v[w2]; best_idx = am_i[w2]; }
      if ((unsigned)best_idx >= (unsigned)AXW_COLD(n_vocab)) best_idx = 0;  // as in the pollers' copy of this merge
      wg_barrier();  // B5
      const int gi = step - 3;
      if (AXW_COLD(forced)) {
        if (gi < AXW_COLD(n_forced)) tok = AXW_COLD(forced)[gi];
      } else {
        if (best_idx == AXW_COLD(eot) || step + 1 >= AXW_COLD(n_ctx) || n_out >= AXW_COLD(max_new)) { n_done = 1; break; }
        ++n_out;
        tok = best_idx;
      }
    }
  }

  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no LDS-DMA may still be in flight when the workgroup's LDS is released
  if (PROF) {
    __syncthreads();
    if (tid < 64) AXW_COLD(prof)[(long)wg * 64 + tid] = prof_acc[tid];
  }
  if (wg == 0 && tid == 0) {
    AXW_COLD(n_out)[0] = n_out;
    AXW_COLD(state)->step = steps_run;
    AXW_COLD(state)->n_done = n_done;
  }
#undef AXW_COLD
#undef AXW_BARRIER_CHECK
#undef AXW_STAMP
#undef AXW_TL
}

// ---------------------------------------------------------------------------------------- host side
int decode_persistent_grid(int d_model, int n_cu) {
  int g = n_cu < d_model ? n_cu : d_model;  // every workgroup owns at least one row of the narrowest layer
  return g < 256 ? g : 256;                 // the argmax merge reads 2 granules per workgroup with 512 poller lanes
}
bool decode_persistent_supported(int d_model, int n_head, int n_layer, int n_cu) {
  if (n_head * 64 != d_model) return false;
  const int P = decode_persistent_grid(d_model, n_cu);
  // one (layer, head) self-attention cache per workgroup; the rest take the 3 * n_head cross-attention units of a layer
  if (P - n_layer * n_head < 2 * kCrossSplit * n_head) return false;  // and units of one workgroup >= 2 layers apart
  switch (d_model) { case 128: case 256: case 384: case 512: case 768: case 1280: return true; default: return false; }
}
// 16 d-wide buffers, the argmax pairs at 16 d (up to 512 granules), the fold's statistics at 16 d + 512 (one 16-granule line
// per row producer, at most d / 16 of them), the error word last
size_t decode_persistent_gran_bytes(int d_model, int grid) { return ((size_t)16 * d_model + 512 + (size_t)d_model + 64 + 0 * (size_t)grid) * 8; }

static size_t persist_lds_bytes(int d) {
  return (size_t)kKvBytes + ((size_t)4 * d + d / 8 + NCW * kPS + 2 * NPW + 4 + 64 + 16 + 16 + 16 + 64 + NCW * 64) * 4 + 64 * 8 + 64;
}

template <int LD, int CD, int LF, int CF, bool PROF, bool QF>
static hipError_t launch_one_prof(const PersistParams& p, int grid, hipStream_t s) {
  const size_t lds = persist_lds_bytes(8 * LD * CD);
  auto kfn = decode_persistent_kernel<LD, CD, LF, CF, PROF, QF>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(PT), lds, s, p);
  return hipGetLastError();
}

// the profiling stamps are a separate instantiation: the production kernel carries none of their code
template <int LD, int CD, int LF, int CF>
static hipError_t launch_one(const PersistParams& p, int grid, hipStream_t s) {
  if constexpr (8 * LD * CD <= 768) {  // the query fold, when the engine built its arena (p.qf)
    if (p.qf) return p.prof ? launch_one_prof<LD, CD, LF, CF, true, true>(p, grid, s) : launch_one_prof<LD, CD, LF, CF, false, true>(p, grid, s);
  }
  return p.prof ? launch_one_prof<LD, CD, LF, CF, true, false>(p, grid, s) : launch_one_prof<LD, CD, LF, CF, false, false>(p, grid, s);
}

hipError_t launch_decode_persistent(const PersistParams& p, int d_model, int grid, hipStream_t s) {
  if (p.n_clip >= 2) return launch_decode_persistent2(p, d_model, grid, s);  // two or three clips: decode_persistent2.hip
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

}  // inline namespace AXW_NS
}  // namespace axw
