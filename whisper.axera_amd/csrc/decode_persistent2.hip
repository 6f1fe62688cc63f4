// decode_persistent2.hip — the greedy decode loops of TWO clips as a single persistent launch (gfx950).
//
// The one-clip launch (decode_persistent.hip) spends two thirds of a decoder layer waiting for hand-offs: eight dependent
// all-to-all exchanges of 1.1-1.7 us each against ~9 us of arithmetic (profiles/r04_persist_phases_summary.txt). Here every
// phase runs for clip 0 and then for clip 1 with the SAME weight rows in registers: while clip 0's outputs travel to their
// consumers, clip 1's rows are computed, and the other way round. Two clips per call then cost one launch of ~1.3x the
// one-clip launch instead of two launches (the reference's loop, Whisper.cpp:207-222, stays per clip: each clip stops at
// its own eot / budget and rides along, ignored, until the other one has stopped too).
// What does NOT fit twice is LDS: a workgroup's 128 KB of K/V tiles (the self-attention cache of its (layer, head) or its
// cross-attention unit's 512 keys) belong to clip 0 exactly as in the one-clip launch; clip 1's self-attention cache lives in
// global memory (PersistParams::self_k1 / self_v1, the per-owner layouts of the LDS cache, read past L1) and its cross K/V
// are read from their slot; the d-wide input vector, the query and the argmax scratch exist once per clip (3.4 KB), the two
// wide LDS vectors (mlp hidden, cross-attention partial records) are shared and handed over with one more barrier.
// The kernel body is the one-clip kernel's, phase by phase, with a clip loop around every phase (NC = 2); it is a separate
// translation unit so that the one-clip kernel — the headline path — keeps its register allocation to the instruction.
#include "decode_persistent_common.hpp"

namespace axw {
inline namespace AXW_NS {

// ---------------------------------------------------------------------------------------- the kernel
// d_model = 8*LD*CD (rows with K = d: LD lanes x CD 16-byte chunks), 4*d_model = 8*LF*CF.
// NC = clips per launch. 2: every phase runs for clip 0 and then for clip 1 (PersistParams::n_clip): the weight rows of a
// phase are in registers once and serve both clips, and while one clip's outputs travel to their consumers (1.1-1.7 us per
// hand-off, two thirds of a one-clip layer) the other clip's rows are computed. LDS holds one clip's K/V tiles and no more
// (128 KB of 160), so clip 1's self-attention cache and cross K/V stay in global memory and its attention blocks read
// them from there; the d-wide input vectors and the attention query exist once per clip (3.3 KB more LDS), the two wide
// LDS vectors (mlp hidden, cross-attention partial records) are shared and handed over with one more barrier.
// QF: the cross-attention query folded through the output projection, as in the one-clip launch (decode_persistent.hip, round 5):
// the row producers of y1 also publish T = A0 + M a + d and the sums of their slice of x1 = x0 + y1, a unit finishes its head's
// query from them; the per-clip query LayerNorm stage and query-row phase are gone. Per clip: g . x0 [D] in the wide LDS vector
// (act + D .., free between the QKV stage and the partial records), and 68 words for A0, the x0 slice and the statistics' shift.
template <int LD, int CD, int LF, int CF, bool PROF, int NC, bool QF>
__global__ __launch_bounds__(PT) void decode_persistent_kernel(PersistParams p) {
  static_assert(NC == 2 || NC == 3, "two or three clips per launch");
  constexpr int D = 8 * LD * CD, F = 8 * LF * CF, H = D / 64;
  static_assert(F == 4 * D, "mlp width");
  // a poller lane owns PAIRS of adjacent vector elements: pair tid + j*PL (j < GPD) = elements 2*pair, 2*pair + 1
  constexpr int GPD = (D / 2 + PL - 1) / PL, GD = 2 * GPD, NPART = H * kCrossSplit * kPS;
  constexpr int NPP = NPART / 2, NPP1 = (NPP + 1) / 2, GP1 = (NPP1 + PL - 1) / PL, GP2 = (NPP - NPP1 + CT - 1) / CT;  // partial-record pairs, split between the roles
  static_assert(NPART % 2 == 0 && kPS % 2 == 0 && kRec % 2 == 0 && D % 2 == 0, "pair polls need even layouts");
  constexpr int NU = kCrossSplit * H;  // cross-attention units per layer and clip
  constexpr int NUC = NC * NU;         // ... per layer: unit r = clip r / NU, (head, key range) r % NU
  // granule buffers (u64 units)
  constexpr int O_QKV = 0, O_ATT = 3 * D, O_Y1 = 4 * D, O_CQ = 5 * D, O_PART = 6 * D, O_Y2 = 10 * D, O_HID = 11 * D, O_Y3 = 15 * D,
                O_AMAX = 16 * D, O_STAT = 16 * D + 512;  // (QF) statistics of row producer p: granules O_STAT + 16 p, + 1
  static_assert(NPART <= 3 * D + D / 8 && NU * kRec <= 4 * D, "partial buffer");
  static_assert(kCrossSplit * NCW == 24, "cross-attention key blocks");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  h16* sK = reinterpret_cast<h16*>(smem);                  // [8 blk][8][64 keys][8]  (blocked, lane = key)
  h16* sV = sK + NCW * 4096;                                // cross tiles: [512 keys][64]; self-attention cache: per block [8 (key/8)][64 dims][8 keys]
  float* act = reinterpret_cast<float*>(smem + kKvBytes);    // [F + D/8] input vector of the current rows phase
  float* wpart = act + F + D / 8;                            // [NCW][kPS] per-wave attention partials
  float* red = wpart + NCW * kPS;                            // [2*NPW] LayerNorm partial sums
  unsigned* qs = reinterpret_cast<unsigned*>(red + 2 * NPW);  // [64] query of the attention phase as packed h16 pairs: [32] hi, [32] lo
  float* am_v = reinterpret_cast<float*>(qs) + 64;           // [16] argmax scratch
  int* am_i = reinterpret_cast<int*>(am_v + 16);             // [16]
  int* ctl = am_i + 16;                                      // [16]: 0 give-up flag, 1 argmax of the step
  float* pk = reinterpret_cast<float*>(ctl + 16);            // [64] this workgroup's rows of the phase, assembled for the one-instruction publish
  float* pscr = pk + 64;                                     // [NCW][64] probability transpose scratch
  long long* prof_acc = reinterpret_cast<long long*>(pscr + NCW * 64);  // [64] per-phase time sums + one layer's absolute timeline (profiling runs only)
  // every clip after the first: its own d-wide input vector [D], attention query [64], argmax scratch [32] and the self-attention
  // k, v rows of its CURRENT step [64 words = 128 h16] (persist2_lds_bytes)
  constexpr int XW = D + 64 + 32 + 64;
  float* const xbase = reinterpret_cast<float*>(prof_acc + 64);
  float* actc[NC];
  unsigned* qsc[NC];
  float* am_vc[NC];
  int* am_ic[NC];
  h16* kvtc[NC];
  actc[0] = act; qsc[0] = qs; am_vc[0] = am_v; am_ic[0] = am_i; kvtc[0] = nullptr;
#pragma unroll
  for (int c = 1; c < NC; ++c) {
    actc[c] = xbase + (c - 1) * XW;
    qsc[c] = reinterpret_cast<unsigned*>(actc[c] + D);
    am_vc[c] = reinterpret_cast<float*>(qsc[c] + 64);
    am_ic[c] = reinterpret_cast<int*>(qsc[c] + 64) + 16;
    kvtc[c] = reinterpret_cast<h16*>(qsc[c] + 64 + 32);
  }
  float* wpart1 = xbase + (NC - 1) * XW;  // [NPW][kPS] + [NPW][64]: the poller waves' own attention scratch (they run the later clips'
  float* pscr1 = wpart1 + NPW * kPS;      //  self-attention blocks while the compute waves run clip 0's)
  float* const qfs = pscr1 + NPW * 64;    // QF: per clip [32] A0 of this producer's rows, [32] its slice of x0, [4] the statistics' shift
  auto xgc = [&](int c) { return act + D + c * D; };   // QF: g_cross . x0 of clip c
  auto a0s = [&](int c) { return qfs + c * 68; };
  auto x0s = [&](int c) { return qfs + c * 68 + 32; };
  auto shs = [&](int c) { return qfs + c * 68 + 64; };

  // tid is re-derived behind an opaque asm at the top of every layer: without it the compiler hoists every
  // per-thread address of every phase out of the step loop and keeps >100 registers of loop invariants alive
  int tid = threadIdx.x;
  const bool poller = tid < PL;  // wave-uniform
  const int P = gridDim.x, wg = blockIdx.x;
  const int L = p.n_layer;
  u64* const G = p.gran;
  const __amdgpu_buffer_rsrc_t GR = __builtin_amdgcn_make_buffer_rsrc((void*)p.gran, 0, (NC - 1) * (int)(p.gran_clip_u64 * 8) + p.gran_bytes, 0x27000);
  const int gco = (int)p.gran_clip_u64;  // granule index of clip c's area: c * gco

  // self-attention ownership: unit (l, h) -> workgroup P-1-(l*H+h). The other NS workgroups take the cross-attention
  // units: unit u of layer l -> workgroup (l*NU + u) % NS.
  const int sa_unit = P - 1 - wg;
  const int sa_layer = sa_unit < L * H ? sa_unit / H : -1, sa_head = sa_unit % H;
  const int NS = P - L * H;
  // producers of the d-row phases (one pass of CT/LD resp. CT/LF rows each, workgroups 0..): only they consume the
  // attention outputs / cross-attention partials / mlp hidden vector; every other workgroup skips those three phases
  // altogether (no polls, no barriers): a hand-off is the faster the fewer workgroups poll it (-5 % decode time)
  constexpr int NP_D = (D + CT / LD - 1) / (CT / LD), NP_F2 = (D + CT / LF - 1) / (CT / LF);
  // Row roles go by a ROTATED workgroup index (rwg 0 = the first self-attention owner): the d-wide layers' producers are then
  // workgroups that own a head (busy with attention in one layer of twelve), not the ones that run a cross-attention unit in
  // most layers — with two clips interleaved, a producer that is also a unit holder puts one clip's rows behind the other
  // clip's attention block.
  const int rwg = (wg - NS + P) % P;
  const bool in_o = rwg < NP_D, in_f2 = rwg < NP_F2;
  // (Measured and switched off: a head's owner that produces no QKV rows running clip 0's self-attention block BEFORE clip 1's
  //  QKV LayerNorm — its query is on its way already — 134.0 -> 143.8 ms per pair: the owners' later phases slip behind.)
  constexpr bool sa_first = false;

  if (p.fault && wg == 0) return;  // test hook: a workgroup that never publishes; everybody else must give up and drain
  for (int i = tid; i < kKvBytes / 16; i += PT) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};  // masked keys must be finite
  if (tid < 16) ctl[tid] = 0;
  if (PROF && tid < 64) prof_acc[tid] = 0;
  __syncthreads();

  long long t_last = PROF ? wall_clock64() : 0;
  // pollers stamp slots 0..15 (thread 0), compute waves 16..31 (thread PL)
#define AXW_TL(IDX) \
  if (PROF && tl_on && (tid == 0 || tid == PL) && (p.prof_clip || prof_acc[32 + (IDX)] == 0)) prof_acc[32 + (IDX)] = wall_clock64();
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
  int tok[NC], n_out[NC];
  bool fin[NC];  // this clip has emitted eot / spent its budget: it rides along (its results are ignored) until the other one has, too
#pragma unroll
  for (int c = 0; c < NC; ++c) { tok[c] = AXW_COLD(sot)[0]; n_out[c] = 0; fin[c] = false; }
  int n_done = 0, steps_run = 0;

  // Cross-attention unit of this workgroup in the t-th layer of the LAUNCH (t = step * L + l), or -1. The units of
  // consecutive layers take consecutive ranges of NU workgroups modulo NS, counted over the whole launch and not per
  // step: 2 * NU <= NS then keeps the two units of any workgroup at least two layers apart across the step boundary
  // as well. (Counted per step, the last layer's range wrapped onto the first layer's of the next step whenever
  // L * NU > NS — large-v3-turbo: 4 x 60 units on 176 workgroups — and a workgroup staged the next step's K tiles over
  // the ones its last-layer unit had not used yet: logits off by 4e-2 at every step of that model.)
  // Two clips: NC * NU units per layer, each ONE clip's (head, key range) — a workgroup's 128 KB of tiles are one clip's, as
  // in the one-clip launch, and twice as many workgroups are busy per layer. NC * NU <= NS only guarantees one unit per
  // workgroup and layer, so a workgroup may own units in CONSECUTIVE layers: the tiles of the next unit are requested
  // only after this layer's attention block is through (kv_piece below).
  auto ca_unit_of = [&](int t) -> int {
    if (wg >= NS) return -1;
    int r = (wg - (int)(((long)t * NUC) % NS)) % NS;
    if (r < 0) r += NS;
    return r < NUC ? r : -1;
  };

  // per-clip argmax scratch (clip 1's sits behind its query)

  if (poller) {
    // ======================================================================================= pollers
    float x[NC][GD];    // residual stream of every clip, element tid + k*PL
    float lg[GD], lb[GD];
    float shift[NC];    // LayerNorm variance shift (previous mean): sums stay small without a second pass
#pragma unroll
    for (int c = 0; c < NC; ++c) shift[c] = 0.f;
    auto el = [&](int k) { return 2 * (tid + (k >> 1) * PL) + (k & 1); };  // vector element of register slot k
    float g2[GD];       // QF, row producers: the cross-attention LayerNorm's gain of the next layer to run (requested a stage ahead)
    auto g2_prefetch = [&](int layer) {
#pragma unroll
      for (int k = 0; k < GD; ++k) { const int i = el(k); g2[k] = (QF && in_o && i < D) ? p.fl[(long)layer * DecArena::f_stride(D) + DecArena::F_CROSS_LN_W * D + i] : 0.f; }
    };
    g2_prefetch(0);
    const int row0 = rwg * (CT / LD);  // QF: first row of this producer's slice of the d-wide layers
    auto ln_prefetch = [&](const float* g, const float* be) {
#pragma unroll
      for (int k = 0; k < GD; ++k) {
        const int i = el(k);
        lg[k] = i < D ? g[i] : 0.f;
        lb[k] = i < D ? be[i] : 0.f;
      }
    };
    // the pairs of a d-wide vector of clip C that starts at granule `base` of its area
#define AXW_PAIRS_D(BASE, C) [&](int j) { const int pr = tid + j * PL; return 2 * pr < D ? (C) * gco + (BASE) + 2 * pr : -1; }

    // x[C] += y, LayerNorm into clip C's input vector: two workgroup barriers
#define AXW_LN_STAGE(Y, ADD, FAIL, CODE, C) AXW_LN_STAGE_X(Y, ADD, FAIL, CODE, C, false)
    // XGW (QF, row producers, QKV stage): also leave g_cross . x, this producer's slice of x and the mean for the compute waves
#define AXW_LN_STAGE_X(Y, ADD, FAIL, CODE, C, XGW)                                          \
  {                                                                                          \
    float s1 = 0.f, s2 = 0.f;                                                                \
    _Pragma("unroll") for (int k = 0; k < GD; ++k) {                                         \
      if (el(k) < D) {                                                                       \
        if (ADD) x[C][k] += __uint_as_float(Y[k]);                                           \
        const float t = x[C][k] - shift[C];                                                  \
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
    const float mean = shift[C] + dm, rstd = rsqrtf(var + 1e-5f);                            \
    _Pragma("unroll") for (int k = 0; k < GD; ++k) {                                         \
      const int i = el(k);                                                                   \
      if (i < D) actc[C][i] = (x[C][k] - mean) * rstd * lg[k] + lb[k];                       \
      if (QF && (XGW) && i < D) {                                                            \
        xgc(C)[i] = x[C][k] * g2[k];                                                         \
        if (i >= row0 && i < row0 + CT / LD) x0s(C)[i - row0] = x[C][k];                     \
      }                                                                                      \
    }                                                                                        \
    if (QF && (XGW) && tid == 0) shs(C)[0] = mean;                                           \
    shift[C] = mean;                                                                         \
    wg_barrier();                                                                            \
  }

    for (int step = 0; step < p.total_steps; ++step) {
      // x = token_embedding[tok] + positional_embedding[step]   (export_onnx.py:334-336)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < GD; ++k) {
          const int i = el(k);
          x[c][k] = i < D ? (float)AXW_COLD(tok_emb)[(long)tok[c] * D + i] + AXW_COLD(pos)[(long)step * D + i] : 0.f;
        }
      ln_prefetch(p.fl + DecArena::F_ATTN_LN_W * D, p.fl + DecArena::F_ATTN_LN_B * D);

      for (int l = 0; l < L; ++l) {
        asm volatile("" : "+v"(tid));
        const bool tl_on = step == p.total_steps / 2 && l == L / 2;
        const float* FL = p.fl + (long)l * DecArena::f_stride(D);
        const unsigned tag = (unsigned)(step * L + l + 1);
        // lanes 0-31: q, 32-63: k, 64-95: v of the head, two adjacent dims each
        auto qkv_pair = [&](int c) { return [&, c](int) { return tid < 96 ? c * gco + O_QKV + (tid >> 5) * D + sa_head * 64 + 2 * (tid & 31) : -1; }; };
        auto stage_q = [&](const unsigned (&v)[2], unsigned* q) {  // dims 2 tid, 2 tid + 1 as one packed (hi, lo) pair
          unsigned hi, lo;
          h16split2(__uint_as_float(v[0]), __uint_as_float(v[1]), hi, lo);
          q[tid] = hi;
          q[32 + tid] = lo;
        };
        // self-attention owner, clip 0: collect q, k, v of the head; append k, v to the LDS cache; its blocks are the compute waves'
#define AXW_SA0_POLL                                                                                                   \
  {                                                                                                                    \
    unsigned v[2];                                                                                                     \
    const bool fail = gather2<1>(GR, tag, v, p.err, ctl, qkv_pair(0));                                                 \
    if (tid < 32) stage_q(v, qs);                                                                                      \
    else if (tid < 96) {                                                                                               \
      _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                                  \
        const int dd = 2 * (tid & 31) + e;                                                                             \
        const float val = __uint_as_float(v[e]);                                                                       \
        if (tid < 64) /* K row `step`, blocked [blk][d/8][key%64][8] */                                                \
          sK[(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (h16)val;                            \
        else          /* V row `step`, TRANSPOSED per block: [blk][key%64 / 8][dim][8 keys] */                         \
          sV[(step >> 6) * 4096 + ((step >> 3) & 7) * 512 + dd * 8 + (step & 7)] = (h16)val;                           \
      }                                                                                                                \
    }                                                                                                                  \
    if (fail) ctl[0] = 1;                                                                                              \
    AXW_STAMP(2)                                                                                                       \
    AXW_BARRIER_CHECK(0x200 + l)                                                                                       \
    AXW_STAMP(3)                                                                                                       \
  }
        // ---- QKV
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          unsigned y[GD];
          bool fail = false;
          if (l > 0) fail = gather2<GPD>(GR, tag - 1, y, p.err, ctl, AXW_PAIRS_D(O_Y3, c));
          AXW_STAMP(0)
          AXW_TL(0)
          AXW_LN_STAGE_X(y, l > 0, fail, 0x100 + l, c, in_o)
          AXW_STAMP(1)
          AXW_TL(1)
          if (c == 0 && l == sa_layer && sa_first) AXW_SA0_POLL
        }
        if constexpr (QF) ln_prefetch(FL + DecArena::F_MLP_LN_W * D, FL + DecArena::F_MLP_LN_B * D);
        else ln_prefetch(FL + DecArena::F_CROSS_LN_W * D, FL + DecArena::F_CROSS_LN_B * D);
        // ---- self-attention owner (clip 0: LDS cache, clip 1: global memory)
        if (l == sa_layer) {
          if (!sa_first) AXW_SA0_POLL
#pragma unroll
          for (int c = 1; c < NC; ++c) {
            // The later clips' caches live in global memory (there is one LDS region, and it is clip 0's). Their blocks are run by the POLLER
            // waves — eight of them, with registers to spare, idle while the compute waves run clip 0's blocks: each requests its
            // block (the rows of the EARLIER steps; nothing of this step is needed from memory) before clip 1's query has been
            // polled for, holds it in registers, and puts this step's row (LDS) into its place. Straight-line code with selects:
            // 64 registers of block behind a branch were 64 registers of copies through scratch.
            const int nblk = (step >> 6) + 1;
            const int lane = tid & 63, pw = __builtin_amdgcn_readfirstlane(tid >> 6);
            u32x4 kr[8], vr[8];
            {
              const bool on = pw < nblk;  // a block without a single key of this clip yet: all zeros, every key masked
              const h16* kc = p.self_k1 + (c - 1) * p.self_clip_stride + (long)sa_unit * (NCW * 4096) + (on ? pw : 0) * 4096;
              const h16* vc = p.self_v1 + (c - 1) * p.self_clip_stride + (long)sa_unit * (NCW * 4096) + (on ? pw : 0) * 4096;
              const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kc, 0, on ? 8192 : 0, 0x27000);  // (0 bytes: loads return 0, no traffic)
              const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vc, 0, on ? 8192 : 0, 0x27000);
#pragma unroll
              for (int i = 0; i < 8; ++i) kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (i * 512 + lane * 8) * 2, 0, 1);  // sc0: past L1
#pragma unroll
              for (int i = 0; i < 8; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, (i * 512 + lane * 8) * 2, 0, 1);
            }
            unsigned v[2];
            const bool fail = gather2<1>(GR, tag, v, p.err, ctl, qkv_pair(c));
            if (tid < 32) stage_q(v, qsc[c]);
            else if (tid < 96) {
              h16* kd = p.self_k1 + (c - 1) * p.self_clip_stride + (long)sa_unit * (NCW * 4096);
              h16* vd = p.self_v1 + (c - 1) * p.self_clip_stride + (long)sa_unit * (NCW * 4096);
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const int dd = 2 * (tid & 31) + e;
                const float val = __uint_as_float(v[e]);
                // the global stores are for the LATER steps (nobody waits for them); this step's row travels through LDS.
                // Why the readers of the next step see them without a fence: gfx9 counts a wave's loads AND stores in one
                // in-order counter, so by the time this wave has consumed the result of ANY later poll (s_waitcnt vmcnt on a
                // load issued behind these stores: the very next gather) the stores have been acknowledged by L2; the reading
                // waves are on this CU, at least one workgroup barrier behind that point (a whole decoder step, in fact), and
                // read with sc0 = past the L1 that may still hold the line from an earlier step
                if (tid < 64) kd[(step >> 6) * 4096 + (dd >> 3) * 512 + (step & 63) * 8 + (dd & 7)] = (h16)val;
                else vd[(step >> 6) * 4096 + ((step >> 3) & 7) * 512 + dd * 8 + (step & 7)] = (h16)val;
                kvtc[c][(tid < 64 ? 0 : 64) + dd] = (h16)val;
              }
            }
            if (fail) ctl[0] = 1;
            AXW_BARRIER_CHECK(0x200 + l)
            {
              const bool mine = pw == (step >> 6);  // this step's key is in this wave's block: key step % 64
              const int i_s = (step >> 3) & 7, w_s = (step & 7) >> 1;
              const unsigned nv = reinterpret_cast<const unsigned short*>(kvtc[c])[64 + lane];
              const bool klane = mine && lane == (step & 63);
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const u32x4 kn = *reinterpret_cast<const u32x4*>(kvtc[c] + i * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  kr[i][e] = klane ? kn[e] : kr[i][e];
                  const unsigned pv = (step & 1) ? ((vr[i][e] & 0xffffu) | (nv << 16)) : ((vr[i][e] & 0xffff0000u) | nv);
                  vr[i][e] = (mine && i == i_s && e == w_s) ? pv : vr[i][e];
                }
              }
            }
            attn_block_regs(kr, vr, qsc[c], pw * 64 + lane <= step, pscr1 + pw * 64, wpart1 + pw * kPS, lane);
            // the poller wave that arrives last merges the block partials and publishes
            __builtin_amdgcn_wave_barrier();
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(ctl + 4, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = __builtin_amdgcn_readfirstlane(old);
            if ((old + 1) % NPW == 0) {
              float m, lt, ov;
              merge_partials(wpart1, nblk, lane, &m, &lt, &ov);
              gput(G + c * gco + O_ATT + sa_head * 64 + lane, tag, ov / lt);
            }
          }
        }
        // ---- attention output projection
        if (in_o) {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            unsigned y[GD];
            const bool fail = gather2<GPD>(GR, tag, y, p.err, ctl, AXW_PAIRS_D(O_ATT, c));
#pragma unroll
            for (int k = 0; k < GD; ++k) { const int i = el(k); if (i < D) actc[c][i] = __uint_as_float(y[k]); }
            if (fail) ctl[0] = 1;
            AXW_STAMP(4)
            AXW_TL(2)
            AXW_BARRIER_CHECK(0x300 + l)
          }
        }
        const int cu = ca_unit_of(step * L + l);
        if constexpr (QF) {
          // ---- cross-attention unit: T of the head + the producers' statistics of ITS clip -> the head's query (decode_persistent.hip)
          if (cu >= 0) {
            const int ca_clip = cu / NU, ca_head = (cu % NU) / kCrossSplit;
            float shc = shift[0];
#pragma unroll
            for (int c = 1; c < NC; ++c) shc = ca_clip == c ? shift[c] : shc;
            const bool fail = qfold_unit_query<D, NP_D>(GR, tag, tid, ca_clip * gco + O_CQ, ca_clip * gco + O_STAT,
                                                        AXW_COLD(qf) + (long)l * qfold_stride(D) + (long)D * D, ca_head, shc, qs, p.err, ctl);
            if (fail) ctl[0] = 1;
            AXW_STAMP(7)
            AXW_BARRIER_CHECK(0x500 + l)
            AXW_STAMP(8)
          }
        }
        // ---- cross-attention query
#pragma unroll
        for (int c = 0; c < NC && !QF; ++c) {
          unsigned y[GD];
          const bool fail = gather2<GPD>(GR, tag, y, p.err, ctl, AXW_PAIRS_D(O_Y1, c));
          AXW_STAMP(5)
          AXW_TL(3)
          AXW_LN_STAGE(y, true, fail, 0x400 + l, c)
          AXW_STAMP(6)
          AXW_TL(4)
          // ---- cross-attention unit of THIS clip: collect the head's query (a unit of clip 0 runs before clip 1's query rows:
          //      its query is already on its way, and nothing it needs waits behind the other clip's LayerNorm)
          if (cu >= 0 && cu / NU == c) {
            const int ca_clip = cu / NU, ca_head = (cu % NU) / kCrossSplit;
            unsigned v[2];
            const bool fail = gather2<1>(GR, tag, v, p.err, ctl, [&](int) { return tid < 32 ? ca_clip * gco + O_CQ + ca_head * 64 + 2 * tid : -1; });
            if (tid < 32) {
              unsigned hi, lo;
              h16split2(__uint_as_float(v[0]), __uint_as_float(v[1]), hi, lo);
              qs[tid] = hi;
              qs[32 + tid] = lo;
            }
            if (fail) ctl[0] = 1;
            AXW_STAMP(7)
            AXW_BARRIER_CHECK(0x500 + l)
            AXW_STAMP(8)
          }
        }
        if constexpr (!QF) ln_prefetch(FL + DecArena::F_MLP_LN_W * D, FL + DecArena::F_MLP_LN_B * D);
        // ---- cross-attention output projection: merge the partials of every head
        if (in_o) {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            // the pollers collect the first NP1 granules of the partial records, the (idle) compute waves the rest
            unsigned y[2 * GP1];  // pair pi of the records: record pi / 33, granules 2 * (pi % 33), +1
            const bool fail = gather2<GP1>(GR, tag, y, p.err, ctl, [&](int j) { const int pi = tid + j * PL; return pi < NPP1 ? c * gco + O_PART + (pi / (kPS / 2)) * kRec + 2 * (pi % (kPS / 2)) : -1; });
            float* pbuf = act + D;  // [H][kCrossSplit][66] (shared by the clips: the merge below is done with it before the other clip's records arrive)
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
                actc[c][i] = merge_cross_records(pbuf, i);
              }
            }
            wg_barrier();
            AXW_STAMP(10)
            AXW_TL(6)
          }
        }
        // ---- mlp.0
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if constexpr (QF) {  // x += y1, then += y2: the unfolded launch's order of additions
            unsigned y[2 * GD];
            const bool fail = gather2<2 * GPD>(GR, tag, y, p.err, ctl, [&](int j) { const int pr = tid + (j % GPD) * PL; return 2 * pr < D ? c * gco + (j < GPD ? O_Y1 : O_Y2) + 2 * pr : -1; });
#pragma unroll
            for (int k = 0; k < GD; ++k) if (el(k) < D) x[c][k] = (x[c][k] + __uint_as_float(y[k])) + __uint_as_float(y[GD + k]);
            AXW_STAMP(11)
            AXW_TL(7)
            AXW_LN_STAGE(y, false, fail, 0x700 + l, c)
          } else {
            unsigned y[GD];
            const bool fail = gather2<GPD>(GR, tag, y, p.err, ctl, AXW_PAIRS_D(O_Y2, c));
            AXW_STAMP(11)
            AXW_TL(7)
            AXW_LN_STAGE(y, true, fail, 0x700 + l, c)
          }
          AXW_STAMP(12)
          AXW_TL(8)
        }
        g2_prefetch(l + 1 < L ? l + 1 : 0);
        if (l + 1 < L) ln_prefetch(FL + DecArena::f_stride(D) + DecArena::F_ATTN_LN_W * D, FL + DecArena::f_stride(D) + DecArena::F_ATTN_LN_B * D);
        else ln_prefetch(AXW_COLD(ln_w), AXW_COLD(ln_b));
        // ---- mlp.2: the 4d-wide hidden vector is the largest hand-off; the pollers collect its first half, the compute
        //      waves (idle until it is complete, their rows already in registers) the second half. One LDS vector for both
        //      clips: clip 1's halves are written once the compute waves are done with clip 0's rows (one more barrier).
        if (in_f2) {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            constexpr int GH = (F / 4 + PL - 1) / PL;  // pairs per lane of the first half
            unsigned y[2 * GH];
            const bool fail = gather2<GH>(GR, tag, y, p.err, ctl, [&](int j) { const int pi = tid + j * PL; return pi < F / 4 ? c * gco + O_HID + 2 * pi : -1; });
            if (c >= 1) wg_barrier();  // the compute waves have read the previous clip's hidden vector
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
        }
      }  // layers

      steps_run = step + 1;
      asm volatile("" : "+v"(tid));
      if (step < 3) {  // SOT steps: feed the next forced token, logits are discarded (Whisper.cpp:214-217)
#pragma unroll
        for (int c = 0; c < NC; ++c) tok[c] = AXW_COLD(sot)[step + 1];
        continue;
      }
      // ---- final LayerNorm for the vocabulary projection, then merge the argmax partials of every workgroup
      {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          unsigned y[GD];
          const bool fail = gather2<GPD>(GR, (unsigned)(step * L + L), y, p.err, ctl, AXW_PAIRS_D(O_Y3, c));
          AXW_LN_STAGE(y, true, fail, 0x900, c)
        }
        AXW_STAMP(14)
        wg_barrier();  // B3: the compute waves have their workgroup argmax
        bool fail2 = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          unsigned v[2];  // {value, row index} of workgroup tid: one pair
          fail2 |= gather2<1>(GR, (unsigned)(step + 1), v, p.err, ctl, [&](int) { return tid < P ? c * gco + O_AMAX + 2 * tid : -1; });
          // first max wins (Whisper.cpp:42-45)
          float cv = tid < P ? __uint_as_float(v[0]) : -INFINITY;
          int ci = tid < P ? (int)v[1] : 0x7fffffff;
          wave_argmax(cv, ci);  // DPP + row swaps: the __shfl_xor form is twelve dependent ds_bpermute round trips on the token's path
          if ((tid & 63) == 0) { am_vc[c][tid >> 6] = cv; am_ic[c][tid >> 6] = ci; }
        }
        if (fail2) ctl[0] = 1;
        AXW_STAMP(15)
        AXW_BARRIER_CHECK(0xA00)  // B4
      }
      int best[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float cv = am_vc[c][0];
        int best_idx = am_ic[c][0];
        for (int w2 = 1; w2 < NPW; ++w2)
          if (am_vc[c][w2] > cv || (am_vc[c][w2] == cv && am_ic[c][w2] < best_idx)) { cv = am_vc[c][w2]; best_idx = am_ic[c][w2]; }
        // no logit compared greater than -inf (all NaN / -inf: non-finite audio): std::max_element returns index 0
        // (Whisper.cpp:42-45); never let the "no candidate" index reach the embedding lookup
        if ((unsigned)best_idx >= (unsigned)AXW_COLD(n_vocab)) best_idx = 0;
        best[c] = best_idx;
      }
      wg_barrier();  // B5: am_v/am_i are free again
      const int gi = step - 3;
      if (NC == 1 && AXW_COLD(forced)) {
        if (wg == 0 && tid == 0 && AXW_COLD(argmax_dump) && gi <= AXW_COLD(n_forced)) AXW_COLD(argmax_dump)[gi] = best[0];
        if (gi < AXW_COLD(n_forced)) tok[0] = AXW_COLD(forced)[gi];
      } else {
        // Whisper.cpp:219-222 per clip: eot, the context's end or the clip's budget ends ITS loop; with two clips the one
        // that is through rides along (same feed, results ignored) until the other is, too
        bool all_fin = true;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int mn = c == 0 ? AXW_COLD(max_new) : (c == 1 ? AXW_COLD(max_new1) : AXW_COLD(max_new2));
          if (!fin[c] && (best[c] == AXW_COLD(eot) || step + 1 >= AXW_COLD(n_ctx) || n_out[c] >= mn)) fin[c] = true;
          if (!fin[c]) {
            if (wg == 0 && tid == 0) (c == 0 ? AXW_COLD(out_ids) : AXW_COLD(out_ids1) + (long)(c - 1) * AXW_COLD(n_ctx))[n_out[c]] = best[c];
            ++n_out[c];
            tok[c] = best[c];
          }
          all_fin &= fin[c];
        }
        if (all_fin) { n_done = NC; break; }
      }
    }
#undef AXW_LN_STAGE
#undef AXW_LN_STAGE_X
#undef AXW_SA0_POLL
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
    constexpr bool kEarlyFc2 = NC == 1 && CF <= 6;
    const bool early_fc2 = kEarlyFc2 && in_f2 && !is_fc1;
    const int pk_d = 0;  // d rows in passes of CT/LD (or CT/LF) rows: never more producers than workgroups (P <= d)
    // two register sets for the d-wide layers are enough: a phase computes from one while the next phase's rows land
    // in the other (qkv A, o B, cq A, co B, mlp.0 A, [mlp.2 F], next qkv / vocabulary A)
    RowSet<LD, CD> ra, rb;
    RowSet<LF, CF> rs_fc2;
    ra.prefetch(p.wl, p.fl + DecArena::F_B_QKV * D, D, 3 * D, rwg, P, ctid, pk_qkv);
    {  // the first layer's cross-attention unit has no previous layer to hide behind
      const int cu0 = ca_unit_of(0);
      if (cu0 >= 0) {  // its K tiles; the V tiles are layer 0's own pieces
        const int lane = ctid & 63, cw = __builtin_amdgcn_readfirstlane(ctid >> 6);
        const int u0 = cu0 % NU;
        const long off = (long)(cu0 / NU) * p.cross_clip_stride + (long)(u0 / kCrossSplit) * 24 * 4096 + (long)((u0 % kCrossSplit) * NCW + cw) * 4096;
        for (int i = 0; i < 8; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(p.cross_k + off + i * 512 + lane * 8), (lds_ptr_t)(sK + cw * 4096 + i * 512), 16, 0, kKvAux);
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
        // Cross K/V tiles are constant during the utterance: they are staged into LDS ahead of their use (LDS-DMA, 16 x 1 KiB
        // per wave), a few instructions after each publish, so that no publish waits behind a burst of DMA requests. A
        // workgroup may own units in consecutive layers (ca_unit_of), so the region is free only once THIS layer's block is
        // through: the K tiles (pieces 0-7) of the next layer's unit go out behind the publishes that follow the block, its
        // V tiles (8-15) behind the next layer's publishes that precede it.
        const int ln = l + 1 < L ? l + 1 : 0;
        const int cun = ca_unit_of(step * L + l + 1);
        auto kv_piece = [&](int i0, int i1) {
          const int un = i0 < 8 ? cun : cu, lay = i0 < 8 ? ln : l;  // (a call never straddles piece 8)
          if (un < 0) return;
          const int uu = un % NU;
          const int kb = (uu % kCrossSplit) * NCW + cw;  // 64-key block of this wave (24 blocks = t_pad 1536)
          const long off = (long)(un / NU) * p.cross_clip_stride + (long)lay * p.cross_layer_stride + (long)(uu / kCrossSplit) * 24 * 4096 + (long)kb * 4096;
          for (int i = i0; i < i1; ++i) {
            const h16* src = (i < 8 ? p.cross_k : p.cross_v) + off + (i & 7) * 512 + lane * 8;
            h16* dst = (i < 8 ? sK : sV) + cw * 4096 + (i & 7) * 512;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)dst, 16, 0, kKvAux);
          }
        };
        // self-attention of one head over keys 0..step (export_onnx.py:103-147: the -60000 mask + the separate current-token
        // column of the reference equal causal attention), clip 0: the LDS cache. No second workgroup barrier: the compute wave
        // that arrives last merges the block partials and publishes.
#define AXW_SA0_COMP                                                                                                   \
  {                                                                                                                    \
    AXW_BARRIER_CHECK(0x200 + l)                                                                                       \
    const int nblk = (step >> 6) + 1;                                                                                  \
    if (cw < nblk) attn_block<true>(sK + cw * 4096, sV + cw * 4096, qs, cw * 64 + lane <= step, pscr + cw * 64, wpart + cw * kPS, lane); \
    __builtin_amdgcn_wave_barrier();                                                                                   \
    int old = 0;                                                                                                       \
    if (lane == 0) old = __hip_atomic_fetch_add(ctl + 3, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);           \
    old = __builtin_amdgcn_readfirstlane(old);                                                                         \
    if ((old + 1) % NCW == 0) {                                                                                        \
      float m, lt, ov;                                                                                                 \
      merge_partials(wpart, nblk, lane, &m, &lt, &ov);                                                                 \
      gput(G + O_ATT + sa_head * 64 + lane, tag, ov / lt);                                                             \
    }                                                                                                                  \
    AXW_STAMP(18)                                                                                                      \
    AXW_TL(11)                                                                                                         \
  }
        // ---- QKV rows (export_onnx.py:245-247)
        float res[2];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          AXW_BARRIER_CHECK(0x100 + l)
          wg_barrier();
          AXW_STAMP(16)
          AXW_TL(18)
          ra.run(w_qkv, b_qkv, D, actc[c], ctid, res);
          ra.publish(ctid, res, pk, ctl + 2, G + c * gco + O_QKV, tag, [](float v) { return v; });
          if (c == 0) {  // the next phase's rows are requested behind the FIRST clip's publish: they land under the second clip's rows
            rb.prefetch(w_o, b_o, D, D, rwg, P, ctid, pk_d);
            kv_piece(8, 11);
          }
          AXW_STAMP(17)
          AXW_TL(10)
          if (c == 0 && l == sa_layer && sa_first) AXW_SA0_COMP
        }
        // QF, row producers: A0 = W_cq (g . x0) of this slot's row for EVERY clip (g . x0 has been in LDS since the clip's QKV stage);
        // the rows of W_cq take the set the QKV rows leave. In the layer whose head this workgroup owns, behind clip 0's blocks.
        auto a0_rows = [&]() {
          if constexpr (QF) {
            if (in_o) {
#pragma unroll
              for (int c = 0; c < NC; ++c) {
                float ra0[2];
                ra.run(w_cq, nullptr, D, xgc(c), ctid, ra0);
                if (ctid % LD == 0) a0s(c)[ctid / LD] = ra0[0];
              }
            }
          }
        };
        if constexpr (QF) {
          if (in_o) ra.prefetch(w_cq, nullptr, D, D, rwg, P, ctid, pk_d);
          if (l != sa_layer) a0_rows();
        }
        // ---- self-attention (clip 1's blocks are the poller waves' — see there: only the hand-over of its query is shared)
        if (l == sa_layer) {
          if (!sa_first) AXW_SA0_COMP
          a0_rows();
#pragma unroll
          for (int c = 1; c < NC; ++c) AXW_BARRIER_CHECK(0x200 + l)
        }
        // ---- attention output projection
        if constexpr (QF) {
          if (in_o) {
            // y1 rows as always, T = A0 + M a + d for the same rows and the two sums of this slice of x1 = x0 + y1, clip by clip:
            // three lines, one store instruction each, all by the compute wave that arrives last (decode_persistent.hip)
            const float* qfl = AXW_COLD(qf) + (long)l * qfold_stride(D);
            const int slot = ctid / LD, row = rb.r0 + slot < rb.r1 ? rb.r0 + slot : rb.r0;
            RowSetF32<LD, CD> rm;
            rm.prefetch(qfl, qfl + (long)D * D, D, row, rb.r0 + slot < rb.r1, ctid);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              AXW_BARRIER_CHECK(0x300 + l)
              AXW_STAMP(19)
              AXW_TL(19)
              rb.run(w_o, b_o, D, actc[c], ctid, res);
              const float tq = rm.run(actc[c], ctid);
              const int j = ctid % LD, nrows = rb.r1 - rb.r0;
              if (j == 0 && slot < nrows) {
                pk[slot] = res[0];
                pk[32 + slot] = tq + a0s(c)[slot];
                pscr[slot] = (x0s(c)[slot] + res[0]) - shs(c)[0];
              }
              qfold_publish(lane, pk, pscr, ctl + 2, G + c * gco, O_CQ, O_STAT, O_Y1, rb.r0, nrows, rwg, tag);
            }
          }
          rb.prefetch(w_co, b_co, D, D, rwg, P, ctid, pk_d);
          kv_piece(11, 16);
          AXW_STAMP(20)
          AXW_TL(12)
          // ---- cross-attention over one third of the 1536 padded keys, if this workgroup holds a unit of this layer
          if (cu >= 0) {
            const int ca_clip = cu / NU, ca_head = (cu % NU) / kCrossSplit, ca_split = cu % kCrossSplit;
            AXW_BARRIER_CHECK(0x500 + l)
            cross_unit_block(sK, sV, qs, pscr, wpart, ctl + 3, G + ca_clip * gco + O_PART + (ca_head * kCrossSplit + ca_split) * kRec, tag, ca_split,
                             p.n_audio_ctx, cw, lane);
            AXW_STAMP(23)
            AXW_TL(14)
          }
        } else {
          if (in_o) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              AXW_BARRIER_CHECK(0x300 + l)
              AXW_STAMP(19)
              AXW_TL(19)
              rb.run(w_o, b_o, D, actc[c], ctid, res);
              rb.publish(ctid, res, pk, ctl + 2, G + c * gco + O_Y1, tag, [](float v) { return v; });
            }
          }
          ra.prefetch(w_cq, b_cq, D, D, rwg, P, ctid, pk_d);
          kv_piece(11, 16);
          AXW_STAMP(20)
          AXW_TL(12)
        }
        // ---- cross-attention query (export_onnx.py:221-230)
#pragma unroll
        for (int c = 0; c < NC && !QF; ++c) {
          AXW_BARRIER_CHECK(0x400 + l)
          wg_barrier();
          AXW_STAMP(21)
          AXW_TL(20)
          ra.run(w_cq, b_cq, D, actc[c], ctid, res);
          ra.publish(ctid, res, pk, ctl + 2, G + c * gco + O_CQ, tag, [](float v) { return v; });
          if (c == 0) rb.prefetch(w_co, b_co, D, D, rwg, P, ctid, pk_d);
          AXW_STAMP(22)
          AXW_TL(13)
          // ---- cross-attention over one third of the 1536 padded keys: the unit of THIS clip, if this workgroup holds one
          if (cu >= 0 && cu / NU == c) {
            const int ca_clip = cu / NU, ca_head = (cu % NU) / kCrossSplit, ca_split = cu % kCrossSplit;
            AXW_BARRIER_CHECK(0x500 + l)
            cross_unit_block(sK, sV, qs, pscr, wpart, ctl + 3, G + ca_clip * gco + O_PART + (ca_head * kCrossSplit + ca_split) * kRec, tag, ca_split,
                             p.n_audio_ctx, cw, lane);
            AXW_STAMP(23)
            AXW_TL(14)
          }
        }
        // ---- cross-attention output projection
        if (in_o) {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            {
              unsigned y[2 * GP2];
              const bool fail = gather2<GP2>(GR, tag, y, p.err, ctl, [&](int j) { const int pi = NPP1 + ctid + j * CT; return pi < NPP ? c * gco + O_PART + (pi / (kPS / 2)) * kRec + 2 * (pi % (kPS / 2)) : -1; });
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
            rb.run(w_co, b_co, D, actc[c], ctid, res);
            rb.publish(ctid, res, pk, ctl + 2, G + c * gco + O_Y2, tag, [](float v) { return v; });
          }
        }
        ra.prefetch(w_fc1, b_fc1, D, F, rwg, P, ctid, pk_f);
        if constexpr (kEarlyFc2) {
          if (early_fc2) rs_fc2.prefetch(w_fc2, b_fc2, F, D, rwg, P, ctid, pk_d);
        }
        kv_piece(0, 3);
        AXW_STAMP(25)
        AXW_TL(15)
        // ---- mlp.0 + GELU (export_onnx.py:298)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          AXW_BARRIER_CHECK(0x700 + l)
          wg_barrier();
          AXW_STAMP(26)
          AXW_TL(22)
          ra.run(w_fc1, b_fc1, D, actc[c], ctid, res);
          ra.publish(ctid, res, pk, ctl + 2, G + c * gco + O_HID, tag, [](float v) { return gelu_erf(v); });
          if (c == 0) kv_piece(3, 5);
          // (behind the LAST clip's publish: next to the mlp.0 rows, which the second clip still needs, the 6 chunks per lane of
          //  the mlp.2 rows do not fit the register budget — as loads followed by vmcnt(0) + a scratch store they cost 2 us per layer)
          if (c == NC - 1 && !early_fc2) rs_fc2.prefetch(w_fc2, b_fc2, F, D, rwg, P, ctid, pk_d);
          AXW_STAMP(27)
          AXW_TL(16)
        }
        // ---- mlp.2
        if (in_f2) {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            {
              constexpr int GH = (F / 4 + CT - 1) / CT;
              unsigned y[2 * GH];
              const bool fail = gather2<GH>(GR, tag, y, p.err, ctl, [&](int j) { const int pi = ctid + j * CT; return pi < F / 4 ? c * gco + O_HID + F / 2 + 2 * pi : -1; });
              if (c >= 1) wg_barrier();  // (with the pollers: everybody is done with the previous clip's hidden vector; this wave's own reads of it are behind it)
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
            rs_fc2.publish(ctid, res, pk, ctl + 2, G + c * gco + O_Y3, tag, [](float v) { return v; });
            AXW_STAMP(31)
          }
        }
        // next consumer of the residual stream: the next layer's QKV rows, the vocabulary projection, or the next step
        if (l + 1 < L) ra.prefetch(w_qkv + DecArena::w_stride(D), b_qkv + DecArena::f_stride(D), D, 3 * D, rwg, P, ctid, pk_qkv);
        else if (step >= 3) ra.prefetch(AXW_COLD(tok_emb), nullptr, D, AXW_COLD(n_vocab), rwg, P, ctid);
        else ra.prefetch(p.wl, p.fl + DecArena::F_B_QKV * D, D, 3 * D, rwg, P, ctid, pk_qkv);
        kv_piece(5, 8);
        AXW_STAMP(29)
        AXW_TL(17)
      }  // layers

      steps_run = step + 1;
      asm volatile("" : "+v"(ctid));
      if (step < 3) {
#pragma unroll
        for (int c = 0; c < NC; ++c) tok[c] = AXW_COLD(sot)[step + 1];
        continue;
      }
      // ---- logits = token_embedding . ln(x)  (tied embedding, export_onnx.py:364-385) + argmax (first max wins, Whisper.cpp:42-45)
      {
        constexpr int SD = CT / LD;
        const int lane = ctid & 63, cw = __builtin_amdgcn_readfirstlane(ctid >> 6);
        const int N = AXW_COLD(n_vocab);
        const int slot = ctid / LD, j = ctid % LD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {  // the final LayerNorm of every clip (one stage per clip on the pollers' side)
          AXW_BARRIER_CHECK(0x900)
          wg_barrier();
        }
        // activations of this lane's chunks stay in registers over all passes (wide models: re-read from LDS per pass,
        // the registers are needed for the rows in flight); clip 1's come from LDS (its own vector)
        constexpr bool kActInRegs = CD <= 3;
        float4 a[kActInRegs ? CD : 1][2];
        if constexpr (kActInRegs) {
#pragma unroll
          for (int i = 0; i < CD; ++i) {
            a[i][0] = *reinterpret_cast<const float4*>(act + (j + LD * i) * 8);
            a[i][1] = *reinterpret_cast<const float4*>(act + (j + LD * i) * 8 + 4);
          }
        }
        float bv[NC];
        int bi[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { bv[c] = -INFINITY; bi[c] = 0x7fffffff; }
        float* dump = (NC == 1 && AXW_COLD(logits_dump)) ? AXW_COLD(logits_dump) + (long)(step - 3) * N : nullptr;
        const int r0 = ra.r0, r1 = ra.r1;
        auto consume = [&](const u32x4 (&wr)[CD], int row) {
          float acc;
          if constexpr (kActInRegs) acc = rows_dot_reg<LD, CD>(wr, a);
          else acc = rows_dot<LD, CD>(wr, act, ctid);
          float accx[NC];  // the later clips' activations come from LDS (their own vectors)
#pragma unroll
          for (int c = 1; c < NC; ++c) accx[c] = rows_dot<LD, CD>(wr, actc[c], ctid);
          if (j == 0) {
            if (dump) dump[row] = acc;
            if (acc > bv[0]) { bv[0] = acc; bi[0] = row; }
#pragma unroll
            for (int c = 1; c < NC; ++c) if (accx[c] > bv[c]) { bv[c] = accx[c]; bi[c] = row; }
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
        ra.prefetch(p.wl, p.fl + DecArena::F_B_QKV * D, D, 3 * D, rwg, P, ctid, pk_qkv);
        // workgroup argmax: lanes with j == 0 hold candidates; the lower index wins ties
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (j != 0) { bv[c] = -INFINITY; bi[c] = 0x7fffffff; }
          wave_argmax(bv[c], bi[c]);
          if (lane == 0) { am_vc[c][8 + cw] = bv[c]; am_ic[c][8 + cw] = bi[c]; }
        }
        // compute waves only: named exchange through LDS, then wave 0 publishes. The pollers sit at B3 meanwhile.
        wg_barrier();  // B3 (all waves)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (ctid == 0) {
            float fv = bv[c];
            int fi = bi[c];
            for (int w2 = 1; w2 < NCW; ++w2)
              if (am_vc[c][8 + w2] > fv || (am_vc[c][8 + w2] == fv && am_ic[c][8 + w2] < fi)) { fv = am_vc[c][8 + w2]; fi = am_ic[c][8 + w2]; }
            am_vc[c][8] = fv; am_ic[c][8] = fi;
          }
          __builtin_amdgcn_wave_barrier();
          if (ctid < 2)  // value and index of this workgroup's best row in ONE store instruction
            gput_u(G + c * gco + O_AMAX + 2 * wg + ctid, (unsigned)(step + 1), ctid == 0 ? __float_as_uint(am_vc[c][8]) : (unsigned)am_ic[c][8]);
        }
        AXW_STAMP(30)
        AXW_BARRIER_CHECK(0xA00)  // B4
      }
      int best[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float cv = am_vc[c][0];
        int best_idx = am_ic[c][0];
        for (int w2 = 1; w2 < NPW; ++w2)
          if (am_vc[c][w2] > cv || (am_vc[c][w2] == cv && am_ic[c][w2] < best_idx)) { cv = am_vc[c][w2]; best_idx = am_ic[c][w2]; }
        if ((unsigned)best_idx >= (unsigned)AXW_COLD(n_vocab)) best_idx = 0;  // as in the pollers' copy of this merge
        best[c] = best_idx;
      }
      wg_barrier();  // B5
      const int gi = step - 3;
      if (NC == 1 && AXW_COLD(forced)) {
        if (gi < AXW_COLD(n_forced)) tok[0] = AXW_COLD(forced)[gi];
      } else {
        bool all_fin = true;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int mn = c == 0 ? AXW_COLD(max_new) : (c == 1 ? AXW_COLD(max_new1) : AXW_COLD(max_new2));
          if (!fin[c] && (best[c] == AXW_COLD(eot) || step + 1 >= AXW_COLD(n_ctx) || n_out[c] >= mn)) fin[c] = true;
          if (!fin[c]) { ++n_out[c]; tok[c] = best[c]; }
          all_fin &= fin[c];
        }
        if (all_fin) { n_done = NC; break; }
      }
    }
  }

  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no LDS-DMA may still be in flight when the workgroup's LDS is released
  if (PROF) {
    __syncthreads();
    if (tid < 64) AXW_COLD(prof)[(long)wg * 64 + tid] = prof_acc[tid];
  }
  if (wg == 0 && tid == 0) {
    AXW_COLD(n_out)[0] = n_out[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) AXW_COLD(n_out1)[c - 1] = n_out[c];
    AXW_COLD(state)->step = steps_run;
    AXW_COLD(state)->n_done = n_done;
  }
#undef AXW_SA0_COMP
#undef AXW_COLD
#undef AXW_BARRIER_CHECK
#undef AXW_STAMP
#undef AXW_TL
}

// ---------------------------------------------------------------------------------------- host side
static size_t persist2_lds_bytes(int d, int nc) {
  // the one-clip launch's LDS + per later clip its d-wide input vector, query (64 words), argmax scratch (32) and the
  // self-attention k, v rows of its current step (64) + the poller waves' attention scratch
  return (size_t)kKvBytes + ((size_t)4 * d + d / 8 + NCW * kPS + 2 * NPW + 64 + 16 + 16 + 16 + 64 + NCW * 64) * 4 + 64 * 8 + 64 +
         ((size_t)(nc - 1) * (d + 64 + 32 + 64) + NPW * kPS + NPW * 64 + (size_t)nc * 68) * 4;  // (+ the query fold's 68 words per clip)
}
int decode_persistent_max_clips(int d_model, int n_head, int n_layer, int grid) {
  // every linear layer must be ONE pass of rows per workgroup (a second pass overwrites the rows the next clip still
  // needs): true up to d_model 768, not for 1280 (mlp.0: 20 rows per workgroup in passes of 16)
  switch (d_model) { case 128: case 256: case 384: case 512: case 768: break; default: return 1; }
  int nc = 1;
  // every workgroup that owns no self-attention head takes at most ONE cross-attention unit (a clip's head and key range) per
  // layer, and the later clips' vectors must fit what the K/V region leaves of the CU's 160 KB of LDS
  while (nc < 3 && grid - n_layer * n_head >= (nc + 1) * kCrossSplit * n_head && persist2_lds_bytes(d_model, nc + 1) <= 160 * 1024) ++nc;
  return nc;
}

template <int LD, int CD, int LF, int CF, int NC, bool PROF, bool QF>
static hipError_t launch_multi_q(const PersistParams& p, int grid, hipStream_t s) {
  const size_t lds = persist2_lds_bytes(8 * LD * CD, NC);
  auto kfn = decode_persistent_kernel<LD, CD, LF, CF, PROF, NC, QF>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(PT), lds, s, p);
  return hipGetLastError();
}
template <int LD, int CD, int LF, int CF, int NC, bool PROF = false>
static hipError_t launch_multi(const PersistParams& p, int grid, hipStream_t s) {  // the query fold where the engine built its arena
  return p.qf ? launch_multi_q<LD, CD, LF, CF, NC, PROF, true>(p, grid, s) : launch_multi_q<LD, CD, LF, CF, NC, PROF, false>(p, grid, s);
}
template <int LD, int CD, int LF, int CF>
static hipError_t launch_nc(const PersistParams& p, int grid, hipStream_t s) {
  return p.n_clip == 2 ? launch_multi<LD, CD, LF, CF, 2>(p, grid, s) : launch_multi<LD, CD, LF, CF, 3>(p, grid, s);
}

hipError_t launch_decode_persistent2(const PersistParams& p, int d_model, int grid, hipStream_t s) {
  if ((p.n_clip != 2 && p.n_clip != 3) || p.forced || p.logits_dump || p.argmax_dump) return hipErrorInvalidValue;
  // the in-kernel timeline (AX_WHISPER_PERSIST_PROF): Whisper-small's shape. Phase sums cover all clips; the
  // absolute stamps of one layer are clip 0's (first writer) or clip 1's (last writer, AX_WHISPER_PERSIST_PROF_CLIP=1)
  if (p.prof) {
    if (d_model != 768) return hipErrorInvalidValue;
    return p.n_clip == 2 ? launch_multi<32, 3, 64, 6, 2, true>(p, grid, s) : launch_multi<32, 3, 64, 6, 3, true>(p, grid, s);
  }
  switch (d_model) {
    case 128: return launch_nc<16, 1, 32, 2>(p, grid, s);
    case 256: return launch_nc<32, 1, 64, 2>(p, grid, s);
    case 384: return launch_nc<16, 3, 64, 3>(p, grid, s);
    case 512: return launch_nc<32, 2, 64, 4>(p, grid, s);
    case 768: return launch_nc<32, 3, 64, 6>(p, grid, s);
    default: return hipErrorInvalidValue;
  }
}

}  // inline namespace AXW_NS
}  // namespace axw
