// engine_decode.cpp — axw::Engine: the decode paths (launch-per-phase, clip-block / split-K sequences, step graphs, the persistent
// launches) — Whisper::run_decoder + the host loop of the reference (Whisper.cpp:207-222,290-346).
#include "engine_impl.hpp"

namespace axw {
inline namespace AXW_NS {

// ------------------------------------------------------------------------------ decoder
void Engine::reset_decode_state(int batch, const int* max_new_clip) {
  hipStream_t s = stream();
  {  // per-clip id budgets (a ragged batch); without them every clip gets the whole context
    std::vector<int> mn(batch, cfg_.n_text_ctx);
    if (max_new_clip)
      for (int b = 0; b < batch; ++b) mn[b] = max_new_clip[b] > 0 ? max_new_clip[b] : cfg_.n_text_ctx;
    HIP_CHECK(hipMemcpyAsync(d_max_new_clip_, mn.data(), (size_t)batch * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));  // mn is a stack vector
  }
  HIP_CHECK(hipMemsetAsync(d_state_, 0, sizeof(DecState), s));
  HIP_CHECK(hipMemsetAsync(d_done_, 0, (size_t)batch * 4, s));
  HIP_CHECK(hipMemsetAsync(d_off_, 0, (size_t)batch * 4, s));
  HIP_CHECK(hipMemsetAsync(d_nout_, 0, (size_t)batch * 4, s));
  std::vector<int> t0(batch, sot_seq_[0]);
  HIP_CHECK(hipMemcpyAsync(d_tok_, t0.data(), (size_t)batch * 4, hipMemcpyHostToDevice, s));
  // x of step 0; every later step's embedding is produced by the previous step's advance kernel
  launch_embed(tok_emb_, dec_pos_, d_tok_, d_off_, d_xdec_, batch, cfg_.n_text_state, s);
  HIP_CHECK(hipStreamSynchronize(s));
}

// One decoder step for `batch` slots: the launch sequence that is captured into the step graph.
void Engine::enqueue_decode_step(int batch, int max_new, const int* d_forced, int n_forced, float* d_logits,
                                 long logits_stride, int* d_argmax) {
  const int d = cfg_.n_text_state, H = cfg_.n_text_head, L = cfg_.n_text_layer, Tc = cfg_.n_text_ctx;
  hipStream_t s = stream();
  // 3+ clips: the clip-block sequence (a 4-clip step: 0.78 ms through the GEMV family, 0.59 through clip-block GEMMs);
  // one clip that cannot use the persistent launch and two clips that cannot either stay on the GEMV family
  if (batch > gemv_max_) {
    enqueue_decode_step_batched(batch, max_new, d_forced, n_forced, d_logits, logits_stride, d_argmax);
    return;
  }

  // the VALU GEMV handles <= 4 clips per launch; tile the batch (gemv_max_ <= 4: one tile)
  auto gemv = [&](GemvParams p, auto&& offset) {
    for (int b0 = 0; b0 < batch; b0 += 4) {
      GemvParams q = p;
      q.batch = std::min(4, batch - b0);
      offset(q, b0);
      if (step_mask_ & 1) launch_gemv(q, s);
    }
  };
  auto attn = [&](const h16* kc, const h16* vc, long stride, int n_keys, int cap_blocks, float* part, int n_split) {
    DecAttnParams a{};
    a.q = d_qdec_; a.k = kc; a.v = vc; a.kv_batch_stride = stride; a.part = part; a.n_split = n_split;
    a.batch = batch; a.n_head = H; a.d_model = d; a.n_keys = n_keys; a.cap_blocks = cap_blocks; a.state = d_state_;
    a.off = d_off_;
    a.done = d_forced ? d_done_none_ : d_done_;
    if (step_mask_ & 2) launch_decode_attention(a, s);
  };

  const long self_stride = (long)H * Tc * 64, cross_stride = (long)H * t_pad_ * 64;
  for (int l = 0; l < L; ++l) {
    const DecLayerW& w = dec_[l];
    h16* sk = d_self_k_ + (size_t)l * cap_ * self_stride;
    h16* sv = d_self_v_ + (size_t)l * cap_ * self_stride;
    const h16* ck = d_cross_k_ + (size_t)l * cap_ * cross_stride;
    const h16* cv = d_cross_v_ + (size_t)l * cap_ * cross_stride;
    GemvParams p{};
    // q,k,v = Linear(attn_ln(x)); k,v appended to the self cache at row `step` (export_onnx.py:245-247, Whisper.cpp:328-342)
    p.W = w.w_qkv; p.bias = w.b_qkv; p.N = 3 * d; p.K = d;
    p.prologue = PRO_LAYERNORM; p.in = d_xdec_; p.ln_w = w.attn_ln_w; p.ln_b = w.attn_ln_b;
    p.epilogue = GEPI_QKV_CACHE; p.out = d_qdec_; p.k_cache = sk; p.v_cache = sv; p.kv_batch_stride = self_stride;
    p.d_model = d; p.n_ctx_pad = Tc; p.state = d_state_; p.off = d_off_;
    gemv(p, [&](GemvParams& q, int b0) { q.in += (long)b0 * d; q.out += (long)b0 * d; q.k_cache += b0 * self_stride; q.v_cache += b0 * self_stride; q.off += b0; });
    attn(sk, sv, self_stride, -1, Tc / 64, d_part_self_, split_self_);
    // x += out(attention)
    p = GemvParams{};
    p.W = w.w_o; p.bias = w.b_o; p.N = d; p.K = d;
    p.prologue = PRO_ATTN_COMBINE; p.part = d_part_self_; p.n_split = split_self_; p.n_head = H;
    p.epilogue = GEPI_RESID; p.out = d_xdec_; p.state = d_state_;
    gemv(p, [&](GemvParams& q, int b0) { q.part += (long)b0 * H * split_self_ * 66; q.out += (long)b0 * d; });
    // cross attention (export_onnx.py:221-230)
    p = GemvParams{};
    p.W = w.w_cq; p.bias = w.b_cq; p.N = d; p.K = d;
    p.prologue = PRO_LAYERNORM; p.in = d_xdec_; p.ln_w = w.cross_ln_w; p.ln_b = w.cross_ln_b;
    p.epilogue = GEPI_STORE; p.out = d_qdec_; p.state = d_state_;
    gemv(p, [&](GemvParams& q, int b0) { q.in += (long)b0 * d; q.out += (long)b0 * d; });
    attn(ck, cv, cross_stride, cfg_.n_audio_ctx, t_pad_ / 64, d_part_cross_, split_cross_);
    p = GemvParams{};
    p.W = w.w_co; p.bias = w.b_co; p.N = d; p.K = d;
    p.prologue = PRO_ATTN_COMBINE; p.part = d_part_cross_; p.n_split = split_cross_; p.n_head = H;
    p.epilogue = GEPI_RESID; p.out = d_xdec_; p.state = d_state_;
    gemv(p, [&](GemvParams& q, int b0) { q.part += (long)b0 * H * split_cross_ * 66; q.out += (long)b0 * d; });
    // mlp (export_onnx.py:298)
    p = GemvParams{};
    p.W = w.w_fc1; p.bias = w.b_fc1; p.N = 4 * d; p.K = d;
    p.prologue = PRO_LAYERNORM; p.in = d_xdec_; p.ln_w = w.mlp_ln_w; p.ln_b = w.mlp_ln_b;
    p.epilogue = GEPI_GELU; p.out = d_hid_; p.state = d_state_;
    gemv(p, [&](GemvParams& q, int b0) { q.in += (long)b0 * d; q.out += (long)b0 * 4 * d; });
    p = GemvParams{};
    p.W = w.w_fc2; p.bias = w.b_fc2; p.N = d; p.K = 4 * d;
    p.prologue = PRO_PLAIN; p.in = d_hid_;
    p.epilogue = GEPI_RESID; p.out = d_xdec_; p.state = d_state_;
    gemv(p, [&](GemvParams& q, int b0) { q.in += (long)b0 * 4 * d; q.out += (long)b0 * d; });
  }
  // logits = token_embedding . ln(x) (tied, export_onnx.py:364-385) fused with the argmax partials
  GemvParams p{};
  p.W = tok_emb_; p.bias = nullptr; p.N = cfg_.n_vocab; p.K = d;
  p.prologue = PRO_LAYERNORM; p.in = d_xdec_; p.ln_w = dec_ln_w_; p.ln_b = dec_ln_b_;
  p.epilogue = GEPI_LOGITS; p.state = d_state_; p.off = d_off_; p.amax_val = d_amax_val_; p.amax_idx = d_amax_idx_; p.amax_stride = n_amax_part_;
  p.skip_before_step = 3; p.logits_dump = d_logits; p.logits_dump_stride = logits_stride;
  gemv(p, [&](GemvParams& q, int b0) {
    q.in += (long)b0 * d; q.amax_val += (long)b0 * n_amax_part_; q.amax_idx += (long)b0 * n_amax_part_; q.off += b0;
    if (q.logits_dump) q.logits_dump += (long)b0 * logits_stride;
  });
  AdvanceParams a{};
  a.amax_val = d_amax_val_; a.amax_idx = d_amax_idx_; a.n_part = gemv_grid(p); a.amax_stride = n_amax_part_;
  a.state = d_state_; a.off = d_off_; a.tok = d_tok_; a.done = d_done_; a.n_out = d_nout_; a.out_ids = d_out_ids_; a.batch = batch;
  a.n_ctx = Tc; a.eot = cfg_.eot; a.max_new = max_new; a.n_vocab = cfg_.n_vocab; a.max_new_clip = d_max_new_clip_; a.sot = d_sot_;
  a.forced = d_forced; a.n_forced = n_forced; a.argmax_dump = d_argmax;
  a.tok_emb = tok_emb_; a.pos = dec_pos_; a.x = d_xdec_; a.d_model = d;
  a.done_host = d_forced ? nullptr : d_done_live_;
  if (step_mask_ & 4) launch_advance(a, s);
}

// bench "attn_stamp" (step_mask_ bit 16): the next {min begin, max end} slot, with what the launch is
unsigned long long* Engine::next_stamp(int layer, int cross, int b0, int nb) {
  if (!(step_mask_ & 16) || !d_stamp_) return nullptr;
  if (stamp_meta_.size() >= kStampLaunches) return nullptr;
  stamp_meta_.push_back({layer, cross, b0, nb});
  return d_stamp_ + 2 * kStampWgs * (stamp_meta_.size() - 1);  // room for kStampWgs workgroups per launch
}

// Decoder layers of clips [b0, b0 + nb) as clip-block GEMMs (decode_cgemm_kernel): LayerNorm is the prologue of its
// consumer and the residual add the epilogue of its producer, so a layer is 7 launches instead of 11
// (AX_WHISPER_BATCHED_LN=0: the older sequence with a separate LayerNorm/h16-pair preparation launch and split-K
// partials). b0 is a multiple of 16: every per-clip buffer of the range starts at a whole clip block.
void Engine::enqueue_layers_cblock(int b0, int nb, hipStream_t s, bool forced, bool one_branch) {
  const int d = cfg_.n_text_state, H = cfg_.n_text_head, L = cfg_.n_text_layer, Tc = cfg_.n_text_ctx;
  const long self_stride = (long)H * Tc * 64, cross_stride = (long)H * t_pad_ * 64;
  const long frag0 = (long)(b0 / 16) * 512;  // fragment-major pair layouts: clip blocks are 512 elements apart within a k-step
  float* x = d_xdec_ + (long)b0 * d;
  float* qd = d_qdec_ + (long)b0 * d;
  h16 *att_hi = d_att_[0] + frag0, *att_lo = d_att_[1] + frag0, *hid_hi = d_hidp_[0] + frag0, *hid_lo = d_hidp_[1] + frag0;
  const int* done = (forced ? d_done_none_ : d_done_) + b0;  // (never null: the attention launches read the flag unconditionally)
  auto cgemm = [&](const h16* W, const float* bias, int N, int K, int epi, int rt) {
    DecCGemmParams c{};
    c.W = W; c.bias = bias; c.N = N; c.K = K; c.batch = nb; c.nbs = nbs_; c.epilogue = epi; c.rt = rt;
    c.d_model = d; c.n_ctx_pad = Tc; c.state = d_state_; c.off = d_off_ + b0;
    return c;
  };
  const bool fuse_cq = true;  // the cross-attention workgroups project their own queries (d_model <= 1024)
  // Workgroups per (clip, head) of the cross-attention launch (its key blocks divided among them, at least four blocks
  // = one per wave each): at few clips one workgroup per (clip, head) leaves most CUs idle behind 24 sequential blocks
  // (3 clips: attention 0.245 -> 0.209 ms per step with 6 splits); from ~24 clips on there are enough (clip, head)
  // pairs and splitting only repeats the query projection (64 clips: 525 -> 588 ms with 2 splits).
  // At most ONE workgroup per CU (round 5; the bound was 320): every split repeats the LayerNorm and the query projection and adds a
  // record to the fold, and a second workgroup on a CU shares its memory path — decode ms per call at Whisper-small dims by splits
  // (profiles/r05_cross_split_sweep.txt): 4 clips 6: 249, 4: 241, 3: 241, 2: 246 | 6 clips 4: 259, 3: 250, 2: 251 |
  // 8 clips 6: 289, 4: 263, 3: 262, 2: 261 | 10 clips 2: 264.5, 1: 267 | 12 clips 2: 283, 1: 272.
  int cross_split = 1;
  {
    const int blocks = t_pad_ / 64;
    for (int c : {6, 4, 3, 2})
      if (c <= kCrossSplitMax && blocks % c == 0 && nb * H * c <= std::max(n_cu_, 64)) { cross_split = c; break; }
    if (cross_split_env_ > 0 && cross_split_env_ <= kCrossSplitMax && blocks % cross_split_env_ == 0) cross_split = cross_split_env_;
  }
  // Folded query: a split repeats nothing but a 3 KB gather, so the grid is sized for whole waves of workgroups instead (three
  // 136-register workgroups per CU are resident): AX_WHISPER_CROSS_SPLIT_FOLD forces a count (sweeps)
  int cross_split_fold = cross_split;
  {
    static const int env = [] { const char* e = getenv("AX_WHISPER_CROSS_SPLIT_FOLD"); return e ? atoi(e) : 0; }();
    const int blocks = t_pad_ / 64;
    if (env > 0 && env <= kCrossSplitMax && blocks % env == 0) cross_split_fold = env;
  }
  static const int gemm_stamp_point = [] { const char* e = getenv("AX_WHISPER_GEMM_STAMP_POINT"); return e ? atoi(e) : 0; }();
  int cur_layer = 0;
  // kind: 2 qkv, 3 o, 4 co, 5 fc1, 6 fc2 (0 / 1 are the attention launches) — bench "attn_stamp" puts every launch of the step on one axis
  auto cgo = [&](DecCGemmParams c, int kind) {
    if (step_mask_ & 16) { c.stamp = next_stamp(cur_layer, kind, b0, nb); c.stamp_point = gemm_stamp_point; }
    if (step_mask_ & 1) launch_decode_cgemm(c, s);
  };
  auto attn = [&](const h16* kc, const h16* vc, long stride, int n_keys, int cap_blocks) {
    DecAttnParams a{};
    a.q = qd; a.k = kc; a.v = vc; a.kv_batch_stride = stride; a.part = nullptr; a.n_split = 1;
    a.batch = nb; a.n_head = H; a.d_model = d; a.n_keys = n_keys; a.cap_blocks = cap_blocks; a.state = d_state_;
    a.off = d_off_ + b0;
    a.done = done;
    a.done_late = one_branch ? 1 : 0;  // the step is ONE chain of dependent launches (decoder.hip); tail branches of a multi-branch step keep the early check
    a.out_hi = att_hi; a.out_lo = att_lo; a.nbs = nbs_;
    return a;
  };
  const int n_blk = (nb + 15) / 16;
  // two row tiles per workgroup where one would make more workgroups than can be resident at once
  // (fewer, fatter workgroups at few clips — two row tiles from 100 or from 40 workgroups on — measured 2 % slower at 4 and 8 clips)
  auto rt_for = [&](int N) { return (N / 16) * n_blk > 512 ? 2 : 1; };

  for (int l = 0; l < L; ++l) {
    cur_layer = l;
    const DecLayerW& w = dec_[l];
    const DecLayerWP& wq = dec_packed_[l];
    h16* sk = d_self_k_ + ((size_t)l * cap_ + b0) * self_stride;
    h16* sv = d_self_v_ + ((size_t)l * cap_ + b0) * self_stride;
    const h16* ck = d_cross_k_ + ((size_t)l * cap_ + b0) * cross_stride;
    const h16* cv = d_cross_v_ + ((size_t)l * cap_ + b0) * cross_stride;
    // query fold (decode_gemm.hip "QUERY FOLD"): the QKV launch also computes A0 = W_cq (g_cross . x0), the o launch T = A0 + M a + d
    // and the block statistics of the new residual rows; the cross-attention workgroups start with their query
    // Measured (profiles/r06_cblock_qfold_ab.txt, Whisper-small, step at t = 224): 4 clips 0.4915 -> 0.4784 ms, 8: 0.5317 -> 0.5119,
    // 16: 0.6244 -> 0.6115; 64 clips (two branches): 1.091 -> 1.135 — the attention launches gain 1 % there (the W_cq rows come from L2
    // beside 147 MB of K/V per launch) and the two fatter GEMM launches cost 7 % of the chain. So: one-branch steps only.
    const bool qfold = !cfold_.empty() && fuse_cq && d <= 1024 && (one_branch || cfold_all_);
    float* a0 = d_a0_ + (long)b0 * d;
    float* statp = d_statp_ + (long)b0 * (d / 16) * 2;
    DecCGemmParams c = qfold ? cgemm(wq.w_qkv, cfold_[l].b_qkv4, 4 * d, d, GEPI_QKV_CACHE, rt_for(4 * d))
                             : cgemm(wq.w_qkv, w.b_qkv, 3 * d, d, GEPI_QKV_CACHE, rt_for(3 * d));
    c.x = x; c.ln_w = w.attn_ln_w; c.ln_b = w.attn_ln_b;
    c.out = qd; c.k_cache = sk; c.v_cache = sv; c.kv_batch_stride = self_stride;
    if (qfold) { c.fold_row0 = 3 * d; c.ln_w2 = w.cross_ln_w; c.out2 = a0; }
    cgo(c, 2);
    if (step_mask_ & 2) { DecAttnParams a = attn(sk, sv, self_stride, -1, Tc / 64); a.stamp = next_stamp(l, 0, b0, nb); launch_decode_attention(a, s); }
    c = qfold ? cgemm(wq.w_o, cfold_[l].b_o2, 2 * d, d, GEPI_RESID, 1) : cgemm(wq.w_o, w.b_o, d, d, GEPI_RESID, 1);
    c.a_hi = att_hi; c.a_lo = att_lo; c.out = x;
    if (qfold) { c.fold_row0 = d; c.W_lo = wq.m_lo; c.out2 = a0; c.stat_part = statp; }
    cgo(c, 3);
    if (qfold) {
      DecAttnParams a = attn(ck, cv, cross_stride, cfg_.n_audio_ctx, t_pad_ / 64);
      a.q = nullptr;
      a.n_split = cross_split_fold;
      a.mpart = d_attn_mpart_ + (long)b0 * H * kCrossSplitMax * 66;
      a.mcnt = d_attn_mcnt_ + (long)b0 * H;
      a.tq = a0; a.stat_part = statp; a.fold_s = cfold_[l].s; a.fold_c = cfold_[l].c;
      a.stamp = next_stamp(l, 1, b0, nb);
      if (step_mask_ & 2) launch_decode_attention(a, s);
    } else if (fuse_cq && d <= 1024) {  // the cross-attention workgroups project their own queries (decode_attention_kernel<1>)
      DecAttnParams a = attn(ck, cv, cross_stride, cfg_.n_audio_ctx, t_pad_ / 64);
      a.q = nullptr;
      a.n_split = cross_split;
      a.mpart = d_attn_mpart_ + (long)b0 * H * kCrossSplitMax * 66;
      a.mcnt = d_attn_mcnt_ + (long)b0 * H;
      a.x = x; a.ln_w = w.cross_ln_w; a.ln_b = w.cross_ln_b; a.wq = w.w_cq; a.bq = w.b_cq;
      a.stamp = next_stamp(l, 1, b0, nb);
      if (step_mask_ & 2) launch_decode_attention(a, s);
    } else {
      c = cgemm(wq.w_cq, w.b_cq, d, d, GEPI_STORE, 1);
      c.x = x; c.ln_w = w.cross_ln_w; c.ln_b = w.cross_ln_b; c.out = qd;
      cgo(c, 7);
      if (step_mask_ & 2) { DecAttnParams a = attn(ck, cv, cross_stride, cfg_.n_audio_ctx, t_pad_ / 64); a.stamp = next_stamp(l, 1, b0, nb); launch_decode_attention(a, s); }
    }
    c = cgemm(wq.w_co, w.b_co, d, d, GEPI_RESID, 1);
    c.a_hi = att_hi; c.a_lo = att_lo; c.out = x;
    cgo(c, 4);
    c = cgemm(wq.w_fc1, w.b_fc1, 4 * d, d, GEPI_GELU, rt_for(4 * d));
    c.x = x; c.ln_w = w.mlp_ln_w; c.ln_b = w.mlp_ln_b; c.out_hi = hid_hi; c.out_lo = hid_lo;
    cgo(c, 5);
    c = cgemm(wq.w_fc2, w.b_fc2, d, 4 * d, GEPI_RESID, 1);
    c.a_hi = hid_hi; c.a_lo = hid_lo; c.out = x;
    cgo(c, 6);
  }
}

// Branches of the batched step graph (see enqueue_decode_step_batched), whole clip blocks each (the last one may be
// partial). Measured on MI355X, Whisper-small, step t = 224 (profiles/r03_branch_table.txt; A/B/A/B per clip count):
//   up to 21 clips  1 branch   (21 clips: 0.720 ms with one, 0.723 with two)
//   22 .. 39        2 branches (16 + rest: 22 clips 0.809 -> 0.735 ms, 24: 0.818 -> 0.754, 28: 0.840 -> 0.790, 31: 0.853 -> 0.817;
//                               an attention launch of more than 256 workgroups — 22 clips x 12 heads — leaves a few CUs with two
//                               workgroups and everybody waits for them; two launches side by side do not)
//   40 .. 48        3 branches (16 + 16 + rest: 40 clips 0.955 -> 0.936 ms, 44: 1.006 -> 0.988, 48: 1.034 -> 1.005)
//   49 .. 64        2 branches (4 branches at 56 / 64 clips: 1.093 -> 1.13-1.16 / 1.152 -> 1.22 ms)
//   65 .. 96        3 branches (32 + 32 + rest: 72 clips 1.338 -> 1.301 ms, 80: 1.374 -> 1.367, 96: 1.664 -> 1.568)
//   97 and more     2 branches (112 clips: 1.745 with two, 1.811 with three; 128: 1.907 / 1.969; 192, 256: within 1 %)
// One step costs 17.9 us per clip at 64 clips, 14.9 at 128, 13.1 at 256 (the chain of small GEMMs is paid once per step): the
// slot scheduler's rate grows with its slot count (profiles/r03_big_batches.txt).
// AX_WHISPER_DECODE_BRANCHES overrides (1, 2, 3 or 4).
int Engine::decode_branches(int batch) const {
  static const int forced = [] { const char* e = getenv("AX_WHISPER_DECODE_BRANCHES"); return e ? atoi(e) : 0; }();
  const int min_per = 6;  // fewest clips the last branch may be left with
  int n = forced > 0 ? forced : (batch < 22 ? 1 : batch < 40 ? 2 : batch <= 48 ? 3 : batch <= 64 ? 2 : batch <= 96 ? 3 : 2);
  n = std::min(n, kMaxBranches);
  // every branch gets whole clip blocks; the last one at least min_per clips
  while (n > 1) {
    const int per = ((batch + n - 1) / n + 15) / 16 * 16;
    if (batch - (n - 1) * per >= min_per) break;
    --n;
  }
  return std::max(n, 1);
}

// Batched variant (3+ clips): LayerNorm -> h16 pairs (act_prep), MFMA GEMMs that read the weights once for the
// whole batch, one attention workgroup per (clip, head) writing its output directly (no split partials).
void Engine::enqueue_decode_step_batched(int batch, int max_new, const int* d_forced, int n_forced, float* d_logits,
                                         long logits_stride, int* d_argmax) {
  const int d = cfg_.n_text_state, H = cfg_.n_text_head, L = cfg_.n_text_layer, Tc = cfg_.n_text_ctx;
  hipStream_t s = stream();
  const long self_stride = (long)H * Tc * 64, cross_stride = (long)H * t_pad_ * 64;

  auto gemm = [&](DecGemmParams p, auto&& offset) {
    for (int b0 = 0; b0 < batch; b0 += 64) {
      DecGemmParams q = p;
      q.batch = std::min(64, batch - b0);
      q.a_hi += (long)(b0 / 16) * 512;  // fragment-major: clip blocks are 512 elements apart within a k-step
      q.a_lo += (long)(b0 / 16) * 512;
      q.nbs = nbs_;
      q.off = d_off_ + b0;
      offset(q, b0);
      if (step_mask_ & 1) launch_decode_gemm(q, s);
    }
  };
  // residual GEMMs write split-K partial sums; the next LayerNorm prep folds them (+ bias) into x, in fixed order
  int pend_n = 0;
  const float* pend_bias = nullptr;
  auto ln = [&](const float* g, const float* be) {
    if (step_mask_ & 8)
      launch_act_prep(d_xdec_, g, be, d_act_[0], d_act_[1], batch, d, true, nbs_, d_part_, pend_n, cap_, pend_bias, s);
    pend_n = 0;
  };
  auto ksplit_for = [&](int K) {
    const int KS = K / 32;
    for (int k = 4; k > 1; --k)
      if (KS % k == 0 && KS / k >= 8) return k;
    return 1;
  };
  auto resid = [&](const h16* W, const float* bias, int K, const h16* ahi, const h16* alo) {
    DecGemmParams p{};
    p.W = W; p.bias = nullptr; p.N = d; p.K = K; p.a_hi = ahi; p.a_lo = alo; p.epilogue = GEPI_PARTIAL; p.rt = 1;
    p.d_model = d; p.n_ctx_pad = Tc; p.state = d_state_; p.out = d_part_; p.ksplit = ksplit_for(K); p.part_batch = cap_;
    gemm(p, [&](DecGemmParams& q, int b0) { q.out += (long)b0 * d; });
    pend_n = p.ksplit;
    pend_bias = bias;
  };
  int stamp_layer = 0;
  auto attn = [&](const h16* kc, const h16* vc, long stride, int n_keys, int cap_blocks) {
    DecAttnParams a{};
    a.q = d_qdec_; a.k = kc; a.v = vc; a.kv_batch_stride = stride; a.part = nullptr; a.n_split = 1;
    a.batch = batch; a.n_head = H; a.d_model = d; a.n_keys = n_keys; a.cap_blocks = cap_blocks; a.state = d_state_;
    a.off = d_off_;
    a.done = d_forced ? d_done_none_ : d_done_;
    a.out_hi = d_att_[0]; a.out_lo = d_att_[1]; a.nbs = nbs_;
    if (n_keys >= 0) {  // cross-attention: few (clip, head) pairs leave CUs with one workgroup beside CUs with two (turbo, 16 clips: 320)
      int c = 1;
      for (int k : {6, 4, 3, 2})
        if (k <= kCrossSplitMax && cap_blocks % k == 0 && batch * H * k <= 640) { c = k; break; }
      if (cross_split_env_ > 0 && cross_split_env_ <= kCrossSplitMax && cap_blocks % cross_split_env_ == 0) c = cross_split_env_;
      a.n_split = c;
      a.mpart = d_attn_mpart_;
      a.mcnt = d_attn_mcnt_;
    }
    a.stamp = next_stamp(stamp_layer, n_keys >= 0 ? 1 : 0, 0, batch);
    if (step_mask_ & 2) launch_decode_attention(a, s);
  };
  auto base = [&](const h16* W, const float* bias, int N, int K, const h16* ahi, const h16* alo, int epi) {
    DecGemmParams p{};
    p.W = W; p.bias = bias; p.N = N; p.K = K; p.a_hi = ahi; p.a_lo = alo; p.epilogue = epi; p.rt = 1;
    p.d_model = d; p.n_ctx_pad = Tc; p.state = d_state_;
    return p;
  };

  if (batched_ln_) {
    // Clip-block sequence (enqueue_layers_cblock). With 32+ clips the batch runs as 2 BRANCHES of whole clip blocks that
    // fork here and join before the vocabulary projection: inside a captured step they become parallel branches of
    // the ONE step graph, so one branch's latency-bound chain of small GEMMs overlaps the other's bandwidth-bound
    // attention launches (a single chain leaves the chip idle between its ~85 dependent launches).
    const int nbr = decode_branches(batch);
    if (nbr == 1) {
      enqueue_layers_cblock(0, batch, s, d_forced != nullptr, true);
    } else {
      const int per = ((batch + nbr - 1) / nbr + 15) / 16 * 16;
      HIP_CHECK(hipEventRecord(ev_fork_, s));
      for (int i = 0; i < nbr; ++i) {
        const int b0 = i * per, nb = std::min(per, batch - b0);
        if (nb <= 0) break;
        hipStream_t bs = i == 0 ? s : branch_stream_[i - 1];
        if (i > 0) HIP_CHECK(hipStreamWaitEvent(bs, ev_fork_, 0));
        enqueue_layers_cblock(b0, nb, bs, d_forced != nullptr, false);
        if (i > 0) {
          HIP_CHECK(hipEventRecord(ev_join_[i - 1], bs));
          HIP_CHECK(hipStreamWaitEvent(s, ev_join_[i - 1], 0));
        }
      }
    }
  }
  for (int l = 0; l < L && !batched_ln_; ++l) {
    stamp_layer = l;
    const DecLayerW& w = dec_[l];
    h16* sk = d_self_k_ + (size_t)l * cap_ * self_stride;
    h16* sv = d_self_v_ + (size_t)l * cap_ * self_stride;
    const h16* ck = d_cross_k_ + (size_t)l * cap_ * cross_stride;
    const h16* cv = d_cross_v_ + (size_t)l * cap_ * cross_stride;
    ln(w.attn_ln_w, w.attn_ln_b);
    const DecLayerWP& wp = dec_packed_[l];
    DecGemmParams p = base(wp.w_qkv, w.b_qkv, 3 * d, d, d_act_[0], d_act_[1], GEPI_QKV_CACHE);
    p.out = d_qdec_; p.k_cache = sk; p.v_cache = sv; p.kv_batch_stride = self_stride;
    gemm(p, [&](DecGemmParams& q, int b0) { q.out += (long)b0 * d; q.k_cache += b0 * self_stride; q.v_cache += b0 * self_stride; });
    attn(sk, sv, self_stride, -1, Tc / 64);
    resid(wp.w_o, w.b_o, d, d_att_[0], d_att_[1]);
    ln(w.cross_ln_w, w.cross_ln_b);
    p = base(wp.w_cq, w.b_cq, d, d, d_act_[0], d_act_[1], GEPI_STORE);
    p.out = d_qdec_;
    gemm(p, [&](DecGemmParams& q, int b0) { q.out += (long)b0 * d; });
    attn(ck, cv, cross_stride, cfg_.n_audio_ctx, t_pad_ / 64);
    resid(wp.w_co, w.b_co, d, d_att_[0], d_att_[1]);
    ln(w.mlp_ln_w, w.mlp_ln_b);
    p = base(wp.w_fc1, w.b_fc1, 4 * d, d, d_act_[0], d_act_[1], GEPI_GELU);
    p.out_hi = d_hidp_[0]; p.out_lo = d_hidp_[1];
    gemm(p, [&](DecGemmParams& q, int b0) { q.out_hi += (long)(b0 / 16) * 512; q.out_lo += (long)(b0 / 16) * 512; });
    resid(wp.w_fc2, w.b_fc2, 4 * d, d_hidp_[0], d_hidp_[1]);
  }
  ln(dec_ln_w_, dec_ln_b_);
  DecGemmParams p = base(tok_emb_packed_, nullptr, cfg_.n_vocab, d, d_act_[0], d_act_[1], GEPI_LOGITS);
  const int vocab_rt = decode_logits_resident_ok(d, batch) ? 0 : logits_rt();
  p.rt = vocab_rt;
  p.amax_val = d_amax_val_; p.amax_idx = d_amax_idx_; p.amax_stride = n_amax_part_;
  p.skip_before_step = 3; p.logits_dump = d_logits; p.logits_dump_stride = logits_stride;
  gemm(p, [&](DecGemmParams& q, int b0) {
    q.amax_val += (long)b0 * n_amax_part_; q.amax_idx += (long)b0 * n_amax_part_;
    if (q.logits_dump) q.logits_dump += (long)b0 * logits_stride;
  });
  AdvanceParams a{};
  a.amax_val = d_amax_val_; a.amax_idx = d_amax_idx_; a.n_part = decode_gemm_grid(cfg_.n_vocab, vocab_rt); a.amax_stride = n_amax_part_;
  a.state = d_state_; a.off = d_off_; a.tok = d_tok_; a.done = d_done_; a.n_out = d_nout_; a.out_ids = d_out_ids_; a.batch = batch;
  a.n_ctx = Tc; a.eot = cfg_.eot; a.max_new = max_new; a.n_vocab = cfg_.n_vocab; a.max_new_clip = d_max_new_clip_; a.sot = d_sot_;
  a.forced = d_forced; a.n_forced = n_forced; a.argmax_dump = d_argmax;
  a.tok_emb = tok_emb_; a.pos = dec_pos_; a.x = d_xdec_; a.d_model = d;
  a.done_host = d_forced ? nullptr : d_done_live_;
  if (step_mask_ & 4) launch_advance(a, s);
}

// The persistent launch needs every workgroup resident at once; when it gives up (CUs taken by somebody else) the
// engine serves the next `backoff` one-clip requests through the launch-per-phase path and then tries again: 8, 32,
// 128, ... requests (capped at 4096), back to 8 after a success. The fast path is never lost for good on a shared box.
bool Engine::persistent_usable() {
  if (!persistent_ok_) return false;
  if (persist_skip_ > 0) {
    --persist_skip_;
    if (persist_skip_ == 0) cfg_.ints["persistent_decode"] = 1;  // the next request re-arms it
    return false;
  }
  return true;
}
void Engine::persistent_gave_up() {
  persist_backoff_ = std::min(persist_backoff_ ? persist_backoff_ * 4 : 8, 4096);
  persist_skip_ = persist_backoff_;
  ++persist_giveups_;
  cfg_.ints["persistent_decode"] = 0;
  cfg_.ints["persistent_giveups"] = persist_giveups_;
}
void Engine::persistent_succeeded() {
  persist_backoff_ = 0;
  cfg_.ints["persistent_decode"] = 1;
}

// After a failed capture the engine's own streams may be left in capture state ("operation failed due to a previous error
// during capture" on everything enqueued afterwards): they are replaced.
void Engine::recover_streams() {
  (void)hipGetLastError();
  auto renew = [](hipStream_t& st) {
    if (!st) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool bad = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    if (!bad) return;
    (void)hipStreamDestroy(st);
    st = nullptr;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  };
  renew(own_stream_);
  for (auto& b : branch_stream_) renew(b);
  (void)hipGetLastError();
}

// the streams the batched step of `batch` clips forks into (enqueue_decode_step_batched): branch i runs on branch_stream_[i - 1]
void Engine::ensure_branch_streams(int batch) {
  if (batch <= gemv_max_ || !batched_ln_) return;
  const int nbr = decode_branches(batch);
  for (int i = 1; i < nbr - 1 && i < kMaxBranches - 1; ++i)
    if (!branch_stream_[i]) HIP_CHECK(hipStreamCreateWithFlags(&branch_stream_[i], hipStreamNonBlocking));
}

hipGraphExec_t Engine::step_graph(int batch, int max_new) {
  const long key = ((long)batch * 1024 + max_new) * 32 + step_mask_;
  auto it = graphs_.find(key);
  if (it != graphs_.end()) return it->second;
  hipStream_t s = stream();
  hipGraph_t graph = nullptr;
  // nobody on this device allocates, copies synchronously or captures while this capture is open (iengine.hpp)
  std::lock_guard<std::recursive_mutex> capture_lock(device_capture_mutex(device_));
  ensure_branch_streams(batch);  // before the capture opens
  HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  hipError_t cap_err = hipSuccess;
  try {
    enqueue_decode_step(batch, max_new, nullptr, 0, nullptr, 0, nullptr);
    cap_err = hipStreamEndCapture(s, &graph);
  } catch (...) {
    (void)hipStreamEndCapture(s, &graph);
    recover_streams();
    throw;
  }
  if (cap_err != hipSuccess || !graph) {  // an invalidated capture must not leave the engine's streams unusable for good
    recover_streams();
    throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(cap_err) + " capturing the decoder step");
  }
  // graph and exec are released on every path out of here (the probe below may throw)
  struct Holder {
    hipGraph_t g;
    hipGraphExec_t e = nullptr;
    ~Holder() {
      if (e) (void)hipGraphExecDestroy(e);
      if (g) (void)hipGraphDestroy(g);
    }
  } hold{graph};
  HIP_CHECK(hipGraphInstantiate(&hold.e, graph, nullptr, nullptr, 0));
  {  // which hardware queue the graph's second branch runs on decides the slot stream's rate (engine_stream.cpp:
     // graph_branch_shares_queue); AX_WHISPER_ALIGN_QUEUES=0: as it falls.
     // ORDER: the probe REPLAYS the step, i.e. runs real decoder steps on whatever state the buffers hold. Every caller
     // therefore asks for its graph BEFORE it sets up the decode state of a request (greedy_loop, stream_begin, bench).
    static const bool align = [] { const char* e = getenv("AX_WHISPER_ALIGN_QUEUES"); return !(e && e[0] == '0'); }();
    if (align && !user_stream_ && batch > gemv_max_ && batched_ln_ && decode_branches(batch) >= 2 && step_mask_ == 15) {
      constexpr size_t kMaxPadStreams = 8;  // per engine, whatever the number of distinct (batch, max_new) graphs a server sees
      int tries = 0;
      bool aligned = graph_branch_shares_queue(hold.e, branch_stream_[0]);
      while (!aligned && tries < 4 && pad_streams_.size() < kMaxPadStreams) {
        HIP_CHECK(hipGraphExecDestroy(hold.e));
        hold.e = nullptr;
        hipStream_t pad = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&pad, hipStreamNonBlocking));
        pad_streams_.push_back(pad);
        HIP_CHECK(hipGraphInstantiate(&hold.e, graph, nullptr, nullptr, 0));
        ++tries;
        aligned = graph_branch_shares_queue(hold.e, branch_stream_[0]);  // the exec that is kept is the one that was probed
      }
      cfg_.ints["graph_queue_tries"] = tries;
      cfg_.ints["graph_queue_aligned"] = aligned ? 1 : 0;
    }
  }
  hipGraphExec_t exec = hold.e;
  hold.e = nullptr;  // kept: owned by graphs_ from here on
  graphs_[key] = exec;
  return exec;
}

// Whisper.cpp:207-222. Returns the number of decoder steps executed.
int Engine::greedy_loop(int batch, int max_new, const int* max_new_clip) {
  const int Tc = cfg_.n_text_ctx;
  if (max_new <= 0 || max_new > Tc - 4) max_new = Tc - 4;
  // One clip: the persistent launch. Two or three clips: ONE multi-clip persistent launch, phase by phase (one clip's rows are
  // computed while the others' hand-offs are in flight; decode_persistent2.hip) — Whisper-small, 444 ids per clip: 134 ms per
  // pair against 2 x 116 ms for one launch per clip (shapes without a multi-clip launch, AX_WHISPER_PERSIST2=0) and 316 ms through
  // the launch-per-phase path. Each clip stops at its own eot / budget.
  // (asked ONCE per request: persistent_usable() counts a back-off down)
  const bool usable = batch <= std::max(2, persist_max_clips_) && persistent_usable();
  if (usable && batch >= 2 && batch <= persist_max_clips_) {
    int mn[3] = {max_new, -1, -1};
    for (int b = 0; b < batch; ++b) mn[b] = (max_new_clip && max_new_clip[b] > 0) ? std::min(max_new, max_new_clip[b]) : max_new;
    const int st = run_persistent(mn[0], nullptr, 0, nullptr, nullptr, 0, mn[1], mn[2]);
    if (st >= 0) { persistent_succeeded(); return st; }
    persistent_gave_up();
  } else if (usable && batch <= 2) {
    int steps = 0, b = 0;
    for (; b < batch; ++b) {
      int mn = max_new;
      if (max_new_clip && max_new_clip[b] > 0) mn = std::min(mn, max_new_clip[b]);
      const int st = run_persistent(mn, nullptr, 0, nullptr, nullptr, b);
      if (st < 0) break;
      steps = std::max(steps, st);
    }
    if (b == batch) { persistent_succeeded(); return steps; }
    persistent_gave_up();  // these utterances (and the next few) take the launch-per-phase path
  }
  hipGraphExec_t g = step_graph(batch, max_new);  // (a fresh multi-branch graph is probed with replays: before the state is set)
  reset_decode_state(batch, max_new_clip);
  hipStream_t s = stream();
  const int total = std::min(Tc, 4 + max_new);
  const int kPoll = 8;  // steps between done-counter polls; at most 2*kPoll steps run past the last eot
  hipEvent_t pe[2] = {ev_[3], ev_[4]};
  int steps = 0, polls = 0;
  for (int st = 0; st < total; ++st) {
    HIP_CHECK(hipGraphLaunch(g, s));
    ++steps;
    if ((st + 1) % kPoll == 0 && st >= 4) {
      if (polls >= 1) {  // look at the poll issued kPoll steps ago (keeps the queue full)
        HIP_CHECK(hipEventSynchronize(pe[(polls - 1) & 1]));
        if (h_poll_[(polls - 1) & 1] >= batch) break;
      }
      HIP_CHECK(hipMemcpyAsync(&h_poll_[polls & 1], &d_state_->n_done, 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipEventRecord(pe[polls & 1], s));
      ++polls;
    }
  }
  return steps;
}

// max_new1 >= 0 (max_new2 >= 0): TWO (THREE) clips in this launch — slots `slot`, `slot + 1` (, `slot + 2`), budgets max_new / max_new1
// (/ max_new2) (greedy decode only)
int Engine::run_persistent(int max_new, const int* d_forced, int n_forced, float* d_logits, int* d_argmax, int slot, int max_new1, int max_new2) {
  // The launch needs every workgroup resident at once (one per CU): two of them in flight on one GPU could each hold
  // part of the CUs and starve the other until both give up. Handles of one process on one device take turns.
  // (one mutex per device, shared by the bfloat16 and the half build of this file: iengine.hpp)
  std::lock_guard<std::mutex> launch_lock(persistent_launch_mutex(device_));
  hipStream_t s = stream();
  const int Tc = cfg_.n_text_ctx, H = cfg_.n_text_head;
  PersistParams p{};
  p.wl = dec_w_arena_; p.fl = dec_f_arena_;
  p.qf = d_qfold_;
  p.tok_emb = tok_emb_; p.pos = dec_pos_; p.ln_w = dec_ln_w_; p.ln_b = dec_ln_b_;
  p.cross_k = d_cross_k_ + (size_t)slot * H * t_pad_ * 64;  // this clip's slot, layer 0
  p.cross_v = d_cross_v_ + (size_t)slot * H * t_pad_ * 64;
  p.cross_layer_stride = (long)cap_ * H * t_pad_ * 64;
  p.n_layer = cfg_.n_text_layer; p.n_vocab = cfg_.n_vocab; p.n_ctx = Tc; p.n_audio_ctx = cfg_.n_audio_ctx;
  p.eot = cfg_.eot; p.max_new = max_new;
  p.total_steps = d_forced || d_logits || d_argmax ? 4 + n_forced : std::min(Tc, 4 + std::max(max_new, std::max(max_new1, max_new2)));
  p.n_clip = 1;
  if (max_new1 >= 0) {
    p.n_clip = max_new2 >= 0 ? 3 : 2;
    if (persist_max_clips_ < p.n_clip || d_forced || d_logits || d_argmax || slot + p.n_clip > cap_) throw std::runtime_error("run_persistent: that many clips are unsupported here");
    p.cross_clip_stride = (long)H * t_pad_ * 64;
    p.self_k1 = d_self_k1_; p.self_v1 = d_self_v1_;
    p.gran_clip_u64 = (long)(gran_bytes_ / 8);
    p.out_ids1 = d_out_ids_ + (size_t)(slot + 1) * Tc; p.n_out1 = d_nout_ + slot + 1; p.max_new1 = max_new1;
    p.max_new2 = max_new2; p.self_clip_stride = (long)(self1_bytes_ / 2);
  }
  p.sot = d_sot_;
  p.forced = d_forced; p.n_forced = n_forced; p.logits_dump = d_logits; p.argmax_dump = d_argmax;
  p.gran = d_gran_;
  p.gran_bytes = (int)gran_bytes_;
  p.err = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(d_gran_) + gran_bytes_ - 8);
  p.out_ids = d_out_ids_ + (size_t)slot * Tc; p.n_out = d_nout_ + slot; p.state = d_state_;
  long long* d_prof = nullptr;
  const char* prof_path = getenv("AX_WHISPER_PERSIST_PROF");  // debugging aid: per-workgroup, per-phase time of the launch
  if (prof_path) {
    HIP_CHECK(hipMalloc((void**)&d_prof, (size_t)persist_grid_ * 64 * 8));
    HIP_CHECK(hipMemset(d_prof, 0, (size_t)persist_grid_ * 64 * 8));
  }
  p.prof = d_prof;
  { const char* pc = getenv("AX_WHISPER_PERSIST_PROF_CLIP"); p.prof_clip = pc && pc[0] == '1'; }
  p.fault = getenv("AX_WHISPER_PERSIST_FAULT") ? 1 : 0;
  HIP_CHECK(hipMemsetAsync(d_gran_, 0, p.n_clip * gran_bytes_, s));
  HIP_CHECK(hipMemsetAsync(d_state_, 0, sizeof(DecState), s));
  HIP_CHECK(hipMemsetAsync(d_nout_ + slot, 0, 4 * p.n_clip, s));
  if (p.n_clip >= 2) {  // keys beyond a clip's position are masked, but their values must be finite
    HIP_CHECK(hipMemsetAsync(d_self_k1_, 0, (p.n_clip - 1) * self1_bytes_, s));
    HIP_CHECK(hipMemsetAsync(d_self_v1_, 0, (p.n_clip - 1) * self1_bytes_, s));
  }
  HIP_CHECK(launch_decode_persistent(p, cfg_.n_text_state, persist_grid_, s));
  HIP_CHECK(hipMemcpyAsync(&h_poll_[8], p.err, 4, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipMemcpyAsync(&h_poll_[9], &d_state_->step, 4, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  if (d_prof) {
    std::vector<long long> hp((size_t)persist_grid_ * 64);
    HIP_CHECK(hipMemcpy(hp.data(), d_prof, hp.size() * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_prof);
    if (FILE* f = fopen(prof_path, "w")) {
      fprintf(f, "# steps %d grid %d; rows = workgroups, columns 0-31 = phase tick sums, 32-63 = absolute ticks of one layer (100 MHz)\n", h_poll_[9], persist_grid_);
      for (int g = 0; g < persist_grid_; ++g) {
        for (int i = 0; i < 64; ++i) fprintf(f, "%lld ", hp[(size_t)g * 64 + i]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
  if (h_poll_[8] != 0) {
    fprintf(stderr, "[ax_whisper] persistent decode gave up (code 0x%x); falling back to the launch-per-phase path\n", (unsigned)h_poll_[8]);
    return -1;
  }
  return h_poll_[9];
}

void Engine::fetch_ids(int batch, int32_t* ids, int* n_ids) {
  hipStream_t s = stream();
  HIP_CHECK(hipMemcpyAsync(ids, d_out_ids_, (size_t)batch * cfg_.n_text_ctx * 4, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipMemcpyAsync(n_ids, d_nout_, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
}

}  // inline namespace AXW_NS
}  // namespace axw
