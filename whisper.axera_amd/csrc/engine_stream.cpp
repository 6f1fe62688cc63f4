// engine_stream.cpp — axw::Engine: utterance slots refilled while the others decode (AX_WHISPER_Stream*) and the bench hooks.
#include "engine_impl.hpp"

namespace axw {
inline namespace AXW_NS {

// ------------------------------------------------------------------------------ slot refill (continuous batching)
// The reference stops every utterance at its own eot (Whisper.cpp:219-222) and serves requests one by one
// (WhisperHTTPServer.hpp:37-100). With per-slot offsets (common.hpp: DecState) a slot whose clip has finished takes the
// next clip while the other slots decode on; the step graph is the one the batched loop replays.
void Engine::require_no_stream(const char* what) const {
  if (stream_slots_ > 0) throw std::runtime_error(std::string(what) + ": a slot stream is open on this handle (AX_WHISPER_StreamClose first)");
}

__global__ static void slot_reset_kernel(int slot, int max_new, const int* sot, int* off, int* tok, int* done, int* n_out, int* max_new_clip,
                                         const h16* tok_emb, const float* pos, float* x, int d) {
  const int t = sot[0];
  if (threadIdx.x == 0) { off[slot] = 0; tok[slot] = t; n_out[slot] = 0; max_new_clip[slot] = max_new; done[slot] = 0; }
  for (int c = threadIdx.x; c < d; c += blockDim.x) x[(long)slot * d + c] = (float)tok_emb[(long)t * d + c] + pos[c];  // position 0
}

void Engine::stream_open(int n_slots) {
  std::lock_guard<std::recursive_mutex> capture_lock(device_capture_mutex(device_));
  HIP_CHECK(hipSetDevice(device_));
  if (n_slots < 1) throw std::runtime_error("stream_open: n_slots must be >= 1");
  if (user_stream_) throw std::runtime_error("stream_open: not with a caller-supplied stream (AX_WHISPER_SetStream)");
  stream_close();
  ensure_capacity(std::max(n_slots, 3));
  const int n = std::max(n_slots, 3);  // the step sequence of 3+ slots handles any mix of idle and active slots
  hipStream_t s = stream();
  (void)step_graph(n, cfg_.n_text_ctx - 4);  // captured here, outside the serving loop (and probed with replays: before the state is set)
  reset_decode_state(n);
  std::vector<int> ones(n, 1);         // every slot idle: its attention launches return at once
  HIP_CHECK(hipMemcpy(d_done_, ones.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  if (h_admit_ring_) { (void)hipHostFree(h_admit_ring_); h_admit_ring_ = nullptr; }
  HIP_CHECK(hipHostMalloc((void**)&h_admit_ring_, (size_t)kAdmitRing * 2 * cap_ * 4, hipHostMallocDefault));
  admit_seq_ = 0;
  while ((int)ev_admit_.size() < n) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ev_admit_.push_back(e);
  }
  HIP_CHECK(hipStreamSynchronize(s));
  slot_state_.assign(n, kIdle);
  slot_max_new_.assign(n, 0);
  memset(h_done_live_, 0, (size_t)cap_ * 4);
  step_seq_ = 0;
  stream_slots_ = n;
  stream_user_slots_ = n_slots;
  cfg_.ints["stream_slots"] = n_slots;
}

void Engine::stream_close() {
  if (stream_slots_ == 0) return;
  (void)hipStreamSynchronize(admit_stream_);
  (void)hipStreamSynchronize(stream());
  stream_slots_ = 0;
  stream_user_slots_ = 0;
  slot_state_.clear();
  cfg_.ints["stream_slots"] = 0;
}

void Engine::stream_admit(const int* slots, const float* const* pcm, const int* n_samples, const int* max_new, int count) {
  HIP_CHECK(hipSetDevice(device_));
  if (stream_slots_ == 0) throw std::runtime_error("stream_admit: no stream open");
  if (count < 1 || count > stream_user_slots_) throw std::runtime_error("stream_admit: count out of range");
  for (int i = 0; i < count; ++i) {
    // the slots the caller opened, not the 3 the step graph is rounded up to: finished_slots of StreamStep is [n_slots]
    if (slots[i] < 0 || slots[i] >= stream_user_slots_) throw std::runtime_error("stream_admit: slot out of range");
    if (slot_state_[slots[i]] != kIdle) throw std::runtime_error("stream_admit: slot " + std::to_string(slots[i]) + " is busy");
    for (int j = 0; j < i; ++j) if (slots[j] == slots[i]) throw std::runtime_error("stream_admit: a slot is listed twice");
    if (n_samples[i] < 1) throw std::runtime_error("empty audio clip");
  }
  const int Tc = cfg_.n_text_ctx;
  // front-end + encoder of these clips as ONE batched pass on the admission stream (encoder scratch of clip indices
  // 0..count-1; the decode step touches none of it), cross K/V scattered straight into the slots, which stay idle — their
  // attention launches skip them — until stream_step has seen the event
  struct StreamSwap {  // run_frontend / run_encoder enqueue on stream(): point it at the admission stream for this call
    hipStream_t& u; hipStream_t keep;
    StreamSwap(hipStream_t& us, hipStream_t to) : u(us), keep(us) { u = to; }
    ~StreamSwap() { u = keep; }
  } swap(user_stream_, admit_stream_);
  // Nothing below waits for an earlier pass's ENCODER: the ring entry of this pass was last used kAdmitRing passes ago, the
  // PCM staging rows by the pass before (its uploads are the first thing it enqueued)
  const int ring = (int)(admit_seq_ % kAdmitRing);
  if (admit_seq_ >= kAdmitRing) HIP_CHECK(hipEventSynchronize(ev_ring_[ring]));
  if (admit_seq_ > 0) HIP_CHECK(hipEventSynchronize(ev_upload_));
  int* h_ns = h_admit_ring_ + (size_t)ring * 2 * cap_;
  int* h_map = h_ns + cap_;
  memcpy(h_map, slots, (size_t)count * 4);
  HIP_CHECK(hipMemcpyAsync(d_slot_map_, h_map, (size_t)count * 4, hipMemcpyHostToDevice, admit_stream_));
  upload_pcm(pcm, n_samples, count);
  HIP_CHECK(hipEventRecord(ev_upload_, admit_stream_));
  run_frontend(d_pcm_, (int)pcm_stride_, n_samples, count, false, true, h_ns);
  run_encoder(count, d_slot_map_);
  HIP_CHECK(hipEventRecord(ev_ring_[ring], admit_stream_));
  ++admit_seq_;
  for (int i = 0; i < count; ++i) {
    HIP_CHECK(hipEventRecord(ev_admit_[slots[i]], admit_stream_));
    h_done_live_[slots[i]] = 0;
    slot_state_[slots[i]] = kEncoding;
    const int mn = max_new ? max_new[i] : 0;
    slot_max_new_[slots[i]] = (mn > 0 && mn < Tc - 4) ? mn : Tc - 4;
  }
}

// Up to n decoder steps. Between two steps the host looks at the host-mapped done flags (advance_kernel raises a clip's flag,
// behind a system-scope fence, the moment its ids are final): a finished slot is seen without a copy or a wait, and a slot
// whose encoder has finished joins before the next step. (Measured and not kept: extra slots holding already-encoded clips
// that take over the moment a decoding slot frees — the step then runs its linear layers over more rows and its attention
// launches over more workgroups, and that costs more than the refill latency it removes: 32 + 8 slots 224 -> 202 clips/s.)
int Engine::stream_step(int n_steps, int* finished_slots) {
  HIP_CHECK(hipSetDevice(device_));
  if (stream_slots_ == 0) throw std::runtime_error("stream_step: no stream open");
  hipStream_t s = stream();
  const int n = stream_slots_;
  auto n_in = [&](int st) { int c = 0; for (int i = 0; i < n; ++i) c += slot_state_[i] == st; return c; };
  auto harvest = [&] {
    for (int i = 0; i < n; ++i)
      if (slot_state_[i] == kActive && __atomic_load_n(&h_done_live_[i], __ATOMIC_ACQUIRE)) slot_state_[i] = kFinished;
  };
  // slots whose encoder has finished join; if nothing decodes the loop waits for the first encoder
  auto activate_ready = [&] {
    int active = n_in(kActive);
    for (int i = 0; i < n; ++i) {
      if (slot_state_[i] != kEncoding) continue;
      hipError_t q = hipEventQuery(ev_admit_[i]);
      if (q == hipErrorNotReady && active == 0) { HIP_CHECK(hipEventSynchronize(ev_admit_[i])); q = hipSuccess; }
      if (q == hipErrorNotReady) continue;
      HIP_CHECK(q);
      hipLaunchKernelGGL(slot_reset_kernel, dim3(1), dim3(256), 0, s, i, slot_max_new_[i], d_sot_, d_off_, d_tok_, d_done_, d_nout_,
                         d_max_new_clip_, tok_emb_, dec_pos_, d_xdec_, cfg_.n_text_state);
      slot_state_[i] = kActive;
      ++active;
    }
  };
  hipGraphExec_t g = step_graph(n, cfg_.n_text_ctx - 4);
  // The host runs two steps ahead of the device (it waits for step k-2 before it enqueues step k): the queue never runs dry,
  // and what the host sees in the flags is at most two steps old, so a waiting clip takes a freed slot within two steps.
  for (int st = 0; st < std::max(1, n_steps); ++st) {
    if (step_seq_ >= 2) HIP_CHECK(hipEventSynchronize(ev_step_[(step_seq_ - 2) % 3]));
    harvest();
    activate_ready();
    if (n_in(kActive) == 0) break;  // nothing decodes and nothing is ready: a step would be the GEMM chain for nobody
    HIP_CHECK(hipGraphLaunch(g, s));
    HIP_CHECK(hipEventRecord(ev_step_[step_seq_ % 3], s));
    ++step_seq_;
  }
  harvest();
  int n_fin = 0;
  for (int i = 0; i < n; ++i)
    if (slot_state_[i] == kFinished) finished_slots[n_fin++] = i;
  return n_fin;
}

void Engine::stream_collect(int slot, int32_t* ids, int* n_ids) {
  HIP_CHECK(hipSetDevice(device_));
  if (stream_slots_ == 0) throw std::runtime_error("stream_collect: no stream open");
  if (slot < 0 || slot >= stream_user_slots_ || slot_state_[slot] != kFinished) throw std::runtime_error("stream_collect: slot has not finished");
  // on its own stream: the slot's ids are final (its done flag was seen), the decoder steps queued meanwhile do not touch them
  HIP_CHECK(hipMemcpyAsync(ids, d_out_ids_ + (size_t)slot * cfg_.n_text_ctx, (size_t)cfg_.n_text_ctx * 4, hipMemcpyDeviceToHost, copy_stream_));
  HIP_CHECK(hipMemcpyAsync(n_ids, d_nout_ + slot, 4, hipMemcpyDeviceToHost, copy_stream_));
  HIP_CHECK(hipStreamSynchronize(copy_stream_));
  slot_state_[slot] = kIdle;
}

// ------------------------------------------------------------------------------ stored 16-bit tensors: non-finite scan
// (parity battery under trained-model statistics: outlier channels, FFN hidden values in the thousands — a half tensor
// that overflowed would show here even where the logits still look plausible)
__global__ static void scan16_kernel(const h16* __restrict__ p, size_t n, unsigned long long* bad, unsigned* maxbits) {
  unsigned long long nb = 0;
  float mx = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = (float)p[i];
    if (v != v || fabsf(v) > 3.0e38f) ++nb; else mx = fmaxf(mx, fabsf(v));
  }
  if (nb) atomicAdd(bad, nb);
  atomicMax(maxbits, __float_as_uint(mx));  // non-negative floats order like their bit patterns
}

int Engine::scan_stored16(int batch, int n_max, char (*names)[32], long long* nonfinite, float* maxabs) {
  require_no_stream("scan_stored16");
  std::lock_guard<std::recursive_mutex> capture_lock(device_capture_mutex(device_));
  HIP_CHECK(hipSetDevice(device_));
  if (batch < 1 || batch > cap_) throw std::runtime_error("scan_stored16: batch outside the allocated slots");
  const size_t B = batch, d = cfg_.n_text_state, T = cfg_.n_audio_ctx, L = cfg_.n_text_layer, H = cfg_.n_text_head, Tc = cfg_.n_text_ctx;
  struct Buf { const char* name; const h16* p; size_t n; };
  std::vector<Buf> bufs = {
      {"enc.mel", d_mel_tm_, B * mel_rows_ * cfg_.n_mels}, {"enc.conv1", d_h1_, B * h1_rows_ * d}, {"enc.ln", d_ln_, B * T * d},
      {"enc.q", d_q_, B * T * d}, {"enc.k", d_k_, B * T * d}, {"enc.vt", d_vt_, B * d * t_pad_}, {"enc.attn", d_attn_, B * T * d},
      {"enc.ffn_hidden", d_ffn_, B * T * 4 * d},
      {"cross_k", d_cross_k_, L * (size_t)cap_ * H * t_pad_ * 64}, {"cross_v", d_cross_v_, L * (size_t)cap_ * H * t_pad_ * 64},
      {"self_k", d_self_k_, L * (size_t)cap_ * H * Tc * 64}, {"self_v", d_self_v_, L * (size_t)cap_ * H * Tc * 64},
      {"dec.act_hi", d_act_[0], (size_t)nbs_ * 16 * d}, {"dec.act_lo", d_act_[1], (size_t)nbs_ * 16 * d},
      {"dec.att_hi", d_att_[0], (size_t)nbs_ * 16 * d}, {"dec.att_lo", d_att_[1], (size_t)nbs_ * 16 * d},
      {"dec.hid_hi", d_hidp_[0], (size_t)nbs_ * 16 * 4 * d}, {"dec.hid_lo", d_hidp_[1], (size_t)nbs_ * 16 * 4 * d},
  };
  if (d_self_k1_) {
    bufs.push_back({"persist.self_k1", d_self_k1_, self1_bytes_ / 2 * (size_t)std::max(persist_max_clips_ - 1, 1)});
    bufs.push_back({"persist.self_v1", d_self_v1_, self1_bytes_ / 2 * (size_t)std::max(persist_max_clips_ - 1, 1)});
  }
  const int n = std::min<int>(n_max, (int)bufs.size());
  unsigned long long* d_res = nullptr;
  HIP_CHECK(hipMalloc((void**)&d_res, (size_t)n * 16));
  HIP_CHECK(hipMemset(d_res, 0, (size_t)n * 16));
  hipStream_t s = stream();
  for (int i = 0; i < n; ++i)
    scan16_kernel<<<1024, 256, 0, s>>>(bufs[i].p, bufs[i].n, d_res + 2 * i, reinterpret_cast<unsigned*>(d_res + 2 * i + 1));
  std::vector<unsigned long long> h((size_t)n * 2);
  HIP_CHECK(hipMemcpyAsync(h.data(), d_res, (size_t)n * 16, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  (void)hipFree(d_res);
  for (int i = 0; i < n; ++i) {
    snprintf(names[i], 32, "%s", bufs[i].name);
    nonfinite[i] = (long long)h[2 * i];
    const unsigned bits = (unsigned)(h[2 * i + 1] & 0xffffffffu);
    memcpy(&maxabs[i], &bits, 4);
  }
  return n;
}

// ------------------------------------------------------------------------------ which of the engine's streams run side by side?
// The runtime multiplexes a process's HIP streams onto a few hardware queues (four by default); two streams on one queue
// execute one after the other. A spinner kernel on stream A and a time stamp on stream B, issued right behind it: if the stamp is
// taken before the spinner ends, A and B are on different queues.
__global__ static void queue_probe_spin(unsigned long long* out, long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  out[0] = (unsigned long long)t0;
  out[1] = (unsigned long long)wall_clock64();
}
__global__ static void queue_probe_mark(unsigned long long* out) { out[0] = (unsigned long long)wall_clock64(); }

static bool streams_concurrent(hipStream_t a, hipStream_t b, unsigned long long* d_buf /*[4]*/) {
  unsigned long long h[4] = {0, 0, 0, 0};
  HIP_CHECK(hipStreamSynchronize(a));
  HIP_CHECK(hipStreamSynchronize(b));
  HIP_CHECK(hipMemset(d_buf, 0, 32));
  queue_probe_spin<<<1, 64, 0, a>>>(d_buf, 30000);  // 300 us on the 100 MHz wall clock
  queue_probe_mark<<<1, 64, 0, b>>>(d_buf + 2);
  HIP_CHECK(hipStreamSynchronize(a));
  HIP_CHECK(hipStreamSynchronize(b));
  HIP_CHECK(hipMemcpy(h, d_buf, 32, hipMemcpyDeviceToHost));
  return h[2] != 0 && h[2] + 5000 < h[1];  // stamped at least 50 us before the spinner ended
}

// hipGraphInstantiate gives every parallel branch of a captured step a stream of its own, and the runtime deals a process's
// streams onto its hardware queues (four by default) round-robin: which queue the branch lands on depends on how many streams the
// process has created before. Measured with 0..5 unused pad streams created ahead (profiles/r05_stream_queue_root_cause.txt,
// stream64, period 4): 335 / 367 / 341 / 345 clips/s — the "bimodality" of rounds 3-4 was this draw (plain step replays do not care:
// 1.130-1.142 ms; the slot stream does, because admission passes and result copies run beside the steps).
// The one good place is the queue of branch_stream_[0], which only ever carries the capture and never a replay. So a freshly
// instantiated multi-branch step is PROBED — a 3 ms spinner on branch_stream_[0], then one replay: if a branch shares that queue the
// replay takes > 3 ms instead of ~1 — and re-instantiated behind one more pad stream until it does (at most 4 times, ~5 ms each, once
// per captured graph). The replays run on whatever the decode state holds: callers reset the state AFTER they have the graph.
bool Engine::graph_branch_shares_queue(hipGraphExec_t exec, hipStream_t other) {
  hipStream_t s = stream();
  // everything the probe owns is released on every path out of it (a HIP_CHECK throw included)
  struct Probe {
    unsigned long long* d_buf = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Probe() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (d_buf) (void)hipFree(d_buf);
    }
  } pr;
  HIP_CHECK(hipMalloc((void**)&pr.d_buf, 32));
  HIP_CHECK(hipEventCreate(&pr.e0));
  HIP_CHECK(hipEventCreate(&pr.e1));
  HIP_CHECK(hipStreamSynchronize(s));
  HIP_CHECK(hipStreamSynchronize(other));
  HIP_CHECK(hipGraphLaunch(exec, s));  // first launch of a fresh exec: paid here, not inside the measurement
  HIP_CHECK(hipStreamSynchronize(s));
  auto replay_ms = [&](bool with_spinner) {
    if (with_spinner) queue_probe_spin<<<1, 64, 0, other>>>(pr.d_buf, 300000);  // 3 ms
    HIP_CHECK(hipEventRecord(pr.e0, s));
    HIP_CHECK(hipGraphLaunch(exec, s));
    HIP_CHECK(hipEventRecord(pr.e1, s));
    HIP_CHECK(hipEventSynchronize(pr.e1));
    HIP_CHECK(hipStreamSynchronize(other));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, pr.e0, pr.e1));
    return ms;
  };
  // a replay of many clips (or on a device another handle keeps busy) takes milliseconds by itself: the spinner shows as
  // ~3 ms ON TOP of the unloaded replay when a branch waits behind it, and as nothing when it does not
  const float base = replay_ms(false);
  const float loaded = replay_ms(true);
  return loaded > base + 2.0f;
}

float Engine::bench(const std::string& what, int batch, int arg, int iters) {
  require_no_stream("bench");
  if (what == "queue_probe") {
    // bit i set: pair i runs side by side. Pairs: 0 main-admission, 1 main-branch0, 2 main-copies, 3 admission-branch0,
    // 4 admission-copies, 5 branch0-copies (diagnostic of the slot stream's process-to-process bimodality)
    HIP_CHECK(hipSetDevice(device_));
    std::lock_guard<std::recursive_mutex> capture_lock(device_capture_mutex(device_));
    unsigned long long* d_buf = nullptr;
    HIP_CHECK(hipMalloc((void**)&d_buf, 32));
    hipStream_t st[4] = {own_stream_, admit_stream_, branch_stream_[0], copy_stream_};
    const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    int mask = 0;
    for (int i = 0; i < 6; ++i)
      if (streams_concurrent(st[pa[i]], st[pb[i]], d_buf)) mask |= 1 << i;
    (void)hipFree(d_buf);
    return (float)mask;
  }
  HIP_CHECK(hipSetDevice(device_));
  ensure_capacity(batch);
  hipStream_t s = stream();
  hipEvent_t a, b;
  HIP_CHECK(hipEventCreate(&a));
  HIP_CHECK(hipEventCreate(&b));
  float ms = 0.f;
  if (what == "decode_step" || what == "decode_gemv" || what == "decode_attn") {
    // decode_gemv / decode_attn: the same captured step with only the GEMV / only the attention launches
    step_mask_ = what == "decode_step" ? 15 : (what == "decode_gemv" ? 1 : 2);
    struct Restore { int& m; ~Restore() { m = 15; } } restore{step_mask_};
    const int Tc = cfg_.n_text_ctx;
    reset_decode_state(batch);
    hipGraphExec_t g = step_graph(batch, Tc - 4);
    arg = std::max(0, std::min(arg, Tc - 1 - iters));
    DecState st{arg, 0, 0, 0};
    std::vector<int> offs(batch, arg);  // every slot at position `arg`
    HIP_CHECK(hipMemcpy(d_state_, &st, sizeof(st), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_off_, offs.data(), (size_t)batch * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipGraphLaunch(g, s));  // warm
    st.step = arg;
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipMemcpy(d_state_, &st, sizeof(st), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_off_, offs.data(), (size_t)batch * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) HIP_CHECK(hipGraphLaunch(g, s));
    HIP_CHECK(hipEventRecord(b, s));
  } else if (what == "attn_stamp") {
    // One replay of the production step graph (all launches, every branch) at decode offset `arg` whose decode_attention
    // launches stamp their own {first workgroup start, last workgroup end}; the table goes to $AX_WHISPER_ATTN_STAMP
    // (default attn_stamps.csv). Returns the length of the UNION of the attention intervals in ms: K/V bytes of the step
    // over that time is the rate the attention launches achieve while the other branch's launches run beside them.
    if (batch <= gemv_max_) throw std::runtime_error("bench attn_stamp: the batched decode sequences only (3+ clips)");
    // (a launch has batch * heads workgroups, or up to 640 when few (clip, head) pairs are split along the keys)
    if (std::max<long>((long)batch * cfg_.n_text_head, 640) > (long)kStampWgs) throw std::runtime_error("bench attn_stamp: too many workgroups per launch");
    if (!d_stamp_) {
      std::lock_guard<std::recursive_mutex> capture_lock(device_capture_mutex(device_));  // an allocation (iengine.hpp)
      d_stamp_ = (unsigned long long*)dalloc((size_t)2 * kStampWgs * kStampLaunches * 8, true);
      allocs_.push_back(d_stamp_);
    }
    step_mask_ = 15 | 16;
    struct Restore { int& m; ~Restore() { m = 15; } } restore{step_mask_};
    const int Tc = cfg_.n_text_ctx;
    reset_decode_state(batch);
    const long key = ((long)batch * 1024 + (Tc - 4)) * 32 + step_mask_;
    auto old = graphs_.find(key);
    if (old != graphs_.end()) { (void)hipGraphExecDestroy(old->second); graphs_.erase(old); }
    stamp_meta_.clear();
    hipGraphExec_t g = step_graph(batch, Tc - 4);
    const int warm_replays = iters >= 100 ? iters - 100 : 0;
    arg = std::max(0, std::min(arg, Tc - 4 - warm_replays));  // every replay advances the clips by one position
    DecState st{arg, 0, 0, 0};
    std::vector<int> offs(batch, arg);
    const size_t n_words = (size_t)2 * kStampWgs * kStampLaunches;
    std::vector<unsigned long long> raw(n_words), got(2 * kStampLaunches);
    std::vector<std::pair<double, double>> iv;
    double best_union = 0.0;
    std::string table;
    for (int rep = 0; rep < 3; ++rep) {  // the first repetitions warm the caches; the last one is reported
      HIP_CHECK(hipMemcpy(d_state_, &st, sizeof(st), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(d_off_, offs.data(), (size_t)batch * 4, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemset(d_stamp_, 0, n_words * 8));
      HIP_CHECK(hipDeviceSynchronize());
      HIP_CHECK(hipEventRecord(a, s));
      // arg2 (iters >= 100): `iters - 100` replays back to back BEFORE the stamped one, so that the stamped step starts the way
      // a step of the loop does — behind its predecessor, both branches already queued (a lone replay's second branch starts
      // ~250 us late: the host is still enqueuing its nodes)
      for (int k = 0; k < warm_replays; ++k) HIP_CHECK(hipGraphLaunch(g, s));
      HIP_CHECK(hipGraphLaunch(g, s));
      HIP_CHECK(hipEventRecord(b, s));
      HIP_CHECK(hipStreamSynchronize(s));
      HIP_CHECK(hipMemcpy(raw.data(), d_stamp_, n_words * 8, hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < stamp_meta_.size(); ++i) {  // a launch = the earliest start and the latest end of its workgroups
      unsigned long long lo = ~0ull, hi = 0ull;
      for (size_t w = 0; w < kStampWgs; ++w) {
        const unsigned long long bg = raw[(i * kStampWgs + w) * 2], en = raw[(i * kStampWgs + w) * 2 + 1];
        if (bg) lo = std::min(lo, bg);
        hi = std::max(hi, en);
      }
      got[2 * i] = lo;
      got[2 * i + 1] = hi;
    }
    float step_ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&step_ms, a, b));
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < stamp_meta_.size(); ++i) t0 = std::min(t0, got[2 * i]);
    const double keys_self = arg + 1, d_ = cfg_.n_text_state;
    char line[256];
    snprintf(line, sizeof line, "# batch %d, decode offset %d, %zu attention launches, %d replays back to back before the stamped one (all %d: %.3f us, hipEvents); times in us from the stamped step's first attention start (100 MHz wall clock)\n",
             batch, arg, stamp_meta_.size(), warm_replays, warm_replays + 1, step_ms * 1e3);
    table += line;
    table += "launch,kind,layer,first_clip,clips,begin_us,end_us,duration_us,kv_bytes,GBs\n";
    for (size_t i = 0; i < stamp_meta_.size(); ++i) {
      const StampMeta& m = stamp_meta_[i];
      const double bg = (double)(got[2 * i] - t0) * 0.01, en = (double)(got[2 * i + 1] - t0) * 0.01;
      static const char* const kinds[] = {"self", "cross", "qkv", "o", "co", "fc1", "fc2", "cq"};
      const bool is_attn = m.cross <= 1;
      const double bytes = is_attn ? (double)m.nb * 2.0 * 2.0 * d_ * (m.cross ? (double)cfg_.n_audio_ctx : keys_self) : 0.0;
      if (is_attn) iv.push_back({bg, en});
      snprintf(line, sizeof line, "%zu,%s,%d,%d,%d,%.2f,%.2f,%.2f,%.0f,%.1f\n", i, kinds[m.cross & 7], m.layer, m.b0, m.nb, bg, en, en - bg, bytes,
               en > bg ? bytes / ((en - bg) * 1e-6) / 1e9 : 0.0);
      table += line;
    }
    std::sort(iv.begin(), iv.end());
    double cur_b = -1, cur_e = -1;
    for (auto& x : iv) {
      if (x.first > cur_e) { best_union += cur_e - cur_b; cur_b = x.first; cur_e = x.second; }
      else cur_e = std::max(cur_e, x.second);
    }
    best_union += cur_e - cur_b;
    snprintf(line, sizeof line, "# union of the attention intervals: %.2f us\n", best_union);
    table += line;
    const char* path = getenv("AX_WHISPER_ATTN_STAMP");
    if (FILE* f = fopen(path ? path : "attn_stamps.csv", "w")) { fputs(table.c_str(), f); fclose(f); }
    { auto it = graphs_.find(key); if (it != graphs_.end()) { (void)hipGraphExecDestroy(it->second); graphs_.erase(it); } }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return (float)(best_union * 1e-3);
  } else if (what == "encoder") {
    run_encoder(batch);
    HIP_CHECK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) run_encoder(batch);
    HIP_CHECK(hipEventRecord(b, s));
  } else if (what == "frontend") {
    std::vector<int> ns(batch, 480000);
    run_frontend(d_pcm_, (int)pcm_stride_, ns.data(), batch, false);
    HIP_CHECK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) run_frontend(d_pcm_, (int)pcm_stride_, ns.data(), batch, false);
    HIP_CHECK(hipEventRecord(b, s));
  } else {
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    throw std::runtime_error("bench: unknown target '" + what + "'");
  }
  HIP_CHECK(hipEventSynchronize(b));
  HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return ms;
}

}  // inline namespace AXW_NS
}  // namespace axw
