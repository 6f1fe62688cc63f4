// engine.cpp — host side of the MI355X Whisper engine (see engine.hpp for what it replaces): construction, weights, slot buffers,
// front-end, encoder and the entry points. The decode paths: engine_decode.cpp; utterance slots and bench hooks: engine_stream.cpp.
#include "engine_impl.hpp"

#include "host_io.hpp"

namespace axw {
inline namespace AXW_NS {

static int dtype_code(const std::string& d) { return d == "F32" ? 0 : d == "BF16" ? 1 : 2; }

// host: the bits of one stored h16 value -> float (bfloat16: the upper half of the fp32 pattern; half: IEEE binary16)
static float h16_bits_to_float(uint16_t bits) {
#if AXW_F16
  _Float16 h;
  memcpy(&h, &bits, 2);
  return (float)h;
#else
  const uint32_t u = (uint32_t)bits << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

// ------------------------------------------------------------------------------ construction
Engine::Engine(const std::string& model_type, const std::string& model_path, const std::string& language, int device,
               int max_batch) {
  try {
    construct(model_type, model_path, language, device, max_batch);
  } catch (...) {
    // the destructor of a partially constructed object never runs: a failed Init (missing / corrupt weights, shape
    // mismatch, out of memory) must give back the stream, the events, the pinned buffers and every weight already uploaded
    destroy();
    throw;
  }
}

void Engine::construct(const std::string& model_type, const std::string& model_path, const std::string& language, int device,
                       int max_batch) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
    throw std::runtime_error("no HIP device visible: the MI355X engine has no CPU fallback");
  if (device < 0) {
    const char* e = getenv("AX_WHISPER_DEVICE");
    device = e ? atoi(e) : 0;
  }
  if (device >= n_dev) throw std::runtime_error("HIP device ordinal out of range");
  device_ = device;
  // Host-side file work first, OUTSIDE the allocation / capture mutex: configuration, vocabulary, the t2s dictionaries, and the
  // weights file paged in. Creating a handle at run time (another model or dtype) then holds the mutex only for device work —
  // allocations, uploads from memory, conversions — and not for seconds of disk reads, during which every serving handle's
  // StreamOpen / capacity growth / first graph capture would have waited (iengine.hpp).
  const std::string dir = model_path + "/" + model_type;
  load_config(dir, model_type, language);
  tokens_ = load_token_table(dir + "/" + model_type + "-tokens.txt");
  load_t2s(model_path);
  SafeTensors weights_file(dir + "/" + model_type + ".safetensors");
  (void)weights_file.page_in();
  std::lock_guard<std::recursive_mutex> capture_lock(device_capture_mutex(device_));  // allocations + synchronous copies (iengine.hpp)
  HIP_CHECK(hipSetDevice(device_));
  device_set_ = true;
  HIP_CHECK(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
  for (auto& e : ev_) HIP_CHECK(hipEventCreate(&e));
  // (the second branch's stream; the third and fourth are created when a step graph first needs them: every stream takes a turn
  // on the runtime's four hardware queues, and up to 64 clips only four of the engine's streams ever work side by side)
  HIP_CHECK(hipStreamCreateWithFlags(&branch_stream_[0], hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&admit_stream_, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
  if (const char* e = getenv("AX_WHISPER_PAD_STREAMS")) {  // diagnostic (profiles/scripts/stream_mode_probe.py): shifts which hardware
    for (int i = 0; i < atoi(e) && i < 8; ++i) {            // queue every LATER stream of the process lands on (graph-internal ones too)
      hipStream_t pad = nullptr;
      HIP_CHECK(hipStreamCreateWithFlags(&pad, hipStreamNonBlocking));
      pad_streams_.push_back(pad);
    }
  }
  for (auto& e : ev_ring_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : ev_step_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&ev_upload_, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
  for (auto& e : ev_join_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

  load_weights(weights_file);

  if (max_batch <= 0) {
    const char* e = getenv("AX_WHISPER_MAX_BATCH");
    max_batch = e ? atoi(e) : 1;
  }
  HIP_CHECK(hipHostMalloc((void**)&h_poll_, 64 * sizeof(int), hipHostMallocDefault));
  {  // persistent batch-1 decode: one workgroup per CU for the whole utterance (decode_persistent.hip)
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device_));
    n_cu_ = prop.multiProcessorCount;
    const char* mode = getenv("AX_WHISPER_DECODE");
    const bool want = !(mode && std::string(mode) == "graph");
    persistent_ok_ = want && decode_persistent_supported(cfg_.n_text_state, cfg_.n_text_head, cfg_.n_text_layer, prop.multiProcessorCount);
    if (persistent_ok_) {
      persist_grid_ = decode_persistent_grid(cfg_.n_text_state, prop.multiProcessorCount);
      gran_bytes_ = decode_persistent_gran_bytes(cfg_.n_text_state, persist_grid_);
      d_gran_ = (u64*)dalloc(3 * gran_bytes_, true);  // one area per clip of a multi-clip launch
      allocs_.push_back(d_gran_);
      // two or three clips per launch (decode_persistent2.hip): the later clips' self-attention caches live in global memory
      // "0" / "1": one launch per clip, as before round 4; "2": at most two clips per launch (A/B, tests); default: up to three
      const char* e2 = getenv("AX_WHISPER_PERSIST2");
      persist_max_clips_ = decode_persistent_max_clips(cfg_.n_text_state, cfg_.n_text_head, cfg_.n_text_layer, persist_grid_);
      if (e2 && e2[0] >= '0' && e2[0] <= '9') persist_max_clips_ = std::max(1, std::min(persist_max_clips_, atoi(e2)));
      // the one-clip launch with the cross-attention query folded through the output projection (decode_persistent.hip, round 5):
      // M = W_cq diag(g) W_o in fp32 and three vectors per layer, built once here; AX_WHISPER_QFOLD=0: the unfolded launch
      {
        const char* eq = getenv("AX_WHISPER_QFOLD");
        if (cfg_.n_text_state <= 768 && !(eq && eq[0] == '0')) {
          d_qfold_ = (float*)dalloc(qfold_floats(cfg_.n_text_state, cfg_.n_text_layer) * sizeof(float));
          allocs_.push_back(d_qfold_);
          launch_qfold_build(dec_w_arena_, dec_f_arena_, d_qfold_, cfg_.n_text_state, cfg_.n_text_layer, own_stream_);
          HIP_CHECK(hipStreamSynchronize(own_stream_));
        }
      }
      if (persist_max_clips_ >= 2) {
        self1_bytes_ = (size_t)cfg_.n_text_layer * cfg_.n_text_head * 8 * 4096 * 2;  // one later clip's cache (K; V alike)
        d_self_k1_ = (h16*)dalloc((persist_max_clips_ - 1) * self1_bytes_, true);
        d_self_v1_ = (h16*)dalloc((persist_max_clips_ - 1) * self1_bytes_, true);
        allocs_.push_back(d_self_k1_);
        allocs_.push_back(d_self_v1_);
      }
    }
  }
  cfg_.ints["persistent_two_clips"] = persist_max_clips_ >= 2 ? 1 : 0;
  cfg_.ints["persistent_max_clips"] = persist_max_clips_;
  cfg_.ints["persistent_decode"] = persistent_ok_ ? 1 : 0;  // visible through AX_WHISPER_GetConfigInt
  cfg_.ints["persistent_qfold"] = d_qfold_ ? 1 : 0;
  cfg_.ints["persistent_giveups"] = 0;
  {  // batched decode as clip-block GEMMs with LayerNorm prologue / residual epilogue (enqueue_decode_step_batched)
    const char* e = getenv("AX_WHISPER_BATCHED_LN");
    const int d = cfg_.n_text_state;
    // measured on MI355X: faster for d_model 768 at 16-64 clips (+2..8 %); slower for 1280 at 16-32 clips (-3..6 %, also
    // with eight k-steps in flight per wave) and equal at 64, which keeps the split-K sequence
    batched_ln_ = !(e && e[0] == '0') && d % 128 == 0 && (d <= 1024 || (e && e[0] == '2' && d <= 1280));  // '2': force (A/B runs)
    cfg_.ints["batched_ln"] = batched_ln_ ? 1 : 0;
    build_cblock_fold();  // (needs the fold arena above and batched_ln_)
  }
  {
    enc_split_k_ = true;
    // A/B and test switches of the batched decode sequence (read per engine)
    if (const char* t = getenv("AX_WHISPER_GEMV_MAX")) gemv_max_ = std::max(1, std::min(4, atoi(t)));
    if (const char* t = getenv("AX_WHISPER_CROSS_SPLIT")) cross_split_env_ = atoi(t);
    // encoder attention: rescale threshold of the running softmax maximum (tests run 0 = rescale on every increase)
    if (const char* t = getenv("AX_WHISPER_ENC_RESCALE_THR")) enc_rescale_thr_ = std::max(0.f, std::min(16.f, (float)atof(t)));
  }
  cfg_.ints["t2s"] = t2s_ ? 1 : 0;
  cfg_.ints["fp16"] = AXW_F16;  // 16-bit storage / MFMA operand type of this engine: 0 bfloat16, 1 IEEE half
  ensure_capacity(std::max(1, max_batch));
  HIP_CHECK(hipStreamSynchronize(own_stream_));
}

Engine::~Engine() { destroy(); }

void Engine::destroy() {
  if (!device_set_) return;  // nothing was created
  std::lock_guard<std::recursive_mutex> capture_lock(device_capture_mutex(device_));
  (void)hipSetDevice(device_);
  (void)hipDeviceSynchronize();
  free_slot_buffers();  // also destroys the captured step graphs
  for (void* p : allocs_) (void)hipFree(p);
  allocs_.clear();
  if (load_stage_) { (void)hipFree(load_stage_); load_stage_ = nullptr; }
  if (d_over_) { (void)hipFree(d_over_); d_over_ = nullptr; over_cap_ = 0; }
  if (h_poll_) { (void)hipHostFree(h_poll_); h_poll_ = nullptr; }
  for (auto& e : ev_) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  for (auto& e : ev_join_) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  if (ev_fork_) { (void)hipEventDestroy(ev_fork_); ev_fork_ = nullptr; }
  for (auto& b : branch_stream_) if (b) { (void)hipStreamDestroy(b); b = nullptr; }
  if (admit_stream_) { (void)hipStreamDestroy(admit_stream_); admit_stream_ = nullptr; }
  if (copy_stream_) { (void)hipStreamDestroy(copy_stream_); copy_stream_ = nullptr; }
  for (auto& ps : pad_streams_) (void)hipStreamDestroy(ps);
  pad_streams_.clear();
  for (auto& e : ev_ring_) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  for (auto& e : ev_step_) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  if (ev_upload_) { (void)hipEventDestroy(ev_upload_); ev_upload_ = nullptr; }
  if (h_admit_ring_) { (void)hipHostFree(h_admit_ring_); h_admit_ring_ = nullptr; }
  for (auto& e : ev_admit_) if (e) (void)hipEventDestroy(e);
  ev_admit_.clear();
  if (h_done_live_) { (void)hipHostFree(h_done_live_); h_done_live_ = nullptr; }
  if (own_stream_) { (void)hipStreamDestroy(own_stream_); own_stream_ = nullptr; }
}

void* Engine::dalloc(size_t bytes, bool zero) {
  void* p = nullptr;
  HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 256)));
  if (zero) {
    // the engine's streams are non-blocking (not ordered against the null stream): finish the fill before any
    // kernel on them can touch the buffer
    hipError_t e = hipMemset(p, 0, std::max<size_t>(bytes, 256));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      (void)hipFree(p);
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " zero-filling a device buffer");
    }
  }
  return p;
}

// Whisper.cpp:86-101,129-139,241-251: config keys, comma-joined language lists, SOT sequence.
void Engine::load_config(const std::string& dir, const std::string& type, const std::string& language) {
  JsonValue j = JsonParser(read_text_file(dir + "/" + type + "_config.json")).parse();
  for (auto& kv : j.obj)
    if (kv.second.kind == JsonValue::Number) cfg_.ints[kv.first] = kv.second.as_int();
  auto geti = [&](const char* k) { return (int)j.at(k).as_int(); };
  // feature_mode: "axera_cpp" (default) = the C++ runtime's pipeline, the drop-in target (Whisper.cpp:
  // 151-184); "openai" = the front-end of the fp32 ONNX / PyTorch lineage (generate_data.py:162-176), for comparing
  // against that lineage. From the config file (key "feature_mode"), overridden by env AX_WHISPER_FEATURE_MODE.
  {
    std::string fm = j.has("feature_mode") && j.at("feature_mode").kind == JsonValue::String ? j.at("feature_mode").as_str() : "axera_cpp";
    if (const char* e = getenv("AX_WHISPER_FEATURE_MODE")) fm = e;
    if (fm != "axera_cpp" && fm != "openai") throw std::runtime_error("feature_mode must be axera_cpp or openai, not '" + fm + "'");
    feature_openai_ = fm == "openai";
    cfg_.ints["feature_mode_openai"] = feature_openai_ ? 1 : 0;
  }
  cfg_.n_mels = geti("n_mels");
  cfg_.n_vocab = geti("n_vocab");
  cfg_.n_text_state = geti("n_text_state");
  cfg_.n_text_ctx = geti("n_text_ctx");
  cfg_.n_text_layer = geti("n_text_layer");
  cfg_.n_text_head = j.has("n_text_head") ? geti("n_text_head") : cfg_.n_text_state / 64;
  cfg_.n_audio_ctx = j.has("n_audio_ctx") ? geti("n_audio_ctx") : 1500;
  cfg_.n_audio_state = j.has("n_audio_state") ? geti("n_audio_state") : cfg_.n_text_state;
  cfg_.n_audio_head = j.has("n_audio_head") ? geti("n_audio_head") : cfg_.n_audio_state / 64;
  cfg_.n_audio_layer = geti("n_audio_layer");
  cfg_.sot = geti("sot");
  cfg_.eot = geti("eot");
  cfg_.transcribe = geti("transcribe");
  cfg_.no_timestamps = geti("no_timestamps");
  for (auto& t : split_csv(j.at("all_language_tokens").as_str())) cfg_.lang_tokens.push_back(std::stoi(t));
  cfg_.lang_codes = split_csv(j.at("all_language_codes").as_str());
  if (cfg_.lang_tokens.size() != cfg_.lang_codes.size() || cfg_.lang_codes.empty())
    throw std::runtime_error("config: all_language_tokens / all_language_codes mismatch");
  if (cfg_.n_audio_state != cfg_.n_text_state) throw std::runtime_error("config: n_audio_state != n_text_state unsupported");
  if (cfg_.n_text_state % 128 != 0 || cfg_.n_text_state / cfg_.n_text_head != 64 || cfg_.n_audio_state / cfg_.n_audio_head != 64)
    throw std::runtime_error("config: d_model must be a multiple of 128 with head_dim 64");
  if (cfg_.n_audio_ctx != 1500 || cfg_.n_text_ctx != 448) throw std::runtime_error("config: expected n_audio_ctx 1500 and n_text_ctx 448");
  if (cfg_.n_text_state > 2048) throw std::runtime_error("config: d_model > 2048 unsupported");
  // get_lang_token (Whisper.cpp:241-251): unknown language falls back to DEFAULT_LANG "zh"
  auto it = std::find(cfg_.lang_codes.begin(), cfg_.lang_codes.end(), language);
  if (it == cfg_.lang_codes.end()) it = std::find(cfg_.lang_codes.begin(), cfg_.lang_codes.end(), std::string("zh"));
  if (it == cfg_.lang_codes.end()) it = cfg_.lang_codes.begin();
  effective_lang_ = *it;  // Whisper.cpp:244-248: an unknown language silently becomes DEFAULT_LANG
  sot_seq_[0] = cfg_.sot;
  sot_seq_[1] = cfg_.lang_tokens[it - cfg_.lang_codes.begin()];
  sot_seq_[2] = cfg_.transcribe;
  sot_seq_[3] = cfg_.no_timestamps;
  for (int i = 0; i < 4; ++i) cfg_.ints["sot_seq" + std::to_string(i)] = sot_seq_[i];
}

// Slaney mel filterbank, arithmetic as librosa.h:102-144 (fp32), stored transposed [201][n_mels].
// librosa.filters.mel as upstream's mel_filters.npz was generated (feature_mode openai): ramps in float64, narrowed,
// scaled by the float64 Slaney norm, narrowed again; stored transposed [201][n_mels].
static std::vector<float> make_mel_basis_t_librosa(int n_mels) {
  const int n_f = kBins;
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  const double max_mel = min_log_mel + std::log(8000.0 / min_log_hz) / logstep;
  std::vector<double> mel_f(n_mels + 2);
  for (int i = 0; i < n_mels + 2; ++i) {
    const double mel = max_mel * (double)i / (double)(n_mels + 1);
    mel_f[i] = mel >= min_log_mel ? min_log_hz * std::exp(logstep * (mel - min_log_mel)) : f_sp * mel;
  }
  std::vector<float> out((size_t)n_f * n_mels);
  for (int m = 0; m < n_mels; ++m) {
    const double fd0 = mel_f[m + 1] - mel_f[m], fd1 = mel_f[m + 2] - mel_f[m + 1], enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
    for (int k = 0; k < n_f; ++k) {
      const double freq = (double)k * 16000.0 / kNFFT;
      const double w = std::min(-(mel_f[m] - freq) / fd0, (mel_f[m + 2] - freq) / fd1);
      out[(size_t)k * n_mels + m] = (float)((double)(float)std::max(w, 0.0) * enorm);
    }
  }
  return out;
}

static std::vector<float> make_mel_basis_t(int n_mels) {
  const int sr = 16000, n_fft = kNFFT, n_f = kBins, fmin = 0, fmax = 8000;
  const float f_min = 0.f, f_sp = 200.f / 3.f, min_log_hz = 1000.f;
  const float min_log_mel = (min_log_hz - f_min) / f_sp, logstep = logf(6.4f) / 27.f;
  auto hz_to_mel = [&](int hz) {
    float mel = (hz - f_min) / f_sp;
    if (hz >= min_log_hz) mel = min_log_mel + logf(hz / min_log_hz) / logstep;
    return mel;
  };
  const float min_mel = hz_to_mel(fmin), max_mel = hz_to_mel(fmax);
  const int nm2 = n_mels + 2;
  std::vector<float> mel_f(nm2);
  const float stepv = (max_mel - min_mel) / (float)(nm2 - 1);
  for (int i = 0; i < nm2; ++i) {
    float mel = (i == nm2 - 1) ? max_mel : min_mel + (float)i * stepv;  // Eigen LinSpaced
    mel_f[i] = (mel > min_log_mel) ? expf((mel - min_log_mel) * logstep) * min_log_hz : mel * f_sp + f_min;
  }
  std::vector<float> out((size_t)n_f * n_mels);
  for (int m = 0; m < n_mels; ++m) {
    const float fd0 = mel_f[m + 1] - mel_f[m], fd1 = mel_f[m + 2] - mel_f[m + 1];
    const float enorm = (float)(2.0 / (double)(mel_f[m + 2] - mel_f[m]));
    for (int k = 0; k < n_f; ++k) {
      const float freq = ((float)k * sr) / n_fft;
      const float lower = -(mel_f[m] - freq) / fd0, upper = (mel_f[m + 2] - freq) / fd1;
      out[(size_t)k * n_mels + m] = std::max(0.f, std::min(lower, upper)) * enorm;
    }
  }
  return out;
}

void Engine::load_weights(const SafeTensors& st) {
  const int d = cfg_.n_text_state, nm = cfg_.n_mels, L = cfg_.n_text_layer, Le = cfg_.n_audio_layer;
  hipStream_t s = own_stream_;

  size_t max_bytes = 0;
  auto note = [&](const std::string& n) { max_bytes = std::max(max_bytes, st.get(n).nbytes); };
  note("decoder.token_embedding.weight");
  note("encoder.conv2.weight");
  note("encoder.blocks.0.mlp.0.weight");
  load_stage_ = dalloc(max_bytes);  // a member: freed by destroy() when a later tensor throws
  void* const stage = load_stage_;

  auto check = [&](const std::string& n, std::vector<int64_t> shape) -> const TensorView& {
    const TensorView& t = st.get(n);
    if (t.shape != shape) throw std::runtime_error("tensor '" + n + "' has an unexpected shape");
    return t;
  };
  // upload one tensor into the staging buffer; conversions run on `s` in order
  auto up = [&](const TensorView& t) {
    HIP_CHECK(hipStreamSynchronize(s));  // staging buffer is reused
    HIP_CHECK(hipMemcpy(stage, t.data, t.nbytes, hipMemcpyHostToDevice));
  };
  auto to_f32 = [&](const std::string& n, std::vector<int64_t> shape) {
    const TensorView& t = check(n, shape);
    float* dst = (float*)dalloc((size_t)t.numel() * 4);
    allocs_.push_back(dst);
    up(t);
    launch_convert_to_f32(stage, dtype_code(t.dtype), dst, t.numel(), s);
    return dst;
  };
  auto to_h16_into = [&](const std::string& n, std::vector<int64_t> shape, h16* dst) {
    const TensorView& t = check(n, shape);
    up(t);
    launch_convert_to_h16(stage, dtype_code(t.dtype), dst, t.numel(), s);
  };
  auto to_f32_into = [&](const std::string& n, std::vector<int64_t> shape, float* dst) {
    const TensorView& t = check(n, shape);
    up(t);
    launch_convert_to_f32(stage, dtype_code(t.dtype), dst, t.numel(), s);
  };
  auto new_h16 = [&](size_t n) { h16* p = (h16*)dalloc(n * 2, true); allocs_.push_back(p); return p; };
  auto new_f32 = [&](size_t n) { float* p = (float*)dalloc(n * 4, true); allocs_.push_back(p); return p; };

  // conv stem: [Cout][Cin][3] -> GEMM weights with k-major taps (gemm.hip header)
  conv1_k_ = ((3 * nm + 63) / 64) * 64;
  conv1_w_ = new_h16((size_t)d * conv1_k_);
  {
    const TensorView& t = check("encoder.conv1.weight", {d, nm, 3});
    up(t);
    launch_conv_weight_pack(stage, dtype_code(t.dtype), conv1_w_, d, nm, conv1_k_, s);
  }
  conv2_w_ = new_h16((size_t)d * 3 * d);
  {
    const TensorView& t = check("encoder.conv2.weight", {d, d, 3});
    up(t);
    launch_conv_weight_pack(stage, dtype_code(t.dtype), conv2_w_, d, d, 3 * d, s);
  }
  conv1_b_ = to_f32("encoder.conv1.bias", {d});
  conv2_b_ = to_f32("encoder.conv2.bias", {d});
  if (st.has("encoder.positional_embedding")) {
    enc_pos_ = to_f32("encoder.positional_embedding", {cfg_.n_audio_ctx, d});
  } else {  // upstream sinusoids(n_audio_ctx, d)
    std::vector<float> pe((size_t)cfg_.n_audio_ctx * d);
    const int half = d / 2;
    const float inc = logf(10000.f) / (float)(half - 1);
    for (int t = 0; t < cfg_.n_audio_ctx; ++t)
      for (int c = 0; c < half; ++c) {
        float v = (float)t * expf(-inc * (float)c);
        pe[(size_t)t * d + c] = sinf(v);
        pe[(size_t)t * d + half + c] = cosf(v);
      }
    enc_pos_ = new_f32(pe.size());
    HIP_CHECK(hipMemcpy(enc_pos_, pe.data(), pe.size() * 4, hipMemcpyHostToDevice));
  }
  ln_post_w_ = to_f32("encoder.ln_post.weight", {d});
  ln_post_b_ = to_f32("encoder.ln_post.bias", {d});

  // attention block: q,k,v rows concatenated [3d][d]; key has no bias (upstream: bias=False)
  auto load_attn = [&](const std::string& pre, h16*& w_qkv, float*& b_qkv, h16*& w_o, float*& b_o) {
    w_qkv = new_h16((size_t)3 * d * d);
    b_qkv = new_f32((size_t)3 * d);
    to_h16_into(pre + ".query.weight", {d, d}, w_qkv);
    to_h16_into(pre + ".key.weight", {d, d}, w_qkv + (size_t)d * d);
    to_h16_into(pre + ".value.weight", {d, d}, w_qkv + (size_t)2 * d * d);
    to_f32_into(pre + ".query.bias", {d}, b_qkv);
    to_f32_into(pre + ".value.bias", {d}, b_qkv + 2 * d);
    w_o = new_h16((size_t)d * d);
    to_h16_into(pre + ".out.weight", {d, d}, w_o);
    b_o = to_f32(pre + ".out.bias", {d});
  };
  auto load_mlp = [&](const std::string& pre, h16*& w1, float*& b1, h16*& w2, float*& b2) {
    w1 = new_h16((size_t)4 * d * d);
    to_h16_into(pre + ".mlp.0.weight", {4 * d, d}, w1);
    b1 = to_f32(pre + ".mlp.0.bias", {4 * d});
    w2 = new_h16((size_t)4 * d * d);
    to_h16_into(pre + ".mlp.2.weight", {d, 4 * d}, w2);
    b2 = to_f32(pre + ".mlp.2.bias", {d});
  };

  enc_.resize(Le);
  for (int i = 0; i < Le; ++i) {
    const std::string pre = "encoder.blocks." + std::to_string(i);
    EncLayer& e = enc_[i];
    e.ln1_w = to_f32(pre + ".attn_ln.weight", {d});
    e.ln1_b = to_f32(pre + ".attn_ln.bias", {d});
    load_attn(pre + ".attn", e.w_qkv, e.b_qkv, e.w_o, e.b_o);
    e.ln2_w = to_f32(pre + ".mlp_ln.weight", {d});
    e.ln2_b = to_f32(pre + ".mlp_ln.bias", {d});
    load_mlp(pre, e.w_fc1, e.b_fc1, e.w_fc2, e.b_fc2);
  }

  // cross K/V projection of every decoder layer as ONE GEMM: rows [all K | all V] (gemm.hip EPI_CROSS_KV)
  w_cross_kv_ = new_h16((size_t)2 * L * d * d);
  b_cross_kv_ = new_f32((size_t)2 * L * d);
  // decoder layer weights live in two arenas with a fixed per-layer stride (layout: DecArena in common.hpp), so the
  // persistent decode kernel derives every address from two base pointers with scalar arithmetic
  dec_w_arena_ = new_h16((size_t)L * DecArena::w_stride(d));
  dec_f_arena_ = new_f32((size_t)L * DecArena::f_stride(d));
  dec_.resize(L);
  for (int i = 0; i < L; ++i) {
    const std::string pre = "decoder.blocks." + std::to_string(i);
    DecLayerW& w = dec_[i];
    h16* wb = dec_w_arena_ + (size_t)i * DecArena::w_stride(d);
    float* fb = dec_f_arena_ + (size_t)i * DecArena::f_stride(d);
    const size_t dd = (size_t)d * d;
    h16 *w_qkv = wb + DecArena::W_QKV * dd, *w_o = wb + DecArena::W_O * dd, *w_cq = wb + DecArena::W_CQ * dd,
         *w_co = wb + DecArena::W_CO * dd, *w_fc1 = wb + DecArena::W_FC1 * dd, *w_fc2 = wb + DecArena::W_FC2 * dd;
    auto f = [&](int off) { return fb + (size_t)off * d; };
    to_f32_into(pre + ".attn_ln.weight", {d}, f(DecArena::F_ATTN_LN_W));
    to_f32_into(pre + ".attn_ln.bias", {d}, f(DecArena::F_ATTN_LN_B));
    // q,k,v rows concatenated [3d][d]; key has no bias (upstream: bias=False): the arena is zero-initialised
    to_h16_into(pre + ".attn.query.weight", {d, d}, w_qkv);
    to_h16_into(pre + ".attn.key.weight", {d, d}, w_qkv + dd);
    to_h16_into(pre + ".attn.value.weight", {d, d}, w_qkv + 2 * dd);
    to_f32_into(pre + ".attn.query.bias", {d}, f(DecArena::F_B_QKV));
    to_f32_into(pre + ".attn.value.bias", {d}, f(DecArena::F_B_QKV) + 2 * d);
    to_h16_into(pre + ".attn.out.weight", {d, d}, w_o);
    to_f32_into(pre + ".attn.out.bias", {d}, f(DecArena::F_B_O));
    to_f32_into(pre + ".cross_attn_ln.weight", {d}, f(DecArena::F_CROSS_LN_W));
    to_f32_into(pre + ".cross_attn_ln.bias", {d}, f(DecArena::F_CROSS_LN_B));
    to_h16_into(pre + ".cross_attn.query.weight", {d, d}, w_cq);
    to_f32_into(pre + ".cross_attn.query.bias", {d}, f(DecArena::F_B_CQ));
    to_h16_into(pre + ".cross_attn.key.weight", {d, d}, w_cross_kv_ + (size_t)i * d * d);
    to_h16_into(pre + ".cross_attn.value.weight", {d, d}, w_cross_kv_ + (size_t)(L + i) * d * d);
    to_f32_into(pre + ".cross_attn.value.bias", {d}, b_cross_kv_ + (size_t)(L + i) * d);
    to_h16_into(pre + ".cross_attn.out.weight", {d, d}, w_co);
    to_f32_into(pre + ".cross_attn.out.bias", {d}, f(DecArena::F_B_CO));
    to_f32_into(pre + ".mlp_ln.weight", {d}, f(DecArena::F_MLP_LN_W));
    to_f32_into(pre + ".mlp_ln.bias", {d}, f(DecArena::F_MLP_LN_B));
    to_h16_into(pre + ".mlp.0.weight", {4 * d, d}, w_fc1);
    to_f32_into(pre + ".mlp.0.bias", {4 * d}, f(DecArena::F_B_FC1));
    to_h16_into(pre + ".mlp.2.weight", {d, 4 * d}, w_fc2);
    to_f32_into(pre + ".mlp.2.bias", {d}, f(DecArena::F_B_FC2));
    w.attn_ln_w = f(DecArena::F_ATTN_LN_W); w.attn_ln_b = f(DecArena::F_ATTN_LN_B);
    w.cross_ln_w = f(DecArena::F_CROSS_LN_W); w.cross_ln_b = f(DecArena::F_CROSS_LN_B);
    w.mlp_ln_w = f(DecArena::F_MLP_LN_W); w.mlp_ln_b = f(DecArena::F_MLP_LN_B);
    w.w_qkv = w_qkv; w.w_o = w_o; w.w_cq = w_cq; w.w_co = w_co; w.w_fc1 = w_fc1; w.w_fc2 = w_fc2;
    w.b_qkv = f(DecArena::F_B_QKV); w.b_o = f(DecArena::F_B_O); w.b_cq = f(DecArena::F_B_CQ); w.b_co = f(DecArena::F_B_CO);
    w.b_fc1 = f(DecArena::F_B_FC1); w.b_fc2 = f(DecArena::F_B_FC2);
  }
  tok_emb_ = new_h16((size_t)cfg_.n_vocab * d);
  to_h16_into("decoder.token_embedding.weight", {cfg_.n_vocab, d}, tok_emb_);
  dec_pos_ = to_f32("decoder.positional_embedding", {cfg_.n_text_ctx, d});
  dec_ln_w_ = to_f32("decoder.ln.weight", {d});
  dec_ln_b_ = to_f32("decoder.ln.bias", {d});
  // fragment-major copies of the decoder weights for the batched (MFMA) decode path (decode_gemm.hip)
  auto pack = [&](const h16* w, int N, int K) {
    h16* wp = new_h16((size_t)((N + 15) / 16) * 16 * K);
    launch_pack_weight_frag(w, wp, N, K, s);
    return (const h16*)wp;
  };
  dec_packed_.resize(L);
  for (int i = 0; i < L; ++i) {
    // fragment-major row blocks are contiguous, so packed [W_qkv; W_cq] is packed W_qkv followed by packed W_cq, and the query
    // fold's M_hi rows follow packed W_o (build_cblock_fold fills them; d % 16 == 0 for every Whisper size)
    h16* qc = new_h16((size_t)4 * d * d);
    launch_pack_weight_frag(dec_[i].w_qkv, qc, 3 * d, d, s);
    launch_pack_weight_frag(dec_[i].w_cq, qc + (size_t)3 * d * d, d, d, s);
    dec_packed_[i].w_qkv = qc;
    dec_packed_[i].w_cq = qc + (size_t)3 * d * d;
    h16* om = new_h16((size_t)2 * d * d);
    launch_pack_weight_frag(dec_[i].w_o, om, d, d, s);
    dec_packed_[i].w_o = om;
    dec_packed_[i].m_hi = om + (size_t)d * d;
    dec_packed_[i].m_lo = nullptr;
    dec_packed_[i].w_co = pack(dec_[i].w_co, d, d);
    dec_packed_[i].w_fc1 = pack(dec_[i].w_fc1, 4 * d, d);
    dec_packed_[i].w_fc2 = pack(dec_[i].w_fc2, d, 4 * d);
  }
  tok_emb_packed_ = pack(tok_emb_, cfg_.n_vocab, d);
  HIP_CHECK(hipStreamSynchronize(s));
  HIP_CHECK(hipFree(stage));
  load_stage_ = nullptr;

  // front-end constants: DFT twiddles (double -> f32), periodic Hann (librosa.h:81), mel basis
  std::vector<float> tw(2 * kNFFT), win(kNFFT);
  for (int i = 0; i < kNFFT; ++i) {
    tw[2 * i] = (float)cos(2.0 * M_PI * i / kNFFT);
    tw[2 * i + 1] = (float)sin(2.0 * M_PI * i / kNFFT);
    win[i] = feature_openai_ ? (float)(0.5 * (1.0 - cos(2.0 * M_PI * i / kNFFT)))  // torch.hann_window
                             : 0.5f * (1.f - cosf((float)i * 2.f * (float)M_PI / (float)kNFFT));  // librosa.h:81
  }
  std::vector<float> mb = feature_openai_ ? make_mel_basis_t_librosa(nm) : make_mel_basis_t(nm);
  twiddle_ = new_f32(tw.size());
  window_ = new_f32(win.size());
  mel_basis_t_ = new_f32(mb.size());
  HIP_CHECK(hipMemcpy(twiddle_, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(window_, win.data(), win.size() * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(mel_basis_t_, mb.data(), mb.size() * 4, hipMemcpyHostToDevice));
  d_sot_ = (int*)dalloc(16);
  allocs_.push_back(d_sot_);
  HIP_CHECK(hipMemcpy(d_sot_, sot_seq_, 16, hipMemcpyHostToDevice));
}

// ------------------------------------------------------------------------------ slot buffers
void Engine::free_slot_buffers() {
  for (auto& g : graphs_) (void)hipGraphExecDestroy(g.second);
  graphs_.clear();
  for (void* p : slot_allocs_) (void)hipFree(p);
  slot_allocs_.clear();
  if (h_pcm_) { (void)hipHostFree(h_pcm_); h_pcm_ = nullptr; }
  cap_ = 0;
}

void Engine::ensure_capacity(int batch) {
  if (batch <= cap_) return;
  std::lock_guard<std::recursive_mutex> capture_lock(device_capture_mutex(device_));
  HIP_CHECK(hipDeviceSynchronize());
  free_slot_buffers();
  const int B = batch, d = cfg_.n_text_state, nm = cfg_.n_mels, H = cfg_.n_text_head, L = cfg_.n_text_layer;
  const int T = cfg_.n_audio_ctx, Tc = cfg_.n_text_ctx;
  auto A = [&](size_t bytes, bool zero = false) { void* p = dalloc(bytes, zero); slot_allocs_.push_back(p); return p; };
  pcm_stride_ = 2 * 480000;  // staging row of a clip (60 s; the window itself is 30 s); longer clips put their tails into d_over_
  d_pcm_ = (float*)A((size_t)B * pcm_stride_ * 4, true);
  HIP_CHECK(hipHostMalloc((void**)&h_pcm_, (size_t)B * pcm_stride_ * 4, hipHostMallocDefault));
  d_nsamp_ = (int*)A((size_t)B * 4);
  d_over_off_ = (long long*)A((size_t)B * 8, true);
  d_gmax_ = (unsigned*)A((size_t)B * 4);
  d_logmel_ = (float*)A((size_t)B * kFramesOut * nm * 4);
  d_mel_ref_ = (float*)A((size_t)B * nm * kFramesOut * 4);
  d_mel_tm_ = (h16*)A((size_t)B * mel_rows_ * nm * 2 + 4096, true);
  d_h1_ = (h16*)A((size_t)B * h1_rows_ * d * 2 + 4096, true);
  d_x_ = (float*)A((size_t)B * T * d * 4);
  d_ln_ = (h16*)A((size_t)B * T * d * 2);
  d_q_ = (h16*)A((size_t)B * T * d * 2);
  d_k_ = (h16*)A((size_t)B * T * d * 2);
  d_vt_ = (h16*)A((size_t)B * d * t_pad_ * 2, true);
  d_attn_ = (h16*)A((size_t)B * T * d * 2);
  d_ffn_ = (h16*)A((size_t)B * T * 4 * d * 2);
  d_enc_part_ = (float*)A((size_t)4 * kEncPartClips * T * d * 4);  // split-K partials of the encoder's residual GEMMs (few clips only)
  d_cross_k_ = (h16*)A((size_t)L * B * H * t_pad_ * 64 * 2, true);
  d_cross_v_ = (h16*)A((size_t)L * B * H * t_pad_ * 64 * 2, true);
  d_self_k_ = (h16*)A((size_t)L * B * H * Tc * 64 * 2, true);
  d_self_v_ = (h16*)A((size_t)L * B * H * Tc * 64 * 2, true);
  d_xdec_ = (float*)A((size_t)B * d * 4, true);
  d_a0_ = (float*)A((size_t)B * d * 4, true);
  d_statp_ = (float*)A((size_t)B * (d / 16 + 1) * 2 * 4, true);
  d_qdec_ = (float*)A((size_t)B * d * 4, true);
  d_hid_ = (float*)A((size_t)B * 4 * d * 4, true);
  nbs_ = (B + 15) / 16;
  for (int i = 0; i < 2; ++i) {  // fragment-major h16 (hi, lo) activation pairs of the batched (MFMA) decode path
    d_act_[i] = (h16*)A((size_t)nbs_ * 16 * d * 2, true);
    d_att_[i] = (h16*)A((size_t)nbs_ * 16 * d * 2, true);
    d_hidp_[i] = (h16*)A((size_t)nbs_ * 16 * 4 * d * 2, true);
  }
  split_cross_ = B <= 2 ? 8 : 3;  // the VALU path serves <= 4 clips; larger batches use one split per (clip, head)
  split_self_ = 2;
  d_part_ = (float*)A((size_t)4 * B * d * 4, true);  // split-K partials of the batched residual GEMMs
  d_part_self_ = (float*)A((size_t)B * H * split_self_ * 66 * 4, true);
  d_part_cross_ = (float*)A((size_t)B * H * split_cross_ * 66 * 4, true);
  GemvParams lp{};
  lp.N = cfg_.n_vocab; lp.K = d;
  n_amax_part_ = std::max(gemv_grid(lp), decode_gemm_grid(cfg_.n_vocab, logits_rt()));
  d_amax_val_ = (float*)A((size_t)n_amax_part_ * B * 4, true);
  d_amax_idx_ = (int*)A((size_t)n_amax_part_ * B * 4, true);
  d_tok_ = (int*)A((size_t)B * 4, true);
  d_done_ = (int*)A((size_t)B * 4, true);
  d_done_none_ = (int*)A((size_t)B * 4, true);  // all zero, never written: the "nobody has finished" flags of teacher-forced decodes
  d_off_ = (int*)A((size_t)B * 4, true);
  d_slot_map_ = (int*)A((size_t)B * 4, true);
  if (h_done_live_) { (void)hipHostFree(h_done_live_); h_done_live_ = nullptr; }
  HIP_CHECK(hipHostMalloc((void**)&h_done_live_, (size_t)B * 4, hipHostMallocMapped));
  memset(h_done_live_, 0, (size_t)B * 4);
  HIP_CHECK(hipHostGetDevicePointer((void**)&d_done_live_, h_done_live_, 0));
  d_attn_mpart_ = (float*)A((size_t)B * cfg_.n_text_head * kCrossSplitMax * 66 * 4, true);
  d_attn_mcnt_ = (unsigned*)A((size_t)B * cfg_.n_text_head * 4, true);  // zero: every launch leaves its tickets at zero
  d_nout_ = (int*)A((size_t)B * 4, true);
  d_max_new_clip_ = (int*)A((size_t)B * 4, true);
  d_out_ids_ = (int*)A((size_t)B * Tc * 4, true);
  d_state_ = (DecState*)A(sizeof(DecState), true);
  cap_ = B;
  cfg_.ints["decode_branches"] = (B > 4 && batched_ln_) ? decode_branches(B) : 1;  // at full capacity (bench.py reads it)
}

// ------------------------------------------------------------------------------ front-end
void Engine::upload_pcm(const float* const* pcm, const int* n_samples, int batch) {
  // Clips longer than a staging row: the reference computes the log-mel of the WHOLE input and takes the maximum over
  // all of its frames before it keeps 3000 of them (Whisper.cpp:158-172), so every sample counts for the clamp floor.
  // The first pcm_stride_ samples go through the pinned staging rows as always; the tails are packed into d_over_.
  std::vector<long long> off(batch, 0);
  size_t over_total = 0;
  for (int b = 0; b < batch; ++b) {
    // (the reflect-pad index of the STFT is 2n - 2 - j in 32-bit arithmetic: 2^29 samples = 9.3 hours is the cap)
    if (n_samples[b] > (1 << 29)) throw std::runtime_error("clip " + std::to_string(b) + ": more than 2^29 samples");
    off[b] = (long long)over_total;
    if (n_samples[b] > pcm_stride_) over_total += (size_t)n_samples[b] - (size_t)pcm_stride_;
  }
  over_used_ = over_total > 0;
  // rare path (clips beyond a 60 s staging row): allocation and synchronous copies, which must not run beside another
  // handle's stream capture (iengine.hpp)
  std::unique_lock<std::recursive_mutex> capture_lock(device_capture_mutex(device_), std::defer_lock);
  if (over_total > 0) {
    capture_lock.lock();
    HIP_CHECK(hipStreamSynchronize(stream()));  // an earlier pass on this stream may still read the tails
  }
  if (over_total > over_cap_) {
    if (d_over_) { (void)hipFree(d_over_); d_over_ = nullptr; over_cap_ = 0; }
    HIP_CHECK(hipMalloc((void**)&d_over_, over_total * 4));
    over_cap_ = over_total;
  }
  if (over_used_) HIP_CHECK(hipMemcpy(d_over_off_, off.data(), (size_t)batch * 8, hipMemcpyHostToDevice));
  for (int b = 0; b < batch; ++b) {
    const int n = (int)std::min<long>(n_samples[b], pcm_stride_);
    // pcm_data is copied, not retained (api.cpp:151-152). The ABI asks for samples in [-1, 1] (ax_whisper_api.h:89 of the
    // reference); NaN / Inf samples turn every mel value, hence every logit, into NaN: refuse them here (-1 at the ABI)
    // instead of decoding garbage. Finite out-of-range samples pass through as they do in the reference.
    float* dst = h_pcm_ + (size_t)b * pcm_stride_;
    memcpy(dst, pcm[b], (size_t)n * 4);
    unsigned bad = 0;
    for (int i = 0; i < n; ++i) bad |= !std::isfinite(dst[i]);  // vectorises: ~0.1 ms per 30 s clip
    const float* tail = pcm[b] + n;
    const size_t n_tail = (size_t)n_samples[b] - (size_t)n;
    for (size_t i = 0; i < n_tail; ++i) bad |= !std::isfinite(tail[i]);
    if (bad) throw std::runtime_error("clip " + std::to_string(b) + ": non-finite PCM sample (NaN or Inf)");
    HIP_CHECK(hipMemcpyAsync(d_pcm_ + (size_t)b * pcm_stride_, h_pcm_ + (size_t)b * pcm_stride_, (size_t)n * 4,
                             hipMemcpyHostToDevice, stream()));
    if (n_tail) HIP_CHECK(hipMemcpy(d_over_ + off[b], tail, n_tail * 4, hipMemcpyHostToDevice));  // rare path: pageable, synchronous
  }
}

// staged: the clips came through upload_pcm (tails of clips beyond a staging row sit in d_over_); otherwise d_pcm is the
// caller's device buffer and a clip ends at its row's end
void Engine::run_frontend(const float* d_pcm, int stride, const int* n_samples, int batch, bool want_ref_layout, bool staged,
                          int* pinned_ns) {
  std::vector<int> ns(batch);
  int max_frames = 1;
  for (int b = 0; b < batch; ++b) {
    if (n_samples[b] < 1) throw std::runtime_error("empty audio clip");
    ns[b] = staged ? n_samples[b] : std::min(n_samples[b], stride);
    max_frames = std::max(max_frames, 1 + ns[b] / kHop);
  }
  if (feature_openai_) max_frames = kFramesOut;
  if (pinned_ns) {  // the caller keeps this buffer alive until the pass has run: no wait here
    memcpy(pinned_ns, ns.data(), (size_t)batch * 4);
    HIP_CHECK(hipMemcpyAsync(d_nsamp_, pinned_ns, (size_t)batch * 4, hipMemcpyHostToDevice, stream()));
  } else {
    HIP_CHECK(hipMemcpyAsync(d_nsamp_, ns.data(), (size_t)batch * 4, hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));  // ns is a stack vector
  }
  FrontendParams p{};
  p.pcm = d_pcm; p.stride = stride; p.n_samples = d_nsamp_; p.batch = batch; p.n_mels = cfg_.n_mels;
  p.twiddle = twiddle_; p.window = window_; p.mel_basis = mel_basis_t_;
  p.logmel = d_logmel_; p.gmax = d_gmax_;
  p.mel_ref = want_ref_layout ? d_mel_ref_ : nullptr;
  p.mel_tm = d_mel_tm_; p.mel_rows = mel_rows_; p.max_frames = max_frames; p.openai = feature_openai_ ? 1 : 0;
  if (staged && over_used_) { p.overflow = d_over_; p.over_off = d_over_off_; }
  launch_frontend(p, stream());
}

// The clip-block step's query fold (decode_gemm.hip "QUERY FOLD"): M of the fold arena as an (hi, lo) h16 pair in fragment order
// behind packed W_o, and the per-layer vectors. Needs the fold arena (d_model <= 768) and the clip-block sequence;
// AX_WHISPER_CBLOCK_QFOLD=0 keeps the fused query projection (A/B, tests).
void Engine::build_cblock_fold() {
  const char* e = getenv("AX_WHISPER_CBLOCK_QFOLD");
  const int d = cfg_.n_text_state, L = cfg_.n_text_layer;
  cfg_.ints["cblock_qfold"] = 0;
  if (!d_qfold_ || !batched_ln_ || d % 64 != 0 || (e && e[0] == '0')) return;
  hipStream_t s = own_stream_;
  cfold_.resize(L);
  for (int l = 0; l < L; ++l) {
    const float* qf = d_qfold_ + (size_t)l * qfold_floats(d, 1);  // per-layer block: M [d][d], then d, s, c [d] each (decode_persistent_common.hpp)
    const float* vecs = qf + (size_t)d * d;  // d, s, c
    DecLayerWP& wp = dec_packed_[l];
    wp.m_lo = (h16*)dalloc((size_t)d * d * 2, true);
    allocs_.push_back(wp.m_lo);
    launch_pack_weight_frag_split(qf, wp.m_hi, wp.m_lo, d, d, s);
    CblockFold& c = cfold_[l];
    c.b_qkv4 = (float*)dalloc((size_t)4 * d * 4, true);  // [b_qkv; 0]: the A0 rows have no bias
    c.b_o2 = (float*)dalloc((size_t)2 * d * 4, true);    // [b_o; d]
    allocs_.push_back(c.b_qkv4);
    allocs_.push_back(c.b_o2);
    HIP_CHECK(hipMemcpyAsync(c.b_qkv4, dec_[l].b_qkv, (size_t)3 * d * 4, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c.b_o2, dec_[l].b_o, (size_t)d * 4, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c.b_o2 + d, vecs, (size_t)d * 4, hipMemcpyDeviceToDevice, s));
    c.s = vecs + d;
    c.c = vecs + 2 * d;
  }
  HIP_CHECK(hipStreamSynchronize(s));
  cfold_all_ = e && e[0] == '2';
  cfg_.ints["cblock_qfold"] = cfold_all_ ? 2 : 1;
}

// ------------------------------------------------------------------------------ encoder
void Engine::run_encoder(int batch, const int* d_slot_map) {
  const int d = cfg_.n_text_state, nm = cfg_.n_mels, T = cfg_.n_audio_ctx, H = cfg_.n_audio_head, L = cfg_.n_text_layer;
  hipStream_t s = stream();
  GemmParams g{};
  // conv1 + GELU (export_onnx.py:158): A row t = mel frames t-1,t,t+1 (time-major, row 0 = zero pad)
  g.A = d_mel_tm_; g.lda = nm; g.a_batch_stride = (long)mel_rows_ * nm;
  g.W = conv1_w_; g.bias = conv1_b_;
  g.C = d_h1_ + d; g.ldc = d; g.c_batch_stride = (long)h1_rows_ * d;  // output row t -> h1 row t+1
  g.M = kFramesOut; g.N = d; g.K = conv1_k_; g.batch = batch; g.d_model = d; g.epilogue = EPI_BIAS_GELU_BF16;
  launch_gemm(g, s);
  // conv2 (stride 2) + GELU + positional embedding (export_onnx.py:159-176): A row t = h1 rows 2t-1,2t,2t+1
  g = GemmParams{};
  g.A = d_h1_; g.lda = 2 * d; g.a_batch_stride = (long)h1_rows_ * d;
  g.W = conv2_w_; g.bias = conv2_b_; g.aux = enc_pos_;
  g.C = d_x_; g.ldc = d; g.c_batch_stride = (long)T * d;
  g.M = T; g.N = d; g.K = 3 * d; g.batch = batch; g.d_model = d; g.epilogue = EPI_GELU_POS_F32;
  launch_gemm(g, s);

  auto linear = [&](const h16* A, int K, const h16* W, const float* bias, void* C, int N, int epi) {
    GemmParams q{};
    q.A = A; q.lda = K; q.a_batch_stride = (long)T * K;
    q.W = W; q.bias = bias; q.C = C; q.ldc = N; q.c_batch_stride = (long)T * N;
    q.M = T; q.N = N; q.K = K; q.batch = batch; q.d_model = d; q.epilogue = epi;
    launch_gemm(q, s);
  };
  // Residual GEMMs of a one-clip encoder pass have 72 tiles of 128x128 for 256 CUs: split K (2-4 slices along grid.y)
  // and let the LayerNorm that always follows fold the fp32 partials into x in fixed order (deterministic, no atomics).
  int pend_n = 0;
  const float* pend_bias = nullptr;
  auto resid = [&](const h16* A, int K, const h16* W, const float* bias) {
    const int tiles = (d / 128) * ((T + 127) / 128) * batch, nk = K / 64;
    int split = 1;
    if (enc_split_k_ && d % 128 == 0 && batch <= kEncPartClips && tiles * 2 <= 256)
      for (int sp = 4; sp > 1; --sp)
        if (nk % sp == 0 && nk / sp >= 4 && tiles * sp <= 512) { split = sp; break; }
    if (split == 1) { linear(A, K, W, bias, d_x_, d, EPI_RESID_F32); return; }
    GemmParams q{};
    q.A = A; q.lda = K; q.a_batch_stride = (long)T * K;
    q.W = W; q.M = T; q.N = d; q.K = K; q.batch = batch; q.d_model = d; q.epilogue = EPI_PARTIAL_F32;
    q.ksplit = split; q.part = d_enc_part_; q.part_stride = (long)kEncPartClips * T * d;
    launch_gemm(q, s);
    pend_n = split;
    pend_bias = bias;
  };
  auto layernorm = [&](const float* g, const float* be) {
    launch_layernorm_bf16(d_x_, g, be, d_ln_, (long)batch * T, d, s, d_enc_part_, pend_n, (long)kEncPartClips * T * d, pend_bias);
    pend_n = 0;
  };
  for (int l = 0; l < cfg_.n_audio_layer; ++l) {
    const EncLayer& e = enc_[l];
    layernorm(e.ln1_w, e.ln1_b);
    GemmParams q{};
    q.A = d_ln_; q.lda = d; q.a_batch_stride = (long)T * d;
    q.W = e.w_qkv; q.bias = e.b_qkv;
    q.C = d_q_; q.c_batch_stride = (long)T * d;
    q.C2 = d_k_; q.c2_batch_stride = (long)T * d;
    q.C3 = d_vt_; q.c3_batch_stride = (long)d * t_pad_;
    q.M = T; q.N = 3 * d; q.K = d; q.batch = batch; q.d_model = d; q.t_pad = t_pad_; q.epilogue = EPI_QKV;
    if (enc_split_k_ && (3 * d / 128) * ((T + 127) / 128) * batch <= 256) {
      // few tiles (one clip: 144 + 72): the Q,K launch and the V launch (swapped operands) run side by side
      HIP_CHECK(hipEventRecord(ev_fork_, s));
      HIP_CHECK(hipStreamWaitEvent(branch_stream_[0], ev_fork_, 0));
      q.qkv_part = 1;
      launch_gemm(q, s);
      q.qkv_part = 2;
      launch_gemm(q, branch_stream_[0]);
      HIP_CHECK(hipEventRecord(ev_join_[0], branch_stream_[0]));
      HIP_CHECK(hipStreamWaitEvent(s, ev_join_[0], 0));
    } else {
      launch_gemm(q, s);
    }
    launch_encoder_attention(d_q_, d_k_, d_vt_, d_attn_, batch, T, t_pad_, d, H, s, enc_rescale_thr_);
    resid(d_attn_, d, e.w_o, e.b_o);
    layernorm(e.ln2_w, e.ln2_b);
    linear(d_ln_, d, e.w_fc1, e.b_fc1, d_ffn_, 4 * d, EPI_BIAS_GELU_BF16);
    resid(d_ffn_, 4 * d, e.w_fc2, e.b_fc2);
  }
  layernorm(ln_post_w_, ln_post_b_);
  // cross K/V of all decoder layers (export_onnx.py:205-210), written in the decoder's layouts
  GemmParams c{};
  c.A = d_ln_; c.lda = d; c.a_batch_stride = (long)T * d;
  c.W = w_cross_kv_; c.bias = b_cross_kv_;
  c.C = d_cross_k_; c.C2 = d_cross_v_;
  c.kv_slot_map = d_slot_map;  // an admission pass (stream_admit) scatters its clips into whichever slots are idle
  c.M = T; c.N = 2 * L * d; c.K = d; c.batch = batch; c.d_model = d; c.t_pad = t_pad_;
  c.n_batch_total = cap_; c.n_layer = L; c.epilogue = EPI_CROSS_KV;
  launch_gemm(c, s);
}

// ------------------------------------------------------------------------------ public entry points
void Engine::run_tokens(const float* const* pcm, const float* d_pcm, int d_stride, const int* n_samples, int batch, int max_new,
                        int32_t* ids, int* n_ids, const int* max_new_clip) {
  if (batch < 1) throw std::runtime_error("batch must be >= 1");
  require_no_stream("run_tokens");
  HIP_CHECK(hipSetDevice(device_));
  auto t0 = std::chrono::steady_clock::now();
  ensure_capacity(batch);
  hipStream_t s = stream();
  HIP_CHECK(hipEventRecord(ev_[0], s));
  if (pcm) {
    upload_pcm(pcm, n_samples, batch);
    run_frontend(d_pcm_, (int)pcm_stride_, n_samples, batch, false, true);
  } else {
    run_frontend(d_pcm, d_stride, n_samples, batch, false);
  }
  HIP_CHECK(hipEventRecord(ev_[1], s));
  run_encoder(batch);
  HIP_CHECK(hipEventRecord(ev_[2], s));
  const int steps = greedy_loop(batch, max_new, max_new_clip);
  fetch_ids(batch, ids, n_ids);
  // stage timings (events 3/4 are reused by the poll; bracket decode with a fresh record)
  HIP_CHECK(hipEventRecord(ev_[3], s));
  HIP_CHECK(hipEventSynchronize(ev_[3]));
  (void)hipEventElapsedTime(&timings[0], ev_[0], ev_[1]);
  (void)hipEventElapsedTime(&timings[1], ev_[1], ev_[2]);
  (void)hipEventElapsedTime(&timings[2], ev_[2], ev_[3]);
  timings[3] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  timings[4] = (float)steps;
}

// Whisper.cpp:231-236: zh transcripts pass through OpenCC's t2s.json. The reference resolves "t2s.json" (and the two
// .ocd2 dictionaries it names) relative to the working directory; here: $AX_WHISPER_OPENCC_DIR, the working directory,
// then the model directory. Missing files are not an error (the text then stays as decoded), a broken file is.
void Engine::load_t2s(const std::string& model_path) {
  if (effective_lang_ != "zh") return;
  std::vector<std::string> dirs;
  if (const char* e = getenv("AX_WHISPER_OPENCC_DIR")) dirs.push_back(std::string(e) + "/");
  dirs.push_back("");
  dirs.push_back(model_path + "/");
  for (const std::string& d : dirs) {
    std::ifstream f(d + "t2s.json");
    if (!f.is_open()) continue;
    t2s_.reset(new T2SConverter(d + "t2s.json"));
    return;
  }
}

std::string Engine::transcript(const int32_t* ids, int n) const {
  std::string s = detokenize(ids, n);
  return t2s_ ? t2s_->convert(s) : s;
}

// Whisper.cpp:224-229 with bounds checks: bytes are concatenated, ids beyond the table skipped.
std::string Engine::detokenize(const int32_t* ids, int n) const {
  std::string out;
  for (int i = 0; i < n; ++i)
    if (ids[i] >= 0 && (size_t)ids[i] < tokens_.size()) out += tokens_[ids[i]];
  return out;
}

void Engine::compute_mel(const float* pcm, int n_samples, float* mel_out) {
  // an open stream's admission pass may still be reading the staging rows and front-end buffers this call overwrites
  require_no_stream("compute_mel");
  HIP_CHECK(hipSetDevice(device_));
  ensure_capacity(1);
  const float* arr[1] = {pcm};
  upload_pcm(arr, &n_samples, 1);
  run_frontend(d_pcm_, (int)pcm_stride_, &n_samples, 1, true, true);
  HIP_CHECK(hipMemcpyAsync(mel_out, d_mel_ref_, (size_t)cfg_.n_mels * kFramesOut * 4, hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipStreamSynchronize(stream()));
}

void Engine::encode_mel(const float* mel, int batch) {
  require_no_stream("encode_mel");
  HIP_CHECK(hipSetDevice(device_));
  ensure_capacity(batch);
  HIP_CHECK(hipMemcpyAsync(d_mel_ref_, mel, (size_t)batch * cfg_.n_mels * kFramesOut * 4, hipMemcpyHostToDevice, stream()));
  launch_mel_to_tm(d_mel_ref_, d_mel_tm_, batch, cfg_.n_mels, mel_rows_, stream());
  run_encoder(batch);
  HIP_CHECK(hipStreamSynchronize(stream()));
}

// back to the reference's layout [n_text_layer][1500][d] fp32 (export_onnx.py:212-213)
void Engine::get_cross_kv(int slot, float* k_out, float* v_out) {
  std::lock_guard<std::recursive_mutex> capture_lock(device_capture_mutex(device_));
  HIP_CHECK(hipSetDevice(device_));
  if (slot < 0 || slot >= cap_) throw std::runtime_error("slot out of range");
  const int d = cfg_.n_text_state, H = cfg_.n_text_head, L = cfg_.n_text_layer, T = cfg_.n_audio_ctx;
  const size_t per = (size_t)H * t_pad_ * 64;
  std::vector<uint16_t> hk(per), hv(per);
  HIP_CHECK(hipStreamSynchronize(stream()));
  for (int l = 0; l < L; ++l) {
    HIP_CHECK(hipMemcpy(hk.data(), d_cross_k_ + ((size_t)l * cap_ + slot) * per, per * 2, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hv.data(), d_cross_v_ + ((size_t)l * cap_ + slot) * per, per * 2, hipMemcpyDeviceToHost));
    for (int h = 0; h < H; ++h)
      for (int t = 0; t < T; ++t)
        for (int c = 0; c < 64; ++c) {
          const uint16_t kbits = hk[(size_t)h * t_pad_ * 64 + (size_t)(t >> 6) * 4096 + (c >> 3) * 512 + (t & 63) * 8 + (c & 7)];
          const uint16_t vbits = hv[(size_t)h * t_pad_ * 64 + (size_t)t * 64 + c];
          const float kf = h16_bits_to_float(kbits), vf = h16_bits_to_float(vbits);
          k_out[((size_t)l * T + t) * d + h * 64 + c] = kf;
          v_out[((size_t)l * T + t) * d + h * 64 + c] = vf;
        }
  }
}

void Engine::decode_forced(int batch, const int32_t* forced, int n_forced, float* logits, int32_t* argmax_ids) {
  require_no_stream("decode_forced");
  std::lock_guard<std::recursive_mutex> capture_lock(device_capture_mutex(device_));
  HIP_CHECK(hipSetDevice(device_));
  if (batch < 1 || batch > cap_) throw std::runtime_error("decode_forced: batch exceeds the encoded slots");
  if (n_forced < 0 || n_forced + 4 > cfg_.n_text_ctx) throw std::runtime_error("decode_forced: n_forced out of range");
  hipStream_t s = stream();
  const int nv = cfg_.n_vocab, rows = n_forced + 1;
  // device scratch of this call, freed on every path out (a HIP_CHECK below may throw)
  struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
  } b_forced, b_arg, b_logits;
  HIP_CHECK(hipMalloc(&b_forced.p, std::max<size_t>((size_t)batch * n_forced * 4, 256)));
  HIP_CHECK(hipMalloc(&b_arg.p, (size_t)batch * rows * 4));
  if (logits) HIP_CHECK(hipMalloc(&b_logits.p, (size_t)batch * rows * nv * 4));
  int* d_forced = (int*)b_forced.p;
  int* d_arg = (int*)b_arg.p;
  float* d_logits = (float*)b_logits.p;
  if (n_forced) HIP_CHECK(hipMemcpy(d_forced, forced, (size_t)batch * n_forced * 4, hipMemcpyHostToDevice));
  bool done = false;
  if (batch == 1 && persistent_usable()) {
    if (run_persistent(cfg_.n_text_ctx, d_forced, n_forced, d_logits, d_arg) >= 0) { done = true; persistent_succeeded(); }
    else persistent_gave_up();
  }
  if (!done) reset_decode_state(batch);
  if (!done) ensure_branch_streams(batch);
  for (int st = 0; !done && st < 4 + n_forced; ++st) {
    const int gi = st - 3;
    float* lrow = (d_logits && gi >= 0) ? d_logits + (size_t)gi * nv : nullptr;
    enqueue_decode_step(batch, cfg_.n_text_ctx, d_forced, n_forced, lrow, (long)rows * nv, d_arg);
  }
  HIP_CHECK(hipStreamSynchronize(s));
  if (logits) HIP_CHECK(hipMemcpy(logits, d_logits, (size_t)batch * rows * nv * 4, hipMemcpyDeviceToHost));
  if (argmax_ids) HIP_CHECK(hipMemcpy(argmax_ids, d_arg, (size_t)batch * rows * 4, hipMemcpyDeviceToHost));
}

void Engine::decode_greedy(int batch, int max_new, const int* max_new_clip, int32_t* ids, int* n_ids) {
  require_no_stream("decode_greedy");
  HIP_CHECK(hipSetDevice(device_));
  if (batch < 1 || batch > cap_) throw std::runtime_error("decode_greedy: batch exceeds the encoded slots");
  hipStream_t s = stream();
  HIP_CHECK(hipEventRecord(ev_[2], s));
  const int steps = greedy_loop(batch, max_new, max_new_clip);
  fetch_ids(batch, ids, n_ids);
  HIP_CHECK(hipEventRecord(ev_[3], s));
  HIP_CHECK(hipEventSynchronize(ev_[3]));
  timings[0] = timings[1] = 0.f;
  (void)hipEventElapsedTime(&timings[2], ev_[2], ev_[3]);
  timings[3] = timings[2];
  timings[4] = (float)steps;
}

}  // inline namespace AXW_NS

#if AXW_F16
IEngine* make_engine_f16(const std::string& model_type, const std::string& model_path, const std::string& language, int device, int max_batch) {
  return new hf::Engine(model_type, model_path, language, device, max_batch);
}
#else
IEngine* make_engine_bf16(const std::string& model_type, const std::string& model_path, const std::string& language, int device, int max_batch) {
  return new bf::Engine(model_type, model_path, language, device, max_batch);
}
#endif
}  // namespace axw
