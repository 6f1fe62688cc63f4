// engine.cpp — host side of the MI355X Whisper engine (see engine.hpp for what it replaces).
#include "engine.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "host_io.hpp"

namespace axw {
inline namespace AXW_NS {

#define HIP_CHECK(expr)                                                                                  \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess)                                                                                \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr);       \
  } while (0)

// Launch-per-row-block form of the batched vocabulary projection (used where the register-resident form does not fit:
// d_model 1280 beyond 48 clips): weight-row tiles of 16 rows per workgroup, two per wave (1 / 2 / 4 measured alike).
static int logits_rt() { return 2; }

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
  for (auto& e : ev_ring_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : ev_step_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&ev_upload_, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
  for (auto& e : ev_join_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

  const std::string dir = model_path + "/" + model_type;
  load_config(dir, model_type, language);
  tokens_ = load_token_table(dir + "/" + model_type + "-tokens.txt");
  load_t2s(model_path);
  load_weights(dir + "/" + model_type + ".safetensors");

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
  cfg_.ints["persistent_giveups"] = 0;
  {  // batched decode as clip-block GEMMs with LayerNorm prologue / residual epilogue (enqueue_decode_step_batched)
    const char* e = getenv("AX_WHISPER_BATCHED_LN");
    const int d = cfg_.n_text_state;
    // measured on MI355X: faster for d_model 768 at 16-64 clips (+2..8 %); slower for 1280 at 16-32 clips (-3..6 %, also
    // with eight k-steps in flight per wave) and equal at 64, which keeps the split-K sequence
    batched_ln_ = !(e && e[0] == '0') && d % 128 == 0 && (d <= 1024 || (e && e[0] == '2' && d <= 1280));  // '2': force (A/B runs)
    cfg_.ints["batched_ln"] = batched_ln_ ? 1 : 0;
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

void Engine::load_weights(const std::string& path) {
  SafeTensors st(path);
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
    dec_packed_[i].w_qkv = pack(dec_[i].w_qkv, 3 * d, d);
    dec_packed_[i].w_o = pack(dec_[i].w_o, d, d);
    dec_packed_[i].w_cq = pack(dec_[i].w_cq, d, d);
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
    a.done = d_forced ? nullptr : d_done_;
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
void Engine::enqueue_layers_cblock(int b0, int nb, hipStream_t s, bool forced) {
  const int d = cfg_.n_text_state, H = cfg_.n_text_head, L = cfg_.n_text_layer, Tc = cfg_.n_text_ctx;
  const long self_stride = (long)H * Tc * 64, cross_stride = (long)H * t_pad_ * 64;
  const long frag0 = (long)(b0 / 16) * 512;  // fragment-major pair layouts: clip blocks are 512 elements apart within a k-step
  float* x = d_xdec_ + (long)b0 * d;
  float* qd = d_qdec_ + (long)b0 * d;
  h16 *att_hi = d_att_[0] + frag0, *att_lo = d_att_[1] + frag0, *hid_hi = d_hidp_[0] + frag0, *hid_lo = d_hidp_[1] + frag0;
  const int* done = forced ? nullptr : d_done_ + b0;
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
  int cross_split = 1;
  {
    const int blocks = t_pad_ / 64;
    for (int c : {6, 4, 3, 2})
      if (c <= kCrossSplitMax && blocks % c == 0 && nb * H * c <= 320) { cross_split = c; break; }
    if (cross_split_env_ > 0 && cross_split_env_ <= kCrossSplitMax && blocks % cross_split_env_ == 0) cross_split = cross_split_env_;
  }
  auto cgo = [&](const DecCGemmParams& c) { if (step_mask_ & 1) launch_decode_cgemm(c, s); };
  auto attn = [&](const h16* kc, const h16* vc, long stride, int n_keys, int cap_blocks) {
    DecAttnParams a{};
    a.q = qd; a.k = kc; a.v = vc; a.kv_batch_stride = stride; a.part = nullptr; a.n_split = 1;
    a.batch = nb; a.n_head = H; a.d_model = d; a.n_keys = n_keys; a.cap_blocks = cap_blocks; a.state = d_state_;
    a.off = d_off_ + b0;
    a.done = done;
    a.out_hi = att_hi; a.out_lo = att_lo; a.nbs = nbs_;
    return a;
  };
  const int n_blk = (nb + 15) / 16;
  // two row tiles per workgroup where one would make more workgroups than can be resident at once
  auto rt_for = [&](int N) { return (N / 16) * n_blk > 512 ? 2 : 1; };

  for (int l = 0; l < L; ++l) {
    const DecLayerW& w = dec_[l];
    const DecLayerWP& wq = dec_packed_[l];
    h16* sk = d_self_k_ + ((size_t)l * cap_ + b0) * self_stride;
    h16* sv = d_self_v_ + ((size_t)l * cap_ + b0) * self_stride;
    const h16* ck = d_cross_k_ + ((size_t)l * cap_ + b0) * cross_stride;
    const h16* cv = d_cross_v_ + ((size_t)l * cap_ + b0) * cross_stride;
    DecCGemmParams c = cgemm(wq.w_qkv, w.b_qkv, 3 * d, d, GEPI_QKV_CACHE, rt_for(3 * d));
    c.x = x; c.ln_w = w.attn_ln_w; c.ln_b = w.attn_ln_b;
    c.out = qd; c.k_cache = sk; c.v_cache = sv; c.kv_batch_stride = self_stride;
    cgo(c);
    if (step_mask_ & 2) { DecAttnParams a = attn(sk, sv, self_stride, -1, Tc / 64); a.stamp = next_stamp(l, 0, b0, nb); launch_decode_attention(a, s); }
    c = cgemm(wq.w_o, w.b_o, d, d, GEPI_RESID, 1);
    c.a_hi = att_hi; c.a_lo = att_lo; c.out = x;
    cgo(c);
    if (fuse_cq && d <= 1024) {  // the cross-attention workgroups project their own queries (decode_attention_kernel<true>)
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
      cgo(c);
      if (step_mask_ & 2) { DecAttnParams a = attn(ck, cv, cross_stride, cfg_.n_audio_ctx, t_pad_ / 64); a.stamp = next_stamp(l, 1, b0, nb); launch_decode_attention(a, s); }
    }
    c = cgemm(wq.w_co, w.b_co, d, d, GEPI_RESID, 1);
    c.a_hi = att_hi; c.a_lo = att_lo; c.out = x;
    cgo(c);
    c = cgemm(wq.w_fc1, w.b_fc1, 4 * d, d, GEPI_GELU, rt_for(4 * d));
    c.x = x; c.ln_w = w.mlp_ln_w; c.ln_b = w.mlp_ln_b; c.out_hi = hid_hi; c.out_lo = hid_lo;
    cgo(c);
    c = cgemm(wq.w_fc2, w.b_fc2, d, 4 * d, GEPI_RESID, 1);
    c.a_hi = hid_hi; c.a_lo = hid_lo; c.out = x;
    cgo(c);
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
    a.done = d_forced ? nullptr : d_done_;
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
      enqueue_layers_cblock(0, batch, s, d_forced != nullptr);
    } else {
      const int per = ((batch + nbr - 1) / nbr + 15) / 16 * 16;
      HIP_CHECK(hipEventRecord(ev_fork_, s));
      for (int i = 0; i < nbr; ++i) {
        const int b0 = i * per, nb = std::min(per, batch - b0);
        if (nb <= 0) break;
        hipStream_t bs = i == 0 ? s : branch_stream_[i - 1];
        if (i > 0) HIP_CHECK(hipStreamWaitEvent(bs, ev_fork_, 0));
        enqueue_layers_cblock(b0, nb, bs, d_forced != nullptr);
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
  hipGraphExec_t exec = nullptr;
  HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  HIP_CHECK(hipGraphDestroy(graph));
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
  if (batch >= 2 && batch <= persist_max_clips_ && persistent_usable()) {
    int mn[3] = {max_new, -1, -1};
    for (int b = 0; b < batch; ++b) mn[b] = (max_new_clip && max_new_clip[b] > 0) ? std::min(max_new, max_new_clip[b]) : max_new;
    const int st = run_persistent(mn[0], nullptr, 0, nullptr, nullptr, 0, mn[1], mn[2]);
    if (st >= 0) { persistent_succeeded(); return st; }
    persistent_gave_up();
  } else if (batch <= 2 && persistent_usable()) {
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
  reset_decode_state(batch, max_new_clip);
  hipGraphExec_t g = step_graph(batch, max_new);
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
  (void)step_graph(n, cfg_.n_text_ctx - 4);  // captured here, outside the serving loop
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

float Engine::bench(const std::string& what, int batch, int arg, int iters) {
  require_no_stream("bench");
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
      const double bytes = (double)m.nb * 2.0 * 2.0 * d_ * (m.cross ? (double)cfg_.n_audio_ctx : keys_self);
      iv.push_back({bg, en});
      snprintf(line, sizeof line, "%zu,%s,%d,%d,%d,%.2f,%.2f,%.2f,%.0f,%.1f\n", i, m.cross ? "cross" : "self", m.layer, m.b0, m.nb, bg, en, en - bg, bytes,
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
