// api.cpp — the C ABI of libax_whisper.so (include/ax_whisper_api.h).
//
// The four legacy entry points keep the reference's contract (cpp/src/api/ax_whisper_api.cpp:
// 48-56 Init, 69-74 Uninit, 88-124 RunFile, 139-163 RunPCM): NULL / -1 on failure, *result set to
// NULL before any failure after the argument checks, result strdup'd for the caller to free().
// Differences, all fixes of reference defects: no exception crosses the ABI
// (json / file errors become NULL), a failed Init does not leak, a handle is serialised by a
// mutex (the reference's handle is not re-entrant although whisper_srv calls it from a thread
// pool). The OpenCC Traditional->Simplified pass of zh transcripts (Whisper.cpp:231-236) is applied by t2s.hpp when
// the reference's t2s.json + .ocd2 dictionaries are found; AX_WHISPER_Detokenize returns the raw bytes.
#include "../../include/ax_whisper_api.h"

#include <climits>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "iengine.hpp"
#include "host_io.hpp"
#include "t2s.hpp"
#include "multi_device.hpp"

using Engine = axw::IEngine;

namespace axw {
std::recursive_mutex& device_capture_mutex(int device) {
  // Default: ONE mutex for the whole process, not one per device: whether an allocation on device 1 can invalidate a
  // thread-local capture on device 0 was never observable on a one-GPU box, so the exclusion does not depend on the answer.
  // The cost is that engines of one AX_WHISPER_InitMulti handle are constructed one after the other (seconds, once per
  // handle), and that creating a handle at run time stalls StreamOpen / capacity growth / first-time graph capture of every
  // serving handle for that long (INTEGRATION.md "Threading"). AX_WHISPER_CAPTURE_MUTEX=device: one mutex per device — for
  // multi-GPU hosts once the first 8-GPU run has shown that captures on different devices do not disturb each other.
  static std::recursive_mutex mu[65];
  static const bool per_device = [] { const char* e = getenv("AX_WHISPER_CAPTURE_MUTEX"); return e && !strcmp(e, "device"); }();
  return per_device ? mu[1 + (device & 63)] : mu[0];
}
std::mutex& persistent_launch_mutex(int device) {
  static std::mutex mu[64];  // one per device: engines of different GPUs never wait for each other
  return mu[device & 63];
}
}  // namespace axw

namespace {
thread_local std::string g_init_error;
struct Handle {
  axw::DeviceGroup<Engine> group;  // one engine per device; the legacy entry points create exactly one
  std::string last_error;
  std::mutex err_mu;
  void set_error(const std::string& e) {
    std::lock_guard<std::mutex> lk(err_mu);
    last_error = e;
  }
};
inline Handle* H(AX_WHISPER_HANDLE h) { return static_cast<Handle*>(h); }

// f(primary engine) under that engine's mutex; exceptions become -1 + last_error
template <typename F>
int guarded(AX_WHISPER_HANDLE handle, F&& f) {
  Handle* h = H(handle);
  if (!h || h->group.size() == 0) return -1;
  try {
    Engine& e = h->group.primary();
    std::lock_guard<std::mutex> lock(e.mutex());
    f(e);
    return 0;
  } catch (const std::exception& e) {
    h->set_error(e.what());
    fprintf(stderr, "[ax_whisper] %s\n", e.what());
    return -1;
  } catch (...) {
    h->set_error("unknown error");
    return -1;
  }
}

// f(device group): the sharded entry points; every shard locks its own engine (multi_device.hpp)
template <typename F>
int guarded_group(AX_WHISPER_HANDLE handle, F&& f) {
  Handle* h = H(handle);
  if (!h || h->group.size() == 0) return -1;
  try {
    f(h->group);
    return 0;
  } catch (const std::exception& e) {
    h->set_error(e.what());
    fprintf(stderr, "[ax_whisper] %s\n", e.what());
    return -1;
  } catch (...) {
    h->set_error("unknown error");
    return -1;
  }
}

int visible_devices() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    throw std::runtime_error("no HIP device visible: the MI355X engine has no CPU fallback");
  return n;
}
AX_WHISPER_HANDLE init_devices(const char* model_type, const char* model_path, const char* language,
                               const std::vector<int>& devices, int max_batch) {
  std::unique_ptr<Handle> h(new Handle());
  const int G = (int)devices.size();
  (void)visible_devices();  // no GPU: fail with that message, before any file is touched (there is no CPU fallback)
  std::vector<std::unique_ptr<Engine>> engines(G);
  // engines load side by side (one host thread per device): each uploads its own replica of the weights
  // 16-bit storage type of the engine: AX_WHISPER_DTYPE=bf16|fp16 when set, else the dtype of the weights file
  // (F16 -> half; BF16 and F32 -> bfloat16)
  const std::string wfile = std::string(model_path) + "/" + model_type + "/" + model_type + ".safetensors";
  bool f16 = false;
  if (const char* e = getenv("AX_WHISPER_DTYPE")) {
    const std::string v = e;
    if (v == "fp16" || v == "f16" || v == "half") f16 = true;
    else if (v != "bf16") throw std::runtime_error("AX_WHISPER_DTYPE must be bf16 or fp16");
  } else {
    axw::SafeTensors st(wfile);
    f16 = st.get("decoder.token_embedding.weight").dtype == "F16";
  }
  axw::run_sharded(G, G, [&](int w, int, int) {
    engines[w].reset(f16 ? axw::make_engine_f16(model_type, model_path, language, devices[w], max_batch)
                         : axw::make_engine_bf16(model_type, model_path, language, devices[w], max_batch));
  });
  for (auto& e : engines) h->group.add(std::move(e));
  return h.release();  // a throw above destroys every engine already built (the reference leaks: ax_whisper_api.cpp:49-53)
}

template <typename F>
AX_WHISPER_HANDLE init_guarded(F&& f) {
  try {
    return f();
  } catch (const std::exception& e) {
    g_init_error = e.what();
    fprintf(stderr, "[ax_whisper] init failed: %s\n", e.what());
    return nullptr;
  } catch (...) {
    g_init_error = "unknown error";
    return nullptr;
  }
}

}  // namespace

extern "C" {

AX_WHISPER_API AX_WHISPER_HANDLE AX_WHISPER_InitMulti(const char* model_type, const char* model_path, const char* language,
                                                      const int* devices, int n_devices, int max_batch_per_device) {
  if (!model_type || !model_path || !language) {
    g_init_error = "null argument";
    return nullptr;
  }
  return init_guarded([&]() -> AX_WHISPER_HANDLE {
    const int n_vis = visible_devices();
    std::vector<int> devs;
    if (devices && n_devices > 0) {
      // test hook: AX_WHISPER_ALLOW_DUPLICATE_DEVICES=1 lets a one-GPU box run several engines (the sharding, the
      // worker threads, the join) against real devices; production lists must name each device once
      const char* dup = getenv("AX_WHISPER_ALLOW_DUPLICATE_DEVICES");
      if (dup && dup[0] == '1') {
        for (int i = 0; i < n_devices; ++i) {
          if (devices[i] < 0 || devices[i] >= n_vis) throw std::runtime_error("device ordinal out of range");
          devs.push_back(devices[i]);
        }
      } else {
        std::string list;
        for (int i = 0; i < n_devices; ++i) list += (i ? "," : "") + std::to_string(devices[i] < 0 ? n_vis : devices[i]);
        devs = axw::parse_device_list(list, n_vis);  // range + duplicate checks
      }
    } else {
      const char* e = getenv("AX_WHISPER_DEVICES");
      devs = axw::parse_device_list(e ? e : "all", n_vis);
    }
    return init_devices(model_type, model_path, language, devs, max_batch_per_device);
  });
}

AX_WHISPER_API AX_WHISPER_HANDLE AX_WHISPER_InitEx(const char* model_type, const char* model_path, const char* language,
                                                   int device, int max_batch) {
  if (!model_type || !model_path || !language) {
    g_init_error = "null argument";
    return nullptr;
  }
  // device < 0 and AX_WHISPER_DEVICES set ("all" or "0,1,..."): the unchanged callers of the legacy Init (whisper_cli,
  // a reference-side application) get one engine per listed device without a source change
  if (device < 0 && getenv("AX_WHISPER_DEVICES")) return AX_WHISPER_InitMulti(model_type, model_path, language, nullptr, 0, max_batch);
  return init_guarded([&]() -> AX_WHISPER_HANDLE { return init_devices(model_type, model_path, language, {device}, max_batch); });
}

AX_WHISPER_API AX_WHISPER_HANDLE AX_WHISPER_Init(const char* model_type, const char* model_path, const char* language) {
  return AX_WHISPER_InitEx(model_type, model_path, language, -1, 0);
}

AX_WHISPER_API void AX_WHISPER_Uninit(AX_WHISPER_HANDLE handle) {
  delete H(handle);  // NULL-safe (ax_whisper_api.cpp:69-74); the group destroys its engines
}

AX_WHISPER_API int AX_WHISPER_VisibleDeviceCount(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

AX_WHISPER_API int AX_WHISPER_GetDeviceCount(AX_WHISPER_HANDLE handle) {
  Handle* h = H(handle);
  return h ? h->group.size() : -1;
}

AX_WHISPER_API int AX_WHISPER_RunPCM(AX_WHISPER_HANDLE handle, float* pcm_data, int num_samples, char** result) {
  if (!handle || !pcm_data || !result) return -1;
  *result = nullptr;
  return AX_WHISPER_RunPCMBatch(handle, &pcm_data, &num_samples, 1, result);
}

AX_WHISPER_API int AX_WHISPER_RunFile(AX_WHISPER_HANDLE handle, const char* wav_file, char** result) {
  if (!handle || !wav_file || !result) return -1;
  *result = nullptr;
  axw::WavData wav;
  std::string err;
  if (!axw::load_audio_file(wav_file, wav, err)) {
    H(handle)->set_error("load wav failed: " + err);
    fprintf(stderr, "[ax_whisper] load wav failed: %s\n", err.c_str());
    return -1;
  }
  if (wav.mono.empty()) {
    H(handle)->set_error("wav file holds no samples");
    return -1;
  }
  if (wav.sample_rate != 16000)  // the reference silently mis-transcribes (no resampler anywhere in its C++)
    fprintf(stderr, "[ax_whisper] warning: %s is %d Hz, expected 16000 Hz (see cpp/resample_wav.sh of the reference)\n",
            wav_file, wav.sample_rate);
  return AX_WHISPER_RunPCM(handle, wav.mono.data(), (int)wav.mono.size(), result);
}

AX_WHISPER_API int AX_WHISPER_RunPCMBatchTokens(AX_WHISPER_HANDLE handle, const float* const* pcm, const int* num_samples,
                                                int batch, int max_new, int32_t* ids, int* n_ids) {
  if (!handle || !pcm || !num_samples || !ids || !n_ids || batch < 1) return -1;
  return guarded_group(handle, [&](axw::DeviceGroup<Engine>& g) {
    g.run_tokens(pcm, num_samples, batch, max_new, g.primary().config().n_text_ctx, ids, n_ids);
  });
}

AX_WHISPER_API int AX_WHISPER_RunDeviceBatchTokens(AX_WHISPER_HANDLE handle, const float* d_pcm, int stride,
                                                   const int* num_samples, int batch, int max_new, int32_t* ids, int* n_ids) {
  if (!handle || !d_pcm || !num_samples || !ids || !n_ids || batch < 1) return -1;
  return guarded(handle, [&](Engine& e) { e.run_tokens(nullptr, d_pcm, stride, num_samples, batch, max_new, ids, n_ids); });
}

AX_WHISPER_API int AX_WHISPER_RunDeviceBatchTokensRagged(AX_WHISPER_HANDLE handle, const float* d_pcm, int stride,
                                                         const int* num_samples, int batch, int max_new,
                                                         const int* max_new_clip, int32_t* ids, int* n_ids) {
  if (!handle || !d_pcm || !num_samples || !ids || !n_ids || batch < 1) return -1;
  return guarded(handle, [&](Engine& e) { e.run_tokens(nullptr, d_pcm, stride, num_samples, batch, max_new, ids, n_ids, max_new_clip); });
}

AX_WHISPER_API int AX_WHISPER_RunPCMBatch(AX_WHISPER_HANDLE handle, const float* const* pcm, const int* num_samples, int batch,
                                          char** results) {
  if (!handle || !pcm || !num_samples || !results || batch < 1) return -1;
  for (int b = 0; b < batch; ++b) results[b] = nullptr;
  return guarded_group(handle, [&](axw::DeviceGroup<Engine>& g) {
    Engine& e = g.primary();
    const int Tc = e.config().n_text_ctx;
    std::vector<int32_t> ids((size_t)batch * Tc);
    std::vector<int> n(batch);
    g.run_tokens(pcm, num_samples, batch, 0, Tc, ids.data(), n.data());
    for (int b = 0; b < batch; ++b) results[b] = strdup(e.transcript(ids.data() + (size_t)b * Tc, n[b]).c_str());  // host only
  });
}

AX_WHISPER_API int AX_WHISPER_Detokenize(AX_WHISPER_HANDLE handle, const int32_t* ids, int n, char** result) {
  if (!handle || (!ids && n > 0) || !result) return -1;
  *result = nullptr;
  return guarded(handle, [&](Engine& e) { *result = strdup(e.detokenize(ids, n).c_str()); });
}

AX_WHISPER_API int AX_WHISPER_Transcript(AX_WHISPER_HANDLE handle, const int32_t* ids, int n, char** result) {
  if (!handle || (!ids && n > 0) || !result) return -1;
  *result = nullptr;
  return guarded(handle, [&](Engine& e) { *result = strdup(e.transcript(ids, n).c_str()); });
}

AX_WHISPER_API int AX_WHISPER_ConvertT2S(const char* config_path, const char* text, char** result) {
  if (!config_path || !text || !result) return -1;
  *result = nullptr;
  try {
    axw::T2SConverter conv(config_path);
    *result = strdup(conv.convert(text).c_str());
    return *result ? 0 : -1;
  } catch (const std::exception& e) {
    g_init_error = e.what();
    return -1;
  } catch (...) {
    g_init_error = "unknown error";
    return -1;
  }
}

// the two byte paths of the drop-in boundary on their own, host only (no handle, no GPU): parity tests hold them bit-equal to
// the reference's AudioFile.h / base64.cpp compiled into oracle/_ref (tests/test_byte_paths.py)
AX_WHISPER_API int AX_WHISPER_LoadAudioFile(const char* path, float** samples, int* n_samples, int* info) {
  if (!path || !samples || !n_samples) return -1;
  *samples = nullptr;
  *n_samples = 0;
  try {
    axw::WavData wav;
    std::string err;
    if (!axw::load_audio_file(path, wav, err)) { g_init_error = "load wav failed: " + err; return -1; }
    *samples = static_cast<float*>(malloc(std::max<size_t>(wav.mono.size(), 1) * sizeof(float)));
    if (!*samples) return -1;
    memcpy(*samples, wav.mono.data(), wav.mono.size() * sizeof(float));
    *n_samples = (int)wav.mono.size();
    if (info) { info[0] = wav.sample_rate; info[1] = wav.channels; }
    return 0;
  } catch (const std::exception& e) {
    g_init_error = e.what();
    return -1;
  } catch (...) {
    g_init_error = "unknown error";
    return -1;
  }
}

AX_WHISPER_API int AX_WHISPER_DetokenizeWithTable(const char* tokens_path, const int32_t* ids, int n, char** result, int* n_bytes) {
  if (!tokens_path || (n > 0 && !ids) || !result || !n_bytes) return -1;
  *result = nullptr;
  *n_bytes = 0;
  try {
    const std::vector<std::string> table = axw::load_token_table(tokens_path);
    std::string s;
    for (int i = 0; i < n; ++i)
      if (ids[i] >= 0 && (size_t)ids[i] < table.size()) s += table[(size_t)ids[i]];
    *result = static_cast<char*>(malloc(s.size() + 1));
    if (!*result) return -1;
    memcpy(*result, s.data(), s.size());
    (*result)[s.size()] = 0;
    *n_bytes = (int)s.size();
    return 0;
  } catch (const std::exception& e) {
    g_init_error = e.what();
    return -1;
  } catch (...) {
    g_init_error = "unknown error";
    return -1;
  }
}

AX_WHISPER_API int AX_WHISPER_GetConfigInt(AX_WHISPER_HANDLE handle, const char* key) {
  Handle* h = H(handle);
  if (!h || !key || h->group.size() == 0) return INT_MIN;
  if (!strcmp(key, "n_devices")) return h->group.size();
  if (!strcmp(key, "persistent_giveups")) {  // a count: summed over the handle's engines
    long sum = 0;
    for (int i = 0; i < h->group.size(); ++i) {
      Engine& e = h->group.at(i);
      std::lock_guard<std::mutex> lock(e.mutex());
      auto it = e.config().ints.find(key);
      if (it != e.config().ints.end()) sum += it->second;
    }
    return (int)sum;
  }
  Engine& e = h->group.primary();
  std::lock_guard<std::mutex> lock(e.mutex());  // a few values change while the engine runs (persistent_decode, ...)
  auto& m = e.config().ints;
  auto it = m.find(key);
  return it == m.end() ? INT_MIN : (int)it->second;
}

AX_WHISPER_API const char* AX_WHISPER_LastError(AX_WHISPER_HANDLE handle) {
  Handle* h = H(handle);
  if (!h) return g_init_error.c_str();
  // a copy taken under the lock: server threads and device workers may set_error() concurrently, and the pointer handed
  // out must not dangle when they do (valid until this thread's next call)
  thread_local std::string copy;
  {
    std::lock_guard<std::mutex> lk(h->err_mu);
    copy = h->last_error;
  }
  return copy.c_str();
}

AX_WHISPER_API int AX_WHISPER_SetStream(AX_WHISPER_HANDLE handle, void* hip_stream) {
  return guarded(handle, [&](Engine& e) { e.set_stream(hip_stream); });
}

AX_WHISPER_API int AX_WHISPER_ComputeMel(AX_WHISPER_HANDLE handle, const float* pcm, int num_samples, float* mel_out) {
  if (!handle || !pcm || !mel_out || num_samples < 1) return -1;
  return guarded(handle, [&](Engine& e) { e.compute_mel(pcm, num_samples, mel_out); });
}

AX_WHISPER_API int AX_WHISPER_EncodeMel(AX_WHISPER_HANDLE handle, const float* mel, int batch) {
  if (!handle || !mel || batch < 1) return -1;
  return guarded(handle, [&](Engine& e) { e.encode_mel(mel, batch); });
}

AX_WHISPER_API int AX_WHISPER_GetCrossKV(AX_WHISPER_HANDLE handle, int slot, float* k_out, float* v_out) {
  if (!handle || !k_out || !v_out) return -1;
  return guarded(handle, [&](Engine& e) { e.get_cross_kv(slot, k_out, v_out); });
}

AX_WHISPER_API int AX_WHISPER_ScanStored16(AX_WHISPER_HANDLE handle, int batch, int n_max, char* names, int64_t* nonfinite,
                                           float* maxabs, int* n_out) {
  if (!handle || !names || !nonfinite || !maxabs || !n_out || n_max < 1) return -1;
  return guarded(handle, [&](Engine& e) {
    static_assert(sizeof(long long) == sizeof(int64_t), "int64_t");
    *n_out = e.scan_stored16(batch, n_max, reinterpret_cast<char(*)[32]>(names), reinterpret_cast<long long*>(nonfinite), maxabs);
  });
}

AX_WHISPER_API int AX_WHISPER_DecodeForced(AX_WHISPER_HANDLE handle, int batch, const int32_t* forced, int n_forced,
                                           float* logits, int32_t* argmax_ids) {
  if (!handle || (n_forced > 0 && !forced)) return -1;
  return guarded(handle, [&](Engine& e) { e.decode_forced(batch, forced, n_forced, logits, argmax_ids); });
}

AX_WHISPER_API int AX_WHISPER_DecodeGreedy(AX_WHISPER_HANDLE handle, int batch, int max_new, int32_t* ids, int* n_ids) {
  return AX_WHISPER_DecodeGreedyRagged(handle, batch, max_new, nullptr, ids, n_ids);
}

AX_WHISPER_API int AX_WHISPER_DecodeGreedyRagged(AX_WHISPER_HANDLE handle, int batch, int max_new, const int* max_new_clip,
                                                 int32_t* ids, int* n_ids) {
  if (!handle || !ids || !n_ids) return -1;
  return guarded(handle, [&](Engine& e) { e.decode_greedy(batch, max_new, max_new_clip, ids, n_ids); });
}

AX_WHISPER_API int AX_WHISPER_StreamOpen(AX_WHISPER_HANDLE handle, int n_slots) {
  return guarded(handle, [&](Engine& e) { e.stream_open(n_slots); });
}
AX_WHISPER_API int AX_WHISPER_StreamAdmit(AX_WHISPER_HANDLE handle, int slot, const float* pcm, int num_samples, int max_new) {
  if (!handle || !pcm || num_samples < 1) return -1;
  return guarded(handle, [&](Engine& e) { const float* arr[1] = {pcm}; e.stream_admit(&slot, arr, &num_samples, &max_new, 1); });
}
AX_WHISPER_API int AX_WHISPER_StreamAdmitBatch(AX_WHISPER_HANDLE handle, const int* slots, const float* const* pcm,
                                               const int* num_samples, const int* max_new, int count) {
  if (!handle || !slots || !pcm || !num_samples || count < 1) return -1;
  for (int i = 0; i < count; ++i) if (!pcm[i] || num_samples[i] < 1) return -1;
  return guarded(handle, [&](Engine& e) { e.stream_admit(slots, pcm, num_samples, max_new, count); });
}
AX_WHISPER_API int AX_WHISPER_StreamStep(AX_WHISPER_HANDLE handle, int n_steps, int* finished_slots, int* n_finished) {
  if (!handle || !finished_slots || !n_finished) return -1;
  *n_finished = 0;
  return guarded(handle, [&](Engine& e) { *n_finished = e.stream_step(n_steps, finished_slots); });
}
AX_WHISPER_API int AX_WHISPER_StreamCollect(AX_WHISPER_HANDLE handle, int slot, int32_t* ids, int* n_ids) {
  if (!handle || !ids || !n_ids) return -1;
  return guarded(handle, [&](Engine& e) { e.stream_collect(slot, ids, n_ids); });
}
AX_WHISPER_API int AX_WHISPER_StreamClose(AX_WHISPER_HANDLE handle) {
  return guarded(handle, [&](Engine& e) { e.stream_close(); });
}

AX_WHISPER_API int AX_WHISPER_GetTimings(AX_WHISPER_HANDLE handle, float* out5) {
  if (!handle || !out5) return -1;
  return guarded(handle, [&](Engine& e) { memcpy(out5, e.timings, sizeof(float) * 5); });
}

AX_WHISPER_API int AX_WHISPER_Bench(AX_WHISPER_HANDLE handle, const char* what, int batch, int arg, int iters, float* ms_total) {
  if (!handle || !what || !ms_total || batch < 1 || iters < 1) return -1;
  return guarded(handle, [&](Engine& e) { *ms_total = e.bench(what, batch, arg, iters); });
}

}  // extern "C"
