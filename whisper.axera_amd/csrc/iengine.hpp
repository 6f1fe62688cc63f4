// iengine.hpp — the dtype-independent face of the engine, as the C ABI (api.cpp) sees it.
//
// engine.{hpp,cpp} and every kernel file are compiled twice (bfloat16 and IEEE-half storage, see common.hpp); each
// build defines axw::<ns>::Engine : IEngine and one factory below. api.cpp picks the factory from the dtype of the
// model's weights file, so one libax_whisper.so serves "Whisper-small bf16" and "Whisper-turbo fp16" alike.
#pragma once

#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace axw {

struct ModelConfig {
  int n_mels = 0, n_audio_ctx = 1500, n_audio_state = 0, n_audio_head = 0, n_audio_layer = 0;
  int n_vocab = 0, n_text_ctx = 448, n_text_state = 0, n_text_head = 0, n_text_layer = 0;
  int sot = 0, eot = 0, transcribe = 0, translate = 0, no_timestamps = 0;
  std::vector<int> lang_tokens;
  std::vector<std::string> lang_codes;
  std::map<std::string, long> ints;  // every integer-valued key of the config file
};

class IEngine {
 public:
  virtual ~IEngine() {}
  // full path, host PCM or device PCM; ids [batch][n_text_ctx], n_ids [batch]
  // max_new_clip: optional host [batch] per-clip id budgets (<= 0: none), each capped by max_new
  virtual void run_tokens(const float* const* pcm, const float* d_pcm, int d_stride, const int* n_samples, int batch, int max_new,
                          int32_t* ids, int* n_ids, const int* max_new_clip = nullptr) = 0;
  virtual std::string detokenize(const int32_t* ids, int n) const = 0;
  // detokenize + the reference's zh post-pass (Traditional -> Simplified, Whisper.cpp:231-236) when its OpenCC data files were found
  virtual std::string transcript(const int32_t* ids, int n) const = 0;
  // stage-level
  virtual void compute_mel(const float* pcm, int n_samples, float* mel_out) = 0;
  virtual void encode_mel(const float* mel, int batch) = 0;
  virtual void get_cross_kv(int slot, float* k_out, float* v_out) = 0;
  virtual void decode_forced(int batch, const int32_t* forced, int n_forced, float* logits, int32_t* argmax_ids) = 0;
  // max_new_clip: optional host [batch] per-clip id budgets (<= 0: none), each capped by max_new
  virtual void decode_greedy(int batch, int max_new, const int* max_new_clip, int32_t* ids, int* n_ids) = 0;
  // utterance slots refilled while the others decode (include/ax_whisper_api.h: AX_WHISPER_Stream*)
  virtual void stream_open(int n_slots) = 0;
  virtual void stream_admit(const int* slots, const float* const* pcm, const int* n_samples, const int* max_new, int count) = 0;
  virtual int stream_step(int n_steps, int* finished_slots) = 0;  // returns the number of finished slots written
  virtual void stream_collect(int slot, int32_t* ids, int* n_ids) = 0;
  virtual void stream_close() = 0;
  // every 16-bit tensor the engine STORES between kernels (encoder activations of `batch` clips, cross / self K/V caches, the
  // decoder's activation pairs): non-finite count and max |x| per buffer; returns the number of buffers reported (<= n_max)
  virtual int scan_stored16(int batch, int n_max, char (*names)[32], long long* nonfinite, float* maxabs) = 0;
  virtual float bench(const std::string& what, int batch, int arg, int iters) = 0;
  virtual void set_stream(void* hip_stream) = 0;
  virtual const ModelConfig& config() const = 0;
  virtual const char* dtype_name() const = 0;  // "bf16" | "fp16"

  std::mutex& mutex() { return mu_; }
  float timings[5] = {0, 0, 0, 0, 0};

 protected:
  std::mutex mu_;
};

// One persistent decode launch at a time per GPU, whichever build (bfloat16 / half) the launching handle belongs to:
// the launch needs every CU, and two of them co-resident would each hold part of the chip until both give up. Defined
// once in api.cpp (the engine's translation units are compiled twice).
std::mutex& persistent_launch_mutex(int device);

// Stream capture of a decoder step (hipStreamBeginCapture ... EndCapture) and the calls that may invalidate SOMEBODY ELSE's
// capture on the same device (allocation, synchronous copies, device-wide synchronisation: engine construction, capacity
// growth, StreamOpen, the stage-level entry points) take this mutex — ONE for the whole process (round 4; it was one per
// device), whichever handle, device or build they belong to: two handles on one GPU — whisper_srv --devices 0,0, or a bf16
// and an fp16 model side by side — would otherwise break each other's capture ("operation failed due to a previous error
// during capture"), and whether handles on DIFFERENT GPUs can was never observable on a one-GPU box. Held for
// milliseconds, a few times per handle lifetime (engine construction: for its whole duration). Defined once in api.cpp.
std::recursive_mutex& device_capture_mutex(int device);

IEngine* make_engine_bf16(const std::string& model_type, const std::string& model_path, const std::string& language, int device, int max_batch);
IEngine* make_engine_f16(const std::string& model_type, const std::string& model_path, const std::string& language, int device, int max_batch);

}  // namespace axw
