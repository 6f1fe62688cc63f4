/**
 * ax_whisper_api.h — C ABI of libax_whisper.so, MI355X (gfx950) build.
 *
 * Drop-in boundary of the reference (paths relative to the reference tree):
 *   AX_WHISPER_Init     replaces cpp/src/api/ax_whisper_api.h:54  (impl ax_whisper_api.cpp:48)
 *   AX_WHISPER_Uninit   replaces cpp/src/api/ax_whisper_api.h:67  (impl :69)
 *   AX_WHISPER_RunFile  replaces cpp/src/api/ax_whisper_api.h:81  (impl :88)
 *   AX_WHISPER_RunPCM   replaces cpp/src/api/ax_whisper_api.h:98  (impl :139)
 * Same names, argument meaning, ownership (malloc'd result, caller free()s) and error
 * behaviour (NULL / -1). The application no longer calls AX_SYS_Init / AX_ENGINE_Init
 * (whisper_cli.cpp:37-61): the library initialises HIP inside Init.
 *
 * Everything below the four legacy symbols is an ADDITION with no reference counterpart:
 * batched entry points (the reference is strictly batch 1, Whisper.hpp:15-25), token-id
 * variants for parity tests, device-resident inputs for benchmarking, and stage-level
 * entry points so each stage can be checked against the CPU oracle.
 *
 * Plain C types only: pointers, ints, sizes. No C++ or torch types cross this boundary.
 */
#ifndef _AX_WHISPER_API_H_
#define _AX_WHISPER_API_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AX_WHISPER_API __attribute__((visibility("default")))

typedef void* AX_WHISPER_HANDLE;

/* ---- legacy (byte-compatible with the reference) ------------------------------------ */

/** model files: {model_path}/{model_type}/{model_type}.safetensors (replaces the two
 *  .axmodel NPU blobs), {model_type}-tokens.txt, {model_type}_config.json.
 *  language not in the config falls back to "zh" (Whisper.cpp:241-251). NULL on failure. */
AX_WHISPER_API AX_WHISPER_HANDLE AX_WHISPER_Init(const char* model_type, const char* model_path,
                                                 const char* language);
AX_WHISPER_API void AX_WHISPER_Uninit(AX_WHISPER_HANDLE handle);
/** 16 kHz WAV (int16/int24/int32/float32 PCM) or AIFF / uncompressed AIFF-C (the reference's AudioFile reads both,
 *  AudioFile.h:450-501); stereo is averaged. 0 ok, -1 error. */
AX_WHISPER_API int AX_WHISPER_RunFile(AX_WHISPER_HANDLE handle, const char* wav_file, char** result);
/** 16 kHz mono f32 PCM in [-1, 1]. *result is malloc'd; caller frees. 0 ok, -1 error. */
AX_WHISPER_API int AX_WHISPER_RunPCM(AX_WHISPER_HANDLE handle, float* pcm_data, int num_samples,
                                     char** result);

/* ---- additions: init / info ---------------------------------------------------------- */

/** device: HIP device ordinal (-1: env AX_WHISPER_DEVICE, else 0). max_batch: number of
 *  utterance slots to allocate (<=0: env AX_WHISPER_MAX_BATCH, else 1; grows on demand). */
AX_WHISPER_API AX_WHISPER_HANDLE AX_WHISPER_InitEx(const char* model_type, const char* model_path,
                                                   const char* language, int device, int max_batch);
/** One engine per device behind ONE handle (utterance-level data parallelism, SURVEY 8e: the reference is one
 *  utterance at a time, Whisper.cpp:186-239). devices: n_devices HIP ordinals; NULL / n_devices <= 0: the list in env
 *  AX_WHISPER_DEVICES ("0,1,4" or "all"), else every visible device. Weights are replicated; RunPCMBatch /
 *  RunPCMBatchTokens split a batch into contiguous blocks of ceil(batch / devices) clips, run every block on its
 *  device from its own host thread and return when all have finished (no collective: each device copies its ids
 *  back itself). Every other entry point uses the first device. The legacy AX_WHISPER_Init does the same when
 *  AX_WHISPER_DEVICES is set, so existing callers need no source change. NULL on failure. */
AX_WHISPER_API AX_WHISPER_HANDLE AX_WHISPER_InitMulti(const char* model_type, const char* model_path,
                                                      const char* language, const int* devices, int n_devices,
                                                      int max_batch_per_device);
/** HIP devices this process can see (0 when there is none): lets a host program that links only this C ABI
 *  (whisper_srv) create one handle per device. */
AX_WHISPER_API int AX_WHISPER_VisibleDeviceCount(void);
/** Engines (devices) behind the handle; -1 for a NULL handle. */
AX_WHISPER_API int AX_WHISPER_GetDeviceCount(AX_WHISPER_HANDLE handle);
/** Integer config value by the key names of {type}_config.json (n_mels, n_vocab, eot, ...),
 *  plus "sot_seq0".."sot_seq3". Returns INT32_MIN for an unknown key. */
AX_WHISPER_API int AX_WHISPER_GetConfigInt(AX_WHISPER_HANDLE handle, const char* key);
/** Last error text of this handle (or of the last failed Init when handle is NULL). */
AX_WHISPER_API const char* AX_WHISPER_LastError(AX_WHISPER_HANDLE handle);
/** Run all device work of this handle on the caller's hipStream_t (NULL: the library's own). */
AX_WHISPER_API int AX_WHISPER_SetStream(AX_WHISPER_HANDLE handle, void* hip_stream);

/* ---- additions: batched / token-id entry points -------------------------------------- */

/** ids: [batch][n_text_ctx] int32 (row b holds n_ids[b] generated ids, eot excluded);
 *  max_new <= 0 means "until eot or context" (Whisper.cpp:219-222). */
AX_WHISPER_API int AX_WHISPER_RunPCMBatchTokens(AX_WHISPER_HANDLE handle, const float* const* pcm,
                                                const int* num_samples, int batch, int max_new,
                                                int32_t* ids, int* n_ids);
/** results: caller-provided array of `batch` char*; each entry malloc'd, caller frees. */
AX_WHISPER_API int AX_WHISPER_RunPCMBatch(AX_WHISPER_HANDLE handle, const float* const* pcm,
                                          const int* num_samples, int batch, char** results);
/** PCM already resident in HBM: d_pcm is a DEVICE pointer to [batch][stride] f32. */
AX_WHISPER_API int AX_WHISPER_RunDeviceBatchTokens(AX_WHISPER_HANDLE handle, const float* d_pcm,
                                                   int stride, const int* num_samples, int batch,
                                                   int max_new, int32_t* ids, int* n_ids);
/** The same with a per-clip id budget (host [batch], may be NULL; <= 0: none), each capped by max_new: a RAGGED batch in one
 *  call — front-end, encoder and the loop — whose clips leave the loop at different steps the way real utterances reach eot
 *  at different steps (bench.py's realistic-length leg: synthetic weights never emit eot by themselves). */
AX_WHISPER_API int AX_WHISPER_RunDeviceBatchTokensRagged(AX_WHISPER_HANDLE handle, const float* d_pcm, int stride,
                                                         const int* num_samples, int batch, int max_new,
                                                         const int* max_new_clip, int32_t* ids, int* n_ids);
/** ids -> bytes (base64 table of {type}-tokens.txt, Whisper.cpp:224-229); ids >= the table
 *  size are skipped. A table entry ends at its first NUL byte, as in the reference (its table load strcpy's the decoded
 *  entry, Whisper.cpp:115-127, and appends it as a C string): the bytes of an entry from a NUL onward are dropped
 *  (id 188 and the like decode to nothing). *result malloc'd. */
AX_WHISPER_API int AX_WHISPER_Detokenize(AX_WHISPER_HANDLE handle, const int32_t* ids, int n, char** result);
/** The zh post-pass of Whisper::run (cpp/src/Whisper.cpp:231-236: opencc::SimpleConverter("t2s.json").Convert)
 *  on its own: config_path names an OpenCC JSON configuration (cpp/t2s.json) whose .ocd2 dictionaries sit next
 *  to it; text is UTF-8. *result malloc'd. Run* apply it themselves for language "zh" when t2s.json is found in
 *  $AX_WHISPER_OPENCC_DIR, the working directory (the reference's rule) or the model directory. Host-only. */
AX_WHISPER_API int AX_WHISPER_ConvertT2S(const char* config_path, const char* text, char** result);

/** The file decode of AX_WHISPER_RunFile on its own, host only (no handle, no GPU): WAV / AIFF -> the mono f32 samples RunFile
 *  feeds the engine (cpp/src/AudioFile.h:450-501,1241-1243 + the stereo average of ax_whisper_api.cpp:105-113). *samples is
 *  malloc'd, the caller frees it; info (may be NULL): [0] sample rate, [1] channels. 0 ok, -1 error. */
AX_WHISPER_API int AX_WHISPER_LoadAudioFile(const char* path, float** samples, int* n_samples, int* info);
/** ids -> bytes through a {type}-tokens.txt file alone, host only (Whisper.cpp:115-127 table load, :224-229 + base64.cpp:84-120
 *  decode): the bytes AX_WHISPER_Detokenize returns, with their count (entries end at their first NUL as there, so the
 *  result holds none; n_bytes saves the caller a strlen). *result malloc'd. */
AX_WHISPER_API int AX_WHISPER_DetokenizeWithTable(const char* tokens_path, const int32_t* ids, int n, char** result, int* n_bytes);

/* ---- additions: stage-level entry points (parity tests, profiling) ------------------- */

/** Whisper::preprocess (Whisper.cpp:151-184) on the GPU. mel_out: host [n_mels*3000] f32. */
AX_WHISPER_API int AX_WHISPER_ComputeMel(AX_WHISPER_HANDLE handle, const float* pcm, int num_samples,
                                         float* mel_out);
/** Encoder + cross-KV projection for `batch` clips given host mels [batch][n_mels*3000];
 *  results stay in the handle's slots 0..batch-1. */
AX_WHISPER_API int AX_WHISPER_EncodeMel(AX_WHISPER_HANDLE handle, const float* mel, int batch);
/** Copy slot's cross K/V back as fp32 [n_text_layer][1500][n_text_state] (reference layout). */
AX_WHISPER_API int AX_WHISPER_GetCrossKV(AX_WHISPER_HANDLE handle, int slot, float* k_out, float* v_out);
/** Parity aid: scan every 16-bit tensor the engine keeps between kernels (encoder activations of clips 0..batch-1, cross and
 *  self K/V caches, the decoder's activation pairs) after EncodeMel / Decode*: names [n_max][32] chars, nonfinite [n_max]
 *  (NaN or Inf elements), maxabs [n_max] (largest finite |x|), *n_out buffers reported. A trained model's outlier channels
 *  and FFN peaks must stay inside the storage type's range (65504 for the fp16 build). */
AX_WHISPER_API int AX_WHISPER_ScanStored16(AX_WHISPER_HANDLE handle, int batch, int n_max, char* names, int64_t* nonfinite,
                                           float* maxabs, int* n_out);
/** Teacher-forced decode over the slots filled by EncodeMel: after the 4 SOT steps feed
 *  forced[b][0..n_forced-1]; logits: host [batch][n_forced+1][n_vocab] f32 (may be NULL);
 *  argmax_ids: host [batch][n_forced+1] (may be NULL). */
AX_WHISPER_API int AX_WHISPER_DecodeForced(AX_WHISPER_HANDLE handle, int batch, const int32_t* forced,
                                           int n_forced, float* logits, int32_t* argmax_ids);
/** Greedy decode over the slots filled by EncodeMel (same loop as RunPCM*). */
AX_WHISPER_API int AX_WHISPER_DecodeGreedy(AX_WHISPER_HANDLE handle, int batch, int max_new,
                                           int32_t* ids, int* n_ids);
/** The same loop over a RAGGED batch: max_new_clip[b] (host, may be NULL; <= 0: none) caps the ids of clip b, so
 *  clips leave the loop at different steps the way real utterances reach eot at different steps
 *  (Whisper.cpp:219-222). A finished clip keeps its slot but no longer streams its K/V. */
AX_WHISPER_API int AX_WHISPER_DecodeGreedyRagged(AX_WHISPER_HANDLE handle, int batch, int max_new,
                                                 const int* max_new_clip, int32_t* ids, int* n_ids);
/** ids -> the text RunPCM* would have returned for them: detokenised bytes + the reference's zh post-pass (OpenCC t2s,
 *  Whisper.cpp:224-236) when its data files were found. *result is malloc'd; the caller frees it. */
AX_WHISPER_API int AX_WHISPER_Transcript(AX_WHISPER_HANDLE handle, const int32_t* ids, int n, char** result);

/* ---- additions: utterance slots that are refilled while the others decode (continuous batching) ---------------
 * The reference decodes ONE utterance per call and stops it at its own eot (Whisper.cpp:207-222); its server hands
 * requests to the handle one by one (WhisperHTTPServer.hpp:37-100). Here every utterance slot has its own decode offset,
 * so a slot whose clip has finished can take the next clip while the other slots decode on: no clip waits for the
 * slowest one of a micro-batch. Primary engine of the handle (one handle per GPU, as whisper_srv --devices runs them).
 *   StreamOpen(n_slots)  all slots idle; the batched entry points above are refused until StreamClose
 *   StreamAdmit(slot..)  front-end + encoder of one clip on a second stream into an idle slot's cross-K/V; the slot joins
 *                        the decode loop at the first StreamStep after its encoder has finished
 *   StreamStep(n_steps)  up to n decoder steps over all slots (captured step graph); finished slots are seen through
 *                        host-mapped flags between two steps (no wait), their successors join at once; returns the slots
 *                        that have finished
 *   StreamCollect(slot)  ids of a finished slot; the slot is idle again */
AX_WHISPER_API int AX_WHISPER_StreamOpen(AX_WHISPER_HANDLE handle, int n_slots);
/** max_new <= 0: until eot or the end of the context. pcm is copied before the call returns. -1 if the slot is busy. */
AX_WHISPER_API int AX_WHISPER_StreamAdmit(AX_WHISPER_HANDLE handle, int slot, const float* pcm, int num_samples, int max_new);
/** `count` clips at once into `count` idle slots (any slots, any order): one front-end + encoder pass for all of them —
 *  a batched pass costs a fraction of count one-clip passes. max_new may be NULL. -1 if any slot is busy (none admitted). */
AX_WHISPER_API int AX_WHISPER_StreamAdmitBatch(AX_WHISPER_HANDLE handle, const int* slots, const float* const* pcm,
                                               const int* num_samples, const int* max_new, int count);
/** finished_slots: host [n_slots]; *n_finished: how many were written. Slots stay "finished" until collected. */
AX_WHISPER_API int AX_WHISPER_StreamStep(AX_WHISPER_HANDLE handle, int n_steps, int* finished_slots, int* n_finished);
/** ids: host [n_text_ctx]. */
AX_WHISPER_API int AX_WHISPER_StreamCollect(AX_WHISPER_HANDLE handle, int slot, int32_t* ids, int* n_ids);
AX_WHISPER_API int AX_WHISPER_StreamClose(AX_WHISPER_HANDLE handle);

/** Stage timings of the last Run* / DecodeGreedy* call, ms (hipEvent): [0] front-end, [1] encoder,
 *  [2] decode loop, [3] whole call (wall), [4] decode steps executed. */
AX_WHISPER_API int AX_WHISPER_GetTimings(AX_WHISPER_HANDLE handle, float* out5);
/** Time `iters` launches of one named piece on the handle's stream with hipEvents; returns
 *  total ms in *ms_total. what: "decode_step" (one captured step graph at decode offset
 *  `arg`), "encoder", "frontend", or a kernel name listed in DESIGN.md. */
AX_WHISPER_API int AX_WHISPER_Bench(AX_WHISPER_HANDLE handle, const char* what, int batch, int arg,
                                    int iters, float* ms_total);

#ifdef __cplusplus
}
#endif

#endif /* _AX_WHISPER_API_H_ */
