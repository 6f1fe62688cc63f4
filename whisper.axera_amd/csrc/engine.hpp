// engine.hpp — the MI355X Whisper engine behind the AX_WHISPER_* C ABI.
//
// Replaces class Whisper (cpp/src/Whisper.hpp:28-59, cpp/src/Whisper.cpp) together with the two
// AxModelRunner NPU executors it owns (cpp/src/ax_model_runner/ax_model_runner.hpp:23-80): model
// directory loading (Whisper.cpp:86-149), preprocess (:151-184), encoder call (:190-195),
// cross-KV hand-off (:260-288, here: none — the encoder writes the decoder's layouts in place),
// the greedy loop (:207-222) and detokenisation (:224-229). The reference handles one utterance at
// a time; this engine runs B utterance slots through every stage as one batch.
#pragma once

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"
#include "iengine.hpp"
#include "t2s.hpp"

namespace axw {
class SafeTensors;  // host_io.hpp
inline namespace AXW_NS {

class Engine final : public IEngine {
 public:
  Engine(const std::string& model_type, const std::string& model_path, const std::string& language, int device, int max_batch);
  ~Engine() override;
  Engine(const Engine&) = delete;

  void run_tokens(const float* const* pcm, const float* d_pcm, int d_stride, const int* n_samples, int batch, int max_new,
                  int32_t* ids, int* n_ids, const int* max_new_clip = nullptr) override;
  std::string detokenize(const int32_t* ids, int n) const override;
  std::string transcript(const int32_t* ids, int n) const override;
  bool has_t2s() const { return (bool)t2s_; }
  void compute_mel(const float* pcm, int n_samples, float* mel_out) override;
  void encode_mel(const float* mel, int batch) override;
  void get_cross_kv(int slot, float* k_out, float* v_out) override;
  void decode_forced(int batch, const int32_t* forced, int n_forced, float* logits, int32_t* argmax_ids) override;
  void decode_greedy(int batch, int max_new, const int* max_new_clip, int32_t* ids, int* n_ids) override;
  void stream_open(int n_slots) override;
  void stream_admit(const int* slots, const float* const* pcm, const int* n_samples, const int* max_new, int count) override;
  int stream_step(int n_steps, int* finished_slots) override;
  void stream_collect(int slot, int32_t* ids, int* n_ids) override;
  void stream_close() override;
  int scan_stored16(int batch, int n_max, char (*names)[32], long long* nonfinite, float* maxabs) override;
  float bench(const std::string& what, int batch, int arg, int iters) override;
  void set_stream(void* s) override { user_stream_ = static_cast<hipStream_t>(s); }
  const ModelConfig& config() const override { return cfg_; }
  const char* dtype_name() const override { return kDtypeName; }
  const int* sot_seq() const { return sot_seq_; }

 private:
  struct EncLayer {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    h16 *w_qkv, *w_o, *w_fc1, *w_fc2;
    float *b_qkv, *b_o, *b_fc1, *b_fc2;
  };

  void construct(const std::string& model_type, const std::string& model_path, const std::string& language, int device, int max_batch);
  void destroy();  // idempotent: the destructor's work, also run when the constructor throws
  hipStream_t stream() const { return user_stream_ ? user_stream_ : own_stream_; }
  void* dalloc(size_t bytes, bool zero = false);
  void load_config(const std::string& dir, const std::string& type, const std::string& language);
  void load_weights(const SafeTensors& st);
  void load_t2s(const std::string& model_path);
  void ensure_capacity(int batch);
  void free_slot_buffers();
  void upload_pcm(const float* const* pcm, const int* n_samples, int batch);
  // pinned_ns: optional pinned host [batch] the clip lengths are staged through (then nothing here waits for the stream)
  void run_frontend(const float* d_pcm, int stride, const int* n_samples, int batch, bool want_ref_layout, bool staged = false,
                    int* pinned_ns = nullptr);
  void run_encoder(int batch, const int* d_slot_map = nullptr);  // cross K/V of clip b goes to slot d_slot_map[b] (device), else b
  void reset_decode_state(int batch, const int* max_new_clip = nullptr);
  void enqueue_decode_step(int batch, int max_new, const int* d_forced, int n_forced, float* d_logits, long logits_stride,
                           int* d_argmax);
  void enqueue_decode_step_batched(int batch, int max_new, const int* d_forced, int n_forced, float* d_logits,
                                   long logits_stride, int* d_argmax);
  void enqueue_layers_cblock(int b0, int nb, hipStream_t s, bool forced, bool one_branch);
  int decode_branches(int batch) const;
  void ensure_branch_streams(int batch);
  hipGraphExec_t step_graph(int batch, int max_new);
  void recover_streams();
  int greedy_loop(int batch, int max_new, const int* max_new_clip = nullptr);
  // batch 1: the whole loop as one persistent launch (decode_persistent.hip); returns steps run, -1 if it gave up
  int run_persistent(int max_new, const int* d_forced, int n_forced, float* d_logits, int* d_argmax, int slot = 0, int max_new1 = -1, int max_new2 = -1);
  void fetch_ids(int batch, int32_t* ids, int* n_ids);

  ModelConfig cfg_;
  int sot_seq_[4] = {0, 0, 0, 0};
  std::vector<std::string> tokens_;
  std::unique_ptr<T2SConverter> t2s_;  // zh only
  std::string effective_lang_;
  bool feature_openai_ = false;  // feature_mode "openai" (engine.cpp load_config)
  int device_ = 0;
  bool device_set_ = false;
  void* load_stage_ = nullptr;  // staging buffer of load_weights
  hipStream_t own_stream_ = nullptr, user_stream_ = nullptr;
  std::vector<hipStream_t> pad_streams_;  // align_graph_queue() (engine_stream.cpp); AX_WHISPER_PAD_STREAMS (diagnostic)
  bool graph_branch_shares_queue(hipGraphExec_t exec, hipStream_t other);
  static constexpr int kMaxBranches = 4;   // parallel branches of the batched step graph
  hipStream_t branch_stream_[kMaxBranches - 1] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork_ = nullptr, ev_join_[kMaxBranches - 1] = {nullptr, nullptr, nullptr};
  std::vector<void*> allocs_;       // weights + constants (freed at destruction)
  std::vector<void*> slot_allocs_;  // capacity-dependent buffers

  // weights
  h16 *conv1_w_ = nullptr, *conv2_w_ = nullptr, *w_cross_kv_ = nullptr, *tok_emb_ = nullptr;
  float *conv1_b_ = nullptr, *conv2_b_ = nullptr, *enc_pos_ = nullptr, *ln_post_w_ = nullptr, *ln_post_b_ = nullptr;
  float *b_cross_kv_ = nullptr, *dec_pos_ = nullptr, *dec_ln_w_ = nullptr, *dec_ln_b_ = nullptr;
  int conv1_k_ = 0;
  std::vector<EncLayer> enc_;
  std::vector<DecLayerW> dec_;       // per-layer views into the two arenas below
  h16* dec_w_arena_ = nullptr; float* dec_f_arena_ = nullptr;
  // w_qkv and w_cq are ONE buffer of 4 d rows ([W_qkv; W_cq]); w_o heads a buffer of 2 d rows whose second half takes the query
  // fold's M_hi, m_lo its lo halves (decode_gemm.hip "QUERY FOLD"; filled by build_cblock_fold)
  struct DecLayerWP { const h16 *w_qkv, *w_o, *w_cq, *w_co, *w_fc1, *w_fc2; h16 *m_hi, *m_lo; };
  struct CblockFold { float *b_qkv4, *b_o2; const float *s, *c; };  // per layer: [b_qkv; 0], [b_o; d], s = W_cq g, c = W_cq beta + b_cq
  std::vector<CblockFold> cfold_;      // empty: the clip-block step keeps the fused query projection
  bool cfold_all_ = false;             // AX_WHISPER_CBLOCK_QFOLD=2: also in multi-branch steps (A/B, tests)
  float *d_a0_ = nullptr, *d_statp_ = nullptr;  // [B][d] A0 -> T; [B][d/16][2] block statistics of the residual rows
  void build_cblock_fold();
  std::vector<DecLayerWP> dec_packed_;  // fragment-major copies for the batched decode path
  const h16* tok_emb_packed_ = nullptr;
  int nbs_ = 1;                         // allocated clip blocks of 16
  // front-end constants
  float *twiddle_ = nullptr, *window_ = nullptr, *mel_basis_t_ = nullptr;
  int* d_sot_ = nullptr;

  // capacity-dependent
  int cap_ = 0;
  int t_pad_ = 1536, mel_rows_ = 3004, h1_rows_ = 3002;
  float* d_pcm_ = nullptr; long pcm_stride_ = 0; float* h_pcm_ = nullptr;
  // clips longer than a staging row (60 s): their tails, packed, so that the clamp floor comes from ALL frames of the
  // input however long it is (Whisper.cpp:158-172); grown on demand, empty for ordinary requests
  float* d_over_ = nullptr; size_t over_cap_ = 0; long long* d_over_off_ = nullptr; bool over_used_ = false;
  int* d_nsamp_ = nullptr; unsigned* d_gmax_ = nullptr; float* d_logmel_ = nullptr; float* d_mel_ref_ = nullptr;
  h16 *d_mel_tm_ = nullptr, *d_h1_ = nullptr, *d_ln_ = nullptr, *d_q_ = nullptr, *d_k_ = nullptr, *d_vt_ = nullptr,
       *d_attn_ = nullptr, *d_ffn_ = nullptr;
  float* d_x_ = nullptr;
  static constexpr int kCrossSplitMax = 6;  // workgroups per (clip, head) of the batched cross-attention launch at few clips
  static constexpr int kEncPartClips = 2;  // split-K of the encoder's residual GEMMs pays for at most this many clips
  float* d_enc_part_ = nullptr;
  bool enc_split_k_ = true;
  float enc_rescale_thr_ = 8.f;  // launch_encoder_attention
  int gemv_max_ = 2;             // clips per call up to which the decoder step uses the GEMV family (AX_WHISPER_GEMV_MAX, <= 4)
  int cross_split_env_ = 0;      // AX_WHISPER_CROSS_SPLIT: workgroups per (clip, head) of the batched cross-attention, 0 = by clip count
  h16 *d_cross_k_ = nullptr, *d_cross_v_ = nullptr, *d_self_k_ = nullptr, *d_self_v_ = nullptr;
  float *d_xdec_ = nullptr, *d_qdec_ = nullptr, *d_hid_ = nullptr, *d_part_self_ = nullptr, *d_part_cross_ = nullptr;
  h16 *d_act_[2] = {nullptr, nullptr}, *d_att_[2] = {nullptr, nullptr}, *d_hidp_[2] = {nullptr, nullptr};
  float* d_part_ = nullptr;
  float* d_amax_val_ = nullptr; int* d_amax_idx_ = nullptr; int n_amax_part_ = 0;
  float* d_attn_mpart_ = nullptr;     // batched cross-attention in splits: partials and tickets (DecAttnParams::mpart / mcnt)
  unsigned* d_attn_mcnt_ = nullptr;
  int *d_tok_ = nullptr, *d_done_ = nullptr, *d_done_none_ = nullptr, *d_nout_ = nullptr, *d_out_ids_ = nullptr, *d_max_new_clip_ = nullptr;
  int* d_off_ = nullptr;   // per-slot offsets (common.hpp: DecState)
  int n_cu_ = 0;            // compute units of the device
  // slot refill (stream_*): a slot is idle -> encoding (admitted, encoder in flight on admit_stream_) -> active (decoding)
  // -> finished (done flag seen) -> idle again after stream_collect
  enum SlotState : int { kIdle = 0, kEncoding = 1, kActive = 2, kFinished = 3 };
  int stream_slots_ = 0;                 // > 0: a stream is open (slots of the step graph: at least 3)
  int stream_user_slots_ = 0;            // the n_slots the caller asked for: the slot indices it may use
  std::vector<int> slot_state_, slot_max_new_;
  std::vector<hipEvent_t> ev_admit_;     // one per slot: its encoder has finished
  hipStream_t admit_stream_ = nullptr;
  int* h_done_live_ = nullptr; int* d_done_live_ = nullptr;  // host-mapped [cap] (and its device alias): advance_kernel raises a clip's
                                                             // flag the moment it finishes; the host reads it without any wait
  hipEvent_t ev_step_[3] = {nullptr, nullptr, nullptr};  // the host stays two steps ahead of the device (stream_step)
  long step_seq_ = 0;
  hipStream_t copy_stream_ = nullptr;    // stream_collect's D2H copies: never behind the queued decoder steps
  int* d_slot_map_ = nullptr;            // [cap]: clip index of an admission pass -> slot
  // admission passes do not wait for one another's encoder: clip lengths and slot maps go through a pinned ring of
  // kAdmitRing entries (an entry is reused once the pass that filled it has finished), the PCM staging rows are reused once
  // the previous pass's uploads have landed
  static constexpr int kAdmitRing = 4;
  int* h_admit_ring_ = nullptr;          // pinned [kAdmitRing][2][cap]
  hipEvent_t ev_ring_[kAdmitRing] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_upload_ = nullptr;
  long admit_seq_ = 0;
  void require_no_stream(const char* what) const;
  DecState* d_state_ = nullptr;
  int* h_poll_ = nullptr;  // pinned
  int split_self_ = 2, split_cross_ = 6;
  int step_mask_ = 15;  // bench only: 1 GEMV/GEMM launches, 2 attention launches, 4 advance, 8 act_prep, 16 attention launches stamp themselves
  // bench "attn_stamp": every decode_attention launch of a captured step gets a {min begin, max end} slot (DecAttnParams::stamp)
  struct StampMeta { int layer, cross, b0, nb; };
  static constexpr size_t kStampWgs = 4096, kStampLaunches = 256;
  unsigned long long* d_stamp_ = nullptr;
  std::vector<StampMeta> stamp_meta_;
  unsigned long long* next_stamp(int layer, int cross, int b0, int nb);
  // persistent batch-1 decode
  bool batched_ln_ = false;         // batched decode: clip-block GEMM sequence (AX_WHISPER_BATCHED_LN=0 disables)
  bool persistent_ok_ = false;      // model shape supported and not disabled (AX_WHISPER_DECODE=graph)
  int persist_max_clips_ = 1;       // clips per persistent launch: up to 3 for d_model <= 768 (AX_WHISPER_PERSIST2=<n> caps it, 0 = 1)
  h16 *d_self_k1_ = nullptr, *d_self_v1_ = nullptr; size_t self1_bytes_ = 0;  // the later clips' self-attention caches of that launch
  int persist_skip_ = 0, persist_backoff_ = 0, persist_giveups_ = 0;  // re-arming after a give-up (engine_decode.cpp)
  bool persistent_usable();
  void persistent_gave_up();
  void persistent_succeeded();
  int persist_grid_ = 0;
  u64* d_gran_ = nullptr; size_t gran_bytes_ = 0;
  float* d_qfold_ = nullptr;  // query-fold arena of the one-clip launch (d_model <= 768), nullptr = unfolded
  std::map<long, hipGraphExec_t> graphs_;  // key: batch * 1024 + max_new
  hipEvent_t ev_[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

}  // inline namespace AXW_NS
}  // namespace axw
