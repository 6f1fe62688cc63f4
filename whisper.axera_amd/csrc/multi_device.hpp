// multi_device.hpp — utterance-level data parallelism over the GPUs of one node, inside ONE process (host only).
//
// The unit being sharded is the reference's one-utterance Whisper::run (cpp/src/Whisper.cpp:186-239): every utterance's
// front-end, encoder and decode loop share nothing but read-only weights, so a batch of B clips splits into contiguous
// blocks of ceil(B / G) clips, one block per device, with no collective at all inside a process: each
// device's engine writes its block of the caller's result arrays directly (plain D2H per device). The multi-process
// form of the same partitioning (one rank per GPU, RCCL all_gather of the ids) is whisper.axera_amd/dp.py + bench.py.
//
// Header-only and free of HIP types on purpose: tests/test_multi_device.py instantiates DeviceGroup with a stand-in
// engine under plain g++ to check the sharding and the joining on a box without a GPU.
#pragma once

#include <atomic>
#include <cstdint>
#include <exception>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace axw {

// contiguous block [lo, hi) of ceil(n / world) items for worker `rank`; trailing workers may get fewer or none
// (the same rule as dp.shard_range, so the in-process and the multi-process partitions agree)
inline void shard_range(int n, int rank, int world, int* lo, int* hi) {
  const int per = world > 0 ? (n + world - 1) / world : n;
  const long l = (long)rank * per;
  *lo = (int)(l < n ? l : n);
  *hi = *lo + per < n ? *lo + per : n;
}

// fn(worker, lo, hi) for every non-empty shard, one host thread per worker (worker 0 on the calling thread); joins all
// of them, then rethrows the first failure (by worker index) with the worker named in the message.
template <typename Fn>
void run_sharded(int n, int world, Fn&& fn) {
  if (n <= 0 || world <= 0) return;
  std::vector<std::string> errors(world);
  std::vector<char> failed(world, 0);
  auto body = [&](int w) {
    int lo, hi;
    shard_range(n, w, world, &lo, &hi);
    if (hi <= lo) return;
    try {
      fn(w, lo, hi);
    } catch (const std::exception& e) {
      failed[w] = 1;
      errors[w] = e.what();
    } catch (...) {
      failed[w] = 1;
      errors[w] = "unknown error";
    }
  };
  std::vector<std::thread> threads;
  for (int w = 1; w < world; ++w) {
    int lo, hi;
    shard_range(n, w, world, &lo, &hi);
    if (hi > lo) threads.emplace_back(body, w);
  }
  body(0);
  for (auto& t : threads) t.join();
  for (int w = 0; w < world; ++w)
    if (failed[w]) throw std::runtime_error("device worker " + std::to_string(w) + ": " + errors[w]);
}

// G engines, one per device. E needs: std::mutex& mutex(); void run_tokens(const float* const* pcm, const float* d_pcm,
// int d_stride, const int* n_samples, int batch, int max_new, int32_t* ids, int* n_ids).
template <typename E>
class DeviceGroup {
 public:
  DeviceGroup() = default;
  void add(std::unique_ptr<E> e) { engines_.push_back(std::move(e)); }
  int size() const { return (int)engines_.size(); }
  E& at(int i) { return *engines_.at(i); }
  E& primary() { return *engines_.at(0); }

  // Host PCM of `batch` clips -> ids [batch][n_ctx], n_ids [batch]. With one engine (or one clip) this is the
  // engine's own call; otherwise workers = min(G, batch) engines each take one contiguous block. An engine is
  // serialised by its own mutex (a handle may be shared between threads), engines of different devices run concurrently.
  void run_tokens(const float* const* pcm, const int* n_samples, int batch, int max_new, int n_ctx, int32_t* ids, int* n_ids) {
    if (batch < 1) throw std::runtime_error("batch must be >= 1");
    const int G = size(), world = G < batch ? G : batch;
    // calls with fewer clips than devices start at a rotating device, so that concurrent small requests on one handle
    // (a server thread pool calling AX_WHISPER_RunPCM) spread over the GPUs instead of queueing on the first engine
    const unsigned first = world < G ? next_.fetch_add((unsigned)world) % (unsigned)G : 0u;
    run_sharded(batch, world, [&](int w, int lo, int hi) {
      E& e = *engines_[(first + (unsigned)w) % (unsigned)G];
      std::lock_guard<std::mutex> lock(e.mutex());
      e.run_tokens(pcm + lo, nullptr, 0, n_samples + lo, hi - lo, max_new, ids + (size_t)lo * n_ctx, n_ids + lo);
    });
  }

 private:
  std::vector<std::unique_ptr<E>> engines_;
  std::atomic<unsigned> next_{0};
};

// "0,2,5" / "all" / "" -> device ordinals (all = 0..n_visible-1). Throws on a malformed list or an ordinal out of range.
inline std::vector<int> parse_device_list(const std::string& s, int n_visible) {
  std::vector<int> out;
  if (s.empty() || s == "all") {
    for (int i = 0; i < n_visible; ++i) out.push_back(i);
    return out;
  }
  size_t p = 0;
  while (p <= s.size()) {
    size_t q = s.find(',', p);
    if (q == std::string::npos) q = s.size();
    const std::string tok = s.substr(p, q - p);
    if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos)
      throw std::runtime_error("bad device list '" + s + "'");
    const int d = std::stoi(tok);
    if (d >= n_visible) throw std::runtime_error("device " + tok + " of '" + s + "' is not visible (" + std::to_string(n_visible) + " devices)");
    for (int o : out)
      if (o == d) throw std::runtime_error("device " + tok + " listed twice in '" + s + "'");
    out.push_back(d);
    p = q + 1;
  }
  return out;
}

}  // namespace axw
