// CPU check of whisper.axera_amd/csrc/multi_device.hpp with a stand-in engine (no GPU, no HIP): the sharding of a
// batch over G devices, the joining, error propagation, and the device-list parser. Built and run by
// tests/test_multi_device.py with plain g++.
#include <atomic>
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <set>

#include "multi_device.hpp"

struct FakeEngine {
  int device;
  std::mutex mu;
  std::vector<std::pair<int, const float*>> calls;  // (batch, first clip pointer)
  int fail_on_value = -1;
  static std::atomic<int> concurrent, max_concurrent, wait_ms;
  explicit FakeEngine(int d) : device(d) {}
  std::mutex& mutex() { return mu; }
  void run_tokens(const float* const* pcm, const float* d_pcm, int, const int* n_samples, int batch, int max_new, int32_t* ids, int* n_ids) {
    assert(d_pcm == nullptr);
    const int c = ++concurrent;
    int m = max_concurrent.load();
    while (c > m && !max_concurrent.compare_exchange_weak(m, c)) {}
    // stay inside the call until a second worker has been seen inside too (or 2 s: a single worker / a serial bug), so the
    // side-by-side check below does not depend on how fast a loaded machine starts threads
    for (int i = 0; i < wait_ms.load() && max_concurrent.load() < 2; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    calls.push_back({batch, pcm[0]});
    for (int b = 0; b < batch; ++b) {
      if ((int)pcm[b][0] == fail_on_value) { --concurrent; throw std::runtime_error("clip " + std::to_string((int)pcm[b][0]) + " is poisoned"); }
      // ids are a function of the clip content only: [value, value + 1, ...], n = samples % 5 + 1, capped by max_new
      int n = n_samples[b] % 5 + 1;
      if (max_new > 0 && n > max_new) n = max_new;
      n_ids[b] = n;
      for (int i = 0; i < n; ++i) ids[(size_t)b * 448 + i] = (int32_t)pcm[b][0] + i;
    }
    --concurrent;
  }
};
std::atomic<int> FakeEngine::concurrent{0}, FakeEngine::max_concurrent{0}, FakeEngine::wait_ms{0};

static void check_shards() {
  for (int n : {0, 1, 5, 7, 64, 512, 513})
    for (int world : {1, 2, 3, 8}) {
      int next = 0;
      for (int r = 0; r < world; ++r) {
        int lo, hi;
        axw::shard_range(n, r, world, &lo, &hi);
        assert(lo == next || (lo == n && hi == n));
        assert(lo <= hi && hi <= n);
        next = hi > next ? hi : next;
      }
      assert(next == n);
    }
  int lo, hi;
  axw::shard_range(512, 3, 8, &lo, &hi);
  assert(lo == 192 && hi == 256);  // BASELINE configs[4]: 64 clips per GPU
}

static void run_case(int G, int B, int max_new) {
  axw::DeviceGroup<FakeEngine> g;
  for (int d = 0; d < G; ++d) g.add(std::unique_ptr<FakeEngine>(new FakeEngine(d)));
  std::vector<std::vector<float>> clips(B);
  std::vector<const float*> ptrs(B);
  std::vector<int> lens(B);
  for (int b = 0; b < B; ++b) { clips[b].assign(3 + b % 4, (float)(1000 + 10 * b)); ptrs[b] = clips[b].data(); lens[b] = (int)clips[b].size(); }
  std::vector<int32_t> ids((size_t)B * 448, -1), want((size_t)B * 448, -1);
  std::vector<int> n(B, -1), wn(B, -1);
  FakeEngine::max_concurrent = 0;
  FakeEngine::wait_ms = (G < B ? G : B) > 1 ? 2000 : 0;
  g.run_tokens(ptrs.data(), lens.data(), B, max_new, 448, ids.data(), n.data());
  FakeEngine::wait_ms = 0;
  FakeEngine single(99);
  single.run_tokens(ptrs.data(), nullptr, 0, lens.data(), B, max_new, want.data(), wn.data());
  assert(n == wn && ids == want);  // joined result == one engine over the whole batch, in order
  const int world = G < B ? G : B, per = (B + world - 1) / world;
  int covered = 0;
  for (int d = 0; d < G; ++d) {
    FakeEngine& e = g.at(d);
    if (d < world && d * per < B) {
      assert(e.calls.size() == 1);
      assert(e.calls[0].second == ptrs[d * per]);                              // contiguous block d
      assert(e.calls[0].first == (d * per + per < B ? per : B - d * per));
      covered += e.calls[0].first;
    } else {
      assert(e.calls.empty());
    }
  }
  assert(covered == B);
  if (world > 1) assert(FakeEngine::max_concurrent.load() > 1);  // the devices really ran side by side
}

static void check_errors() {
  axw::DeviceGroup<FakeEngine> g;
  for (int d = 0; d < 4; ++d) g.add(std::unique_ptr<FakeEngine>(new FakeEngine(d)));
  g.at(2).fail_on_value = 1000 + 10 * 5;  // clip 5 lives in block 2 of 4 x 2
  const int B = 8;
  std::vector<std::vector<float>> clips(B);
  std::vector<const float*> ptrs(B);
  std::vector<int> lens(B, 4);
  for (int b = 0; b < B; ++b) { clips[b].assign(4, (float)(1000 + 10 * b)); ptrs[b] = clips[b].data(); }
  std::vector<int32_t> ids((size_t)B * 448);
  std::vector<int> n(B);
  bool threw = false;
  try {
    g.run_tokens(ptrs.data(), lens.data(), B, 0, 448, ids.data(), n.data());
  } catch (const std::exception& e) {
    threw = true;
    assert(strstr(e.what(), "device worker 2") && strstr(e.what(), "poisoned"));
  }
  assert(threw);
  for (int d = 0; d < 4; ++d) assert(g.at(d).calls.size() == 1);  // every worker was joined (none left running)
  bool t2 = false;
  try { g.run_tokens(ptrs.data(), lens.data(), 0, 0, 448, ids.data(), n.data()); } catch (const std::exception&) { t2 = true; }
  assert(t2);
}

static void check_rotation() {
  // fewer clips than devices: consecutive calls start at consecutive devices (concurrent 1-clip requests spread out)
  axw::DeviceGroup<FakeEngine> g;
  for (int d = 0; d < 3; ++d) g.add(std::unique_ptr<FakeEngine>(new FakeEngine(d)));
  std::vector<float> clip(4, 7.f);
  const float* ptr = clip.data();
  int len = 4, n = 0;
  std::vector<int32_t> ids(448);
  for (int call = 0; call < 7; ++call) g.run_tokens(&ptr, &len, 1, 0, 448, ids.data(), &n);
  assert(g.at(0).calls.size() == 3 && g.at(1).calls.size() == 2 && g.at(2).calls.size() == 2);
  // two clips on three devices: devices (1, 2) after seven 1-clip calls, then (0, 1)
  const float* two[2] = {ptr, ptr};
  int lens[2] = {4, 4}, ns[2];
  std::vector<int32_t> ids2(2 * 448);
  g.run_tokens(two, lens, 2, 0, 448, ids2.data(), ns);
  assert(g.at(1).calls.size() == 3 && g.at(2).calls.size() == 3 && g.at(0).calls.size() == 3);
  // a full batch always uses every device, block w on device w
  const float* six[6] = {ptr, ptr, ptr, ptr, ptr, ptr};
  int lens6[6] = {4, 4, 4, 4, 4, 4}, ns6[6];
  std::vector<int32_t> ids6(6 * 448);
  g.run_tokens(six, lens6, 6, 0, 448, ids6.data(), ns6);
  for (int d = 0; d < 3; ++d) assert(g.at(d).calls.size() == 4 && g.at(d).calls.back().first == 2);
}

static void check_device_lists() {
  using axw::parse_device_list;
  assert((parse_device_list("all", 3) == std::vector<int>{0, 1, 2}));
  assert((parse_device_list("", 2) == std::vector<int>{0, 1}));
  assert((parse_device_list("2,0", 4) == std::vector<int>{2, 0}));
  for (const char* bad : {"0,,1", "a", "0,4", "1,1", "-1", "0,"}) {
    bool threw = false;
    try { parse_device_list(bad, 4); } catch (const std::exception&) { threw = true; }
    assert(threw);
  }
}

int main() {
  check_shards();
  for (int G : {1, 2, 3, 8})
    for (int B : {1, 2, 5, 8, 64, 65}) run_case(G, B, B % 2 ? 0 : 3);
  check_errors();
  check_rotation();
  check_device_lists();
  printf("multi_device ok\n");
  return 0;
}
