// whisper_srv.cpp — HTTP entry point, MI355X build.
//
// Keeps the reference server's contract (cpp/whisper_srv.cpp:10-70, cpp/src/WhisperHTTPServer.hpp:
// 37-100): flags --port (8080), --model_type/-t, --model_path/-p, --language/-l; route POST /asr
// with Content-Type application/octet-stream and a body of raw little-endian f32 PCM (16 kHz mono);
// replies {"success": true, "text": ...}; 400 with the reference's error strings for a wrong
// content type / empty body / size % 4 != 0 / failed run. Adds GET /health.
//
// What is new: the reference calls one non-re-entrant handle from cpp-httplib's thread pool with
// no lock. Here connection threads only parse requests and enqueue them; one
// scheduler thread PER DEVICE (--devices all | 0,1,...; default: the one device of AX_WHISPER_Init)
// drains the shared queue on its own handle — utterance-level data parallelism inside a GPU and across
// the GPUs of the node with no cross-device synchronisation: an idle device simply takes the next requests.
//   --scheduler slots (default): --max_batch utterance SLOTS that are refilled as they finish
//       (AX_WHISPER_Stream*): every clip stops at its own eot, as the reference's one-by-one loop does
//       (Whisper.cpp:219-222), and the freed slot takes the next request while the other slots decode on.
//       A lone request on an idle device still goes through the one-clip path (the persistent launch). More slots
//       serve more clips per second (Whisper-small, clips of 60-150 ids: 341 / 456 / 489 clips/s with 64 / 128 / 256).
//   --scheduler batches: micro-batches (up to --max_batch clips, waiting at most --batch_wait_ms for
//       stragglers) through AX_WHISPER_RunPCMBatch; a batch returns when its slowest clip has finished.
// Connections: at most --max_conns at a time (503 beyond), a body of at most --max_body_mb (413 beyond; 30 s of
// audio is 1.92 MB), --recv_timeout_s per read, and a body that ends before its Content-Length is a 400, not a
// transcription of half a clip.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/time.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ax_whisper_api.h"
#include "http_request.hpp"

struct Job {
  std::vector<float> pcm;
  std::promise<std::pair<bool, std::string>> done;
  std::chrono::steady_clock::time_point arrived = std::chrono::steady_clock::now();
};

static std::mutex g_mu;
static std::condition_variable g_cv;
static std::deque<Job*> g_queue;
static std::atomic<bool> g_stop{false};
static std::atomic<int> g_conns{0}, g_busy_slots{0};
static std::atomic<long> g_served{0};
static std::vector<AX_WHISPER_HANDLE> g_models;  // for /health
static int g_max_conns = 256, g_recv_timeout_s = 10;
static size_t g_max_body = (size_t)16 << 20;

static std::string json_escape(const std::string& s) {
  std::string o;
  for (unsigned char c : s) {
    switch (c) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      default:
        if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o += (char)c;
    }
  }
  return o;
}

static void batcher(AX_WHISPER_HANDLE model, int max_batch, int wait_ms) {
  while (!g_stop) {
    std::vector<Job*> jobs;
    {
      std::unique_lock<std::mutex> lk(g_mu);
      g_cv.wait(lk, [] { return g_stop || !g_queue.empty(); });
      if (g_stop) break;
      auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(wait_ms);
      while ((int)g_queue.size() < max_batch && g_cv.wait_until(lk, deadline) != std::cv_status::timeout) {}
      while (!g_queue.empty() && (int)jobs.size() < max_batch) { jobs.push_back(g_queue.front()); g_queue.pop_front(); }
    }
    const int n = (int)jobs.size();
    if (n == 0) continue;  // another device's batcher took the requests while this one waited for stragglers
    std::vector<const float*> ptrs(n);
    std::vector<int> lens(n);
    std::vector<char*> texts(n, nullptr);
    for (int i = 0; i < n; ++i) { ptrs[i] = jobs[i]->pcm.data(); lens[i] = (int)jobs[i]->pcm.size(); }
    int rc = AX_WHISPER_RunPCMBatch(model, ptrs.data(), lens.data(), n, texts.data());
    std::vector<int> ok(n, rc == 0);
    if (rc != 0 && n > 1) {
      // one bad request (e.g. non-finite samples) must not fail the strangers batched with it: run them one by one
      for (int i = 0; i < n; ++i) {
        free(texts[i]);
        texts[i] = nullptr;
        ok[i] = AX_WHISPER_RunPCMBatch(model, &ptrs[i], &lens[i], 1, &texts[i]) == 0;
      }
    }
    for (int i = 0; i < n; ++i) {
      jobs[i]->done.set_value({ok[i] && texts[i], texts[i] ? std::string(texts[i]) : std::string()});
      free(texts[i]);  // the reference never frees it (WhisperHTTPServer.hpp:77-90)
      ++g_served;
    }
  }
}

// Slots that are refilled as they finish (AX_WHISPER_Stream*). One request alone on an idle device — or two or three, where the
// model has a multi-clip launch — takes the batch entry point instead (the persistent launch is 2x faster than a step sequence with one
// or two live slots).
// Admission policy (--min_admit N --admit_wait_ms T): while something is decoding, an admission pass is held back until N
// requests can be admitted together or the oldest waiting request is T ms old — an encoder pass over 8 clips costs 0.72 ms
// per clip against 1.74 ms alone (bench.py stream64.encoder_pass_by_group_size). Default N = 1 (admit as they come): on
// MI355X at 64 slots and 60-150 ids per clip that measures FASTER than N = 8 (366 against 354 clips/s, bench.py stream64),
// because a slot that waits for company idles for whole decoder steps; the knob is for models whose encoder dominates.
static void slot_scheduler(AX_WHISPER_HANDLE model, int n_slots, int wait_ms, int steps_per_call, int min_admit, int admit_wait_ms) {
  const int Tc = std::max(1, AX_WHISPER_GetConfigInt(model, "n_text_ctx"));
  std::vector<Job*> owner(n_slots, nullptr);
  std::vector<int> fin(std::max(n_slots, 3));
  std::vector<int32_t> ids(Tc);
  int busy = 0, reported = 0;
  bool open = false;
  const int launch_clips = AX_WHISPER_GetConfigInt(model, "persistent_max_clips");  // clips one persistent launch decodes (1-3)
  auto report = [&] { g_busy_slots += busy - reported; reported = busy; };  // /health sums the devices
  auto fail = [](Job* j) { j->done.set_value({false, std::string()}); };
  auto run_alone = [&](Job* j) {
    const float* ptr = j->pcm.data();
    int len = (int)j->pcm.size();
    char* text = nullptr;
    const bool ok = AX_WHISPER_RunPCMBatch(model, &ptr, &len, 1, &text) == 0 && text;
    j->done.set_value({ok, text ? std::string(text) : std::string()});
    free(text);
    ++g_served;
  };
  while (!g_stop) {
    std::vector<Job*> take;
    {
      std::unique_lock<std::mutex> lk(g_mu);
      if (busy == 0) {
        g_cv.wait(lk, [] { return g_stop || !g_queue.empty(); });
        if (g_stop) break;
        if (g_queue.size() == 1 && wait_ms > 0) g_cv.wait_for(lk, std::chrono::milliseconds(wait_ms), [] { return g_stop || g_queue.size() > 1; });
      }
      const int room = n_slots - busy, can = std::min(room, (int)g_queue.size());
      const bool hold = busy > 0 && can > 0 && can < std::min(min_admit, room) &&
                        std::chrono::steady_clock::now() - g_queue.front()->arrived < std::chrono::milliseconds(admit_wait_ms);
      while (!hold && !g_queue.empty() && busy + (int)take.size() < n_slots) { take.push_back(g_queue.front()); g_queue.pop_front(); }
    }
    if (busy == 0 && take.size() == 1) {  // alone on an idle device
      if (open) { AX_WHISPER_StreamClose(model); open = false; }
      run_alone(take[0]);
      continue;
    }
    // (persistent_decode is 0 while the engine backs off after a give-up — CUs taken by another process: the group path would then
    // run the launch-per-phase sequence synchronously, ~316 ms per pair against ~240 ms through slots, with no admission meanwhile)
    if (busy == 0 && take.size() >= 2 && (int)take.size() <= launch_clips && AX_WHISPER_GetConfigInt(model, "persistent_decode") == 1) {
      // two or three requests on an idle device: their greedy loops in ONE persistent launch (134 ms for a pair of Whisper-small
      // clips of 444 ids against ~240 ms for two live slots of the step sequence); a refused group (one bad request) is served one by one
      if (open) { AX_WHISPER_StreamClose(model); open = false; }
      const int n = (int)take.size();
      const float* ptrs[3];
      int lens[3];
      char* texts[3] = {nullptr, nullptr, nullptr};
      for (int i = 0; i < n; ++i) { ptrs[i] = take[i]->pcm.data(); lens[i] = (int)take[i]->pcm.size(); }
      bool ok = AX_WHISPER_RunPCMBatch(model, ptrs, lens, n, texts) == 0;
      for (int i = 0; i < n; ++i) ok = ok && texts[i];
      if (ok) {
        for (int i = 0; i < n; ++i) { take[i]->done.set_value({true, std::string(texts[i])}); ++g_served; }
      } else {
        for (int i = 0; i < n; ++i) run_alone(take[i]);
      }
      for (int i = 0; i < n; ++i) free(texts[i]);
      continue;
    }
    if (!take.empty() && !open) {
      if (AX_WHISPER_StreamOpen(model, n_slots) != 0) { for (Job* j : take) run_alone(j); continue; }
      open = true;
    }
    if (!take.empty()) {
      // every request taken now goes through ONE batched front-end + encoder pass into the idle slots; when the pass is
      // refused (one bad request: e.g. non-finite samples) the requests are admitted one by one so that only that one fails
      std::vector<int> sl;
      std::vector<const float*> ptrs;
      std::vector<int> lens;
      for (int i = 0, s2 = 0; i < (int)take.size(); ++i) {
        while (owner[s2]) ++s2;
        sl.push_back(s2++);
        ptrs.push_back(take[i]->pcm.data());
        lens.push_back((int)take[i]->pcm.size());
      }
      if (AX_WHISPER_StreamAdmitBatch(model, sl.data(), ptrs.data(), lens.data(), nullptr, (int)take.size()) == 0) {
        for (size_t i = 0; i < take.size(); ++i) owner[sl[i]] = take[i];
        busy += (int)take.size();
      } else {
        for (size_t i = 0; i < take.size(); ++i) {
          if (AX_WHISPER_StreamAdmit(model, sl[i], ptrs[i], lens[i], 0) != 0) { fail(take[i]); continue; }
          owner[sl[i]] = take[i];
          ++busy;
        }
      }
    }
    report();
    if (busy == 0) continue;
    int nfin = 0;
    if (AX_WHISPER_StreamStep(model, steps_per_call, fin.data(), &nfin) != 0) {
      for (int sl = 0; sl < n_slots; ++sl) if (owner[sl]) { fail(owner[sl]); owner[sl] = nullptr; }
      busy = 0;
      AX_WHISPER_StreamClose(model);
      open = false;
      continue;
    }
    bool collect_failed = false;
    for (int i = 0; i < nfin && !collect_failed; ++i) {
      const int sl = fin[i];
      if (sl < 0 || sl >= n_slots || !owner[sl]) continue;
      int n = 0;
      char* text = nullptr;
      if (AX_WHISPER_StreamCollect(model, sl, ids.data(), &n) != 0) { collect_failed = true; break; }
      const bool ok = AX_WHISPER_Transcript(model, ids.data(), n, &text) == 0 && text;
      owner[sl]->done.set_value({ok, text ? std::string(text) : std::string()});
      free(text);
      owner[sl] = nullptr;
      --busy;
      ++g_served;
    }
    if (collect_failed) {
      // the engine still holds that slot as finished-but-uncollected and would refuse every later admission into it: the
      // stream is closed and reopened by the next requests, as after a failed step (the requests in flight fail)
      for (int sl = 0; sl < n_slots; ++sl) if (owner[sl]) { fail(owner[sl]); owner[sl] = nullptr; }
      busy = 0;
      AX_WHISPER_StreamClose(model);
      open = false;
    }
    report();
  }
  for (int sl = 0; sl < n_slots; ++sl) if (owner[sl]) fail(owner[sl]);
  if (open) AX_WHISPER_StreamClose(model);
}

static void send_response(int fd, int status, const std::string& body) {
  const char* reason = status == 200 ? "OK" : status == 400 ? "Bad Request" : status == 404 ? "Not Found" : status == 413 ? "Payload Too Large"
                       : status == 503 ? "Service Unavailable" : "Internal Server Error";
  std::string h = "HTTP/1.1 " + std::to_string(status) + " " + reason +
                  "\r\nContent-Type: application/json\r\nContent-Length: " + std::to_string(body.size()) +
                  "\r\nAccess-Control-Allow-Origin: *\r\nAccess-Control-Allow-Methods: POST, GET, OPTIONS\r\n"
                  "Access-Control-Allow-Headers: Content-Type, X-Array-Name, X-Array-Description, X-Array-Size\r\n"
                  "Connection: close\r\n\r\n";
  std::string out = h + body;
  size_t off = 0;
  while (off < out.size()) {
    ssize_t w = send(fd, out.data() + off, out.size() - off, MSG_NOSIGNAL);
    if (w <= 0) break;
    off += (size_t)w;
  }
}

struct ConnGuard { ~ConnGuard() { --g_conns; } };

static void serve(int fd) {
  ConnGuard guard;
  std::string buf;
  char tmp[65536];
  size_t hdr_end = std::string::npos;
  while (hdr_end == std::string::npos && buf.size() < (1u << 20)) {
    ssize_t r = recv(fd, tmp, sizeof tmp, 0);
    if (r <= 0) { close(fd); return; }
    buf.append(tmp, (size_t)r);
    hdr_end = buf.find("\r\n\r\n");
  }
  if (hdr_end == std::string::npos) { close(fd); return; }
  axw::HttpHead hh;
  axw::parse_http_head(buf, hh);  // csrc/http_request.hpp: the parsing itself is socket-free (and runs under ASan / UBSan in tests/)
  if (!hh.length_ok) {
    send_response(fd, 400, R"({"error": "Bad Content-Length"})");
    shutdown(fd, SHUT_RDWR);
    close(fd);
    return;
  }
  const size_t clen = hh.content_length;
  if (hh.expect_continue) {
    const char* c = "HTTP/1.1 100 Continue\r\n\r\n";
    (void)!send(fd, c, strlen(c), MSG_NOSIGNAL);
  }
  if (clen > g_max_body) {  // 30 s of audio is 1.92 MB; nothing longer than the cap is read at all
    send_response(fd, 413, R"({"error": "Request body too large"})");
    shutdown(fd, SHUT_RDWR);
    close(fd);
    return;
  }
  std::string body = buf.substr(hdr_end + 4);
  while (body.size() < clen) {
    ssize_t r = recv(fd, tmp, sizeof tmp, 0);  // SO_RCVTIMEO bounds every read
    if (r <= 0) break;
    body.append(tmp, (size_t)r);
  }
  if (body.size() < clen) {  // the peer closed (or stalled) before the body was complete: not half a clip's transcript
    send_response(fd, 400, R"({"error": "Request body is incomplete"})");
    shutdown(fd, SHUT_RDWR);
    close(fd);
    return;
  }
  const axw::HttpRoute route = axw::http_route(hh);
  if (route == axw::HttpRoute::Health) {
    long giveups = 0;
    for (AX_WHISPER_HANDLE m : g_models) giveups += std::max(0, AX_WHISPER_GetConfigInt(m, "persistent_giveups"));
    size_t queued;
    { std::lock_guard<std::mutex> lk(g_mu); queued = g_queue.size(); }
    char hb[256];
    snprintf(hb, sizeof hb, "{\"status\": \"ok\", \"devices\": %d, \"queued\": %zu, \"busy_slots\": %d, \"connections\": %d, \"served\": %ld, \"persistent_giveups\": %ld}",
             (int)g_models.size(), queued, g_busy_slots.load(), g_conns.load(), g_served.load(), giveups);
    send_response(fd, 200, hb);
  } else if (route == axw::HttpRoute::Options) {
    send_response(fd, 200, "{}");
  } else if (route == axw::HttpRoute::NotFound) {
    send_response(fd, 404, R"({"error": "Not found"})");
  } else if (const char* bad = axw::asr_request_error(hh, body.size())) {
    send_response(fd, 400, bad);
  } else {
    {
      Job job;
      job.pcm.resize(body.size() / sizeof(float));
      memcpy(job.pcm.data(), body.data(), body.size());                                      // hpp:103-113
      auto fut = job.done.get_future();
      { std::lock_guard<std::mutex> lk(g_mu); g_queue.push_back(&job); }
      g_cv.notify_all();
      auto res = fut.get();
      if (!res.first) send_response(fd, 400, R"({"error": "Run model failed!"})");           // hpp:79-83
      else send_response(fd, 200, "{\n  \"success\": true,\n  \"text\": \"" + json_escape(res.second) + "\"\n}");  // hpp:86-90
    }
  }
  shutdown(fd, SHUT_RDWR);
  close(fd);
}

int main(int argc, char** argv) {
  int port = 8080, max_batch = 64, wait_ms = 5, steps_per_call = 8, max_body_mb = 16, min_admit = 1, admit_wait_ms = 20;
  std::string model_type = "turbo", model_path = "../models-mi355x", language = "zh", devices, scheduler = "slots";
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto val = [&](const char* lng, const char* sht, std::string& dst) -> bool {
      std::string l = std::string("--") + lng;
      if (a.rfind(l + "=", 0) == 0) { dst = a.substr(l.size() + 1); return true; }
      if ((a == l || (sht && a == sht)) && i + 1 < argc) { dst = argv[++i]; return true; }
      return false;
    };
    std::string v;
    if (val("port", nullptr, v)) { port = atoi(v.c_str()); continue; }
    if (val("max_batch", nullptr, v)) { max_batch = std::max(1, atoi(v.c_str())); continue; }
    if (val("batch_wait_ms", nullptr, v)) { wait_ms = std::max(0, atoi(v.c_str())); continue; }
    if (val("model_type", "-t", model_type) || val("model_path", "-p", model_path) || val("language", "-l", language)) continue;
    if (val("devices", nullptr, devices) || val("scheduler", nullptr, scheduler)) continue;
    if (val("steps_per_call", nullptr, v)) { steps_per_call = std::max(1, atoi(v.c_str())); continue; }
    if (val("min_admit", nullptr, v)) { min_admit = std::max(1, atoi(v.c_str())); continue; }
    if (val("admit_wait_ms", nullptr, v)) { admit_wait_ms = std::max(0, atoi(v.c_str())); continue; }
    if (val("max_conns", nullptr, v)) { g_max_conns = std::max(1, atoi(v.c_str())); continue; }
    if (val("max_body_mb", nullptr, v)) { max_body_mb = std::max(1, atoi(v.c_str())); continue; }
    if (val("recv_timeout_s", nullptr, v)) { g_recv_timeout_s = std::max(1, atoi(v.c_str())); continue; }
    fprintf(stderr, "usage: %s [--port 8080] [-t model_type] [-p model_path] [-l language] [--max_batch 64] [--batch_wait_ms 5] [--devices all|0,1,..]\n"
                    "          [--scheduler slots|batches] [--steps_per_call 8] [--min_admit 1] [--admit_wait_ms 20] [--max_conns 256] [--max_body_mb 16] [--recv_timeout_s 10]\n", argv[0]);
    return a == "--help" || a == "-?" ? 0 : 1;
  }
  printf("port: %d\n", port);
  printf("model_path: %s\n", model_path.c_str());
  printf("model_type: %s\n", model_type.c_str());
  printf("language: %s\n", language.c_str());

  // one handle (one engine, one batcher thread) per device
  std::vector<int> devs;
  if (devices.empty()) {
    devs.push_back(-1);  // AX_WHISPER_Init's default device
  } else if (devices == "all") {
    for (int d = 0; d < AX_WHISPER_VisibleDeviceCount(); ++d) devs.push_back(d);
  } else {
    size_t p = 0;
    while (p < devices.size()) {
      size_t q = devices.find(',', p);
      if (q == std::string::npos) q = devices.size();
      devs.push_back(atoi(devices.substr(p, q - p).c_str()));
      p = q + 1;
    }
  }
  std::vector<AX_WHISPER_HANDLE> models;
  for (int d : devs) {
    // (the slot stream runs at least three slots underneath: capacity for them is allocated here, not at the first request)
    AX_WHISPER_HANDLE m = AX_WHISPER_InitEx(model_type.c_str(), model_path.c_str(), language.c_str(), d, std::max(max_batch, 3));
    if (!m) {
      printf("init server failed!\n");
      for (AX_WHISPER_HANDLE o : models) AX_WHISPER_Uninit(o);
      return -1;
    }
    models.push_back(m);
  }
  if (models.empty()) { printf("init server failed!\n"); return -1; }
  printf("devices: %d\n", (int)models.size());
  if (scheduler != "slots" && scheduler != "batches") { printf("--scheduler must be slots or batches\n"); return 1; }
  printf("scheduler: %s\n", scheduler.c_str());
  g_models = models;
  g_max_body = (size_t)max_body_mb << 20;

  int srv = socket(AF_INET, SOCK_STREAM, 0);
  int one = 1;
  setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  addr.sin_addr.s_addr = htonl(INADDR_ANY);  // the reference listens on 0.0.0.0 (hpp:33)
  addr.sin_port = htons((uint16_t)port);
  if (bind(srv, (sockaddr*)&addr, sizeof addr) != 0 || listen(srv, 128) != 0) { perror("bind/listen"); return -1; }
  printf("Start server at port %d, POST binary stream to IP:%d/asr\n", port, port);
  fflush(stdout);

  std::vector<std::thread> bts;
  for (AX_WHISPER_HANDLE m : models) {
    if (scheduler == "slots" && max_batch >= 2) bts.emplace_back(slot_scheduler, m, max_batch, wait_ms, steps_per_call, min_admit, admit_wait_ms);
    else bts.emplace_back(batcher, m, max_batch, wait_ms);
  }
  signal(SIGPIPE, SIG_IGN);
  for (;;) {
    int fd = accept(srv, nullptr, nullptr);
    if (fd < 0) break;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    timeval tv{g_recv_timeout_s, 0};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);  // a peer that stops sending holds its thread this long at most
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
    if (++g_conns > g_max_conns) {  // one thread per connection, so connections are capped
      send_response(fd, 503, R"({"error": "Too many connections"})");
      shutdown(fd, SHUT_RDWR);
      close(fd);
      --g_conns;
      continue;
    }
    std::thread(serve, fd).detach();
  }
  g_stop = true;
  g_cv.notify_all();
  for (auto& t : bts) t.join();
  for (AX_WHISPER_HANDLE m : models) AX_WHISPER_Uninit(m);
  return 0;
}
