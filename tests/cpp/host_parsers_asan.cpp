// host_parsers_asan.cpp — the host byte parsers under AddressSanitizer + UndefinedBehaviorSanitizer with seeded mutations.
//
// Built by tests/test_host_parsers_sanitized.py with plain g++ (-fsanitize=address,undefined -fno-sanitize-recover=all):
// CPU only, never on the GPU box. Every parser that consumes bytes an outsider controls runs on its well-formed fixtures and
// on truncated / bit-flipped / length-smashed mutations of them; a parser may REFUSE an input (return false, throw
// std::exception) but may not crash, read out of bounds, overflow, hang or invoke undefined behaviour.
//   csrc/host_io.hpp      JSON, safetensors header, token table (base64), WAV, AIFF / AIFF-C
//   csrc/t2s.hpp          OpenCC .ocd2 dictionaries (marisa trie image) + the conversion over a loaded dictionary
//   csrc/http_request.hpp whisper_srv's request head (network-facing)
// The reference's counterparts: cpp/src/AudioFile.h:450-776, cpp/src/utils/WhisperHTTPServer.hpp:50-71.
//
//   host_parsers_asan <fixture dir> <cases per fixture> <seed>
// <fixture dir>: *.wav *.aiff (audio), *.tiktoken (token table), *.ocd2, *.json, *.safetensors
#include <dirent.h>
#include <unistd.h>

#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

#include "host_io.hpp"
#include "http_request.hpp"
#include "t2s.hpp"

using Bytes = std::vector<uint8_t>;

static uint64_t g_rng = 1;
static uint64_t rnd() {  // xorshift64*: seeded, the same cases on every run
  g_rng ^= g_rng >> 12; g_rng ^= g_rng << 25; g_rng ^= g_rng >> 27;
  return g_rng * 2685821657736338717ull;
}
static size_t below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

static Bytes read_file(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return Bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// one mutation of `in`: the kinds a damaged or hostile file shows
static Bytes mutate(const Bytes& in) {
  Bytes b = in;
  if (b.empty()) return b;
  switch (below(7)) {
    case 0: b.resize(below(b.size())); break;                                      // truncation anywhere
    case 1: for (int i = 0, n = 1 + (int)below(8); i < n; ++i) b[below(b.size())] ^= (uint8_t)(1u << below(8)); break;  // bit flips
    case 2: for (int i = 0, n = 1 + (int)below(4); i < n; ++i) {                   // interesting bytes
      static const uint8_t v[] = {0x00, 0xFF, 0x7F, 0x80, 0x01, 0xFE, '"', '\\', '[', '{', '=', '\n', '\r'};
      b[below(b.size())] = v[below(sizeof v)];
    } break;
    case 3: {                                                                     // a 32-bit field smashed to a boundary value
      static const uint32_t v[] = {0xFFFFFFFFu, 0xFFFFFFF8u, 0x80000000u, 0x7FFFFFFFu, 0u, 1u, 0xFFFFFFF7u, 0x00010000u};
      if (b.size() >= 4) { size_t o = below(b.size() - 3); uint32_t x = v[below(8)]; if (below(2)) x = __builtin_bswap32(x); memcpy(&b[o], &x, 4); }
    } break;
    case 4: {                                                                     // a 64-bit field smashed
      static const uint64_t v[] = {~0ull, ~0ull - 7, 1ull << 63, (1ull << 63) - 1, 1ull << 32, 0ull};
      if (b.size() >= 8) { size_t o = below(std::min<size_t>(b.size() - 7, 256)); uint64_t x = v[below(6)]; memcpy(&b[o], &x, 8); }
    } break;
    case 5: {                                                                     // a block duplicated or dropped
      size_t o = below(b.size()), n = 1 + below(std::min<size_t>(64, b.size() - o));
      if (below(2)) b.insert(b.begin() + (long)o, in.begin() + (long)o, in.begin() + (long)(o + n));
      else b.erase(b.begin() + (long)o, b.begin() + (long)(o + n));
    } break;
    default: {                                                                    // a run of random bytes
      size_t o = below(b.size()), n = 1 + below(std::min<size_t>(32, b.size() - o));
      for (size_t i = 0; i < n; ++i) b[o + i] = (uint8_t)rnd();
    }
  }
  return b;
}

static long g_cases = 0, g_refused = 0;
template <class F>
static void guarded(F&& f) {
  ++g_cases;
  try {
    if (!f()) ++g_refused;
  } catch (const std::exception&) {
    ++g_refused;
  }
}

static void run_audio(const Bytes& b) {
  guarded([&] {
    axw::WavData w; std::string err;
    const bool ok = b.size() >= 4 && !memcmp(b.data(), "FORM", 4) ? axw::load_aiff(b, w, err) : axw::load_wav_bytes(b, w, err);
    if (ok) { volatile float acc = 0; for (float v : w.mono) acc = acc + v; (void)acc; }  // every sample was really written
    return ok;
  });
}
static void run_tokens(const Bytes& b) {
  guarded([&] {
    std::istringstream is(std::string(b.begin(), b.end()));
    auto t = axw::parse_token_table(is);
    size_t n = 0; for (auto& s : t) n += s.size();
    return n > 0;
  });
}
static void run_json(const Bytes& b) {
  guarded([&] {
    const std::string s(b.begin(), b.end());
    axw::JsonValue j = axw::JsonParser(s).parse();
    if (j.kind == axw::JsonValue::Object) for (auto& kv : j.obj) { (void)kv.second.has("x"); if (kv.second.kind == axw::JsonValue::Number) (void)kv.second.as_int(); }
    return true;
  });
}
static void run_safetensors(const Bytes& b) {
  guarded([&] {
    std::map<std::string, axw::TensorView> t;
    axw::SafeTensors::parse_image(b.data(), b.size(), t);
    // what the engine does with a view: read its first and last byte
    volatile unsigned acc = 0;
    for (auto& kv : t) if (kv.second.nbytes) acc = acc + kv.second.data[0] + kv.second.data[kv.second.nbytes - 1];
    (void)acc;
    return !t.empty();
  });
}
static void run_ocd2(const Bytes& b) {
  guarded([&] {
    axw::T2SDict d = axw::T2SDict::load_ocd2_bytes(std::string(b.begin(), b.end()), "fuzz");
    size_t len = 0;
    const std::string text = "\xE8\xAA\xAA\xE8\xA9\xB1 test \xE4\xBA\xA4\xE6\x98\x93\xE5\xB9\xBE\xE4\xB9\x8E\xE5\x81\x9C\xE6\xAD\xA2";
    for (size_t p = 0; p < text.size(); ++p) (void)d.match_prefix(text, p, &len);
    return !d.map.empty();
  });
}
static void run_http(const Bytes& b) {
  guarded([&] {
    const std::string s(b.begin(), b.end());
    axw::HttpHead h;
    if (!axw::parse_http_head(s, h)) return false;
    (void)axw::http_route(h);
    const size_t body = s.size() - (h.header_end + 4);
    return axw::asr_request_error(h, body) == nullptr && h.length_ok;
  });
}

static bool ends_with(const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && !s.compare(s.size() - n, n, suf); }

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <fixture dir> <cases per fixture> <seed>\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  const long per = atol(argv[2]);
  g_rng = strtoull(argv[3], nullptr, 10) * 2654435761ull + 88172645463325252ull;
  alarm(900);  // a parser that hangs on some input ends the run (SIGALRM) instead of the test suite's patience

  std::vector<std::string> names;
  if (DIR* d = opendir(dir.c_str())) {
    while (dirent* e = readdir(d)) if (e->d_name[0] != '.') names.push_back(e->d_name);
    closedir(d);
  }
  std::sort(names.begin(), names.end());
  long fixtures = 0;
  for (const std::string& n : names) {
    const Bytes b = read_file(dir + "/" + n);
    void (*run)(const Bytes&) = nullptr;
    long mult = 1;
    if (ends_with(n, ".wav") || ends_with(n, ".aiff")) run = run_audio;
    else if (ends_with(n, ".tiktoken")) { run = run_tokens; }
    else if (ends_with(n, ".ocd2")) { run = run_ocd2; }
    else if (ends_with(n, ".json")) { run = run_json; mult = 4; }
    else if (ends_with(n, ".safetensors")) { run = run_safetensors; mult = 4; }
    else if (ends_with(n, ".http")) { run = run_http; mult = 4; }
    if (!run) continue;
    ++fixtures;
    const long before = g_refused;
    run(b);  // the fixture itself
    if (g_refused != before && !ends_with(n, ".bad.http")) { fprintf(stderr, "fixture %s was refused unmutated\n", n.c_str()); return 1; }
    // big fixtures (the token table, the phrase dictionary) cost milliseconds per parse: fewer cases each
    long cases = per * mult;
    if (b.size() > (1u << 18)) cases = std::max<long>(per / 8, 16);
    for (long i = 0; i < cases; ++i) run(mutate(b));
    // mutations of mutations: damage accumulates
    Bytes m = b;
    for (long i = 0; i < cases / 4; ++i) { m = mutate(m); if (m.empty()) m = b; run(m); }
  }
  // JSON / HTTP shapes no fixture holds
  {
    std::string deep(200000, '[');
    run_json(Bytes(deep.begin(), deep.end()));
    std::string deepo;
    for (int i = 0; i < 50000; ++i) deepo += "{\"a\":";
    run_json(Bytes(deepo.begin(), deepo.end()));
    const char* heads[] = {
        "POST /asr HTTP/1.1\r\nContent-Length: -1\r\n\r\n", "POST /asr HTTP/1.1\r\nContent-Length: 99999999999999999999999\r\n\r\n",
        "POST /asr HTTP/1.1\r\nX-Content-Length: 8\r\nContent-Type: application/octet-stream\r\n\r\nabcd", "\r\n\r\n", "\r\n\r\n\r\n",
        "POST /asr HTTP/1.1\r\nContent-Length:\r\n\r\n", "POST /asr HTTP/1.1\r\nContent-Length: 0x10\r\n\r\n", "content-length: 5\r\n\r\n",
        "GET /health HTTP/1.1\r\n\r\n", "OPTIONS * HTTP/1.1\r\n\r\n", "POST /asr HTTP/1.1\r\nContent-Type:\r\nContent-Length: 4\r\n\r\nabcd"};
    for (const char* h : heads) { const std::string s(h); run_http(Bytes(s.begin(), s.end())); }
  }
  printf("host parsers sanitized: %ld fixtures, %ld cases, %ld refused, 0 faults\n", fixtures, g_cases, g_refused);
  return 0;
}
