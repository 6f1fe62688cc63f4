// ref_bytes_driver.cpp — builds the REFERENCE's own byte paths into oracle/_ref/libref_bytes.so.
//
// TEST INFRASTRUCTURE ONLY. This file is ours; what it compiles against stays where it lies under /root/reference and is
// never copied into this repository (oracle/Makefile target `ref`, outputs into oracle/_ref/ only):
//   * cpp/src/AudioFile.h        — WAV / AIFF decode (load :450,501; int16 / 32768 :1241-1243; 8/24/32-bit and float paths)
//   * cpp/src/base64.cpp (+ .h)  — base64_decode of a token-table entry (:84-120), compiled as its own object
// Two pieces of the reference sit in files that cannot be compiled here (they include the closed AXera BSP headers) and are
// restated below, line by line, with their citations:
//   * cpp/src/api/ax_whisper_api.cpp:105-113 — channel 0, and (L + R) / 2 in place when the file is stereo
//   * cpp/src/Whisper.cpp:115-127            — the token file loop: one line per id, the text up to the first blank
//   * cpp/src/Whisper.cpp:224-229            — ids -> bytes: base64_decode of every id's entry, appended as a C string
#include <limits>  // AudioFile.h uses std::numeric_limits without including it (the reference gets it transitively)

#include <AudioFile.h>
#include <base64.h>

#include <cstring>
#include <fstream>
#include <string>
#include <vector>

// -> number of mono samples (or -1); out may be NULL to ask for the size. info: [0] sample rate, [1] channels, [2] bit depth
extern "C" int ref_load_audio_mono(const char* path, float* out, int cap, int* info) {
  AudioFile<float> audio_file;
  audio_file.shouldLogErrorsToConsole(false);
  if (!audio_file.load(path)) return -1;                       // ax_whisper_api.cpp:100-103
  auto& samples = audio_file.samples[0];                       // :105
  int n_samples = (int)samples.size();                         // :106
  if (audio_file.isStereo())                                   // :109
    for (int i = 0; i < n_samples; i++) samples[i] = (samples[i] + audio_file.samples[1][i]) / 2;  // :110-112
  if (info) { info[0] = (int)audio_file.getSampleRate(); info[1] = audio_file.getNumChannels(); info[2] = audio_file.getBitDepth(); }
  if (out) std::memcpy(out, samples.data(), sizeof(float) * (size_t)std::min(cap, n_samples));
  return n_samples;
}

// the reference's base64_decode on one table entry -> decoded length j (its return value); str receives what its strcpy
// leaves there (the bytes up to the first NUL)
extern "C" int ref_base64_decode(const char* code, int code_len, char* str /*[1024]*/) {
  return base64_decode((const uint8*)code, (uint32)code_len, str);
}

static std::vector<std::string> g_tokens;
// Whisper.cpp:115-127 -> number of table entries
extern "C" int ref_load_tokens(const char* token_path) {
  g_tokens.clear();
  std::ifstream fs(token_path);
  if (!fs.is_open()) return -1;
  std::string line;
  while (std::getline(fs, line)) {
    size_t i = line.find(' ');
    g_tokens.push_back(line.substr(0, i));
  }
  return (int)g_tokens.size();
}
extern "C" int ref_token_entry(int id, char* out, int cap) {
  if (id < 0 || id >= (int)g_tokens.size()) return -1;
  std::snprintf(out, cap, "%s", g_tokens[id].c_str());
  return (int)g_tokens[id].size();
}
// Whisper.cpp:224-229 with the buffer wide enough for every entry (the reference's char str[32] overflows on the one
// 33-byte token, SURVEY B8: undefined behaviour is not a parity target) -> bytes written (no terminator counted)
extern "C" int ref_detokenize(const int* ids, int n, char* out, int cap) {
  std::string s;
  for (int k = 0; k < n; ++k) {
    const int i = ids[k];
    if (i < 0 || i >= (int)g_tokens.size()) continue;           // the reference would index out of bounds (B8)
    if (g_tokens[i].size() & 3) continue;                       // "=" of the last line (id 50256, decodes to nothing): base64.cpp:87
                                                                // asserts on it; with NDEBUG it reads past the entry
    char str[1024];
    base64_decode((const uint8*)g_tokens[i].c_str(), (uint32)g_tokens[i].size(), str);
    s += str;
  }
  const int m = (int)std::min<size_t>(s.size(), (size_t)cap);
  std::memcpy(out, s.data(), (size_t)m);
  return m;
}
