// host_io.hpp — host-side file formats of the model directory and the audio input.
//
//   * a small JSON reader for {type}_config.json (the reference uses nlohmann::json,
//     cpp/src/Whisper.cpp:98-100; only objects/strings/numbers are needed here)
//   * a safetensors reader (mmap) for {type}.safetensors
//   * {type}-tokens.txt: "<base64> <rank>" lines (cpp/src/Whisper.cpp:115-127) + base64 decode
//     (cpp/src/base64.cpp:84-120), bounds-checked (fixes SURVEY B8)
//   * WAV and AIFF readers with the reference's sample conventions (cpp/src/AudioFile.h:1241-1243:
//     int16 -> /32768; ax_whisper_api.cpp:105-113: stereo -> (L+R)/2; AIFF: AudioFile.h:643-776)
#pragma once

#include <array>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace axw {

// ------------------------------------------------------------------------------- JSON
struct JsonValue {
  enum Kind { Null, Bool, Number, String, Object, Array } kind = Null;
  double num = 0;
  bool b = false;
  std::string str;
  std::map<std::string, JsonValue> obj;
  std::vector<JsonValue> arr;

  bool has(const std::string& k) const { return kind == Object && obj.count(k); }
  const JsonValue& at(const std::string& k) const {
    auto it = obj.find(k);
    if (kind != Object || it == obj.end()) throw std::runtime_error("json: missing key '" + k + "'");
    return it->second;
  }
  long as_int() const {
    if (kind == Number) {
      if (!(num >= -9.2e18 && num <= 9.2e18)) throw std::runtime_error("json: number out of range");  // (also NaN) a cast outside long's range is undefined
      return (long)num;
    }
    if (kind == String) {
      try { return std::stol(str); } catch (const std::exception&) { throw std::runtime_error("json: not a number"); }
    }
    if (kind == Bool) return b;
    throw std::runtime_error("json: not a number");
  }
  const std::string& as_str() const {
    if (kind != String) throw std::runtime_error("json: not a string");
    return str;
  }
};

class JsonParser {
 public:
  explicit JsonParser(const std::string& s) : s_(s) {}
  JsonValue parse() {
    JsonValue v = value();
    ws();
    if (p_ != s_.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& s_;
  size_t p_ = 0;
  [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("json: ") + m + " at " + std::to_string(p_)); }
  void ws() { while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\t' || s_[p_] == '\r')) ++p_; }
  char peek() { ws(); if (p_ >= s_.size()) fail("unexpected end"); return s_[p_]; }
  int depth_ = 0;
  struct Depth {  // containers nest by recursion: a hostile header of 100 000 '[' must not exhaust the stack
    JsonParser& p;
    explicit Depth(JsonParser& q) : p(q) { if (++p.depth_ > 64) p.fail("nesting too deep"); }
    ~Depth() { --p.depth_; }
  };
  JsonValue value() {
    Depth guard(*this);
    char c = peek();
    JsonValue v;
    if (c == '{') {
      v.kind = JsonValue::Object; ++p_;
      if (peek() == '}') { ++p_; return v; }
      for (;;) {
        if (peek() != '"') fail("expected key");
        std::string k = string();
        if (peek() != ':') fail("expected ':'");
        ++p_;
        v.obj[k] = value();
        char d = peek(); ++p_;
        if (d == '}') break;
        if (d != ',') fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      v.kind = JsonValue::Array; ++p_;
      if (peek() == ']') { ++p_; return v; }
      for (;;) {
        v.arr.push_back(value());
        char d = peek(); ++p_;
        if (d == ']') break;
        if (d != ',') fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      v.kind = JsonValue::String; v.str = string();
    } else if (c == 't' && s_.compare(p_, 4, "true") == 0) { v.kind = JsonValue::Bool; v.b = true; p_ += 4;
    } else if (c == 'f' && s_.compare(p_, 5, "false") == 0) { v.kind = JsonValue::Bool; v.b = false; p_ += 5;
    } else if (c == 'n' && s_.compare(p_, 4, "null") == 0) { p_ += 4;
    } else {
      size_t e = p_;
      while (e < s_.size() && (isdigit((unsigned char)s_[e]) || strchr("+-.eE", s_[e]))) ++e;
      if (e == p_) fail("unexpected character");
      v.kind = JsonValue::Number;
      try { v.num = std::stod(s_.substr(p_, e - p_)); } catch (const std::exception&) { fail("bad number"); }
      p_ = e;
    }
    return v;
  }
  std::string string() {
    std::string out; ++p_;
    while (p_ < s_.size() && s_[p_] != '"') {
      char c = s_[p_++];
      if (c == '\\') {
        if (p_ >= s_.size()) fail("bad escape");
        char e = s_[p_++];
        switch (e) {
          case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
          case 'b': out += '\b'; break; case 'f': out += '\f'; break;
          case 'u': {
            if (p_ + 4 > s_.size()) fail("bad \\u");
            unsigned cp = 0;
            for (int i = 0; i < 4; ++i) {
              const char h = s_[p_ + i];
              const int dv = h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : -1;
              if (dv < 0) fail("bad \\u");
              cp = cp * 16 + (unsigned)dv;
            }
            p_ += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += e;
        }
      } else out += c;
    }
    if (p_ >= s_.size()) fail("unterminated string");
    ++p_;
    return out;
  }
};

inline std::string read_text_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) throw std::runtime_error("cannot open " + path);
  std::stringstream ss; ss << f.rdbuf();
  return ss.str();
}

inline std::vector<std::string> split_csv(const std::string& s) {
  std::vector<std::string> out; std::string cur;
  for (char c : s) { if (c == ',') { out.push_back(cur); cur.clear(); } else cur += c; }
  if (!s.empty()) out.push_back(cur);
  return out;
}

// ------------------------------------------------------------------------------- safetensors
struct TensorView {
  std::string dtype;            // "BF16" | "F16" | "F32"
  std::vector<int64_t> shape;
  const uint8_t* data = nullptr;
  size_t nbytes = 0;
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

class SafeTensors {
 public:
  explicit SafeTensors(const std::string& path) {
    try {
      parse(path);
    } catch (...) {
      release();  // a throwing constructor never runs the destructor: give the mapping and the descriptor back here
      throw;
    }
  }
  ~SafeTensors() { release(); }
  SafeTensors(const SafeTensors&) = delete;
  bool has(const std::string& n) const { return tensors_.count(n) != 0; }
  // Read the whole file into the page cache now (one byte per page behind a WILLNEED hint), so that the uploads that follow —
  // which the engine runs under its per-device allocation / capture mutex — copy from memory and never wait for the disk.
  size_t page_in() const {
    (void)madvise((void*)base_, size_, MADV_WILLNEED);
    size_t acc = 0;
    for (size_t o = 0; o < size_; o += 4096) acc += base_[o];
    return acc;
  }
  const TensorView& get(const std::string& n) const {
    auto it = tensors_.find(n);
    if (it == tensors_.end()) throw std::runtime_error("weights file lacks tensor '" + n + "'");
    return it->second;
  }

 private:
  void release() {
    if (base_ && base_ != MAP_FAILED) munmap((void*)base_, size_);
    base_ = nullptr;
    if (fd_ >= 0) close(fd_);
    fd_ = -1;
  }
  void parse(const std::string& path) {
    fd_ = open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw std::runtime_error("cannot open weights file " + path);
    struct stat st;
    if (fstat(fd_, &st) != 0) throw std::runtime_error("cannot stat weights file " + path);
    size_ = (size_t)st.st_size;
    if (size_ < 8) throw std::runtime_error("weights file too small");
    base_ = (const uint8_t*)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (base_ == MAP_FAILED) { base_ = nullptr; throw std::runtime_error("mmap failed for " + path); }
    parse_image(base_, size_, tensors_);
  }

 public:
  // The header of a safetensors image [base, base + size): every tensor's dtype, shape and byte range, checked against the
  // image's bounds (the views point into the image). Separate from the mmap so that tests can hand it arbitrary bytes.
  static void parse_image(const uint8_t* base, size_t size, std::map<std::string, TensorView>& tensors) {
    if (size < 8) throw std::runtime_error("weights file too small");
    uint64_t hl; memcpy(&hl, base, 8);
    if (hl > size - 8) throw std::runtime_error("bad safetensors header length");  // no 8 + hl: it can wrap
    std::string hdr((const char*)base + 8, (size_t)hl);
    JsonValue j = JsonParser(hdr).parse();
    if (j.kind != JsonValue::Object) throw std::runtime_error("safetensors header is not an object");
    const uint8_t* data0 = base + 8 + hl;
    const size_t data_bytes = size - 8 - (size_t)hl;
    for (auto& kv : j.obj) {
      if (kv.first == "__metadata__") continue;
      TensorView t;
      t.dtype = kv.second.at("dtype").as_str();
      size_t numel = 1;
      for (auto& d : kv.second.at("shape").arr) {
        if (d.as_int() < 0) throw std::runtime_error("tensor '" + kv.first + "': negative dimension");
        t.shape.push_back(d.as_int());
        if (d.as_int() != 0 && numel > (size_t)1 << 48) throw std::runtime_error("tensor '" + kv.first + "': shape too large");  // the product below cannot wrap
        numel *= (size_t)d.as_int();
        if (numel > (size_t)1 << 48) throw std::runtime_error("tensor '" + kv.first + "': shape too large");
      }
      auto& off = kv.second.at("data_offsets").arr;
      if (off.size() != 2) throw std::runtime_error("tensor '" + kv.first + "': bad data_offsets");
      if (off.at(0).as_int() < 0 || off.at(1).as_int() < 0) throw std::runtime_error("tensor '" + kv.first + "': negative offset");
      size_t s = (size_t)off.at(0).as_int(), e = (size_t)off.at(1).as_int();
      if (e < s || e > data_bytes) throw std::runtime_error("tensor '" + kv.first + "' out of file bounds");
      t.data = data0 + s; t.nbytes = e - s;
      size_t es = t.dtype == "F32" ? 4 : (t.dtype == "BF16" || t.dtype == "F16") ? 2 : 0;
      if (!es) throw std::runtime_error("tensor '" + kv.first + "': unsupported dtype " + t.dtype);
      if (numel * es != t.nbytes) throw std::runtime_error("tensor '" + kv.first + "': size mismatch");
      tensors[kv.first] = t;
    }
  }

 private:
  int fd_ = -1; size_t size_ = 0; const uint8_t* base_ = nullptr;
  std::map<std::string, TensorView> tensors_;
};

// ------------------------------------------------------------------------------- tokens
inline bool base64_decode(const std::string& in, std::string& out) {
  // a magic static: built once, thread-safe (the engines of AX_WHISPER_InitMulti load their token tables side by side)
  static const std::array<int8_t, 256> map = [] {
    std::array<int8_t, 256> m;
    m.fill(-1);
    const char* a = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    for (int i = 0; i < 64; ++i) m[(unsigned char)a[i]] = (int8_t)i;
    return m;
  }();
  out.clear();
  uint32_t acc = 0; int bits = 0;
  for (unsigned char c : in) {
    if (c == '=') break;
    int v = map[c];
    if (v < 0) return false;
    acc = (acc << 6) | (uint32_t)v; bits += 6;
    if (bits >= 8) { bits -= 8; out += (char)((acc >> bits) & 0xFF); }
  }
  return true;
}

// Decoded byte strings, index = rank (= line index). The reference keeps the base64 text and
// decodes per token at run time (Whisper.cpp:224-229); decoding once at load is equivalent.
inline std::vector<std::string> parse_token_table(std::istream& f);
inline std::vector<std::string> load_token_table(const std::string& path) {
  std::ifstream f(path);
  if (!f.is_open()) throw std::runtime_error("cannot open tokens file " + path);
  return parse_token_table(f);
}
inline std::vector<std::string> parse_token_table(std::istream& f) {
  std::vector<std::string> table; std::string line;
  while (std::getline(f, line)) {
    size_t i = line.find(' ');
    std::string b64 = line.substr(0, i), bytes;
    if (!base64_decode(b64, bytes)) throw std::runtime_error("bad base64 in tokens file line " + std::to_string(table.size()));
    // the reference appends an entry as a C string (base64.cpp:117 strcpy, Whisper.cpp:228 `s += str`): bytes from a NUL on
    // never reach the text (id 188 decodes to one NUL byte: it contributes nothing, and the text goes on behind it)
    bytes.resize(strlen(bytes.c_str()));
    table.push_back(bytes);
  }
  return table;
}

// ------------------------------------------------------------------------------- WAV
struct WavData { int sample_rate = 0; int channels = 0; std::vector<float> mono; };

inline bool load_wav_bytes(const std::vector<uint8_t>& d, WavData& out, std::string& err);
inline bool load_wav(const std::string& path, WavData& out, std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) { err = "cannot open " + path; return false; }
  std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  return load_wav_bytes(d, out, err);
}
inline bool load_wav_bytes(const std::vector<uint8_t>& d, WavData& out, std::string& err) {
  if (d.size() < 44 || memcmp(d.data(), "RIFF", 4) || memcmp(d.data() + 8, "WAVE", 4)) { err = "not a RIFF/WAVE file"; return false; }
  auto u16 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8); };
  auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
  size_t p = 12; int fmt = 0, ch = 0, bits = 0, rate = 0; size_t data_off = 0, data_len = 0;
  while (p + 8 <= d.size()) {
    uint32_t len = u32(p + 4);
    if (!memcmp(d.data() + p, "fmt ", 4) && p + 8 + 16 <= d.size()) {
      fmt = u16(p + 8); ch = u16(p + 10); rate = u32(p + 12); bits = u16(p + 22);
      if (fmt == 0xFFFE && len >= 26 && p + 8 + 26 <= d.size()) fmt = u16(p + 8 + 24);  // WAVE_FORMAT_EXTENSIBLE sub-format
    } else if (!memcmp(d.data() + p, "data", 4)) {
      data_off = p + 8; data_len = std::min<size_t>(len, d.size() - data_off); break;
    }
    p += (size_t)8 + (size_t)len + (len & 1);  // in size_t: 8 + len wraps in 32 bits (a chunk length of 0xFFFFFFF8 would never advance)
  }
  if (!data_off || ch < 1 || !(fmt == 1 || fmt == 3)) { err = "unsupported WAV (need PCM or IEEE float)"; return false; }
  int bps = bits / 8;
  if (bits % 8 != 0) { err = "unsupported bit depth"; return false; }
  if (!((fmt == 1 && (bps == 1 || bps == 2 || bps == 3 || bps == 4)) || (fmt == 3 && bps == 4))) { err = "unsupported bit depth"; return false; }
  size_t frames = data_len / ((size_t)bps * ch);
  out.sample_rate = rate; out.channels = ch; out.mono.resize(frames);
  auto sample = [&](size_t frame, int c) -> float {
    const uint8_t* s = d.data() + data_off + (frame * ch + c) * bps;
    if (fmt == 3) { float v; memcpy(&v, s, 4); return v; }
    if (bps == 1) return ((int)s[0] - 128) / 128.f;
    if (bps == 2) { int16_t v; memcpy(&v, s, 2); return (float)v / 32768.f; }  // AudioFile.h:1241-1243
    if (bps == 3) { int32_t v = (s[0] | (s[1] << 8) | (s[2] << 16)); if (v & 0x800000) v |= ~0xFFFFFF; return (float)v / 8388608.f; }
    int32_t v; memcpy(&v, s, 4); return (float)((double)v / 2147483648.0);
  };
  for (size_t i = 0; i < frames; ++i)
    out.mono[i] = (ch == 2) ? (sample(i, 0) + sample(i, 1)) / 2 : sample(i, 0);  // api.cpp:105-113
  return true;
}

// AIFF / AIFF-C (the reference's AudioFile reads both, cpp/src/AudioFile.h:643-776; RunFile hands it whatever the caller
// names): FORM container, big-endian chunk sizes, COMM = channels / frames / bits / 80-bit extended sample rate, SSND =
// offset + big-endian samples. Conventions as the reference: 8-bit signed / 128, 16-bit / 32768, 24-bit / 8388608, 32-bit
// integer / INT32_MAX (AIFF) or IEEE float (AIFF-C), stereo averaged by the caller's rule (ax_whisper_api.cpp:105-113).
inline bool load_aiff(const std::vector<uint8_t>& d, WavData& out, std::string& err) {
  auto be16 = [&](size_t o) { return ((uint32_t)d[o] << 8) | d[o + 1]; };
  auto be32 = [&](size_t o) { return (be16(o) << 16) | be16(o + 2); };
  if (d.size() < 12 || memcmp(d.data(), "FORM", 4)) { err = "not a FORM/AIFF file"; return false; }
  const bool aifc = !memcmp(d.data() + 8, "AIFC", 4);
  if (!aifc && memcmp(d.data() + 8, "AIFF", 4)) { err = "not a FORM/AIFF file"; return false; }
  size_t p = 12, comm = 0, ssnd = 0, ssnd_len = 0;
  while (p + 8 <= d.size()) {
    const uint32_t len = be32(p + 4);
    if (!memcmp(d.data() + p, "COMM", 4) && p + 8 + 18 <= d.size()) comm = p;
    else if (!memcmp(d.data() + p, "SSND", 4) && p + 16 <= d.size()) { ssnd = p; ssnd_len = len; }
    p += 8 + (size_t)len + (len & 1);
  }
  if (!comm || !ssnd) { err = "AIFF file lacks a COMM or SSND chunk"; return false; }
  const int ch = (int)be16(comm + 8), bits = (int)be16(comm + 14);
  const size_t frames_decl = be32(comm + 10);
  // sample rate: 80-bit IEEE 754 extended, big-endian (sign/exponent 16 bits, 64-bit mantissa with explicit integer bit)
  const int expo = (int)(be16(comm + 16) & 0x7fff) - 16383;
  uint64_t mant = ((uint64_t)be32(comm + 18) << 32) | be32(comm + 22);
  const double rate = mant == 0 ? 0.0 : std::ldexp((double)mant, expo - 63);
  if (ch < 1 || ch > 2) { err = "AIFF file is neither mono nor stereo"; return false; }
  if (bits != 8 && bits != 16 && bits != 24 && bits != 32) { err = "unsupported bit depth"; return false; }
  if (aifc) {  // only uncompressed AIFF-C: 'NONE' (big-endian integers) or 32-bit 'fl32' / 'FL32' floats
    if (comm + 8 + 22 > d.size()) { err = "truncated AIFF-C COMM chunk"; return false; }
    const uint8_t* ct = d.data() + comm + 26;
    const bool none = !memcmp(ct, "NONE", 4), fl32 = !memcmp(ct, "fl32", 4) || !memcmp(ct, "FL32", 4);
    if (!(none || (fl32 && bits == 32))) { err = "compressed AIFF-C is not supported"; return false; }
    if (none && bits == 32) { /* integers */ } else if (fl32) { /* floats, below */ }
  }
  const bool is_float = aifc && bits == 32 && (!memcmp(d.data() + comm + 26, "fl32", 4) || !memcmp(d.data() + comm + 26, "FL32", 4));
  const size_t offset = be32(ssnd + 8), start = ssnd + 16 + offset, bps = (size_t)bits / 8;
  if (start > d.size()) { err = "AIFF sound data offset out of bounds"; return false; }
  const size_t avail = std::min<size_t>(d.size() - start, ssnd_len >= 8 + offset ? ssnd_len - 8 - offset : 0);
  const size_t frames = std::min(frames_decl, avail / (bps * ch));
  // (an 80-bit rate beyond int's range — or a NaN — must not reach the cast: undefined behaviour)
  out.sample_rate = (rate >= 0.0 && rate < 2147483000.0) ? (int)(rate + 0.5) : 0; out.channels = ch; out.mono.resize(frames);
  auto sample = [&](size_t frame, int c) -> float {
    const uint8_t* s = d.data() + start + (frame * ch + c) * bps;
    if (bps == 1) return (float)(int8_t)s[0] / 128.f;
    if (bps == 2) return (float)(int16_t)((s[0] << 8) | s[1]) / 32768.f;
    if (bps == 3) { int32_t v = (s[0] << 16) | (s[1] << 8) | s[2]; if (v & 0x800000) v |= ~0xFFFFFF; return (float)v / 8388608.f; }
    const uint32_t u = ((uint32_t)s[0] << 24) | ((uint32_t)s[1] << 16) | ((uint32_t)s[2] << 8) | s[3];
    if (is_float) { float f; memcpy(&f, &u, 4); return f; }
    return (float)(int32_t)u / 2147483647.f;
  };
  for (size_t i = 0; i < frames; ++i) out.mono[i] = (ch == 2) ? (sample(i, 0) + sample(i, 1)) / 2 : sample(i, 0);
  return true;
}

// RIFF/WAVE or FORM/AIFF by the file's magic (AudioFile.h:450-501 decides the same way)
inline bool load_audio_file(const std::string& path, WavData& out, std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) { err = "cannot open " + path; return false; }
  char magic[4] = {0, 0, 0, 0};
  f.read(magic, 4);
  f.close();
  if (!memcmp(magic, "FORM", 4)) {
    std::ifstream g(path, std::ios::binary);
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(g)), std::istreambuf_iterator<char>());
    return load_aiff(d, out, err);
  }
  return load_wav(path, out, err);
}

}  // namespace axw
