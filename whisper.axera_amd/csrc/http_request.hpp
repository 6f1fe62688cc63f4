// http_request.hpp — the request-head parsing and the /asr request validation of whisper_srv, as pure functions over bytes
// (no sockets), so that they can be exercised by the sanitizer / mutation harness (tests/cpp/host_parsers_asan.cpp).
//
// The contract is the reference server's (cpp/src/utils/WhisperHTTPServer.hpp:50-71): POST /asr with
// Content-Type application/octet-stream and a non-empty body of whole f32 samples; its error strings are kept verbatim.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <string>

namespace axw {

struct HttpHead {
  std::string first;            // the request line
  std::string content_type;     // lower-cased value of Content-Type ("" if absent)
  size_t content_length = 0;    // 0 if absent or malformed
  bool length_ok = true;        // false: a Content-Length header is present but is not a plain decimal number
  bool expect_continue = false;
  size_t header_end = std::string::npos;  // offset of the blank line's "\r\n\r\n" in the buffer
};

inline std::string http_lower(std::string s) {
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)(c >= 'A' && c <= 'Z' ? c + 32 : c); });
  return s;
}

// Value of header `name` (lower case, with the colon) in the lower-cased head: matched at the START of a header line only
// ("x-content-length: 9" is not a Content-Length), leading blanks trimmed, up to the line's end. found = false if absent.
inline std::string http_header_value(const std::string& lhead, const std::string& name, bool* found = nullptr) {
  size_t p = 0;
  if (found) *found = false;
  for (;;) {
    p = lhead.find(name, p);
    if (p == std::string::npos) return std::string();
    if (p >= 2 && lhead[p - 2] == '\r' && lhead[p - 1] == '\n') break;  // (p == 0 would be the request line: never a header)
    p += 1;
  }
  size_t b = p + name.size();
  size_t e = lhead.find("\r\n", b);
  if (e == std::string::npos) e = lhead.size();
  while (b < e && (lhead[b] == ' ' || lhead[b] == '\t')) ++b;
  while (e > b && (lhead[e - 1] == ' ' || lhead[e - 1] == '\t')) --e;
  if (found) *found = true;
  return lhead.substr(b, e - b);
}

// true once `buf` holds a complete head (h.header_end is set); false: more bytes are needed
inline bool parse_http_head(const std::string& buf, HttpHead& h) {
  h = HttpHead{};
  const size_t he = buf.find("\r\n\r\n");
  if (he == std::string::npos) return false;
  h.header_end = he;
  const std::string head = buf.substr(0, he);
  const std::string lhead = http_lower(head);
  h.first = head.substr(0, head.find("\r\n"));
  bool has_len = false;
  const std::string cl = http_header_value(lhead, "content-length:", &has_len);
  if (has_len) {
    // a plain decimal number of at most 15 digits: no sign, no hex, nothing strtoul would silently accept or wrap
    h.length_ok = !cl.empty() && cl.size() <= 15 && std::all_of(cl.begin(), cl.end(), [](char c) { return c >= '0' && c <= '9'; });
    if (h.length_ok) h.content_length = (size_t)strtoull(cl.c_str(), nullptr, 10);
  }
  h.content_type = http_header_value(lhead, "content-type:");
  h.expect_continue = http_header_value(lhead, "expect:").find("100-continue") != std::string::npos;
  return true;
}

enum class HttpRoute { Health, Options, Asr, NotFound };
inline HttpRoute http_route(const HttpHead& h) {
  if (h.first.rfind("GET /health", 0) == 0) return HttpRoute::Health;
  if (h.first.rfind("OPTIONS ", 0) == 0) return HttpRoute::Options;
  if (h.first.rfind("POST /asr", 0) == 0) return HttpRoute::Asr;
  return HttpRoute::NotFound;
}

// POST /asr: nullptr if the request can be served, else the reference's error body (status 400)
inline const char* asr_request_error(const HttpHead& h, size_t body_bytes) {
  if (h.content_type.find("application/octet-stream") == std::string::npos)              // hpp:50-55
    return R"({"error": "Content-Type must be application/octet-stream"})";
  if (body_bytes == 0) return R"({"error": "Request body is empty"})";                     // hpp:58-62
  if (body_bytes % sizeof(float) != 0) return R"({"error": "Data size must be multiple of 4 bytes"})";  // hpp:65-71
  return nullptr;
}

}  // namespace axw
