// t2s.hpp — Traditional -> Simplified Chinese post-pass of zh transcripts (host side, no GPU work).
//
// Replaces `opencc::SimpleConverter converter("t2s.json"); result = converter.Convert(s);`
// (cpp/src/Whisper.cpp:231-236) without linking OpenCC (the reference ships AArch64-only static libraries).
// It reads the SAME data files the reference deploys next to its binaries — cpp/t2s.json and the two dictionaries it
// names, cpp/TSPhrases.ocd2 and cpp/TSCharacters.ocd2 — and applies OpenCC's algorithm for that configuration:
//   * dictionaries: OpenCC "ocd2" = the magic "OPENCC_MARISA_0.2.5", a marisa-trie 0.2.x image of the keys, then the
//     serialized values (u32 item count, u32 value bytes, the NUL-terminated values back to back, then per key a u16
//     value count and a u16 byte length per value); values are indexed by the trie's key id. Only enumeration is
//     needed here, so the trie is walked with plain bit scans instead of marisa's rank/select indices and caches;
//   * segmentation "mmseg": at every position the longest dictionary key that prefixes the rest of the text becomes a
//     segment; characters with no match accumulate into one segment until the next match;
//   * conversion chain: every segment is rewritten left to right — longest prefix match in the first dictionary of
//     the group that matches at all, its first value is emitted; no match: the UTF-8 character is copied.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_io.hpp"

namespace axw {

namespace ocd2 {

struct Reader {
  const std::string& b;
  size_t p;
  // (p may sit a few pad bytes beyond the end after a vector; n comes from the file: no p + n, it can wrap)
  void need(size_t n) const { if (p > b.size() || n > b.size() - p) throw std::runtime_error("ocd2: truncated file"); }
  uint32_t u32() { need(4); uint32_t v; memcpy(&v, b.data() + p, 4); p += 4; return v; }
  uint64_t u64() { need(8); uint64_t v; memcpy(&v, b.data() + p, 8); p += 8; return v; }
  uint16_t u16() { need(2); uint16_t v; memcpy(&v, b.data() + p, 2); p += 2; return v; }
  // marisa Vector<T>: u64 byte count, the objects, padding to 8 bytes
  std::string vec(size_t item) {
    const uint64_t n = u64();
    if (n % item) throw std::runtime_error("ocd2: bad vector size");
    need(n);
    std::string d = b.substr(p, n);
    p += n + (8 - n % 8) % 8;
    return d;
  }
};

struct BitVector {
  std::vector<uint64_t> units;
  uint32_t size = 0, num1 = 0;
  std::vector<uint32_t> ones, zeros;  // positions (enumeration only: no rank/select indices needed)
  void read(Reader& r) {
    const std::string u = r.vec(8);
    units.resize(u.size() / 8);
    if (!u.empty()) memcpy(units.data(), u.data(), u.size());  // (an empty vector's data() may be null: not a memcpy argument)
    size = r.u32();
    num1 = r.u32();
    r.vec(12);  // rank index
    r.vec(4);   // select0 index
    r.vec(4);   // select1 index
    if (((size_t)size + 63) / 64 > units.size()) throw std::runtime_error("ocd2: bad bit vector");  // in size_t: size + 63 wraps in 32 bits
    for (uint32_t i = 0; i < size; ++i) (get(i) ? ones : zeros).push_back(i);
  }
  bool get(size_t i) const {
    if (i >= size) throw std::runtime_error("ocd2: bit index out of range");  // node ids come from the file's own links
    return (units[i >> 6] >> (i & 63)) & 1;
  }
  size_t rank1(size_t i) const { return std::lower_bound(ones.begin(), ones.end(), (uint32_t)i) - ones.begin(); }
  size_t select1(size_t k) const { if (k >= ones.size()) throw std::runtime_error("ocd2: select out of range"); return ones[k]; }
};

struct FlagVector {
  std::vector<uint64_t> units;
  uint32_t value_size = 0, mask = 0;
  void read(Reader& r) {
    const std::string u = r.vec(8);
    units.resize(u.size() / 8 + 1, 0);
    memcpy(units.data(), u.data(), u.size());
    value_size = r.u32();
    mask = r.u32();
    r.u64();  // number of values
    if (value_size > 32) throw std::runtime_error("ocd2: bad flag vector");
  }
  uint32_t get(size_t i) const {
    const size_t pos = i * value_size, w = pos >> 6, o = pos & 63;
    if (w + 1 >= units.size()) throw std::runtime_error("ocd2: flag index out of range");  // (one spare unit is allocated)
    uint64_t v = units[w] >> o;
    if (o + value_size > 64) v |= units[w + 1] << (64 - o);
    return (uint32_t)v & mask;
  }
};

struct Trie {  // one level of marisa's LoudsTrie
  BitVector louds, terminal, link;
  std::string bases;
  FlagVector extras;
  std::string tail_buf;
  BitVector tail_end;
  std::unique_ptr<Trie> next;
  uint32_t num_l1 = 0;

  size_t get_link(size_t node) const { return (uint8_t)bases[node] | ((size_t)extras.get(link.rank1(node)) << 8); }
  void restore_link(size_t lnk, std::string& out) const {
    if (next) { next->restore(lnk, out); return; }
    if (lnk >= tail_buf.size()) throw std::runtime_error("ocd2: tail offset out of range");
    if (tail_end.size == 0) {
      for (size_t o = lnk; o < tail_buf.size() && tail_buf[o]; ++o) out.push_back(tail_buf[o]);
    } else {
      for (size_t o = lnk; o < tail_buf.size() && o < tail_end.size; ++o) { out.push_back(tail_buf[o]); if (tail_end.get(o)) break; }
    }
  }
  // a node of a NEXT-level trie spells its string walking up to the root
  void restore(size_t node, std::string& out) const {
    for (int guard = 0; guard < 4096; ++guard) {
      if (node >= bases.size()) throw std::runtime_error("ocd2: node out of range");
      if (link.get(node)) restore_link(get_link(node), out);
      else out.push_back(bases[node]);
      if (node <= num_l1) return;
      node = louds.select1(node) - node - 1;
    }
    throw std::runtime_error("ocd2: trie walk does not terminate");
  }
  // key of the top-level trie by key id (marisa reverse lookup)
  std::string key(size_t id) const {
    size_t node = terminal.select1(id);
    std::string out;
    if (node == 0) return out;
    for (int guard = 0; guard < 4096; ++guard) {
      if (node >= bases.size()) throw std::runtime_error("ocd2: node out of range");
      if (link.get(node)) {
        std::string t;
        restore_link(get_link(node), t);
        out.append(t.rbegin(), t.rend());
      } else {
        out.push_back(bases[node]);
      }
      if (node <= num_l1) return std::string(out.rbegin(), out.rend());
      node = louds.select1(node) - node - 1;
    }
    throw std::runtime_error("ocd2: trie walk does not terminate");
  }
};

}  // namespace ocd2

// One OpenCC dictionary: key -> first value, plus the longest key (bytes).
struct T2SDict {
  std::unordered_map<std::string, std::string> map;
  size_t max_key = 0;

  static T2SDict load_ocd2(const std::string& path) { return load_ocd2_bytes(read_text_file(path), path); }
  // the dictionary image itself (tests hand it mutated bytes); `path` only names it in messages
  static T2SDict load_ocd2_bytes(const std::string& b, const std::string& path) {
    static const char kMagic[] = "OPENCC_MARISA_0.2.5";
    static const char kMarisa[16] = {'W', 'e', ' ', 'l', 'o', 'v', 'e', ' ', 'M', 'a', 'r', 'i', 's', 'a', '.', '\0'};
    const size_t ml = sizeof(kMagic) - 1;
    if (b.size() < ml + 16 || memcmp(b.data(), kMagic, ml) != 0 || memcmp(b.data() + ml, kMarisa, 16) != 0)
      throw std::runtime_error("ocd2: '" + path + "' is not an OPENCC_MARISA_0.2.5 dictionary");
    ocd2::Reader r{b, ml + 16};
    // the levels are written depth first: fields of level 1, fields of level 2, ..., then (innermost first) each level's
    // cache, root child count and config word. A level has a successor iff it has links but no tail of its own.
    std::vector<ocd2::Trie*> levels;
    std::unique_ptr<ocd2::Trie> top(new ocd2::Trie());
    for (ocd2::Trie* t = top.get();;) {
      t->louds.read(r); t->terminal.read(r); t->link.read(r);
      t->bases = r.vec(1);
      t->extras.read(r);
      t->tail_buf = r.vec(1);
      t->tail_end.read(r);
      levels.push_back(t);
      if (t->link.num1 == 0 || !t->tail_buf.empty() || levels.size() >= 16) break;
      t->next.reset(new ocd2::Trie());
      t = t->next.get();
    }
    for (size_t i = levels.size(); i-- > 0;) {
      r.vec(12);  // cache
      levels[i]->num_l1 = r.u32();
      r.u32();    // config flags
    }
    const size_t n_keys = top->terminal.num1;
    // serialized values
    const uint32_t n_items = r.u32(), total = r.u32();
    if (n_items != n_keys) throw std::runtime_error("ocd2: key / value count mismatch");
    r.need(total);
    const size_t vbuf = r.p;
    r.p += total;
    T2SDict d;
    size_t off = 0;
    for (uint32_t i = 0; i < n_items; ++i) {
      const uint16_t nv = r.u16();
      std::string first;
      for (uint16_t v = 0; v < nv; ++v) {
        const uint16_t len = r.u16();  // bytes incl. the terminating NUL
        if (len == 0 || off + len > total) throw std::runtime_error("ocd2: bad value length");
        if (v == 0) first.assign(b.data() + vbuf + off, len - 1);
        off += len;
      }
      if (nv == 0) continue;
      const std::string key = top->key(i);
      if (key.empty()) continue;
      d.max_key = std::max(d.max_key, key.size());
      d.map.emplace(key, first);
    }
    return d;
  }

  // longest key that prefixes s[pos..]; nullptr if none. *len = its byte length.
  const std::string* match_prefix(const std::string& s, size_t pos, size_t* len) const {
    const size_t lim = std::min(max_key, s.size() - pos);
    for (size_t l = lim; l > 0; --l) {
      if (pos + l < s.size() && ((unsigned char)s[pos + l] & 0xC0) == 0x80) continue;  // not a character boundary
      auto it = map.find(s.substr(pos, l));
      if (it != map.end()) { *len = l; return &it->second; }
    }
    return nullptr;
  }
};

inline size_t utf8_char_len(const std::string& s, size_t pos) {
  const unsigned char c = (unsigned char)s[pos];
  size_t n = c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xE ? 3 : (c >> 3) == 0x1E ? 4 : 1;
  return std::min(n, s.size() - pos);
}

// The converter an OpenCC JSON configuration describes (types used by t2s.json: mmseg segmentation over an ocd2
// dictionary; a chain of conversions whose dictionary is an ocd2 file or a group of them).
class T2SConverter {
 public:
  explicit T2SConverter(const std::string& config_path) {
    const size_t slash = config_path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? std::string() : config_path.substr(0, slash + 1);
    JsonValue j = JsonParser(read_text_file(config_path)).parse();
    auto load = [&](const JsonValue& d) -> std::shared_ptr<T2SDict> {
      if (d.at("type").as_str() != "ocd2") throw std::runtime_error("opencc config: unsupported dictionary type '" + d.at("type").as_str() + "'");
      const std::string f = dir + d.at("file").as_str();
      auto it = cache_.find(f);
      if (it != cache_.end()) return it->second;
      auto p = std::make_shared<T2SDict>(T2SDict::load_ocd2(f));
      cache_[f] = p;
      return p;
    };
    const JsonValue& seg = j.at("segmentation");
    if (seg.at("type").as_str() != "mmseg") throw std::runtime_error("opencc config: unsupported segmentation");
    seg_ = load(seg.at("dict"));
    for (const JsonValue& c : j.at("conversion_chain").arr) {
      const JsonValue& d = c.at("dict");
      std::vector<std::shared_ptr<T2SDict>> group;
      if (d.at("type").as_str() == "group") for (const JsonValue& g : d.at("dicts").arr) group.push_back(load(g));
      else group.push_back(load(d));
      chain_.push_back(group);
    }
  }

  std::string convert(const std::string& text) const {
    // mmseg: phrase segments and runs of unmatched characters
    std::vector<std::string> segs;
    std::string run;
    for (size_t p = 0; p < text.size();) {
      size_t len = 0;
      if (seg_->match_prefix(text, p, &len)) {
        if (!run.empty()) { segs.push_back(run); run.clear(); }
        segs.push_back(text.substr(p, len));
      } else {
        len = utf8_char_len(text, p);
        run.append(text, p, len);
      }
      p += len;
    }
    if (!run.empty()) segs.push_back(run);
    for (const auto& group : chain_) {
      for (std::string& s : segs) {
        std::string out;
        for (size_t p = 0; p < s.size();) {
          size_t len = 0;
          const std::string* v = nullptr;
          for (const auto& d : group)
            if ((v = d->match_prefix(s, p, &len))) break;  // DictGroup: the first dictionary that matches at all
          if (v) out += *v;
          else { len = utf8_char_len(s, p); out.append(s, p, len); }
          p += len;
        }
        s.swap(out);
      }
    }
    std::string res;
    for (const std::string& s : segs) res += s;
    return res;
  }

 private:
  std::map<std::string, std::shared_ptr<T2SDict>> cache_;
  std::shared_ptr<T2SDict> seg_;
  std::vector<std::vector<std::shared_ptr<T2SDict>>> chain_;
};

}  // namespace axw
