// flvis_amd: reader of DBoW3 vocabulary files -- the host side of `Vocabulary vocTmp(vocFile)` in the loop-closing nodelet
// (src/backend/vo_loopclosing.cpp:1095-1099, the launch files point it at voc/voc_orb.dbow3).  The tree lands in the flat
// arrays flvis_hip_bow_set_vocabulary takes; flvis_hip_bow_load_vocabulary is the two steps in one call.
//
// The three on-disk layouts Vocabulary::load accepts (3rdPartLib/DBow3/src/Vocabulary.cpp:1082-1096) are read:
//   * binary   (Vocabulary::toStream / fromStream, :1180-1256 / :1335-1407): u64 magic 88877711233, bool compressed, u32 node
//     count; then -- plain, or as QuickLZ 1.5 level-1 blocks of 10000 bytes each -- k, L, scoring, weighting, one record per
//     non-root node (id, parent, f64 weight, cv::Mat header cols/rows/type + the descriptor bytes), the word table (word id,
//     node id).  `save(filename)` compresses by default, so that is the layout a .dbow3 file normally has.
//   * text     (load_fromtxt, :1259-1332; the ORB-SLAM2 ORBvoc.txt layout): "k L scoring weighting", then one line per node
//     "parent is_leaf d0 .. d31 weight"; node ids and word ids count up in file order.
//   * OpenCV FileStorage YAML, optionally gzipped (load(fs), :1411-1462): vocabulary: {k, L, scoringType, weightingType,
//     nodes: [{nodeId, parentId, weight, descriptor: "dbw3 <type> <cols> b0 b1 .."}], words: [{wordId, nodeId}]}.
// In all three the children of a node keep the order in which the file lists them: the descent of transform() takes the FIRST
// child at minimum Hamming distance, so the order is part of the result.
//
// Only what the device path implements is accepted: 32-byte CV_8U descriptors (ORB), weighting TF_IDF or TF (both add the leaf's
// stored weight per feature), scoring L1_NORM.  Anything else is refused with a message, never approximated.
#include <zlib.h>

#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/flvis_hip.h"
#include "ctx.hpp"

struct flvis_voc_file {
  int k = 0, L = 0, scoring = 0, weighting = 0, format = 0;
  int n_nodes = 0, n_words = 0;
  std::vector<int> child_ptr, child_idx, word_id;
  std::vector<uint8_t> desc;
  std::vector<double> weight;
};

namespace {

struct VocError {
  std::string msg;
};
[[noreturn]] void bad(const std::string& m) { throw VocError{m}; }

// ---- QuickLZ 1.5, compression level 1, no streaming buffer (the settings of 3rdPartLib/DBow3/src/quicklz.h:25,31) -------------
// block := header, then items steered by 32-bit control words (LSB first, a set bit 31 is the end marker, so a word steers 31
// items).  Control bit 1: a match -- 16 bits {len-2 : 4, hash : 12}, or, when the low nibble is 0, 24 bits with the length in
// the third byte; the source is the position the 4096-entry table holds for `hash`.  The table is not transmitted: both sides
// enter position p under hash(bytes p..p+2) = ((v >> 12) ^ v) & 4095 as soon as p+2 is known, up to the start of the last match
// / three bytes behind the literal front.  Control bit 0: literals.  The last 10 bytes of a block are always literals, one per
// control bit.  header := flags (bit 0 compressed, bit 1 four-byte sizes), compressed size, decompressed size.
struct Qlz1 {
  int32_t table[4096];
  static uint32_t rd(const uint8_t* p, const uint8_t* end, int n) {  // little-endian read, zero beyond the block
    uint32_t v = 0;
    for (int i = 0; i < n; i++)
      if (p + i < end) v |= (uint32_t)p[i] << (8 * i);
    return v;
  }
  static size_t header_size(uint8_t flags) { return (flags & 2) ? 9 : 3; }

  // decodes one block starting at src (at most `avail` bytes readable); appends to out; returns the compressed size consumed
  size_t block(const uint8_t* src, size_t avail, std::vector<uint8_t>& out) {
    if (avail < 3) bad("compressed vocabulary: truncated block header");
    const uint8_t flags = src[0];
    const size_t hs = header_size(flags);
    if (avail < hs) bad("compressed vocabulary: truncated block header");
    const int nb = (flags & 2) ? 4 : 1;
    const size_t csize = rd(src + 1, src + avail, nb), dsize = rd(src + 1 + nb, src + avail, nb);
    if (csize < hs || csize > avail) bad("compressed vocabulary: block size beyond the end of the file");
    if (((flags >> 2) & 3) != 1 && (flags & 1)) bad("compressed vocabulary: QuickLZ level other than 1");
    const uint8_t* end = src + csize;
    const size_t base = out.size();
    if (!(flags & 1)) {  // stored
      if (csize - hs != dsize) bad("compressed vocabulary: stored block with inconsistent sizes");
      out.insert(out.end(), src + hs, end);
      return csize;
    }
    // (a 3-byte token yields at most 255 bytes: a size no block of this length can decode to is refused before it is allocated)
    if (dsize > 90 * csize + 64) bad("compressed vocabulary: implausible decompressed size");
    out.resize(base + dsize);
    uint8_t* dst0 = out.data() + base;
    for (int i = 0; i < 4096; i++) table[i] = -1;
    const uint8_t* s = src + hs;
    int64_t d = 0, hashed = -1;  // d: bytes written; hashed: last position entered in the table
    const int64_t n = (int64_t)dsize, last_matchstart = n - 1 - 6 - 4;
    auto hash_upto = [&](int64_t upto) {
      while (hashed < upto) {
        hashed++;
        const uint32_t v = dst0[hashed] | ((uint32_t)dst0[hashed + 1] << 8) | ((uint32_t)dst0[hashed + 2] << 16);
        table[((v >> 12) ^ v) & 4095] = (int32_t)hashed;
      }
    };
    uint32_t cword = 1;
    for (;;) {
      if (cword == 1) {
        if (s + 4 > end) bad("compressed vocabulary: control word beyond the block");
        cword = rd(s, end, 4);
        s += 4;
      }
      if (cword & 1) {
        cword >>= 1;
        const uint32_t f = rd(s, end, 3);
        const int32_t from = table[(f >> 4) & 0xfff];
        int64_t len;
        if (f & 0xf) {
          len = (f & 0xf) + 2;
          s += 2;
        } else {
          len = (f >> 16) & 0xff;
          s += 3;
        }
        if (s > end) bad("compressed vocabulary: match token beyond the block");
        if (from < 0 || from > d - 3) bad("compressed vocabulary: match refers to data not yet decoded");
        if (len < 3 || d + len > n - 4) bad("compressed vocabulary: match runs past the block");
        for (int64_t i = 0; i < len; i++) dst0[d + i] = dst0[from + i];  // forward byte copy: an overlap repeats the pattern
        d += len;
        hash_upto(d - len);
        hashed = d - 1;
      } else if (d < last_matchstart) {
        static const int run[16] = {4, 0, 1, 0, 2, 0, 1, 0, 3, 0, 1, 0, 2, 0, 1, 0};  // literal flags in a row, at most 4
        const int r = run[cword & 0xf];
        if (s + r > end) bad("compressed vocabulary: literals beyond the block");
        for (int i = 0; i < r; i++) dst0[d + i] = s[i];
        cword >>= r;
        d += r;
        s += r;
        hash_upto(d - 3);
      } else {
        while (d < n) {
          if (cword == 1) {
            s += 4;
            cword = 1u << 31;
          }
          if (s >= end) bad("compressed vocabulary: literals beyond the block");
          dst0[d++] = *s++;
          cword >>= 1;
        }
        return csize;
      }
    }
  }
};

// ---- the tree as the files describe it: one record per non-root node, in file order -----------------------------------------
struct Builder {
  struct Rec {
    uint32_t id, parent;
    double weight;
    uint8_t desc[32];
  };
  std::vector<Rec> recs;
  std::vector<std::pair<uint32_t, uint32_t>> words;  // (word id, node id)
  int k = 0, L = 0, scoring = 0, weighting = 0;

  void finish(flvis_voc_file& v, size_t n_nodes) {
    if (n_nodes < 2 || n_nodes > (size_t)INT_MAX / 64) bad("vocabulary: node count out of range");
    if (recs.size() != n_nodes - 1) bad("vocabulary: the file does not hold one record per non-root node");
    if (scoring != 0) bad("vocabulary: scoring type " + std::to_string(scoring) + " (only L1_NORM = 0 is implemented on the device)");
    if (weighting != 0 && weighting != 1)
      bad("vocabulary: weighting type " + std::to_string(weighting) + " (only TF_IDF = 0 and TF = 1 are implemented on the device)");
    v.k = k, v.L = L, v.scoring = scoring, v.weighting = weighting;
    v.n_nodes = (int)n_nodes;
    v.child_ptr.assign(n_nodes + 1, 0);
    v.word_id.assign(n_nodes, -1);
    v.weight.assign(n_nodes, 0.0);
    v.desc.assign(n_nodes * 32, 0);
    std::vector<char> seen(n_nodes, 0);
    seen[0] = 1;
    for (const Rec& r : recs) {
      if (r.id == 0 || r.id >= n_nodes || r.parent >= n_nodes || r.parent == r.id) bad("vocabulary: node or parent id out of range");
      if (seen[r.id]) bad("vocabulary: node " + std::to_string(r.id) + " listed twice");
      seen[r.id] = 1;
      v.child_ptr[r.parent + 1]++;
      v.weight[r.id] = r.weight;
      memcpy(&v.desc[(size_t)r.id * 32], r.desc, 32);
    }
    for (size_t n = 0; n < n_nodes; n++) v.child_ptr[n + 1] += v.child_ptr[n];
    v.child_idx.assign(recs.size(), 0);
    std::vector<int> fill(v.child_ptr.begin(), v.child_ptr.end() - 1);
    for (const Rec& r : recs) v.child_idx[fill[r.parent]++] = (int)r.id;  // file order within a parent
    if (v.child_ptr[1] == 0) bad("vocabulary: the root has no children");
    // every node must hang below the root (a cycle among non-root nodes would leave the descent nothing to reach)
    {
      std::vector<int> stack{0};
      size_t reached = 0;
      std::vector<char> mark(n_nodes, 0);
      while (!stack.empty()) {
        const int n = stack.back();
        stack.pop_back();
        if (mark[n]) bad("vocabulary: the node links do not form a tree");
        mark[n] = 1;
        reached++;
        for (int c = v.child_ptr[n]; c < v.child_ptr[n + 1]; c++) stack.push_back(v.child_idx[c]);
      }
      if (reached != n_nodes) bad("vocabulary: nodes that are not reachable from the root");
    }
    int n_words = 0;
    for (auto& w : words) {
      if (w.second >= n_nodes || w.first >= (uint32_t)INT_MAX - 1) bad("vocabulary: word table entry out of range");
      v.word_id[w.second] = (int)w.first;
      if ((int)w.first + 1 > n_words) n_words = (int)w.first + 1;
    }
    for (size_t n = 1; n < n_nodes; n++)
      if (v.child_ptr[n + 1] == v.child_ptr[n] && v.word_id[n] < 0) bad("vocabulary: leaf node " + std::to_string(n) + " has no word id");
    v.n_words = n_words;
  }
};

struct Cursor {  // bounds-checked little-endian reads of the binary layout
  const uint8_t* p;
  const uint8_t* end;
  template <class T>
  T get() {
    if ((size_t)(end - p) < sizeof(T)) bad("vocabulary: unexpected end of the binary stream");
    T v;
    memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  void bytes(void* dst, size_t n) {
    if ((size_t)(end - p) < n) bad("vocabulary: unexpected end of the binary stream");
    memcpy(dst, p, n);
    p += n;
  }
};

const uint64_t kMagic = 88877711233ull;

void parse_binary(const std::vector<uint8_t>& file, flvis_voc_file& v) {
  Cursor c{file.data(), file.data() + file.size()};
  c.get<uint64_t>();
  const bool compressed = c.get<uint8_t>() != 0;
  const uint32_t nnodes = c.get<uint32_t>();
  if (nnodes == 0) bad("vocabulary: the file holds an empty vocabulary");
  std::vector<uint8_t> plain;
  if (compressed) {
    const uint32_t chunks = c.get<uint32_t>();
    std::unique_ptr<Qlz1> q(new Qlz1());
    for (uint32_t i = 0; i < chunks; i++) c.p += q->block(c.p, (size_t)(c.end - c.p), plain);
    c = Cursor{plain.data(), plain.data() + plain.size()};
  }
  v.format = compressed ? 1 : 0;
  Builder b;
  b.k = c.get<int32_t>();
  b.L = c.get<int32_t>();
  b.scoring = c.get<int32_t>();
  b.weighting = c.get<int32_t>();
  // a record is 60 bytes (id, parent, weight, cv::Mat header, 32 descriptor bytes): a node count the stream cannot hold is refused
  // before anything of that size is allocated
  if ((uint64_t)(nnodes - 1) * 60 > (uint64_t)(c.end - c.p)) bad("vocabulary: the node count exceeds what the file holds");
  b.recs.resize(nnodes - 1);
  for (auto& r : b.recs) {
    r.id = c.get<uint32_t>();
    r.parent = c.get<uint32_t>();
    r.weight = c.get<double>();
    const int32_t cols = c.get<int32_t>(), rows = c.get<int32_t>(), type = c.get<int32_t>();
    if (type != 0 || rows != 1 || cols != 32) bad("vocabulary: descriptors are not 1x32 CV_8U (only ORB vocabularies are supported)");
    c.bytes(r.desc, 32);
  }
  const uint32_t nwords = c.get<uint32_t>();
  if (nwords > nnodes) bad("vocabulary: more words than nodes");
  b.words.resize(nwords);
  for (auto& w : b.words) {
    w.first = c.get<uint32_t>();
    w.second = c.get<uint32_t>();
  }
  b.finish(v, nnodes);
}

// numbers of a text line; strtod accepts what operator>> does for the layouts at hand
bool next_number(const char*& p, const char* end, double& out) {
  while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
  if (p >= end) return false;
  char* q = nullptr;
  out = strtod(p, &q);
  if (q == p) return false;
  p = q;
  return true;
}

void parse_txt(const std::vector<uint8_t>& file, flvis_voc_file& v) {
  const char* p = (const char*)file.data();
  const char* end = p + file.size();
  auto line_end = [&](const char* s) {
    while (s < end && *s != '\n') s++;
    return s;
  };
  Builder b;
  {
    const char* le = line_end(p);
    double h[4];
    for (int i = 0; i < 4; i++)
      if (!next_number(p, le, h[i])) bad("vocabulary text file: the first line must read 'k L scoring weighting'");
    for (int i = 0; i < 4; i++)
      if (!(h[i] >= -1e6 && h[i] <= 1e6)) bad("vocabulary text file: this is not a vocabulary header");  // (also refuses NaN: the casts below are defined)
    b.k = (int)h[0], b.L = (int)h[1], b.scoring = (int)h[2], b.weighting = (int)h[3];
    if (b.k < 0 || b.k > 20 || b.L < 1 || b.L > 10 || b.scoring < 0 || b.scoring > 5 || b.weighting < 0 || b.weighting > 3)
      bad("vocabulary text file: this is not a vocabulary header");  // Vocabulary.cpp:1271
    p = le < end ? le + 1 : end;
  }
  uint32_t next_word = 0;
  while (p < end) {
    const char* le = line_end(p);
    std::vector<double> num;
    double x;
    const char* q = p;
    while (next_number(q, le, x)) num.push_back(x);
    p = le < end ? le + 1 : end;
    if (num.empty()) break;  // an empty line ends the node list (Vocabulary.cpp:1294)
    if (num.size() != 2 + 32 + 1) bad("vocabulary text file: a node line must hold parent, leaf flag, 32 descriptor bytes and the weight");
    Builder::Rec r;
    r.id = (uint32_t)b.recs.size() + 1;
    if (num[0] < 0 || num[0] >= (double)r.id) bad("vocabulary text file: a node's parent must come before it");
    r.parent = (uint32_t)num[0];
    for (int i = 0; i < 32; i++) {
      if (!(num[2 + i] >= 0.0 && num[2 + i] < 256.0)) bad("vocabulary text file: a descriptor byte outside 0 .. 255");
      r.desc[i] = (uint8_t)(float)num[2 + i];
    }
    if (!std::isfinite(num[34])) bad("vocabulary text file: a node weight that is not a number");
    r.weight = (double)(float)num[34];  // the reference reads every field of the line as float
    b.recs.push_back(r);
    if (num[1] > 0) b.words.push_back({next_word++, r.id});
  }
  v.format = 2;
  b.finish(v, b.recs.size() + 1);
}

// ---- OpenCV FileStorage YAML: just the shapes Vocabulary::save(fs) writes ------------------------------------------------------
// a decimal node / word id of the FileStorage text: all digits, below 2^32 (strtoul alone would wrap or saturate silently)
uint32_t parse_id(const std::string& val, const char* what) {
  errno = 0;
  char* endp = nullptr;
  const unsigned long long v = strtoull(val.c_str(), &endp, 10);
  if (val.empty() || endp == val.c_str() || errno != 0 || v > 0xffffffffull) bad(std::string("vocabulary yaml: bad ") + what);
  return (uint32_t)v;
}

struct Yaml {
  const char* p;
  const char* end;
  void skip_ws() {
    while (p < end) {
      if (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')
        p++;
      else if (*p == '#')
        while (p < end && *p != '\n') p++;
      else
        break;
    }
  }
  // position behind "key:" (searched from the current position); false when absent
  bool seek_key(const char* key) {
    const size_t n = strlen(key);
    for (const char* s = p; s + n + 1 <= end; s++)
      if (!memcmp(s, key, n) && s[n] == ':' && (s == p || s[-1] == ' ' || s[-1] == '\n' || s[-1] == '{' || s[-1] == ',')) {
        p = s + n + 1;
        return true;
      }
    return false;
  }
  double scalar() {
    skip_ws();
    double x;
    if (!next_number(p, end, x)) bad("vocabulary yaml: number expected");
    return x;
  }
  // one flow mapping "{ key:value, key:"string", ... }" -> callbacks; returns false at the closing ']' of the sequence
  template <class F>
  bool flow_map(F&& on_pair) {
    skip_ws();
    while (p < end && (*p == '-' || *p == ',')) {
      p++;
      skip_ws();
    }
    if (p >= end || *p != '{') return false;  // ']' of a flow sequence, the next key of a block sequence, or the end of the file
    p++;
    for (;;) {
      skip_ws();
      if (p >= end) bad("vocabulary yaml: unterminated mapping");
      if (*p == '}') {
        p++;
        return true;
      }
      if (*p == ',' || *p == ':') {  // "{:" is FileStorage's flow-map opener in older files
        p++;
        continue;
      }
      const char* k0 = p;
      while (p < end && *p != ':' && *p != '}' && *p != ',') p++;
      if (p >= end || *p != ':') bad("vocabulary yaml: 'key:value' expected");
      std::string key(k0, p);
      while (!key.empty() && (key.back() == ' ' || key.back() == '"')) key.pop_back();
      while (!key.empty() && (key[0] == ' ' || key[0] == '"')) key.erase(0, 1);
      p++;
      skip_ws();
      std::string val;
      if (p < end && (*p == '"' || *p == '\'')) {
        const char qc = *p++;
        while (p < end && *p != qc) {
          if (*p == '\\' && p + 1 < end) p++;
          val.push_back(*p == '\n' ? ' ' : *p);
          p++;
        }
        if (p >= end) bad("vocabulary yaml: unterminated string");
        p++;
      } else {
        const char* v0 = p;
        while (p < end && *p != ',' && *p != '}') p++;
        val.assign(v0, p);
      }
      on_pair(key, val);
    }
  }
};

void parse_descriptor_string(const std::string& s, uint8_t* out) {  // DescManip::fromString (DescManip.cpp:164-199)
  const char* p = s.c_str();
  const char* end = p + s.size();
  std::vector<double> num;
  double x;
  const bool tagged = s.substr(0, 10).find("dbw3") != std::string::npos;
  if (tagged) p = strstr(p, "dbw3") + 4;
  while (next_number(p, end, x)) num.push_back(x);
  size_t off = 0;
  if (tagged) {
    if (num.size() < 2 || (int)num[0] != 0 || (int)num[1] != 32) bad("vocabulary yaml: descriptors are not 32-byte CV_8U (only ORB vocabularies are supported)");
    off = 2;
  }
  if (num.size() - off != 32) bad("vocabulary yaml: a descriptor string must hold 32 bytes");
  for (int i = 0; i < 32; i++) {
    if (!(num[off + i] >= 0.0 && num[off + i] < 256.0)) bad("vocabulary yaml: a descriptor byte outside 0 .. 255");
    out[i] = (uint8_t)(int)num[off + i];
  }
}

void parse_yaml(const std::vector<uint8_t>& file, flvis_voc_file& v) {
  Yaml y{(const char*)file.data(), (const char*)file.data() + file.size()};
  if (!y.seek_key("vocabulary")) bad("vocabulary file: neither a DBoW3 binary, a .txt vocabulary nor a FileStorage yaml with a 'vocabulary' node");
  Builder b;
  auto need = [&](const char* key) {
    if (!y.seek_key(key)) bad(std::string("vocabulary yaml: '") + key + "' missing");
  };
  need("k");
  b.k = (int)y.scalar();
  need("L");
  b.L = (int)y.scalar();
  need("scoringType");
  b.scoring = (int)y.scalar();
  need("weightingType");
  b.weighting = (int)y.scalar();
  need("nodes");
  y.skip_ws();
  if (y.p >= y.end || *y.p != '[') {
    // block sequence ("- { .. }" lines): flow_map skips the dashes; the list ends at the 'words' key
  } else {
    y.p++;
  }
  uint32_t max_id = 0;
  for (;;) {
    y.skip_ws();
    if (y.p + 6 <= y.end && !memcmp(y.p, "words:", 6)) break;
    Builder::Rec r{};
    bool have[4] = {false, false, false, false};
    const bool more = y.flow_map([&](const std::string& k, const std::string& val) {
      if (k == "nodeId") r.id = parse_id(val, "nodeId"), have[0] = true;
      else if (k == "parentId") r.parent = parse_id(val, "parentId"), have[1] = true;
      else if (k == "weight") r.weight = strtod(val.c_str(), nullptr), have[2] = true;
      else if (k == "descriptor") parse_descriptor_string(val, r.desc), have[3] = true;
    });
    if (!more) {
      if (y.p < y.end && *y.p == ']') y.p++;
      break;
    }
    if (!(have[0] && have[1] && have[2] && have[3])) bad("vocabulary yaml: a node needs nodeId, parentId, weight and descriptor");
    if (r.id > max_id) max_id = r.id;
    b.recs.push_back(r);
  }
  need("words");
  y.skip_ws();
  if (y.p < y.end && *y.p == '[') y.p++;
  for (;;) {
    uint32_t wid = 0, nid = 0;
    bool have[2] = {false, false};
    const bool more = y.flow_map([&](const std::string& k, const std::string& val) {
      if (k == "wordId") wid = parse_id(val, "wordId"), have[0] = true;
      else if (k == "nodeId") nid = parse_id(val, "nodeId"), have[1] = true;
    });
    if (!more) break;
    if (!(have[0] && have[1])) bad("vocabulary yaml: a word needs wordId and nodeId");
    b.words.push_back({wid, nid});
  }
  v.format = 3;
  b.finish(v, b.recs.size() + 1);  // m_nodes.resize(fn.size() + 1), Vocabulary.cpp:1429
  (void)max_id;
}

std::vector<uint8_t> read_file(const char* path) {  // through zlib: FileStorage opens .gz transparently, plain files pass through
  gzFile f = gzopen(path, "rb");
  if (!f) bad(std::string("cannot open ") + path);
  std::vector<uint8_t> data;
  std::vector<uint8_t> buf(1 << 20);
  for (;;) {
    const int n = gzread(f, buf.data(), (unsigned)buf.size());
    if (n < 0) {
      gzclose(f);
      bad(std::string("read error on ") + path);
    }
    if (n == 0) break;
    data.insert(data.end(), buf.begin(), buf.begin() + n);
  }
  gzclose(f);
  return data;
}

}  // namespace

extern "C" {

int flvis_voc_file_open(const char* path, flvis_voc_file** out, char* err, int errlen) {
  auto fail = [&](const std::string& m) {
    if (err && errlen > 0) snprintf(err, errlen, "%s", m.c_str());
    return (int)FLVIS_ERR_CONFIG;
  };
  if (!path || !out) return FLVIS_ERR_INVALID_ARG;
  *out = nullptr;
  flvis_voc_file* v = nullptr;
  try {
    const std::vector<uint8_t> file = read_file(path);
    v = new flvis_voc_file();
    uint64_t magic = 0;
    if (file.size() >= 8) memcpy(&magic, file.data(), 8);
    if (magic == kMagic)
      parse_binary(file, *v);
    else if (std::string(path).find(".txt") != std::string::npos)  // Vocabulary.cpp:1088
      parse_txt(file, *v);
    else
      parse_yaml(file, *v);
  } catch (const VocError& e) {
    delete v;
    return fail(e.msg);
  } catch (const std::exception& e) {
    delete v;
    return fail(std::string("vocabulary file: ") + e.what());
  }
  *out = v;
  return FLVIS_OK;
}

int flvis_voc_file_info(const flvis_voc_file* v, int* info8) {
  if (!v || !info8) return FLVIS_ERR_INVALID_ARG;
  const int vals[8] = {v->n_nodes, v->n_words, v->k, v->L, v->scoring, v->weighting, (int)v->child_idx.size(), v->format};
  memcpy(info8, vals, sizeof(vals));
  return FLVIS_OK;
}

int flvis_voc_file_arrays(const flvis_voc_file* v, const int** child_ptr, const int** child_idx, const uint8_t** desc, const double** weight,
                          const int** word_id) {
  if (!v || !child_ptr || !child_idx || !desc || !weight || !word_id) return FLVIS_ERR_INVALID_ARG;
  *child_ptr = v->child_ptr.data();
  *child_idx = v->child_idx.data();
  *desc = v->desc.data();
  *weight = v->weight.data();
  *word_id = v->word_id.data();
  return FLVIS_OK;
}

void flvis_voc_file_close(flvis_voc_file* v) { delete v; }

int flvis_hip_bow_load_vocabulary(flvis_ctx* ctx, const char* path) {
  if (!ctx) return FLVIS_ERR_INVALID_ARG;
  if (!path) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_load_vocabulary: no path");
  flvis_voc_file* v = nullptr;
  char msg[512] = {0};
  const int rc = flvis_voc_file_open(path, &v, msg, (int)sizeof(msg));
  if (rc != FLVIS_OK) return ctx->fail(rc, std::string("bow_load_vocabulary: ") + msg);
  // leaves carry their word id; inner nodes get 0 (never read)
  std::vector<int> wid(v->word_id);
  for (int& w : wid)
    if (w < 0) w = 0;
  const int r = flvis_hip_bow_set_vocabulary(ctx, v->n_nodes, v->child_ptr.data(), v->child_idx.data(), v->desc.data(), v->weight.data(),
                                             wid.data());
  flvis_voc_file_close(v);
  return r;
}
}
