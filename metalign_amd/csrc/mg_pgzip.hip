// mg_pgzip.hip — ONE gzip stream inflated by MANY host threads (host code only: no kernel in this file).
//
// The reference takes `.fq.gz` / `.fa.gz` as ordinary input (scripts/select_db.py:146-148 sniffs the type behind the
// `.gz`; kmc reads gzip itself, :50-52).  zlib inflates ~0.3 GB/s of text per core, so a plain gzip stream fed the GPU
// pipeline of mg_stream.hip at a thousandth of what it hashes.  A deflate stream CAN be entered in the middle — the way
// pugz / rapidgzip do it:
//
//   * the compressed file is cut into chunks of a few MB.  In each chunk a thread looks for the first bit position at which a
//     DYNAMIC-Huffman block with BFINAL = 0 begins — by trying every bit position and rejecting what cannot be one: the
//     code-length code must be a complete prefix code, the literal/length and distance code lengths must decode without
//     running over, the literal/length code must be complete and hold the end-of-block symbol, the distance code complete
//     (or a single code); then the block is decoded on trial: every literal must be TEXT (these files are FASTQ / FASTA /
//     SAM), every length / distance symbol legal, the block must end, and what follows must again look like a block;
//   * from there the thread decodes with an UNKNOWN 32 KB window: the output is 16-bit symbols, a byte, or 0x8000 + i = "byte
//     i of the 32 KB in front of this chunk's output" (a back-reference that reaches behind the chunk's start; copies of
//     such symbols stay symbols).  It stops at the block boundary where the next chunk's thread started;
//   * the chunks' ends are chained (a chunk whose start was a false positive is simply run over by its predecessor, which
//     never finds its boundary there), the window is handed from chunk to chunk — only the last 32 KB of a chunk have to be
//     resolved for that — and every chunk is resolved to bytes in parallel; CRC-32 and ISIZE of every member are checked
//     from per-piece CRCs (crc32_combine).
//
// Anything this cannot enter (binary data, stored / fixed blocks only) is still decoded correctly: the first chunk's thread
// runs through everything it meets; only the speed is then one thread's.  Truncated or corrupt streams are errors.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mg_internal.h"
#include "mg_pgzip.h"

namespace mg {

namespace {

// ---- bit reader over the mapped file (deflate packs bits LSB first) ----
struct Bits {
  const uint8_t* p;
  uint64_t nbits;   // size of the input in bits
  uint64_t pos;     // next bit
  bool over = false;
  // up to 57 bits at pos (zero beyond the end)
  inline uint64_t peek() const {
    const uint64_t byte = pos >> 3;
    uint64_t v = 0;
    const uint64_t nbytes = nbits >> 3;
    if (byte + 8 <= nbytes) memcpy(&v, p + byte, 8);
    else for (uint64_t i = 0; byte + i < nbytes && i < 8; ++i) v |= (uint64_t)p[byte + i] << (8 * i);
    return v >> (pos & 7);
  }
  inline uint32_t get(int n) {  // n <= 32
    const uint32_t v = (uint32_t)(peek() & ((1ull << n) - 1ull));
    pos += (uint64_t)n;
    if (pos > nbits) over = true;
    return v;
  }
};

// ---- canonical Huffman decoding tables: a primary table indexed by the next PB bits, overflow into sub-tables ----
struct Huff {
  // entry: bits 0..4 = code length (0 = invalid), bits 5..15 = symbol; or bit 31 set: bits 0..4 = extra index bits,
  // bits 5..25 = offset of the sub-table
  std::vector<uint32_t> t;
  int pb = 0;
  // lens[n] (0..15).  ok_incomplete_single: a code with exactly one symbol of length 1 is accepted (distance codes).
  // Returns false for over-subscribed or incomplete codes.  *nused = symbols with a non-zero length.
  bool build(const uint8_t* lens, int n, int primary, bool ok_incomplete_single, int* nused) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    *nused = n - count[0];
    if (*nused == 0) { t.assign(1u << primary, 0u); pb = primary; return true; }  // (no code at all: every look-up is invalid)
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
      left <<= 1;
      left -= count[l];
      if (left < 0) return false;  // over-subscribed
    }
    if (left > 0 && !(ok_incomplete_single && *nused == 1 && count[1] == 1)) return false;  // incomplete
    int maxlen = 15;
    while (maxlen > 1 && count[maxlen] == 0) --maxlen;
    pb = primary < maxlen ? primary : maxlen;
    uint32_t next[16];
    {
      uint32_t c = 0;
      next[0] = 0;
      for (int l = 1; l <= 15; ++l) { c = (c + (l > 1 ? (uint32_t)count[l - 1] : 0u)) << 1; next[l] = c; }  // (no codes of length 0)
    }
    t.assign(1u << pb, 0u);
    // sub-tables: for every primary prefix of a long code, as many extra bits as the longest code that shares it
    auto rev = [](uint32_t v, int bits) { uint32_t r = 0; for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; } return r; };
    struct Sym { uint32_t code; uint8_t len; uint16_t sym; };
    std::vector<Sym> longs;
    for (int s = 0; s < n; ++s) {
      const int l = lens[s];
      if (!l) continue;
      const uint32_t c = next[l]++;
      const uint32_t r = rev(c, l);  // as the bits arrive
      if (l <= pb) {
        for (uint32_t i = r; i < (1u << pb); i += 1u << l) t[i] = (uint32_t)l | ((uint32_t)s << 5);
      } else {
        longs.push_back(Sym{r, (uint8_t)l, (uint16_t)s});
      }
    }
    if (!longs.empty()) {
      std::vector<int> need(1u << pb, 0);
      for (const Sym& y : longs) { const uint32_t pre = y.code & ((1u << pb) - 1u); if (y.len - pb > need[pre]) need[pre] = y.len - pb; }
      for (uint32_t pre = 0; pre < (1u << pb); ++pre) {
        if (!need[pre]) continue;
        const uint32_t off = (uint32_t)t.size();
        t.resize(t.size() + (1u << need[pre]), 0u);
        t[pre] = 0x80000000u | (uint32_t)need[pre] | (off << 5);
      }
      for (const Sym& y : longs) {
        const uint32_t pre = y.code & ((1u << pb) - 1u);
        const int sb = (int)(t[pre] & 31u);
        const uint32_t off = (t[pre] >> 5) & 0x1fffffu;
        const uint32_t hi = y.code >> pb;
        for (uint32_t i = hi; i < (1u << sb); i += 1u << (y.len - pb)) t[off + i] = (uint32_t)y.len | ((uint32_t)y.sym << 5);
      }
    }
    return true;
  }
  // symbol at the reader's position, or -1; consumes its bits
  inline int decode(Bits& b) const {
    const uint64_t v = b.peek();
    uint32_t e = t[v & ((1u << pb) - 1u)];
    if (e & 0x80000000u) {
      const int sb = (int)(e & 31u);
      e = t[((e >> 5) & 0x1fffffu) + ((v >> pb) & ((1u << sb) - 1u))];
    }
    const int l = (int)(e & 31u);
    if (l == 0) return -1;
    b.pos += (uint64_t)l;
    if (b.pos > b.nbits) { b.over = true; return -1; }
    return (int)(e >> 5);
  }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline bool is_text(uint32_t c) { return (c >= 32 && c < 127) || c == '\n' || c == '\r' || c == '\t'; }

struct BlockCodes { Huff lit, dist; };

// The header of a dynamic block (the 3 block bits already consumed).  strict: the checks that reject a false block start.
bool read_dynamic_header(Bits& b, BlockCodes& bc, bool strict) {
  const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
  if (b.over || hlit > 286 || hdist > 30) return false;
  uint8_t cl[19] = {0};
  for (int i = 0; i < hclen; ++i) cl[kClOrder[i]] = (uint8_t)b.get(3);
  if (b.over) return false;
  Huff clh;
  int nused = 0;
  if (!clh.build(cl, 19, 7, false, &nused) || nused == 0) return false;
  uint8_t lens[286 + 30 + 138];
  int n = 0;
  const int total = hlit + hdist;
  while (n < total) {
    const int s = clh.decode(b);
    if (s < 0) return false;
    if (s < 16) { lens[n++] = (uint8_t)s; continue; }
    int rep, val = 0;
    if (s == 16) { if (n == 0) return false; val = lens[n - 1]; rep = 3 + (int)b.get(2); }
    else if (s == 17) rep = 3 + (int)b.get(3);
    else rep = 11 + (int)b.get(7);
    if (b.over || n + rep > total) return false;
    while (rep--) lens[n++] = (uint8_t)val;
  }
  if (lens[256] == 0) return false;  // no end-of-block symbol
  int nl = 0, nd = 0;
  if (!bc.lit.build(lens, hlit, 11, false, &nl)) return false;
  if (!bc.dist.build(lens + hlit, hdist, 9, true, &nd)) return false;
  if (strict && nl < 2) return false;  // (a block of one repeated symbol: legal, never seen in text, common among false starts)
  return true;
}

void fixed_codes(BlockCodes& bc) {
  uint8_t l[288];
  for (int i = 0; i < 144; ++i) l[i] = 8;
  for (int i = 144; i < 256; ++i) l[i] = 9;
  for (int i = 256; i < 280; ++i) l[i] = 7;
  for (int i = 280; i < 288; ++i) l[i] = 8;
  int n = 0;
  bc.lit.build(l, 288, 11, false, &n);
}

// the fixed distance code is incomplete by definition: build it without the completeness test
struct FixedDist { Huff h; FixedDist() { h.pb = 5; h.t.assign(32, 0u); for (uint32_t s = 0; s < 30; ++s) { uint32_t r = 0, v = s; for (int i = 0; i < 5; ++i) { r = (r << 1) | (v & 1u); v >>= 1; } h.t[r] = 5u | (s << 5); } } };

}  // namespace

// One chunk's speculative output.
struct PGzip::Chunk {
  // (raw buffers: std::vector would zero what the decoder is about to overwrite — and its 20 MB blocks came fresh from mmap, a
  // page fault per 4 KB, on every chunk until glibc's mmap threshold had adapted: the first 130 MB took four times what the
  // next took; recycled through the decoder's pool)
  template <class T>
  struct Raw {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    Raw() = default;
    Raw(const Raw&) = delete;
    Raw& operator=(const Raw&) = delete;
    Raw(Raw&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    Raw& operator=(Raw&& o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
    ~Raw() { free(p); }
    bool reserve(size_t want) {
      if (want <= cap) return true;
      size_t nc = cap ? cap : (size_t)(1u << 20);
      while (nc < want) nc += nc / 2;
      size_t bytes = (nc * sizeof(T) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
      void* q = aligned_alloc(2u << 20, bytes);
      if (!q) return false;
      // (MADV_HUGEPAGE was tried: where huge pages are not to be had — this container — every fault pays a failed compaction,
      // 0.9 s instead of 0.28 s for 130 MB on first use; opt-in)
      const bool thp = mg::dbg("pgzip_thp") != 0;
      if (thp) (void)madvise(q, bytes, MADV_HUGEPAGE);
      if (n) memcpy(q, p, n * sizeof(T));
      free(p);
      p = (T*)q;
      cap = bytes / sizeof(T);
      return true;
    }
    size_t size() const { return n; }
  };
  using SymBuf = Raw<uint16_t>;
  using ByteBuf = Raw<uint8_t>;
  uint64_t index = 0;
  uint64_t start_bit = 0;       // where decoding started (a block boundary, or the middle of the gzip header chain for chunk 0)
  bool have_start = false;
  bool at_member_start = false; // start_bit is the first byte of a gzip member header (the stream's start, or after a trailer)
  uint64_t end_bit = 0;         // block boundary (or end of stream) where decoding stopped
  bool eof = false;             // the stream ended in this chunk
  SymBuf sym;                   // bytes, or 0x8000 + index into the 32 KB window in front of the chunk
  struct MemberEnd { uint64_t out_pos; uint32_t crc, isize; };
  std::vector<MemberEnd> ends;  // members that END in this chunk's output (out_pos = symbols of the chunk before the end)
  std::string error;
  ByteBuf bytes;                // resolved
  std::vector<std::pair<uint32_t, uint64_t>> piece_crc;  // (crc, length) of the pieces between member ends
};

namespace {

// Skips a gzip member header at byte-aligned b.pos; false = not a gzip header (or truncated: *trunc).
bool skip_gzip_header(Bits& b, bool* trunc) {
  *trunc = false;
  const uint64_t byte = b.pos >> 3, nbytes = b.nbits >> 3;
  const uint8_t* h = b.p + byte;
  // what there is of the first three bytes decides: anything that is not the start of a member is trailing garbage (1 to 9 bytes of
  // padding after the last member are as good as 4000: gzip / zcat ignore them); a member's start that is cut short is a truncated file
  static const uint8_t magic[3] = {0x1f, 0x8b, 8};
  for (uint64_t i = 0; i < 3 && byte + i < nbytes; ++i)
    if (h[i] != magic[i]) return false;
  if (byte + 10 > nbytes) { *trunc = byte < nbytes; return false; }
  const uint8_t flg = h[3];
  uint64_t at = byte + 10;
  if (flg & 4) { if (at + 2 > nbytes) { *trunc = true; return false; } at += 2 + (uint64_t)(b.p[at] | (b.p[at + 1] << 8)); }
  if (flg & 8) { while (at < nbytes && b.p[at]) ++at; ++at; }
  if (flg & 16) { while (at < nbytes && b.p[at]) ++at; ++at; }
  if (flg & 2) at += 2;
  if (at > nbytes) { *trunc = true; return false; }
  b.pos = at << 3;
  return true;
}

}  // namespace

// Decodes from c.start_bit until the first block boundary q with stop(q) true, or the end of the stream.
// validate: the trial decode of a candidate start (ONE block, literals must be text; nothing is kept).
// Returns false on a corrupt stream (c.error set).
static bool decode_from(const uint8_t* data, uint64_t nbytes, PGzip::Chunk& c, const std::vector<uint64_t>* stops, uint64_t hard_stop_bit,
                        bool validate, uint64_t* validate_end) {
  static const FixedDist fixed_dist;
  Bits b{data, nbytes * 8, c.start_bit};
  PGzip::Chunk::SymBuf& out = c.sym;
  auto push = [&](uint16_t v) -> bool { if (out.n == out.cap && !out.reserve(out.n + 1)) return false; out.p[out.n++] = v; return true; };
  uint64_t member_base = 0;  // symbols of this chunk in front of the current member (references must not reach behind a member's start)
  bool in_known_member = false;  // the member began inside this chunk: its window is known to be empty
  if (c.at_member_start) {
    bool trunc = false;
    if (!skip_gzip_header(b, &trunc)) { c.error = trunc ? "the gzip stream ends inside a member header" : "not a gzip header"; return false; }
    in_known_member = true;
  }
  BlockCodes bc;
  uint64_t trial_out = 0;
  for (;;) {
    // ---- a block boundary ----
    if (!validate) {
      const uint64_t q = b.pos;
      if (q != c.start_bit) {
        bool stop = q >= hard_stop_bit;
        if (!stop && stops) {
          // (ascending; few of them: a linear look is fine)
          for (uint64_t s : *stops) { if (s == q) { stop = true; break; } if (s > q) break; }
        }
        if (stop) { c.end_bit = q; return true; }
      }
    }
    const uint32_t bfinal = b.get(1), btype = b.get(2);
    if (b.over) { c.error = "the gzip stream ends in the middle of a member (truncated file)"; return false; }
    if (btype == 3) { c.error = "corrupt deflate data (block type 3)"; return false; }
    if (btype == 0) {
      b.pos = (b.pos + 7) & ~7ull;
      const uint32_t len = b.get(16), nlen = b.get(16);
      if (b.over || (len ^ nlen) != 0xffffu) { c.error = b.over ? "the gzip stream ends in the middle of a member (truncated file)" : "corrupt stored block"; return false; }
      const uint64_t byte = b.pos >> 3;
      if (byte + len > nbytes) { c.error = "the gzip stream ends in the middle of a member (truncated file)"; return false; }
      if (validate) { for (uint32_t i = 0; i < len; ++i) if (!is_text(data[byte + i])) return false; trial_out += len; }
      else { if (!out.reserve(out.n + len)) { c.error = "out of memory"; return false; } for (uint32_t i = 0; i < len; ++i) out.p[out.n++] = data[byte + i]; }
      b.pos += (uint64_t)len * 8;
    } else {
      const Huff* dist;
      if (btype == 1) { fixed_codes(bc); dist = &fixed_dist.h; }
      else {
        if (!read_dynamic_header(b, bc, validate)) {
          if (validate) return false;
          c.error = b.over ? "the gzip stream ends in the middle of a member (truncated file)" : "corrupt dynamic block header";
          return false;
        }
        dist = &bc.dist;
      }
      if (!validate) {
        // ---- the fast loop: a local 64-bit bit buffer refilled eight bytes at a time, raw output pointer; up to three
        // literals per refill (3 x 15 bits), a length + distance pair needs at most 48.  Leaves to the careful loop below
        // within 16 bytes of the input's end. ----
        const uint32_t* lt = bc.lit.t.data();
        const uint32_t lmask = (1u << bc.lit.pb) - 1u;
        const int lpb = bc.lit.pb;
        const uint32_t* dt = dist->t.data();
        const uint32_t dmask = (1u << dist->pb) - 1u;
        const int dpb = dist->pb;
        const uint8_t* ip = data + (b.pos >> 3);
        const uint8_t* const safe = nbytes > 16 ? data + nbytes - 16 : data;
        uint64_t bitbuf = 0;
        unsigned bitcnt = 0;
        const bool fast = ip < safe;
        bool eob = false, bad = false;
        uint64_t n = out.n;
#define MG_REFILL() do { uint64_t w_; memcpy(&w_, ip, 8); bitbuf |= w_ << bitcnt; ip += (63 - bitcnt) >> 3; bitcnt |= 56; } while (0)
#define MG_LOOKUP(e_, tab_, mask_, pb_) do { e_ = tab_[bitbuf & mask_]; if (e_ & 0x80000000u) { const unsigned sb_ = e_ & 31u; e_ = tab_[((e_ >> 5) & 0x1fffffu) + ((bitbuf >> pb_) & ((1u << sb_) - 1u))]; } } while (0)
        if (fast) {  // the first refill starts at a byte; the bits of that byte in front of the position go
          MG_REFILL();
          const unsigned skip = (unsigned)(b.pos & 7);
          bitbuf >>= skip;
          bitcnt -= skip;
        }
        while (fast && ip < safe) {
          if (n + 4 * 258 > out.cap) { out.n = n; if (!out.reserve(n + (1u << 20))) { c.error = "out of memory"; return false; } }
          uint16_t* o = out.p;
          MG_REFILL();
          uint32_t e;
          MG_LOOKUP(e, lt, lmask, lpb);
          unsigned l = e & 31u;
          if (!l) { bad = true; break; }
          uint32_t sy = e >> 5;
          if (sy < 256) {
            bitbuf >>= l; bitcnt -= l;
            o[n++] = (uint16_t)sy;
            MG_LOOKUP(e, lt, lmask, lpb);
            l = e & 31u;
            if (!l) { bad = true; break; }
            sy = e >> 5;
            if (sy < 256) {
              bitbuf >>= l; bitcnt -= l;
              o[n++] = (uint16_t)sy;
              MG_LOOKUP(e, lt, lmask, lpb);
              l = e & 31u;
              if (!l) { bad = true; break; }
              sy = e >> 5;
              if (sy < 256) {
                bitbuf >>= l; bitcnt -= l;
                o[n++] = (uint16_t)sy;
                continue;
              }
            }
            // a length symbol (or the end of the block) after one or two literals: it needs a full buffer
            MG_REFILL();
          }
          bitbuf >>= l; bitcnt -= l;
          if (sy == 256) { eob = true; break; }
          if (sy > 285) { bad = true; break; }
          const unsigned le = kLenExtra[sy - 257];
          const uint32_t len = kLenBase[sy - 257] + (uint32_t)(bitbuf & ((1u << le) - 1u));
          bitbuf >>= le; bitcnt -= le;
          MG_LOOKUP(e, dt, dmask, dpb);
          l = e & 31u;
          const uint32_t ds = e >> 5;
          if (!l || ds > 29) { bad = true; break; }
          bitbuf >>= l; bitcnt -= l;
          const unsigned de = kDistExtra[ds];
          const uint32_t d = kDistBase[ds] + (uint32_t)(bitbuf & ((1u << de) - 1u));
          bitbuf >>= de; bitcnt -= de;
          if (in_known_member && (uint64_t)d > n - member_base) { bad = true; break; }
          if ((uint64_t)d <= n) {
            const uint16_t* src = o + n - d;
            uint16_t* dst = o + n;
            if (d >= len) memcpy(dst, src, (size_t)len * 2);  // (no overlap)
            else for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
          } else {
            for (uint32_t i = 0; i < len; ++i) {
              const uint64_t at = n + i;
              if ((uint64_t)d <= at) o[at] = o[at - d];
              else o[at] = (uint16_t)(0x8000u + (32768u - (uint32_t)((uint64_t)d - at)));
            }
          }
          n += len;
        }
#undef MG_REFILL
#undef MG_LOOKUP
        out.n = n;
        if (fast) b.pos = (uint64_t)(ip - data) * 8 - bitcnt;
        if (bad) { c.error = "corrupt deflate data"; return false; }
        if (eob) goto block_done;
      }
      for (;;) {
        const int s = bc.lit.decode(b);
        if (s < 0) { if (validate) return false; c.error = b.over ? "the gzip stream ends in the middle of a member (truncated file)" : "corrupt deflate data"; return false; }
        if (s < 256) {
          if (validate) { if (!is_text((uint32_t)s)) return false; if (++trial_out > (8u << 20)) return false; }
          else if (!push((uint16_t)s)) { c.error = "out of memory"; return false; }
          continue;
        }
        if (s == 256) break;
        if (s > 285) { if (validate) return false; c.error = "corrupt deflate data (length symbol)"; return false; }
        const uint32_t len = kLenBase[s - 257] + b.get(kLenExtra[s - 257]);
        const int ds = dist->decode(b);
        if (ds < 0 || ds > 29) { if (validate) return false; c.error = b.over ? "the gzip stream ends in the middle of a member (truncated file)" : "corrupt deflate data (distance symbol)"; return false; }
        const uint32_t d = kDistBase[ds] + b.get(kDistExtra[ds]);
        if (b.over) { if (validate) return false; c.error = "the gzip stream ends in the middle of a member (truncated file)"; return false; }
        if (validate) { trial_out += len; if (trial_out > (8u << 20)) return false; continue; }
        const uint64_t n = out.n;
        if (in_known_member && (uint64_t)d > n - member_base) { c.error = "corrupt deflate data (distance beyond the member's start)"; return false; }
        if (!out.reserve(n + len)) { c.error = "out of memory"; return false; }
        out.n = n + len;
        uint16_t* o = out.p;
        if ((uint64_t)d <= n) {
          for (uint32_t i = 0; i < len; ++i) o[n + i] = o[n + i - d];
        } else {
          // reaches behind the chunk's start: symbol 0x8000 + position in the 32 KB window in front of the chunk
          for (uint32_t i = 0; i < len; ++i) {
            const uint64_t at = n + i;
            if ((uint64_t)d <= at) o[at] = o[at - d];
            else o[at] = (uint16_t)(0x8000u + (32768u - (uint32_t)((uint64_t)d - at)));
          }
        }
      }
    }
  block_done:
    if (validate) {
      // one block decoded as text: what follows must again look like a block (or the member's end)
      if (bfinal) { *validate_end = b.pos; return true; }
      Bits nb = b;
      const uint32_t f2 = nb.get(1), t2 = nb.get(2);
      (void)f2;
      if (nb.over || t2 == 3) return false;
      if (t2 == 2) { BlockCodes bc2; if (!read_dynamic_header(nb, bc2, true)) return false; }
      *validate_end = b.pos;
      return true;
    }
    if (bfinal) {
      // ---- the member's end: trailer, then another member, trailing garbage, or the end of the file ----
      b.pos = (b.pos + 7) & ~7ull;
      const uint64_t byte = b.pos >> 3;
      if (byte + 8 > nbytes) { c.error = "the gzip stream ends in the middle of a member (truncated file)"; return false; }
      uint32_t crc, isz;
      memcpy(&crc, data + byte, 4);
      memcpy(&isz, data + byte + 4, 4);
      c.ends.push_back(PGzip::Chunk::MemberEnd{out.n, crc, isz});
      b.pos += 64;
      member_base = out.n;
      in_known_member = true;
      const uint64_t at = b.pos >> 3;
      if (at >= nbytes) { c.end_bit = b.pos; c.eof = true; return true; }
      bool trunc = false;
      if (!skip_gzip_header(b, &trunc)) {
        if (trunc) { c.error = "the gzip stream ends inside a member header"; return false; }
        c.end_bit = b.pos;  // trailing garbage (zero padding, a tape block's fill): ignored, as gzip / zcat do
        c.eof = true;
        return true;
      }
    }
  }
}

// First position in [from_bit, to_bit) where a dynamic block with BFINAL = 0 plausibly starts.
static bool find_block_start(const uint8_t* data, uint64_t nbytes, uint64_t from_bit, uint64_t to_bit, uint64_t* found) {
  Bits b{data, nbytes * 8, from_bit};
  for (uint64_t p = from_bit; p < to_bit; ++p) {
    b.pos = p;
    b.over = false;
    const uint64_t v = b.peek();
    if ((v & 7u) != 4u) continue;  // BFINAL = 0, BTYPE = 2 (the two type bits arrive low bit first: 10b -> value 2)
    // cheap rejections before any table is built
    const uint32_t hlit = (uint32_t)((v >> 3) & 31u), hdist = (uint32_t)((v >> 8) & 31u);
    if (hlit > 29 || hdist > 29) continue;
    {
      // the code-length code (3 bits per length, HCLEN + 4 of them, 17 bits in) must be a COMPLETE prefix code: Kraft's sum
      // over 2^(7 - len) = 2^7 — integer arithmetic on one 57-bit look, no table: rejects ~99 % of what got this far
      const int hclen = (int)((v >> 13) & 15u) + 4;
      b.pos = p + 17;
      const uint64_t w = b.peek();
      uint32_t kraft = 0, nz = 0;
      for (int i = 0; i < hclen; ++i) { const uint32_t l = (uint32_t)((w >> (3 * i)) & 7u); if (l) { kraft += 128u >> l; ++nz; } }
      if (kraft != 128u || nz < 2) continue;
    }
    PGzip::Chunk trial;
    trial.start_bit = p;
    uint64_t vend = 0;
    if (decode_from(data, nbytes, trial, nullptr, ~0ull, true, &vend)) { *found = p; return true; }
  }
  return false;
}

struct PGzip::Queue { std::deque<PGzip::Chunk::ByteBuf> q; };

// Buffers go round: a chunk's symbol buffer back to the pool once it is resolved, a byte buffer once the reader has copied it
// out — fresh memory per chunk costs a page fault per 4 KB (measured: the first 130 MB took 4 x as long as the next).
struct PGzip::Pools {
  std::mutex m;
  std::vector<PGzip::Chunk::SymBuf> syms;
  std::vector<PGzip::Chunk::ByteBuf> bytes;
  PGzip::Chunk::SymBuf take_sym() {
    std::lock_guard<std::mutex> lk(m);
    if (syms.empty()) return PGzip::Chunk::SymBuf();
    PGzip::Chunk::SymBuf b = std::move(syms.back());
    syms.pop_back();
    b.n = 0;
    return b;
  }
  void give_sym(PGzip::Chunk::SymBuf&& b) { std::lock_guard<std::mutex> lk(m); if (b.p && syms.size() < 256) syms.push_back(std::move(b)); }
  PGzip::Chunk::ByteBuf take_bytes() {
    std::lock_guard<std::mutex> lk(m);
    if (bytes.empty()) return PGzip::Chunk::ByteBuf();
    PGzip::Chunk::ByteBuf b = std::move(bytes.back());
    bytes.pop_back();
    b.n = 0;
    return b;
  }
  void give_bytes(PGzip::Chunk::ByteBuf&& b) { std::lock_guard<std::mutex> lk(m); if (b.p && bytes.size() < 256) bytes.push_back(std::move(b)); }
};

// ---------------------------------------------------------------------------------------------------------------------
PGzip::PGzip() : queue_(new Queue()), pools_(new Pools()) {}

PGzip::~PGzip() {
  {
    std::lock_guard<std::mutex> lk(m_);
    stop_ = true;
  }
  cv_space_.notify_all();
  cv_data_.notify_all();
  if (coordinator_.joinable()) coordinator_.join();
  if (map_ && map_ != MAP_FAILED) munmap(const_cast<uint8_t*>(map_), (size_t)size_);
  if (fd_ >= 0 && own_fd_) close(fd_);
}

std::unique_ptr<PGzip> PGzip::open(int fd, bool own_fd, uint64_t fsize, int nthreads, uint64_t chunk_bytes, std::string* err) {
  std::unique_ptr<PGzip> g(new PGzip());
  g->fd_ = fd;
  g->own_fd_ = own_fd;
  g->size_ = fsize;
  g->nthreads_ = nthreads < 1 ? 1 : nthreads;
  g->chunk_ = chunk_bytes < (64u << 10) ? (64u << 10) : chunk_bytes;
  if (fsize) {
    void* p = mmap(nullptr, (size_t)fsize, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) { if (err) *err = std::string("mmap failed: ") + strerror(errno); if (!own_fd) g->fd_ = -1; return nullptr; }
    g->map_ = (const uint8_t*)p;
    (void)madvise(p, (size_t)fsize, MADV_SEQUENTIAL);
  }
  g->coordinator_ = std::thread([raw = g.get()] { raw->run(); });
  return g;
}

template <class F>
static void parallel_for(int nthreads, uint64_t n, F&& fn) {
  if (n == 0) return;
  std::atomic<uint64_t> next{0};
  const int t = (int)((uint64_t)nthreads < n ? (uint64_t)nthreads : n);
  std::vector<std::thread> th;
  auto body = [&] { for (;;) { const uint64_t i = next.fetch_add(1); if (i >= n) return; fn(i); } };
  for (int i = 1; i < t; ++i) th.emplace_back(body);
  body();
  for (auto& x : th) x.join();
}

void PGzip::fail_with(const std::string& e) {
  std::lock_guard<std::mutex> lk(m_);
  if (error_.empty()) error_ = e;
  done_ = true;
  cv_data_.notify_all();
}

void PGzip::run() {
  const uint8_t* data = map_;
  const uint64_t nbytes = size_;
  if (nbytes == 0) { std::lock_guard<std::mutex> lk(m_); done_ = true; cv_data_.notify_all(); return; }
  const uint64_t per_batch = (uint64_t)nthreads_ * 2;
  const bool timing = mg::dbg("pgzip_timing") != 0;
  double t_find = 0, t_dec = 0, t_chain = 0, t_res = 0, t_wait = 0;
  uint64_t n_used = 0, n_all = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  uint64_t resume_bit = 0;        // where the next batch's first chunk starts
  bool resume_member = true;      // ... at a gzip member header
  std::vector<uint8_t> window(32768, 0);  // the 32 KB in front of the next chunk's output
  uint32_t run_crc = (uint32_t)crc32(0L, Z_NULL, 0);
  uint64_t run_len = 0;
  for (;;) {
    double t0 = now();
    {
      std::unique_lock<std::mutex> lk(m_);
      cv_space_.wait(lk, [&] { return stop_ || queued_bytes_ < max_queued_; });
      if (stop_) return;
    }
    t_wait += now() - t0; t0 = now();
    // the batch: chunk 0 starts at resume_bit; the others look for a start in their own byte range
    const uint64_t first_chunk = resume_bit / 8 / chunk_;
    const uint64_t nchunks_total = (nbytes + chunk_ - 1) / chunk_;
    uint64_t b1 = first_chunk + per_batch;
    if (b1 > nchunks_total) b1 = nchunks_total;
    const uint64_t n = b1 - first_chunk;
    std::vector<Chunk> ch(n);
    ch[0].index = first_chunk; ch[0].start_bit = resume_bit; ch[0].have_start = true; ch[0].at_member_start = resume_member;
    parallel_for(nthreads_, n - 1, [&](uint64_t i) {
      Chunk& c = ch[i + 1];
      c.index = first_chunk + i + 1;
      uint64_t from = c.index * chunk_ * 8, to = (c.index + 1) * chunk_ * 8;
      if (to > nbytes * 8) to = nbytes * 8;
      if (from <= resume_bit) from = resume_bit + 1;
      uint64_t f = 0;
      if (from < to && find_block_start(data, nbytes, from, to, &f)) { c.have_start = true; c.start_bit = f; }
    });
    t_find += now() - t0; t0 = now();
    std::vector<uint64_t> starts;
    for (uint64_t i = 1; i < n; ++i) if (ch[i].have_start) starts.push_back(ch[i].start_bit);
    const uint64_t batch_end_bit = b1 >= nchunks_total ? ~0ull : b1 * chunk_ * 8;
    parallel_for(nthreads_, n, [&](uint64_t i) {
      Chunk& c = ch[i];
      if (!c.have_start) return;
      c.sym = pools_->take_sym();
      if (!c.sym.reserve((size_t)(chunk_ * 5))) { c.error = "out of memory"; return; }
      (void)decode_from(data, nbytes, c, &starts, batch_end_bit, false, nullptr);
    });
    t_dec += now() - t0; t0 = now();
    // ---- chain the chunks: the next used chunk is the one whose start is where this one stopped ----
    std::vector<uint64_t> used;
    uint64_t cur = 0;
    bool eof = false;
    for (;;) {
      Chunk& c = ch[cur];
      used.push_back(cur);
      if (!c.error.empty()) { fail_with(c.error); return; }
      if (c.eof) { eof = true; break; }
      uint64_t nxt = 0;
      for (uint64_t j = cur + 1; j < n; ++j) if (ch[j].have_start && ch[j].start_bit == c.end_bit) { nxt = j; break; }
      if (!nxt) break;  // stopped at the batch's end (or ran over the starts that were false)
      cur = nxt;
    }
    resume_bit = ch[used.back()].end_bit;
    resume_member = false;
    // ---- windows: only the last 32 KB of every used chunk have to be resolved for the next one ----
    std::vector<std::vector<uint8_t>> win_in(used.size());
    for (size_t u = 0; u < used.size(); ++u) {
      Chunk& c = ch[used[u]];
      win_in[u] = window;
      const uint64_t m = c.sym.size();
      std::vector<uint8_t> nw(32768);
      if (m >= 32768) {
        for (uint64_t i = 0; i < 32768; ++i) { const uint16_t s = c.sym.p[m - 32768 + i]; nw[i] = s & 0x8000u ? window[s & 0x7fffu] : (uint8_t)s; }
      } else {
        memcpy(nw.data(), window.data() + m, (size_t)(32768 - m));
        for (uint64_t i = 0; i < m; ++i) { const uint16_t s = c.sym.p[i]; nw[32768 - m + i] = s & 0x8000u ? window[s & 0x7fffu] : (uint8_t)s; }
      }
      window.swap(nw);
    }
    t_chain += now() - t0; t0 = now();
    n_used += used.size(); n_all += n;
    // ---- resolve every used chunk to bytes, CRC of the pieces between member ends (parallel) ----
    parallel_for(nthreads_, used.size(), [&](uint64_t u) {
      Chunk& c = ch[used[u]];
      const std::vector<uint8_t>& w = win_in[u];
      const uint64_t m = c.sym.size();
      c.bytes = pools_->take_bytes();
      if (!c.bytes.reserve((size_t)m + 1)) { c.error = "out of memory"; return; }
      c.bytes.n = (size_t)m;
      const uint16_t* s = c.sym.p;
      uint8_t* o = c.bytes.p;
      for (uint64_t i = 0; i < m; ++i) o[i] = s[i] & 0x8000u ? w[s[i] & 0x7fffu] : (uint8_t)s[i];
      pools_->give_sym(std::move(c.sym));
      uint64_t at = 0;
      for (size_t e = 0; e <= c.ends.size(); ++e) {
        const uint64_t to = e < c.ends.size() ? c.ends[e].out_pos : m;
        uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
        uint64_t left = to - at;
        const uint8_t* p = o + at;
        while (left) { const uInt step = (uInt)(left > (1u << 30) ? (1u << 30) : left); crc = (uint32_t)crc32(crc, p, step); p += step; left -= step; }
        c.piece_crc.emplace_back(crc, to - at);
        at = to;
      }
    });
    // ---- members' CRC-32 / ISIZE, in order; hand the bytes over ----
    for (size_t u = 0; u < used.size(); ++u) {
      Chunk& c = ch[used[u]];
      for (size_t e = 0; e < c.piece_crc.size(); ++e) {
        run_crc = (uint32_t)crc32_combine(run_crc, c.piece_crc[e].first, (z_off_t)c.piece_crc[e].second);
        run_len += c.piece_crc[e].second;
        if (e < c.ends.size()) {
          if (run_crc != c.ends[e].crc || (uint32_t)run_len != c.ends[e].isize) { fail_with("gzip member fails its CRC-32 / length check (corrupt data)"); return; }
          run_crc = (uint32_t)crc32(0L, Z_NULL, 0);
          run_len = 0;
        }
      }
      if (!c.error.empty()) { fail_with(c.error); return; }
      std::lock_guard<std::mutex> lk(m_);
      queued_bytes_ += c.bytes.size();
      queue_->q.emplace_back(std::move(c.bytes));
      cv_data_.notify_all();
    }
    t_res += now() - t0;
    if (eof) break;
    if (resume_bit >= nbytes * 8) { fail_with("the gzip stream ends in the middle of a member (truncated file)"); return; }
  }
  if (timing)
    fprintf(stderr, "[pgzip] %d threads, chunk %llu: find %.3f s, decode %.3f s, chain+windows %.3f s, resolve+crc+queue %.3f s, waiting for the reader %.3f s; %llu of %llu chunks used\n",
            nthreads_, (unsigned long long)chunk_, t_find, t_dec, t_chain, t_res, t_wait, (unsigned long long)n_used, (unsigned long long)n_all);
  std::lock_guard<std::mutex> lk(m_);
  done_ = true;
  cv_data_.notify_all();
}

int64_t PGzip::read(uint8_t* dst, uint64_t cap) {
  uint64_t got = 0;
  while (got < cap) {
    std::unique_lock<std::mutex> lk(m_);
    cv_data_.wait(lk, [&] { return !queue_->q.empty() || done_; });
    if (queue_->q.empty()) {
      if (!error_.empty()) return -1;
      break;  // the end of the stream
    }
    Chunk::ByteBuf& front = queue_->q.front();
    const uint64_t avail = front.size() - front_off_;
    const uint64_t take = avail < cap - got ? avail : cap - got;
    const uint8_t* src = front.p + front_off_;
    lk.unlock();
    memcpy(dst + got, src, (size_t)take);  // (the front buffer is only ever popped by this thread)
    lk.lock();
    got += take;
    front_off_ += take;
    if (front_off_ == front.size()) {
      queued_bytes_ -= front.size();
      pools_->give_bytes(std::move(front));
      queue_->q.pop_front();
      front_off_ = 0;
      cv_space_.notify_all();
    }
  }
  return (int64_t)got;
}

std::string PGzip::error() {
  std::lock_guard<std::mutex> lk(m_);
  return error_;
}

}  // namespace mg

// ---- C ABI: a gzip file's text through the parallel decoder, for callers outside the streaming pipeline (plain host code:
// needs no device and no mg_init) ----
struct mg_gunzip {
  std::unique_ptr<mg::PGzip> g;
};

extern "C" {

int mg_gunzip_open(const char* path, int nthreads, mg_gunzip** out) {
  if (!path || !out) return mg::fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return mg::fail(MG_ERR_ARG, "cannot open %s: %s", path, strerror(errno));
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); return mg::fail(MG_ERR_ARG, "cannot stat %s", path); }
  if (nthreads <= 0) { unsigned hw = std::thread::hardware_concurrency(); nthreads = (int)(hw == 0 ? 4 : (hw > 64 ? 64 : hw)); }
  std::string err;
  uint64_t chunk = 1ull << 20;
  if (mg::dbg("pgzip_chunk") > 0) chunk = (uint64_t)mg::dbg("pgzip_chunk");
  std::unique_ptr<mg::PGzip> g = mg::PGzip::open(fd, true, (uint64_t)sb.st_size, nthreads, chunk, &err);
  if (!g) { close(fd); return mg::fail(MG_ERR_ARG, "%s: %s", path, err.c_str()); }
  *out = new mg_gunzip{std::move(g)};
  return MG_OK;
}

int mg_gunzip_read(mg_gunzip* h, uint8_t* dst, uint64_t cap, uint64_t* n) {
  if (!h || !n || (cap && !dst)) return mg::fail(MG_ERR_ARG, "null argument");
  const int64_t got = h->g->read(dst, cap);
  if (got < 0) return mg::fail(MG_ERR_ARG, "%s", h->g->error().c_str());
  *n = (uint64_t)got;
  return MG_OK;
}

void mg_gunzip_close(mg_gunzip* h) { delete h; }

// ---- zcat of many small files (the selected genomes, scripts/select_db.py:103-105) ----
// One file: every member inflated by zlib (gzip or zlib header, as `zlib.decompressobj(47)`); bytes after the last member that
// do not begin another member are ignored, as gzip / zcat do.  false + *why for a stream that is not gzip, corrupt, or ends
// inside a member: such a file contributes NOTHING (zcat's partial output is not reproduced; the caller reports the file).
static bool zcat_one(const char* path, std::string* text, std::string* why) {
  text->clear();
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { *why = strerror(errno); return false; }
  struct stat sb;
  if (fstat(fd, &sb) != 0) { *why = strerror(errno); close(fd); return false; }
  std::vector<uint8_t> in((size_t)sb.st_size);
  size_t got = 0;
  while (got < in.size()) {
    const ssize_t r = pread(fd, in.data() + got, in.size() - got, (off_t)got);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) break;
    got += (size_t)r;
  }
  close(fd);
  if (got != in.size()) { *why = "short read"; return false; }
  if (in.empty()) return true;
  z_stream z;
  memset(&z, 0, sizeof(z));
  if (inflateInit2(&z, 47) != Z_OK) { *why = "zlib initialisation failed"; return false; }
  z.next_in = in.data();
  z.avail_in = (uInt)(in.size() > 0x7fffffffu ? 0x7fffffffu : in.size());
  size_t fed_to = z.avail_in;  // bytes of `in` handed to zlib so far
  text->resize(in.size() * 4 + (1u << 16));
  size_t at = 0;
  bool ok = true, in_member = false;
  for (;;) {
    if (at == text->size()) text->resize(text->size() * 2);
    const size_t room = text->size() - at;
    z.next_out = reinterpret_cast<Bytef*>(&(*text)[at]);
    z.avail_out = (uInt)(room > 0x7fffffffu ? 0x7fffffffu : room);
    const uInt out0 = z.avail_out;
    if (z.avail_in == 0 && fed_to < in.size()) {
      const size_t more = in.size() - fed_to;
      z.avail_in = (uInt)(more > 0x7fffffffu ? 0x7fffffffu : more);
      fed_to += z.avail_in;
    }
    if (z.avail_in) in_member = true;
    const int rc = inflate(&z, Z_NO_FLUSH);
    at += out0 - z.avail_out;
    if (rc == Z_STREAM_END) {
      in_member = false;
      const size_t left = (size_t)z.avail_in + (in.size() - fed_to);
      if (left == 0) break;
      const uint8_t* nx = z.avail_in ? z.next_in : in.data() + fed_to;
      const bool again = nx[0] == 0x1f && (left < 2 || nx[1] == 0x8b);
      if (!again) break;  // trailing garbage
      if (inflateReset(&z) != Z_OK) { ok = false; *why = "zlib reset failed"; break; }
      continue;
    }
    if (rc == Z_OK) continue;
    if (rc == Z_BUF_ERROR && z.avail_in == 0 && fed_to == in.size()) {
      ok = false; *why = "the gzip stream ends in the middle of a member"; break;
    }
    if (rc == Z_BUF_ERROR) continue;  // (no room: the buffer grows at the top of the loop)
    ok = false; *why = z.msg ? z.msg : "not a gzip stream"; break;
  }
  (void)in_member;
  inflateEnd(&z);
  if (!ok) { text->clear(); return false; }
  text->resize(at);
  return true;
}

int mg_zcat_files(const char* const* paths, uint64_t nfiles, const char* out_path, int nthreads, uint64_t* bytes_out, uint8_t* failed) {
  if (!out_path || (nfiles && !paths)) return mg::fail(MG_ERR_ARG, "null argument");
  if (nthreads <= 0) {
    const unsigned hw = std::thread::hardware_concurrency();
    nthreads = (int)(hw == 0 ? 4 : (hw > 32 ? 32 : hw));
  }
  if ((uint64_t)nthreads > nfiles) nthreads = nfiles ? (int)nfiles : 1;
  const int fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) return mg::fail(MG_ERR_ARG, "cannot open %s: %s", out_path, strerror(errno));
  // In WINDOWS of files whose text is about a gigabyte (estimated from their sizes): inflated by all threads, written at their
  // offsets, released — a selection of thousands of genomes does not sit in host memory whole (the reference streams one zcat per
  // genome into the append handle, scripts/select_db.py:99-105).  Order and bytes are those of one pass.
  std::vector<std::string> why(nfiles);
  std::vector<uint8_t> bad(nfiles, 0);
  std::atomic<bool> werr{false};
  uint64_t base = 0;
  const uint64_t budget = 1ull << 30;
  for (uint64_t w0 = 0; w0 < nfiles && !werr.load();) {
    uint64_t w1 = w0, est = 0;
    while (w1 < nfiles) {
      struct stat sb;
      const uint64_t sz = stat(paths[w1], &sb) == 0 ? (uint64_t)sb.st_size : 0;
      if (w1 > w0 && est + 5 * sz > budget) break;
      est += 5 * sz;
      ++w1;
    }
    const uint64_t nw = w1 - w0;
    std::vector<std::string> text(nw);
    std::atomic<uint64_t> next{0};
    auto inflate_some = [&]() {
      for (;;) {
        const uint64_t i = next.fetch_add(1);
        if (i >= nw) return;
        bad[w0 + i] = zcat_one(paths[w0 + i], &text[i], &why[w0 + i]) ? 0 : 1;
      }
    };
    {
      std::vector<std::thread> th;
      for (int t = 1; t < nthreads && (uint64_t)t < nw; ++t) th.emplace_back(inflate_some);
      inflate_some();
      for (auto& t : th) t.join();
    }
    std::vector<uint64_t> at(nw + 1, base);
    for (uint64_t i = 0; i < nw; ++i) at[i + 1] = at[i] + text[i].size();
    for (uint64_t i = 0; i < nw; ++i)
      if (bad[w0 + i]) fprintf(stderr, "zcat: %s: %s\n", paths[w0 + i], why[w0 + i].c_str());  // (as zcat: a line on stderr, and on with the rest)
    next.store(0);
    auto write_some = [&]() {
      for (;;) {
        const uint64_t i = next.fetch_add(1);
        if (i >= nw) return;
        size_t done = 0;
        while (done < text[i].size()) {
          const ssize_t w = pwrite(fd, text[i].data() + done, text[i].size() - done, (off_t)(at[i] + done));
          if (w < 0 && errno == EINTR) continue;
          if (w <= 0) { werr.store(true); return; }
          done += (size_t)w;
        }
      }
    };
    {
      std::vector<std::thread> th;
      const int wt = nthreads > 8 ? 8 : nthreads;
      for (int t = 1; t < wt && (uint64_t)t < nw; ++t) th.emplace_back(write_some);
      write_some();
      for (auto& t : th) t.join();
    }
    base = at[nw];
    w0 = w1;
  }
  std::vector<uint64_t> at(nfiles + 1, base);
  const int cerr = close(fd);
  if (werr.load() || cerr != 0) return mg::fail(MG_ERR_ARG, "writing %s failed: %s", out_path, strerror(errno));
  if (bytes_out) *bytes_out = at[nfiles];
  if (failed) for (uint64_t i = 0; i < nfiles; ++i) failed[i] = bad[i];
  return MG_OK;
}

}  // extern "C"
