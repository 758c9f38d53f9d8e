// mg_stream.hip — files -> HBM -> the stages, pipelined inside the library.
//
// The reference hands its reads file to `kmc` and its SAM text to a Python loop (scripts/select_db.py:43-52, `.gz` is
// expected input :146-148; scripts/map_and_profile.py:201-217).  Here the text goes through in PIECES:
//
//   reader threads      page cache -> page-locked slots (plain file: positional reads, any number of threads in
//                       parallel; gzip: zlib inflate straight into the slots — one thread for a plain gzip stream,
//                       which cannot be entered in the middle, all threads for BGZF, whose blocks say how long they are)
//   DMA stream          slot i -> device text buffer (i mod 3), behind a HEADROOM in front of which the unfinished
//                       last record (line) of piece i - 1 is placed ON THE DEVICE — so the upload of piece i + 1 does not
//                       depend on where piece i was cut and runs while piece i is parsed
//   library stream      [carry | piece] -> mg_reads_parse_prefix_dev / the SAM tokeniser -> the consumer
//                       (mg_sketch_stream_add_dev: ONE set of counting tables for the whole file)
//
// Nothing of the text ever exists as a host array beyond the slots; the host never looks at a byte of it.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mg_inflate.h"
#include "mg_internal.h"
#include "mg_pgzip.h"

namespace mg {

// ---------------------------------------------------------------------------------------------------------------------
// byte sources
// ---------------------------------------------------------------------------------------------------------------------
struct Source {
  std::string error;  // set by fill() on failure (a reader thread must not touch the library's error text)
  int error_code = MG_ERR_ARG;
  virtual ~Source() {}
  // Piece i of the byte stream into dst (at most cap bytes).  *last = this is the stream's final piece (possibly empty).
  // Returns the bytes written, -1 on error.  Called for i = 0, 1, 2, ... — by ANY thread in any order when parallel(),
  // else by one thread in order.
  virtual int64_t fill(uint64_t i, uint8_t* dst, uint64_t cap, bool* last) = 0;
  virtual bool parallel() const = 0;
};

struct PlainSource : Source {
  int fd = -1;
  uint64_t off = 0, len = 0, cap_ = 0;
  ~PlainSource() override { if (fd >= 0) close(fd); }
  bool parallel() const override { return true; }
  int64_t fill(uint64_t i, uint8_t* dst, uint64_t cap, bool* last) override {
    const uint64_t at = i * cap;
    const uint64_t want = at >= len ? 0 : (len - at < cap ? len - at : cap);
    *last = at + want >= len;
    uint64_t got = 0;
    while (got < want) {
      const ssize_t n = pread(fd, dst + got, want - got, (off_t)(off + at + got));
      if (n < 0) { error = std::string("read failed: ") + strerror(errno); return -1; }
      if (n == 0) { error = "file is shorter than its size said"; return -1; }
      got += (uint64_t)n;
    }
    return (int64_t)got;
  }
};

// A gzip (or zlib) stream, members concatenated or not: inflated in order by ONE thread, straight into the slots.
struct GzipSource : Source {
  int fd = -1;
  z_stream zs;
  bool zs_live = false, ended = false, in_member = false, at_boundary = false;
  std::vector<uint8_t> in;
  uint64_t in_off = 0;  // file offset of the next compressed byte to read
  GzipSource() { memset(&zs, 0, sizeof(zs)); in.resize(4u << 20); }
  ~GzipSource() override { if (zs_live) inflateEnd(&zs); if (fd >= 0) close(fd); }
  bool parallel() const override { return false; }
  int64_t fill(uint64_t, uint8_t* dst, uint64_t cap, bool* last) override {
    *last = false;
    if (ended) { *last = true; return 0; }
    if (!zs_live) {
      if (inflateInit2(&zs, 15 + 32) != Z_OK) { error = "inflateInit2 failed"; return -1; }  // gzip or zlib header, detected
      zs_live = true;
    }
    zs.next_out = dst;
    uint64_t room = cap;
    while (room > 0) {
      if (zs.avail_in == 0) {
        const ssize_t n = pread(fd, in.data(), in.size(), (off_t)in_off);
        if (n < 0) { error = std::string("read failed: ") + strerror(errno); return -1; }
        if (n == 0) {  // end of the file
          if (in_member) { error = "the gzip stream ends in the middle of a member (truncated file)"; return -1; }
          ended = true;
          break;
        }
        in_off += (uint64_t)n;
        zs.next_in = in.data();
        zs.avail_in = (uInt)n;
      }
      if (at_boundary) {
        // between members only another gzip member may follow; anything else — zero padding, a tape block's fill — is
        // trailing garbage, which gzip / zcat ignore with a warning (the reference feeds `.gz` files to kmc and zcat,
        // scripts/select_db.py:105,146-148)
        if (zs.next_in[0] != 0x1f || (zs.avail_in > 1 && zs.next_in[1] != 0x8b)) { ended = true; break; }
        at_boundary = false;
      }
      zs.avail_out = (uInt)(room > 0x40000000ull ? 0x40000000ull : room);
      const uInt before = zs.avail_out;
      in_member = true;
      const int rc = inflate(&zs, Z_NO_FLUSH);
      room -= before - zs.avail_out;
      if (rc == Z_STREAM_END) {
        in_member = false;  // a member ended: another may follow (bgzip, cat a.gz b.gz, pigz -i)
        at_boundary = true;
        if (inflateReset(&zs) != Z_OK) { error = "inflateReset failed"; return -1; }
      } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
        error = std::string("inflate failed: ") + (zs.msg ? zs.msg : "corrupt gzip data");
        return -1;
      }
    }
    if (ended) *last = true;
    return (int64_t)(cap - room);
  }
};

// The same stream inflated by MANY host threads (mg_pgzip.hip: deflate entered in the middle, the pugz / rapidgzip scheme):
// the decoder runs ahead in the background, one reader thread copies its output into the slots in order.
struct ParallelGzipSource : Source {
  std::unique_ptr<PGzip> g;
  bool ended = false;
  bool parallel() const override { return false; }
  int64_t fill(uint64_t, uint8_t* dst, uint64_t cap, bool* last) override {
    *last = false;
    if (ended) { *last = true; return 0; }
    const int64_t n = g->read(dst, cap);
    if (n < 0) { error = g->error(); return -1; }
    if ((uint64_t)n < cap) { ended = true; *last = true; }
    return n;
  }
};

// BGZF (bgzip / htslib): gzip members of at most 64 KB that carry their own compressed size (extra field 'BC'), so the
// file can be indexed without inflating anything and every block inflated independently — by all reader threads.
struct BgzfSource : Source {
  struct Block { uint64_t coff; uint32_t csize, isize; };
  struct Group { uint64_t first, count, bytes; };
  int fd = -1;
  std::vector<Block> blocks;
  std::vector<Group> groups;  // consecutive blocks of at most one piece together
  ~BgzfSource() override { if (fd >= 0) close(fd); }
  bool parallel() const override { return true; }

  // true when the file is BGZF from its first to its last byte (every member has the BC field); builds the index
  static bool is_bgzf_header(const uint8_t* h) {
    return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 && h[12] == 'B' && h[13] == 'C' &&
           h[14] == 2 && h[15] == 0;
  }
  bool index(uint64_t fsize, uint64_t cap) {
    uint64_t at = 0;
    uint8_t h[18], tail[4];
    while (at < fsize) {
      if (fsize - at < 28 || pread(fd, h, 18, (off_t)at) != 18 || !is_bgzf_header(h)) return false;
      const uint32_t csize = (uint32_t)(h[16] | (h[17] << 8)) + 1u;
      if (csize < 26 || at + csize > fsize) return false;
      if (pread(fd, tail, 4, (off_t)(at + csize - 4)) != 4) return false;
      const uint32_t isize = (uint32_t)tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
      if (isize > 65536u) return false;
      blocks.push_back(Block{at, csize, isize});
      at += csize;
    }
    Group g{0, 0, 0};
    for (uint64_t b = 0; b < blocks.size(); ++b) {
      if (g.count && g.bytes + blocks[b].isize > cap) { groups.push_back(g); g = Group{b, 0, 0}; }
      ++g.count;
      g.bytes += blocks[b].isize;
    }
    groups.push_back(g);  // (an empty file has one empty group: the final piece)
    return true;
  }
  int64_t fill(uint64_t i, uint8_t* dst, uint64_t cap, bool* last) override {
    *last = i + 1 >= groups.size();
    if (i >= groups.size()) return 0;
    const Group& g = groups[i];
    if (g.bytes > cap) { error = "BGZF group larger than a slot"; return -1; }
    std::vector<uint8_t> cbuf(65536 + 64);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) { error = "inflateInit2 failed"; return -1; }  // raw deflate: the header is skipped by hand
    uint64_t out = 0;
    for (uint64_t b = g.first; b < g.first + g.count; ++b) {
      const Block& blk = blocks[b];
      if (pread(fd, cbuf.data(), blk.csize, (off_t)blk.coff) != (ssize_t)blk.csize) { error = "short read of a BGZF block"; inflateEnd(&zs); return -1; }
      const uint32_t xlen = (uint32_t)(cbuf[10] | (cbuf[11] << 8));
      const uint32_t hdr = 12 + xlen;
      if (hdr + 8 > blk.csize) { error = "corrupt BGZF block header"; inflateEnd(&zs); return -1; }
      zs.next_in = cbuf.data() + hdr;
      zs.avail_in = blk.csize - hdr - 8;
      zs.next_out = dst + out;
      zs.avail_out = blk.isize;
      const int rc = inflate(&zs, Z_FINISH);
      if (rc != Z_STREAM_END || zs.avail_out != 0) {
        error = std::string("inflate of a BGZF block failed: ") + (zs.msg ? zs.msg : "size mismatch");
        inflateEnd(&zs);
        return -1;
      }
      out += blk.isize;
      inflateReset(&zs);
    }
    inflateEnd(&zs);
    return (int64_t)out;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// A plain FASTQ or SAM file THINNED by the reader threads to what the device parsers read.  From the page cache both
// files go up at the PCIe link's rate (3.0 + 3.5 GB per 10M reads: 0.14 s of a 0.28 s pair of command lines), and half
// of a FASTQ record (the quality line, the '+', most of the header) and two thirds of a SAM line (SEQ and QUAL, of which
// only len(SEQ) is used, scripts/map_and_profile.py:213) are never looked at on the device.
//   FASTQ  a record -> ">\n<sequence line>\n": the device parses single-line FASTA.  The reader that owns a record's
//          header line checks what k_read_lengths checks (header '@...', separator '+...') and what the parser checks
//          at the end of the file (only blank lines after the last whole record).  Which line of its record a line is
//          follows from the number of newlines in front of it: every reader counts its raw piece first and waits for
//          the counts of the pieces before it.
//   SAM    a line with at least 11 fields of str.split(): SEQ -> MG_THIN_MARK + len(SEQ) in decimal (a '*' stays),
//          QUAL -> '*'; every other byte, white space included, is kept, so the tokeniser sees the same fields, the
//          same errors and the same line numbers.  k_sam_parse takes the length from the mark (its `thin` flag).
// A raw piece is read into the slot it will leave from, behind a little room, and thinned in place (the output never
// overtakes the input: a FASTQ record shrinks, a SAM line grows by one byte only when SEQ and QUAL are one character
// each); a reader owns the lines / records that START in its piece and reads on past its end to finish the last one.
// OPT-IN (mg_debug_set("stream_thin", 1)), because it does not pay on the hosts it was measured on (profiles/r04/stream_ceilings.txt): the plain
// streams run at ~50 GB/s of text, which is the PCIe link's rate AND about what these hosts read out of the page cache at all
// (tools/pread_probe.py: 60 GB/s at four threads, less with more).  Thinning takes bytes off the link, not out of the page cache,
// and adds work per byte: with the scanner below (newline / white-space positions 32 bytes per step; the first form, one memchr per
// line, thinned 1.8 GB/s per reader) 16 readers draw level with 4 plain ones on FASTQ (0.066 s for 3.2 GB) and stay behind on SAM
// (0.097 against 0.078 s for 3.7 GB).  On a host whose memory outruns its link it is one environment variable away.
// ---------------------------------------------------------------------------------------------------------------------
static inline bool h_is_ws(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); }

// first whitespace byte in [p, end), or end (the device's scan_to_ws: eight bytes per step)
static inline const uint8_t* h_scan_to_ws(const uint8_t* p, const uint8_t* end) {
  while (p + 8 <= end) {
    uint64_t x;
    memcpy(&x, p, 8);
    uint64_t m = (x - 0x2121212121212121ull) & ~x & 0x8080808080808080ull;
    while (m) {
      const unsigned b = (unsigned)__builtin_ctzll(m) >> 3;
      if (h_is_ws((uint8_t)(x >> (8 * b)))) return p + b;
      m &= m - 1;
    }
    p += 8;
  }
  while (p < end && !h_is_ws(*p)) ++p;
  return p;
}

// ---- candidate positions, 32 bytes per step where the host has AVX2 ----
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define MG_HOST_AVX2 1
__attribute__((target("avx2"))) static inline uint32_t mask32_eq(const uint8_t* p, uint8_t v) {
  const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
  return (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, _mm256_set1_epi8((char)v)));
}
__attribute__((target("avx2"))) static inline uint32_t mask32_le(const uint8_t* p, uint8_t v) {
  const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
  return (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_min_epu8(x, _mm256_set1_epi8((char)v)), x));
}
static bool host_has_avx2() {  // (knob no_avx2: the byte-by-byte scanner, for the tests)
  static const bool yes = __builtin_cpu_supports("avx2") && !dbg("no_avx2");  // (read once per process)
  return yes;
}
#endif

static inline uint64_t h_count_nl(const uint8_t* p, uint64_t n) {
  uint64_t c = 0, i = 0;
#ifdef MG_HOST_AVX2
  if (host_has_avx2())
    for (; i + 32 <= n; i += 32) c += (uint64_t)__builtin_popcount(mask32_eq(p + i, '\n'));
#endif
  for (; i < n; ++i) c += p[i] == '\n';
  return c;
}

// The positions in [*, *n) of the bytes that are '\n' (LE = false) or <= 0x20 (LE = true: every white-space byte of str.split()
// is one, and a few control characters that are not), in ascending order.  *n may grow between calls (the reader reads on).
template <bool LE>
struct CandIter {
  const uint8_t* base;
  const uint64_t* n;
  uint64_t pos;    // everything below pos has been looked at
  uint64_t wbase = 0;
  uint32_t mask = 0;
  CandIter(const uint8_t* b, const uint64_t* nn, uint64_t from) : base(b), n(nn), pos(from) {}
  // the next candidate at or after the last one returned + 1; *n when there is none (so far)
  uint64_t next() {
    for (;;) {
      if (mask) {
        const unsigned bit = (unsigned)__builtin_ctz(mask);
        mask &= mask - 1;
        return wbase + bit;
      }
#ifdef MG_HOST_AVX2
      if (pos + 32 <= *n && host_has_avx2()) {
        mask = LE ? mask32_le(base + pos, 0x20) : mask32_eq(base + pos, '\n');
        wbase = pos;
        pos += 32;
        continue;
      }
#endif
      while (pos < *n) {
        const uint8_t c = base[pos++];
        if (LE ? c <= 0x20 : c == '\n') return pos - 1;
      }
      return *n;
    }
  }
  // forget what lies below `from` (the caller jumped ahead, e.g. past a record it has taken)
  void skip_to(uint64_t from) {
    if (from <= pos) {
      if (mask && from > wbase) mask &= from - wbase >= 32 ? 0u : ~((1u << (from - wbase)) - 1u);
    } else {
      mask = 0;
      pos = from;
    }
  }
};

struct ThinSource : Source {
  enum Kind { kFastq = 0, kSam = 1 };
  int fd = -1;
  Kind kind = kFastq;
  uint64_t len = 0, raw = 0, npieces = 1;
  static constexpr uint64_t kRoom = 4096;  // in front of the raw bytes: what a piece of absurd SAM lines may grow by
  // FASTQ: newlines per raw piece, as the readers come to know them
  std::mutex m;
  std::condition_variable cv;
  std::vector<int64_t> nl;
  bool dead = false;  // a reader failed: nobody waits for its count
  ~ThinSource() override { if (fd >= 0) close(fd); }
  bool parallel() const override { return true; }

  // the raw piece size for slots of `cap` bytes
  static uint64_t raw_for(uint64_t cap) {
    const uint64_t tail = cap / 4 > (4u << 20) ? (4u << 20) : cap / 4;
    return cap > kRoom + tail + 1 ? cap - kRoom - tail : 1;
  }
  void setup(uint64_t file_len, uint64_t cap) {
    len = file_len;
    raw = raw_for(cap);
    npieces = len ? (len + raw - 1) / raw : 1;
    if (kind == kFastq) nl.assign(npieces, -1);
  }
  bool read_at(uint8_t* dst, uint64_t n, uint64_t at, std::string* why) {
    uint64_t got = 0;
    while (got < n) {
      const ssize_t r = pread(fd, dst + got, n - got, (off_t)(at + got));
      if (r < 0 && errno == EINTR) continue;
      if (r < 0) { *why = std::string("read failed: ") + strerror(errno); return false; }
      if (r == 0) { *why = "file is shorter than its size said"; return false; }
      got += (uint64_t)r;
    }
    return true;
  }
  int64_t give_up(const std::string& why, int code = MG_ERR_ARG) {
    {
      std::lock_guard<std::mutex> lk(m);
      if (!dead) { error = why; error_code = code; }  // (the first failure is the one reported)
      dead = true;
    }
    cv.notify_all();
    return -1;
  }

  int64_t fill(uint64_t i, uint8_t* dst, uint64_t cap, bool* last) override {
    *last = i + 1 >= npieces;
    if (i >= npieces || len == 0) return 0;
    const uint64_t a = i * raw, b = a + raw < len ? a + raw : len;
    uint8_t* const in0 = dst + kRoom;          // raw byte a lands here; byte a - 1 right in front of it
    uint64_t have = b - a;                     // raw bytes in the slot so far: [a, a + have)
    std::string io_error;  // (this call's own: `error` belongs to whichever reader failed first)
    if (a > 0) { if (!read_at(in0 - 1, have + 1, a - 1, &io_error)) return give_up(io_error); }
    else if (!read_at(in0, have, 0, &io_error)) return give_up(io_error);
    if (kind == kFastq) {
      const int64_t c = (int64_t)h_count_nl(in0, have);
      { std::lock_guard<std::mutex> lk(m); nl[i] = c; }
      cv.notify_all();
    }
    // more of the file behind the piece, for the line / record that starts in it and ends after it
    auto more = [&]() -> int {  // 1: got some, 0: end of the file, -1: no room / error
      if (a + have >= len) return 0;
      const uint64_t room = cap - kRoom - have;
      if (room == 0) return -1;
      uint64_t n = len - (a + have);
      if (n > room) n = room;
      if (n > (256u << 10)) n = 256u << 10;
      if (!read_at(in0 + have, n, a + have, &io_error)) return -1;
      have += n;
      return 1;
    };
    const auto too_long = [&]() {
      if (!io_error.empty()) return give_up(io_error);
      return give_up("a record of more than " + std::to_string((unsigned long long)(cap - kRoom - raw)) +
                         " bytes past its piece does not fit the streaming pieces", MG_ERR_CAPACITY);
    };
    // the first line that starts in [a, b)
    uint64_t s = 0;
    const bool fresh = a == 0 || in0[-1] == '\n';
    if (!fresh) {
      const uint8_t* q = static_cast<const uint8_t*>(memchr(in0, '\n', (size_t)(b - a)));
      if (!q) return 0;  // one line runs through the whole piece: its owner is an earlier piece
      s = (uint64_t)(q - in0) + 1;
    }
    uint8_t* out = dst;
    if (kind == kSam) {
      CandIter<true> it(in0, &have, s);
      while (s < b - a) {
        // the line's fields = the gaps between its white-space bytes (str.split()); the line ends at its '\n' or with the file
        const uint8_t* lb = in0 + s;
        uint64_t prev = s;  // one past the last white-space byte seen (s: none yet)
        int nf = 0;
        uint64_t f9b = 0, f9e = 0, f10b = 0, f10e = 0, e = 0;
        bool eof = false;
        for (;;) {
          uint64_t c = it.next();
          if (c >= have) {
            const int r = more();
            if (r < 0) return too_long();
            if (r > 0) continue;
            eof = true;
            c = have;
          }
          if (!eof && !h_is_ws(in0[c])) continue;  // (a control character inside a field)
          if (c > prev) {
            if (nf == 9) { f9b = prev; f9e = c; }
            if (nf == 10) { f10b = prev; f10e = c; }
            ++nf;
          }
          prev = c + 1;
          if (eof || in0[c] == '\n') { e = c; break; }
        }
        const uint8_t* le = in0 + e;
        if (le > lb && *lb != '@' && nf >= 11) {
          const uint8_t *p9b = in0 + f9b, *p9e = in0 + f9e, *p10b = in0 + f10b, *p10e = in0 + f10e;
          const bool star = f9e - f9b == 1 && *p9b == '*';
          char num[24];
          const int nd = star ? 0 : snprintf(num, sizeof(num), "%llu", (unsigned long long)(f9e - f9b));
          // every part is written at or below where it came from as long as the rewritten SEQ ends no later than the original did
          // (it is longer only for a one-character SEQ: kRoom such lines per piece before this gives up)
          if (out + (p9b - lb) + (star ? 1 : 1 + nd) > p9e) return give_up("a piece of SAM text grew while it was thinned");
          memmove(out, lb, (size_t)(p9b - lb)); out += p9b - lb;
          if (star) *out++ = '*';
          else { *out++ = (uint8_t)MG_THIN_MARK; memcpy(out, num, (size_t)nd); out += nd; }
          memmove(out, p9e, (size_t)(p10b - p9e)); out += p10b - p9e;
          *out++ = '*';
          memmove(out, p10e, (size_t)(le - p10e)); out += le - p10e;
        } else {
          memmove(out, lb, (size_t)(le - lb)); out += le - lb;
        }
        if (!eof) *out++ = '\n';
        s = e + 1;
        if (eof) break;
      }
      return (int64_t)(out - dst);
    }
    // FASTQ: the index of the line at s = the newlines in front of it
    uint64_t before = 0;
    {
      std::unique_lock<std::mutex> lk(m);
      for (uint64_t j = 0; j < i; ++j) {
        cv.wait(lk, [&] { return dead || nl[j] >= 0; });
        if (dead) return -1;
        before += (uint64_t)nl[j];
      }
    }
    CandIter<false> it(in0, &have, s);
    // end of the line the scanner stands in: offset of its '\n', or of the end of the file (*eof); -1: no room / error
    auto line_end = [&](bool* eof) -> int64_t {
      *eof = false;
      for (;;) {
        const uint64_t c = it.next();
        if (c < have) return (int64_t)c;
        const int r = more();
        if (r < 0) return -1;
        if (r == 0) { *eof = true; return (int64_t)have; }
      }
    };
    uint64_t L = before + (fresh ? 0 : 1);
    while (s < b - a) {
      bool eof = false;
      if (L % 4 != 0) {  // a line of a record that an earlier piece owns
        const int64_t e = line_end(&eof);
        if (e < 0) return too_long();
        if (eof) break;
        s = (uint64_t)e + 1;
        ++L;
        continue;
      }
      uint64_t lb[4], le[4];
      int k = 0;
      uint64_t at = s;
      for (; k < 4; ++k) {
        const int64_t e = line_end(&eof);
        if (e < 0) return too_long();
        lb[k] = at; le[k] = (uint64_t)e;
        at = (uint64_t)e + 1;
        if (eof) { if (le[k] > lb[k]) ++k; break; }  // (nothing behind the last newline is not a line)
      }
      if (k < 4) {  // the end of the file inside a record: only blank lines may be left (the parser's rule)
        for (int j = 0; j < k; ++j) {
          const uint64_t n = le[j] - lb[j];
          if (!(n == 0 || (n == 1 && in0[lb[j]] == '\r')))
            return give_up("reads text: " + std::to_string((unsigned long long)(L + (uint64_t)k)) + " lines is not a whole number of 4-line records");
        }
        break;
      }
      auto stripped = [&](int j) { return le[j] > lb[j] && in0[le[j] - 1] == '\r' ? le[j] - 1 : le[j]; };
      if (stripped(0) == lb[0] || in0[lb[0]] != '@' || stripped(2) == lb[2] || in0[lb[2]] != '+')
        return give_up("reads file: malformed record " + std::to_string((unsigned long long)(L / 4)) + " (header / separator line)");
      *out++ = '>';
      *out++ = '\n';
      memmove(out, in0 + lb[1], (size_t)(le[1] - lb[1]));
      out += le[1] - lb[1];
      *out++ = '\n';
      s = at;
      L += 4;
      if (eof) break;
    }
    return (int64_t)(out - dst);
  }
};

static bool looks_gzip(int fd) {
  uint8_t h[2];
  return pread(fd, h, 2, 0) == 2 && h[0] == 0x1f && h[1] == 0x8b;
}

// ---------------------------------------------------------------------------------------------------------------------
// resources kept between calls (page-locking a slot costs milliseconds: the slots stay)
// ---------------------------------------------------------------------------------------------------------------------
struct StreamRes {
  hipStream_t copy = nullptr;
  std::vector<uint8_t*> slots;
  uint64_t slot_bytes = 0;
  std::vector<hipEvent_t> ev_slot;   // the DMA out of slot s has run
  hipEvent_t ev_h2d[3] = {nullptr, nullptr, nullptr};     // device buffer b holds its piece
  hipEvent_t ev_parsed[3] = {nullptr, nullptr, nullptr};  // ... and nothing reads it any more
};
static StreamRes g_res;

void stream_release_all() {
  StreamRes& r = g_res;
  if (r.copy) { (void)hipStreamSynchronize(r.copy); (void)hipStreamDestroy(r.copy); }
  for (uint8_t* p : r.slots) (void)hipHostFree(p);
  for (hipEvent_t e : r.ev_slot) (void)hipEventDestroy(e);
  for (int b = 0; b < 3; ++b) {
    if (r.ev_h2d[b]) (void)hipEventDestroy(r.ev_h2d[b]);
    if (r.ev_parsed[b]) (void)hipEventDestroy(r.ev_parsed[b]);
  }
  r = StreamRes();
}

static int ensure_res(uint64_t slot_bytes, size_t nslots) {
  StreamRes& r = g_res;
  if (!r.copy) MG_HIP(hipStreamCreateWithFlags(&r.copy, hipStreamNonBlocking));
  if (r.slot_bytes != slot_bytes) {
    for (uint8_t* p : r.slots) (void)hipHostFree(p);
    r.slots.clear();
    r.slot_bytes = slot_bytes;
  }
  while (r.slots.size() < nslots) {
    uint8_t* p = nullptr;
    MG_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), slot_bytes, hipHostMallocDefault));
    r.slots.push_back(p);
  }
  while (r.ev_slot.size() < nslots) {
    hipEvent_t e = nullptr;
    MG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    r.ev_slot.push_back(e);
  }
  for (int b = 0; b < 3; ++b) {
    if (!r.ev_h2d[b]) MG_HIP(hipEventCreateWithFlags(&r.ev_h2d[b], hipEventDisableTiming));
    if (!r.ev_parsed[b]) MG_HIP(hipEventCreateWithFlags(&r.ev_parsed[b], hipEventDisableTiming));
  }
  return MG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// the pipeline
// ---------------------------------------------------------------------------------------------------------------------
// consumer(d_text, nbytes, final, &consumed): the library-stream side of one piece; [consumed, nbytes) is carried.
using Consumer = std::function<int(const uint8_t*, uint64_t, bool, uint64_t*)>;

struct SlotState {
  std::mutex m;
  std::condition_variable cv;
  int64_t filled = -1;        // piece number the slot holds (ready for the DMA)
  int64_t dma_queued = -1;    // piece number whose DMA out of the slot has been queued (ev_slot recorded behind it)
  int64_t bytes = 0;
  bool last = false;
};

static int run_pipeline(Source& src, uint64_t chunk_bytes, int nthreads, const Consumer& consume, uint64_t* pieces_out) {
  Context& c = ctx();
  if (chunk_bytes < (1u << 16)) chunk_bytes = 1u << 16;
  chunk_bytes = (chunk_bytes + 4095) & ~4095ull;
  const uint64_t headroom = chunk_bytes / 4 > (4u << 20) ? chunk_bytes / 4 : (chunk_bytes < (4u << 20) ? chunk_bytes : (4u << 20));
  if (nthreads < 1) nthreads = 1;
  if (!src.parallel()) nthreads = 1;
  const size_t nslots = (size_t)nthreads + 2;
  MG_TRY(ensure_res(chunk_bytes, nslots));
  StreamRes& r = g_res;
  DevBuf dtext[3];
  for (int b = 0; b < 3; ++b) MG_TRY(dtext[b].alloc(headroom + chunk_bytes + 64));
  std::vector<std::unique_ptr<SlotState>> st(nslots);
  for (auto& p : st) p.reset(new SlotState());
  std::atomic<uint64_t> next_piece{0};
  std::atomic<bool> stop{false}, failed{false};
  std::atomic<int64_t> last_piece{-1};  // known once some reader has seen the end of the stream

  auto reader = [&]() {
    (void)hipSetDevice(c.device);
    for (;;) {
      const uint64_t i = next_piece.fetch_add(1);
      const int64_t lp = last_piece.load();
      if (stop.load() || (lp >= 0 && (int64_t)i > lp)) return;
      const size_t s = i % nslots;
      SlotState& ss = *st[s];
      if (i >= nslots) {  // the slot's previous piece must have left it
        std::unique_lock<std::mutex> lk(ss.m);
        ss.cv.wait(lk, [&] { return stop.load() || ss.dma_queued == (int64_t)(i - nslots); });
        if (stop.load()) return;
        lk.unlock();
        (void)hipEventSynchronize(r.ev_slot[s]);
      }
      bool last = false;
      const int64_t n = src.fill(i, r.slots[s], chunk_bytes, &last);
      {
        std::lock_guard<std::mutex> lk(ss.m);
        if (n < 0) { failed.store(true); ss.bytes = 0; ss.last = true; }
        else { ss.bytes = n; ss.last = last; }
        ss.filled = (int64_t)i;
      }
      if (last || n < 0) {
        int64_t expect = -1;
        last_piece.compare_exchange_strong(expect, (int64_t)i);
      }
      ss.cv.notify_all();
      if (n < 0) return;
    }
  };
  std::vector<std::thread> threads;
  for (int t = 0; t < nthreads; ++t) threads.emplace_back(reader);
  auto shut = [&]() {
    stop.store(true);
    for (auto& p : st) { std::lock_guard<std::mutex> lk(p->m); p->cv.notify_all(); }
    for (auto& t : threads) if (t.joinable()) t.join();
  };

  int rc = MG_OK;
  uint64_t carry = 0, npieces = 0;
  hipStream_t main_st = c.stream;
  // piece j's DMA is queued BEFORE piece j - 1 is parsed: the copy engine works while this thread waits in the parser
  auto queue_dma = [&](uint64_t j, int64_t* bytes, bool* last) -> int {
    const size_t s = j % nslots;
    const int b = (int)(j % 3);
    SlotState& ss = *st[s];
    {
      std::unique_lock<std::mutex> lk(ss.m);
      ss.cv.wait(lk, [&] { return ss.filled == (int64_t)j; });
      *bytes = ss.bytes;
      *last = ss.last;
    }
    if (failed.load()) return fail(src.error_code, "%s", src.error.empty() ? "reading the input failed" : src.error.c_str());
    if (j >= 3) MG_HIP(hipStreamWaitEvent(r.copy, r.ev_parsed[b], 0));  // the buffer's previous piece has been parsed
    if (*bytes > 0)
      MG_HIP(hipMemcpyAsync(dtext[b].as<uint8_t>() + headroom, r.slots[s], (size_t)*bytes, hipMemcpyHostToDevice, r.copy));
    MG_HIP(hipEventRecord(r.ev_slot[s], r.copy));
    MG_HIP(hipEventRecord(r.ev_h2d[b], r.copy));
    {
      std::lock_guard<std::mutex> lk(ss.m);
      ss.dma_queued = (int64_t)j;
    }
    ss.cv.notify_all();
    return MG_OK;
  };

  int64_t cur_bytes = 0, nxt_bytes = 0;
  bool cur_last = false, nxt_last = false;
  rc = queue_dma(0, &cur_bytes, &cur_last);
  for (uint64_t j = 0; rc == MG_OK; ++j) {
    if (!cur_last) rc = queue_dma(j + 1, &nxt_bytes, &nxt_last);
    if (rc != MG_OK) break;
    const int b = (int)(j % 3);
    rc = hipStreamWaitEvent(main_st, r.ev_h2d[b], 0) == hipSuccess ? MG_OK : fail(MG_ERR_HIP, "hipStreamWaitEvent failed");
    if (rc != MG_OK) break;
    const uint8_t* text = dtext[b].as<uint8_t>() + headroom - carry;
    const uint64_t nbytes = carry + (uint64_t)cur_bytes;
    uint64_t consumed = 0;
    rc = consume(text, nbytes, cur_last, &consumed);
    if (rc != MG_OK) break;
    ++npieces;
    if (cur_last) break;
    if (consumed > nbytes) consumed = nbytes;
    const uint64_t left = nbytes - consumed;
    if (left > headroom) {
      rc = fail(MG_ERR_CAPACITY, "a record of more than %llu bytes does not fit the streaming pieces", (unsigned long long)headroom);
      break;
    }
    if (left) {
      const int nb = (int)((j + 1) % 3);
      if (hipMemcpyAsync(dtext[nb].as<uint8_t>() + headroom - left, text + consumed, (size_t)left, hipMemcpyDeviceToDevice, main_st) != hipSuccess) {
        rc = fail(MG_ERR_HIP, "carry copy failed");
        break;
      }
    }
    carry = left;
    if (hipEventRecord(r.ev_parsed[b], main_st) != hipSuccess) { rc = fail(MG_ERR_HIP, "hipEventRecord failed"); break; }
    cur_bytes = nxt_bytes;
    cur_last = nxt_last;
  }
  shut();
  (void)hipStreamSynchronize(r.copy);
  if (rc != MG_OK) (void)hipStreamSynchronize(main_st);  // (what was queued may still read the buffers released below)
  if (pieces_out) *pieces_out = npieces;
  return rc;
}

// *chunk_bytes: 0 = the default of the source's kind (32 MB: tools/stream_probe.py — 3.2 GB of FASTQ, 8 readers: 16 MB pieces
// 0.089 s, 32 MB 0.071 s, 64 MB 0.067 s warm, but 0.105 against 0.122 s with the page-locked slots still to be allocated, which is
// what a one-shot command line pays; BGZF 16 MB: many inflating threads, each with a slot).
// thin_kind: -1, or what a plain file read from its first to its last byte may be thinned as (ThinSource::Kind); *thinned tells.
static int open_source(const char* path, uint64_t offset, uint64_t length, uint64_t* chunk_bytes, std::unique_ptr<Source>* out, bool* gz,
                       int thin_kind = -1, bool* thinned = nullptr) {
  if (thinned) *thinned = false;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return fail(MG_ERR_ARG, "cannot open %s: %s", path, strerror(errno));
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); return fail(MG_ERR_ARG, "cannot stat %s", path); }
  const uint64_t fsize = (uint64_t)sb.st_size;
  *gz = looks_gzip(fd);
  if (*gz) {
    if (offset || length) { close(fd); return fail(MG_ERR_ARG, "a byte range of a gzip file cannot be streamed"); }
    std::unique_ptr<BgzfSource> bz(new BgzfSource());
    bz->fd = fd;
    uint64_t bc = *chunk_bytes ? *chunk_bytes : (16ull << 20);
    if (bc < (1u << 16)) bc = 1u << 16;
    if (bz->index(fsize, bc)) { *chunk_bytes = bc; *out = std::move(bz); return MG_OK; }
    bz->fd = -1;  // not BGZF: one deflate stream
    if (!*chunk_bytes) *chunk_bytes = 32ull << 20;
    unsigned hw = std::thread::hardware_concurrency();
    int threads = (int)(hw == 0 ? 4 : (hw > 64 ? 64 : hw));
    if (dbg("gzip_threads") > 0) threads = (int)dbg("gzip_threads");
    if (threads > 1) {
      // ... entered in the middle by every core the box has (knob gzip_threads = 1: zlib, one thread, as rounds 2-3)
      std::string err;
      uint64_t pc = 1ull << 20;
      if (threads > 32) threads = 32;  // (two chunks per thread in flight, ~15 MB each inflated: a gigabyte of host memory at 32)
      if (dbg("pgzip_chunk") > 0) pc = (uint64_t)dbg("pgzip_chunk");
      std::unique_ptr<ParallelGzipSource> pg(new ParallelGzipSource());
      pg->g = PGzip::open(fd, true, fsize, threads, pc, &err);
      if (!pg->g) { close(fd); return fail(MG_ERR_ARG, "%s: %s", path, err.c_str()); }
      *out = std::move(pg);
      return MG_OK;
    }
    std::unique_ptr<GzipSource> g(new GzipSource());
    g->fd = fd;
    *out = std::move(g);
    return MG_OK;
  }
  if (offset > fsize) { close(fd); return fail(MG_ERR_ARG, "offset beyond the end of %s", path); }
  if (!*chunk_bytes) *chunk_bytes = 32ull << 20;
  if (thin_kind >= 0 && thinned && offset == 0 && length == 0) {
    if (dbg("stream_thin")) {  // opt-in: see the note at ThinSource
      uint64_t cb = *chunk_bytes < (1u << 16) ? (1u << 16) : *chunk_bytes;  // (the pipeline's own rounding of the slot size)
      cb = (cb + 4095) & ~4095ull;
      std::unique_ptr<ThinSource> t(new ThinSource());
      t->fd = fd;
      t->kind = thin_kind == 0 ? ThinSource::kFastq : ThinSource::kSam;
      t->setup(fsize, cb);
      *thinned = true;
      *out = std::move(t);
      return MG_OK;
    }
  }
  std::unique_ptr<PlainSource> p(new PlainSource());
  p->fd = fd;
  p->off = offset;
  p->len = length ? (offset + length > fsize ? fsize - offset : length) : fsize - offset;
  *out = std::move(p);
  return MG_OK;
}

// Thinning is host work per byte (a reader does ~4 GB/s of it against ~12 GB/s of plain reads): twice the readers, when the
// box has them and nobody fixed the number.
static int thin_threads(int n) {
  if (dbg("stream_threads") > 0) return n;
  const unsigned hw = std::thread::hardware_concurrency();
  const int want = n * 2;
  return hw >= (unsigned)want * 2 ? want : n;
}

static int default_threads() {
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 4;
  if (dbg("stream_threads") > 0) return (int)dbg("stream_threads");
  return (int)(hw > 8 ? 8 : hw);
}

// A `.gz` file (gzip or BGZF) whose compressed bytes go to the device and are inflated there (mg_inflate.hip), unless
// mg_inflate_config turned that off: -> fd >= 0 and its size; -1: not such a file (the host readers take it)
// (... or too large a file: the device inflater holds the WHOLE compressed file in device memory beside the pieces of text in
// flight; a file above half of what is free — or above the knob inflate_dev_max_bytes — goes through the host inflater, whose
// memory is bounded by its pieces.)
static int open_for_device_inflate(const char* path, uint64_t offset, uint64_t length, uint64_t* fsize) {
  if (!inflate_dev_enabled() || offset || length) return -1;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return -1;
  struct stat sb;
  if (fstat(fd, &sb) != 0 || !looks_gzip(fd)) { close(fd); return -1; }
  *fsize = (uint64_t)sb.st_size;
  size_t free_b = 0, total_b = 0;
  uint64_t cap = hipMemGetInfo(&free_b, &total_b) == hipSuccess ? (uint64_t)free_b / 2 : 0;
  if (dbg("inflate_dev_max_bytes") > 0 && (uint64_t)dbg("inflate_dev_max_bytes") < cap) cap = (uint64_t)dbg("inflate_dev_max_bytes");
  if (*fsize > cap) { close(fd); return -1; }
  return fd;
}

}  // namespace mg

using namespace mg;

extern "C" {

int mg_sketch_stream_add_file(mg_sketch_stream* ss, const char* path, int format, uint64_t offset, uint64_t length,
                              uint64_t chunk_bytes, int nthreads) {
  MG_REQUIRE_READY();
  if (!ss || !path) return fail(MG_ERR_ARG, "null argument");
  if (format < 0 || format > 2) return fail(MG_ERR_ARG, "format must be 0 (fastq), 1 (single-line fasta) or 2 (fasta)");
  {
    uint64_t fsize = 0;
    const int gfd = open_for_device_inflate(path, offset, length, &fsize);
    if (gfd >= 0) {
      Consumer consume_gz = [&](const uint8_t* d_text, uint64_t nbytes, bool final, uint64_t* consumed) -> int {
        mg_reads* rd = nullptr;
        MG_TRY(mg_reads_parse_prefix_dev(d_text, nbytes, format, final ? 1 : 0, consumed, &rd));
        const uint8_t* d_b = nullptr;
        const uint64_t* d_o = nullptr;
        int rc = mg_reads_device_ptrs(rd, &d_b, &d_o);
        if (rc == MG_OK) rc = mg_sketch_stream_add_dev(ss, d_b, d_o, mg_reads_count(rd), mg_reads_nbases(rd));
        mg_reads_free(rd);
        return rc;
      };
      bool started = false;
      const int rc = inflate_file_pipeline(gfd, fsize, consume_gz, &started);
      close(gfd);
      if (!(rc == MG_ERR_NOMEM && !started)) return rc;
      // (no room on the device for the compressed file after all, and nothing consumed yet: the host inflater takes it)
    }
  }
  std::unique_ptr<Source> src;
  bool gz = false;
  bool thinned = false;
  const bool auto_threads = nthreads <= 0;
  MG_TRY(open_source(path, offset, length, &chunk_bytes, &src, &gz, format == 0 ? (int)ThinSource::kFastq : -1, &thinned));
  if (nthreads <= 0) nthreads = default_threads();
  if (gz && src->parallel()) {  // BGZF: inflating is the work — every core the box has
    unsigned hw = std::thread::hardware_concurrency();
    if (dbg("stream_threads") <= 0 && hw > (unsigned)nthreads) nthreads = (int)(hw > 32 ? 32 : hw);
  }
  if (thinned && auto_threads) nthreads = thin_threads(nthreads);
  const int dev_format = thinned ? 1 : format;  // (a thinned FASTQ piece is single-line FASTA: ">", the sequence line)
  Consumer consume = [&](const uint8_t* d_text, uint64_t nbytes, bool final, uint64_t* consumed) -> int {
    mg_reads* rd = nullptr;
    MG_TRY(mg_reads_parse_prefix_dev(d_text, nbytes, dev_format, final ? 1 : 0, consumed, &rd));
    const uint8_t* d_b = nullptr;
    const uint64_t* d_o = nullptr;
    int rc = mg_reads_device_ptrs(rd, &d_b, &d_o);
    if (rc == MG_OK) rc = mg_sketch_stream_add_dev(ss, d_b, d_o, mg_reads_count(rd), mg_reads_nbases(rd));
    mg_reads_free(rd);  // (stream-ordered: the hashing kernel queued above still reads it; the pool hands it out behind that)
    return rc;
  };
  return run_pipeline(*src, chunk_bytes, nthreads, consume, nullptr);
}

// SAM (paf = 0) or PAF text file -> alignment records on the device, through the same pipeline: every piece is cut at its
// last newline, tokenised (mg_sam_tokenize_dev's rules; the previous retained QNAME carried from piece to piece so that
// the new-read bit is the whole file's) and its records appended.  Replaces the per-line Python of map_and_process,
// scripts/map_and_profile.py:201-217.  A line the reference cannot parse: MG_ERR_ARG with err_kind (the caller streams
// the file through the host tokeniser, which raises what the reference raises).
int mg_sam_stream_file(const char* path, int paf, const mg_acc_index* ix, uint64_t offset, uint64_t length,
                       uint64_t chunk_bytes, int nthreads, mg_sam_batch** out, int* err_kind, uint64_t* err_line) {
  MG_REQUIRE_READY();
  if (!path || !ix || !out) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  if (err_kind) *err_kind = 0;
  if (err_line) *err_line = 0;
  std::unique_ptr<Source> src;
  bool gz = false;
  bool thinned = false;
  const bool auto_threads = nthreads <= 0;
  uint64_t gsize = 0;
  int gfd = open_for_device_inflate(path, offset, length, &gsize);
  if (gfd >= 0) {  // (room for the compressed file on the device? probed here, so that the host inflater can still take the file)
    DevBuf probe;
    if (probe.alloc(((gsize + 3) & ~3ull) + 64) != MG_OK) { close(gfd); gfd = -1; }
  }
  if (gfd < 0) MG_TRY(open_source(path, offset, length, &chunk_bytes, &src, &gz, paf ? -1 : (int)ThinSource::kSam, &thinned));
  if (nthreads <= 0) nthreads = default_threads();
  if (thinned && auto_threads) nthreads = thin_threads(nthreads);
  std::vector<std::unique_ptr<mg_sam_batch>> parts;
  std::string prev;
  uint64_t total = 0;
  Consumer consume = [&](const uint8_t* d_text, uint64_t nbytes, bool final, uint64_t* consumed) -> int {
    mg_sam_batch* b = nullptr;
    MG_TRY(aln_tokenize_prefix_dev(d_text, nbytes, ix, prev.c_str(), paf != 0, final, consumed, &b, err_kind, err_line, thinned));
    prev = b->last_qname;
    total += b->nrecs;
    parts.emplace_back(b);
    return MG_OK;
  };
  if (gfd >= 0) {
    const int rc = inflate_file_pipeline(gfd, gsize, consume, nullptr);
    close(gfd);
    MG_TRY(rc);
  } else {
    MG_TRY(run_pipeline(*src, chunk_bytes, nthreads, consume, nullptr));
  }
  std::unique_ptr<mg_sam_batch> all(new mg_sam_batch());
  all->last_qname = prev;
  all->nrecs = total;
  if (parts.size() == 1) {
    all->recs = std::move(parts[0]->recs);
  } else {
    MG_TRY(all->recs.alloc((total + 1) * sizeof(mg_aln_rec)));
    uint64_t at = 0;
    hipStream_t st = ctx().stream;
    for (auto& p : parts) {
      if (p->nrecs)
        MG_HIP(hipMemcpyAsync(all->recs.as<mg_aln_rec>() + at, p->recs.p, p->nrecs * sizeof(mg_aln_rec), hipMemcpyDeviceToDevice, st));
      at += p->nrecs;
    }
    MG_HIP(hipStreamSynchronize(st));
  }
  *out = all.release();
  return MG_OK;
}

// What the file readers hand to the device for a plain FASTQ (kind 0) or SAM (kind 1) file, written to out_path instead: the
// pieces of `piece_bytes` slots thinned by nthreads readers, in order.  Host code only (no device, no mg_init): the thinning can be
// checked where there is no GPU.
int mg_stream_thin_file(const char* path, int kind, uint64_t piece_bytes, int nthreads, const char* out_path) {
  if (!path || !out_path) return fail(MG_ERR_ARG, "null argument");
  if (kind != 0 && kind != 1) return fail(MG_ERR_ARG, "kind must be 0 (fastq) or 1 (sam)");
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return fail(MG_ERR_ARG, "cannot open %s: %s", path, strerror(errno));
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); return fail(MG_ERR_ARG, "cannot stat %s", path); }
  uint64_t cb = piece_bytes ? piece_bytes : (32ull << 20);
  if (cb < (1u << 16)) cb = 1u << 16;
  cb = (cb + 4095) & ~4095ull;
  ThinSource t;
  t.fd = fd;
  t.kind = kind == 0 ? ThinSource::kFastq : ThinSource::kSam;
  t.setup((uint64_t)sb.st_size, cb);
  std::vector<std::vector<uint8_t>> outs(t.npieces);
  std::atomic<uint64_t> next{0};
  std::atomic<bool> failed{false};
  auto work = [&]() {
    std::vector<uint8_t> buf(cb);
    for (;;) {
      const uint64_t i = next.fetch_add(1);
      if (i >= t.npieces) return;
      bool last = false;
      const int64_t n = t.fill(i, buf.data(), cb, &last);
      if (n < 0) { failed.store(true); return; }
      outs[i].assign(buf.begin(), buf.begin() + n);
    }
  };
  {
    if (nthreads < 1) nthreads = 1;
    std::vector<std::thread> th;
    for (int k = 1; k < nthreads; ++k) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
  }
  if (failed.load()) return fail(t.error_code, "%s", t.error.empty() ? "thinning failed" : t.error.c_str());
  FILE* fo = fopen(out_path, "wb");
  if (!fo) return fail(MG_ERR_ARG, "cannot open %s: %s", out_path, strerror(errno));
  bool ok = true;
  for (auto& o : outs) ok = ok && (o.empty() || fwrite(o.data(), 1, o.size(), fo) == o.size());
  ok = (fclose(fo) == 0) && ok;
  return ok ? MG_OK : fail(MG_ERR_ARG, "writing %s failed", out_path);
}

}  // extern "C"
