// mg_ingest.hip — device-side ingest: text already in HBM -> the structures the hot
// path consumes.
//
//   FASTQ / single-line FASTA text  -> bases u8[] + offsets u64[R+1]          (mg_reads_parse_dev)
//     replaces the reads parsing inside kmc (scripts/select_db.py:45-52)
//   SAM text                        -> 16-byte alignment records              (mg_sam_tokenize_dev)
//     replaces the per-line Python of map_and_process: '@' / short-line / unmapped filter
//     (scripts/map_and_profile.py:202-213), parse_flag (:104-111), the CIGAR walk of filter_line
//     (:88-95), RNAME -> accession row (:217) and the `read != prev_read` test (:220)
//
// Both start from a line index (positions of '\n'), built with one counting pass, one scan and one
// marking pass; everything after is one thread per line.  Byte work, HBM-bound; no rocPRIM in the
// counting passes besides the exclusive scans.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "mg_internal.h"

namespace mg {

constexpr int kIB = 256;          // threads per block
constexpr int kBytesPerThread = 16;

// ---------------------------------------------------------------------------------------------
// line index
// ---------------------------------------------------------------------------------------------
// Thread g of the line-index kernels looks at the 16-byte ALIGNED window of memory number g counted from the aligned address
// at or below `text`: text offsets [16 g - mis, 16 g - mis + 16), mis = address of text modulo 16.  A piece of a streamed file
// begins wherever the carried record put it (mg_stream.hip), so `text` is rarely aligned itself; with windows fixed to the
// text's own start every thread of such a piece took the byte-by-byte path below.
__device__ __forceinline__ int64_t nl_window(const uint8_t* text, uint64_t g) {
  return (int64_t)(g * kBytesPerThread) - (int64_t)(reinterpret_cast<uintptr_t>(text) & 15);
}

__device__ __forceinline__ uint32_t nl_mask16(const uint8_t* __restrict__ text, uint64_t nbytes, int64_t base) {
  // bit j set <=> text[base + j] == '\n' (offsets outside [0, nbytes) never)
  uint32_t m = 0;
  if (base >= 0 && (uint64_t)base + kBytesPerThread <= nbytes) {
    const uint4 v = *reinterpret_cast<const uint4*>(text + base);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) m |= (((w[q] >> (8 * b)) & 0xffu) == 0x0au ? 1u : 0u) << (4 * q + b);
  } else {
    for (int j = 0; j < kBytesPerThread; ++j) {
      const int64_t o = base + j;
      if (o >= 0 && (uint64_t)o < nbytes && text[o] == '\n') m |= 1u << j;
    }
  }
  return m;
}

__global__ __launch_bounds__(kIB) void k_count_newlines(const uint8_t* __restrict__ text, uint64_t nbytes,
                                                        uint32_t* __restrict__ blk_count) {
  __shared__ uint32_t wsum[kIB / 64];
  const int64_t base = nl_window(text, (uint64_t)blockIdx.x * kIB + threadIdx.x);
  uint32_t c = base < (int64_t)nbytes ? __popc(nl_mask16(text, nbytes, base)) : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// line_end[i] = byte offset of the i-th '\n'
__global__ __launch_bounds__(kIB) void k_mark_newlines(const uint8_t* __restrict__ text, uint64_t nbytes,
                                                       const uint64_t* __restrict__ blk_base,
                                                       uint64_t* __restrict__ line_end) {
  __shared__ uint32_t wsum[kIB / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = nl_window(text, (uint64_t)blockIdx.x * kIB + threadIdx.x);
  const uint32_t m = base < (int64_t)nbytes ? nl_mask16(text, nbytes, base) : 0;
  const uint32_t c = __popc(m);
  uint32_t inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t p = __shfl_up(inc, o, 64);
    if (lane >= o) inc += p;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = 0;
  for (int w = 0; w < wave; ++w) before += wsum[w];
  uint64_t at = blk_base[blockIdx.x] + before + inc - c;
  uint32_t mm = m;
  while (mm) {
    const int j = __ffs(mm) - 1;
    line_end[at++] = (uint64_t)(base + j);
    mm &= mm - 1;
  }
}

// ---------------------------------------------------------------------------------------------
// FASTQ / FASTA
// ---------------------------------------------------------------------------------------------
// Line l spans [l == 0 ? 0 : line_end[l-1] + 1, line_end[l]) ; a trailing '\r' is not part of the line.
__device__ __forceinline__ void line_span(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_end,
                                          uint64_t l, uint64_t& beg, uint64_t& end) {
  beg = l == 0 ? 0 : line_end[l - 1] + 1;
  end = line_end[l];
  if (end > beg && text[end - 1] == '\r') --end;
}

// lines_per_rec = 4 (FASTQ) or 2 (single-line FASTA); sequence = line 1 of each record.
// err[0] = min record index with a malformed header (UINT64_MAX if none)
__global__ void k_read_lengths(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_end, uint64_t nrec,
                               int lines_per_rec, uint8_t head, uint32_t* __restrict__ lens,
                               unsigned long long* __restrict__ err) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; r < nrec; r += stride) {
    uint64_t hb, he, sb, se;
    line_span(text, line_end, r * lines_per_rec, hb, he);
    line_span(text, line_end, r * lines_per_rec + 1, sb, se);
    bool bad = he == hb || text[hb] != head;
    if (lines_per_rec == 4) {
      uint64_t pb, pe;
      line_span(text, line_end, r * 4 + 2, pb, pe);
      bad |= pe == pb || text[pb] != '+';
    }
    if (bad) atomicMin(err, (unsigned long long)r);
    lens[r] = (uint32_t)(se - sb);
  }
}

// One wavefront per read: copy its sequence line into the compact base buffer.
__global__ __launch_bounds__(256) void k_gather_reads(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_end,
                                                      uint64_t nrec, int lines_per_rec, const uint64_t* __restrict__ offs,
                                                      uint8_t* __restrict__ bases) {
  const int lane = threadIdx.x & 63;
  uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (; r < nrec; r += nw) {
    const uint64_t sb = line_end[r * lines_per_rec] + 1;
    const uint64_t o = offs[r], n = offs[r + 1] - o;
    for (uint64_t i = lane; i < n; i += 64) bases[o + i] = text[sb + i];
  }
}

// ---- FASTA with sequences over any number of lines (format 2) ----
// The host parser's rules (metalign_amd/formats.py::read_sequences): a line that starts with '>' opens a record;
// every other line after the first header is stripped of white space at both ends and appended; lines before the
// first header are ignored.
__device__ __forceinline__ bool is_space(uint8_t ch) {
  return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n' || ch == '\v' || ch == '\f';
}
__device__ __forceinline__ void stripped_span(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_end,
                                              uint64_t l, uint64_t& beg, uint64_t& end) {
  beg = l == 0 ? 0 : line_end[l - 1] + 1;
  end = line_end[l];
  while (end > beg && is_space(text[end - 1])) --end;
  while (beg < end && is_space(text[beg])) ++beg;
}

__global__ void k_fasta_lines(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_end, uint64_t nlines,
                              uint32_t* __restrict__ flags, uint32_t* __restrict__ lens) {
  uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; l < nlines; l += stride) {
    const uint64_t raw = l == 0 ? 0 : line_end[l - 1] + 1;
    const bool head = raw < line_end[l] && text[raw] == '>';
    uint64_t b, e;
    stripped_span(text, line_end, l, b, e);
    flags[l] = head ? 1u : 0u;
    lens[l] = head ? 0u : (uint32_t)(e - b);
  }
}

// 1 + the index of the last header line (0: none) — where a piece of a streamed FASTA file is cut
__global__ void k_fasta_last_header(const uint32_t* __restrict__ flags, uint64_t nlines, unsigned long long* __restrict__ out) {
  uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long best = 0;
  for (; l < nlines; l += stride)
    if (flags[l]) best = l + 1;
  if (best) atomicMax(out, best);
}

// sequence lines in front of the first header belong to no record
__global__ void k_fasta_orphans(const uint32_t* __restrict__ flags, const uint64_t* __restrict__ rank, uint64_t nlines,
                                uint32_t* __restrict__ lens) {
  uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; l < nlines; l += stride)
    if (!flags[l] && rank[l] == 0) lens[l] = 0;
}

__global__ void k_fasta_offsets(const uint32_t* __restrict__ flags, const uint64_t* __restrict__ rank,
                                const uint64_t* __restrict__ pos, uint64_t nlines, uint64_t nrec, uint64_t nbases,
                                uint64_t* __restrict__ offsets) {
  uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  if (l == 0) offsets[nrec] = nbases;
  for (; l < nlines; l += stride)
    if (flags[l]) offsets[rank[l]] = pos[l];
}

// One wavefront per line: its stripped bytes go to the record's place in the compact base buffer.
__global__ __launch_bounds__(256) void k_fasta_gather(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_end,
                                                      const uint32_t* __restrict__ lens, const uint64_t* __restrict__ pos,
                                                      uint64_t nlines, uint8_t* __restrict__ bases) {
  const int lane = threadIdx.x & 63;
  uint64_t l = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (; l < nlines; l += nw) {
    const uint32_t n = lens[l];
    if (!n) continue;
    uint64_t b, e;
    stripped_span(text, line_end, l, b, e);
    const uint64_t o = pos[l];
    for (uint32_t i = lane; i < n; i += 64) bases[o + i] = text[b + i];
  }
}

// ---------------------------------------------------------------------------------------------
// SAM
// ---------------------------------------------------------------------------------------------
enum SamErr : uint32_t { kErrNone = 0, kErrKey = 1, kErrIndex = 2, kErrValue = 3, kErrZeroDiv = 4, kErrOverflow = 5 };

__device__ __forceinline__ bool is_ws(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); }
__device__ __forceinline__ bool is_alpha(uint8_t c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }

// First whitespace byte in [p, end), or end.  A SAM line is mostly SEQ and QUAL (2 x 150 of ~370 bytes) that only have
// to be stepped over: eight bytes per load once p is aligned, with a SWAR test for "some byte below 0x21" (the classic
// has-less-than; bytes >= 0x80 never match); a candidate is then checked against the exact whitespace set of
// str.split().  One byte per load made the one-thread-per-line parser latency-bound at 106 GB/s of text.
__device__ __forceinline__ uint64_t scan_to_ws(const uint8_t* __restrict__ text, uint64_t p, uint64_t end) {
  while (p < end && ((reinterpret_cast<uintptr_t>(text) + p) & 7u)) {
    if (is_ws(text[p])) return p;
    ++p;
  }
  while (p + 8 <= end) {
    const uint64_t x = *reinterpret_cast<const uint64_t*>(text + p);
    uint64_t m = (x - 0x2121212121212121ull) & ~x & 0x8080808080808080ull;
    while (m) {  // candidates in ascending order (a borrow can only flag bytes ABOVE a true one: checked like the rest)
      const uint32_t b = (uint32_t)__builtin_ctzll(m) >> 3;
      if (is_ws((uint8_t)(x >> (8 * b)))) return p + b;
      m &= m - 1;
    }
    p += 8;
  }
  while (p < end && !is_ws(text[p])) ++p;
  return p;
}

__device__ __forceinline__ uint64_t fnv1a(const uint8_t* p, uint32_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint32_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

struct AccTable {
  const uint64_t* slot_hash;   // 0 = empty
  const uint32_t* slot_row;
  uint64_t mask;               // slots - 1
  const uint8_t* names;        // concatenated accession strings
  const uint64_t* name_off;    // [nacc + 1]
};

__device__ __forceinline__ int64_t acc_lookup(const AccTable& t, const uint8_t* s, uint32_t n) {
  uint64_t h = fnv1a(s, n);
  if (h == 0) h = 1;
  for (uint64_t p = h & t.mask;; p = (p + 1) & t.mask) {
    const uint64_t sh = t.slot_hash[p];
    if (sh == 0) return -1;
    if (sh == h) {
      const uint32_t row = t.slot_row[p];
      const uint64_t b = t.name_off[row], e = t.name_off[row + 1];
      if (e - b == n) {
        bool same = true;
        for (uint32_t i = 0; i < n && same; ++i) same = t.names[b + i] == s[i];
        if (same) return row;
      }
    }
  }
}

constexpr uint32_t kQnameInline = 256;  // (SAM: QNAME is at most 254 characters)
struct LineOut {
  mg_aln_rec rec;      // ref_new without the new-read bit
  uint64_t qbeg;       // QNAME span in the text
  uint32_t qlen;
  uint32_t retained;
};

// One thread per line: everything except the new-read bit.  err: [0] = first failing line (atomicMin),
// kinds[line] holds the failure kind for the host to look up.
__global__ void k_sam_parse(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_end, uint64_t nlines,
                            AccTable acc, LineOut* __restrict__ out, uint32_t* __restrict__ retained,
                            unsigned long long* __restrict__ err, uint32_t* __restrict__ err_kind, uint32_t thin) {
  uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; l < nlines; l += stride) {
    LineOut o;
    o.retained = 0;
    o.qbeg = 0; o.qlen = 0;
    o.rec.ref_new = o.rec.matched = o.rec.total = o.rec.flag_len = 0;
    const uint64_t beg = l == 0 ? 0 : line_end[l - 1] + 1;
    const uint64_t end = line_end[l];
    uint32_t kind = kErrNone;
    if (end > beg && text[beg] != '@') {  // line.startswith('@') is tested BEFORE strip() (:204)
      // fields of line.strip().split(): maximal runs of non-whitespace; only the first 12 matter
      uint64_t fb[12], fe[12];
      int nf = 0;
      uint64_t p = beg;
      while (p < end && nf < 12) {
        while (p < end && is_ws(text[p])) ++p;
        if (p >= end) break;
        fb[nf] = p;
        p = scan_to_ws(text, p, end);
        fe[nf] = p;
        ++nf;
      }
      if (nf >= 6) {
        // FLAG: int(splits[1])
        uint64_t flag = 0;
        bool ok = fe[1] > fb[1], neg = false;
        uint64_t q = fb[1];
        if (ok && (text[q] == '+' || text[q] == '-')) { neg = text[q] == '-'; ++q; }
        ok = ok && q < fe[1];
        for (; ok && q < fe[1]; ++q) {
          const uint8_t ch = text[q];
          if (ch < '0' || ch > '9') ok = false; else flag = (flag * 10 + (ch - '0')) & 0xffffffffffffull;
        }
        if (neg) flag = 0 - flag;  // Python's & on a negative int sees two's complement bits
        if (!ok) {
          kind = kErrValue;
        } else {
          const bool star = fe[5] - fb[5] == 1 && text[fb[5]] == '*';
          if (!((flag & 4) || star)) {
            // retained line (:211-213)
            const int64_t row = acc_lookup(acc, text + fb[2], (uint32_t)(fe[2] - fb[2]));
            if (row < 0) kind = kErrKey;
            uint64_t matched = 0, total = 0, cur = 0;
            for (uint64_t c = fb[5]; c < fe[5] && kind == kErrNone; ++c) {
              const uint8_t ch = text[c];
              if (is_alpha(ch)) {
                if (ch == 'M') matched += cur;
                total += cur;
                cur = 0;
              } else if (ch >= '0' && ch <= '9') {
                cur = cur * 10 + (ch - '0');
              } else {
                kind = kErrValue;  // int(ch) on a non-digit, e.g. '=' (:90-93)
              }
            }
            if (kind == kErrNone && nf < 12) kind = kErrIndex;  // splits[11] (:97)
            if (kind == kErrNone) {  // int(splits[11][5:])
              uint64_t t = fb[11] + 5;
              bool tok = t < fe[11];
              if (tok && (text[t] == '+' || text[t] == '-')) ++t;
              tok = tok && t < fe[11];
              for (; tok && t < fe[11]; ++t) tok = text[t] >= '0' && text[t] <= '9';
              if (!tok) kind = kErrValue;
            }
            if (kind == kErrNone && total == 0) kind = kErrZeroDiv;
            uint64_t slen = (fe[9] - fb[9] == 1 && text[fb[9]] == '*') ? 0 : fe[9] - fb[9];
            if (thin && text[fb[9]] == MG_THIN_MARK) {  // a line thinned by the file reader (mg_stream.hip): SEQ = mark + its length in decimal
              slen = 0;
              for (uint64_t c = fb[9] + 1; c < fe[9]; ++c) slen = slen * 10 + (uint64_t)(text[c] - '0');
            }
            if (kind == kErrNone && (slen > MG_REC_MAX_SEQLEN || matched > 0xffffffffull || total > 0xffffffffull))
              kind = kErrOverflow;
            if (kind == kErrNone) {
              o.retained = 1;
              o.rec.ref_new = (uint32_t)row;
              o.rec.matched = (uint32_t)matched;
              o.rec.total = (uint32_t)total;
              o.rec.flag_len = (uint32_t)(flag & MG_REC_FLAG_MASK) | ((uint32_t)slen << MG_REC_LEN_SHIFT);
              o.qbeg = fb[0];
              o.qlen = (uint32_t)(fe[0] - fb[0]);
            }
          }
        }
      }
    }
    if (kind != kErrNone) {
      atomicMin(err, (unsigned long long)l);
      err_kind[l] = kind;
    }
    out[l] = o;
    retained[l] = o.retained;
  }
}

// PAF replay adaptor, one thread per line: minimap2 PAF -> the same records, by the rules of
// metalign_amd/map_and_profile.py::tokenise_paf (the reference itself reads SAM, scripts/map_and_profile.py:87,97,
// 142-144,211,217: this is an adaptor, not a parity path).  Fields are split on TAB only (the trailing CR / LF
// stripped); lines with fewer than 12 fields are skipped; RNAME <- column 6; FLAG <- 16 if strand '-', + 256 if the
// LAST `tp:A:` tag is S; with a `cg:Z:` tag (the last one) matched = sum of M, total = all ops + the query bases outside
// [qstart, qend); without, matched = column 10 and total = column 2; len(SEQ) <- query length, 0 for secondaries.
__device__ __forceinline__ bool paf_int(const uint8_t* t, uint64_t b, uint64_t e, int64_t* v) {  // int(field): [+-]digits
  bool neg = false;
  if (b < e && (t[b] == '+' || t[b] == '-')) { neg = t[b] == '-'; ++b; }
  if (b >= e) return false;
  int64_t x = 0;
  for (; b < e; ++b) {
    if (t[b] < '0' || t[b] > '9') return false;
    x = x * 10 + (t[b] - '0');
    if (x > (1ll << 40)) return false;  // (reported as overflow by the caller's range checks anyway)
  }
  *v = neg ? -x : x;
  return true;
}

__global__ void k_paf_parse(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_end, uint64_t nlines,
                            AccTable acc, LineOut* __restrict__ out, uint32_t* __restrict__ retained,
                            unsigned long long* __restrict__ err, uint32_t* __restrict__ err_kind) {
  uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; l < nlines; l += stride) {
    LineOut o;
    o.retained = 0;
    o.qbeg = 0; o.qlen = 0;
    o.rec.ref_new = o.rec.matched = o.rec.total = o.rec.flag_len = 0;
    const uint64_t beg = l == 0 ? 0 : line_end[l - 1] + 1;
    uint64_t end = line_end[l];
    while (end > beg && (text[end - 1] == '\r' || text[end - 1] == '\n')) --end;  // rstrip('\r\n')
    uint32_t kind = kErrNone;
    // the first 12 fields; the tags after them are scanned in place
    uint64_t fb[12], fe[12];
    int nf = 0;
    uint64_t p = beg;
    while (nf < 12) {
      fb[nf] = p;
      while (p < end && text[p] != '\t') ++p;
      fe[nf] = p;
      ++nf;
      if (p >= end) break;
      ++p;  // the TAB
    }
    const bool more = nf == 12 && fe[11] < end;  // a TAB follows the 12th field: tags
    if (nf == 12) {
      int64_t qlen = 0, qs = 0, qe = 0, nmatch = 0;
      if (!paf_int(text, fb[1], fe[1], &qlen) || !paf_int(text, fb[2], fe[2], &qs) || !paf_int(text, fb[3], fe[3], &qe))
        kind = kErrValue;
      uint32_t flag = (fe[4] - fb[4] == 1 && text[fb[4]] == '-') ? 16u : 0u;
      // tags: the LAST tp:A: and the LAST cg:Z: win (a dict built left to right)
      bool secondary = false, have_cg = false;
      uint64_t cgb = 0, cge = 0;
      if (more) {
        uint64_t t = fe[11] + 1;
        while (t <= end) {
          uint64_t te = t;
          while (te < end && text[te] != '\t') ++te;
          if (te - t > 5) {
            if (text[t] == 't' && text[t + 1] == 'p' && text[t + 2] == ':' && text[t + 3] == 'A')
              secondary = te - t == 6 && text[t + 5] == 'S';
            if (text[t] == 'c' && text[t + 1] == 'g' && text[t + 2] == ':' && text[t + 3] == 'Z') { have_cg = true; cgb = t + 5; cge = te; }
          }
          if (te >= end) break;
          t = te + 1;
        }
      }
      if (secondary) flag |= 256u;
      int64_t matched = 0, total = 0;
      if (kind == kErrNone) {
        if (have_cg) {
          int64_t cur = 0;
          for (uint64_t c = cgb; c < cge; ++c) {
            const uint8_t ch = text[c];
            if (ch >= '0' && ch <= '9') { cur = cur * 10 + (ch - '0'); if (cur > (1ll << 40)) cur = 1ll << 40; }
            else { if (ch == 'M') matched += cur; total += cur; cur = 0; }
          }
          total += qlen - (qe - qs);
        } else {
          if (!paf_int(text, fb[9], fe[9], &nmatch)) kind = kErrValue;
          matched = nmatch;
          total = qlen;
        }
      }
      if (kind == kErrNone && total == 0) kind = kErrZeroDiv;
      const int64_t slen = secondary ? 0 : qlen;
      int64_t row = -1;
      if (kind == kErrNone) {
        row = acc_lookup(acc, text + fb[5], (uint32_t)(fe[5] - fb[5]));
        if (row < 0) kind = kErrKey;
      }
      if (kind == kErrNone && (slen < 0 || slen > (int64_t)MG_REC_MAX_SEQLEN || matched < 0 || matched > 0xffffffffll ||
                               total < 0 || total > 0xffffffffll))
        kind = kErrOverflow;
      if (kind == kErrNone) {
        o.retained = 1;
        o.rec.ref_new = (uint32_t)row;
        o.rec.matched = (uint32_t)matched;
        o.rec.total = (uint32_t)total;
        o.rec.flag_len = flag | ((uint32_t)slen << MG_REC_LEN_SHIFT);
        o.qbeg = fb[0];
        o.qlen = (uint32_t)(fe[0] - fb[0]);
      }
    }
    if (kind != kErrNone) {
      atomicMin(err, (unsigned long long)l);
      err_kind[l] = kind;
    }
    out[l] = o;
    retained[l] = o.retained;
  }
}

// rank = exclusive prefix of retained flags: list the retained lines in order
__global__ void k_sam_list(const uint32_t* __restrict__ retained, const uint64_t* __restrict__ rank, uint64_t nlines,
                           uint64_t* __restrict__ ret_line) {
  uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; l < nlines; l += stride)
    if (retained[l]) ret_line[rank[l]] = l;
}

// record r = retained line ret_line[r]; new-read bit = QNAME differs from the previous retained line's
// (the first one is compared with prev_qname, the QNAME carried over from the previous chunk; empty = none)
__global__ void k_sam_emit(const uint8_t* __restrict__ text, const LineOut* __restrict__ lines,
                           const uint64_t* __restrict__ ret_line, uint64_t nret, const uint8_t* __restrict__ prev_qname,
                           uint32_t prev_len, mg_aln_rec* __restrict__ recs) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; r < nret; r += stride) {
    const LineOut cur = lines[ret_line[r]];
    const uint8_t* pq;
    uint32_t pl;
    if (r == 0) { pq = prev_qname; pl = prev_len; }
    else { const LineOut pv = lines[ret_line[r - 1]]; pq = text + pv.qbeg; pl = pv.qlen; }
    bool same = pl == cur.qlen;
    for (uint32_t i = 0; same && i < cur.qlen; ++i) same = pq[i] == text[cur.qbeg + i];
    mg_aln_rec rec = cur.rec;
    if (!same) rec.ref_new |= MG_REC_NEW_BIT;
    recs[r] = rec;
  }
}

// Shared: build the line index of a text buffer.  If the text does not end in '\n' the last partial
// line is terminated virtually.  -> d_line_end (scratch "ing_lines"), *nlines.
// last_end (optional): the end of the last line, brought back in the same round trip as the marks' completion (a piece of a
// stream is consumed up to there).  Two host synchronisations per call: the count, then the marks.
static int build_line_index(const uint8_t* d_text, uint64_t nbytes, uint64_t** d_line_end, uint64_t* nlines,
                            bool* virtual_last, bool allow_virtual = true, uint64_t* last_end = nullptr) {
  Context& c = ctx();
  hipStream_t st = c.stream;
  *nlines = 0;
  *virtual_last = false;
  if (nbytes == 0) { *d_line_end = (uint64_t*)scratch("ing_lines", 16); return *d_line_end ? MG_OK : MG_ERR_NOMEM; }
  const uint64_t per_block = (uint64_t)kIB * kBytesPerThread;
  // (the threads' windows are aligned in MEMORY: up to 15 bytes of slack in front of the text)
  const uint64_t nblocks = (nbytes + (kBytesPerThread - 1) + per_block - 1) / per_block;
  if (nblocks > 0x7fffffffull) return fail(MG_ERR_ARG, "text too large for one ingest call");
  uint32_t* d_cnt = (uint32_t*)scratch("ing_blk_cnt", nblocks * sizeof(uint32_t));
  uint64_t* d_base = (uint64_t*)scratch("ing_blk_base", (nblocks + 1) * sizeof(uint64_t));
  if (!d_cnt || !d_base) return MG_ERR_NOMEM;
  uint64_t total = 0;
  uint8_t last = 0;
  {
    ProfScope ps("ingest_lines");
    hipLaunchKernelGGL(k_count_newlines, dim3((unsigned)nblocks), dim3(kIB), 0, st, d_text, nbytes, d_cnt);
    MG_HIP(hipGetLastError());
    uint64_t* pin = host_words();
    MG_HIP(hipMemcpyAsync(pin + 13, d_text + nbytes - 1, 1, hipMemcpyDeviceToHost, st));  // (rides on the scan's synchronisation)
    MG_TRY(exclusive_sum_u32_to_u64(d_cnt, d_base, nblocks, &total));
    last = *reinterpret_cast<const volatile uint8_t*>(pin + 13);
    *virtual_last = allow_virtual && last != '\n';  // (a piece of a stream: what follows the last newline is not a line yet)
    uint64_t* d_le = (uint64_t*)scratch("ing_lines", (total + 2) * sizeof(uint64_t));
    if (!d_le) return MG_ERR_NOMEM;
    hipLaunchKernelGGL(k_mark_newlines, dim3((unsigned)nblocks), dim3(kIB), 0, st, d_text, nbytes, d_base, d_le);
    MG_HIP(hipGetLastError());
    if (*virtual_last) MG_HIP(hipMemcpyAsync(d_le + total, &nbytes, sizeof(uint64_t), hipMemcpyHostToDevice, st));
    const uint64_t nl = total + (*virtual_last ? 1 : 0);
    if (last_end && nl) MG_HIP(hipMemcpyAsync(pin + 12, d_le + (nl - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st));
    if (last_end) *last_end = nl ? pin[12] : 0;
    *d_line_end = d_le;
  }
  *nlines = total + (*virtual_last ? 1 : 0);
  return MG_OK;
}

}  // namespace mg

using namespace mg;

// (mg_reads, mg_acc_index, mg_sam_batch: mg_internal.h)

extern "C" {

// final: the text is a whole file (or the last piece of one).  Otherwise it is a piece of a stream that BEGINS on a record
// boundary: the whole records in it are parsed and *consumed = the byte where the first incomplete record begins (the
// caller carries [consumed, nbytes) to the front of the next piece).
static int reads_parse_impl(const uint8_t* d_text, uint64_t nbytes, int format, bool final, uint64_t* consumed, mg_reads** out) {
  MG_REQUIRE_READY();
  if (!out) return fail(MG_ERR_ARG, "null out handle");
  *out = nullptr;
  if (consumed) *consumed = final ? nbytes : 0;
  if (format < 0 || format > 2) return fail(MG_ERR_ARG, "format must be 0 (fastq), 1 (single-line fasta) or 2 (fasta)");
  if (nbytes > 0 && !d_text) return fail(MG_ERR_ARG, "null device text");
  Context& c = ctx();
  hipStream_t st = c.stream;
  std::unique_ptr<mg_reads> rd(new mg_reads());
  uint64_t* d_le = nullptr;
  uint64_t nlines = 0;
  bool vlast = false;
  MG_TRY(build_line_index(d_text, nbytes, &d_le, &nlines, &vlast, final));
  if (format == 2) {  // sequences over any number of lines
    uint32_t* d_flag = (uint32_t*)scratch("ing_fa_flag", (nlines + 1) * sizeof(uint32_t));
    uint32_t* d_llen = (uint32_t*)scratch("ing_fa_len", (nlines + 1) * sizeof(uint32_t));
    uint64_t* d_rank = (uint64_t*)scratch("ing_fa_rank", (nlines + 1) * sizeof(uint64_t));
    uint64_t* d_pos = (uint64_t*)scratch("ing_fa_pos", (nlines + 1) * sizeof(uint64_t));
    if (!d_flag || !d_llen || !d_rank || !d_pos) return MG_ERR_NOMEM;
    uint64_t nrec = 0, nbases = 0;
    ProfScope ps("ingest_reads");
    if (nlines) {
      const unsigned grid = grid_for(nlines, 256, (unsigned)c.num_cus * 8);
      hipLaunchKernelGGL(k_fasta_lines, dim3(grid), dim3(256), 0, st, d_text, d_le, nlines, d_flag, d_llen);
      MG_HIP(hipGetLastError());
      if (!final) {
        // a record ends where the next header line begins: everything from the LAST header line on waits for the next piece
        unsigned long long* d_last = (unsigned long long*)scratch("ing_fa_last", 2 * sizeof(unsigned long long));
        if (!d_last) return MG_ERR_NOMEM;
        MG_HIP(hipMemsetAsync(d_last, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_fasta_last_header, dim3(grid), dim3(256), 0, st, d_flag, nlines, d_last);
        MG_HIP(hipGetLastError());
        uint64_t* pin = host_words();
        MG_HIP(hipMemcpyAsync(pin + 12, d_last, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        MG_HIP(hipStreamSynchronize(st));
        const uint64_t last = pin[12];  // 1 + line index, 0 = none
        // (no header in the piece at all: a record longer than the piece, or junk in front of the first header —
        // nothing is consumed and the caller, whose carry then outgrows its room, takes the whole-file path)
        const uint64_t use = last ? last - 1 : 0;
        const uint64_t cut_line = use;  // consumed = the start of line `use` (= the end of line use - 1, + 1)
        if (cut_line == 0) {
          if (consumed) *consumed = 0;
        } else {
          MG_HIP(hipMemcpyAsync(pin + 12, d_le + (cut_line - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, st));
          MG_HIP(hipStreamSynchronize(st));
          if (consumed) *consumed = pin[12] + 1;
        }
        nlines = use;
      }
    }
    if (nlines) {
      const unsigned grid = grid_for(nlines, 256, (unsigned)c.num_cus * 8);
      MG_TRY(exclusive_sum_u32_to_u64(d_flag, d_rank, nlines, &nrec));
      hipLaunchKernelGGL(k_fasta_orphans, dim3(grid), dim3(256), 0, st, d_flag, d_rank, nlines, d_llen);
      MG_HIP(hipGetLastError());
      MG_TRY(exclusive_sum_u32_to_u64(d_llen, d_pos, nlines, &nbases));
    }
    rd->nreads = nrec;
    rd->nbases = nbases;
    MG_TRY(rd->offsets.alloc((nrec + 2) * sizeof(uint64_t)));
    MG_TRY(rd->bases.alloc(nbases + 16));
    if (nrec == 0) {
      MG_HIP(hipMemsetAsync(rd->offsets.p, 0, 2 * sizeof(uint64_t), st));
    } else {
      hipLaunchKernelGGL(k_fasta_offsets, dim3(grid_for(nlines, 256, (unsigned)c.num_cus * 8)), dim3(256), 0, st, d_flag,
                         d_rank, d_pos, nlines, nrec, nbases, rd->offsets.as<uint64_t>());
      hipLaunchKernelGGL(k_fasta_gather, dim3(grid_for(nlines, 4, (unsigned)c.num_cus * 16)), dim3(256), 0, st, d_text, d_le,
                         d_llen, d_pos, nlines, rd->bases.as<uint8_t>());
      MG_HIP(hipGetLastError());
    }
    *out = rd.release();
    return MG_OK;
  }
  const int lpr = format == 0 ? 4 : 2;
  const uint64_t nrec = nlines / lpr;
  const uint64_t left = final ? nlines % lpr : 0;  // (a piece: the lines of an incomplete record are the next piece's)
  if (!final && consumed) {
    *consumed = 0;
    if (nrec) {
      uint64_t* pin = host_words();
      MG_HIP(hipMemcpyAsync(pin + 12, d_le + (nrec * lpr - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      MG_HIP(hipStreamSynchronize(st));
      *consumed = pin[12] + 1;
    }
  }
  if (left) {  // only blank lines may follow the last whole record
    uint64_t le[5] = {0, 0, 0, 0, 0};
    const uint64_t first = nrec * lpr;  // first leftover line
    const uint64_t from = first ? first - 1 : 0;
    MG_HIP(hipMemcpyAsync(le, d_le + from, (nlines - from) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st));
    bool blank = true;
    for (uint64_t l = first; l < nlines; ++l) {
      const uint64_t end = le[l - from];
      const uint64_t beg = l == 0 ? 0 : le[l - 1 - from] + 1;
      blank = blank && end <= beg + 1;  // empty, or a lone '\r'
    }
    if (!blank) return fail(MG_ERR_ARG, "reads text: %llu lines is not a whole number of %d-line records",
                            (unsigned long long)nlines, lpr);
  }
  rd->nreads = nrec;
  MG_TRY(rd->offsets.alloc((nrec + 2) * sizeof(uint64_t)));
  if (nrec == 0) {
    MG_HIP(hipMemsetAsync(rd->offsets.p, 0, 2 * sizeof(uint64_t), st));
    MG_TRY(rd->bases.alloc(16));
    *out = rd.release();
    return MG_OK;
  }
  uint32_t* d_len = (uint32_t*)scratch("ing_len", nrec * sizeof(uint32_t));
  unsigned long long* d_err = (unsigned long long*)scratch("ing_err", sizeof(unsigned long long));
  if (!d_len || !d_err) return MG_ERR_NOMEM;
  MG_HIP(hipMemsetAsync(d_err, 0xff, sizeof(unsigned long long), st));
  {
    ProfScope ps("ingest_reads");
    hipLaunchKernelGGL(k_read_lengths, dim3(grid_for(nrec, 256, (unsigned)c.num_cus * 8)), dim3(256), 0, st, d_text, d_le,
                       nrec, lpr, (uint8_t)(format == 0 ? '@' : '>'), d_len, d_err);
    MG_HIP(hipGetLastError());
    MG_TRY(exclusive_sum_u32_to_u64(d_len, rd->offsets.as<uint64_t>(), nrec, &rd->nbases));
    unsigned long long h_err = 0;
    MG_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st));
    if (h_err != ~0ull) return fail(MG_ERR_ARG, "reads file: malformed record %llu (header / separator line)", h_err);
    MG_TRY(rd->bases.alloc(rd->nbases + 16));
    hipLaunchKernelGGL(k_gather_reads, dim3(grid_for(nrec, 4, (unsigned)c.num_cus * 16)), dim3(256), 0, st, d_text, d_le,
                       nrec, lpr, rd->offsets.as<uint64_t>(), rd->bases.as<uint8_t>());
    MG_HIP(hipGetLastError());
  }
  *out = rd.release();
  return MG_OK;
}

int mg_reads_parse_dev(const uint8_t* d_text, uint64_t nbytes, int format, mg_reads** out) {
  return reads_parse_impl(d_text, nbytes, format, true, nullptr, out);
}

int mg_reads_parse_prefix_dev(const uint8_t* d_text, uint64_t nbytes, int format, int final, uint64_t* consumed, mg_reads** out) {
  if (!consumed) return fail(MG_ERR_ARG, "null consumed");
  return reads_parse_impl(d_text, nbytes, format, final != 0, consumed, out);
}

int mg_reads_parse(const uint8_t* text, uint64_t nbytes, int format, mg_reads** out) {
  MG_REQUIRE_READY();
  DevBuf d_text;
  MG_TRY(d_text.alloc(nbytes + 16));
  MG_TRY(mg_memcpy_h2d(d_text.p, text, nbytes));
  return mg_reads_parse_dev(d_text.as<uint8_t>(), nbytes, format, out);
}

uint64_t mg_reads_count(const mg_reads* r) { return r ? r->nreads : 0; }
uint64_t mg_reads_nbases(const mg_reads* r) { return r ? r->nbases : 0; }

int mg_reads_device_ptrs(const mg_reads* r, const uint8_t** d_bases, const uint64_t** d_offsets) {
  if (!r) return fail(MG_ERR_ARG, "null reads");
  if (d_bases) *d_bases = r->bases.as<uint8_t>();
  if (d_offsets) *d_offsets = r->offsets.as<uint64_t>();
  return MG_OK;
}

int mg_reads_download(const mg_reads* r, uint8_t* bases, uint64_t* offsets) {
  MG_REQUIRE_READY();
  if (!r) return fail(MG_ERR_ARG, "null reads");
  if (bases && r->nbases) MG_TRY(mg_memcpy_d2h(bases, r->bases.p, r->nbases));
  if (offsets) MG_TRY(mg_memcpy_d2h(offsets, r->offsets.p, (r->nreads + 1) * sizeof(uint64_t)));
  return MG_OK;
}

void mg_reads_free(mg_reads* r) { delete r; }

// ---- accession table ----
static uint64_t host_fnv1a(const char* p, uint64_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint64_t i = 0; i < n; ++i) { h ^= (uint8_t)p[i]; h *= 0x100000001b3ull; }
  return h ? h : 1;
}

int mg_acc_index_build(const char* names, const uint64_t* name_offsets, uint32_t nacc, mg_acc_index** out) {
  MG_REQUIRE_READY();
  if (!out || (nacc && (!names || !name_offsets))) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  std::unique_ptr<mg_acc_index> ix(new mg_acc_index());
  uint64_t slots = 16;
  while (slots < 2ull * nacc + 2) slots <<= 1;
  std::vector<uint64_t> sh(slots, 0);
  std::vector<uint32_t> sr(slots, 0);
  for (uint32_t i = 0; i < nacc; ++i) {
    const uint64_t b = name_offsets[i], e = name_offsets[i + 1];
    const uint64_t h = host_fnv1a(names + b, e - b);
    uint64_t p = h & (slots - 1);
    bool dup = false;
    while (sh[p] != 0) {
      if (sh[p] == h) {
        const uint64_t ob = name_offsets[sr[p]], oe = name_offsets[sr[p] + 1];
        if (oe - ob == e - b && memcmp(names + ob, names + b, e - b) == 0) { dup = true; break; }
      }
      p = (p + 1) & (slots - 1);
    }
    // a repeated accession keeps its LAST row, as the reference's dict assignment does (:77)
    sh[p] = h;
    sr[p] = i;
    (void)dup;
  }
  const uint64_t nb = nacc ? name_offsets[nacc] : 0;
  MG_TRY(ix->slot_hash.alloc(slots * sizeof(uint64_t)));
  MG_TRY(ix->slot_row.alloc(slots * sizeof(uint32_t)));
  MG_TRY(ix->names.alloc(nb + 16));
  MG_TRY(ix->name_off.alloc((nacc + 1ull) * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_h2d(ix->slot_hash.p, sh.data(), slots * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_h2d(ix->slot_row.p, sr.data(), slots * sizeof(uint32_t)));
  if (nb) MG_TRY(mg_memcpy_h2d(ix->names.p, names, nb));
  if (nacc) MG_TRY(mg_memcpy_h2d(ix->name_off.p, name_offsets, (nacc + 1ull) * sizeof(uint64_t)));
  else { uint64_t z = 0; MG_TRY(mg_memcpy_h2d(ix->name_off.p, &z, sizeof(z))); }
  ix->slots = slots;
  ix->nacc = nacc;
  *out = ix.release();
  return MG_OK;
}

void mg_acc_index_free(mg_acc_index* ix) { delete ix; }

static int aln_tokenize_dev(const uint8_t* d_text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname,
                            bool paf, mg_sam_batch** out, int* err_kind, uint64_t* err_line) {
  return mg::aln_tokenize_prefix_dev(d_text, nbytes, ix, prev_qname, paf, true, nullptr, out, err_kind, err_line);
}

}  // extern "C"

namespace mg {
// out: [0..8) QNAME offset, [8..12) length, [16..) its first kQnameInline bytes — of the last retained line.
__global__ void k_sam_last_qname(const uint8_t* __restrict__ text, const LineOut* __restrict__ lines, const uint64_t* __restrict__ list,
                                 uint64_t nret, uint8_t* __restrict__ out) {
  const LineOut& lo = lines[list[nret - 1]];
  if (threadIdx.x == 0) {
    *reinterpret_cast<uint64_t*>(out) = lo.qbeg;
    *reinterpret_cast<uint32_t*>(out + 8) = lo.qlen;
  }
  for (uint32_t i = threadIdx.x; i < lo.qlen && i < kQnameInline; i += blockDim.x) out[16 + i] = text[lo.qbeg + i];
}
}  // namespace mg

// final = false: a PIECE of the text that begins at a line start; the complete lines in it are tokenised and *consumed =
// the byte after the last newline (what follows is carried to the next piece by the caller, mg_stream.hip).
int mg::aln_tokenize_prefix_dev(const uint8_t* d_text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname,
                                bool paf, bool final, uint64_t* consumed, mg_sam_batch** out, int* err_kind, uint64_t* err_line,
                                bool thin) {
  MG_REQUIRE_READY();
  if (!out || !ix) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  if (err_kind) *err_kind = 0;
  if (err_line) *err_line = 0;
  Context& c = ctx();
  hipStream_t st = c.stream;
  std::unique_ptr<mg_sam_batch> sb(new mg_sam_batch());
  if (prev_qname) sb->last_qname = prev_qname;
  uint64_t* d_le = nullptr;
  uint64_t nlines = 0;
  bool vlast = false;
  uint64_t last_end = 0;
  MG_TRY(build_line_index(d_text, nbytes, &d_le, &nlines, &vlast, final, (!final && consumed) ? &last_end : nullptr));
  if (consumed) *consumed = final ? nbytes : 0;
  if (nlines == 0) {
    MG_TRY(sb->recs.alloc(16));
    *out = sb.release();
    return MG_OK;
  }
  if (!final && consumed) *consumed = last_end + 1;
  LineOut* d_lines = (LineOut*)scratch("sam_lines", nlines * sizeof(LineOut));
  uint32_t* d_ret = (uint32_t*)scratch("sam_ret", nlines * sizeof(uint32_t));
  uint64_t* d_rank = (uint64_t*)scratch("sam_rank", (nlines + 1) * sizeof(uint64_t));
  uint32_t* d_kind = (uint32_t*)scratch("sam_kind", nlines * sizeof(uint32_t));
  unsigned long long* d_err = (unsigned long long*)scratch("ing_err", sizeof(unsigned long long));
  const size_t plen = prev_qname ? strlen(prev_qname) : 0;
  uint8_t* d_prev = (uint8_t*)scratch("sam_prev", plen + 16);
  if (!d_lines || !d_ret || !d_rank || !d_kind || !d_err || !d_prev) return MG_ERR_NOMEM;
  MG_HIP(hipMemsetAsync(d_err, 0xff, sizeof(unsigned long long), st));
  if (plen) MG_HIP(hipMemcpyAsync(d_prev, prev_qname, plen, hipMemcpyHostToDevice, st));
  AccTable at{ix->slot_hash.as<uint64_t>(), ix->slot_row.as<uint32_t>(), ix->slots - 1, ix->names.as<uint8_t>(),
              ix->name_off.as<uint64_t>()};
  uint64_t nret = 0;
  {
    ProfScope ps("ingest_sam");
    if (paf)
      hipLaunchKernelGGL(k_paf_parse, dim3(grid_for(nlines, 256, (unsigned)c.num_cus * 16)), dim3(256), 0, st, d_text, d_le,
                         nlines, at, d_lines, d_ret, d_err, d_kind);
    else
      hipLaunchKernelGGL(k_sam_parse, dim3(grid_for(nlines, 256, (unsigned)c.num_cus * 16)), dim3(256), 0, st, d_text, d_le,
                         nlines, at, d_lines, d_ret, d_err, d_kind, thin ? 1u : 0u);
    MG_HIP(hipGetLastError());
    uint64_t* pin = host_words();
    MG_HIP(hipMemcpyAsync(pin + 14, d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));  // (rides on the scan's synchronisation)
    MG_TRY(exclusive_sum_u32_to_u64(d_ret, d_rank, nlines, &nret));
    const unsigned long long h_err = *reinterpret_cast<const volatile unsigned long long*>(pin + 14);
    if (h_err != ~0ull) {
      uint32_t kind = 0;
      MG_HIP(hipMemcpyAsync(&kind, d_kind + h_err, sizeof(kind), hipMemcpyDeviceToHost, st));
      MG_HIP(hipStreamSynchronize(st));
      if (err_kind) *err_kind = (int)kind;
      if (err_line) *err_line = h_err;
      return fail(MG_ERR_ARG, "%s line %llu: parse error kind %u", paf ? "PAF" : "SAM", h_err, kind);
    }
    MG_TRY(sb->recs.alloc((nret + 1) * sizeof(mg_aln_rec)));
    if (nret) {
      uint64_t* d_list = (uint64_t*)scratch("sam_list", nret * sizeof(uint64_t));
      if (!d_list) return MG_ERR_NOMEM;
      hipLaunchKernelGGL(k_sam_list, dim3(grid_for(nlines, 256, (unsigned)c.num_cus * 16)), dim3(256), 0, st, d_ret, d_rank,
                         nlines, d_list);
      hipLaunchKernelGGL(k_sam_emit, dim3(grid_for(nret, 256, (unsigned)c.num_cus * 16)), dim3(256), 0, st, d_text, d_lines,
                         d_list, nret, d_prev, (uint32_t)plen, sb->recs.as<mg_aln_rec>());
      MG_HIP(hipGetLastError());
      // QNAME of the last retained line, for the next chunk: its span and its first kQnameInline bytes in ONE round trip
      // (three dependent ones — line number, span, bytes — were a quarter of a streamed piece's host time)
      uint8_t* d_q = (uint8_t*)scratch("sam_lastq", 16 + kQnameInline);
      if (!d_q) return MG_ERR_NOMEM;
      hipLaunchKernelGGL(k_sam_last_qname, dim3(1), dim3(64), 0, st, d_text, d_lines, d_list, nret, d_q);
      MG_HIP(hipGetLastError());
      std::vector<uint8_t> hq(16 + kQnameInline);
      MG_HIP(hipMemcpyAsync(hq.data(), d_q, hq.size(), hipMemcpyDeviceToHost, st));
      MG_HIP(hipStreamSynchronize(st));
      uint64_t qbeg;
      uint32_t qlen;
      memcpy(&qbeg, hq.data(), 8);
      memcpy(&qlen, hq.data() + 8, 4);
      sb->last_qname.assign(reinterpret_cast<const char*>(hq.data() + 16), qlen < kQnameInline ? qlen : kQnameInline);
      if (qlen > kQnameInline) {  // (a QNAME longer than the SAM format allows: the rest in a second trip)
        sb->last_qname.resize(qlen);
        MG_HIP(hipMemcpyAsync(&sb->last_qname[0], d_text + qbeg, qlen, hipMemcpyDeviceToHost, st));
        MG_HIP(hipStreamSynchronize(st));
      }
    }
  }
  sb->nrecs = nret;
  *out = sb.release();
  return MG_OK;
}

extern "C" {

int mg_sam_tokenize_dev(const uint8_t* d_text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname,
                        mg_sam_batch** out, int* err_kind, uint64_t* err_line) {
  return aln_tokenize_dev(d_text, nbytes, ix, prev_qname, false, out, err_kind, err_line);
}

int mg_paf_tokenize_dev(const uint8_t* d_text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname,
                        mg_sam_batch** out, int* err_kind, uint64_t* err_line) {
  return aln_tokenize_dev(d_text, nbytes, ix, prev_qname, true, out, err_kind, err_line);
}

int mg_paf_tokenize(const uint8_t* text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname,
                    mg_sam_batch** out, int* err_kind, uint64_t* err_line) {
  MG_REQUIRE_READY();
  DevBuf d_text;
  MG_TRY(d_text.alloc(nbytes + 16));
  MG_TRY(mg_memcpy_h2d(d_text.p, text, nbytes));
  return mg_paf_tokenize_dev(d_text.as<uint8_t>(), nbytes, ix, prev_qname, out, err_kind, err_line);
}

int mg_sam_tokenize(const uint8_t* text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname,
                    mg_sam_batch** out, int* err_kind, uint64_t* err_line) {
  MG_REQUIRE_READY();
  DevBuf d_text;
  MG_TRY(d_text.alloc(nbytes + 16));
  MG_TRY(mg_memcpy_h2d(d_text.p, text, nbytes));
  return mg_sam_tokenize_dev(d_text.as<uint8_t>(), nbytes, ix, prev_qname, out, err_kind, err_line);
}

uint64_t mg_sam_batch_count(const mg_sam_batch* b) { return b ? b->nrecs : 0; }
const char* mg_sam_batch_last_qname(const mg_sam_batch* b) { return b ? b->last_qname.c_str() : ""; }

int mg_sam_batch_device_ptr(const mg_sam_batch* b, const mg_aln_rec** d_recs) {
  if (!b || !d_recs) return fail(MG_ERR_ARG, "null argument");
  *d_recs = b->recs.as<mg_aln_rec>();
  return MG_OK;
}

int mg_sam_batch_download(const mg_sam_batch* b, mg_aln_rec* recs) {
  MG_REQUIRE_READY();
  if (!b) return fail(MG_ERR_ARG, "null batch");
  if (b->nrecs) MG_TRY(mg_memcpy_d2h(recs, b->recs.p, b->nrecs * sizeof(mg_aln_rec)));
  return MG_OK;
}

void mg_sam_batch_free(mg_sam_batch* b) { delete b; }

}  // extern "C"
