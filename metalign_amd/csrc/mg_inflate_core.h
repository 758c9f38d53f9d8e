// mg_inflate_core.h — DEFLATE (RFC 1951) inside gzip (RFC 1952), decoded by ONE 64-lane wavefront per job.
//
// The reference takes `.fq.gz` reads as ordinary input (scripts/select_db.py:146-148; kmc inflates them itself, :50-52) and
// `zcat`s the selected genomes (:101-105).  mg_inflate.hip moves that to the device: compressed bytes cross PCIe, the text is
// born in HBM.  This header is the decoder itself, written so that the SAME code compiles for the host (tests/host_inflate_check.cpp
// runs it lane by lane against zlib where there is no GPU) and for gfx950:
//
//   * everything a Huffman decoder does serially is WAVE-UNIFORM code: every lane executes it, the values that come back from LDS
//     or memory go through readfirstlane (MGI_UNI), so the bit buffer, the table look-ups' results, lengths and distances live
//     in SGPRs and the loop runs on the scalar unit; LDS writes of such code are made by one lane (Exec::leader());
//   * what is parallel is written per LANE (Exec::lanes): counting code lengths, ranking the symbols of a length (canonical
//     codes), filling the look-up tables, and EMITTING: the uniform loop only queues up to 64 symbols (literal | length,
//     distance) of at most 4096 output bytes; then lane l produces output bytes l, l + 64, ... of the batch — the owner of a
//     byte is a popcount over a bitmap of the symbols' first bytes, a back-reference is followed (through other symbols of
//     the same batch if need be) to a literal of the batch or to a byte that was in memory before the batch — so no byte of
//     a batch depends on a store of the same batch and every load is independent of every other;
//   * a job that starts in the middle of a deflate stream (mg_inflate.hip finds block starts speculatively) does not know the
//     32 KB in front of it: its output is 16-bit symbols, a byte or 0x8000 + i = "byte i of that window" (the pugz scheme).
//
// Phases are separated by Exec::sync() (a workgroup barrier of the single wavefront); the host runs a phase for lane 0..63
// in turn, which is the same thing as long as the lanes of a phase do not depend on each other — they are written not to.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MGI_HD __host__ __device__ inline
#else
#define MGI_HD inline
#endif
#define MGI_HDI MGI_HD __attribute__((always_inline))
#if defined(__clang__)
#define MGI_UNROLL _Pragma("unroll")
#else
#define MGI_UNROLL
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define MGI_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define MGI_CLOCK() ((uint64_t)__builtin_readcyclecounter())
#else
#define MGI_UNI(x) ((uint32_t)(x))
#define MGI_CLOCK() ((uint64_t)0)
#endif

namespace mgi {

constexpr int LB = 10;                   // primary bits of the literal/length table
constexpr int DB = 8;                    // ... of the distance table
constexpr uint32_t kBatchSyms = 256;     // symbols queued per batch
constexpr uint32_t kBatchBytes = 4096;   // output bytes per batch (bitmap of 64 x 64 bits)
constexpr uint32_t kWindow = 32768;
constexpr uint32_t kInWords = 64;        // compressed words staged in LDS for the window decode
constexpr uint32_t kStepBits = 256;      // bit positions one window step looks at (four per lane)
constexpr uint32_t kStepSyms = 64;       // ... and the symbols it takes at most
constexpr uint32_t kEmitWays = 4;        // output bytes a lane makes side by side in the emission (8: the same time — it is instructions, not waiting)

enum : uint32_t { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_LONG = 3, K_BAD = 4, K_DIST = 5 };

// how a job ended (Result::status)
enum : uint32_t {
  ST_STOP = 1,        // at a block boundary at or behind Job::stop_bit
  ST_END = 2,         // the gzip stream ended: input exhausted at a member boundary, or something that is no member follows
  ST_MEMBER = 3,      // F_ONE_MEMBER: the member's final block and trailer were read
  ST_NEED_MORE = 4,   // ran behind the bytes that are there so far (Job input not final)
  ST_ERR = 16,        // everything from here on is an error
  ST_TRUNC = 16,      // the stream ends inside a member
  ST_BAD_BLOCK = 17,  // block type 3
  ST_BAD_STORED = 18, // LEN / NLEN of a stored block
  ST_BAD_CODES = 19,  // code lengths: over-subscribed, incomplete, no end-of-block, a repeat with nothing to repeat
  ST_BAD_SYMBOL = 20, // an unassigned code, a length / distance symbol that does not exist
  ST_BAD_DIST = 21,   // a distance behind the start of the member (or of the known window)
  ST_BAD_HEADER = 22, // gzip header
  ST_EVENTS_FULL = 23,
};
enum : uint32_t {
  F_HEADER = 1,        // a gzip member header stands at start_bit (a byte position)
  F_ONE_MEMBER = 2,    // stop behind the trailer of the member (BGZF); crc / isize go to the Result
  F_MEMBER_START = 4,  // nothing of the member lies in front of start_bit: a distance behind it is an error
  F_COUNT_ONLY = 8,    // decode and count, write nothing
};

struct Job {
  uint64_t start_bit, stop_bit;  // ST_STOP at the first block boundary >= stop_bit
  uint64_t out_off, out_cap;     // elements (bytes, or 16-bit symbols) in the output buffer; beyond out_cap: counted, not written
  uint32_t flags, pad;
};
struct Result {
  uint64_t end_bit, out_count;
  uint32_t status, overflow;
  uint32_t crc, isize;           // F_ONE_MEMBER: the trailer
  uint32_t nblocks, nevents;
  // where the job's time went (shader clocks): block headers + tables, symbol decoding, emission, the tail row; batches, windows
  uint64_t t_tab, t_dec, t_emit, t_tail;
  uint32_t nbatch, nstep;
  uint64_t t_sub[6];  // (MGI_SUBCLOCKS builds only) inside the symbol decoding: stage refill, lanes' decode, walk, prefix sums, queueing, the rest
};
struct Event {                   // a member ended inside a job (not F_ONE_MEMBER)
  uint64_t out_pos;              // elements the job had produced when it ended
  uint32_t job, crc, isize, pad;
};

// (The tables hold 16-bit entries — lit16 / dist16 below — since the end of round 5.  A first 16-bit layout, code length | symbol << 4 with
// base and extra bits recomputed per look-up, was 20 % slower than 32-bit entries: the kernel is bound by instructions issued.  The
// layout kept decodes an entry with the 32-bit layout's instruction count; tests/host_inflate_check.cpp holds every symbol and length
// of the two layouts against each other.)
struct Shared {
  uint16_t lit[1 << LB];         // 16-bit entries (lit16 / dist16 below): 2.5 KB of tables instead of 5 — twenty-four jobs per CU
  uint16_t dist[1 << DB];
  uint16_t cnt[2][16], first[2][16], offs[2][16];  // (16 bits each; the whole struct is 6 608 bytes: twenty-four jobs per CU)
  uint16_t sorted[320];          // symbols by (length, symbol): [0, 288) literal/length, [288, 320) distance
  uint32_t nshort, err;
  union {
    struct {                     // while a block's header is read and its tables are made
      uint32_t blkcnt[5][16];    // code lengths per block of 64 symbols
      uint16_t shortsym[32], shortrc[32];
      uint8_t cl[320];
      uint8_t pre[128];
    };
    struct {                     // while its symbols are decoded
      uint32_t rec_lo[kBatchSyms], rec_hi[kBatchSyms];  // first byte in the batch | len << 16 ; dist | literal << 16
    };
  };
  uint64_t headbits[kBatchBytes / 64];
  uint16_t headbase[kBatchBytes / 64];
  uint32_t scan[64];             // what Exec::scan sums up
  uint32_t cut, cut_total;
  uint32_t inbuf[kInWords];      // the compressed words the window decode stands in
};

MGI_HD uint32_t bitrev(uint32_t v, uint32_t nbits) {
#if defined(__has_builtin)
#if __has_builtin(__builtin_bitreverse32)
  return __builtin_bitreverse32(v) >> (32 - nbits);
#define MGI_HAVE_BREV 1
#endif
#endif
#ifndef MGI_HAVE_BREV
  uint32_t r = 0;
  for (uint32_t i = 0; i < nbits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
  return r;
#endif
}
MGI_HD uint32_t popc64(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }

// length symbol 257 + i -> base | extra bits << 16 (RFC 1951 3.2.5)
MGI_HD uint32_t len_sym(uint32_t i) {
  if (i < 8) return 3 + i;
  if (i == 28) return 258;
  const uint32_t x = (i - 4) >> 2;
  return (3 + ((4 + (i & 3)) << x)) | (x << 16);
}
MGI_HD uint32_t dist_sym(uint32_t d) {
  if (d < 4) return 1 + d;
  const uint32_t x = (d - 2) >> 1;
  return (1 + ((2 + (d & 1)) << x)) | (x << 16);
}
// table entry of a symbol whose code has l bits: l | extra << 4 | kind << 8 | value << 16
MGI_HD uint32_t lit_entry(uint32_t s, uint32_t l) {
  if (s < 256) return l | (K_LIT << 8) | (s << 16);
  if (s == 256) return l | (K_EOB << 8);
  if (s > 285) return l | (K_BAD << 8);
  const uint32_t b = len_sym(s - 257);
  return l | ((b >> 16) << 4) | (K_LEN << 8) | ((b & 0xffff) << 16);
}
MGI_HD uint32_t dist_entry(uint32_t s, uint32_t l) {
  if (s > 29) return l | (K_BAD << 8);
  const uint32_t b = dist_sym(s);
  return l | ((b >> 16) << 4) | (K_DIST << 8) | ((b & 0xffff) << 16);
}

// The window decode's tables hold 16-bit entries:
//   literal/length  bits 0-3 the code's length (1 .. LB), bit 4 = a length symbol, bits 5-7 its extra bits, bits 8-15 the literal | the
//                   length's base - 3;  length 0 = not for the window decode: bits 4-7 say why (K_EOB with the code's length in bits
//                   8-11, K_LONG, K_BAD; an empty slot reads as K_BAD)
//   distance        bits 0-3 the code's length (1 .. DB), bits 4-8 the distance symbol (base and extra bits are arithmetic: dist_sym);
//                   length 0: bits 4-7 why (K_LONG, K_BAD)
MGI_HD uint32_t lit16(uint32_t s, uint32_t l) {
  if (s < 256) return l | (s << 8);
  if (s == 256) return (K_EOB << 4) | (l << 8);
  if (s > 285) return K_BAD << 4;
  const uint32_t b = len_sym(s - 257);
  return l | (1u << 4) | ((b >> 16) << 5) | (((b & 0xffffu) - 3u) << 8);
}
MGI_HD uint32_t dist16(uint32_t s, uint32_t l) { return s > 29 ? (uint32_t)(K_BAD << 4) : l | (s << 4); }
// ... as the 32-bit entries the scalar path works with (lit_entry / dist_entry)
MGI_HD uint32_t wide_lit(uint32_t c) {
  const uint32_t nb = c & 15u;
  if (nb) {
    const uint32_t is_len = (c >> 4) & 1u;
    return nb | (((c >> 5) & 7u) << 4) | ((is_len ? K_LEN : K_LIT) << 8) | (((c >> 8) + (is_len ? 3u : 0u)) << 16);
  }
  const uint32_t why = (c >> 4) & 15u;
  if (why == K_EOB) return ((c >> 8) & 15u) | (K_EOB << 8);
  if (why == K_LONG) return K_LONG << 8 | 15u;
  return 0u;
}
MGI_HD uint32_t wide_dist(uint32_t c) {
  const uint32_t dn = c & 15u;
  if (dn) return dist_entry((c >> 4) & 31u, dn);
  return ((c >> 4) & 15u) == K_LONG ? (K_LONG << 8 | 15u) : 0u;
}

MGI_HDI uint32_t expand_lit(uint32_t c) {
  if (c & 0x8000u) return K_LONG << 8 | 15u;
  return (c & 15u) ? lit_entry((c >> 4) & 511u, c & 15u) : 0u;
}
MGI_HDI uint32_t expand_dist(uint32_t c) {
  if (c & 0x8000u) return K_LONG << 8 | 15u;
  return (c & 15u) ? dist_entry((c >> 4) & 511u, c & 15u) : 0u;
}

// ---- the bit reader; deflate packs bits LSB first; the input is read as aligned 32-bit words.  U: the reader is wave-uniform
// (what it loads goes through readfirstlane and it lives in SGPRs); a lane's own reader otherwise ----
template <bool U>
struct BitReaderT {
  const uint32_t* in;
  uint64_t nwords;  // words that may be read (zero behind them)
  uint64_t bb;      // the next bc bits
  uint32_t bc;
  uint64_t wi;      // index of the word in nxt
  uint32_t nxt;
  MGI_HD uint32_t load(uint64_t w) const { return w < nwords ? (U ? MGI_UNI(in[w]) : in[w]) : 0u; }
  MGI_HD void init(const uint32_t* p, uint64_t nw, uint64_t bitpos) {
    in = p;
    nwords = nw;
    const uint32_t sh = (uint32_t)bitpos & 31u;
    wi = bitpos >> 5;
    bb = (uint64_t)(load(wi) >> sh);
    bc = 32 - sh;
    ++wi;
    nxt = load(wi);
  }
  MGI_HD void refill() {  // afterwards bc > 32
    if (bc <= 32) {
      bb |= (uint64_t)nxt << bc;
      bc += 32;
      ++wi;
      nxt = load(wi);
    }
  }
  MGI_HD uint64_t pos() const { return wi * 32 - bc; }
  MGI_HD void drop(uint32_t n) { bb >>= n; bc -= n; }
  MGI_HD uint32_t bits(uint32_t n) {  // n <= 32 <= bc
    const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1ull));
    drop(n);
    return v;
  }
};
using BitReader = BitReaderT<true>;

// ---- execution policies: how a phase meets the lanes ----
// Per-lane values that live from one phase to the next (the speculative decode of a window: bits, length, distance | literal of
// the symbol that would start at the lane's bit) are registers on the device and arrays on the host; tot_at / len_at read
// another lane's (v_readlane with a uniform index).  scan: exclusive prefix sums of sh.scan[0..64) in place -> the total.
struct HostExec {
  Shared* sh = nullptr;
  uint32_t r_totp[64], r_whyp[64], r_len[64][4], r_hi[64][4];
  bool leader() const { return true; }
  void sync() const {}
  void set_syms(int lane, uint32_t totp, uint32_t whyp, const uint32_t (&len)[4], const uint32_t (&hi)[4]) {
    r_totp[lane] = totp;
    r_whyp[lane] = whyp;
    for (int j = 0; j < 4; ++j) { r_len[lane][j] = len[j]; r_hi[lane][j] = hi[j]; }
  }
  uint32_t totp_at(uint32_t l) const { return r_totp[l]; }
  uint32_t whyp_at(uint32_t l) const { return r_whyp[l]; }
  // the bit counts of all 256 positions laid out for the walk: word r of 64 positions in one register, a lane per position
  void spread_tots() {}
  uint32_t tot_in_word(uint32_t r, uint32_t off) const { const uint32_t p = 64u * r + (off & 63u); return (r_totp[p >> 2] >> (8u * (p & 3u))) & 0xffu; }
  uint32_t my_len(int lane, int j) const { return r_len[lane][j]; }
  uint32_t my_hi(int lane, int j) const { return r_hi[lane][j]; }
  uint32_t scan() const {
    uint32_t run = 0;
    for (int l = 0; l < 64; ++l) { const uint32_t v = sh->scan[l]; sh->scan[l] = run; run += v; }
    return run;
  }
  template <class F> void lanes(F&& f) const { for (int l = 0; l < 64; ++l) f(l); }
  // every lane produces a value, gets the sum of the lanes below it and the sum of all; -> that sum
  template <class P, class C> uint32_t lanes_scan(P&& produce, C&& consume) const {
    uint32_t v[64], run = 0;
    for (int l = 0; l < 64; ++l) v[l] = produce(l);
    for (int l = 0; l < 64; ++l) { const uint32_t x = v[l]; v[l] = run; run += x; }
    for (int l = 0; l < 64; ++l) consume(l, v[l], run);
    return run;
  }
  static void atomic_inc(uint32_t* p) { ++*p; }
  static uint32_t atomic_fetch_inc(uint32_t* p) { return (*p)++; }
  static void atomic_or64(uint64_t* p, uint64_t v) { *p |= v; }
  static void atomic_min(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
};
#if defined(__HIPCC__)
struct DevExec {
  Shared* sh;
  int lane;
  uint32_t v_totp = 0, v_whyp = 0, v_len[4] = {0, 0, 0, 0}, v_hi[4] = {0, 0, 0, 0};
  uint32_t v_t0 = 0, v_t1 = 0, v_t2 = 0, v_t3 = 0;  // (four names, not an array: indexed by the walk's word it went to scratch)
  __device__ bool leader() const { return lane == 0; }
  __device__ void sync() const { __syncthreads(); }
  __device__ void set_syms(int, uint32_t totp, uint32_t whyp, const uint32_t (&len)[4], const uint32_t (&hi)[4]) {
    v_totp = totp;
    v_whyp = whyp;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v_len[j] = len[j]; v_hi[j] = hi[j]; }
  }
  __device__ uint32_t totp_at(uint32_t l) const { return (uint32_t)__builtin_amdgcn_readlane((int)v_totp, (int)l); }
  __device__ uint32_t whyp_at(uint32_t l) const { return (uint32_t)__builtin_amdgcn_readlane((int)v_whyp, (int)l); }
  // The walk reads one position's bit count per symbol: with four positions packed per lane that was shift, readlane, shift, mask,
  // add in a chain.  Transposed through 256 bytes of LDS (sh.scan is free during a step's first half) position 64 r + i is lane i of
  // register r, and v_readlane takes the walk's offset as it is (the lane select is its low six bits): readlane, add.
  __device__ void spread_tots() {
    sh->scan[lane] = v_totp;
    __syncthreads();
    const uint8_t* b = reinterpret_cast<const uint8_t*>(sh->scan);
    v_t0 = b[lane];
    v_t1 = b[64 + lane];
    v_t2 = b[128 + lane];
    v_t3 = b[192 + lane];
    __syncthreads();
  }
  __device__ uint32_t tot_in_word(uint32_t r, uint32_t off) const {
    const int o = (int)(off & 63u);
    return (uint32_t)(r == 0 ? __builtin_amdgcn_readlane((int)v_t0, o) : r == 1 ? __builtin_amdgcn_readlane((int)v_t1, o)
                      : r == 2 ? __builtin_amdgcn_readlane((int)v_t2, o) : __builtin_amdgcn_readlane((int)v_t3, o));
  }
  __device__ uint32_t my_len(int, int j) const { return v_len[j]; }
  __device__ uint32_t my_hi(int, int j) const { return v_hi[j]; }
  __device__ uint32_t scan() const {
    const uint32_t v = sh->scan[lane];
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = (uint32_t)__shfl_up((int)x, d, 64);
      if (lane >= d) x += y;
    }
    sh->scan[lane] = x - v;
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
  }
  template <class F> __device__ void lanes(F&& f) const { f(lane); }
  // (prefix sums over the wavefront in registers: four row_shr steps inside the rows of 16 lanes, row_bcast:15 into rows 1 and 3,
  // row_bcast:31 into rows 2 and 3 — no LDS, no barrier)
  __device__ static uint32_t dpp_inclusive(uint32_t v) {
    uint32_t x = v;
#if defined(__HIP_DEVICE_COMPILE__)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
#endif
    return x;
  }
  template <class P, class C> __device__ uint32_t lanes_scan(P&& produce, C&& consume) const {
    const uint32_t v = produce(lane);
    const uint32_t x = dpp_inclusive(v);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    consume(lane, x - v, total);
    return total;
  }
  __device__ static void atomic_inc(uint32_t* p) { __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ static uint32_t atomic_fetch_inc(uint32_t* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ static void atomic_or64(uint64_t* p, uint64_t v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ static void atomic_min(uint32_t* p, uint32_t v) { __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
};
#endif

// ---- canonical Huffman codes -> look-up table + what the slow path (codes longer than the table's bits) needs ----
// which = 0: literal/length code, cl[base, base + n), table sh.lit of LB bits; 1: distance code, table sh.dist of DB bits.
// strict: what the block-start finder asks of a candidate (a complete code, or at most one distance code).
// Returns 0, or ST_BAD_CODES.
template <class Exec>
MGI_HD uint32_t build_table(Exec& ex, Shared& sh, int which, uint32_t base, uint32_t n, bool table, bool strict = false) {
  const uint32_t TB = which ? DB : LB;
  uint16_t* tab = which ? sh.dist : sh.lit;
  ex.lanes([&](int lane) {
    if (lane < 16) sh.cnt[which][lane] = 0;
    for (uint32_t i = (uint32_t)lane; i < 80; i += 64) (&sh.blkcnt[0][0])[i] = 0;
    if (lane == 0) sh.nshort = 0;
    if (table)
      for (uint32_t i = (uint32_t)lane; i < (1u << TB); i += 64) tab[i] = 0;
  });
  ex.sync();
  ex.lanes([&](int lane) {
    for (uint32_t s = (uint32_t)lane; s < n; s += 64) Exec::atomic_inc(&sh.blkcnt[s >> 6][sh.cl[base + s]]);
  });
  ex.sync();
  // uniform: counts, Kraft, first code and first rank of every length
  uint32_t used = 0, code = 0, rank = 0, c1 = 0;
  int32_t left = 1;
  bool over = false;
  for (uint32_t l = 1; l <= 15; ++l) {
    uint32_t c = 0;
    for (uint32_t b = 0; b * 64 < n; ++b) c += MGI_UNI(sh.blkcnt[b][l]);
    if (l == 1) c1 = c;
    left = (left << 1) - (int32_t)c;
    if (left < 0) over = true;
    if (ex.leader()) { sh.cnt[which][l] = c; sh.first[which][l] = code; sh.offs[which][l] = rank; }
    code = (code + c) << 1;
    rank += c;
    used += c;
  }
  if (over) return ST_BAD_CODES;
  if (left > 0) {
    // zlib (inflate_table): an incomplete code is accepted only when every code has one bit — and never for the literal/length
    // code of a block the finder is to trust
    if (used != 0 && !(c1 == used)) return ST_BAD_CODES;
    if (strict && (which == 0 || used > 1)) return ST_BAD_CODES;
  }
  ex.sync();
  if (!table) return 0;
  ex.lanes([&](int lane) {
    for (uint32_t s0 = 0; s0 < n; s0 += 64) {
      const uint32_t s = s0 + (uint32_t)lane;
      if (s >= n) break;
      const uint32_t l = sh.cl[base + s];
      if (!l) continue;
      uint32_t r = 0;
      for (uint32_t b = 0; b < (s0 >> 6); ++b) r += sh.blkcnt[b][l];
      for (uint32_t j = 0; j < (uint32_t)lane; ++j) r += sh.cl[base + s0 + j] == l;
      sh.sorted[(which ? 288 : 0) + sh.offs[which][l] + r] = (uint16_t)s;
      const uint32_t rc = bitrev(sh.first[which][l] + r, l);  // the code as its bits arrive
      const uint16_t e = (uint16_t)(which ? dist16(s, l) : lit16(s, l));
      if (l > TB) {
        tab[rc & ((1u << TB) - 1u)] = (uint16_t)(K_LONG << 4);  // (the slow path finds the length)
      } else if ((1u << (TB - l)) >= 64u) {
        const uint32_t i = Exec::atomic_fetch_inc(&sh.nshort);
        sh.shortsym[i] = (uint16_t)s;
        sh.shortrc[i] = (uint16_t)rc;
      } else {
        for (uint32_t j = rc; j < (1u << TB); j += 1u << l) tab[j] = e;
      }
    }
  });
  ex.sync();
  const uint32_t ns = MGI_UNI(sh.nshort);
  if (ns) {
    ex.lanes([&](int lane) {
      for (uint32_t i = 0; i < ns; ++i) {
        const uint32_t s = sh.shortsym[i], rc = sh.shortrc[i], l = sh.cl[base + s];
        const uint16_t e = (uint16_t)(which ? dist16(s, l) : lit16(s, l));
        for (uint32_t j = (uint32_t)lane; j < (1u << (TB - l)); j += 64) tab[rc + (j << l)] = e;
      }
    });
    ex.sync();
  }
  return 0;
}

// a code longer than the table's bits: canonical search over the lengths (uniform)
MGI_HD uint32_t slow_entry(const Shared& sh, int which, uint64_t bb) {
  const uint32_t TB = which ? DB : LB;
  const uint32_t c15 = bitrev((uint32_t)bb & 0x7fffu, 15);
  for (uint32_t l = TB + 1; l <= 15; ++l) {
    const uint32_t c = c15 >> (15 - l);
    const uint32_t i = c - MGI_UNI(sh.first[which][l]);
    if (i < MGI_UNI(sh.cnt[which][l])) {
      const uint32_t s = MGI_UNI(sh.sorted[(which ? 288 : 0) + MGI_UNI(sh.offs[which][l]) + i]);
      return which ? dist_entry(s, l) : lit_entry(s, l);
    }
  }
  return 0;  // no such code
}

// ---- the header of a dynamic block: code lengths into sh.cl[0, nlit + ndist) ----
template <class Exec>
MGI_HD uint32_t read_dynamic_lengths(Exec& ex, Shared& sh, BitReader& br, uint32_t* nlit_out, uint32_t* ndist_out) {
  br.refill();
  const uint32_t nlit = br.bits(5) + 257, ndist = br.bits(5) + 1, ncl = br.bits(4) + 4;
  if (nlit > 286 || ndist > 30) return ST_BAD_CODES;
  // the code-length code: 3-bit lengths in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
  const uint64_t order_lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 |
                            5ull << 45 | 11ull << 50 | 4ull << 55;
  const uint64_t order_hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
  uint64_t plen = 0;  // 3 bits per symbol
  uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = 0; i < ncl; ++i) {
    br.refill();
    const uint32_t l = br.bits(3);
    const uint32_t sym = (uint32_t)((i < 12 ? order_lo >> (5 * i) : order_hi >> (5 * (i - 12))) & 31u);
    plen |= (uint64_t)l << (3 * sym);
    // (a fixed-trip unrolled update keeps cnt[] in registers)
    for (uint32_t k = 1; k < 8; ++k) cnt[k] += l == k;
  }
  int32_t left = 1;
  uint32_t next[8];
  {
    uint32_t code = 0;
    for (uint32_t l = 1; l < 8; ++l) {
      left = (left << 1) - (int32_t)cnt[l];
      if (left < 0) return ST_BAD_CODES;
      next[l] = code;
      code = (code + cnt[l]) << 1;
    }
  }
  if (left != 0) return ST_BAD_CODES;  // (zlib: an incomplete code-length code is an error)
  ex.sync();                           // (whoever read sh.pre before is done)
  for (uint32_t s = 0; s < 19; ++s) {
    const uint32_t l = (uint32_t)(plen >> (3 * s)) & 7u;
    if (!l) continue;
    uint32_t c = 0;
    for (uint32_t k = 1; k < 8; ++k) if (k == l) { c = next[k]; next[k] = c + 1; }
    const uint32_t rc = bitrev(c, l);
    if (ex.leader())
      for (uint32_t j = rc; j < 128; j += 1u << l) sh.pre[j] = (uint8_t)(l | (s << 3));
  }
  ex.sync();
  const uint32_t total = nlit + ndist;
  uint32_t i = 0, prev = 0;
  while (i < total) {
    br.refill();
    const uint32_t e = MGI_UNI(sh.pre[(uint32_t)br.bb & 127u]);
    br.drop(e & 7u);
    const uint32_t s = e >> 3;
    if (s < 16) {
      if (ex.leader()) sh.cl[i] = (uint8_t)s;
      prev = s;
      ++i;
      continue;
    }
    uint32_t rep, val = 0;
    if (s == 16) {
      if (i == 0) return ST_BAD_CODES;
      rep = 3 + br.bits(2);
      val = prev;
    } else if (s == 17) {
      rep = 3 + br.bits(3);
    } else {
      rep = 11 + br.bits(7);
    }
    if (i + rep > total) return ST_BAD_CODES;
    if (ex.leader())
      for (uint32_t j = 0; j < rep; ++j) sh.cl[i + j] = (uint8_t)val;
    if (s != 16) prev = 0;
    i += rep;
  }
  ex.sync();
  if (MGI_UNI(sh.cl[256]) == 0) return ST_BAD_CODES;  // no end-of-block code
  *nlit_out = nlit;
  *ndist_out = ndist;
  return 0;
}

template <class Exec>
MGI_HD void fixed_lengths(Exec& ex, Shared& sh) {
  ex.sync();
  ex.lanes([&](int lane) {
    for (uint32_t s = (uint32_t)lane; s < 320; s += 64) sh.cl[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
  });
  ex.sync();
}

// ---- one batch of symbols ----
// One symbol by the scalar path (uniform): what the window decode leaves over — the end of a block, codes longer than the tables'
// bits.  -> K_LIT / K_LEN with *len, *hi (distance | literal << 16), K_EOB, or K_BAD.
MGI_HD uint32_t decode_one(const Shared& sh, BitReader& br, uint32_t* len, uint32_t* hi) {
  br.refill();
  uint32_t e = wide_lit(MGI_UNI(sh.lit[(uint32_t)br.bb & ((1u << LB) - 1u)]));
  if (((e >> 8) & 7u) == K_LONG) e = slow_entry(sh, 0, br.bb);
  if ((e & 15u) == 0) return K_BAD;
  br.drop(e & 15u);
  const uint32_t kind = (e >> 8) & 7u;
  if (kind == K_LIT) { *len = 1; *hi = (e >> 16) << 16; return K_LIT; }
  if (kind != K_LEN) return kind == K_EOB ? K_EOB : K_BAD;
  *len = (e >> 16) + br.bits((e >> 4) & 15u);
  br.refill();
  uint32_t d = wide_dist(MGI_UNI(sh.dist[(uint32_t)br.bb & ((1u << DB) - 1u)]));
  if (((d >> 8) & 7u) == K_LONG) d = slow_entry(sh, 1, br.bb);
  if ((d & 15u) == 0 || ((d >> 8) & 7u) != K_DIST) return K_BAD;
  br.drop(d & 15u);
  *hi = (d >> 16) + br.bits((d >> 4) & 15u);
  return K_LEN;
}

// The symbol that WOULD start at a bit, decoded by one lane from the 48 bits there (b0: the first 32, b1: the rest): both table
// look-ups, extra bits included.  -> *tot = its bits (at most 48) with *len / *hi, or *tot = 0: the scalar path's business
// (*len = why: K_EOB, K_LONG, K_BAD).
MGI_HDI void decode_at(const Shared& sh, uint32_t b0, uint32_t b1, uint32_t* tot, uint32_t* len, uint32_t* hi) {
  // (no branches: the lanes of a wavefront hold literals and matches side by side, both ways would be walked anyway; a literal's
  // distance look-up reads some entry and is thrown away)
  const uint32_t e = sh.lit[b0 & ((1u << LB) - 1u)];
  const uint32_t nb = e & 15u, is_len = (e >> 4) & 1u, xb = (e >> 5) & 7u, v = e >> 8;  // (a literal's extra bits are 0: c1 = nb)
  const uint32_t length = v + 3u + ((b0 >> nb) & ((1u << xb) - 1u));
  const uint32_t c1 = nb + xb;  // at most 15
  const uint32_t y = (uint32_t)((((uint64_t)b1 << 32) | b0) >> c1);
  const uint32_t d = sh.dist[y & ((1u << DB) - 1u)];
  const uint32_t dn = d & 15u, ds = (d >> 4) & 31u;
  const uint32_t dx = ds < 2u ? 0u : (ds - 2u) >> 1;                              // dist_sym, branch-free
  const uint32_t dbase = ds < 2u ? 1u + ds : 1u + ((2u + (ds & 1u)) << dx);
  const uint32_t distance = dbase + ((y >> dn) & ((1u << dx) - 1u));
  const bool ok = nb != 0u;
  const bool is_lit = ok && !is_len, is_match = ok && is_len && dn != 0u;
  const uint32_t wl = (e >> 4) & 15u, wd = (d >> 4) & 15u;                        // why, of an entry without a length
  const uint32_t why = !ok ? (wl ? wl : (uint32_t)K_BAD) : (wd == K_LONG ? (uint32_t)K_LONG : (uint32_t)K_BAD);
  *tot = is_lit ? nb : is_match ? c1 + dn + dx : 0u;
  *len = is_lit ? 1u : is_match ? length : why;
  *hi = is_lit ? v << 16 : is_match ? distance : 0u;
}

// the word of a window's 256-bit membership set that holds lane's four positions; those four bits
MGI_HDI uint64_t lane_word(int lane, uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3) {
  const uint32_t r = (uint32_t)lane >> 4;
  return r == 0 ? s0 : r == 1 ? s1 : r == 2 ? s2 : s3;
}
MGI_HDI uint32_t lane_members(int lane, uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3) {
  return (uint32_t)(lane_word(lane, s0, s1, s2, s3) >> (4u * ((uint32_t)lane & 15u))) & 15u;
}

// Symbols until the batch is full (-> 0), the block ends (-> 1), or an ST_ error.  Every step looks at a window of 256 bit
// positions: lane l decodes the four symbols that would start at bits 4l .. 4l + 3 (decode_at; the compressed words come from a
// small stage in LDS), a uniform walk from bit 0 — one v_readlane per symbol — picks the positions that really start one, and their
// lanes queue the symbols: place in the batch from a popcount, first output byte from a prefix sum of the lengths, a bit in the
// bitmap of first bytes.  queue = false: count only.
template <class Exec>
#if defined(MGI_SUBCLOCKS)
#define MGI_SUB(i) do { const uint64_t now_ = MGI_CLOCK(); if (sub) sub[i] += now_ - tick_; tick_ = now_; } while (0)
#else
#define MGI_SUB(i) do { } while (0)
#endif
// *inbuf_base: which compressed words sh.inbuf holds (~0: none) — it lives from batch to batch of a job: a batch starts where the one
// before ended, in words the stage mostly still holds (every batch used to begin with a trip to memory).
MGI_HD uint32_t decode_batch(Exec& ex, Shared& sh, BitReader& br, bool queue, uint32_t* nsym_out, uint32_t* T_out, uint64_t* inbuf_base,
                             uint32_t* nstep = nullptr, uint64_t* sub = nullptr) {
  uint32_t nsym = 0, T = 0, rc = 0, steps = 0;
#if defined(MGI_SUBCLOCKS)
  uint64_t tick_ = MGI_CLOCK();
#endif
  (void)sub;
  uint64_t pos = br.pos();
  const uint32_t* in = br.in;
  const uint64_t nwords = br.nwords;
  uint64_t base_w = *inbuf_base;  // sh.inbuf = words [base_w, base_w + kInWords)
  if (queue) {
    ex.sync();
    ex.lanes([&](int lane) { sh.headbits[lane] = 0; });
  }
  for (;;) {
    if (nsym + kStepSyms + 64 > kBatchSyms || T + 258 > kBatchBytes) break;
    ++steps;
    const uint64_t wi = pos >> 5;
    const uint32_t so = (uint32_t)pos & 31u;
    if (base_w == ~0ull || wi < base_w || wi + 11 > base_w + kInWords) {  // (31 + 255 + 51 bits: eleven words)
      ex.sync();
      base_w = wi;
      ex.lanes([&](int lane) {
        for (uint32_t i = (uint32_t)lane; i < kInWords; i += 64) sh.inbuf[i] = base_w + i < nwords ? in[base_w + i] : 0u;
      });
    }
    ex.sync();
    MGI_SUB(0);
    const uint32_t rel = (uint32_t)(wi - base_w);
    ex.lanes([&](int lane) {
      const uint32_t o0 = so + 4u * (uint32_t)lane;
      const uint32_t i = rel + (o0 >> 5), sft = o0 & 31u;
      const uint64_t lo = sh.inbuf[i] | (uint64_t)sh.inbuf[i + 1] << 32;
      const uint64_t x = sft ? (lo >> sft) | ((uint64_t)sh.inbuf[i + 2] << (64u - sft)) : lo;  // the 64 bits at o0
      uint32_t totp = 0, whyp = 0, len[4], hi[4];
      MGI_UNROLL
      for (uint32_t j = 0; j < 4; ++j) {
        const uint64_t y = x >> j;
        uint32_t tot;
        decode_at(sh, (uint32_t)y, (uint32_t)(y >> 32), &tot, &len[j], &hi[j]);
        totp |= (tot ? tot : 255u) << (8u * j);  // (255 "bits": the walk leaves the window at a position the scalar path has to look at)
        if (!tot) whyp |= len[j] << (8u * j);
      }
      ex.set_syms(lane, totp, whyp, len, hi);
    });
    ex.spread_tots();
    MGI_SUB(1);
    // which positions start a symbol: from bit 0, every symbol says where the next one starts
    uint32_t off = 0, cnt = 0;
    uint64_t S0 = 0, S1 = 0, S2 = 0, S3 = 0;  // (four words, not an array: they stay in SGPRs)
    bool special = false;
    // (one loop per 64 positions, so that its word of the set stays in one register pair; a position the scalar path has to look at
    // says "255 bits", which ends the walk by itself — the loop is: read the lane, extract, set the bit, add, compare; the step's
    // symbol limit is looked at between the loops: at most kStepSyms - 1 + 64 symbols)
    uint32_t last = 0;
#define MGI_TOT(p) (ex.tot_in_word((p) >> 6, (p)))
#define MGI_WALK(Sr, R, END)                                                                                 \
    if (off < (END) && cnt < kStepSyms) {                                                                      \
      do {                                                                                                     \
        last = off;                                                                                            \
        Sr |= 1ull << (off & 63u);                                                                             \
        off += ex.tot_in_word(R, off);                                                                         \
      } while (off < (END));                                                                                   \
      cnt += popc64(Sr);                                                                                       \
    }
    if (MGI_TOT(0u) == 255u) {  // (from any other position 255 more bits are behind the window)
      special = true;
    } else {
      MGI_WALK(S0, 0u, 64u)
      MGI_WALK(S1, 1u, 128u)
      MGI_WALK(S2, 2u, 192u)
      MGI_WALK(S3, 3u, 256u)
      if (MGI_TOT(last) == 255u) {  // the last position looked at was the scalar path's: it is no member, the walk stands there
        special = true;
        off = last;
        const uint64_t bit = 1ull << (last & 63u);
        const uint32_t r = last >> 6;
        if (r == 0) S0 &= ~bit; else if (r == 1) S1 &= ~bit; else if (r == 2) S2 &= ~bit; else S3 &= ~bit;
        --cnt;
      }
    }
#undef MGI_WALK
#undef MGI_TOT
    MGI_SUB(2);
    if (cnt) {
      const uint64_t a0 = S0, a1 = S1, a2 = S2, a3 = S3;  // (as found; the cut below may drop the last ones)
      const uint32_t d0 = popc64(a0), d1 = popc64(a1), d2 = popc64(a2);
      // a lane's symbols: place in the batch from the popcounts, first output byte from the prefix sum of the lengths — in one phase
      // (the sums in registers); only a window in which the batch's bytes run out takes the slow way below
      auto queue_lane = [&](int lane, uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3, uint32_t e0, uint32_t e1, uint32_t e2, uint32_t excl) {
        const uint32_t m = lane_members(lane, s0, s1, s2, s3);
        if (!m) return;
        const uint32_t r = (uint32_t)lane >> 4;
        uint32_t i = nsym + (r >= 1 ? e0 : 0u) + (r >= 2 ? e1 : 0u) + (r >= 3 ? e2 : 0u) +  // (sums, not a chain of ==: that became a table in scratch)
                     popc64(lane_word(lane, s0, s1, s2, s3) & ((1ull << (4u * ((uint32_t)lane & 15u))) - 1ull));
        uint32_t dst = T + excl;
        MGI_UNROLL
        for (int j = 0; j < 4; ++j) {
          if (!((m >> j) & 1u)) continue;
          const uint32_t len = ex.my_len(lane, j);
          sh.rec_lo[i] = dst | len << 16;
          sh.rec_hi[i] = ex.my_hi(lane, j);
          Exec::atomic_or64(&sh.headbits[dst >> 6], 1ull << (dst & 63u));
          ++i;
          dst += len;
        }
      };
      uint32_t total = ex.lanes_scan(
          [&](int lane) -> uint32_t {
            const uint32_t m = lane_members(lane, a0, a1, a2, a3);
            uint32_t sum = 0;
            MGI_UNROLL
            for (int j = 0; j < 4; ++j) sum += (m >> j) & 1u ? ex.my_len(lane, j) : 0u;
            return sum;
          },
          [&](int lane, uint32_t excl, uint32_t tot) {
            if (T + tot > kBatchBytes) {
              sh.scan[lane] = excl;
              if (lane == 0) sh.cut = kStepBits;
            } else if (queue) {
              queue_lane(lane, a0, a1, a2, a3, d0, d1, d2, excl);
            }
          });
      MGI_SUB(3);
      if (T + total > kBatchBytes) {  // the batch's bytes run out inside this window: up to the first symbol that does not fit
        ex.sync();
        ex.lanes([&](int lane) {
          const uint32_t m = lane_members(lane, a0, a1, a2, a3);
          uint32_t run = T + sh.scan[lane];
          bool done = false;
          MGI_UNROLL
          for (int j = 0; j < 4; ++j) {
            if (done || !((m >> j) & 1u)) continue;
            if (run + ex.my_len(lane, j) > kBatchBytes) { Exec::atomic_min(&sh.cut, 4u * (uint32_t)lane + (uint32_t)j); done = true; continue; }
            run += ex.my_len(lane, j);
          }
        });
        ex.sync();
        const uint32_t c = MGI_UNI(sh.cut);
        ex.lanes([&](int lane) {
          if ((uint32_t)lane != c >> 2) return;
          const uint32_t m = lane_members(lane, a0, a1, a2, a3);
          uint32_t run = sh.scan[lane];
          MGI_UNROLL
          for (uint32_t j = 0; j < 3; ++j) run += j < (c & 3u) && ((m >> j) & 1u) ? ex.my_len(lane, (int)j) : 0u;
          sh.cut_total = run;
        });
        ex.sync();
        total = MGI_UNI(sh.cut_total);
        auto trim = [&](uint64_t Sr, uint32_t first) -> uint64_t {
          return c < first ? 0ull : c < first + 64u ? Sr & ((1ull << (c & 63u)) - 1ull) : Sr;
        };
        S0 = trim(S0, 0);
        S1 = trim(S1, 64);
        S2 = trim(S2, 128);
        S3 = trim(S3, 192);
        cnt = popc64(S0) + popc64(S1) + popc64(S2) + popc64(S3);
        off = c;
        special = false;
        if (queue) {
          const uint64_t b0 = S0, b1 = S1, b2 = S2, b3 = S3;
          const uint32_t f0 = popc64(b0), f1 = popc64(b1), f2 = popc64(b2);
          ex.lanes([&](int lane) { queue_lane(lane, b0, b1, b2, b3, f0, f1, f2, sh.scan[lane]); });
        }
      }
      nsym += cnt;
      T += total;
      MGI_SUB(4);
    }
    pos += off;
    if (!special) {
      if (cnt == 0) break;  // (the first symbol of the window did not fit the batch any more)
      continue;
    }
    const uint32_t why = (ex.whyp_at(off >> 2) >> (8u * (off & 3u))) & 0xffu;
    if (why == K_BAD) { rc = ST_BAD_SYMBOL; break; }
    BitReader one;
    one.init(in, nwords, pos);
    uint32_t len = 0, hi = 0;
    const uint32_t kind = decode_one(sh, one, &len, &hi);
    if (kind == K_BAD) { rc = ST_BAD_SYMBOL; break; }
    if (kind == K_EOB) { pos = one.pos(); rc = 1; break; }
    if (nsym + 1 > kBatchSyms || T + len > kBatchBytes) break;  // (not taken: the next batch starts with it)
    if (queue) {
      ex.sync();
      if (ex.leader()) {
        sh.rec_lo[nsym] = T | len << 16;
        sh.rec_hi[nsym] = hi;
        sh.headbits[T >> 6] |= 1ull << (T & 63u);
      }
    }
    ++nsym;
    T += len;
    pos = one.pos();
  }
  if (queue && T) {  // how many symbols start before every 64 bytes of the batch
    ex.sync();
    ex.lanes([&](int lane) { sh.scan[lane] = popc64(sh.headbits[lane]); });
    ex.sync();
    ex.scan();
    ex.sync();
    ex.lanes([&](int lane) { sh.headbase[lane] = (uint16_t)sh.scan[lane]; });
    // Matches that copy a match of this batch that copies a match ... (the quality line of one FASTQ record after another): every
    // byte of the last one would walk the whole chain.  A match whose source lies inside ONE earlier match of the batch (one that does
    // not overlap itself) may as well copy from where that one copies: twice over all symbols, distances added (pointer jumping).
    for (int it = 0; it < 2; ++it) {
      ex.sync();
      ex.lanes([&](int lane) {
        for (uint32_t k = (uint32_t)lane; k < nsym; k += 64) {
          const uint32_t lo = sh.rec_lo[k], hi = sh.rec_hi[k];
          const uint32_t dist = hi & 0xffffu, dst = lo & 0xffffu, len = lo >> 16;
          if (dist == 0 || dist > dst || dist < len) continue;  // a literal; a source in front of the batch; a run
          const uint32_t a = dst - dist, b = a + len - 1;
          const uint32_t wa = a >> 6, wb = b >> 6;
          const uint32_t j = sh.headbase[wa] + popc64(sh.headbits[wa] & ((2ull << (a & 63u)) - 1ull)) - 1u;
          const uint32_t je = sh.headbase[wb] + popc64(sh.headbits[wb] & ((2ull << (b & 63u)) - 1ull)) - 1u;
          if (j != je) continue;
          const uint32_t hj = sh.rec_hi[j], dj = hj & 0xffffu;
          if (dj == 0 || dj < (sh.rec_lo[j] >> 16) || dist + dj > 0xffffu) continue;
          sh.rec_hi[k] = dist + dj;
        }
      });
    }
  }
  br.init(in, nwords, pos);
  *inbuf_base = base_w;
  *nsym_out = nsym;
  *T_out = T;
  if (nstep) *nstep += steps;
  MGI_SUB(5);
  return rc;
}

// ---- emission: lane l makes bytes l, l + 64, ... of the batch ----
// out: the job's output (element 0 = the first the job produces); pos0: elements before the batch; floor: the position
// below which nothing may be referenced — the member's start inside the job, 0 when the job starts a member, and
// -32768 for a job that starts inside a member (16-bit output only: positions below 0 become window symbols).
struct Chase {  // one output byte on its way through the batch's records
  uint32_t p, val;
  int64_t q;
  bool act, ld;
};
template <class OutT>
MGI_HDI bool chase_step(Shared& sh, Chase& c, uint64_t pos0, int64_t floor) {  // -> still inside the batch
  const uint32_t w = c.p >> 6;
  const uint32_t own = sh.headbase[w] + popc64(sh.headbits[w] & ((2ull << (c.p & 63u)) - 1ull)) - 1u;
  const uint32_t lo = sh.rec_lo[own], hi = sh.rec_hi[own];
  const uint32_t dist = hi & 0xffffu;
  if (dist == 0) { c.val = hi >> 16; c.act = false; return false; }
  const uint32_t dst = lo & 0xffffu, len = lo >> 16;
  uint32_t off = c.p - dst;
  if (dist < len) off %= dist;  // a match that overlaps itself repeats its first dist bytes
  const int64_t src = (int64_t)pos0 + (int64_t)dst + (int64_t)off - (int64_t)dist;
  if (src >= (int64_t)pos0) { c.p = (uint32_t)(src - (int64_t)pos0); return true; }  // made by this batch: follow it
  c.act = false;
  if (src < floor) { sh.err = ST_BAD_DIST; return false; }
  if (src >= 0) { c.ld = true; c.q = src; return false; }
  c.val = sizeof(OutT) == 2 ? (0x8000u | (uint32_t)(src + (int64_t)kWindow)) : 0u;
  return false;
}
template <class OutT>
MGI_HDI void emit_lane(Shared& sh, int lane, uint32_t T, OutT* out, uint64_t pos0, int64_t floor) {
  // kEmitWays bytes per lane at a time (f, f + 64, f + 128, ...): their chains through the batch's records are independent, so
  // their LDS reads are in flight together; then the loads from memory, then the stores
  constexpr uint32_t W = kEmitWays;
  for (uint32_t f0 = (uint32_t)lane; f0 < T; f0 += 64u * W) {
    Chase c[W];
    MGI_UNROLL
    for (uint32_t i = 0; i < W; ++i) c[i] = Chase{f0 + 64u * i, 0, 0, f0 + 64u * i < T, false};
    for (bool any = true; any;) {
      any = false;
      MGI_UNROLL
      for (uint32_t i = 0; i < W; ++i)
        if (c[i].act) any |= chase_step<OutT>(sh, c[i], pos0, floor);
    }
    MGI_UNROLL
    for (uint32_t i = 0; i < W; ++i)
      if (c[i].ld) c[i].val = out[c[i].q];
    MGI_UNROLL
    for (uint32_t i = 0; i < W; ++i)
      if (f0 + 64u * i < T) out[pos0 + f0 + 64u * i] = (OutT)c[i].val;
  }
}

// ---- gzip member header at byte P (RFC 1952) -> first byte of the deflate data.  0 ok, 1 not a member, 2 ends inside ----
template <bool U = true>
MGI_HD uint32_t in_byte(const uint32_t* in, uint64_t i) {
  const uint32_t w = in[i >> 2];
  return ((U ? MGI_UNI(w) : w) >> (8u * ((uint32_t)i & 3u))) & 0xffu;
}
template <bool U = true>
MGI_HD uint32_t gzip_header(const uint32_t* in, uint64_t nbytes, uint64_t P, uint64_t* data) {
  // what is there of the first three bytes decides between "not a member" (trailing garbage, ignored as gzip does) and "cut"
  const uint32_t magic[3] = {0x1f, 0x8b, 8};
  for (uint32_t i = 0; i < 3; ++i) {
    if (P + i >= nbytes) return i == 0 ? 1u : 2u;
    if (in_byte<U>(in, P + i) != magic[i]) return 1u;
  }
  if (P + 10 > nbytes) return 2u;
  const uint32_t flg = in_byte<U>(in, P + 3);
  uint64_t at = P + 10;
  if (flg & 4u) {  // FEXTRA
    if (at + 2 > nbytes) return 2u;
    at += 2 + (in_byte<U>(in, at) | in_byte<U>(in, at + 1) << 8);
  }
  for (uint32_t f = 8u; f <= 16u; f <<= 1) {  // FNAME, FCOMMENT: zero-terminated
    if (!(flg & f)) continue;
    for (;;) {
      if (at >= nbytes) return 2u;
      if (in_byte<U>(in, at++) == 0) break;
    }
  }
  if (flg & 2u) at += 2;  // FHCRC
  if (at > nbytes) return 2u;
  *data = at;
  return 0u;
}

// ---- a job: blocks from start_bit until a boundary at or behind stop_bit / the end of the member / of the stream ----
// in: the compressed bytes as words, nbytes of them there so far, input_final: that is the whole file.
// out: the job's own output (already offset).  events / nevents / max_events: member ends inside jobs (global, atomic).
struct alignas(16) Sym8 { uint16_t v[8]; };
// tail (16-bit output only; may be null): 32768 symbols, 16-byte aligned — the window BEHIND the job as far as the job knows it:
// its last 32 KB of symbols, or, for a shorter job, the end of the window in front of it (as window symbols) and then its own.
template <class Exec, class OutT>
MGI_HD void run_job(Exec& ex, Shared& sh, const uint32_t* in, uint64_t nbytes, bool input_final, const Job& job, uint32_t jobidx,
                    OutT* out, Result* res, Event* events, uint32_t* nevents, uint32_t max_events, uint16_t* tail = nullptr) {
  const uint64_t nbits = nbytes * 8, nwords = (nbytes + 3) / 4;
  const uint32_t cut = input_final ? ST_TRUNC : ST_NEED_MORE;
  uint64_t outn = 0;
  int64_t floor = (job.flags & F_MEMBER_START) ? 0 : (sizeof(OutT) == 2 ? -(int64_t)kWindow : 0);
  uint32_t status = 0, overflow = 0, nblocks = 0, nev = 0, crc = 0, isize = 0, nbatch = 0, nstep = 0;
  uint64_t t_tab = 0, t_dec = 0, t_emit = 0, t_tail = 0, t_sub[6] = {0, 0, 0, 0, 0, 0};
  bool count_only = (job.flags & F_COUNT_ONLY) != 0;
  uint64_t pos = job.start_bit, inbuf_base = ~0ull;
  if (ex.leader()) sh.err = 0;
  if (job.flags & F_HEADER) {
    uint64_t data = 0;
    const uint32_t h = gzip_header(in, nbytes, pos >> 3, &data);
    if (h) status = h == 1 ? ST_BAD_HEADER : cut;
    pos = data * 8;
  }
  BitReader br;
  br.init(in, nwords, pos);
  while (!status) {
    if (br.pos() >= job.stop_bit) { status = ST_STOP; break; }
    br.refill();
    if (br.pos() + 3 > nbits) { status = cut; break; }
    const uint32_t bfinal = br.bits(1), btype = br.bits(2);
    ++nblocks;
    if (btype == 3) { status = ST_BAD_BLOCK; break; }
    if (btype == 0) {
      br.drop(br.bc & 7u);  // to the byte boundary (pos() is a multiple of 8 exactly when bc is)
      br.refill();
      if (br.pos() + 32 > nbits) { status = cut; break; }
      const uint32_t len = br.bits(16);
      br.refill();
      const uint32_t nlen = br.bits(16);
      if ((len ^ nlen) != 0xffffu) { status = ST_BAD_STORED; break; }
      const uint64_t P = br.pos() >> 3;
      if (P + len > nbytes) { status = cut; break; }
      if (!count_only && outn + len > job.out_cap) { overflow = 1; count_only = true; }
      if (!count_only) {
        const uint8_t* bytes = reinterpret_cast<const uint8_t*>(in);
        ex.lanes([&](int lane) {
          for (uint32_t i = (uint32_t)lane; i < len; i += 64) out[outn + i] = (OutT)bytes[P + i];
        });
      }
      outn += len;
      br.init(in, nwords, (P + len) * 8);
    } else {
      uint32_t nlit = 288, ndist = 32, rc;
      const uint64_t c0 = MGI_CLOCK();
      if (btype == 1) {
        fixed_lengths(ex, sh);
      } else {
        rc = read_dynamic_lengths(ex, sh, br, &nlit, &ndist);
        if (rc) { status = rc; break; }
        if (br.pos() > nbits) { status = cut; break; }
      }
      rc = build_table(ex, sh, 0, 0, nlit, true);
      if (!rc) rc = build_table(ex, sh, 1, nlit, ndist, true);
      if (rc) { status = rc; break; }
      t_tab += MGI_CLOCK() - c0;
      for (;;) {
        uint32_t nsym = 0, T = 0;
        const uint64_t c1 = MGI_CLOCK();
        rc = decode_batch(ex, sh, br, !count_only, &nsym, &T, &inbuf_base, &nstep, t_sub);
        ++nbatch;
        const uint64_t c2 = MGI_CLOCK();
        t_dec += c2 - c1;
        if (rc >= ST_ERR) { status = rc; break; }
        if (br.pos() > nbits) { status = cut; break; }
        if (!count_only && outn + T > job.out_cap) { overflow = 1; count_only = true; }
        if (!count_only && T) {
          ex.sync();
          ex.lanes([&](int lane) { emit_lane<OutT>(sh, lane, T, out, outn, floor); });
          ex.sync();
          const uint32_t er = MGI_UNI(sh.err);
          t_emit += MGI_CLOCK() - c2;
          if (er) { status = er; break; }
        }
        outn += T;
        if (rc == 1) break;
      }
      if (status) break;
    }
    if (bfinal) {  // the member's trailer: CRC-32 and ISIZE at the next byte boundary
      br.drop(br.bc & 7u);
      br.refill();
      if (br.pos() + 64 > nbits) { status = cut; break; }
      crc = br.bits(32);
      br.refill();
      isize = br.bits(32);
      if (job.flags & F_ONE_MEMBER) { status = ST_MEMBER; break; }
      ++nev;
      if (ex.leader()) {
        uint32_t at;
#if defined(__HIP_DEVICE_COMPILE__)
        at = __hip_atomic_fetch_add(nevents, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        at = (*nevents)++;
#endif
        if (at < max_events) {
          events[at].out_pos = outn;
          events[at].job = jobidx;
          events[at].crc = crc;
          events[at].isize = isize;
          events[at].pad = 0;
        } else {
          sh.err = ST_EVENTS_FULL;
        }
      }
      ex.sync();
      if (MGI_UNI(sh.err)) { status = ST_EVENTS_FULL; break; }
      uint64_t data = 0;
      const uint32_t h = gzip_header(in, nbytes, br.pos() >> 3, &data);
      if (h == 1) {  // nothing follows, or something that is no member (padding, garbage: ignored as gzip does): the stream is over
        status = (br.pos() >> 3) >= nbytes && !input_final ? ST_NEED_MORE : ST_END;
        break;
      }
      if (h == 2) { status = cut; break; }
      floor = (int64_t)outn;
      br.init(in, nwords, data * 8);
    }
  }
  const uint64_t c3 = MGI_CLOCK();
  if (sizeof(OutT) == 2 && tail && !count_only && status < ST_ERR) {
    ex.sync();
    ex.lanes([&](int lane) {
      for (uint32_t w0 = (uint32_t)lane * 8u; w0 < kWindow; w0 += 512u) {
        Sym8 x;
        for (uint32_t k = 0; k < 8; ++k) {
          const uint32_t w = w0 + k;
          x.v[k] = outn >= kWindow ? (uint16_t)out[outn - kWindow + w]
                                   : (w < kWindow - outn ? (uint16_t)(0x8000u | (w + (uint32_t)outn)) : (uint16_t)out[w - (kWindow - (uint32_t)outn)]);
        }
        *reinterpret_cast<Sym8*>(tail + w0) = x;
      }
    });
  }
  t_tail = MGI_CLOCK() - c3;
  // garbage decoded from behind the end of the input is the input's end, not a damaged stream
  if (status >= ST_ERR && status != ST_BAD_HEADER && br.pos() > nbits) status = cut;
  if (ex.leader()) {
    res->end_bit = br.pos();
    res->out_count = outn;
    res->status = status;
    res->overflow = overflow;
    res->crc = crc;
    res->isize = isize;
    res->nblocks = nblocks;
    res->nevents = nev;
    res->t_tab = t_tab;
    res->t_dec = t_dec;
    res->t_emit = t_emit;
    res->t_tail = t_tail;
    res->nbatch = nbatch;
    res->nstep = nstep;
    for (int i = 0; i < 6; ++i) res->t_sub[i] = t_sub[i];
  }
}


// =====================================================================================================================
// One job per LANE (k_inflate_lanes): the serial decoder as every inflate is written, 64 of them side by side in a wavefront.
//
// The wavefront-per-job decoder above spends ~50 vector instructions per symbol whichever way it is arranged (a scalar loop on
// the CU's one scalar unit, or 256 speculative decodes of which 16 are symbols); a file of F deflate blocks offers F-fold
// parallelism, and a 10M-read FASTQ has 17 000 of them.  Here a lane is a whole decoder: its bit buffer and counters in registers,
// its look-up tables in LDS (1280 bytes a lane: 9 bits of literal/length code, 6 of distance code, 16-bit entries = code length
// | symbol << 4; longer codes are searched canonically from the per-length counts), what is touched rarely (code lengths, the
// symbols in code order) in a scratch block of memory per job.  A lane is a small state machine and the wavefront runs ROUNDS: in a
// round a lane reads a block header and builds its tables, or decodes one symbol, or copies up to 16 elements of a match (a long
// match takes several rounds, so one lane's 258-byte match does not hold up 63 literals).  The sources of a match are final
// before the match starts (an overlapping match re-reads its first `dist` elements), so a round's loads never wait for its stores.
// =====================================================================================================================
constexpr int LLB = 9, LDB = 6;
struct LaneMem {            // a lane's LDS: 319 words, an odd number, so that equal indices of different lanes fall into different banks
  uint16_t lit[1 << LLB];
  uint16_t dist[1 << LDB];
  uint16_t cnt_[30];        // codes per length 1..15 of either code (the search for long codes derives first codes and ranks from them)
  uint16_t next[16];        // table build: the next code of every length ...
  uint16_t offs[16];        // ... and the rank of its first symbol
  MGI_HD uint16_t& cnt(int which, uint32_t l) { return cnt_[which * 15 + (int)l - 1]; }
};
static_assert(sizeof(LaneMem) == 1276, "a lane's tables");
struct LaneScratch {        // a job's block of memory
  uint16_t sorted[320];     // symbols by (length, symbol): [0, 288) literal/length, [288, 320) distance
  uint8_t cl[320];
};
enum : uint32_t { LS_BLOCK = 0, LS_SYM = 1, LS_COPY = 2, LS_STORED = 3, LS_MEMBER = 4, LS_DONE = 5 };

template <class OutT>
struct LaneDec {
  const uint32_t* in;
  uint64_t nbytes, nbits;
  bool input_final;
  Job job;
  uint32_t jobidx;
  OutT* out;
  LaneMem* mem;
  LaneScratch* scr;
  Event* events;
  uint32_t* nevents;
  uint32_t max_events;
  BitReaderT<false> br;
  uint64_t outn;
  int64_t floor;
  uint32_t state, status, overflow, nblocks, nev, crc, isize, bfinal;
  bool count_only;
  uint64_t cp_base;         // a match: where it starts, its distance and length, elements done, the source offset of the next one
  uint32_t cp_dist, cp_len, cp_i, cp_o;
  uint64_t st_pos;          // a stored block: the next input byte, bytes left
  uint32_t st_left;
  uint32_t rounds;

  MGI_HD uint32_t cut() const { return input_final ? ST_TRUNC : ST_NEED_MORE; }
  MGI_HD void finish(uint32_t st) {
    // garbage decoded from behind the end of the input is the input's end, not a damaged stream
    if (st >= ST_ERR && st != ST_BAD_HEADER && br.pos() > nbits) st = cut();
    status = st;
    state = LS_DONE;
  }
  MGI_HD void init(const uint32_t* in_, uint64_t nbytes_, bool final_, const Job& job_, uint32_t jobidx_, OutT* out_, LaneMem* mem_, LaneScratch* scr_,
                   Event* events_, uint32_t* nevents_, uint32_t max_events_) {
    in = in_; nbytes = nbytes_; nbits = nbytes_ * 8; input_final = final_; job = job_; jobidx = jobidx_; out = out_; mem = mem_; scr = scr_;
    events = events_; nevents = nevents_; max_events = max_events_;
    outn = 0;
    floor = (job.flags & F_MEMBER_START) ? 0 : (sizeof(OutT) == 2 ? -(int64_t)kWindow : 0);
    state = LS_BLOCK; status = 0; overflow = 0; nblocks = 0; nev = 0; crc = 0; isize = 0; bfinal = 0; rounds = 0;
    count_only = (job.flags & F_COUNT_ONLY) != 0;
    cp_base = 0; cp_dist = cp_len = cp_i = cp_o = 0; st_pos = 0; st_left = 0;
    uint64_t pos = job.start_bit;
    br.init(in, (nbytes + 3) / 4, pos);
    if (job.flags & F_HEADER) {
      uint64_t data = 0;
      const uint32_t h = gzip_header<false>(in, nbytes, pos >> 3, &data);
      if (h) { finish(h == 1 ? (uint32_t)ST_BAD_HEADER : cut()); return; }
      br.init(in, (nbytes + 3) / 4, data * 8);
    }
  }

  // a code longer than the table's bits: canonical search from the counts
  MGI_HD uint32_t slow(int which) {
    const uint32_t TB = which ? LDB : LLB;
    const uint32_t c15 = bitrev((uint32_t)br.bb & 0x7fffu, 15);
    uint32_t first = 0, rank = 0;
    for (uint32_t l = 1; l <= 15; ++l) {
      const uint32_t c = mem->cnt(which, l);
      if (l > TB) {
        const uint32_t k = (c15 >> (15 - l)) - first;
        if (k < c) return l | (uint32_t)scr->sorted[(which ? 288 : 0) + rank + k] << 4;
      }
      rank += c;
      first = (first + c) << 1;
    }
    return 0;
  }

  // code lengths cl[base, base + n) -> the lane's table; 0 or ST_BAD_CODES
  MGI_HD uint32_t build(int which, uint32_t base, uint32_t n) {
    const uint32_t TB = which ? LDB : LLB;
    uint16_t* tab = which ? mem->dist : mem->lit;
    for (uint32_t l = 1; l < 16; ++l) mem->cnt(which, l) = 0;
    for (uint32_t s = 0; s < n; ++s) {
      const uint32_t l = scr->cl[base + s];
      if (l) ++mem->cnt(which, l);
    }
    int32_t left = 1;
    uint32_t code = 0, rank = 0, used = 0;
    for (uint32_t l = 1; l <= 15; ++l) {
      const uint32_t c = mem->cnt(which, l);
      left = (left << 1) - (int32_t)c;
      if (left < 0) return ST_BAD_CODES;
      mem->next[l] = (uint16_t)code;
      mem->offs[l] = (uint16_t)rank;
      code = (code + c) << 1;
      rank += c;
      used += c;
    }
    if (left > 0 && used != 0 && mem->cnt(which, 1) != used) return ST_BAD_CODES;  // (zlib: incomplete only when every code has one bit)
    for (uint32_t i = 0; i < (1u << TB); ++i) tab[i] = 0;
    for (uint32_t s = 0; s < n; ++s) {
      const uint32_t l = scr->cl[base + s];
      if (!l) continue;
      const uint32_t c = mem->next[l];
      mem->next[l] = (uint16_t)(c + 1);
      const uint32_t r = mem->offs[l];
      mem->offs[l] = (uint16_t)(r + 1);
      scr->sorted[(which ? 288 : 0) + r] = (uint16_t)s;
      const uint32_t rc = bitrev(c, l);
      if (l > TB) tab[rc & ((1u << TB) - 1u)] = 0x800fu;
      else
        for (uint32_t j = rc; j < (1u << TB); j += 1u << l) tab[j] = (uint16_t)(l | s << 4);
    }
    return 0;
  }

  // the header of a block, its tables
  MGI_HD void block() {
    if (br.pos() >= job.stop_bit) { finish(ST_STOP); return; }
    br.refill();
    if (br.pos() + 3 > nbits) { finish(cut()); return; }
    bfinal = br.bits(1);
    const uint32_t btype = br.bits(2);
    ++nblocks;
    if (btype == 3) { finish(ST_BAD_BLOCK); return; }
    if (btype == 0) {
      br.drop(br.bc & 7u);
      br.refill();
      if (br.pos() + 32 > nbits) { finish(cut()); return; }
      const uint32_t len = br.bits(16);
      br.refill();
      const uint32_t nlen = br.bits(16);
      if ((len ^ nlen) != 0xffffu) { finish(ST_BAD_STORED); return; }
      st_pos = br.pos() >> 3;
      st_left = len;
      if (st_pos + len > nbytes) { finish(cut()); return; }
      state = LS_STORED;
      return;
    }
    uint32_t nlit = 288, ndist = 32;
    if (btype == 1) {
      for (uint32_t s = 0; s < 320; ++s) scr->cl[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
    } else {
      br.refill();
      nlit = br.bits(5) + 257;
      ndist = br.bits(5) + 1;
      const uint32_t ncl = br.bits(4) + 4;
      if (nlit > 286 || ndist > 30) { finish(ST_BAD_CODES); return; }
      const uint64_t order_lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 |
                                5ull << 45 | 11ull << 50 | 4ull << 55;
      const uint64_t order_hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
      uint64_t plen = 0, cnts = 0;  // 3 bits per symbol; 8 bits per length
      for (uint32_t i = 0; i < ncl; ++i) {
        br.refill();
        const uint32_t l = br.bits(3);
        const uint32_t sym = (uint32_t)((i < 12 ? order_lo >> (5 * i) : order_hi >> (5 * (i - 12))) & 31u);
        plen |= (uint64_t)l << (3 * sym);
        cnts += 1ull << (8 * l);
      }
      uint64_t firsts = 0, offs = 0, sorted_lo = 0, sorted_hi = 0;
      {
        int32_t left = 1;
        uint32_t code = 0, rank = 0;
        for (uint32_t l = 1; l < 8; ++l) {
          const uint32_t c = (uint32_t)(cnts >> (8 * l)) & 0xffu;
          left = (left << 1) - (int32_t)c;
          if (left < 0) { finish(ST_BAD_CODES); return; }
          firsts |= (uint64_t)code << (8 * l);
          offs |= (uint64_t)rank << (8 * l);
          for (uint32_t s = 0; s < 19; ++s) {
            if (((uint32_t)(plen >> (3 * s)) & 7u) != l) continue;
            if (rank < 12) sorted_lo |= (uint64_t)s << (5 * rank); else sorted_hi |= (uint64_t)s << (5 * (rank - 12));
            ++rank;
          }
          code = (code + c) << 1;
        }
        if (left != 0) { finish(ST_BAD_CODES); return; }
      }
      const uint32_t total = nlit + ndist;
      uint32_t i = 0, prev = 0;
      while (i < total) {
        br.refill();
        uint32_t code = 0, sym = 32;
        for (uint32_t l = 1; l < 8; ++l) {
          code = (code << 1) | ((uint32_t)(br.bb >> (l - 1)) & 1u);
          const uint32_t k = code - ((uint32_t)(firsts >> (8 * l)) & 0xffu);
          if (k < ((uint32_t)(cnts >> (8 * l)) & 0xffu)) {
            const uint32_t r = ((uint32_t)(offs >> (8 * l)) & 0xffu) + k;
            sym = (uint32_t)((r < 12 ? sorted_lo >> (5 * r) : sorted_hi >> (5 * (r - 12))) & 31u);
            br.drop(l);
            break;
          }
        }
        if (sym == 32) { finish(ST_BAD_CODES); return; }
        uint32_t rep = 1, val = sym;
        if (sym == 16) {
          if (i == 0) { finish(ST_BAD_CODES); return; }
          rep = 3 + br.bits(2);
          val = prev;
        } else if (sym == 17) {
          rep = 3 + br.bits(3);
          val = 0;
        } else if (sym == 18) {
          rep = 11 + br.bits(7);
          val = 0;
        }
        if (i + rep > total) { finish(ST_BAD_CODES); return; }
        if (sym != 16) prev = val;
        for (uint32_t j = 0; j < rep; ++j) scr->cl[i + j] = (uint8_t)val;
        i += rep;
      }
      if (br.pos() > nbits) { finish(cut()); return; }
      if (scr->cl[256] == 0) { finish(ST_BAD_CODES); return; }
    }
    uint32_t rc = build(0, 0, nlit);
    if (!rc) rc = build(1, nlit, ndist);
    if (rc) { finish(rc); return; }
    state = LS_SYM;
  }

  MGI_HD void end_of_block() {
    if (bfinal) { state = LS_MEMBER; return; }
    state = LS_BLOCK;
  }

  // behind a member's last block: the trailer, what follows
  MGI_HD void member() {
    br.drop(br.bc & 7u);
    br.refill();
    if (br.pos() + 64 > nbits) { finish(cut()); return; }
    crc = br.bits(32);
    br.refill();
    isize = br.bits(32);
    if (job.flags & F_ONE_MEMBER) { finish(ST_MEMBER); return; }
    ++nev;
    uint32_t at;
#if defined(__HIP_DEVICE_COMPILE__)
    at = __hip_atomic_fetch_add(nevents, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    at = (*nevents)++;
#endif
    if (at >= max_events) { finish(ST_EVENTS_FULL); return; }
    events[at].out_pos = outn;
    events[at].job = jobidx;
    events[at].crc = crc;
    events[at].isize = isize;
    events[at].pad = 0;
    uint64_t data = 0;
    const uint32_t h = gzip_header<false>(in, nbytes, br.pos() >> 3, &data);
    if (h == 1) { finish((br.pos() >> 3) >= nbytes && !input_final ? (uint32_t)ST_NEED_MORE : (uint32_t)ST_END); return; }
    if (h == 2) { finish(cut()); return; }
    floor = (int64_t)outn;
    br.init(in, (nbytes + 3) / 4, data * 8);
    state = LS_BLOCK;
  }

  MGI_HD void room(uint32_t n) {
    if (!count_only && outn + n > job.out_cap) { overflow = 1; count_only = true; }
  }

  // one symbol
  MGI_HD void symbol() {
    br.refill();
    if (br.pos() > nbits) { finish(cut()); return; }
    uint32_t c = mem->lit[(uint32_t)br.bb & ((1u << LLB) - 1u)];
    if (c & 0x8000u) c = slow(0);
    const uint32_t l = c & 15u;
    if (!l) { finish(ST_BAD_SYMBOL); return; }
    br.drop(l);
    const uint32_t sym = (c >> 4) & 511u;
    if (sym < 256) {
      room(1);
      if (!count_only) out[outn] = (OutT)sym;
      ++outn;
      return;
    }
    if (sym == 256) { end_of_block(); return; }
    if (sym > 285) { finish(ST_BAD_SYMBOL); return; }
    const uint32_t b = len_sym(sym - 257);
    const uint32_t len = (b & 0xffffu) + br.bits(b >> 16);
    br.refill();
    uint32_t d = mem->dist[(uint32_t)br.bb & ((1u << LDB) - 1u)];
    if (d & 0x8000u) d = slow(1);
    const uint32_t dl = d & 15u, ds = (d >> 4) & 511u;
    if (!dl || ds > 29) { finish(ST_BAD_SYMBOL); return; }
    br.drop(dl);
    const uint32_t db = dist_sym(ds);
    const uint32_t dist = (db & 0xffffu) + br.bits(db >> 16);
    if ((int64_t)outn - (int64_t)dist < floor) { finish(ST_BAD_DIST); return; }
    cp_base = outn;
    cp_dist = dist;
    cp_len = len;
    cp_i = 0;
    cp_o = 0;
    state = LS_COPY;
  }

  // up to 16 elements of the match
  MGI_HD void copy() {
    const uint32_t left = cp_len - cp_i, n = left < 16u ? left : 16u;
    room(n);
    if (!count_only) {
      const int64_t q0 = (int64_t)cp_base - (int64_t)cp_dist + (int64_t)cp_i;
      OutT* dst = out + cp_base + cp_i;
      if (cp_dist >= 16 && q0 >= 0 && outn + 16 <= job.out_cap) {
        // sixteen elements as ONE wide access each way (a lane's accesses are its own cache lines: sixteen narrow ones per lane were a
        // thousand requests per round and wavefront, and the round waited for them).  The source lies at least 16 elements back, so it
        // is complete — written by earlier rounds where the match overlaps itself; what is stored beyond the n elements is overwritten
        // by what the job produces next (or lies behind its end, inside its own reservation).
        OutT t[16];
        __builtin_memcpy(t, out + q0, sizeof(t));
        __builtin_memcpy(dst, t, sizeof(t));
        cp_o += n;
        if (cp_o >= cp_dist) cp_o -= cp_dist;
      } else if (cp_dist == 1 && cp_base >= 1 && outn + 16 <= job.out_cap) {  // a run of one element
        const OutT x = out[cp_base - 1];
        OutT t[16];
        MGI_UNROLL
        for (uint32_t i = 0; i < 16; ++i) t[i] = x;
        __builtin_memcpy(dst, t, sizeof(t));
      } else {
        uint32_t v[16];
        uint32_t o = cp_o;
        MGI_UNROLL
        for (uint32_t i = 0; i < 16; ++i) {
          if (i < n) {
            const int64_t q = (int64_t)cp_base - (int64_t)cp_dist + (int64_t)o;
            v[i] = q >= 0 ? (uint32_t)out[q] : (0x8000u | (uint32_t)(q + (int64_t)kWindow));
            ++o;
            if (o == cp_dist) o = 0;  // (an overlapping match repeats its first dist elements: they were there before it began)
          }
        }
        MGI_UNROLL
        for (uint32_t i = 0; i < 16; ++i)
          if (i < n) dst[i] = (OutT)v[i];
        cp_o = o;
      }
    }
    cp_i += n;
    outn = cp_base + cp_i;
    if (cp_i == cp_len) state = LS_SYM;
  }

  MGI_HD void stored() {
    const uint32_t n = st_left < 16u ? st_left : 16u;
    room(n);
    if (!count_only)
      for (uint32_t i = 0; i < n; ++i) out[outn + i] = (OutT)in_byte<false>(in, st_pos + i);
    outn += n;
    st_pos += n;
    st_left -= n;
    if (st_left == 0) {
      br.init(in, (nbytes + 3) / 4, st_pos * 8);
      end_of_block();
    }
  }

  MGI_HD void round() {
    ++rounds;
    if (state == LS_BLOCK) block();
    if (state == LS_MEMBER) member();
    if (state == LS_STORED) stored();
    if (state == LS_SYM) symbol();
    if (state == LS_COPY) copy();
  }

  MGI_HD void result(Result* res) const {
    res->end_bit = br.pos();
    res->out_count = outn;
    res->status = status;
    res->overflow = overflow;
    res->crc = crc;
    res->isize = isize;
    res->nblocks = nblocks;
    res->nevents = nev;
    res->t_tab = res->t_dec = res->t_emit = res->t_tail = 0;
    res->nbatch = 0;
    res->nstep = rounds;
    for (int i = 0; i < 6; ++i) res->t_sub[i] = 0;
  }
};

// ---- block-start finder ----
// Could a dynamic-Huffman block with BFINAL = 0 begin at bit p?  The cheap part, per lane: header fields in range and the
// code-length code a complete prefix code (what zlib, pigz, libdeflate write; a false negative only costs parallelism).
// lo / hi: the 128 bits that start at the candidate position
// strict: the header must also be one an ENCODER writes — zlib, pigz, libdeflate, bgzip and igzip all send as few lengths as they
// can, so the last length of each of the three lists is not zero (unless the list has its minimum size).  The format does not ask
// for it: a true start written otherwise is only found with strict = false (the host falls back to that when a stage finds next to
// nothing).  It is what keeps FALSE starts rare — a false start in front of the true one of its chunk hides it, and the text between
// the two jobs around it then takes a launch of its own (measured: 3 in 16 000 chunks without this rule, ~4 ms each).
// (in two parts: one position in nine passes the first, and the finder runs the second — fifty times the instructions — only on
// those, sixty-four of them at a time)
MGI_HD bool probe_fields(uint64_t lo) {
  // BFINAL = 0, BTYPE = 2 (bits 1-2, LSB first); at most 286 literal/length and 30 distance codes
  return (lo & 7u) == 4u && ((lo >> 3) & 31u) <= 29u && ((lo >> 8) & 31u) <= 29u;
}
// probe_fields for 32 positions at once: bit i of the result = probe_fields(x >> i), x = the 64 bits at the first position (bits
// i .. i + 12 are looked at).  Every test is a conjunction of single bits, so the words shifted against each other do all positions
// together: twenty instructions for 32 positions where one position took twelve.
MGI_HD uint32_t probe_fields_mask(uint64_t x) {
  auto b = [&](uint32_t k) -> uint32_t { return (uint32_t)(x >> k); };
  const uint32_t type = ~b(0) & ~b(1) & b(2);              // BFINAL = 0, BTYPE = 2
  const uint32_t hlit = b(4) & b(5) & b(6) & b(7);         // HLIT  (bits 3..7)  = 30 or 31
  const uint32_t hdist = b(9) & b(10) & b(11) & b(12);     // HDIST (bits 8..12) = 30 or 31
  return type & ~hlit & ~hdist;
}
MGI_HD bool probe_code_lengths(uint64_t lo, uint64_t hi, bool strict = true) {
  const uint32_t ncl = (uint32_t)((lo >> 13) & 15u) + 4u;
  if (strict && ncl > 4u && ((((lo >> 17) | (hi << 47)) >> (3u * (ncl - 1u))) & 7u) == 0u) return false;
  // the 3-bit lengths as three words of up to eight (24 bits each), those behind the ncl-th zeroed; a length l weighs
  // 64 >> (l - 1), and l = 0 shifts everything out (the shift count wraps to 31)
  const uint64_t all = ((lo >> 17) | (hi << 47)) & ((1ull << (3u * ncl)) - 1ull);
  const uint32_t f[3] = {(uint32_t)all & 0xffffffu, (uint32_t)(all >> 24) & 0xffffffu, (uint32_t)(all >> 48)};
  uint32_t kraft = 0;
  for (uint32_t w = 0; w < 3; ++w)
    for (uint32_t i = 0; i < (w < 2 ? 8u : 3u); ++i) kraft += 64u >> ((((f[w] >> (3u * i)) & 7u) - 1u) & 31u);
  return kraft == 128u;
}
MGI_HD bool probe_bits(uint64_t lo, uint64_t hi, bool strict = true) { return probe_fields(lo) && probe_code_lengths(lo, hi, strict); }
MGI_HD bool probe_block_start(const uint32_t* in, uint64_t nwords, uint64_t p, bool strict = true) {
  const uint64_t w = p >> 5;
  const uint32_t s = (uint32_t)p & 31u;
  uint32_t a[4];
  for (uint32_t i = 0; i < 4; ++i) a[i] = w + i < nwords ? in[w + i] : 0u;
  const uint64_t x0 = a[0] | (uint64_t)a[1] << 32, x1 = a[2] | (uint64_t)a[3] << 32;
  return probe_bits(s ? (x0 >> s) | (x1 << (64 - s)) : x0, x1 >> s, strict);
}
// ... and the whole header (uniform): the code lengths decode, both codes are complete (or there is at most one distance code)
template <class Exec>
MGI_HD bool validate_block_start(Exec& ex, Shared& sh, const uint32_t* in, uint64_t nbytes, uint64_t p, bool strict = true) {
  BitReader br;
  br.init(in, (nbytes + 3) / 4, p);
  br.refill();
  br.drop(3);
  uint32_t nlit = 0, ndist = 0;
  if (read_dynamic_lengths(ex, sh, br, &nlit, &ndist)) return false;
  if (br.pos() > nbytes * 8) return false;
  if (strict) {
    const uint32_t last_lit = MGI_UNI(sh.cl[nlit - 1]), last_dist = MGI_UNI(sh.cl[nlit + ndist - 1]);
    if ((nlit > 257 && !last_lit) || (ndist > 1 && !last_dist)) return false;
  }
  if (build_table(ex, sh, 0, 0, nlit, false, true)) return false;
  if (build_table(ex, sh, 1, nlit, ndist, false, true)) return false;
  return true;
}

// The same decision as validate_block_start by ONE lane, in registers only (the finder validates up to 64 candidates at once): the
// code-length code decoded canonically (at most seven compares per symbol), the Kraft sums of both codes kept as the lengths arrive —
// an over-subscribed code is refused at once.
MGI_HD bool light_validate(const uint32_t* in, uint64_t nbytes, uint64_t p, bool strict = true) {
  BitReaderT<false> br;
  br.init(in, (nbytes + 3) / 4, p);
  br.refill();
  br.drop(3);
  const uint32_t nlit = br.bits(5) + 257, ndist = br.bits(5) + 1, ncl = br.bits(4) + 4;
  if (nlit > 286 || ndist > 30) return false;
  const uint64_t order_lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 |
                            5ull << 45 | 11ull << 50 | 4ull << 55;
  const uint64_t order_hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
  uint64_t plen = 0, cnts = 0;  // 3 bits per symbol; 8 bits per length
  for (uint32_t i = 0; i < ncl; ++i) {
    br.refill();
    const uint32_t l = br.bits(3);
    const uint32_t sym = (uint32_t)((i < 12 ? order_lo >> (5 * i) : order_hi >> (5 * (i - 12))) & 31u);
    plen |= (uint64_t)l << (3 * sym);
    cnts += 1ull << (8 * l);
  }
  // canonical code of the code-length code: first code and first rank per length, the symbols in (length, symbol) order
  uint64_t firsts = 0, offs = 0, sorted_lo = 0, sorted_hi = 0;  // 8 bits per length; 5 bits per rank
  {
    int32_t left = 1;
    uint32_t code = 0, rank = 0;
    for (uint32_t l = 1; l < 8; ++l) {
      const uint32_t c = (uint32_t)(cnts >> (8 * l)) & 0xffu;
      left = (left << 1) - (int32_t)c;
      if (left < 0) return false;
      firsts |= (uint64_t)code << (8 * l);
      offs |= (uint64_t)rank << (8 * l);
      for (uint32_t s = 0; s < 19; ++s) {
        if (((uint32_t)(plen >> (3 * s)) & 7u) != l) continue;
        if (rank < 12) sorted_lo |= (uint64_t)s << (5 * rank); else sorted_hi |= (uint64_t)s << (5 * (rank - 12));
        ++rank;
      }
      code = (code + c) << 1;
    }
    if (left != 0) return false;
  }
  const uint32_t total = nlit + ndist;
  uint32_t i = 0, prev = 0, eob = 0;
  uint32_t kl = 0, kd = 0, nd = 0, d1 = 0;  // Kraft sums in units of 2^-15; distance codes used, of one bit
  while (i < total) {
    br.refill();
    uint32_t code = 0, sym = 32;
    for (uint32_t l = 1; l < 8; ++l) {
      code = (code << 1) | ((uint32_t)(br.bb >> (l - 1)) & 1u);
      const uint32_t k = code - ((uint32_t)(firsts >> (8 * l)) & 0xffu);
      if (k < ((uint32_t)(cnts >> (8 * l)) & 0xffu)) {
        const uint32_t r = ((uint32_t)(offs >> (8 * l)) & 0xffu) + k;
        sym = (uint32_t)((r < 12 ? sorted_lo >> (5 * r) : sorted_hi >> (5 * (r - 12))) & 31u);
        br.drop(l);
        break;
      }
    }
    if (sym == 32) return false;
    uint32_t rep = 1, val = sym;
    if (sym == 16) {
      if (i == 0) return false;
      rep = 3 + br.bits(2);
      val = prev;
    } else if (sym == 17) {
      rep = 3 + br.bits(3);
      val = 0;
    } else if (sym == 18) {
      rep = 11 + br.bits(7);
      val = 0;
    }
    if (i + rep > total) return false;
    if (sym != 16) prev = val;
    if (strict && !val && ((i < nlit && nlit <= i + rep && nlit > 257) || (i + rep == total && ndist > 1))) return false;  // (a last length of zero)
    if (val) {
      // (a run may cross from the literal/length lengths into the distance lengths)
      const uint32_t a = i < nlit ? (i + rep < nlit ? rep : nlit - i) : 0u;
      kl += a * (32768u >> val);
      kd += (rep - a) * (32768u >> val);
      nd += rep - a;
      if (val == 1) d1 += rep - a;
      if (i <= 256 && 256 < i + rep) eob = 1;
      if (kl > 32768u || kd > 32768u) return false;
    }
    i += rep;
  }
  if (br.pos() > nbytes * 8 || !eob) return false;
  if (kl != 32768u) return false;                // the literal/length code must be complete
  if (kd != 32768u && nd > 1) return false;      // the distance code complete, or a single code (of one bit: zlib's rule)
  if (kd != 32768u && nd == 1 && d1 != 1) return false;
  return true;
}

// ---- CRC-32 (gzip): combination of the CRCs of adjacent pieces, as zlib's crc32_combine does it since 1.2.12 ----
constexpr uint32_t kCrcPoly = 0xedb88320u;
MGI_HD uint32_t crc_multmodp(uint32_t a, uint32_t b) {  // a(x) * b(x) mod p(x), reflected
  uint32_t m = 1u << 31, p = 0;
  for (;;) {
    if (a & m) {
      p ^= b;
      if ((a & (m - 1)) == 0) break;
    }
    m >>= 1;
    b = b & 1 ? (b >> 1) ^ kCrcPoly : b >> 1;
  }
  return p;
}
// x^(8 n) mod p(x): x2n[k] = x^(2^k) mod p
MGI_HD uint32_t crc_x8n(uint64_t nbytes, const uint32_t* x2n) {
  uint32_t p = 1u << 31;
  uint32_t k = 3;
  while (nbytes) {
    if (nbytes & 1) p = crc_multmodp(x2n[k & 31u], p);
    nbytes >>= 1;
    ++k;
  }
  return p;
}
MGI_HD void crc_make_x2n(uint32_t* x2n) {
  uint32_t p = 1u << 30;  // x^1
  x2n[0] = p;
  for (uint32_t n = 1; n < 32; ++n) x2n[n] = p = crc_multmodp(p, p);
}
// crc of A || B from crc(A), crc(B), len(B)
MGI_HD uint32_t crc_combine(uint32_t crc1, uint32_t crc2, uint64_t len2, const uint32_t* x2n) {
  return crc_multmodp(crc_x8n(len2, x2n), crc1) ^ crc2;
}

}  // namespace mgi
