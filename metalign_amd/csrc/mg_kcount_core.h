// mg_kcount_core.h — stage A of the reference pipeline BY K-MER IDENTITY: the per-lane pieces of k_count_kmers (mg_kcount.hip).
//
// What it replaces: `kmc -k<kmax> -ci2 -cs3` over the reads + `kmc_tools simple ... intersect` with the k_max-mers of the genome
// sketches (scripts/select_db.py:50-59).  KMC counts canonical k-mers and intersects k-mer SETS: nothing on the read side is
// hashed.  So the read side here hashes nothing either: a read k-mer is looked up among the table's sketched k-mers by what it IS.
//
// How: minimizer partitioning (KMC's own signatures are the same idea).  The CANDIDATES of a k-mer are its m-mers (m = 15) that
// have e more bases of the k-mer on either side (e = kc_flank(k): 2 from k = 23 on) — w = k - 14 - 2 e of them; every one has a
// RANK (22 bits of an odd multiplier over the lexicographically smaller strand of the m-mer: the same on either strand), and the
// k-mer's minimizer is a candidate of the smallest rank.  What a k-mer is FILED under is not the 15-mer but the (15 + 2 e)-mer
// around it — 19 bases — hashed to 32 bits (the smaller strand: kc_ext_hash): a table of ten million k-mers holds a sixth of
// all the 15-mers that ever come out smallest of 37 (minima crowd at the bottom of their range), and one read run in six then
// met some unrelated k-mer's 15-mer; with the flanks it is one in a hundred thousand.  Equal k-mers have equal sets of
// smallest-rank candidates: the table's distinct canonical k-mers are filed once under each of theirs (nearly always one; a
// tandem repeat has several — mg_refdb_index_kmers), a read's windows are cut into RUNS of consecutive windows that share
// their minimizer (the leftmost smallest: about w / 2 windows each), and whichever candidate a window chose, the table k-mer
// equal to it is filed there.  Per run: one bit of a gate bitmap over the leading bits of the hash; for the few runs that pass,
// the bucket of table k-mers filed under that hash, each compared against the run's windows (a 16-base signature first, the
// whole canonical k-mer on a signature hit); a match adds one to the k-mer's counter.
//
// The sliding minimum is van Herk / Gil-Werman in registers: m-mers in blocks of w; a window that starts in block b and ends
// in block b + 1 has min(suffix minimum of block b from its first m-mer on, prefix minimum of block b + 1 up to its last):
// one v_min for the prefix, one for the combination, one per m-mer for the suffix minima after the block — no deque, no
// branch, every index a compile-time constant (the block loop is unrolled, the block's keys live in w registers).
//
// Host / device, one source: everything here is per-lane code without cross-lane operations; tests/host_kcount_check.cpp
// compiles it with g++ (MG_HOST_CHECK) and holds it to the oracle where there is no GPU (tests/test_kcount_core_host.py).
#pragma once
#include <cstdint>
#include <utility>

#ifdef MG_HOST_CHECK
#define MG_LDS
#define MG_GLB
#define MG_HD static inline
#define MG_UNIFORM(x) (x)
#else
#include <hip/hip_runtime.h>
#define MG_LDS __attribute__((address_space(3)))
#define MG_GLB __attribute__((address_space(1)))  // device memory, said in the type: a generic pointer into a CALLED function is a FLAT access
#define MG_HD __device__ __forceinline__
#define MG_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))  // the same in every lane: say so (an SGPR, scalar branches)
#endif

// the sample's words that many lanes update (counters, saturation bits, the live gate)
#ifdef MG_HOST_CHECK
#define MG_KC_ADD(ptr, v) ((*(ptr) += (v)) - (v))
#define MG_KC_OR(ptr, v) (*(ptr) |= (v))
#define MG_KC_AND(ptr, v) (*(ptr) &= (v))
#define MG_KC_LOAD(ptr) (*(ptr))
#define MG_KC_CAS(ptr, expected, desired) (*(ptr) = (*(ptr) == (expected)) ? (desired) : *(ptr))
#else
// (a sample's bits are set by lanes all over the device; each XCD's L2 keeps what it has read: a plain load may see a word as it
// was long ago — and a k-mer that is saturated be scanned for again and again.  Device-scope loads for those words.)
#ifdef MG_KC_PLAIN_LOADS  // (A/B builds)
#define MG_KC_LOAD(ptr) (*(ptr))
#else
#define MG_KC_LOAD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
#define MG_KC_ADD(ptr, v) __hip_atomic_fetch_add((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MG_KC_OR(ptr, v) __hip_atomic_fetch_or((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MG_KC_AND(ptr, v) __hip_atomic_fetch_and((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// (nobody asks whether it took: a lost exchange loses a HINT — see kc_count_entry)
#define MG_KC_CAS(ptr, expected, desired)                                                                                          \
  do {                                                                                                                             \
    uint32_t mg_kc_expected_ = (expected);                                                                                         \
    (void)__hip_atomic_compare_exchange_strong((ptr), &mg_kc_expected_, (desired), __ATOMIC_RELAXED, __ATOMIC_RELAXED,             \
                                               __HIP_MEMORY_SCOPE_AGENT);                                                          \
  } while (0)
#endif

namespace mg {

constexpr int kKcM = 15;                     // minimizer length: odd (no m-mer is its own reverse complement), 30 bits
constexpr uint32_t kKcMask = 0x3fffffffu;
constexpr uint32_t kKcNone = 0xffffffffu;    // "no key": above every key, so it is also +infinity of the minima
constexpr uint32_t kKcXor = 0x1b873593u & kKcMask;  // (poly-A is m-mer 0: without this its rank would be the smallest of all)
constexpr uint32_t kKcMul = 0x2545f491u;     // odd
constexpr uint32_t kKcPos = 1023u;           // the low ten bits of a candidate's word: where it starts
constexpr int kKcMinK = kKcM, kKcMaxK = 64;
constexpr int kKcMaxCands = 50;              // candidates of a k-mer: k - 14 at most
constexpr uint32_t kKcSeveral = 0xffu;       // KcEntry::off of an entry whose hash belongs to several candidates of its k-mer
constexpr uint32_t kKcSlots = 4;             // entries of a bucket that sit in its own line (KcIndexView)
constexpr uint32_t kKcMaxRead = 1023;        // window numbers of an event take ten bits: longer reads go through in chunks

// bases of the k-mer that a candidate has on either side, and how many candidates that leaves
// (two from k = 23 on; from k = 53 on as many as keep the candidates at 33 or 34: the walk holds a block of candidates in registers, and
// with the 42 of k = 60 under two flank bases it spilled inside its loop — 2.26 ms per 10M reads against 2.08 at k = 51; the wider
// flanks cost a fifth more runs and nothing else: the filed bases, 15 + 2 e <= 31, still come out of one 64-bit read of the stream)
constexpr int kc_flank(int k) { return k >= 53 ? 2 + (k - 51) / 2 : (k >= 23 ? 2 : (k >= 19 ? 1 : 0)); }
constexpr int kc_cands(int k) { return k - kKcM + 1 - 2 * kc_flank(k); }

// the WORD of a candidate given both strands of its m-mer 2-bit packed (first base most significant, right-aligned in 30 bits)
// and its number `pos` (< 1023): its rank above its number — the smallest word of a window is its leftmost candidate of the
// smallest rank, and says where it is.
// (The rank is the first eleven bases of the smaller strand under a fixed relabelling of the bases — xor with a constant: three
// operations.  An odd multiplier over all fifteen spreads the ranks better, but v_mul_lo_u32 runs at a quarter of the rate and this
// is computed once per base of every read; candidates that tie on eleven bases are told apart by their numbers, and the table
// files a k-mer under every one of its smallest.)
MG_HD uint32_t kc_word(uint32_t f, uint32_t r, uint32_t pos) {
  const uint32_t c = f < r ? f : r;
  return (((c ^ kKcXor) >> 8) << 10) | pos;
}
// what a k-mer is filed under: v = 64 bits of bases from the candidate's first flank base on (the 15 + 2 e bases at the top);
// the smaller strand of those bases, mixed to 32 bits.  Never kKcNone (an unused slot of a bucket has that).
MG_HD uint32_t kc_rc32(uint32_t x);
MG_HD uint32_t kc_ext_hash(uint64_t v, int k) {
  const int nb = 2 * (kKcM + 2 * kc_flank(k));
  const uint64_t f = v >> (64 - nb);
  const uint64_t r = ((((uint64_t)kc_rc32((uint32_t)v)) << 32) | kc_rc32((uint32_t)(v >> 32))) & ((1ull << nb) - 1ull);
  const uint64_t c = f < r ? f : r;  // (38 bits up to k = 52, 62 at most: what lies above the low word goes in through 24-bit multiply-adds, full rate)
  const uint32_t hi = (uint32_t)(c >> 32);
#ifdef MG_HOST_CHECK
  uint32_t h = (uint32_t)c * 0x9E3779B1u + (hi & 0xffffffu) * 0x85EBCBu + (hi >> 24) * 0xC2B2AFu;
#else
  uint32_t h = (uint32_t)c * 0x9E3779B1u + __umul24(hi & 0xffffffu, 0x85EBCBu) + __umul24(hi >> 24, 0xC2B2AFu);
#endif
  h ^= h >> 15; h *= 0x2C1B3C6Du;  // (two quarter-rate multiplies in all: this runs once per run of every read)
  h ^= h >> 13;
  return h & 0xfffffffeu;  // (never kKcNone: one AND instead of a compare and a select)
}
// bits of the gate bitmap for a table of nd distinct k-mers: 2^extra bits per k-mer (one run in 2^extra that has nothing to find
// passes), between 2^16 and 2^32
constexpr uint32_t kKcGateExtra = 5;  // (2^5 bits per k-mer: a run in 32 that has nothing to find passes; 6 measured the same kernel and a pass 2.5 % slower — the sample's copy of the gate is made anew every pass)
constexpr uint32_t kc_gate_bits(uint64_t nd, uint32_t extra) {
  uint32_t b = 0;
  while (b < 32u && (1ull << b) < nd) ++b;
  b += extra;
  return b < 16u ? 16u : (b > 32u ? 32u : b);
}

// ---- the staged tile: a big-endian 2-bit base stream -----------------------------------------------------------------
// Base p of the stream sits in dword p >> 4 at bits [30 - 2 (p & 15), +2): the first base of a dword is its most significant
// pair, so 32 bits taken at any base offset ARE sixteen bases packed first-base-most-significant — what k-mers compare by.
// (The reverse strand of a window is computed from the window itself, kc_revcomp, and only where a signature has matched.)
// `inv` is one bit per base (bit 31 - (p & 31) of dword p >> 5): set = not one of ACGTacgt.

// 32 bits of a 2-bit stream from base p on (sixteen bases)
MG_HD uint32_t kc_ext32(const MG_LDS uint32_t* s, uint32_t p) {
  const uint32_t d = p >> 4, sh = (p & 15u) << 1;
  const uint64_t v = ((uint64_t)s[d] << 32) | s[d + 1];
  return (uint32_t)((v << sh) >> 32);
}
// 32 bits of a 1-bit stream from position p on
MG_HD uint32_t kc_bits32(const MG_LDS uint32_t* s, uint32_t p) {
  const uint32_t d = p >> 5, sh = p & 31u;
  const uint64_t v = ((uint64_t)s[d] << 32) | s[d + 1];
  return (uint32_t)((v << sh) >> 32);
}

// 64 bits of a 2-bit stream from base p on (thirty-two bases)
MG_HD uint64_t kc_ext64(const MG_LDS uint32_t* s, uint32_t p) {
  const uint32_t d = p >> 4, sh = (p & 15u) << 1;
  const uint32_t a = s[d], b = s[d + 1], c = s[d + 2];
#ifdef MG_HOST_CHECK
  const uint32_t hi = (uint32_t)(((((uint64_t)a) << 32 | b) << sh) >> 32), lo = (uint32_t)(((((uint64_t)b) << 32 | c) << sh) >> 32);
#else
  // (two funnel shifts — v_alignbit_b32 takes (hi : lo) >> n for n in 0 .. 31 — and a select for sh = 0, instead of two 64-bit shifts)
  const uint32_t n = (sh ^ 31u) + 1u;  // 32 - sh for sh in 1 .. 31 (written so that the compiler does not make a multiply of it)
  const uint32_t hi = sh ? __builtin_amdgcn_alignbit(a, b, n) : a, lo = sh ? __builtin_amdgcn_alignbit(b, c, n) : b;
#endif
  return ((uint64_t)hi << 32) | lo;
}
// the hash a closed run is looked up by (pos: where its candidate starts in the read that starts at stream position p0)
MG_HD uint32_t kc_run_hash(const MG_LDS uint32_t* fwd, uint32_t p0, uint32_t pos, int k) {
  return kc_ext_hash(kc_ext64(fwd, p0 + pos), k);
}
// An EVENT = a closed run in 32 bits: first window | last window << 10 | its candidate's number << 20 (the two bits above: whatever
// the word's rank left there).  A run of windows that have no minimizer (word kKcNone) says candidate 1023: no event.  The walk
// leaves the first window out (info = last window << 10): a lane's runs follow one another, so whoever reads its list fills it
// in — the window after the previous event's last, or the window the walk started at (kc_event_first).
MG_HD uint32_t kc_event(uint32_t word, uint32_t info) { return (word << 20) | info; }
MG_HD uint32_t kc_event_first(uint32_t ev, uint32_t& next) {  // next: in, this run's first window; out, the following run's
  const uint32_t out = (ev & ~1023u) | next;
  next = ((ev >> 10) & 1023u) + 1u;
  return out;
}
MG_HD bool kc_event_none(uint32_t ev) { return ((ev >> 20) & kKcPos) == kKcPos; }
MG_HD uint32_t kc_event_pos(uint32_t ev) { return (ev >> 20) & kKcPos; }

struct KcWin { uint32_t w[4]; };  // a k-mer, LEFT-aligned: base 0 in the top pair of w[0]; bits below 2k are zero

MG_HD uint32_t kc_keep_mask(int k, int word) {  // the bits of dword `word` that belong to a left-aligned k-mer
  const int nb = 2 * k - 32 * word;
  return nb <= 0 ? 0u : (nb >= 32 ? 0xffffffffu : 0xffffffffu << (32 - nb));
}

MG_HD KcWin kc_ext128(const MG_LDS uint32_t* s, uint32_t p, int k) {
  const uint32_t d = p >> 4, sh = (p & 15u) << 1;
  uint32_t a[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) a[i] = s[d + i];
  KcWin x;
#pragma unroll
  for (int i = 0; i < 4; ++i) x.w[i] = (uint32_t)(((((uint64_t)a[i]) << 32 | a[i + 1]) << sh) >> 32) & kc_keep_mask(k, i);
  return x;
}

MG_HD bool kc_less(const KcWin& a, const KcWin& b) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
  return a.w[3] < b.w[3];
}
MG_HD bool kc_equal(const KcWin& a, const KcWin& b) {
  return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2] && a.w[3] == b.w[3];
}

// the order of the sixteen 2-bit groups of a dword reversed
MG_HD uint32_t kc_rev2(uint32_t x) {
#ifdef MG_HOST_CHECK
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
  x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
  x = (x >> 16) | (x << 16);
#else
  x = __builtin_bitreverse32(x);  // v_bfrev_b32: every bit reversed ...
#endif
  return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);  // ... and the two bits of a group put back in order
}
// reverse complement of sixteen packed bases
MG_HD uint32_t kc_rc32(uint32_t x) { return kc_rev2(~x); }

// reverse complement of a left-aligned k-mer
MG_HD KcWin kc_revcomp(const KcWin& x, int k) {
  // all 64 groups reversed and complemented: the k-mer's groups end up at the BOTTOM of the 128 bits; shift them to the top
  const uint32_t r[4] = {kc_rc32(x.w[3]), kc_rc32(x.w[2]), kc_rc32(x.w[1]), kc_rc32(x.w[0])};
  const int s = 2 * (64 - k);  // left shift, 0 .. 98
  KcWin o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i + (s >> 5), sh = s & 31;
    const uint32_t hi = q < 4 ? r[q < 4 ? q : 3] : 0u, lo = q + 1 < 4 ? r[q + 1 < 4 ? q + 1 : 3] : 0u;
    o.w[i] = (sh ? (hi << sh) | (lo >> (32 - sh)) : hi) & kc_keep_mask(k, i);
  }
  return o;
}

// (hi, lo): a k-mer right-aligned in 128 bits, first base most significant (the table's kmer_hi / kmer_lo) -> left-aligned
MG_HD KcWin kc_from_right(uint64_t hi, uint64_t lo, int k) {
  const int s = 128 - 2 * k;  // 0 .. 98
  uint64_t h, l;
  if (s == 0) { h = hi; l = lo; }
  else if (s < 64) { h = (hi << s) | (lo >> (64 - s)); l = lo << s; }
  else { h = s == 64 ? lo : lo << (s - 64); l = 0; }
  return KcWin{{(uint32_t)(h >> 32), (uint32_t)h, (uint32_t)(l >> 32), (uint32_t)l}};
}

// the lexicographically smaller strand (KMC's canonical k-mer)
MG_HD KcWin kc_canonical(const KcWin& x, int k) {
  const KcWin r = kc_revcomp(x, k);
  return kc_less(r, x) ? r : x;
}

// m-mer number j (bases j .. j + 14) of a left-aligned k-mer, right-aligned in 30 bits
MG_HD uint32_t kc_mmer_at(const KcWin& x, int j) {
  const int d = j >> 4, sh = (j & 15) << 1;
  const uint64_t v = ((uint64_t)x.w[d] << 32) | (d + 1 < 4 ? x.w[d + 1 < 4 ? d + 1 : 3] : 0u);
  return (uint32_t)((v << sh) >> 34);
}
// reverse complement of a 15-mer (30 bits, right-aligned)
MG_HD uint32_t kc_mmer_rc(uint32_t f) { return kc_rc32(f << 2) & kKcMask; }

// 64 bits of a left-aligned k-mer from base j on (zeros beyond its 128 bits)
MG_HD uint64_t kc_sub64(const KcWin& x, int j) {
  const int d = j >> 4, sh = (j & 15) << 1;
  const uint32_t a = x.w[d < 4 ? d : 3], b = d + 1 < 4 ? x.w[d + 1 < 4 ? d + 1 : 3] : 0u, c = d + 2 < 4 ? x.w[d + 2 < 4 ? d + 2 : 3] : 0u;
  const uint32_t hi = (uint32_t)(((((uint64_t)a) << 32 | b) << sh) >> 32), lo = (uint32_t)(((((uint64_t)b) << 32 | c) << sh) >> 32);
  return ((uint64_t)hi << 32) | lo;
}

// What a table k-mer is filed under (the read side slides: kc_walk): the hash of EVERY candidate of the smallest rank — a read
// window equal to this k-mer chooses the leftmost of them as the window stands, the rightmost here if the read shows the other
// strand.  out[0 .. return): the distinct hashes, at most kKcMaxCands (nearly always one).
MG_HD int kc_table_keys(const KcWin& x, int k, uint32_t* out) {
  const int e = kc_flank(k), w = kc_cands(k);
  uint32_t best = kKcNone;
  for (int j = 0; j < w; ++j) {
    const uint32_t f = kc_mmer_at(x, j + e);
    const uint32_t rank = kc_word(f, kc_mmer_rc(f), 0u);
    best = rank < best ? rank : best;
  }
  int n = 0;
  for (int j = 0; j < w; ++j) {
    const uint32_t f = kc_mmer_at(x, j + e);
    if (kc_word(f, kc_mmer_rc(f), 0u) != best) continue;
    const uint32_t h = kc_ext_hash(kc_sub64(x, j), k);
    bool seen = false;
    for (int t = 0; t < n; ++t) seen = seen || out[t] == h;
    if (!seen) out[n++] = h;
  }
  return n;
}
// ... and which candidate of x the hash h (one of kc_table_keys') is of: a read window equal to x that chose candidate number c
// as ITS minimizer starts c windows before it (x's strand) or w - 1 - c (the other) — two windows to compare instead of a run.
// kKcSeveral when more than one candidate has this hash (a homopolymer, a tandem repeat: those runs are scanned whole).
MG_HD uint32_t kc_table_key_offset(const KcWin& x, int k, uint32_t h) {
  const int e = kc_flank(k), w = kc_cands(k);
  uint32_t best = kKcNone, off = kKcSeveral;
  for (int j = 0; j < w; ++j) {
    const uint32_t f = kc_mmer_at(x, j + e);
    const uint32_t rank = kc_word(f, kc_mmer_rc(f), 0u);
    best = rank < best ? rank : best;
  }
  int n = 0;
  for (int j = 0; j < w; ++j) {
    const uint32_t f = kc_mmer_at(x, j + e);
    if (kc_word(f, kc_mmer_rc(f), 0u) != best || kc_ext_hash(kc_sub64(x, j), k) != h) continue;
    off = (uint32_t)j;
    ++n;
  }
  return n == 1 ? off : kKcSeveral;
}

// One distinct canonical k-mer of the table, as the read side meets it (32 bytes, two 16-byte loads).
struct __attribute__((aligned(16))) KcEntry {
  uint32_t w[4];   // the canonical k-mer, left-aligned
  uint32_t head;   // where it is counted: the first pair of the hash-major table that holds it
  uint32_t key;    // what it is filed under here (kc_table_keys: a k-mer with several has an entry for each)
  uint32_t off;    // which of its candidates that hash is of (its number, 0 .. w - 1) — kKcSeveral: more than one of them
  uint32_t pad;    // (the first two entries of a bucket: its entries beyond the four of its line — how many, and where)
};

// ---- staging: sixteen ASCII bases -> one dword of the stream ----------------------------------------------------------
// v: the sixteen bytes as four little-endian dwords (byte 0 = the first base).  Returns the packed dword; *notbase gets a
// non-zero value when any byte is not one of ACGTacgt (per byte: bit 7 of the byte's lane in nz[i]).
MG_HD uint32_t kc_pack4(uint32_t x, uint32_t& nz) {
  const uint32_t u = x & 0xDFDFDFDFu;                       // upper case
  const uint32_t t = (x >> 1) & 0x03030303u;                // A:0 C:1 T:2 G:3
  const uint32_t c = t ^ ((t >> 1) & 0x01010101u);          // A:0 C:1 G:2 T:3
#ifdef MG_HOST_CHECK
  uint32_t letters = 0;
  for (int b = 0; b < 4; ++b) letters |= ((0x54474341u >> (8 * ((c >> (8 * b)) & 3u))) & 0xffu) << (8 * b);
#else
  const uint32_t letters = __builtin_amdgcn_perm(0u, 0x54474341u, c);  // "ACGT"[code] per byte
#endif
  nz = letters ^ u;                                          // a byte that is not its own code's letter is no base
  // the four codes (byte b holds base b) -> eight bits, base 0 most significant: one multiply gathers them (the partial
  // products land in disjoint bit pairs, nothing carries)
  return (c * 0x40100401u) >> 24;
}
MG_HD uint32_t kc_pack16(const uint32_t v[4], uint32_t& notbase) {
  uint32_t n0, n1, n2, n3;
  const uint32_t p = (kc_pack4(v[0], n0) << 24) | (kc_pack4(v[1], n1) << 16) | (kc_pack4(v[2], n2) << 8) | kc_pack4(v[3], n3);
  notbase = n0 | n1 | n2 | n3;
  return p;
}
// the same sixteen bytes -> sixteen "not a base" bits, the first base most significant
MG_HD uint32_t kc_notbase16(const uint32_t v[4]) {
  uint32_t out = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t nz;
    (void)kc_pack4(v[i], nz);
    const uint32_t b = ((((nz & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | nz) >> 7) & 0x01010101u;  // 1 per non-zero byte
    out = (out << 4) | (((b * 0x08040201u) >> 24) & 0xFu);
  }
  return out;
}

// ---- the read side: runs of windows that share their minimizer -------------------------------------------------------
// An event = a closed run: (word, first window | last window << 10), kept as kc_event(word, info); word == kKcNone: nothing
// (skipped by whoever reads the list).  Out: put(slot, word, info) stores the run a lane is ABOUT to close in its list (every step, at the slot after its closed
// ones: whether the run closes is known one compare later — an unconditional store and an add-with-carry instead of a branch);
// a list has kCap slots and one more that takes the stores of a lane whose list is full; any_full(cnt), wave_min(x): the two
// things the walk asks of the whole wavefront, once per block and once per call.
//
// The walk starts at window w0 (wave-uniform) and goes on to the end of the tile's longest read — or to the end of the block in
// which some lane's list filled up.  It returns the window to go on from: the first it has not walked, or the first that a full
// lane has not RECORDED (the window after its last event); whoever empties the lists drops what lies at or beyond that window
// (other lanes may have recorded further: they record it again).  Nothing else is carried from one call to the next: the
// sliding minimum is primed anew (k - 1 bases: what a restart costs) and a run that straddles the restart is two events with
// one key — every window is still in exactly one event.  So whatever empties the lists runs where nothing of the walk is live.
//
// MODE 0: any tile (bases that are no bases: an m-mer over one has no key, a run of such windows is no event; windows over
//         one are sorted out when a run is matched).  1: no such base in the tile, every read as long as the longest.
//         2: no such base, ragged lengths.
// fwd / inv: the staged tile; p0: stream position of this lane's first base; len: its read's length; maxlen: the longest of
// the tile (wave-uniform: so are the loop bounds and the stream's refill points).
template <int K, int MODE, class Out>
MG_HD uint32_t kc_walk(const MG_LDS uint32_t* fwd, const MG_LDS uint32_t* inv, uint32_t p0, uint32_t len, uint32_t maxlen, uint32_t w0,
                       Out& out, uint32_t& cnt) {
  constexpr int M = kKcM, E = kc_flank(K), W = kc_cands(K);  // (candidate j: bases j + E .. j + E + 14 of the read; window i has j = i .. i + W - 1)
  constexpr uint32_t kCap = Out::kCap;
  static_assert(K >= kKcMinK && K <= kKcMaxK, "k out of range for the minimizer path");
  maxlen = MG_UNIFORM(maxlen);
  w0 = MG_UNIFORM(w0);
  if (maxlen < (uint32_t)K) return 0u;  // (uniform) no window in the whole tile
  const uint32_t nw = len >= (uint32_t)K ? len - (uint32_t)K + 1u : 0u;
  const uint32_t nwmax = maxlen - (uint32_t)K + 1u;  // windows of the longest read
  if (w0 >= nwmax) return nwmax;
  const uint32_t nmers = nwmax + (uint32_t)(W - 1);  // ... and its m-mers
  uint32_t f = 0, r = 0, vrun = 0;
  p0 += (uint32_t)E;  // (base u of the walk is base u + E of the read)
  uint32_t word = kc_ext32(fwd, p0 + (w0 & ~15u)), iw = 0;
  if constexpr (MODE == 0) iw = kc_bits32(inv, p0 + (w0 & ~31u));
  uint32_t A[W];  // this block's keys; from the end of the block on, the block's suffix minima
  uint32_t P = kKcNone, Mprev = kKcNone;
  auto take = [&](uint32_t u) -> uint32_t {  // base u comes in; the word of the candidate that ends there (number u - 14)
    if ((u & 15u) == 0) word = kc_ext32(fwd, p0 + u);
    const uint32_t c = (word >> (30u - 2u * (u & 15u))) & 3u;
    // both strands of the m-mer LEFT-aligned in their dwords (bits 31 .. 2): the oldest base falls off the top of f by itself, the
    // newest complement comes into r from above by one funnel shift (what that leaves in r's two low bits is an older base: below
    // every bit a comparison of two different m-mers can turn on — m is odd: no m-mer is its own reverse complement — and below
    // the rank's bits); the word is kc_word's: ((c ^ X) >> 8) << 10 | number, with c << 2 in hand instead of c.  Two operations
    // fewer per base than with right-aligned strands and a mask.
    f = (f << 2) | (c << 2);
#ifdef MG_HOST_CHECK
    r = (r >> 2) | ((c ^ 3u) << 30);
#else
    r = __builtin_amdgcn_alignbit(c ^ 3u, r, 2);
#endif
    const uint32_t ct = f < r ? f : r;
    // ((ct ^ X) & ~1023) | number, written as (ct & ~1023) ^ (a uniform word): one three-input operation
    uint32_t key = (ct & ~kKcPos) ^ (((kKcXor << 2) & ~kKcPos) | (u - (uint32_t)(M - 1)));
    if constexpr (MODE == 0) {
      if ((u & 31u) == 0) iw = kc_bits32(inv, p0 + u);
      const uint32_t bad = (iw >> (31u - (u & 31u))) & 1u;
      vrun = bad ? 0u : vrun + 1u;
      key = vrun >= (uint32_t)M ? key : kKcNone;
    }
    return key;
  };
  for (uint32_t u = w0; u + 1 < w0 + (uint32_t)M; ++u) (void)take(u);
  // The W steps of a block and the W - 2 of its suffix minima are EXPANDED, not looped (fold expressions over the step number):
  // every A[...] is a register, and a block is straight-line code — only the block in which the longest read ends looks, every
  // eighth step, whether it has (what it walks beyond that end is masked like a ragged read's tail).
  // block 0: the first window (m-mers w0 .. w0 + W - 1, all there: w0 < nwmax) opens the first run
  [&]<int... T>(std::integer_sequence<int, T...>) {
    ((A[T] = take(w0 + (uint32_t)(T + M - 1)), P = T == 0 ? A[T] : (A[T] < P ? A[T] : P)), ...);
  }(std::make_integer_sequence<int, W>{});
  Mprev = P;
  if constexpr (MODE != 1) Mprev = w0 < nw ? Mprev : kKcNone;
  uint32_t walked = w0 + 1u;  // the first window not walked
  for (uint32_t b = 1;; ++b) {
    if constexpr (W > 2) {
      [&]<int... T>(std::integer_sequence<int, T...>) {
        ((A[W - 2 - T] = A[W - 2 - T] < A[W - 1 - T] ? A[W - 2 - T] : A[W - 1 - T]), ...);  // A[W-2] .. A[1]
      }(std::make_integer_sequence<int, W - 2>{});
    }
    const uint32_t j0 = w0 + b * (uint32_t)W;  // the block's first m-mer (m-mer j ends at base j + M - 1 and closes window j - (W - 1))
    if (j0 >= nmers || out.any_full(cnt)) break;
    const bool tail = j0 + (uint32_t)W > nmers;
    // (the windows past a lane's own read have no minimizer: one compare against the lane's window count — which is "none" while
    // every lane's read goes on, so that the compare's result goes straight into the select, not through a scalar OR with `tail`)
    // MODE 1 (every read as long as the longest) away from the tile's end: no compare at all — the block's code twice, the one without
    // (two operations of a step's seventeen) for every block but the last of a tile
    auto step = [&]<int T, bool ENDS>() -> bool {
      const uint32_t j = j0 + (uint32_t)T;
      if constexpr (ENDS && T % 8 == 0 && T > 0) { if (j >= nmers) return false; }
      const uint32_t key = take(j + (uint32_t)M - 1u);
      P = T == 0 ? key : (key < P ? key : P);
      uint32_t Mc = P;
      if constexpr (T != W - 1) { const uint32_t s = A[T + 1 < W ? T + 1 : 0]; Mc = s < P ? s : P; }
      const uint32_t i = j - (uint32_t)(W - 1);
      if constexpr (MODE != 1 || ENDS) Mc = i < nw ? Mc : kKcNone;
      out.put(cnt, Mprev, ((i - 1u) & 1023u) << 10);
      const bool ch = Mc != Mprev;
      cnt += ch ? 1u : 0u;  // (add with carry, then the cap: no scalar AND of two compares in between)
      cnt = cnt < kCap ? cnt : kCap;
      Mprev = Mc;
      A[T] = key;
      walked = i + 1u;
      return true;
    };
    bool whole;
    if (tail)
      whole = [&]<int... T>(std::integer_sequence<int, T...>) { return (step.template operator()<T, true>() && ...); }(
          std::make_integer_sequence<int, W>{});
    else
      whole = [&]<int... T>(std::integer_sequence<int, T...>) { return (step.template operator()<T, false>() && ...); }(
          std::make_integer_sequence<int, W>{});
    if (!whole) break;
  }
  walked = walked < nwmax ? walked : nwmax;  // (the tail block may have walked past the end)
  // the run still open (nothing, if its key is kKcNone: a lane whose read has ended closed its last run where it ended)
  out.put(cnt, Mprev, ((walked - 1u) & 1023u) << 10);
  cnt += cnt < kCap ? 1u : 0u;
  // a full list: everything up to its last event is recorded, what follows may not be
  const uint32_t safe = cnt >= kCap ? out.last_window(kCap - 1u) + 1u : walked;
  return out.wave_min(safe < walked ? safe : walked);
}

// the windows of a read that hold no "not a base" bit (KMC's total of k-mers, for the tiles that have such bases)
MG_HD uint32_t kc_clean_windows(const MG_LDS uint32_t* inv, uint32_t p0, uint32_t len, uint32_t maxlen, int k) {
  uint32_t vrun = 0, n = 0, iw = 0;
  for (uint32_t u = 0; u < maxlen; ++u) {
    if ((u & 31u) == 0) iw = kc_bits32(inv, p0 + u);
    const uint32_t bad = (iw >> (31u - (u & 31u))) & 1u;
    vrun = bad ? 0u : vrun + 1u;
    n += (vrun >= (uint32_t)k && u < len) ? 1u : 0u;
  }
  return n;
}

// ---- a run against the table -------------------------------------------------------------------------------------------
// The table's distinct canonical k-mers, laid out so that a run's look-up is ONE access to memory in the usual case:
//   prim   kKcSlots = four 32-byte entries per bucket (bucket of a minimizer = its LOW bits; as many buckets as a power of two >=
//          the k-mers): the bucket's first four entries in (minimizer) order, an unused slot has key kKcNone — one 128-byte line;
//   ovf    the fifth and later entries of the (very few: one bucket in a thousand) buckets that hold more: prim[4 b].pad = how
//          many, prim[4 b + 1].pad = where.  (Two slots per bucket — a 64-byte line — left one bucket in forty with more, and a
//          batch of 64 look-ups then nearly always had a lane that needed a second and third round trip to memory.)
//   an entry's NUMBER: 4 b + s in prim, 4 * buckets + j in ovf — what a sample's saturation bits go by.
// Per sample (mg_kcounts): `live` = a copy of the table's gate bitmap (bit hash >> gshift set <=> some table k-mer is filed
// under a hash with these leading bits) in which a bit is CLEARED once a run has found every k-mer filed under its hash at the
// saturation value — unless the table's `shared` bitmap says that two different hashes of the table have this bit (one in
// 2^kKcGateExtra: their runs keep passing the gate and stop at the saturation bits); ONE bit probe per
// run decides both; `csat` = a counter per entry number: what has been found under it (read with its bucket: at the saturation
// value the entry is skipped); `counts` at the entry's head: the k-mer's occurrences, what stage B reads.
struct KcIndexView {
  MG_GLB uint32_t* live;
  const MG_GLB uint32_t* shared;
  const MG_GLB KcEntry* prim;
  const MG_GLB KcEntry* ovf;
  MG_GLB uint32_t* counts;
  MG_GLB uint32_t* csat;
  uint32_t bmask;         // buckets - 1 (minimizers are minima: their HIGH bits are nearly all zero, the low ones spread)
  uint32_t gshift;        // a hash's gate bit is number hash >> gshift (32 - kc_gate_bits)
  uint32_t cs;            // counters are read as min(counter, cs) (kmc -cs<cs>; 0: exact, nothing ever saturates)
  uint32_t ablate;        // measurements only (knob kc_ablate): 3 = no run is scanned; 4 = signatures only; 5 = no count
  uint32_t epoch;         // 1 .. 255: which of the sample's passes the entry counters' words are of (kc_entry_count)
};

// An entry's counter word: the pass it was written in (top eight bits) above what has been found under the entry in that pass (24
// bits: the lanes in flight on the device times the fifty windows of a run are under ten million).  A word of another pass counts
// as zero — so that nothing has to zero 4 bytes per SLOT of the index before every pass (268 MB at ten million k-mers, 4.3 GB at
// two hundred million); the words are zeroed when the pass number wraps, every 255 passes.
MG_HD uint32_t kc_entry_count(uint32_t word, uint32_t epoch) { return (word >> 24) == epoch ? (word & 0xffffffu) : 0u; }

// an entry out of device memory (class types do not copy out of a qualified address space: as vectors)
#ifdef MG_HOST_CHECK
MG_HD KcEntry kc_load_entry(const KcEntry* ent, uint32_t e) { return ent[e]; }
#else
typedef uint32_t kc_u32x4 __attribute__((ext_vector_type(4)));
MG_HD KcEntry kc_load_entry(const MG_GLB KcEntry* ent, uint32_t e) {
  const MG_GLB kc_u32x4* p = reinterpret_cast<const MG_GLB kc_u32x4*>(ent + e);
  const kc_u32x4 a = p[0], b = p[1];
  return KcEntry{{a.x, a.y, a.z, a.w}, b.x, b.y, b.z, b.w};
}
#endif
// entry number n
MG_HD KcEntry kc_entry(const KcIndexView& ix, uint32_t n) {
  const uint32_t nprim = kKcSlots * (ix.bmask + 1u);
  return n < nprim ? kc_load_entry(ix.prim, n) : kc_load_entry(ix.ovf, n - nprim);
}

MG_HD bool kc_gate(const KcIndexView& ix, uint32_t key) {
  const uint32_t g = key >> ix.gshift;
  return (MG_KC_LOAD(&ix.live[g >> 5]) >> (g & 31u)) & 1u;
}

// no "not a base" bit in [p, p + k)
MG_HD bool kc_window_clean(const MG_LDS uint32_t* inv, uint32_t p, int k) {
  const uint32_t d = p >> 5, sh = p & 31u;
  const uint32_t a = inv[d], b = inv[d + 1], c = inv[d + 2];
  const uint32_t v0 = (uint32_t)(((((uint64_t)a) << 32 | b) << sh) >> 32), v1 = (uint32_t)(((((uint64_t)b) << 32 | c) << sh) >> 32);
  const uint32_t m0 = k >= 32 ? 0xffffffffu : 0xffffffffu << (32 - k);
  const uint32_t m1 = k <= 32 ? 0u : (k >= 64 ? 0xffffffffu : 0xffffffffu << (64 - k));
  return ((v0 & m0) | (v1 & m1)) == 0u;
}


// One table k-mer (words w, left-aligned canonical) against the windows [i1, i2] of the read that starts at stream position p0:
// EVERY window of the run (the entries whose hash is of several candidates; one copy of this code, for the rare).  What every
// window is tested by is its first sixteen bases against the k-mer's two signatures: the bases those tests need — the run's
// windows begin within 50 bases: five dwords — are taken into registers ONCE and shifted along two bits per window; only a
// signature hit reads the window's k-mer from the stream and settles it.  Returns the matches.
MG_HD uint32_t kc_scan_run(const MG_LDS uint32_t* fwd, const MG_LDS uint32_t* inv, int k, bool bad, const uint32_t* ew,
                           uint32_t p0, uint32_t i1, uint32_t i2) {
  constexpr int NS = 5;  // 2 (k - 15) + 32 bits for k <= 64
  const uint32_t p = p0 + i1, d = p >> 4, sh = (p & 15u) << 1;
  uint32_t w[NS];
  {
    uint32_t a[NS + 1];
#pragma unroll
    for (int j = 0; j < NS + 1; ++j) a[j] = fwd[d + j];
#pragma unroll
    for (int j = 0; j < NS; ++j) w[j] = (uint32_t)(((((uint64_t)a[j]) << 32 | a[j + 1]) << sh) >> 32);
  }
  const uint32_t sigmask = kc_keep_mask(k, 0);
  const uint32_t sig_rc = kc_revcomp(KcWin{{ew[0], ew[1], ew[2], ew[3]}}, k).w[0];
  uint64_t hits = 0;
  for (uint32_t i = i1; i <= i2; ++i) {
    const uint32_t x0 = w[0] & sigmask;
    hits |= (x0 == ew[0] || x0 == sig_rc) ? 1ull << (i - i1) : 0ull;
#pragma unroll
    for (int j = 0; j < NS - 1; ++j) w[j] = (w[j] << 2) | (w[j + 1] >> 30);
    w[NS - 1] <<= 2;
  }
  uint32_t found = 0;
  while (hits) {
    const uint32_t i = i1 + (uint32_t)__builtin_ctzll(hits);
    hits &= hits - 1;
    if (bad && !kc_window_clean(inv, p0 + i, k)) continue;
    const KcWin x = kc_ext128(fwd, p0 + i, k), y = kc_revcomp(x, k);
    const KcWin c = kc_less(y, x) ? y : x;
    found += (c.w[0] == ew[0] && c.w[1] == ew[1] && c.w[2] == ew[2] && c.w[3] == ew[3]) ? 1u : 0u;
  }
  return found;
}

// ... and the usual entry, whose hash is of ONE candidate (number off) of its k-mer: the run's candidate starts at base `pos` of
// the read's candidate numbering, so a window equal to the k-mer is window pos - off (as the k-mer stands) or pos - (w - 1 - off)
// (its reverse complement) — if that window is one of the run's (a window that chose another candidate of the same rank is in
// another run, and meets the k-mer's entry under that candidate's hash).  Two windows, each tested by sixteen bases before its
// k-mer is taken from the stream.  Returns the matches: 0, 1, or 2.
MG_HD uint32_t kc_match_windows(const MG_LDS uint32_t* fwd, const MG_LDS uint32_t* inv, int k, bool bad, uint32_t e0, uint32_t e1,
                                uint32_t e2, uint32_t e3, uint32_t off, uint32_t p0, uint32_t pos, uint32_t i1, uint32_t i2) {
  const uint32_t wc = (uint32_t)kc_cands(k);
  const uint32_t ia = pos - off, ib = pos - (wc - 1u - off);  // (below zero: far above i2)
  const uint32_t sigmask = kc_keep_mask(k, 0);
  uint32_t found = 0;
  if (ia >= i1 && ia <= i2 && (kc_ext32(fwd, p0 + ia) & sigmask) == e0 && !(bad && !kc_window_clean(inv, p0 + ia, k))) {
    const KcWin x = kc_ext128(fwd, p0 + ia, k);
    found += (x.w[0] == e0 && x.w[1] == e1 && x.w[2] == e2 && x.w[3] == e3) ? 1u : 0u;
  }
  // (a k-mer that is its own reverse complement, met at the one window both ways: one occurrence)
  if (ib >= i1 && ib <= i2 && !(ib == ia && found) && !(bad && !kc_window_clean(inv, p0 + ib, k))) {
    // the first sixteen bases of the reverse complement are the last sixteen of the window, reversed and complemented
    const uint32_t tail = kc_rc32(kc_ext32(fwd, p0 + ib + (uint32_t)k - 16u));
    if (k < 16 || (tail & sigmask) == e0) {
      const KcWin y = kc_revcomp(kc_ext128(fwd, p0 + ib, k), k);
      found += (y.w[0] == e0 && y.w[1] == e1 && y.w[2] == e2 && y.w[3] == e3) ? 1u : 0u;
    }
  }
  return found;
}

// `found` windows of a run were entry number n (counted at `head`): one add to the k-mer's counter and — counters that saturate —
// one to the ENTRY's own (what the next run that comes to this entry reads with its bucket: at the saturation value it is
// skipped).  Neither add's old value is asked for: nothing waits for them.  `seen`: the entry's word as the run read it; a word
// of an earlier pass is REPLACED (compare-and-swap, result not looked at) instead of added to.  The entry's counter is a HINT: it
// may run behind the k-mer's (a lost exchange, two lanes replacing at once) — then the entry is matched a few times more than
// needed — and never ahead of it (what is added to it is added to the k-mer's counter too), so an entry is skipped only when the
// k-mer's counter has reached the saturation value.
MG_HD void kc_count_entry(const KcIndexView& ix, uint32_t n, uint32_t head, uint32_t found, uint32_t seen) {
  if (!found || ix.ablate == 5u) return;
  (void)MG_KC_ADD(&ix.counts[head], found);
  if (ix.cs) {
    if ((seen >> 24) == ix.epoch) (void)MG_KC_ADD(&ix.csat[n], found);
    else MG_KC_CAS(&ix.csat[n], seen, (ix.epoch << 24) | found);
  }
}

// A run past the gate against its bucket, one lane on its own (the host check, and the statement of what the kernel's batched
// phases — mg_kcount.hip: kc_drain — compute): every entry filed under the run's hash whose counter is not saturated is
// matched; when all of them are saturated the hash's bit is cleared in the sample's gate.
MG_HD uint32_t kc_match_run(const KcIndexView& ix, const MG_LDS uint32_t* fwd, const MG_LDS uint32_t* inv, int k, bool bad, uint32_t key,
                            uint32_t p0, uint32_t pos, uint32_t i1, uint32_t i2) {
  const uint32_t b = key & ix.bmask, nprim = kKcSlots * (ix.bmask + 1u);
  uint32_t found = 0, novf = 0, ovf_at = 0;
  bool any = false, open = false;  // an entry filed under this hash; one of them whose counter is not saturated
  for (uint32_t t = 0; t < kKcSlots + novf; ++t) {
    const uint32_t n = t < kKcSlots ? kKcSlots * b + t : nprim + ovf_at + (t - kKcSlots);
    const KcEntry E = kc_entry(ix, n);
    if (t == 0) novf = E.pad;
    if (t == 1) ovf_at = E.pad;
    if (E.key != key) continue;
    any = true;
    const uint32_t seen = MG_KC_LOAD(&ix.csat[n]);
    if (ix.cs && kc_entry_count(seen, ix.epoch) >= ix.cs) continue;
    open = true;
    const uint32_t f = E.off == kKcSeveral ? kc_scan_run(fwd, inv, k, bad, E.w, p0, i1, i2)
                                           : kc_match_windows(fwd, inv, k, bad, E.w[0], E.w[1], E.w[2], E.w[3], E.off, p0, pos, i1, i2);
    kc_count_entry(ix, n, E.head, f, seen);
    found += f;
  }
  const uint32_t g = key >> ix.gshift;
  // (a run that another hash's bit let through finds no entry of its own: that bit is not its to clear)
  if (ix.cs && any && !open && !((ix.shared[g >> 5] >> (g & 31u)) & 1u) && ((MG_KC_LOAD(&ix.live[g >> 5]) >> (g & 31u)) & 1u))
    MG_KC_AND(&ix.live[g >> 5], ~(1u << (g & 31u)));
  return found;
}

}  // namespace mg
