// mg_profile.hip — Stage C: per-read taxon assignment + abundance histogram, ONE pass over the records.
//
// Replaces the loop of map_and_process (scripts/map_and_profile.py:193-264).
// The reference loop carries one bit across read boundaries: when a read is
// judged Ambiguous the `continue` at :232 skips the append at :257-259, so the
// first line of the NEXT read is dropped.  Whether read g is Ambiguous depends
// on whether its own first line was dropped, so read g is a map
//     A_g : {first line kept, first line dropped} -> {next kept, next dropped}
// and the true state of every read is the prefix composition of these maps
// (associative, not commutative) started from "dropped" (the phantom first
// boundary, :155-156).
//
// k_profile_pass is a single-pass chained scan (decoupled look-back) over tiles of 2048 records:
//   1. the tile's records are loaded once (coalesced 16 B loads), reduced to 8-byte descriptors
//      {taxon, new-read / rejected / pair bits, len(SEQ)} (the fp64 division of filter_line :96-99 is done once
//      per record) and parked in LDS;
//   2. every read leader classifies its read under BOTH hypotheses (process_read :152-176) from LDS; a thread
//      folds its 8 consecutive records into a function of the state entering it: outgoing state, multimapped
//      reads and list entries it would emit; an ordered workgroup scan composes these functions;
//   3. the tile publishes its aggregate AS A FUNCTION of the state entering it (two 64-bit words);
//   4. one wavefront looks back over the preceding tiles' descriptors, 256 per step, composing them with an
//      ordered butterfly until it meets an inclusive prefix, and publishes its own inclusive prefix — tiles are
//      handed out by an atomic ticket, so every predecessor is resident or finished and the look-back cannot
//      dead-lock;
//   5. with the incoming state and the exclusive read / multimapped offsets known, the leaders classify once
//      more (from LDS, true hypothesis only) and commit: unique reads into an LDS-privatised histogram (count,
//      bases, first-seen) that the workgroup keeps across its tiles and flushes once with global atomics,
//      multimapped reads straight into the CSR (SAM order).
// HBM traffic: 16 B per record, once (the previous version read the records three times, wrote and re-read
// 5 B of scratch per record and ran two single-workgroup scans: 17.2 ms at 125 M records).
// COMMIT = false is the same chain without step 5: the shard's composed state map and read count, which a
// multi-GPU job needs from every shard before any of them can commit.
#include <memory>

#include <algorithm>
#include <vector>

#include <atomic>
#include <thread>

#include "mg_internal.h"

namespace mg {

#ifndef MG_K3_LOAD_BATCH
#define MG_K3_LOAD_BATCH 3
#endif
constexpr int kPB = 256;                 // threads per workgroup
constexpr int kItems = 8;                // consecutive records per thread
constexpr int kTile = kPB * kItems;      // records per tile
constexpr int kHalo = 64;                // descriptors staged past the tile end (a read's lines + its closing line)
constexpr int kStaged = kTile + kHalo;

struct Verdict {
  uint32_t kind;  // 0 Ambiguous, 1 unique, 2 multimapped
  uint32_t tax;
  uint64_t hitlen;
  uint32_t nmm;
};

// 8-byte digest of one record: all that process_read needs.
constexpr uint32_t D_NEW = 1u, D_REJ = 2u, D_P1 = 4u, D_P2 = 8u;
constexpr int D_LEN_SHIFT = 12;

__device__ __forceinline__ uint2 make_desc(const mg_aln_rec& r, uint32_t tax, double pct_id) {
  const uint32_t fl = r.flag_len & MG_REC_FLAG_MASK;
  const bool a = (fl & 1u) && (fl & 64u);   // parse_flag :106
  const bool b = (fl & 1u) && (fl & 128u);  // :107
  // filter_line (:86-100) or chimeric (:108,135)
  const bool rej = ((double)r.matched / (double)r.total < pct_id) || (fl & 2048u);
  uint2 d;
  d.x = tax;
  d.y = (r.ref_new >> 31) | (rej ? D_REJ : 0u) | (a ? D_P1 : 0u) | (b ? D_P2 : 0u) |
        ((r.flag_len >> MG_REC_LEN_SHIFT) << D_LEN_SHIFT);
  return d;
}

// LDS slot of the tile-local record index i: one pad slot per kItems, so that the threads' strided
// accesses (thread t owns records kItems*t ..) fall into distinct banks.
__device__ __forceinline__ uint32_t slot(uint32_t i) { return i + (i >> 3); }
constexpr int kSlots = kStaged + (kStaged >> 3) + 1;

// Descriptors of records [t0, t0 + nst) live in LDS; anything past that (a read longer than the halo) is
// rebuilt from HBM.
// The LDS pointer keeps its address space in the type: as a generic pointer inside a struct the compiler fell back
// to FLAT loads in eval_group, and with a FLAT access possibly in flight every wait in the commit loop became
// vmcnt(0) — each read then waited for its own multimapped-list stores to reach memory.
typedef const __attribute__((address_space(3))) unsigned long long* LdsDescPtr;  // a uint2 as one 64-bit word
struct DescView {
  LdsDescPtr lds;
  uint64_t t0;
  uint32_t nst;
  const mg_aln_rec* __restrict__ recs;
  const uint32_t* __restrict__ ref2tax;
  double pct_id;
  __device__ __forceinline__ uint2 operator()(uint64_t i) const {
    if (i - t0 < nst) {
      const unsigned long long w = lds[slot((uint32_t)(i - t0))];
      return make_uint2((uint32_t)w, (uint32_t)(w >> 32));
    }
    const mg_aln_rec r = recs[i];
    return make_desc(r, ref2tax[r.ref_new & MG_REC_REF_MASK], pct_id);
  }
};

// process_read (:152-176) over the lines [s, e), any length, any flags; next_paired = the line that closes the
// read carries pair flags (the `pair1, pair2` passed at :225-226).  mm_out: where to write the taxon list of a
// multimapped read (nullptr = only count it).  The kernel calls this for the reads its straight-line path does
// not cover: both mates mapped (set intersection, :115-125) and reads longer than the staged window.
__device__ __forceinline__ Verdict eval_group(const DescView& at, uint64_t s, uint64_t e, bool next_paired, uint32_t* mm_out) {
  Verdict v{0, 0, 0, 0};
  long p1 = 0, p2 = 0;
  uint64_t nk = 0;
  for (uint64_t i = s; i < e; ++i) {  // appends (:257-258) then clean_read_hits (:130-147)
    const uint2 d = at(i);
    const bool a = d.y & D_P1, b = d.y & D_P2;
    p1 += (a || !(a || b)) ? 1 : 0;
    p2 += b ? 1 : 0;
    if (d.y & D_REJ) {
      if (a) p1 -= 1; else if (b) p2 -= 1;
    } else {
      if (nk == 0) v.tax = d.x;
      ++nk;
    }
    v.hitlen += d.y >> D_LEN_SHIFT;
  }
  if (nk == 0) return v;  // :155-156
  auto kept = [&](uint64_t i) { return !(at(i).y & D_REJ); };
  auto tax = [&](uint64_t i) { return at(i).x; };
  auto emit = [&](uint32_t t) {
    if (mm_out) mm_out[v.nmm] = t;
    ++v.nmm;
  };
  if (next_paired) {                                   // :157
    if (p1 + p2 == 1) { v.kind = 1; return v; }         // :158-160
    if (p1 == 0 || p2 == 0) return v;                   // :116-117 -> :164-165
    uint64_t split = p1 < 0 ? 0 : (uint64_t)p1;
    if (split > nk) split = nk;
    // in_second(t): some kept line with ordinal >= split has taxon t; in_first: ordinal < split
    auto in_second = [&](uint32_t t) {
      uint64_t o = 0;
      for (uint64_t j = s; j < e; ++j) {
        if (!kept(j)) continue;
        if (o >= split && tax(j) == t) return true;
        ++o;
      }
      return false;
    };
    auto in_first = [&](uint32_t t) {
      uint64_t o = 0;
      for (uint64_t j = s; j < e && o < split; ++j) {
        if (!kept(j)) continue;
        if (tax(j) == t) return true;
        ++o;
      }
      return false;
    };
    uint64_t ndistinct = 0, o = 0;
    for (uint64_t i = s; i < e && o < split; ++i) {  // |set(pair1refs) & set(pair2refs)| (:121-122)
      if (!kept(i)) continue;
      const uint32_t t = tax(i);
      if (in_second(t)) {
        bool dup = false;
        for (uint64_t j = s; j < i && !dup; ++j) dup = kept(j) && tax(j) == t;
        if (!dup) ++ndistinct;
      }
      ++o;
    }
    if (ndistinct == 0) return v;                  // :164-165
    if (ndistinct == 1) { v.kind = 1; return v; }  // :166-167 (taxon of the FIRST kept line)
    v.kind = 2;                                    // :168-169
    for (uint64_t i = s; i < e; ++i) {
      if (!kept(i)) continue;
      const uint32_t t = tax(i);
      if (in_first(t) && in_second(t)) emit(t);
    }
    return v;
  }
  if (p1 > 1) {  // single end, multimapped (:172-173)
    v.kind = 2;
    for (uint64_t i = s; i < e; ++i)
      if (kept(i)) emit(tax(i));
    return v;
  }
  v.kind = 1;  // :174-176
  return v;
}

// Counters of clean_read_hits (:130-147) over staged lines [s, e) (tile-local indices).
struct Walk {
  int p1, p2;
  uint32_t nk, tax;  // kept lines, taxon of the first kept line
  uint64_t hsum;     // sum of len(SEQ) over all lines
};
__device__ __forceinline__ void walk_add(Walk& w, const uint2 d) {
  const bool a = d.y & D_P1, b = d.y & D_P2, rej = d.y & D_REJ;
  w.p1 += ((a || !b) ? 1 : 0) - ((rej && a) ? 1 : 0);
  w.p2 += (b ? 1 : 0) - ((rej && !a && b) ? 1 : 0);
  if (!rej) {
    if (w.nk == 0) w.tax = d.x;
    ++w.nk;
  }
  w.hsum += d.y >> D_LEN_SHIFT;
}
// process_read's decision from the counters: kind | nmm << 2; kind 3 = needs the set intersection (eval_group).
__device__ __forceinline__ uint32_t classify(const Walk& w, bool next_paired) {
  if (w.nk == 0) return 0u;                           // :155-156
  if (next_paired) {                                  // :157
    if (w.p1 + w.p2 == 1) return 1u;                  // :158-160
    if (w.p1 == 0 || w.p2 == 0) return 0u;            // :116-117 -> :164-165
    return 3u;
  }
  if (w.p1 > 1) return 2u | (w.nk << 2);              // :172-173
  return 1u;                                          // :174-176
}

// State maps: bit0 = f(0), bit1 = f(1); x = 1 means "first line dropped".
constexpr uint32_t kIdentity = 2u;
__device__ __forceinline__ uint32_t map_then(uint32_t first, uint32_t second) {
  return ((second >> (first & 1u)) & 1u) | (((second >> ((first >> 1) & 1u)) & 1u) << 1);
}

// A run of records as a function of the state x entering it: outgoing state and what it emits.
// pk[x] packs multimapped list entries (bits 0..35), multimapped reads (36..49) and reads (50..63) of at most
// one tile.
struct RunFn {
  uint32_t out;
  uint64_t pk[2];
};
__device__ __forceinline__ RunFn run_then(const RunFn& f, const RunFn& s) {
  RunFn o;
  o.out = map_then(f.out, s.out);
  o.pk[0] = f.pk[0] + ((f.out & 1u) ? s.pk[1] : s.pk[0]);  // selects, not indexing: the pair stays in registers
  o.pk[1] = f.pk[1] + ((f.out & 2u) ? s.pk[1] : s.pk[0]);
  return o;
}
constexpr int kPkMmShift = 36, kPkReadShift = 50;
constexpr uint64_t kPkOneMm = 1ull << kPkMmShift, kPkOneRead = 1ull << kPkReadShift;
__device__ __forceinline__ uint64_t pk_ent(uint64_t pk) { return pk & (kPkOneMm - 1); }
__device__ __forceinline__ uint64_t pk_mm(uint64_t pk) { return (pk >> kPkMmShift) & 0x3fffull; }
__device__ __forceinline__ uint64_t pk_reads(uint64_t pk) { return pk >> kPkReadShift; }
static_assert(kTile < (1 << 13), "tile counters are 13-bit fields in the published aggregate");

// Ordered inclusive scan of RunFn across the workgroup; returns the EXCLUSIVE prefix of this thread and the
// workgroup aggregate.  lds: kPB / 64 entries.
__device__ __forceinline__ RunFn block_scan_runs(const RunFn& mine, RunFn* lds, RunFn* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  RunFn inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    RunFn prev;
    prev.out = __shfl_up(inc.out, o, 64);
    prev.pk[0] = __shfl_up(inc.pk[0], o, 64);
    prev.pk[1] = __shfl_up(inc.pk[1], o, 64);
    if (lane >= o) inc = run_then(prev, inc);
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  RunFn before{kIdentity, {0, 0}}, all{kIdentity, {0, 0}};
#pragma unroll
  for (int w = 0; w < kPB / 64; ++w) {
    const RunFn t = lds[w];
    if (w < wave) before = run_then(before, t);
    all = run_then(all, t);
  }
  RunFn excl;
  excl.out = __shfl_up(inc.out, 1, 64);
  excl.pk[0] = __shfl_up(inc.pk[0], 1, 64);
  excl.pk[1] = __shfl_up(inc.pk[1], 1, 64);
  if (lane == 0) excl = RunFn{kIdentity, {0, 0}};
  __syncthreads();
  *total = all;
  return run_then(before, excl);
}

// ---- tile descriptors of the chained scan ----
// Three 64-bit words per tile, each carrying the status in its top two bits, so that a reader needs no fence:
// it re-reads until the three statuses agree.
//   AGG     w[x] (x = 0, 1) = status | out(x):1 | reads:13 | mm_reads(x):13 | mm_entries(x):35 ;  w[2] = status
//   PREFIX  w[0] = status | map:2 | reads (inclusive) ; w[1] = status | mm_reads ; w[2] = status | mm_entries
constexpr uint64_t ST_AGG = 1ull << 62, ST_PREFIX = 2ull << 62, ST_MASK = 3ull << 62;
constexpr int kDescWords = 4;  // 32-byte stride
constexpr int kTicketLanes = 16;    // interleaved ticket counters ...
constexpr int kTicketStride = 16;   // ... 128 bytes apart
constexpr int kWin = 4;        // descriptors per lane and look-back step (window = 256 tiles)

struct TileFn {  // aggregate of a run of tiles as a function of the state entering it
  uint32_t out;  // 2-bit map
  uint64_t g;    // reads
  uint64_t r[2], e[2];
};
__device__ __forceinline__ TileFn fn_identity() { return TileFn{kIdentity, 0, {0, 0}, {0, 0}}; }
__device__ __forceinline__ TileFn fn_then(const TileFn& f, const TileFn& s) {
  TileFn o;
  o.out = map_then(f.out, s.out);
  o.g = f.g + s.g;
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const uint32_t mid = (f.out >> x) & 1u;
    o.r[x] = f.r[x] + (mid ? s.r[1] : s.r[0]);
    o.e[x] = f.e[x] + (mid ? s.e[1] : s.e[0]);
  }
  return o;
}
__device__ __forceinline__ uint64_t agg_word(uint32_t out_x, uint64_t g, uint64_t r, uint64_t e) {
  return ST_AGG | ((uint64_t)out_x << 61) | (g << 48) | (r << 35) | e;
}
// Descriptor words -> function.  An inclusive prefix is a function of the state entering the SHARD whose
// counts are those of the true path (the look-back only ever evaluates it there); everything farther back is
// already folded into it.
__device__ __forceinline__ TileFn fn_from_words(uint64_t w0, uint64_t w1, uint64_t w2) {
  TileFn f;
  if ((w0 & ST_MASK) == ST_PREFIX) {
    f.out = (uint32_t)(w0 >> 60) & 3u;
    f.g = w0 & ((1ull << 60) - 1);
    f.r[0] = f.r[1] = w1 & ~ST_MASK;
    f.e[0] = f.e[1] = w2 & ~ST_MASK;
  } else {
    f.out = (uint32_t)((w0 >> 61) & 1u) | ((uint32_t)((w1 >> 61) & 1u) << 1);
    f.g = (w0 >> 48) & 0x1fffull;
    f.r[0] = (w0 >> 35) & 0x1fffull; f.e[0] = w0 & ((1ull << 35) - 1);
    f.r[1] = (w1 >> 35) & 0x1fffull; f.e[1] = w1 & ((1ull << 35) - 1);
  }
  return f;
}
__device__ __forceinline__ TileFn fn_shfl_xor(const TileFn& f, int o) {
  TileFn t;
  t.out = __shfl_xor(f.out, o, 64);
  t.g = __shfl_xor(f.g, o, 64);
  t.r[0] = __shfl_xor(f.r[0], o, 64); t.r[1] = __shfl_xor(f.r[1], o, 64);
  t.e[0] = __shfl_xor(f.e[0], o, 64); t.e[1] = __shfl_xor(f.e[1], o, 64);
  return t;
}

__device__ __forceinline__ uint64_t ld_desc(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_desc(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Incoming {  // what a tile learns from its predecessors
  uint32_t map;    // composed map from the shard start to the tile start
  uint64_t g, r, e;
};

constexpr int kLbWaves = 1;     // wavefronts that look back together: kLbWaves x 64 x kWin tiles per round

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kWinTiles = 64 * kWin;             // tiles per look-back window
constexpr int kWinChunks = kWinTiles * 2 / 64;   // 16-byte chunks per lane and window (a descriptor = 2 chunks)
static_assert(kWinChunks == 8, "the batched load below is written for 8 chunks per lane");

// 8 x 16 bytes per lane, L2-bypassing (sc0 sc1: another XCD wrote them), all in flight before one wait.  Chunk j
// of lane l sits at p + (j * 64 + l) * 16 bytes: every instruction reads one contiguous KB.
__device__ __forceinline__ void load_window(const u32x4* p, u32x4 (&v)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %8, off offset:1024 sc0 sc1\n\t"
      "global_load_dwordx4 %2, %8, off offset:2048 sc0 sc1\n\t"
      "global_load_dwordx4 %3, %8, off offset:3072 sc0 sc1\n\t"
      "global_load_dwordx4 %4, %9, off sc0 sc1\n\t"
      "global_load_dwordx4 %5, %9, off offset:1024 sc0 sc1\n\t"
      "global_load_dwordx4 %6, %9, off offset:2048 sc0 sc1\n\t"
      "global_load_dwordx4 %7, %9, off offset:3072 sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p), "v"(p + 256)
      : "memory");
}

// Executed by one whole wavefront: folds the window of 64 * kWin tiles whose nearest member is tile `base`
// (lane l folds the tiles base - kWin*l - c, c = 0 .. kWin-1), stopping at the nearest inclusive prefix.
// Returns once every tile nearer than that prefix (all of them, if there is none) is published.
// The window's descriptors (8 KB, contiguous) are read with coalesced 16-byte loads and handed to their lanes
// through LDS: per-word atomic loads made every look-back ~770 separate 8-byte requests to the memory side, and
// the fabric's request rate, not its latency, bounded the pass at ~80 ns per tile whatever the window or tile size.
// A 16-byte load may tear between its two words; every word carries the status, torn reads fail the check below.
__device__ __forceinline__ TileFn look_window(const uint64_t* __restrict__ desc, int64_t base, bool* found,
                                              u32x4* __restrict__ lbuf) {
  const int lane = threadIdx.x & 63;
  TileFn f;
  int p_lane;
  const int64_t lo = base - (kWinTiles - 1);  // tile of chunk 0 (may be negative: before the first tile)
  for (;;) {
    {
      // chunks of tiles outside [0, base] are not read (the window is clamped into the descriptor array and the
      // lanes below substitute the empty prefix for tiles before 0)
      const int64_t lo_c = lo < 0 ? 0 : lo;
      u32x4 v[8];
      load_window(reinterpret_cast<const u32x4*>(desc + (uint64_t)lo_c * kDescWords) + lane, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) lbuf[j * 64 + lane] = v[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f = fn_identity();
    bool ok = true, pfx = false;
    const int64_t shift = lo < 0 ? -lo : 0;  // the staged window starts at tile max(lo, 0)
#pragma unroll
    for (int c = 0; c < kWin; ++c) {
      const int64_t k = base - (lane * kWin + c);
      uint64_t w0 = ST_PREFIX | ((uint64_t)kIdentity << 60), w1 = ST_PREFIX, w2 = ST_PREFIX;  // before tile 0
      if (k >= 0) {
        const int64_t slot = (k - lo) - shift;  // index of tile k in the staged window
        const u32x4 a = lbuf[slot * 2], b2 = lbuf[slot * 2 + 1];
        w0 = (uint64_t)a.x | ((uint64_t)a.y << 32);
        w1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
        w2 = (uint64_t)b2.x | ((uint64_t)b2.y << 32);
      }
      const uint64_t s0 = w0 & ST_MASK;
      const bool pub = s0 != 0 && s0 == (w1 & ST_MASK) && s0 == (w2 & ST_MASK);
      if (ok && !pfx) {
        if (!pub) ok = false;
        else {
          f = fn_then(fn_from_words(w0, w1, w2), f);  // farther tile: prepend
          pfx = s0 == ST_PREFIX;
        }
      }
    }
    const unsigned long long pm = __ballot(ok && pfx);
    p_lane = pm ? __ffsll((long long)pm) - 1 : 64;
    if (__ballot(lane <= p_lane && !ok) == 0ull) break;
    __builtin_amdgcn_s_sleep(2);
  }
  if (lane > p_lane) f = fn_identity();  // farther than the prefix: already folded into it
  // ordered butterfly over the lanes (lane 63 is the farthest)
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const TileFn t = fn_shfl_xor(f, o);
    f = (lane & o) ? fn_then(f, t) : fn_then(t, f);
  }
  *found = p_lane < 64;
  return f;
}

// Per-workgroup privatised histogram in LDS, flushed with global atomics every kFlushTiles tiles and at the end:
//   use_lds_hist == 1  direct:  ntax <= 4096 bins (dynamic LDS: 12 B per bin; above 2048 taxa two workgroups per CU, like the hashed form)
//   use_lds_hist == 2  hashed:  ntax < 2^18; kHashSlots open-addressed bins keyed by taxon id (a sample hits far
//                      fewer taxa than the table lists).  A taxon that finds no bin within kHashProbe probes goes to
//                      the WORKGROUP'S PRIVATE BINS IN GLOBAL MEMORY (priv_pack / priv_first: [workgroup][ntax]),
//                      updated with WORKGROUP-scope atomics — executed in the L2 of the XCD the workgroup runs on,
//                      not at the memory side like device-scope ones (~14 G/s on this multi-XCD part).  Nothing else
//                      touches a workgroup's copy while the kernel runs; k_bins_reduce sums the copies afterwards (and
//                      leaves them zeroed), and reads none of them when no workgroup overflowed.
//   use_lds_hist == 0  global atomics only (shards of 2^32 reads or more: the bins hold 32-bit read indices).
// A bin is ONE packed 64-bit word (count << 40 | bases) plus the 32-bit shard-relative index of the first read
// seen: one LDS atomic per unique read (the minimum is only attempted when a plain read says it would change
// something — after a workgroup's first tile it almost never does).  Three atomics per read on a few dozen hot bins
// had been 2/3 of the commit pass: LDS atomics to one address serialise, and every workgroup of the CU shares the
// pipeline.  Reads of 2^14 bases or more bypass the bins (kQueueLenLimit), so that kFlushTiles tiles cannot overflow a field.
// kHashProbe slots (double hashing over the packed words) are looked at before the private bins take a record.  Measured in round 4
// on bench.py's configs[2] records (12.5M, 10 001 taxa, ~1000 of them hit uniquely: the sample's 500 genomes and — through reads
// whose first line was dropped — their sibling accessions): 0.46-0.47 ms with the bins hashed against 0.30 ms with direct bins on
// the same records folded onto 2001 taxa, whatever the look-up's form (buckets of four keys, single keys, packed words).  The
// phase clocks (profiles/r04/k3_phases_queue_dump.txt) found the 0.16 ms elsewhere: (1) 8 probes failed for a handful of taxa per
// workgroup, so every pass ended with the reduction over ~90 MB of private copies; (2) the flush every 8 tiles fell in the middle
// of a workgroup's life and stalled the look-back chain behind it; (3) the last flush — hash-ordered slots touch 64 cache lines
// per instruction, 2.3 M atomics from 768 workgroups at the same moment.  Now: 0.36 ms; what is left over direct bins is the
// look-up (one dependent LDS read per read and the wavefront's slowest probe sequence).
// The commit walk queues a uniquely mapped read as ONE 32-bit word (taxon << 14 | bases) in the LDS the look-back used, kItems
// words per thread; reads of 2^14 bases or more go to the global bins at once.
constexpr int kQueueLenBits = 14;
constexpr uint32_t kQueueLenLimit = 1u << kQueueLenBits;
static_assert(kItems <= 32 && kPB * kItems * sizeof(uint32_t) <= sizeof(u32x4) * kWinTiles * 2, "the queue lives in one look-back window");
constexpr uint32_t kHashSlots = 2048, kHashProbe = 8, kFlushTiles = 256;
// ... 8 probes once the table is crowded (more than kHashCrowded keys claimed since the last flush: a sample with more taxa than
// slots must not walk 64 of them per read), 64 while it is not: at configs[2]'s ~1000 taxa in 2048 slots 8 probes failed for a
// handful of taxa per workgroup, every workgroup then had used its private copy, and the reduction over ~90 MB of copies that
// followed was 0.05 of the pass's 0.46 ms.
constexpr uint32_t kHashProbeFree = 64, kHashCrowded = kHashSlots * 3 / 4;
// The hashed form's bin (round 4): taxon + 1 (18 bits) | count (15) | bases (31) in ONE 64-bit word, so that a slot is 12 bytes
// (word + first read seen) and 2048 of them fit beside three workgroups per CU; 0 = empty.  The fields hold kHashFlushTiles tiles
// (15 x 2048 reads of < 2^14 bases each; longer reads bypass the bins), then the bins are flushed — and their keys with them.
constexpr int kHKeyShift = 46, kHCountShift = 31;
constexpr uint32_t kHashFlushTiles = 15, kHashMaxTax = (1u << 18) - 2;
static_assert((uint64_t)kHashFlushTiles * kTile < (1u << 15) && (uint64_t)kHashFlushTiles * kTile * kQueueLenLimit < (1ull << 31), "hashed bin fields");
static_assert(kHashMaxTax < (1u << (32 - kQueueLenBits)), "taxon field of a queue word");
constexpr int kBinCountShift = 40;

// The XCD this wavefront runs on (0..7), from the hardware register.  Atomics without device scope are resolved in that XCD's L2;
// every access to an XCD's copy comes from workgroups of that XCD, the kernel's end writes it back, a later kernel sums the copies.
__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}
constexpr uint32_t kXcds = 8;
// ASSUMPTION, by architecture: on gfx942 / gfx950 a global atomic below device scope is executed in the L2 of the XCD the wavefront
// runs on, so workgroups of ONE XCD that update the same word of that XCD's copy see each other's updates although the scope says
// "workgroup" (under the HIP memory model alone that sharing would be a race); the kernel's end writes the L2 back and the next kernel
// sums the eight copies.  Any other part gets the agent-scope atomic (correct anywhere, resolved at the memory side).
#if defined(__gfx942__) || defined(__gfx950__)
#define MG_L2_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#else
#define MG_L2_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
__device__ __forceinline__ void l2_add64(unsigned long long* p, unsigned long long v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, MG_L2_SCOPE);
}
__device__ __forceinline__ void l2_min64(unsigned long long* p, unsigned long long v) {
  (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, MG_L2_SCOPE);
}

struct PassArgs {
  const mg_aln_rec* recs;
  uint64_t nrecs, ntotal;
  const uint32_t* ref2tax;
  double pct_id;
  uint64_t* desc;                 // [ntiles][kDescWords], zeroed
  unsigned long long* ticket;     // [ticket_lanes][kTicketStride], zeroed
  uint32_t ticket_lanes;
  uint64_t ntiles;
  uint32_t incoming, first_shard;
  uint64_t group_base;
  uint32_t ntax, use_lds_hist;
  uint32_t flush_tiles;           // the LDS bins go to the accumulators every so many tiles (kFlushTiles / kHashFlushTiles)
  unsigned long long *g_count, *g_bases, *g_first, *g_scalars;
  unsigned long long* priv_pack;  // hashed mode: [gridDim.x][ntax] packed bins (count << 40 | bases), all zero between passes ...
  uint32_t* priv_first;           // ... first-seen words (0xffffffff between passes) ...
  unsigned long long* xcd_bins;   // hashed mode: [8 XCDs][count | bases | first seen][ntax], (0, 0, all-ones) between passes
  unsigned long long* dump_pack;  // hashed mode: [gridDim.x][kHashSlots] every workgroup's LDS bins as they are at its end ...
  uint32_t* dump_first;           // ... summed slot by slot by k_bins_reduce
  uint32_t* priv_used;            // ... and the flags: [0] some workgroup used its private bins in this pass, [1 + w] workgroup w did
  uint64_t* mm_offsets; uint32_t* mm_tax; uint64_t* mm_hitlen; uint64_t* mm_read;
  uint64_t* out_tot;              // [0] composed map, [1] reads, [2] multimapped entries, [3] multimapped reads
};

// Phase timing (make K3_PHASES=1; tools/k3_phases.py): thread 0's shader-clock time between the phase boundaries
// of every tile, summed per kernel variant.  Compiled out of the normal build.
#ifdef MG_K3_PHASES
__device__ unsigned long long g_ph[24];
extern "C" int mg_debug_k3_phases(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ph), sizeof(g_ph)) != hipSuccess) return MG_ERR_HIP;
  unsigned long long z[24] = {0};
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_ph), z, sizeof(z)) != hipSuccess) return MG_ERR_HIP;
  return MG_OK;
}
#define PH_DECL __shared__ unsigned long long s_ph[12]; unsigned long long tprev = clock64(); if (threadIdx.x < 12) s_ph[threadIdx.x] = 0
#define PH(i) do { if (tid == 0) { const unsigned long long now_ = clock64(); s_ph[i] += now_ - tprev; tprev = now_; } } while (0)
#define PH_FLUSH(c) do { if (tid == 0) for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_ph[i_ + ((c) ? 12 : 0)], s_ph[i_]); } while (0)
#else
#define PH_DECL
#define PH(i)
#define PH_FLUSH(c)
#endif
template <bool COMMIT>
__device__ __forceinline__ void profile_pass_body(const PassArgs& A) {
  PH_DECL;
  extern __shared__ __attribute__((aligned(16))) unsigned long long hist[];
  __shared__ uint2 s_desc[kSlots];
  __shared__ RunFn s_run[kPB / 64];
  __shared__ uint64_t s_bcast[4];
  __shared__ TileFn s_lb[kLbWaves];
  __shared__ u32x4 s_lbuf[kLbWaves][kWinTiles * 2];  // look-back windows staged for their wavefronts (8 KB each)
  __shared__ uint32_t s_lbf[kLbWaves + 1];
  __shared__ unsigned long long s_ticket;
  __shared__ unsigned long long s_ambig, s_groups;
  __shared__ uint32_t s_nkeys;  // hashed bins: slots claimed since the last flush
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t nbins = A.use_lds_hist == 2 ? kHashSlots : A.ntax;
  unsigned long long* h_pack = hist;                                     // direct: count << 40 | bases; hashed: taxon + 1 | count | bases
  uint32_t* h_first = reinterpret_cast<uint32_t*>(hist + nbins);         // first read seen, relative to the shard
  auto flush_bins = [&](bool reset) {
    for (uint32_t t = tid; t < nbins; t += kPB) {
      const unsigned long long pk = h_pack[t];
      if (pk) {
        const bool hashed = A.use_lds_hist == 2;
        const uint32_t tax = hashed ? (uint32_t)(pk >> kHKeyShift) - 1u : t;
        const unsigned long long cnt = hashed ? (pk >> kHCountShift) & 0x7fffull : pk >> kBinCountShift;
        const unsigned long long bas = hashed ? pk & ((1ull << kHCountShift) - 1) : pk & ((1ull << kBinCountShift) - 1);
        if (cnt && hashed) {  // (a hashed slot may be claimed and not yet counted in)
          // Slots hold taxa in hash order: a wavefront's flush touches 64 different cache lines per instruction where the direct
          // bins' touches four, and ~3000 device-scope atomics from each of 768 workgroups, all at the end of the pass, were
          // 0.08 of its 0.43 ms (phase clocks, profiles/r04).  They go to THIS XCD's copy of the accumulators, resolved in its L2.
          unsigned long long* x = A.xcd_bins + (size_t)xcc_id() * 3 * A.ntax;
          l2_add64(x + tax, cnt);
          l2_add64(x + A.ntax + tax, bas);
          l2_min64(x + 2 * (size_t)A.ntax + tax, (unsigned long long)(A.group_base + h_first[t]));
        } else if (cnt) {
          atomicAdd(&A.g_count[tax], cnt);
          atomicAdd(&A.g_bases[tax], bas);
          atomicMin(&A.g_first[tax], (unsigned long long)(A.group_base + h_first[t]));
        }
        if (reset) { h_pack[t] = 0; h_first[t] = 0xffffffffu; }
      }
    }
    if (reset && tid == 0) s_nkeys = 0;
  };
  uint32_t tiles_binned = 0;
  if (COMMIT && A.use_lds_hist) {
    for (uint32_t t = tid; t < nbins; t += kPB) {
      h_pack[t] = 0; h_first[t] = 0xffffffffu;
    }
  }
  if (tid == 0) { s_ambig = 0; s_groups = 0; s_nkeys = 0; }
  __syncthreads();
  PH(10);

  for (;;) {
    // Tiles are handed out in order, by kTicketLanes interleaved counters (tile = ticket * lanes + lane, a workgroup
    // always draws from lane blockIdx % lanes): returning atomics on ONE address retire at ~12 M/s on this part,
    // which capped the whole pass at ~80 ns per tile whatever its size.  The smallest unfinished tile is always
    // either being processed or next in its lane's queue, so the look-back still cannot dead-lock.
    PH(7);
    if (tid == 0) s_ticket = atomicAdd(A.ticket + (size_t)(blockIdx.x % A.ticket_lanes) * kTicketStride, 1ull);
    __syncthreads();
    const uint64_t tile = s_ticket * A.ticket_lanes + (blockIdx.x % A.ticket_lanes);
    if (tile >= A.ntiles) break;
    PH(0);
    const uint64_t t0 = tile * kTile;
    const uint32_t nst = (uint32_t)(A.ntotal - t0 < (uint64_t)kStaged ? A.ntotal - t0 : (uint64_t)kStaged);  // staged
    const uint32_t nown = (uint32_t)(A.nrecs - t0 < (uint64_t)kTile ? A.nrecs - t0 : (uint64_t)kTile);      // owned
    // 1. records -> descriptors in LDS (coalesced 16-byte loads, one division per record).  All of a thread's record
    //    loads are issued before the first is used, then all of its taxon gathers: as a plain loop this was nine times
    //    two DEPENDENT round trips (record, then ref2tax[record]) — 14.5 k of a tile's 87 k clocks, all of it latency.
    {
      constexpr int kRounds = (kStaged + kPB - 1) / kPB;
      constexpr int kBatch = MG_K3_LOAD_BATCH;  // rounds in flight together (registers: 5 per round)
#pragma unroll
      for (int j0 = 0; j0 < kRounds; j0 += kBatch) {
        mg_aln_rec r[kBatch];
        uint32_t tx[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const uint32_t i = (uint32_t)tid + (uint32_t)(j0 + j) * kPB;
          if (j0 + j < kRounds && i < nst) r[j] = A.recs[t0 + i];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const uint32_t i = (uint32_t)tid + (uint32_t)(j0 + j) * kPB;
          if (j0 + j < kRounds && i < nst) tx[j] = A.ref2tax[r[j].ref_new & MG_REC_REF_MASK];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const uint32_t i = (uint32_t)tid + (uint32_t)(j0 + j) * kPB;
          if (j0 + j < kRounds && i < nst) s_desc[slot(i)] = make_desc(r[j], tx[j], A.pct_id);
        }
      }
    }
    __syncthreads();
    PH(1);
    const DescView at{(LdsDescPtr)s_desc, t0, nst, A.recs, A.ref2tax, A.pct_id};

    // 2. every read is classified under BOTH hypotheses by the thread that owns its first line; the thread's reads fold
    //    into one RunFn.  ONE linear walk: the thread's own eight descriptors come out of LDS in one round trip (registers,
    //    static indices), a read's counters run along under both hypotheses ("dropped" is the same read without its
    //    first line), and the line that opens the next read closes the current one (its pair flags are what
    //    process_read is told, :225-226).  Only a read that runs past the thread's own records goes on reading LDS.
    //    (Rounds 1-3 looked for the closing line first and walked the read a second time: two dependent LDS chains per
    //    read and a wavefront as slow as its longest read.)
    const uint32_t l0 = (uint32_t)tid * kItems;
    RunFn mine{kIdentity, {0, 0}};
    {
      uint32_t st0 = 0, st1 = 1;  // state after the reads closed so far, for the thread entered kept / dropped
      bool open = false;
      uint32_t lead = 0;
      Walk w{0, 0, 0, 0, 0}, w1{0, 0, 0, 0, 0};
      // the read [lead, e) is closed by a line with pair flags np (staged), or runs to global record ge (staged == false)
      auto close = [&](bool staged, bool np, uint64_t ge) {
        uint32_t k0 = staged ? classify(w, np) : 3u, k1 = staged ? classify(w1, np) : 3u;
        // the general evaluator for what the counters do not decide — ONE inlined copy for both hypotheses
        // (every copy is ~3 KB of code in the tile loop; the kernel was 42 KB with six of them)
#pragma unroll 1
        for (uint32_t hyp = 0; hyp < 2; ++hyp) {
          if ((hyp ? k1 : k0) != 3u) continue;
          const Verdict v = eval_group(at, t0 + lead + hyp, ge, np, nullptr);
          const uint32_t kk = v.kind | (v.nmm << 2);
          if (hyp) k1 = kk; else k0 = kk;
        }
        const uint32_t ka = st0 ? k1 : k0, kb = st1 ? k1 : k0;
        if (COMMIT) {
          if ((ka & 3u) == 2u) mine.pk[0] += kPkOneMm + (ka >> 2);
          if ((kb & 3u) == 2u) mine.pk[1] += kPkOneMm + (kb >> 2);
        }
        st0 = (ka & 3u) == 0u;
        st1 = (kb & 3u) == 0u;
      };
      // (a rolled loop, the next line requested from LDS before this one is used: one copy of the code above — unrolled
      // over the thread's eight records it was nine, 222 VGPRs and two workgroups per CU)
      const uint32_t own_end = l0 + kItems < nown ? l0 + kItems : nown;
      uint32_t e = l0;
      uint2 d = e < nst ? s_desc[slot(e)] : make_uint2(0u, 0u);
#pragma unroll 1
      while (e < nst) {
        const uint2 dn = e + 1 < nst ? s_desc[slot(e + 1)] : make_uint2(0u, 0u);
        const bool own = e < own_end;
        if (d.y & D_NEW) {
          if (open) { close(true, d.y & (D_P1 | D_P2), t0 + e); open = false; }
          if (!own) break;
          open = true;
          lead = e;
          w = Walk{0, 0, 0, 0, 0};
          w1 = w;
          walk_add(w, d);
          mine.pk[0] += kPkOneRead;
          mine.pk[1] += kPkOneRead;
        } else if (open) {
          walk_add(w, d);
          walk_add(w1, d);
        } else if (!own) {
          break;
        }
        ++e;
        d = dn;
      }
      if (open) {  // ... past the staged window (or to the end of the shard)
        uint64_t ge = t0 + e;
        while (ge < A.ntotal && !(A.recs[ge].ref_new & MG_REC_NEW_BIT)) ++ge;
        if (ge < A.ntotal) close(false, at(ge).y & (D_P1 | D_P2), ge);  // (no closing line: the read is never processed, :259-264)
      }
      mine.out = st0 | (st1 << 1);
    }
    // 3. ordered scan over the workgroup; tile aggregate
    PH(2);
    RunFn tile_fn;
    const RunFn excl = block_scan_runs(mine, s_run, &tile_fn);
    PH(3);
    // 4. publish the aggregate, look back (wavefronts 0 .. kLbWaves-1, 1024 tiles per round), publish the
    //    inclusive prefix
    const uint64_t tile_g = pk_reads(tile_fn.pk[0]);
    uint64_t* const dsc = A.desc + tile * kDescWords;
    if (tid == 0 && tile + 1 < A.ntiles) {
      st_desc(dsc, agg_word(tile_fn.out & 1u, tile_g, pk_mm(tile_fn.pk[0]), pk_ent(tile_fn.pk[0])));
      st_desc(dsc + 1, agg_word((tile_fn.out >> 1) & 1u, tile_g, pk_mm(tile_fn.pk[1]), pk_ent(tile_fn.pk[1])));
      st_desc(dsc + 2, ST_AGG);
    }
    TileFn acc = fn_identity();  // thread 0: everything between the nearest inclusive prefix and this tile
    if (tile > 0) {
      for (int64_t base = (int64_t)tile - 1;; base -= kLbWaves * 64 * kWin) {
        if (wave < kLbWaves) {
          bool found;
          const TileFn f = look_window(A.desc, base - (int64_t)wave * 64 * kWin, &found, s_lbuf[wave]);
          if (lane == 0) { s_lb[wave] = f; s_lbf[wave] = found ? 1u : 0u; }
        }
        __syncthreads();
        if (tid == 0) {
          uint32_t done = 0;
          for (int w = 0; w < kLbWaves && !done; ++w) {  // near -> far
            acc = fn_then(s_lb[w], acc);
            done = s_lbf[w];
          }
          s_lbf[kLbWaves] = done;
        }
        __syncthreads();
        if (s_lbf[kLbWaves]) break;
      }
    }
    PH(4);
    if (tid == 0) {
      Incoming in{acc.out, acc.g, A.incoming ? acc.r[1] : acc.r[0], A.incoming ? acc.e[1] : acc.e[0]};
      const uint32_t x_in = (in.map >> A.incoming) & 1u;
      const uint32_t map_incl = map_then(in.map, tile_fn.out);
      const uint64_t tpk = x_in ? tile_fn.pk[1] : tile_fn.pk[0];
      const uint64_t g_incl = in.g + tile_g, r_incl = in.r + pk_mm(tpk), e_incl = in.e + pk_ent(tpk);
      if (tile + 1 < A.ntiles) {
        st_desc(dsc, ST_PREFIX | ((uint64_t)map_incl << 60) | g_incl);
        st_desc(dsc + 1, ST_PREFIX | r_incl);
        st_desc(dsc + 2, ST_PREFIX | e_incl);
      } else {  // the last tile closes the shard
        A.out_tot[0] = map_incl; A.out_tot[1] = g_incl;
        if (COMMIT) { A.out_tot[2] = e_incl; A.out_tot[3] = r_incl; A.mm_offsets[r_incl] = e_incl; }
      }
      s_bcast[0] = x_in; s_bcast[1] = in.g; s_bcast[2] = in.r; s_bcast[3] = in.e;
    }
    __syncthreads();
    PH(5);
    if (COMMIT) {
      // 5. commit with the true state: the thread's exclusive prefix, evaluated at the tile's incoming state
      const uint32_t x_in = (uint32_t)s_bcast[0];
      uint32_t st = (excl.out >> x_in) & 1u;
      const uint64_t ex = x_in ? excl.pk[1] : excl.pk[0];
      uint64_t eo = s_bcast[3] + pk_ent(ex);
      uint64_t so = s_bcast[2] + pk_mm(ex);
      uint64_t gidx = A.group_base + s_bcast[1] + pk_reads(ex);
      const uint64_t gidx0 = gidx;      // the thread opens at most kItems reads: gidx0 .. gidx0 + 7
      uint32_t qn = 0, qmask = 0;       // its queue of uniquely mapped reads (s_queue[tid + kPB * i]) and which reads those are
      uint32_t* const s_queue = reinterpret_cast<uint32_t*>(&s_lbuf[0][0]);  // (the look-back of this tile is over)
      uint32_t ambig = (tid == 0 && tile == 0 && A.first_shard) ? 1u : 0u;  // the phantom first boundary (:155-156)
      bool used_priv = false;
      // the same linear walk with the TRUE state: a read's counters run along from the line that opens it (without that
      // line when the state says "dropped") and the read is committed by the line that closes it
      bool open = false;
      uint32_t lead = 0;
      uint64_t my_gidx = 0;
      Walk w{0, 0, 0, 0, 0};
      // commits the read whose lines are [lead + st, e): e staged (e_loc) or global (ge, staged == false)
      auto commit_read = [&](bool staged, uint32_t e_loc, bool np, uint64_t ge) {
        uint32_t kind, tax, nmm = 0;
        uint64_t hitlen;
        bool slow = !staged;
        if (staged) {
          const uint32_t k = classify(w, np);
          kind = k & 3u; nmm = k >> 2; tax = w.tax; hitlen = w.hsum;
          slow = kind == 3u;
        }
        if (slow) {  // verdict first; a multimapped read's taxon list (SAM order) in a second round of the same code
          uint32_t* out = nullptr;
#pragma unroll 1
          for (int round = 0; round < 2; ++round) {
            const Verdict v = eval_group(at, t0 + lead + st, ge, np, out);
            kind = v.kind; tax = v.tax; nmm = v.nmm; hitlen = v.hitlen;
            if (kind != 2u) break;
            out = A.mm_tax + eo;
          }
        }
        if (kind == 0) {
          ++ambig;
        } else if (kind == 1) {
          if (A.use_lds_hist && hitlen < kQueueLenLimit) {
            // binned after the walk, every lane of the wavefront together (below): here only the lanes whose read closes on
            // THIS line are active, and the bin look-up + atomics were paid once per line of the walk
            s_queue[tid + kPB * qn++] = (tax << kQueueLenBits) | (uint32_t)hitlen;
            qmask |= 1u << (uint32_t)(my_gidx - gidx0);
          } else {
            atomicAdd(&A.g_count[tax], 1ull);
            atomicAdd(&A.g_bases[tax], (unsigned long long)hitlen);
            // first_seen only ever decreases: a (possibly stale) value at or below ours means nothing to do
            if (__atomic_load_n(&A.g_first[tax], __ATOMIC_RELAXED) > my_gidx)
              atomicMin(&A.g_first[tax], (unsigned long long)my_gidx);
          }
        } else {
          if (!slow) {
            uint64_t wpos = eo;
            for (uint32_t q = lead + st; q < e_loc; ++q) {  // every kept line (:172-173)
              const uint2 dq = s_desc[slot(q)];
              if (!(dq.y & D_REJ)) A.mm_tax[wpos++] = dq.x;
            }
          }
          A.mm_offsets[so] = eo;
          A.mm_hitlen[so] = hitlen;
          A.mm_read[so] = my_gidx;
          eo += nmm;
          ++so;
        }
        st = kind == 0;
      };
      const uint32_t own_end = l0 + kItems < nown ? l0 + kItems : nown;
      uint32_t e = l0;
      uint2 d = e < nst ? s_desc[slot(e)] : make_uint2(0u, 0u);
#pragma unroll 1
      while (e < nst) {
        const uint2 dn = e + 1 < nst ? s_desc[slot(e + 1)] : make_uint2(0u, 0u);
        const bool own = e < own_end;
        if (d.y & D_NEW) {
          if (open) { commit_read(true, e, d.y & (D_P1 | D_P2), t0 + e); open = false; }
          if (!own) break;
          open = true;
          lead = e;
          my_gidx = gidx++;
          w = Walk{0, 0, 0, 0, 0};
          if (!st) walk_add(w, d);
        } else if (open) {
          walk_add(w, d);
        } else if (!own) {
          break;
        }
        ++e;
        d = dn;
      }
      if (open) {
        uint64_t ge = t0 + e;
        while (ge < A.ntotal && !(A.recs[ge].ref_new & MG_REC_NEW_BIT)) ++ge;
        if (ge < A.ntotal) commit_read(false, e, at(ge).y & (D_P1 | D_P2), ge);  // (else: never processed)
      }
      PH(8);
      // the thread's uniquely mapped reads into the bins: entry i belongs to the read of the i-th set bit of qmask
      const uint32_t max_probe = A.use_lds_hist == 2 && s_nkeys <= kHashCrowded ? kHashProbeFree : kHashProbe;
      for (uint32_t qi = 0; qmask; ++qi) {
        const uint32_t ord = (uint32_t)__builtin_ctz(qmask);
        qmask &= qmask - 1u;
        const uint32_t ent = s_queue[tid + kPB * qi];
        const uint32_t tax = ent >> kQueueLenBits, hitlen = ent & (kQueueLenLimit - 1u);
        const uint32_t rel = (uint32_t)(gidx0 + ord - A.group_base);
        uint32_t bin = tax;
        bool in_lds = A.use_lds_hist == 1;
        if (A.use_lds_hist == 2) {
          // Open addressing with double hashing over the packed words, at most kHashProbe probes: the usual case is ONE 8-byte
          // LDS read and a compare of its key field.  Between two flushes a key never changes once set and a taxon always
          // walks the same slots in the same order, so it cannot sit in two of them; two lanes that race for an empty slot are
          // settled by the compare-and-swap (the loser of another taxon goes on to its next slot).
          static_assert(kHashSlots == 2048, "slot index width");
          const uint32_t hh = tax * 2654435761u;
          uint32_t sl = hh >> 21;                              // 11 bits
          const uint32_t stride = ((hh >> 10) & 2046u) | 1u;   // odd: every slot is reached
          const unsigned long long mine = (unsigned long long)(tax + 1u) << kHKeyShift;
          for (uint32_t step = 0; step < max_probe && !in_lds; ++step) {
            unsigned long long w = h_pack[sl];
            if (w == 0ull) {
              const unsigned long long old = atomicCAS(&h_pack[sl], 0ull, mine);
              w = old == 0ull ? mine : old;
              if (old == 0ull) atomicAdd(&s_nkeys, 1u);
            }
            if ((w >> kHKeyShift) == (unsigned long long)(tax + 1u)) { bin = sl; in_lds = true; }
            sl = (sl + stride) & (kHashSlots - 1);
          }
        }
        if (in_lds) {
          atomicAdd(&h_pack[bin], (1ull << (A.use_lds_hist == 2 ? kHCountShift : kBinCountShift)) + hitlen);
          if (h_first[bin] > rel) atomicMin(&h_first[bin], rel);
        } else {  // no slot within kHashProbe probes: this workgroup's private copy
          unsigned long long* pp = A.priv_pack + (size_t)blockIdx.x * A.ntax + tax;
          uint32_t* pf = A.priv_first + (size_t)blockIdx.x * A.ntax + tax;
          __hip_atomic_fetch_add(pp, (1ull << kBinCountShift) + hitlen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (*pf > rel) __hip_atomic_fetch_min(pf, rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          used_priv = true;
        }
      }
      PH(9);
      if (ambig) atomicAdd(&s_ambig, (unsigned long long)ambig);
      if (used_priv) {
        A.priv_used[1 + blockIdx.x] = 1u;
        if (__hip_atomic_load(A.priv_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicExch(A.priv_used, 1u);
      }
      if (tid == 0) s_groups += tile_g;
      if (A.use_lds_hist && ++tiles_binned == A.flush_tiles) {  // before a packed field can overflow
        __syncthreads();
        flush_bins(true);
        tiles_binned = 0;
      }
    }
    __syncthreads();  // s_desc / s_bcast / s_ticket are reused by the next tile
    PH(6);
  }
  if (COMMIT) {
    __syncthreads();
    if (A.use_lds_hist == 2) {
      // The last flush is a plain copy of the table: every workgroup hashes alike, so slot s holds the SAME taxon in nearly
      // all of them and k_bins_reduce adds a slot's 768 words up in registers — three atomics per run of equal keys instead of
      // three per workgroup and taxon (2.3 M of them, all at the end of the pass: 0.06-0.08 of 0.43 ms by the phase clocks).
      for (uint32_t t = tid; t < kHashSlots; t += kPB) {
        A.dump_pack[(size_t)blockIdx.x * kHashSlots + t] = h_pack[t];
        A.dump_first[(size_t)blockIdx.x * kHashSlots + t] = h_first[t];
      }
    } else if (A.use_lds_hist) flush_bins(false);
    if (tid == 0) {
      if (s_groups) atomicAdd(&A.g_scalars[0], s_groups);
      if (s_ambig) atomicAdd(&A.g_scalars[1], s_ambig);
    }
  }
  PH(11);
  PH_FLUSH(COMMIT);
}

// The two instantiations as kernels.  The map-only pass fits 128 VGPRs (4 wavefronts per SIMD: four workgroups
// per CU instead of three); the commit pass keeps its 154 rather than spill.
template <bool COMMIT>
__global__ __launch_bounds__(kPB) void k_profile_pass(const PassArgs A);
template <>
__global__ __launch_bounds__(kPB) __attribute__((amdgpu_waves_per_eu(4))) void k_profile_pass<false>(const PassArgs A) {
  profile_pass_body<false>(A);
}
#ifndef MG_K3_COMMIT_WAVES
#define MG_K3_COMMIT_WAVES 3
#endif
template <>
__global__ __launch_bounds__(kPB) __attribute__((amdgpu_waves_per_eu(MG_K3_COMMIT_WAVES))) void k_profile_pass<true>(const PassArgs A) {
  profile_pass_body<true>(A);
}

// k_bins_reduce's view of the dumped tables: a block takes kRedSlots consecutive slots x kRedSlices shares of a quarter of the tables
constexpr uint32_t kRedSlots = 32, kRedSlices = 8, kRedCache = 4, kRedY = 4;
constexpr int kRedBatch = 24;
static_assert(kRedSlots * kRedSlices == 256 && kHashSlots % kRedSlots == 0, "one thread per (slot, share)");
struct RedEnt { uint32_t key, first; unsigned long long cnt, bas; };  // key = taxon + 1, 0 = free

// Hashed mode, after the pass: the workgroups' private bins summed into the accumulators — one thread per taxon walks
// the copies (consecutive threads read consecutive words of a copy) and leaves them zeroed for the next pass.  Nothing
// to do (and nothing read) when no workgroup overflowed its LDS bins.
__global__ __launch_bounds__(256) void k_bins_reduce(unsigned long long* __restrict__ priv_pack, uint32_t* __restrict__ priv_first,
                                                     uint32_t* __restrict__ priv_used, unsigned long long* __restrict__ xcd_bins,
                                                     const unsigned long long* __restrict__ dump_pack, const uint32_t* __restrict__ dump_first,
                                                     uint32_t ncopies, uint32_t ntax,
                                                     uint64_t group_base, unsigned long long* __restrict__ g_count,
                                                     unsigned long long* __restrict__ g_bases,
                                                     unsigned long long* __restrict__ g_first) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  // blockIdx.y: a slice of the workgroups' copies (enough blocks to fill the GPU: the private copies are ~100 MB at 10 001 taxa)
  const uint32_t per = (ncopies + gridDim.y - 1) / gridDim.y;
  const uint32_t cbeg = blockIdx.y * per, cend = cbeg + per < ncopies ? cbeg + per : ncopies;
  // The workgroups' last tables.  Every workgroup hashes alike, but which of two colliding taxa took a slot first differs from
  // workgroup to workgroup: a slot holds two or three different keys across them.  A thread reads slot s of 1/32 of the tables into a
  // four-key cache in registers, the eight threads of a slot leave their caches in LDS and one of them merges the 32 entries: three
  // atomics per slot, key and quarter of the workgroups (summing only RUNS of equal keys left 0.7 M atomics — 43 us at the ~16 G/s
  // the part retires them — where this leaves tens of thousands).
  if (blockIdx.x < kHashSlots / kRedSlots && blockIdx.y < kRedY) {
    __shared__ RedEnt s_ent[kRedSlots][kRedSlices][kRedCache];
    const uint32_t sl = threadIdx.x / kRedSlots, sloc = threadIdx.x % kRedSlots;  // (consecutive lanes: consecutive slots, 256 B of a table)
    const uint32_t slot = blockIdx.x * kRedSlots + sloc;
    const uint32_t nsl = kRedY * kRedSlices, dper = (ncopies + nsl - 1) / nsl;
    const uint32_t dbeg = (blockIdx.y * kRedSlices + sl) * dper, dend = dbeg + dper < ncopies ? dbeg + dper : ncopies;
    auto emit = [&](const RedEnt& e) {
      if (e.cnt) {
        const uint32_t tax = e.key - 1u;
        atomicAdd(&g_count[tax], e.cnt);
        atomicAdd(&g_bases[tax], e.bas);
        atomicMin(&g_first[tax], (unsigned long long)(group_base + e.first));
      }
    };
    RedEnt cache[kRedCache];
#pragma unroll
    for (int j = 0; j < kRedCache; ++j) cache[j] = RedEnt{0u, 0xffffffffu, 0ull, 0ull};
    auto add = [&](uint32_t key, unsigned long long n1, unsigned long long b1, uint32_t f1) {
      bool placed = false;
#pragma unroll
      for (int j = 0; j < kRedCache; ++j) {  // (static indices: the cache stays in registers)
        const bool here = !placed && (cache[j].key == key || cache[j].key == 0u);
        if (here) {
          cache[j].key = key; cache[j].cnt += n1; cache[j].bas += b1;
          cache[j].first = f1 < cache[j].first ? f1 : cache[j].first;
        }
        placed = placed || here;
      }
      if (!placed) emit(RedEnt{key, f1, n1, b1});  // a fifth key in one slot: straight to the accumulators
    };
    for (uint32_t c0 = dbeg; c0 < dend; c0 += kRedBatch) {  // (a thread's whole share in ONE round trip at 768 workgroups)
      unsigned long long v[kRedBatch];
      uint32_t f[kRedBatch];
#pragma unroll
      for (int u = 0; u < kRedBatch; ++u) {
        const uint32_t c = c0 + u;
        v[u] = c < dend ? dump_pack[(size_t)c * kHashSlots + slot] : 0ull;
        f[u] = c < dend ? dump_first[(size_t)c * kHashSlots + slot] : 0xffffffffu;
      }
#pragma unroll
      for (int u = 0; u < kRedBatch; ++u) {
        const unsigned long long n1 = (v[u] >> kHCountShift) & 0x7fffull;
        if (n1) add((uint32_t)(v[u] >> kHKeyShift), n1, v[u] & ((1ull << kHCountShift) - 1), f[u]);  // (0: empty, or claimed and never counted in)
      }
    }
#pragma unroll
    for (int j = 0; j < kRedCache; ++j) s_ent[sloc][sl][j] = cache[j];
    __syncthreads();
    if (sl == 0) {
#pragma unroll
      for (int j = 0; j < kRedCache; ++j) cache[j] = RedEnt{0u, 0xffffffffu, 0ull, 0ull};
      for (uint32_t q = 0; q < kRedSlices; ++q)
#pragma unroll
        for (int j = 0; j < kRedCache; ++j) {
          const RedEnt e = s_ent[sloc][q][j];
          if (e.cnt) add(e.key, e.cnt, e.bas, e.first);
        }
#pragma unroll
      for (int j = 0; j < kRedCache; ++j) emit(cache[j]);
    }
  }
  if (blockIdx.y == 0 && t < ntax) {  // the XCDs' copies of the accumulators (always: the LDS bins are flushed into them)
    unsigned long long cnt = 0, bas = 0, first = ~0ull;
#pragma unroll
    for (uint32_t x = 0; x < kXcds; ++x) {
      unsigned long long* b = xcd_bins + (size_t)x * 3 * ntax;
      const unsigned long long c1 = b[t];
      if (c1) {
        cnt += c1; bas += b[ntax + t];
        const unsigned long long f1 = b[2 * (size_t)ntax + t];
        first = f1 < first ? f1 : first;
        b[t] = 0; b[ntax + t] = 0; b[2 * (size_t)ntax + t] = ~0ull;
      }
    }
    if (cnt) {
      atomicAdd(&g_count[t], cnt);
      atomicAdd(&g_bases[t], bas);
      atomicMin(&g_first[t], first);
    }
  }
  if (*priv_used != 0u && t < ntax) {  // (no workgroup overflowed its LDS bins: nothing read)
    unsigned long long cnt = 0, bas = 0;
    uint32_t first = 0xffffffffu;
    for (uint32_t c0 = cbeg; c0 < cend; c0 += 8) {
      unsigned long long v[8];
      uint32_t f[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t c = c0 + u;
        const bool used = c < cend && priv_used[1 + c] != 0u;  // (uniform: most workgroups never leave their LDS bins)
        v[u] = used ? priv_pack[(size_t)c * ntax + t] : 0ull;
        f[u] = used ? priv_first[(size_t)c * ntax + t] : 0xffffffffu;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t c = c0 + u;
        if (v[u]) {
          cnt += v[u] >> kBinCountShift;
          bas += v[u] & ((1ull << kBinCountShift) - 1);
          first = f[u] < first ? f[u] : first;
          priv_pack[(size_t)c * ntax + t] = 0ull;
          priv_first[(size_t)c * ntax + t] = 0xffffffffu;
        }
      }
    }
    if (cnt) {  // (the accumulators may already hold other shards' sums)
      atomicAdd(&g_count[t], cnt);
      atomicAdd(&g_bases[t], bas);
      atomicMin(&g_first[t], (unsigned long long)(group_base + first));
    }
  }
  // (the flags are zeroed by the NEXT pass's k_pass_prepare.  Counting the blocks that are through and letting the last one do it
  // cost 46 us: 2048 returning atomics on one address)
}

// Before a pass: tile descriptors + ticket = 0 and, when asked, the accumulators of a fresh batch (one launch).
// flags: the hashed bins' overflow flags of the pass before (read by its k_bins_reduce, which ran ahead of this on the stream).
__global__ void k_pass_prepare(uint64_t* __restrict__ desc, uint64_t ndesc, uint64_t* __restrict__ count,
                               uint64_t* __restrict__ bases, uint64_t* __restrict__ first, uint64_t* __restrict__ scalars,
                               uint32_t ntax, uint32_t* __restrict__ flags, uint32_t nflags) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t i = i0; i < ndesc; i += stride) desc[i] = 0;
  for (uint64_t i = i0; i < nflags; i += stride) flags[i] = 0;
  if (count) {
    for (uint64_t t = i0; t < ntax; t += stride) { count[t] = 0; bases[t] = 0; first[t] = ~0ull; }
    if (i0 < 2) scalars[i0] = 0;
  }
}

// Accumulators of one batch: count = bases = 0, first_seen = UINT64_MAX, scalars = 0 (one launch).
__global__ void k_acc_reset(uint64_t* __restrict__ count, uint64_t* __restrict__ bases, uint64_t* __restrict__ first,
                            uint64_t* __restrict__ scalars, uint32_t ntax) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < ntax; t += stride) { count[t] = 0; bases[t] = 0; first[t] = ~0ull; }
  if (blockIdx.x == 0 && threadIdx.x < 2) scalars[threadIdx.x] = 0;
}

// resolve_multi_prop (scripts/map_and_profile.py:269-312) over the multimapped CSR the pass left on the device:
// one thread per multimapped read.  The DISTINCT taxa of the read that still have a weight (NaN = dropped by the
// read cutoff or never seen uniquely, :180-188,:428) share its hitlen in proportion to their weights; the shares are
// summed per taxon (LDS-privatised for ntax <= 2048, then global f64 atomics).  Floating-point sums are
// order-dependent: the result equals the host version to ~1e-15 relative, not bit for bit (the reference's own
// order is a Python set's).
__global__ __launch_bounds__(256) void k_resolve_multimapped(const uint64_t* __restrict__ mm_offsets,
                                                             const uint32_t* __restrict__ mm_tax,
                                                             const uint64_t* __restrict__ mm_hitlen,
                                                             const uint64_t* __restrict__ tot,
                                                             const double* __restrict__ weight,
                                                             const double* __restrict__ genome_len, uint32_t ntax,
                                                             uint32_t use_lds, double* __restrict__ extra) {
  extern __shared__ double s_extra[];
  if (use_lds) {
    for (uint32_t t = threadIdx.x; t < ntax; t += blockDim.x) s_extra[t] = 0.0;
    __syncthreads();
  }
  const uint64_t nreads = tot[3];
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nreads; i += stride) {
    const uint64_t o0 = mm_offsets[i], o1 = mm_offsets[i + 1];
    const double hitlen = (double)mm_hitlen[i];
    auto first_kept = [&](uint64_t e, uint32_t t, double w) {  // e is the first entry of taxon t, and t has a weight
      if (w != w) return false;
      for (uint64_t f = o0; f < e; ++f)
        if (mm_tax[f] == t) return false;
      return true;
    };
    double denom = 0.0;
    for (uint64_t e = o0; e < o1; ++e) {
      const uint32_t t = mm_tax[e];
      const double w = weight[t];
      if (first_kept(e, t, w)) denom += w;
    }
    if (denom == 0.0) continue;  // no taxon left (:280-281) or all weights zero (:286-287)
    for (uint64_t e = o0; e < o1; ++e) {
      const uint32_t t = mm_tax[e];
      const double w = weight[t];
      if (!first_kept(e, t, w)) continue;
      double part = (w / denom) * hitlen;
      if (genome_len) part /= genome_len[t];
      if (use_lds) atomicAdd(&s_extra[t], part); else atomicAdd(&extra[t], part);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < ntax; t += blockDim.x)
      if (s_extra[t] != 0.0) atomicAdd(&extra[t], s_extra[t]);
  }
}

}  // namespace mg

using namespace mg;

struct mg_profile {
  const mg_aln_rec* d_recs = nullptr;
  uint64_t nrecs = 0, ntotal = 0;
  const uint32_t* d_ref2tax = nullptr;
  uint32_t nref = 0, ntax = 0;
  double pct_id = 0.5;
  uint64_t ntiles = 0;
  DevBuf desc, tot;          // tile descriptors + ticket (last word); totals of the last pass
  uint8_t map[2] = {0, 1};
  uint64_t ngroups = 0;
  bool have_map = false;     // composed state map / read count known (map-only pass, or read back after commit)
  bool map_launched = false; // the map-only pass is queued (mg_profile_map_launch); its totals are not read back yet
  bool have_mm = false;      // multimapped totals read back (lazily)
  bool committed = false;
  // multimapped CSR (device)
  DevBuf mm_offsets, mm_tax, mm_hitlen, mm_read;
  uint64_t mm_nreads = 0, mm_nentries = 0;
};

namespace {

// One chained-scan pass over the shard.  commit == false: composed state map + read count only.
int launch_pass(mg_profile* p, bool commit, uint32_t incoming, uint32_t first_shard, uint64_t group_base,
                uint64_t* d_count, uint64_t* d_bases, uint64_t* d_first_seen, uint64_t* d_scalars, bool reset_acc = false) {
  Context& c = ctx();
  hipStream_t st = stage_c_stream();
  const uint64_t desc_bytes = (p->ntiles * kDescWords + (uint64_t)kTicketLanes * kTicketStride) * sizeof(uint64_t);
  // (+ one look-back window of slack: a window clamped to the start of a short array is read in full)
  if (!p->desc.p) MG_TRY(p->desc.alloc(desc_bytes + (uint64_t)kWinTiles * kDescWords * sizeof(uint64_t)));
  if (!p->tot.p) MG_TRY(p->tot.alloc(4 * sizeof(uint64_t)));
  {
    const uint64_t ndesc = desc_bytes / sizeof(uint64_t);
    const uint64_t work = ndesc > p->ntax ? ndesc : p->ntax;
    hipLaunchKernelGGL(k_pass_prepare, dim3(grid_for(work, 256, (unsigned)c.num_cus * 4)), dim3(256), 0, st,
                       p->desc.as<uint64_t>(), ndesc, reset_acc ? d_count : (uint64_t*)nullptr, d_bases, d_first_seen, d_scalars,
                       p->ntax, c.k3_flags, c.k3_flags ? c.k3_nflags : 0u);
    MG_HIP(hipGetLastError());
  }
  PassArgs a{};
  a.recs = p->d_recs; a.nrecs = p->nrecs; a.ntotal = p->ntotal;
  a.ref2tax = p->d_ref2tax; a.pct_id = p->pct_id;
  a.desc = p->desc.as<uint64_t>();
  a.ticket = reinterpret_cast<unsigned long long*>(p->desc.as<uint64_t>() + p->ntiles * kDescWords);
  a.ntiles = p->ntiles;
  a.ticket_lanes = kTicketLanes;  // (lowered below to the grid size: every lane needs a workgroup drawing from it)
  a.incoming = incoming; a.first_shard = first_shard; a.group_base = group_base;
  a.ntax = p->ntax;
  a.out_tot = p->tot.as<uint64_t>();
  if (!commit) {
    ProfScope ps("profile_map", st);
    const unsigned grid = grid_for(p->ntiles, 1, (unsigned)c.num_cus * 4);
    if (grid < a.ticket_lanes) a.ticket_lanes = grid;
    hipLaunchKernelGGL(k_profile_pass<false>, dim3(grid), dim3(kPB), 0, st, a);
    MG_HIP(hipGetLastError());
    return MG_OK;
  }
  a.use_lds_hist = p->nrecs >= 0xffffffffull ? 0u : p->ntax <= 4096 ? 1u : p->ntax <= kHashMaxTax ? 2u : 0u;
  if (a.use_lds_hist == 1 && dbg("k3_hashed")) a.use_lds_hist = 2;  // tests: the hashed bins on a small taxonomy
  const size_t lds = a.use_lds_hist == 0 ? 0
                   : a.use_lds_hist == 1 ? (size_t)p->ntax * (sizeof(unsigned long long) + sizeof(uint32_t))
                                         : kHashSlots * (sizeof(unsigned long long) + sizeof(uint32_t));
  a.g_count = (unsigned long long*)d_count; a.g_bases = (unsigned long long*)d_bases;
  a.g_first = (unsigned long long*)d_first_seen; a.g_scalars = (unsigned long long*)d_scalars;
  a.mm_offsets = p->mm_offsets.as<uint64_t>(); a.mm_tax = p->mm_tax.as<uint32_t>();
  a.mm_hitlen = p->mm_hitlen.as<uint64_t>(); a.mm_read = p->mm_read.as<uint64_t>();
  ProfScope ps("profile_pass", st);
  // every workgroup flushes its private histogram once: few, long-lived workgroups
  const unsigned per_cu = lds > 40 * 1024 ? 2u : 3u;
  unsigned grid = grid_for(p->ntiles, 1, (unsigned)c.num_cus * per_cu);
  a.flush_tiles = a.use_lds_hist == 2 ? kHashFlushTiles : kFlushTiles;
  // test hooks: a workgroup at this size lives for ~8 tiles and never reaches a mid-life flush — fewer workgroups, shorter intervals
  { const int64_t v = dbg("k3_flush_tiles"); if (v > 0 && (uint64_t)v < a.flush_tiles) a.flush_tiles = (uint32_t)v; }
  { const int64_t v = dbg("k3_grid"); if (v > 0 && (uint64_t)v < grid) grid = (unsigned)v; }
  if (grid < a.ticket_lanes) a.ticket_lanes = grid;
  if (a.use_lds_hist == 2) {
    // the private overflow bins: a grow-only buffer of the library, all-zero (first-seen words all-ones) between passes —
    // zeroed when it is (re)allocated, re-zeroed by k_bins_reduce wherever a pass wrote
    const uint64_t nb = (uint64_t)c.num_cus * 3 * p->ntax;
    const uint64_t nflags = (uint64_t)c.num_cus * 3 + 1;  // [0] any, [1 + w] workgroup w
    const uint64_t nx = (uint64_t)kXcds * 3 * p->ntax;  // the XCDs' accumulator copies
    const uint64_t nd = (uint64_t)c.num_cus * 3 * kHashSlots;  // the workgroups' last tables (written whole by every pass)
    const uint64_t bytes = (nb + nx + nd) * sizeof(unsigned long long) + (nb + nd) * sizeof(uint32_t) + nflags * sizeof(uint32_t) + 64;
    void*& priv_ptr = c.k3_priv_ptr;
    uint64_t& priv_nb = c.k3_priv_nb;
    uint8_t* buf = (uint8_t*)scratch("k3_priv_bins", bytes);
    if (!buf) return MG_ERR_NOMEM;
    a.priv_pack = reinterpret_cast<unsigned long long*>(buf);
    a.xcd_bins = a.priv_pack + nb;
    a.dump_pack = a.xcd_bins + nx;
    a.priv_first = reinterpret_cast<uint32_t*>(a.dump_pack + nd);
    a.dump_first = a.priv_first + nb;
    a.priv_used = a.dump_first + nd;
    if (buf != priv_ptr || nb != priv_nb) {  // new block, or a different partition of it: establish the invariant
      MG_HIP(hipMemsetAsync(a.priv_pack, 0, nb * sizeof(unsigned long long), st));
      MG_HIP(hipMemsetAsync(a.priv_first, 0xff, nb * sizeof(uint32_t), st));
      MG_HIP(hipMemsetAsync(a.xcd_bins, 0, nx * sizeof(unsigned long long), st));
      for (uint32_t x = 0; x < kXcds; ++x)
        MG_HIP(hipMemsetAsync(a.xcd_bins + ((size_t)x * 3 + 2) * p->ntax, 0xff, p->ntax * sizeof(unsigned long long), st));
      MG_HIP(hipMemsetAsync(a.priv_used, 0, nflags * sizeof(uint32_t), st));
      priv_ptr = buf;
      priv_nb = nb;
      c.k3_flags = a.priv_used;  // (zeroed again by every k_pass_prepare from now on)
      c.k3_nflags = (uint32_t)nflags;
    }
  }
  hipLaunchKernelGGL(k_profile_pass<true>, dim3(grid), dim3(kPB), lds, st, a);
  MG_HIP(hipGetLastError());
  if (a.use_lds_hist == 2) {
    const unsigned rx = (p->ntax + 255) / 256 > kHashSlots / kRedSlots ? (p->ntax + 255) / 256 : kHashSlots / kRedSlots;
    hipLaunchKernelGGL(k_bins_reduce, dim3(rx, 32), dim3(256), 0, st, a.priv_pack, a.priv_first, a.priv_used, a.xcd_bins, a.dump_pack, a.dump_first, grid,
                       p->ntax, group_base, a.g_count, a.g_bases, a.g_first);
    MG_HIP(hipGetLastError());
  }
  return MG_OK;
}

// Totals of the last pass ([0] composed map, [1] reads) are fetched on first use.
int fetch_map(mg_profile* p) {
  if (p->have_map || p->nrecs == 0) { p->have_map = true; return MG_OK; }
  hipStream_t st = stage_c_stream();
  if (!p->committed && !p->map_launched)  // nobody ran over the shard yet: the map-only pass
    MG_TRY(launch_pass(p, false, 0, 0, 0, nullptr, nullptr, nullptr, nullptr));
  uint64_t* h_tot = host_words() + 16;
  MG_HIP(hipMemcpyAsync(h_tot, p->tot.p, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  p->map[0] = (uint8_t)(h_tot[0] & 1u);
  p->map[1] = (uint8_t)((h_tot[0] >> 1) & 1u);
  p->ngroups = h_tot[1];
  p->have_map = true;
  return MG_OK;
}

int fetch_mm(mg_profile* p) {
  if (p->have_mm || p->nrecs == 0) { p->have_mm = true; return MG_OK; }
  hipStream_t st = stage_c_stream();
  uint64_t* h_tot = host_words() + 16;
  MG_HIP(hipMemcpyAsync(h_tot + 2, p->tot.as<uint64_t>() + 2, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  p->mm_nentries = h_tot[2];
  p->mm_nreads = h_tot[3];
  p->have_mm = true;
  return MG_OK;
}

}  // namespace

extern "C" {

int mg_profile_begin_dev(const mg_aln_rec* d_recs, uint64_t nrecs, int has_lookahead, const uint32_t* d_ref2tax,
                         uint32_t nref, uint32_t ntax, double pct_id, mg_profile** out) {
  MG_REQUIRE_READY();
  if (!out) return fail(MG_ERR_ARG, "null out handle");
  *out = nullptr;
  if (nrecs > 0 && (!d_recs || !d_ref2tax)) return fail(MG_ERR_ARG, "null device input");
  std::unique_ptr<mg_profile> p(new mg_profile());
  p->d_recs = d_recs;
  p->nrecs = nrecs;
  p->ntotal = nrecs + (has_lookahead ? 1 : 0);
  p->d_ref2tax = d_ref2tax;
  p->nref = nref;
  p->ntax = ntax;
  p->pct_id = pct_id;
  p->ntiles = (nrecs + kTile - 1) / kTile;
  // nothing is launched here: a single shard commits in one pass; a shard of a multi-GPU job first asks for its
  // composed state map (mg_profile_state_map), which runs the map-only pass.
  *out = p.release();
  return MG_OK;
}

int mg_profile_acc_reset(uint64_t* d_count, uint64_t* d_bases, uint64_t* d_first_seen, uint64_t* d_scalars, uint32_t ntax) {
  MG_REQUIRE_READY();
  if (!d_count || !d_bases || !d_first_seen || !d_scalars) return fail(MG_ERR_ARG, "null argument");
  hipLaunchKernelGGL(k_acc_reset, dim3(grid_for((uint64_t)ntax + 2, 256, 64)), dim3(256), 0, stage_c_stream(), d_count, d_bases,
                     d_first_seen, d_scalars, ntax);
  MG_HIP(hipGetLastError());
  return MG_OK;
}

int mg_profile_map_launch(mg_profile* p) {
  MG_REQUIRE_READY();
  if (!p) return fail(MG_ERR_ARG, "null profile");
  if (p->have_map || p->map_launched || p->committed || p->nrecs == 0) return MG_OK;
  MG_TRY(launch_pass(p, false, 0, 0, 0, nullptr, nullptr, nullptr, nullptr));
  p->map_launched = true;
  return MG_OK;
}

// out[0], out[1] = the shard's composed state map, out[2] = its read count — from the totals the map-only pass left
// on the device (tot == nullptr: an empty shard, the identity map).
__global__ void k_map_words(const uint64_t* __restrict__ tot, long long* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  const uint64_t m = tot ? tot[0] : 2ull;  // bit x = f(x); identity = 0b10
  out[0] = (long long)(m & 1ull);
  out[1] = (long long)((m >> 1) & 1ull);
  out[2] = tot ? (long long)tot[1] : 0;
}

int mg_profile_map_words_dev(mg_profile* p, int64_t* d_out3) {
  MG_REQUIRE_READY();
  if (!p || !d_out3) return fail(MG_ERR_ARG, "null argument");
  if (p->committed) return fail(MG_ERR_STATE, "the commit pass has overwritten the map-only totals");
  MG_TRY(mg_profile_map_launch(p));
  MG_TRY(mg_stage_c_join());  // the map-only pass runs on stage C's stream; the words are written on the main one
  hipLaunchKernelGGL(k_map_words, dim3(1), dim3(64), 0, ctx().stream,
                     p->nrecs ? (const uint64_t*)p->tot.as<uint64_t>() : (const uint64_t*)nullptr,
                     reinterpret_cast<long long*>(d_out3));
  MG_HIP(hipGetLastError());
  return MG_OK;
}

int mg_profile_state_map(const mg_profile* p, uint8_t map[2]) {
  if (!p || !map) return fail(MG_ERR_ARG, "null argument");
  MG_REQUIRE_READY();
  MG_TRY(fetch_map(const_cast<mg_profile*>(p)));
  map[0] = p->map[0];
  map[1] = p->map[1];
  return MG_OK;
}

uint64_t mg_profile_ngroups(const mg_profile* p) {
  if (!p || !ctx().ready) return 0;
  if (fetch_map(const_cast<mg_profile*>(p)) != MG_OK) return 0;
  return p->ngroups;
}

static int commit_impl(mg_profile* p, int incoming_dropped, int first_shard, uint64_t group_base, uint64_t* d_count,
                       uint64_t* d_bases, uint64_t* d_first_seen, uint64_t* d_scalars, bool reset_acc) {
  MG_REQUIRE_READY();
  if (!p || !d_count || !d_bases || !d_first_seen || !d_scalars) return fail(MG_ERR_ARG, "null argument");
  if (p->committed) return fail(MG_ERR_STATE, "profile shard already committed");
  p->committed = true;
  if (p->nrecs == 0) {  // the phantom boundary never happens without a first line: nothing to add
    if (reset_acc) return mg_profile_acc_reset(d_count, d_bases, d_first_seen, d_scalars, p->ntax);
    return MG_OK;
  }
  // multimapped CSR buffers sized by their upper bounds (pooled), so that nothing has to be read back here:
  // entries <= records, multimapped reads <= records
  MG_TRY(p->mm_offsets.alloc((p->nrecs + 2) * sizeof(uint64_t)));
  MG_TRY(p->mm_tax.alloc((p->nrecs + 1) * sizeof(uint32_t)));
  MG_TRY(p->mm_hitlen.alloc((p->nrecs + 1) * sizeof(uint64_t)));
  MG_TRY(p->mm_read.alloc((p->nrecs + 1) * sizeof(uint64_t)));
  return launch_pass(p, true, incoming_dropped ? 1u : 0u, first_shard ? 1u : 0u, group_base, d_count, d_bases,
                     d_first_seen, d_scalars, reset_acc);
}

int mg_profile_commit_dev(mg_profile* p, int incoming_dropped, int first_shard, uint64_t group_base, uint64_t* d_count,
                          uint64_t* d_bases, uint64_t* d_first_seen, uint64_t* d_scalars) {
  return commit_impl(p, incoming_dropped, first_shard, group_base, d_count, d_bases, d_first_seen, d_scalars, false);
}

int mg_profile_commit_reset_dev(mg_profile* p, int incoming_dropped, int first_shard, uint64_t group_base,
                                uint64_t* d_count, uint64_t* d_bases, uint64_t* d_first_seen, uint64_t* d_scalars) {
  return commit_impl(p, incoming_dropped, first_shard, group_base, d_count, d_bases, d_first_seen, d_scalars, true);
}

// (host code: every rounding is the reference's — no fused multiply-add between the share and the sum it goes into)
#pragma clang fp contract(off)
// resolve_multi_prop's per-read split (scripts/map_and_profile.py:269-312) in the reference's ORDER OF ADDITIONS per taxon.  One read's
// shares — its distinct taxa that still have an entry, weight / sum of weights x hitlen [/ genome length] — depend on nothing but the
// read; what must keep its order is every taxon's running sum.  So, for a long list: host threads take contiguous runs of reads and
// sort their shares into one list per OWNER of the taxon (eight neighbouring taxa — a cache line of sums — have one owner); then every
// owner adds its lists up, run after run: each taxon's additions in read order, bit for bit the serial loop.
int mg_multimapped_shares(const uint64_t* mm_offsets, uint64_t nreads, const uint32_t* mm_tax, const uint64_t* mm_hitlen,
                          const double* weight, uint32_t ntax, const double* genome_len, double* extra, uint8_t* touched) {
  if (!extra || !touched || !weight || (nreads && (!mm_offsets || !mm_tax || !mm_hitlen)))
    return fail(MG_ERR_ARG, "null argument");
  for (uint32_t t = 0; t < ntax; ++t) { extra[t] = 0.0; touched[t] = 0; }
  if (nreads == 0) return MG_OK;
  const uint64_t nent = mm_offsets[nreads] - mm_offsets[0];
  const unsigned hw = std::thread::hardware_concurrency();
  unsigned nth = nent >= (1u << 16) && hw >= 8 ? 8u : 1u;  // (a power of two; two or four threads lose to the serial loop: the lists cost more than they save)
  if (const int64_t forced = dbg("shares_threads")) nth = forced >= 8 ? 8u : forced >= 4 ? 4u : forced >= 2 ? 2u : 1u;  // (tests, probes)
  struct Share { uint32_t t; double part; };
  std::vector<std::vector<Share>> lists((size_t)nth * nth);  // [run of reads][owner]
  std::atomic<uint32_t> bad_taxon{0xffffffffu};
  auto shares_of = [&](unsigned run, uint64_t r0, uint64_t r1) {
    std::vector<uint32_t> taxa;
    std::vector<Share>* mine = &lists[(size_t)run * nth];
    if (nth > 1)
      for (unsigned o = 0; o < nth; ++o) mine[o].reserve((size_t)((mm_offsets[r1] - mm_offsets[r0]) / nth + 1024));
    for (uint64_t i = r0; i < r1; ++i) {
      taxa.clear();
      for (uint64_t e = mm_offsets[i]; e < mm_offsets[i + 1]; ++e) {
        const uint32_t t = mm_tax[e];
        if (t >= ntax) { bad_taxon.store(t); return; }
        if (weight[t] == weight[t]) taxa.push_back(t);  // (not NaN: the taxon still has an entry)
      }
      if (taxa.empty()) continue;
      std::sort(taxa.begin(), taxa.end());
      taxa.erase(std::unique(taxa.begin(), taxa.end()), taxa.end());
      double denom = 0.0;
      for (uint32_t t : taxa) denom += weight[t];
      if (denom == 0.0) continue;
      const double hitlen = (double)mm_hitlen[i];
      for (uint32_t t : taxa) {
        double part = (weight[t] / denom) * hitlen;
        if (genome_len) part = part / genome_len[t];
        if (nth == 1) {
          extra[t] += part;
          touched[t] = 1;
        } else {
          mine[(t >> 3) & (nth - 1)].push_back(Share{t, part});
        }
      }
    }
  };
  if (nth == 1) {
    shares_of(0, 0, nreads);
    if (bad_taxon.load() != 0xffffffffu) return fail(MG_ERR_ARG, "multimapped taxon %u outside [0,%u)", bad_taxon.load(), ntax);
    return MG_OK;
  }
  {
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nth; ++k) th.emplace_back(shares_of, k, nreads * k / nth, nreads * (k + 1) / nth);
    for (auto& t : th) t.join();
  }
  if (bad_taxon.load() != 0xffffffffu) return fail(MG_ERR_ARG, "multimapped taxon %u outside [0,%u)", bad_taxon.load(), ntax);
  {
    auto add_up = [&](unsigned owner) {
      std::vector<uint8_t> seen(ntax, 0);  // (64 taxa to a cache line of marks: every owner its own, merged below)
      for (unsigned run = 0; run < nth; ++run)
        for (const Share& s : lists[(size_t)run * nth + owner]) {
          extra[s.t] += s.part;
          seen[s.t] = 1;
        }
      for (uint32_t t0 = owner * 8u; t0 < ntax; t0 += 8u * nth)
        for (uint32_t t = t0; t < t0 + 8u && t < ntax; ++t) touched[t] = seen[t];
    };
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nth; ++k) th.emplace_back(add_up, k);
    for (auto& t : th) t.join();
  }
  return MG_OK;
}

int mg_profile_resolve_multimapped_dev(const mg_profile* p, const double* d_weight, const double* d_genome_len,
                                       double* d_extra) {
  MG_REQUIRE_READY();
  if (!p || !d_weight || !d_extra) return fail(MG_ERR_ARG, "null argument");
  if (!p->committed) return fail(MG_ERR_STATE, "profile shard not committed");
  hipStream_t st = stage_c_stream();
  MG_HIP(hipMemsetAsync(d_extra, 0, (uint64_t)p->ntax * sizeof(double), st));
  if (p->nrecs == 0) return MG_OK;
  const uint32_t use_lds = p->ntax <= 2048 ? 1u : 0u;
  ProfScope ps("resolve_multimapped", st);
  hipLaunchKernelGGL(k_resolve_multimapped, dim3(grid_for(p->nrecs / 4 + 1, 256, (unsigned)ctx().num_cus * 4)), dim3(256),
                     use_lds ? (size_t)p->ntax * sizeof(double) : 0, st, p->mm_offsets.as<uint64_t>(),
                     p->mm_tax.as<uint32_t>(), p->mm_hitlen.as<uint64_t>(), p->tot.as<uint64_t>(), d_weight, d_genome_len,
                     p->ntax, use_lds, d_extra);
  MG_HIP(hipGetLastError());
  return MG_OK;
}

int mg_profile_multimapped_size(const mg_profile* p, uint64_t* nreads, uint64_t* nentries) {
  if (!p) return fail(MG_ERR_ARG, "null profile");
  MG_REQUIRE_READY();
  if (!p->committed) return fail(MG_ERR_STATE, "profile shard not committed");
  MG_TRY(fetch_mm(const_cast<mg_profile*>(p)));
  if (nreads) *nreads = p->mm_nreads;
  if (nentries) *nentries = p->mm_nentries;
  return MG_OK;
}

int mg_profile_multimapped(const mg_profile* p, uint64_t* mm_offsets, uint32_t* mm_tax, uint64_t* mm_hitlen,
                           uint64_t* mm_read) {
  MG_REQUIRE_READY();
  if (!p) return fail(MG_ERR_ARG, "null profile");
  if (!p->committed) return fail(MG_ERR_STATE, "profile shard not committed");
  if (p->nrecs == 0) { if (mm_offsets) mm_offsets[0] = 0; return MG_OK; }
  MG_TRY(fetch_mm(const_cast<mg_profile*>(p)));
  MG_TRY(mg_memcpy_d2h(mm_offsets, p->mm_offsets.p, (p->mm_nreads + 1) * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_d2h(mm_tax, p->mm_tax.p, p->mm_nentries * sizeof(uint32_t)));
  MG_TRY(mg_memcpy_d2h(mm_hitlen, p->mm_hitlen.p, p->mm_nreads * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_d2h(mm_read, p->mm_read.p, p->mm_nreads * sizeof(uint64_t)));
  return MG_OK;
}

void mg_profile_free(mg_profile* p) { delete p; }

int mg_profile_assign(const mg_aln_rec* recs, uint64_t nrecs, const uint32_t* ref2tax, uint32_t nref, uint32_t ntax,
                      double pct_id, uint64_t* out_count, uint64_t* out_bases, uint64_t* out_first_seen,
                      uint64_t* out_tot_rds, uint64_t* out_n_ambig, uint64_t* mm_offsets, uint32_t* mm_tax,
                      uint64_t* mm_hitlen, uint64_t* mm_read, uint64_t mm_cap_reads, uint64_t mm_cap_entries,
                      uint64_t* mm_nreads, uint64_t* mm_nentries) {
  MG_REQUIRE_READY();
  hipStream_t st = stage_c_stream();
  DevBuf d_recs, d_r2t, d_acc;
  MG_TRY(d_recs.alloc((nrecs + 1) * sizeof(mg_aln_rec)));
  MG_TRY(d_r2t.alloc((uint64_t)nref * sizeof(uint32_t)));
  MG_TRY(d_acc.alloc((3 * (uint64_t)ntax + 2) * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_h2d(d_recs.p, recs, nrecs * sizeof(mg_aln_rec)));
  MG_TRY(mg_memcpy_h2d(d_r2t.p, ref2tax, (uint64_t)nref * sizeof(uint32_t)));
  uint64_t* acc = d_acc.as<uint64_t>();
  MG_HIP(hipMemsetAsync(acc, 0, 2 * (uint64_t)ntax * sizeof(uint64_t), st));
  MG_HIP(hipMemsetAsync(acc + 2 * (uint64_t)ntax, 0xff, (uint64_t)ntax * sizeof(uint64_t), st));
  MG_HIP(hipMemsetAsync(acc + 3 * (uint64_t)ntax, 0, 2 * sizeof(uint64_t), st));
  mg_profile* p = nullptr;
  MG_TRY(mg_profile_begin_dev(d_recs.as<mg_aln_rec>(), nrecs, 0, d_r2t.as<uint32_t>(), nref, ntax, pct_id, &p));
  std::unique_ptr<mg_profile, void (*)(mg_profile*)> guard(p, mg_profile_free);
  MG_TRY(mg_profile_commit_dev(p, 1, 1, 0, acc, acc + ntax, acc + 2 * (uint64_t)ntax, acc + 3 * (uint64_t)ntax));
  MG_TRY(mg_memcpy_d2h(out_count, acc, (uint64_t)ntax * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_d2h(out_bases, acc + ntax, (uint64_t)ntax * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_d2h(out_first_seen, acc + 2 * (uint64_t)ntax, (uint64_t)ntax * sizeof(uint64_t)));
  uint64_t sc[2];
  MG_TRY(mg_memcpy_d2h(sc, acc + 3 * (uint64_t)ntax, sizeof(sc)));
  *out_tot_rds = sc[0];
  *out_n_ambig = sc[1];
  MG_TRY(fetch_mm(p));
  *mm_nreads = p->mm_nreads;
  *mm_nentries = p->mm_nentries;
  if (p->mm_nreads > mm_cap_reads || p->mm_nentries > mm_cap_entries)
    return fail(MG_ERR_CAPACITY, "multimapped output needs %llu reads / %llu entries", (unsigned long long)p->mm_nreads,
                (unsigned long long)p->mm_nentries);
  return mg_profile_multimapped(p, mm_offsets, mm_tax, mm_hitlen, mm_read);
}

}  // extern "C"
