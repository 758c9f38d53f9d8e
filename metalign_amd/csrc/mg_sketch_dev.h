// mg_sketch_dev.h — device-side pieces shared by the stage-A kernels (mg_sketch.hip: one k per launch;
// mg_sketch_multi.hip: every k of a query in one launch): wavefront helpers, the partitioned counting table's
// insert, the ASCII -> 4-bit code stage and a lane's view of it.
#pragma once
#include "mg_internal.h"
#include "mg_kmer.h"

namespace mg {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;
constexpr int kCandBuf = 256;  // u64 entries per wavefront

__device__ __forceinline__ void wave_lds_sync() {
  // LDS traffic of one wavefront is executed in order; this only pins the compiler.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint64_t t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// The kernels' `cs` word: the saturation value in the low 30 bits; the top two pin the order of a flush's two look-ups
// for the tests (0: each wavefront adapts, 1: filter first, 2: home slot first) — MG_DEBUG_FLUSH_ORDER=filter|slot.
constexpr uint32_t kCsMask = 0x3fffffffu;
constexpr uint32_t kBucketSlots = 256;  // open-addressed slots per hash-range bucket (power of two)
constexpr uint32_t kBucketTarget = 128;  // expected distinct hashes per bucket at most (load factor <= 1/2; the shift rounds it down by up to 2x)

// One slot of the partitioned counting table: key = hash + 1 (0 = empty) and its occurrence counter side by side, so
// that ONE 16-byte access answers both "is it this key" and "is its counter saturated" (round 1 kept keys and counters
// in two arrays: two dependent random accesses per candidate, each a 128-byte line out of HBM).
struct __attribute__((aligned(16))) Slot {
  unsigned long long key;
  uint32_t cnt;
  uint32_t pad;
};
static_assert(sizeof(Slot) == 16, "slots are loaded with one 16-byte access");

// Insert-or-add into the partitioned counting table: tab[bucket][slot].key holds hash+1 (0 = empty).
// Returns false when the bucket has no free slot.
// cs > 0: counters SATURATE at cs (kmc -cs3, scripts/select_db.py:50): a counter that is seen at cs or above is
// left alone (counters only grow, so a stale look can only cost an unnecessary add), and whoever reads the table
// afterwards takes min(counter, cs).  At 50x coverage nearly every candidate is a repeat of a key whose counter is
// saturated already: it costs one read and no memory-side read-modify-write.
__device__ __forceinline__ bool table_add(Slot* __restrict__ tab, uint64_t bucket, uint64_t h, uint32_t amount, uint32_t cs,
                                          uint32_t first_probe = 0) {
  const unsigned long long v = h + 1;  // hashes are <= 2^64-2, so v is never the empty marker 0
  const uint64_t base = bucket * kBucketSlots;
  uint32_t p = ((uint32_t)h + first_probe) & (kBucketSlots - 1);  // low bits: independent of the bucket id
  for (uint32_t t = first_probe; t < kBucketSlots; ++t) {
    // A plain look first: a slot's key never changes once set, so a (possibly stale, per-XCD cached) read can only
    // err towards "empty", and then the CAS decides.  At 50x coverage most candidates are repeats of a key that is
    // already there: they cost this read and at most one add instead of a returning CAS and an add.
    const uint4 raw = *reinterpret_cast<const uint4*>(tab + base + p);
    unsigned long long old = (unsigned long long)raw.x | ((unsigned long long)raw.y << 32);
    const bool seen = old == v;
    if (old == 0ull) old = atomicCAS(&tab[base + p].key, 0ull, v);
    if (old == 0ull || old == v) {
      if (cs && seen && raw.z >= cs) return true;
      atomicAdd(&tab[base + p].cnt, amount);
      return true;
    }
    p = (p + 1) & (kBucketSlots - 1);
  }
  return false;
}

// Four ASCII bases -> four code bytes: 0..3 = A C G T (either case), 4 = anything else.  SWAR on the dword, done
// once per base while the tile is copied into LDS (14 instructions per 4 bases instead of 9 per base in the walk).
__device__ __forceinline__ uint32_t encode4(uint32_t x) {
  const uint32_t u = x & 0xDFDFDFDFu;                       // upper case
  const uint32_t t = (x >> 1) & 0x03030303u;                // A:0 C:1 T:2 G:3
  const uint32_t c = t ^ ((t >> 1) & 0x01010101u);          // A:0 C:1 G:2 T:3
  const uint32_t d = __builtin_amdgcn_perm(0u, 0x54474341u, c) ^ u;  // "ACGT"[c] != the byte <=> not a base
  const uint32_t nz = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
  return c | (nz >> 5);
}

// Four code bytes -> four nibbles (bits 0..15).
__device__ __forceinline__ uint32_t pack4(uint32_t c) {
  const uint32_t p = (c | (c >> 4)) & 0x00ff00ffu;
  return (p | (p >> 8)) & 0xffffu;
}

// A lane's view of its read in the nibble-packed LDS stage: eight codes per 32-bit word.  The read starts at any
// nibble, so the window of eight codes is a funnel shift of two consecutive words (one v_alignbit per eight bases);
// the word after next is requested a group ahead.  Positions advance in lock step across the wavefront, which makes
// the group changes scalar branches and the nibble offsets scalar operands.
struct CodeStream {
  const uint32_t* d;   // word holding the read's first code
  uint32_t sh;         // bit offset of that code in the word
  uint32_t lo, hi, nxt, w;
  uint32_t g;          // group (pos / 8) that w holds
  __device__ __forceinline__ void open(const uint8_t* stage, uint32_t start) {
    d = reinterpret_cast<const uint32_t*>(stage) + (start >> 3);
    sh = (start & 7u) * 4u;
    lo = d[0]; hi = d[1]; nxt = d[2];
    w = __builtin_amdgcn_alignbit(hi, lo, sh);
    g = 0;
  }
  // pos: wave-uniform; called for 0, 1, 2, ... or (from an even position on) for pairs pos, pos + 1: a group starts
  // at a multiple of 8, which only the first of a pair can be — the test is then a scalar one
  __device__ __forceinline__ uint32_t at(uint32_t pos) {
    if ((pos & 7u) == 0 && pos != 0) {
      g = pos >> 3;
      lo = hi; hi = nxt;
      nxt = d[g + 2];  // may run past the tile into whatever follows in LDS: such positions are >= len and masked
      w = __builtin_amdgcn_alignbit(hi, lo, sh);
    }
    return (w >> ((pos & 7u) * 4u)) & 15u;
  }
};

// Code byte of one base read from HBM (the path for tiles that do not fit the LDS stage).
__device__ __forceinline__ uint32_t encode1(uint32_t b) {
  uint32_t c;
  return decode_base(b, c) ? c : 4u;
}

}  // namespace mg
