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

// The kernels' `cs` word: the saturation value in the low 28 bits; the top two pin the order of a flush's two look-ups
// for the tests (0: each wavefront adapts, 1: filter first, 2: home slot first: knob flush_order); bits 28-29 ablate the look-ups in a
// resident index for diagnostics (knob resident_ablate; tools/k1_dense_ablation.sh).
constexpr uint32_t kCsMask = 0x0fffffffu;
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

// RESIDENT INDEX (mg_filter_make_resident): the same slots, but seeded ONCE with every hash of the genome table and never
// cleared.  A candidate is then either found and counted, or it is not a hash of the table and dropped: no filter word,
// no insert.  What a flush costs is (a) its DEPENDENT round trips to memory and (b) the number of loads its lanes issue,
// each to a line of its own — not bytes (tools/k1_probe.py, 12.5M reads against the 200k-genome table: no look-up at all
// 19.3 ms, every home slot loaded and nothing else 24.8, two slots of one line per candidate 33.0, the counting table's
// slot -> filter word -> compare-and-swap -> add 35.2).  So the index answers with ONE 16-byte load, once:
//  * open addressing as in the counting table (home slot = the hash's low bits, then the slots after it), but the set is
//    static, so the HOME slot knows whether any hash whose home it is lives further on (kMovedOn, set at seeding).  A
//    candidate that is not in its home slot and finds the mark clear is not a hash of the table — 90 % of the misses end
//    there, where probing would go on until an empty slot.  Otherwise it goes back into the wavefront's buffer tagged "one
//    slot on" and is looked at again by the NEXT flush (found / an empty slot / on again): never a second round trip.
//  * `pad` holds the EPOCH the counter belongs to: a sketch call has its own, larger than any before it; a counter of
//    another epoch reads as zero.  Counting never waits for memory either: {cnt, pad} is one 64-bit word, epoch high —
//    atomicMax(word, epoch << 32 | mark) starts the epoch's count at zero unless it has started, atomicAdd(word, 1) counts;
//    both without a return value, same address, program order.  A lane that saw a stale word (another XCD's L2) does the
//    same two and nothing is lost; one that saw the counter at cs does neither.
constexpr uint32_t kMovedOn = 0x80000000u;  // in cnt of a hash's home slot: a hash with this home is not in it
constexpr uint32_t kMaxHops = 63;           // slots a hash may live from home (the fused kernel's tag has six bits)

// J candidates of a lane (hh[j] == kReservedHash: none), hop[j] slots from home.  hit[j]: found and counted; fresh[j]: ... and
// this lane saw no count of this epoch there (for a list of the hashes a pass has touched: several lanes may, for one hash);
// on[j]: not found, and it may be one slot on.
template <int J>
__device__ __forceinline__ void resident_lookup(Slot* const (&bucket)[J], const uint64_t (&hh)[J], const uint32_t (&hop)[J],
                                                uint32_t epoch, uint32_t cs, bool (&hit)[J], bool (&fresh)[J], bool (&on)[J],
                                                uint32_t (&pos)[J], bool count = true) {
  uint4 raw[J];
  Slot* at[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    raw[j] = make_uint4(0, 0, 0, 0);
    pos[j] = ((uint32_t)hh[j] + hop[j]) & (kBucketSlots - 1);  // (the slot within its bucket: what a fresh hit is listed by)
    at[j] = bucket[j] + pos[j];
    if (hh[j] != kReservedHash) raw[j] = *reinterpret_cast<const uint4*>(at[j]);
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const unsigned long long key = (unsigned long long)raw[j].x | ((unsigned long long)raw[j].y << 32);
    const uint32_t mark = raw[j].z & kMovedOn, cnt = raw[j].z & ~kMovedOn, ep = raw[j].w;
    hit[j] = hh[j] != kReservedHash && key == hh[j] + 1;
    on[j] = hh[j] != kReservedHash && !hit[j] && (hop[j] == 0 ? mark != 0 : key != 0ull);
    fresh[j] = hit[j] && ep != epoch;
    if (count && hit[j] && !(ep == epoch && cs && cnt >= cs)) {
      unsigned long long* word = reinterpret_cast<unsigned long long*>(&at[j]->cnt);  // {cnt, pad}: cnt is the low half
      if (ep != epoch) atomicMax(word, ((unsigned long long)epoch << 32) | mark);
      atomicAdd(word, 1ull);
    }
  }
}

// What a pass has touched in a resident index, as a LIST (what turns the index into the pass's sketch without a walk over
// all of its slots) — of SLOT NUMBERS, bucket x 256 + slot: 32-bit keys sort in half the passes and a quarter of the bytes
// of 64-bit hashes, buckets are hash ranges, and the handful of slots a pass touches in one bucket are put in hash order
// when the counters are read (mg_sketch.hip: k_list_*).  A wavefront appends what its lanes saw `fresh` to a chunk of its
// own (one atomic on the list's cursor, counters[3], per kListChunk entries); the list is filled with kNoSlot before the
// launch, so what a wavefront leaves of its last chunk sorts to the end.  A full list is reported like a table overflow
// (counters[2]): whoever resolves the sketch makes it again with room for every hash.
constexpr uint32_t kNoSlot = 0xffffffffu;
constexpr uint32_t kListChunk = 256;
__device__ __forceinline__ void resident_list_append(bool mine, uint32_t h, uint32_t* __restrict__ list, uint64_t cap,
                                                     unsigned long long* __restrict__ counters, uint32_t*& lbase, uint32_t& lfill,
                                                     int lane) {
  const unsigned long long m = __ballot(mine);
  if (m == 0) return;
  const uint32_t c = (uint32_t)__popcll(m);
  if (!lbase || lfill + c > kListChunk) {
    unsigned long long off = 0;
    if (lane == 0) off = atomicAdd(counters + 3, (unsigned long long)kListChunk);
    off = (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)off) |
          ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(off >> 32)) << 32);
    if (off + kListChunk > cap) {
      if (lane == 0) atomicAdd(counters + 2, 1ull);
      lbase = nullptr;
      return;
    }
    lbase = list + off;
    lfill = 0;
  }
  if (mine) lbase[lfill + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = h;
  lfill += c;
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
