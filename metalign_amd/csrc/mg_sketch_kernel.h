// mg_sketch_kernel.h — the one-k stage-A kernel (k_sketch_reads<K, HM>) and the genome-position hasher
// (k_hash_positions<K, HM>) as templates, shared by the two translation units that instantiate them for k = 1..64:
// mg_sketch.hip (hash definition 0, the default) and mg_sketch_cmash.hip (definition 1) — two compilations side by side
// instead of one of twice the length.  See mg_sketch.hip for the design.
#pragma once
#include "mg_internal.h"
#include "mg_kmer.h"
#include "mg_sketch_dev.h"

namespace mg {

// Wave-level candidate sink: LDS staging, then either one reservation in the flat list (list mode) or an
// insert-or-increment per candidate in the partitioned counting table (table mode, shift < 64).
struct CandSink {
  uint64_t* lds;      // this wave's kCandBuf entries
  uint64_t* out;      // list mode: candidate list
  uint64_t cap;       // list mode: entries available in `out`
  unsigned long long* counters;  // [0] candidates produced, [1] k-mers hashed, [2] table overflows
  Slot* tab;                     // table mode: [nbuckets][kBucketSlots] slots (key = hash + 1, 0 = empty; counter)
  unsigned shift;                // bucket = hash >> shift; 64 = list mode (read sketches start at hash 0)
  const uint32_t* fbits;         // optional membership pre-filter (mg_filter): bit (h & fmask) set <=> h may be in the table
  uint64_t fmask;
  uint32_t cs;        // table mode: counters saturate at cs (0 = exact)
  int n;              // entries staged (wave-uniform)
  unsigned long long produced = 0;  // table mode: this lane's candidates inserted so far
  bool slot_first = false;          // table mode: the order of the two look-ups of a flush (wave-uniform)
  uint32_t order = 0;               // 0: adapt; 1 / 2: pinned (tests)
  uint32_t ablate = 0;              // resident index: diagnostics (see flush)
  uint32_t epoch = 0;               // table mode, != 0: `tab` is a resident index (mg_sketch_dev.h) and out / cap the list of hashes touched
  uint32_t* lbase = nullptr;        // ... this wavefront's chunk of that list (slot numbers: `out`, there, is a uint32_t array)
  uint32_t lfill = 0;

  __device__ __forceinline__ bool passes(uint64_t h) const {
    return !fbits || ((fbits[(h & fmask) >> 5] >> (h & 31u)) & 1u);
  }

  // The filter is probed here, a buffer at a time with every lane busy, not in the hashing loop.
  __device__ __forceinline__ void flush(int lane) {
    if (n == 0) return;
    wave_lds_sync();
    if (shift >= 64) {
      for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const uint64_t h = i < n ? lds[i] : 0;
        const bool keep = i < n && passes(h);
        const unsigned long long m = __ballot(keep);
        if (m == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(counters, (unsigned long long)__popcll(m));
        base = __shfl(base, 0, 64) + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && base < cap) out[base] = h;
      }
    } else if (epoch) {
      // resident index (mg_sketch_dev.h).  The one-k kernel serves the k sets without a fused kernel and the rare sketch that
      // is made again: its flush takes the buffer 64 candidates at a time, each lane's candidate round after round until it is
      // found or known absent — four to a lane in one round trip, as the fused kernel does it, costs every one-k kernel 35
      // registers (100 -> 135: three wavefronts per SIMD instead of four) whether it ever sees an index or not.
      // (diagnostics, tools/k1_dense_ablation.sh, knob resident_ablate: 1 = the candidates are dropped, 2 = looked up but not counted
      // and not listed, 3 = counted but not listed)
      if (ablate == 1u) { wave_lds_sync(); n = 0; return; }
#pragma unroll 1
      for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        uint64_t hh[1] = {i < n ? lds[i] : kReservedHash};
        uint32_t hop[1] = {0}, pos[1];
        Slot* bucket[1] = {tab + (hh[0] != kReservedHash ? hh[0] >> shift : 0ull) * kBucketSlots};
        bool hit[1], fresh[1], on[1];
        for (;;) {
          resident_lookup<1>(bucket, hh, hop, epoch, cs, hit, fresh, on, pos, ablate != 2u);
          produced += hit[0];
          if (out && ablate < 2u)
            resident_list_append(fresh[0], (uint32_t)(hh[0] >> shift) * kBucketSlots + pos[0], reinterpret_cast<uint32_t*>(out), cap,
                                 counters, lbase, lfill, lane);
          bool more = false;
          if (on[0] && hop[0] < kMaxHops) { ++hop[0]; more = true; }
          else hh[0] = kReservedHash;
          if (__ballot(more) == 0ull) break;
        }
      }
    } else {
      uint32_t lost = 0, kept = 0;
      // this lane's candidates; then — all in flight together — either their filter words and after those the home
      // slots of the survivors, or the home slots and after those the filter words of the candidates their slot does
      // not hold (same result either way: what is in the table has passed the filter; see MultiSink::flush in
      // mg_sketch_multi.hip for which order a wavefront takes).  A home slot is key and counter in ONE 16-byte access
      // (one candidate after the other — key, then counter, then the next candidate — a flush was nine dependent round
      // trips to memory; now it is two, and a tenth of the kernel's time went with them).
      constexpr int J = kCandBuf / 64;
      uint64_t hh[J];
      uint32_t fw[J];
      uint4 sv[J];
      bool go[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int i = lane + 64 * j;
        hh[j] = i < n ? lds[i] : kReservedHash;
        fw[j] = 0xffffffffu;
        sv[j] = make_uint4(0, 0, 0, 0);
      }
      if (slot_first) {
#pragma unroll
        for (int j = 0; j < J; ++j)
          if (hh[j] != kReservedHash) sv[j] = *reinterpret_cast<const uint4*>(tab + (hh[j] >> shift) * kBucketSlots + ((uint32_t)hh[j] & (kBucketSlots - 1)));
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const unsigned long long key = (unsigned long long)sv[j].x | ((unsigned long long)sv[j].y << 32);
          if (fbits && hh[j] != kReservedHash && key != hh[j] + 1) fw[j] = fbits[(hh[j] & fmask) >> 5];
        }
#pragma unroll
        for (int j = 0; j < J; ++j) go[j] = hh[j] != kReservedHash && ((fw[j] >> (hh[j] & 31u)) & 1u);
      } else {
#pragma unroll
        for (int j = 0; j < J; ++j)
          if (fbits && hh[j] != kReservedHash) fw[j] = fbits[(hh[j] & fmask) >> 5];
#pragma unroll
        for (int j = 0; j < J; ++j) {
          go[j] = hh[j] != kReservedHash && ((fw[j] >> (hh[j] & 31u)) & 1u);
          if (go[j]) sv[j] = *reinterpret_cast<const uint4*>(tab + (hh[j] >> shift) * kBucketSlots + ((uint32_t)hh[j] & (kBucketSlots - 1)));
        }
      }
      int found = 0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const unsigned long long v = hh[j] + 1, key = (unsigned long long)sv[j].x | ((unsigned long long)sv[j].y << 32);
        found += __popcll(__ballot(go[j] && key == v));
        if (!go[j]) continue;
        ++kept;
        if (key == v) {  // the usual case at metagenomic coverage: a repeat
          if (!(cs && sv[j].z >= cs))
            atomicAdd(&tab[(hh[j] >> shift) * kBucketSlots + ((uint32_t)hh[j] & (kBucketSlots - 1))].cnt, 1u);
        } else if (!table_add(tab, hh[j] >> shift, hh[j], 1u, cs, key == 0ull ? 0u : 1u)) {  // empty: claim it; taken: probe on
          ++lost;
        }
      }
      slot_first = order ? order == 2u : 2 * found > n;
      // (counted per lane and added to counters[0] once, at the end of the kernel: an atomic per flush on that one
      // address is what bounded the kernel when the threshold filters little — 1.3 M flushes per 10 M reads at ~24 ns)
      produced += kept;
      if (lost) atomicAdd(counters + 2, (unsigned long long)lost);
    }
    wave_lds_sync();
    n = 0;
  }

  __device__ __forceinline__ void offer(bool hit, uint64_t h, int lane) {
    // (the builtin takes the condition mask as it is; __ballot() first materialises the predicate as an integer)
    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    if (m == 0) return;
    if (hit) lds[n + __popcll(m & ((1ull << lane) - 1ull))] = h;
    n += __popcll(m);
    if (n > kCandBuf - 64) flush(lane);
  }

  // hit = a && b with each condition balloted by its own compare (the ballot of a conjunction is lowered through a
  // materialised integer: two more VALU instructions per base)
  __device__ __forceinline__ void offer2(bool a, bool b, uint64_t h, int lane) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(a) & __builtin_amdgcn_ballot_w64(b);
    if (m == 0) return;
    if (a && b) lds[n + __popcll(m & ((1ull << lane) - 1ull))] = h;
    n += __popcll(m);
    if (n > kCandBuf - 64) flush(lane);
  }
};

// One lane walks its read two bases per iteration.  Straight-line body (invalid bases and
// positions past the end are folded into the run counter instead of branches) so that the two
// independent MurmurHash3 chains of an iteration interleave in the VALU.  CODES: src is the wavefront's
// nibble-packed LDS stage and `start` the nibble index of this lane's read; otherwise src points at the read's
// ASCII bases in HBM.
// MODE 0: any tile.  1: a clean tile (no invalid base in it) of equally long reads.  2: a clean tile of ragged reads.
template <int K, bool CODES, int MODE = 0, int HM = kHashCanonical>
__device__ __forceinline__ void walk_reads(const uint8_t* src, uint32_t start, uint32_t len, uint32_t maxlen_v,
                                           uint64_t hmax, CandSink& sink, uint64_t& kmers, int lane, const uint64_t* htab) {
  Roller<K> roll;
  roll.reset();
  uint32_t nk = 0;
  CodeStream cs;
  if constexpr (CODES) cs.open(src, start);
  // the tile's longest read, as a SCALAR: the position counter, the stream's group changes and nibble offsets and
  // the "is this k-mer complete" tests of the clean walk then live in SGPRs instead of costing VALU issue slots
  const uint32_t maxlen = __builtin_amdgcn_readfirstlane(maxlen_v);
  // The first K-1 bases of a read complete no k-mer: roll them in without hashing (every lane starts its read
  // at pos 0, so this is wave-uniform; it is (K-1)/150 of all steps — 13 % at k = 21, 39 % at k = 60).
  constexpr uint32_t kWarm = (uint32_t)(K - 1) & ~1u;
  const uint32_t warm = kWarm < maxlen ? kWarm : (maxlen & ~1u);
  if constexpr (MODE != 0) {
    // No invalid base anywhere in the tile: no run counter — "this position completes a k-mer" is a scalar condition,
    // the k-mers of a read are counted in closed form.  Equally long reads (MODE 1, the usual tile) need no length
    // test either; ragged reads (MODE 2: trimmed data) pay one compare per base, and a lane past its read's end
    // hashes the next read's bases and offers nothing.
    static_assert(CODES, "the clean walks read the LDS stage");
    constexpr bool RAGGED = MODE == 2;
    for (uint32_t pos = 0; pos < warm; ++pos) roll.push_clean(cs.at(pos) & 3u);
    for (uint32_t pos = warm; pos < maxlen; pos += 2) {
      const uint32_t c0 = cs.at(pos) & 3u, c1 = cs.at(pos + 1) & 3u;  // (c1 past the end: hashed, never offered)
      roll.push_clean(c0);
      const uint64_t h0 = roll.template hash<HM>(htab);
      roll.push_clean(c1);
      const uint64_t h1 = roll.template hash<HM>(htab);
      if (pos + 1 >= (uint32_t)K) {  // (scalar branches: the ballot then is the compare itself)
        if constexpr (RAGGED) sink.offer2(pos < len, h0 <= hmax, h0, lane); else sink.offer(h0 <= hmax, h0, lane);
      }
      if (pos + 2 >= (uint32_t)K && pos + 1 < maxlen) {
        if constexpr (RAGGED) sink.offer2(pos + 1 < len, h1 <= hmax, h1, lane); else sink.offer(h1 <= hmax, h1, lane);
      }
    }
    kmers += len >= (uint32_t)K ? len - (uint32_t)K + 1u : 0u;
    return;
  }
  auto code_at = [&](uint32_t pos) -> uint32_t {
    if constexpr (CODES) {
      const uint32_t c = cs.at(pos);
      return pos < len ? c : 4u;
    } else {
      if (pos >= len) return 4u;
      return encode1(src[pos]);
    }
  };
  for (uint32_t pos = 0; pos < warm; ++pos) {
    const uint32_t c = code_at(pos);
    roll.push(c);
    roll.run = c < 4u ? roll.run : 0;
  }
  for (uint32_t pos = warm; pos < maxlen; pos += 2) {
    const uint32_t c0 = code_at(pos), c1 = code_at(pos + 1);
    roll.push(c0);
    roll.run = c0 < 4u ? roll.run : 0;
    const uint64_t h0 = roll.template hash<HM>(htab);
    const bool full0 = roll.run >= K;
    roll.push(c1);
    roll.run = c1 < 4u ? roll.run : 0;
    const uint64_t h1 = roll.template hash<HM>(htab);
    const bool full1 = roll.run >= K;
    nk += (full0 ? 1u : 0u) + (full1 ? 1u : 0u);
    sink.offer(full0 && h0 <= hmax, h0, lane);
    sink.offer(full1 && h1 <= hmax, h1, lane);
  }
  kmers += nk;
}

// (A/B builds: -DMG_K1_WAVES_BIG=5 holds the kernels of k > 32 — two 64-bit words per packed strand, 112 VGPRs at k = 51, four
// wavefronts per SIMD — to the register budget of five)
#ifndef MG_K1_WAVES_BIG
#define MG_K1_WAVES_BIG 0
#endif
#if MG_K1_WAVES_BIG
#define MG_K1_ATTR(K) __attribute__((amdgpu_waves_per_eu((K) > 32 ? MG_K1_WAVES_BIG : 4)))
#else
#define MG_K1_ATTR(K)
#endif
// counters[0] = candidates produced (may exceed cap: overflow => caller retries), counters[1] = k-mers hashed
template <int K, int HM>
__global__ __launch_bounds__(kBlock) MG_K1_ATTR(K) void k_sketch_reads(const uint8_t* __restrict__ bases,
                                                         const uint64_t* __restrict__ offsets, uint64_t nreads,
                                                         uint64_t hmax, uint64_t* __restrict__ cand, uint64_t cand_cap,
                                                         unsigned long long* __restrict__ counters,
                                                         Slot* __restrict__ tab, unsigned bucket_shift,
                                                         unsigned stage_bytes, const uint32_t* __restrict__ fbits,
                                                         uint64_t fmask, uint32_t cs) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint64_t* htab = fill_hash_tables();  // MurmurHash3's first multiplies (mg_kmer.h)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t* stage = smem + (size_t)wave * stage_bytes;
  uint64_t* cbuf = reinterpret_cast<uint64_t*>(smem + (size_t)kWavesPerBlock * stage_bytes) + wave * kCandBuf;
  CandSink sink{cbuf, cand, cand_cap, counters, tab, bucket_shift, fbits, fmask, cs & kCsMask, 0};
  sink.order = cs >> 30;
  sink.ablate = (cs >> 28) & 3u;
  sink.slot_first = sink.order == 2u;
  if (!fbits) sink.epoch = (uint32_t)fmask;  // no filter words and a "mask": the table is a resident index, this its epoch
  uint64_t kmers = 0;
  const uint64_t ntiles = (nreads + 63) / 64;
  for (uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave; tile < ntiles;
       tile += (uint64_t)gridDim.x * kWavesPerBlock) {
    const uint64_t r0 = tile * 64;
    const uint64_t r = r0 + lane;
    uint64_t beg = 0, end = 0;
    if (r < nreads) { beg = offsets[r]; end = offsets[r + 1]; }
    const uint64_t len = end - beg;  // < 2^32: a read longer than that is rejected on the host side
    const uint64_t maxlen = wave_max_u64(len);
    const uint64_t t_beg = __shfl(beg, 0, 64);
    const uint64_t t_end = wave_max_u64(end);
    const uintptr_t a_first = reinterpret_cast<uintptr_t>(bases) + t_beg;
    const uintptr_t a0 = a_first & ~(uintptr_t)15;
    const uint64_t shift = a_first - a0;
    const uint64_t nbytes = shift + (t_end - t_beg);
    if (nbytes <= 2ull * stage_bytes) {
      // coalesced HBM -> LDS copy of the whole tile (16 B per lane per step), bases -> 4-bit codes on the way:
      // half a byte per base, 30 KB per workgroup instead of 52 — what lets a second grid of this kernel (the next
      // batch's) and the small kernels of the other streams live beside it on the CU
      const uint4* g = reinterpret_cast<const uint4*>(a0);
      uint2* s = reinterpret_cast<uint2*>(stage);
      uint32_t bad = 0;  // an invalid base anywhere in what this lane staged (the 16-byte slop of the neighbours included)
      for (uint64_t i = lane; i * 16 < nbytes; i += 64) {
        const uint4 v = g[i];
        const uint2 p{pack4(encode4(v.x)) | (pack4(encode4(v.y)) << 16), pack4(encode4(v.z)) | (pack4(encode4(v.w)) << 16)};
        bad |= (p.x | p.y) & 0x44444444u;
        s[i] = p;
      }
      wave_lds_sync();
      const uint32_t nstart = (uint32_t)(shift + (beg - t_beg));
      if (__ballot(bad != 0) != 0ull)
        walk_reads<K, true, 0, HM>(stage, nstart, (uint32_t)len, (uint32_t)maxlen, hmax, sink, kmers, lane, htab);
      else if (__ballot(len != maxlen) == 0ull)
        walk_reads<K, true, 1, HM>(stage, nstart, (uint32_t)len, (uint32_t)maxlen, hmax, sink, kmers, lane, htab);
      else
        walk_reads<K, true, 2, HM>(stage, nstart, (uint32_t)len, (uint32_t)maxlen, hmax, sink, kmers, lane, htab);
      wave_lds_sync();
    } else {
      walk_reads<K, false, 0, HM>(bases + beg, 0u, (uint32_t)len, (uint32_t)maxlen, hmax, sink, kmers, lane, htab);
    }
  }
  sink.flush(lane);
  kmers = wave_sum_u64(kmers);
  if (lane == 0 && kmers) atomicAdd(counters + 1, (unsigned long long)kmers);
  const uint64_t produced = wave_sum_u64(sink.produced);
  if (lane == 0 && produced) atomicAdd(counters, (unsigned long long)produced);
}

// Stage A': hash of the k-mer ENDING at every base position of a batch of genomes
// (kReservedHash where there is none).  One lane per run of kChunk positions.
constexpr int kChunk = 64;

template <int K, int HM>
__global__ __launch_bounds__(256) void k_hash_positions(const uint8_t* __restrict__ bases,
                                                        const uint64_t* __restrict__ offsets, uint64_t nseq,
                                                        uint64_t nbases, uint64_t* __restrict__ out) {
  const uint64_t* htab = fill_hash_tables();
  const uint64_t nchunks = (nbases + kChunk - 1) / kChunk;
  for (uint64_t ch = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks;
       ch += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p0 = ch * kChunk;
    const uint64_t p1 = p0 + kChunk < nbases ? p0 + kChunk : nbases;
    // sequence containing p0: last g with offsets[g] <= p0 (empty sequences are skipped by the walk below)
    uint64_t lo = 0, hi = nseq;  // invariant: offsets[lo] <= p0 < offsets[hi]
    while (hi - lo > 1) {
      uint64_t mid = (lo + hi) >> 1;
      if (offsets[mid] <= p0) lo = mid; else hi = mid;
    }
    uint64_t g = lo;
    uint64_t g_beg = offsets[g], g_end = offsets[g + 1];
    Roller<K> roll;
    roll.reset();
    uint64_t p = p0 >= (uint64_t)(K - 1) ? p0 - (K - 1) : 0;
    if (p < g_beg) p = g_beg;  // warm-up never crosses into the previous sequence
    for (; p < p1; ++p) {
      while (p >= g_end) {  // entered the next sequence
        ++g;
        g_beg = g_end;
        g_end = offsets[g + 1];
        roll.run = 0;
      }
      uint32_t c;
      uint64_t h = kReservedHash;
      if (decode_base(bases[p], c)) {
        roll.push(c);
        if (roll.full()) h = roll.template hash<HM>(htab);
      } else {
        roll.run = 0;
      }
      if (p >= p0) out[p] = h;
    }
  }
}

}  // namespace mg
