// mg_sketch.hip — Stage A (read sketch) and Stage A' (genome sketch table).
//
// K1 `k_sketch_reads<K>`: one lane per read, one wavefront per tile of 64
// consecutive reads.  The tile's bases (contiguous in the concatenated read
// buffer) are copied HBM -> LDS with 16-byte coalesced loads and packed to 4-bit
// codes on the way, then every lane rolls its own read out of LDS (mg_kmer.h,
// CodeStream below) and hashes one canonical k-mer per base; tiles without an
// invalid base take a walk whose position bookkeeping is scalar.  Hashes <= hmax are compacted per wavefront (ballot + popcount) into an
// LDS candidate buffer.  A flush inserts its candidates into a COUNTING HASH
// TABLE in HBM that is partitioned by hash range: bucket = leading bits of the
// hash, kBucketSlots open-addressed slots per bucket (atomicCAS to claim a slot,
// atomicAdd on its counter) — the GPU analogue of what KMC does on disk.  The
// table therefore holds each DISTINCT hash once, whatever the coverage, and
// because hashes are uniform every bucket holds ~the same number of distinct
// hashes.  `k_bucket_sort` (one wavefront per bucket) compacts a bucket into
// LDS, sorts its (hash,count) pairs with a bitonic network and writes them
// out; buckets are hash ranges, so their concatenation (`k_bucket_scan` +
// `k_bucket_compact`) is the ascending sketch.  No global radix sort: ~0.05 ms
// instead of 8 onesweep passes (~0.3 ms) at 2.6 M candidates.  Inputs whose
// distinct count defeats the table sizing, and tiny candidate sets, take the
// list path: flat candidate list + rocPRIM sort / run-length (mg_sort.hip).
//
// Replaces: kmc -k60 -ci2 -cs3 (scripts/select_db.py:50-52) + k-mer hashing in
// CMash's streaming query (scripts/select_db.py:73-76).
#include <cstdlib>
#include <memory>

#include "mg_internal.h"
#include "mg_kmer.h"
#include "mg_sketch_dev.h"
#include "mg_sketch_kernel.h"

// (defined with the merge entry points below; sketch_resolve redoes a deferred merge with it)
static int redo_resident(mg_sketch* sk);
static int merge_via_sort(mg_sketch* sk, const uint64_t* d_hashes, const uint32_t* d_counts, uint64_t n, uint64_t s,
                          int any_truncated, uint64_t bound);

namespace mg {

// Per genome: first n distinct values of its sorted hash segment -> out[g*n ..], cnt[g].
__global__ __launch_bounds__(256) void k_take_bottom_n(const uint64_t* __restrict__ sorted,
                                                       const uint64_t* __restrict__ offsets, uint64_t nseq, uint64_t n,
                                                       uint64_t* __restrict__ out, uint32_t* __restrict__ cnt) {
  __shared__ uint32_t wave_tot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t g = blockIdx.x; g < nseq; g += gridDim.x) {
    const uint64_t beg = offsets[g], end = offsets[g + 1];
    uint64_t taken = 0;
    for (uint64_t base = beg; base < end && taken < n; base += 256) {
      const uint64_t i = base + threadIdx.x;
      uint64_t v = kReservedHash;
      bool head = false;
      if (i < end) {
        v = sorted[i];
        head = v != kReservedHash && (i == beg || sorted[i - 1] != v);
      }
      const unsigned long long m = __ballot(head);
      if (lane == 0) wave_tot[wave] = __popcll(m);
      __syncthreads();
      uint32_t before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (w < wave) before += wave_tot[w];
        total += wave_tot[w];
      }
      const uint64_t pos = taken + before + __popcll(m & ((1ull << lane) - 1ull));
      if (head && pos < n) out[g * n + pos] = v;
      taken += total;
      __syncthreads();
    }
    if (threadIdx.x == 0) cnt[g] = (uint32_t)(taken < n ? taken : n);
  }
}

static unsigned bit_length(uint64_t v) {
  unsigned b = 0;
  while (v) { ++b; v >>= 1; }
  return b ? b : 1;
}

// Merge step of the multi-GPU exchange: add (hash,count) pairs into the table; bucket = (hash - lo) >> shift.
__global__ void k_table_insert_pairs(const uint64_t* __restrict__ hashes, const uint32_t* __restrict__ counts, uint64_t n,
                                     uint64_t lo, unsigned shift, uint64_t nbuckets, Slot* __restrict__ tab,
                                     unsigned long long* __restrict__ counters, uint32_t cs) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t lost = 0;
  for (; i < n; i += stride) {
    const uint64_t h = hashes[i];
    uint64_t b = (h - lo) >> shift;
    if (h < lo || b >= nbuckets) { ++lost; continue; }  // outside the declared range: caller falls back
    if (!table_add(tab, b, h, counts[i], cs)) ++lost;
  }
  if (lost) atomicAdd(counters + 2, (unsigned long long)lost);
}

// One WAVEFRONT per bucket (no workgroup barriers: buckets are independent, four in flight per workgroup):
// compact the bucket's occupied slots into the wave's LDS region, sort the (hash,count) pairs by hash with a
// bitonic network (width = next power of two >= n, wave-synchronous steps) and write them back IN PLACE: the
// bucket's first nuniq[b] slots then hold its hashes (no longer hash + 1) ascending, with their counts.  (The whole
// bucket is in registers / LDS before the first store, and a bucket belongs to one wavefront; separate staging
// rows cost 12 B per slot, 13 GB for a dense 400 M-candidate table.)  Every hash of the bucket is distinct already.
__global__ __launch_bounds__(256) void k_bucket_sort(Slot* tab, uint64_t nbuckets, uint32_t* __restrict__ nuniq, uint32_t cs) {
  __shared__ uint64_t s_keys[4][kBucketSlots];
  __shared__ uint32_t s_cnt[4][kBucketSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t* keys = s_keys[wave];
  uint32_t* cnt = s_cnt[wave];
  const uint64_t gwave = (uint64_t)blockIdx.x * 4 + wave, nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t b = gwave; b < nbuckets; b += nwaves) {
    const uint64_t base = b * kBucketSlots;
    uint32_t n = 0;
    // all of the bucket's slots are requested before the first is looked at (one round trip, not one per chunk)
    uint64_t v[kBucketSlots / 64];
    uint32_t vc[kBucketSlots / 64];
#pragma unroll
    for (uint32_t c = 0; c < kBucketSlots / 64; ++c) {
      const uint4 raw = *reinterpret_cast<const uint4*>(tab + base + c * 64 + lane);
      v[c] = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
      vc[c] = raw.z;
    }
#pragma unroll
    for (uint32_t c = 0; c < kBucketSlots / 64; ++c) {
      const unsigned long long m = __ballot(v[c] != 0);
      if (v[c] != 0) {
        const uint32_t d = n + __popcll(m & ((1ull << lane) - 1ull));
        keys[d] = v[c] - 1;
        cnt[d] = (cs && vc[c] > cs) ? cs : vc[c];  // saturating counters: concurrent adds may have overshot
      }
      n += __popcll(m);
    }
    if (lane == 0) nuniq[b] = n;
    if (n == 0) continue;
    uint32_t N = 64;
    while (N < n) N <<= 1;
    for (uint32_t i = n + lane; i < N; i += 64) keys[i] = kReservedHash;
    wave_lds_sync();
    for (uint32_t k = 2; k <= N; k <<= 1) {
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t t = lane; t < N / 2; t += 64) {
          const uint32_t ix = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const uint32_t px = ix | j;
          const uint64_t x = keys[ix], y = keys[px];
          const bool up = (ix & k) == 0;
          if ((x > y) == up) {
            keys[ix] = y; keys[px] = x;
            const uint32_t cx = cnt[ix], cy = cnt[px];
            cnt[ix] = cy; cnt[px] = cx;
          }
        }
        wave_lds_sync();
      }
    }
    for (uint32_t i = lane; i < n; i += 64) { tab[base + i].key = keys[i]; tab[base + i].cnt = cnt[i]; }
    wave_lds_sync();
  }
}

// One block: exclusive sums of nuniq -> offs[0..nbuckets]; meta[0] = total distinct hashes.  Chunks of 4096
// counts go through LDS so that both the loads and the stores are coalesced (a thread that walks its own run of
// counts in HBM pays one round trip per count: 20 us for 14 k buckets, against 4 us this way).
__global__ __launch_bounds__(1024) void k_bucket_scan(const uint32_t* __restrict__ nuniq, uint64_t nbuckets,
                                                      uint64_t* __restrict__ offs, uint64_t* __restrict__ meta) {
  constexpr int kPer = 4, kChunk = 1024 * kPer;
  __shared__ uint32_t s_in[kChunk];
  __shared__ uint64_t s_out[kChunk];
  __shared__ uint64_t wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint64_t carry = 0;
  for (uint64_t base = 0; base < nbuckets; base += kChunk) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const uint64_t i = base + tid + j * 1024;
      s_in[tid + j * 1024] = i < nbuckets ? nuniq[i] : 0u;
    }
    __syncthreads();
    uint32_t v[kPer];
    uint64_t mine = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) { v[j] = s_in[tid * kPer + j]; mine += v[j]; }
    uint64_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t prev = __shfl_up(inc, o, 64);
      if (lane >= o) inc += prev;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint64_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wave) before += wave_tot[w];
      total += wave_tot[w];
    }
    uint64_t run = carry + before + inc - mine;
#pragma unroll
    for (int j = 0; j < kPer; ++j) { s_out[tid * kPer + j] = run; run += v[j]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const uint64_t i = base + tid + j * 1024;
      if (i < nbuckets) offs[i] = s_out[tid + j * 1024];
    }
    carry += total;
    __syncthreads();
  }
  if (tid == 0) { offs[nbuckets] = carry; meta[0] = carry; }
}

// Pack every bucket's distinct hashes / counts at its offset: the concatenation is ascending.
__global__ __launch_bounds__(256) void k_bucket_compact(const Slot* __restrict__ tab,
                                                        const uint32_t* __restrict__ nuniq, const uint64_t* __restrict__ offs,
                                                        uint64_t nbuckets, uint64_t* __restrict__ out_hashes,
                                                        uint32_t* __restrict__ out_counts, uint64_t out_cap) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t b = wave; b < nbuckets; b += nwaves) {
    const uint32_t n = nuniq[b];
    const uint64_t o = offs[b];
    for (uint32_t i = lane; i < n && o + i < out_cap; i += 64) {  // out_cap: see the size check on the host
      const uint4 raw = *reinterpret_cast<const uint4*>(tab + b * kBucketSlots + i);
      out_hashes[o + i] = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
      out_counts[o + i] = raw.z;
    }
  }
}

// ---- resident index (mg_filter_make_resident) ----
// Seed: every hash gets its slot (key = hash + 1; counter 0, epoch 0).  counters[0] += distinct hashes placed,
// counters[1] += hashes without a slot (outside the range, or a full bucket: the index is then not built).
__global__ void k_index_seed(const uint64_t* __restrict__ hashes, uint64_t n, unsigned shift, uint64_t nbuckets,
                             Slot* __restrict__ tab, unsigned long long* __restrict__ counters) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t placed = 0, lost = 0;
  for (; i < n; i += stride) {
    const uint64_t h = hashes[i];
    const uint64_t b = h >> shift;
    if (b >= nbuckets || h == kReservedHash) { ++lost; continue; }
    // its home slot or the first free one after it; a hash that is not at home leaves kMovedOn there (mg_sketch_dev.h)
    const unsigned long long v = h + 1;
    const uint64_t base = b * kBucketSlots;
    const uint32_t home = (uint32_t)h & (kBucketSlots - 1);
    uint32_t t = 0;
    for (; t <= kMaxHops; ++t) {
      const unsigned long long old = atomicCAS(&tab[base + ((home + t) & (kBucketSlots - 1))].key, 0ull, v);
      if (old == 0ull) { ++placed; break; }
      if (old == v) break;  // the same hash in another genome's sketch
    }
    if (t > kMaxHops) ++lost;
    else if (t > 0) atomicOr(&tab[base + home].cnt, kMovedOn);
  }
  if (placed) atomicAdd(counters, (unsigned long long)placed);
  if (lost) atomicAdd(counters + 1, (unsigned long long)lost);
}

// A slot of a resident index counts in this pass: seeded, written in this epoch.
__device__ __forceinline__ bool resident_live(const uint4& raw, uint32_t epoch) {
  return (raw.x | raw.y) != 0u && raw.w == epoch && (raw.z & ~kMovedOn) != 0u;
}

// nuniq[b] = slots of bucket b that count in this pass (one wavefront per bucket).
__global__ __launch_bounds__(256) void k_resident_count(const Slot* __restrict__ tab, uint64_t nbuckets, uint32_t epoch,
                                                        uint32_t* __restrict__ nuniq) {
  const int lane = threadIdx.x & 63;
  const uint64_t gwave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t b = gwave; b < nbuckets; b += nwaves) {
    uint4 raw[kBucketSlots / 64];
#pragma unroll
    for (uint32_t c = 0; c < kBucketSlots / 64; ++c) raw[c] = *reinterpret_cast<const uint4*>(tab + b * kBucketSlots + c * 64 + lane);
    uint32_t n = 0;
#pragma unroll
    for (uint32_t c = 0; c < kBucketSlots / 64; ++c) n += __popcll(__ballot(resident_live(raw[c], epoch)));
    if (lane == 0) nuniq[b] = n;
  }
}

// k_bucket_sort for a resident index: the slots stay where they are (the next pass finds them there); the bucket's
// live (hash, count) pairs are sorted in LDS and written straight to their place in the sketch (offs from
// k_resident_count + k_bucket_scan).
__global__ __launch_bounds__(256) void k_resident_sort_out(const Slot* __restrict__ tab, uint64_t nbuckets, uint32_t epoch,
                                                           const uint64_t* __restrict__ offs, uint32_t cs,
                                                           uint64_t* __restrict__ out_hashes, uint32_t* __restrict__ out_counts,
                                                           uint64_t out_cap) {
  __shared__ uint64_t s_keys[4][kBucketSlots];
  __shared__ uint32_t s_cnt[4][kBucketSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t* keys = s_keys[wave];
  uint32_t* cnt = s_cnt[wave];
  const uint64_t gwave = (uint64_t)blockIdx.x * 4 + wave, nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t b = gwave; b < nbuckets; b += nwaves) {
    uint4 raw[kBucketSlots / 64];
#pragma unroll
    for (uint32_t c = 0; c < kBucketSlots / 64; ++c) raw[c] = *reinterpret_cast<const uint4*>(tab + b * kBucketSlots + c * 64 + lane);
    uint32_t n = 0;
#pragma unroll
    for (uint32_t c = 0; c < kBucketSlots / 64; ++c) {
      const bool live = resident_live(raw[c], epoch);
      const unsigned long long m = __ballot(live);
      if (live) {
        const uint32_t d = n + __popcll(m & ((1ull << lane) - 1ull));
        keys[d] = ((uint64_t)raw[c].x | ((uint64_t)raw[c].y << 32)) - 1;
        const uint32_t seen = raw[c].z & ~kMovedOn;
        cnt[d] = (cs && seen > cs) ? cs : seen;
      }
      n += __popcll(m);
    }
    if (n == 0) continue;
    uint32_t N = 64;
    while (N < n) N <<= 1;
    for (uint32_t i = n + lane; i < N; i += 64) keys[i] = kReservedHash;
    wave_lds_sync();
    for (uint32_t k = 2; k <= N; k <<= 1) {
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t t = lane; t < N / 2; t += 64) {
          const uint32_t ix = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const uint32_t px = ix | j;
          const uint64_t x = keys[ix], y = keys[px];
          const bool up = (ix & k) == 0;
          if ((x > y) == up) {
            keys[ix] = y; keys[px] = x;
            const uint32_t cx = cnt[ix], cy = cnt[px];
            cnt[ix] = cy; cnt[px] = cx;
          }
        }
        wave_lds_sync();
      }
    }
    const uint64_t o = offs[b];
    for (uint32_t i = lane; i < n && o + i < out_cap; i += 64) { out_hashes[o + i] = keys[i]; out_counts[o + i] = cnt[i]; }
    wave_lds_sync();
  }
}

// ---- ... and the sketch from the LIST of slots the pass touched (mg_sketch_dev.h: resident_list_append), sorted ----
// Sorted slot numbers are in hash order bucket by bucket (a bucket is a hash range) but not within a bucket (a slot is the
// hash's low bits): k_list_keys fetches every listed slot's hash and counter, k_list_place puts the few entries of a bucket
// in hash order.  An entry counts if it is a slot (the unused part of the list is kNoSlot, sorted to the end) and differs
// from its predecessor (two lanes may have listed one slot).
__device__ __forceinline__ bool list_entry_counts(const uint32_t* __restrict__ sorted, uint64_t n, uint64_t i) {
  if (i >= n) return false;
  const uint32_t s = sorted[i];
  return s != kNoSlot && (i == 0 || sorted[i - 1] != s);
}
// nuniq[b] = entries that count among the 256 of block b.
__global__ __launch_bounds__(256) void k_list_count(const uint32_t* __restrict__ sorted, uint64_t n, uint32_t* __restrict__ nuniq) {
  __shared__ uint32_t s_n[4];
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long m = __ballot(list_entry_counts(sorted, n, i));
  if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) nuniq[blockIdx.x] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
}
// Per entry: how many entries before it count (before[i]), and the hash and counter of its slot (kReservedHash for an
// entry that does not count).
__global__ __launch_bounds__(256) void k_list_keys(const uint32_t* __restrict__ sorted, uint64_t n, const uint64_t* __restrict__ offs,
                                                   const Slot* __restrict__ tab, uint32_t epoch, uint32_t cs,
                                                   uint32_t* __restrict__ before, uint64_t* __restrict__ keys,
                                                   uint32_t* __restrict__ cnts) {
  __shared__ uint32_t s_n[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool mine = list_entry_counts(sorted, n, i);
  const unsigned long long m = __ballot(mine);
  if (lane == 0) s_n[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (i >= n) return;
  uint64_t at = offs[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) at += s_n[w];
  before[i] = (uint32_t)at;
  uint64_t key = kReservedHash;
  uint32_t seen = 0;
  if (mine) {
    const uint4 raw = *reinterpret_cast<const uint4*>(tab + sorted[i]);
    key = ((uint64_t)raw.x | ((uint64_t)raw.y << 32)) - 1;
    seen = raw.w == epoch ? raw.z & ~kMovedOn : 0u;
    if (cs && seen > cs) seen = cs;
  }
  keys[i] = key;
  cnts[i] = seen;
}
// An entry's place in the sketch: behind everything of earlier buckets (before[first entry of its bucket]) and behind the
// entries of its own bucket with smaller hashes.
__global__ __launch_bounds__(256) void k_list_place(const uint32_t* __restrict__ sorted, uint64_t n, const uint32_t* __restrict__ before,
                                                    const uint64_t* __restrict__ keys, const uint32_t* __restrict__ cnts,
                                                    uint64_t* __restrict__ out_hashes, uint32_t* __restrict__ out_counts,
                                                    uint64_t out_cap) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  if (key == kReservedHash) return;
  const uint32_t bucket = sorted[i] / kBucketSlots;
  uint64_t first = i;
  uint32_t smaller = 0;
  while (first > 0 && sorted[first - 1] / kBucketSlots == bucket) {
    --first;
    smaller += keys[first] < key;  // (an entry that does not count holds kReservedHash: never smaller)
  }
  for (uint64_t j = i + 1; j < n && sorted[j] / kBucketSlots == bucket; ++j) smaller += keys[j] < key;
  const uint64_t at = (uint64_t)before[first] + smaller;
  if (at >= out_cap) return;  // (reported by k_sketch_meta's cap)
  out_hashes[at] = key;
  out_counts[at] = cnts[i];
}

__global__ void k_sketch_split(const uint64_t* __restrict__ hashes, uint64_t n, const uint64_t* __restrict__ bounds,
                               uint32_t nbounds, uint64_t* __restrict__ out_idx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nbounds) return;
  const uint64_t key = bounds[i];
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (hashes[mid] < key) lo = mid + 1; else hi = mid;
  }
  out_idx[i] = lo;
}

// meta[0] = runs, then: apply the complete-part bound (entries > bound dropped) and the s cut;
// meta[1] = kept entries, meta[2] = last kept hash, meta[3] = 1 if anything was cut.
// cap > 0: the sketch's buffers hold cap entries; a table with more distinct hashes than that (its size was an
// estimate) is reported like a table overflow (meta[6]) and the sketch is redone by whoever resolves it.
__global__ void k_sketch_meta(const uint64_t* __restrict__ unique, uint64_t* __restrict__ meta, uint64_t s,
                              uint32_t use_bound, uint64_t bound, const unsigned long long* __restrict__ counters,
                              uint64_t* __restrict__ host_mirror, uint64_t cap) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (counters) { meta[4] = counters[0]; meta[5] = counters[1]; meta[6] = counters[2]; }
  uint64_t runs = meta[0];
  if (cap && runs > cap) {  // k_bucket_compact dropped what did not fit
    meta[6] = (counters ? meta[6] : 0) + 1;
    runs = cap;
    meta[0] = cap;
  }
  uint64_t keep = runs;
  uint64_t cut = 0;
  if (use_bound) {  // upper_bound(unique, bound)
    uint64_t lo = 0, hi = runs;
    while (lo < hi) {
      uint64_t mid = (lo + hi) >> 1;
      if (unique[mid] <= bound) lo = mid + 1; else hi = mid;
    }
    keep = lo;
    cut = 1;  // a truncated input makes the union a truncated sketch
  }
  if (s > 0 && keep > s) { keep = s; cut = 1; }
  meta[1] = keep;
  meta[2] = keep ? unique[keep - 1] : 0;
  meta[3] = cut;
  if (host_mirror) {  // page-locked host words, written over the bus: the host reads them after the stream sync
#pragma unroll
    for (int i = 0; i < 7; ++i) host_mirror[i] = meta[i];
  }
}

// counts[i] = min(counts[i], cs) for the meta[0] runs a run-length / reduce-by-key pass just wrote (the list path and
// the sorting merge count exactly; the sketch's counters saturate at cs).
__global__ void k_clamp_counts(uint32_t* __restrict__ counts, const uint64_t* __restrict__ meta, uint32_t cs) {
  const uint64_t n = meta[0];
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if (counts[i] > cs) counts[i] = cs;
}

// Finalise a sketch whose (hash,count) runs were written straight into its own buffers; d_meta[0] = runs.
// One read-back: [meta 0..3 | counters 0..2] -> pinned words 4..10.
static int adopt_runs(mg_sketch* sk, uint64_t* d_meta, uint64_t s, bool use_bound, uint64_t bound,
                      const unsigned long long* d_counters = nullptr, uint64_t* h_counters = nullptr) {
  hipStream_t st = ctx().stream;
  if (ctx().count_sat)  // (after the table path the counts are clamped already; this pass then changes nothing)
    hipLaunchKernelGGL(k_clamp_counts, dim3(ctx().num_cus * 4), dim3(256), 0, st, sk->counts.as<uint32_t>(), d_meta,
                       ctx().count_sat);
  hipLaunchKernelGGL(k_sketch_meta, dim3(1), dim3(64), 0, st, sk->hashes.as<uint64_t>(), d_meta, s,
                     (uint32_t)(use_bound ? 1 : 0), bound, (const unsigned long long*)nullptr, (uint64_t*)nullptr, (uint64_t)0);
  uint64_t* pin = host_words();
  MG_HIP(hipMemcpyAsync(pin + 4, d_meta, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  if (d_counters) MG_HIP(hipMemcpyAsync(pin + 8, d_counters, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  sk->n = pin[5];
  sk->last_hash = pin[6];
  sk->truncated = pin[7] ? 1 : 0;
  if (h_counters) { h_counters[0] = pin[8]; h_counters[1] = pin[9]; h_counters[2] = pin[10]; }
  return MG_OK;
}

template <int K>
static int launch_sketch_reads(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, uint64_t hmax,
                               uint64_t* d_cand, uint64_t cap, unsigned long long* d_counters, Slot* d_tab,
                               unsigned bucket_shift, unsigned stage_bytes, const mg_filter* filter = nullptr,
                               uint32_t epoch = 0) {
  Context& c = ctx();
  size_t lds = (size_t)kWavesPerBlock * (stage_bytes + kCandBuf * sizeof(uint64_t));
  if (dbg("lds_pad") > 0) lds += (size_t)dbg("lds_pad");  // occupancy experiments only
  const uint64_t ntiles = (nreads + 63) / 64;
  // enough resident blocks to fill every CU at the LDS-limited occupancy, grid-stride over the rest
  unsigned per_cu = (unsigned)(160 * 1024 / (lds + kHashTabEntries * sizeof(uint64_t)));  // (+ the static hash tables)
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  // On a stage-A stream (a pipelined job) the caller may cap the resident workgroups per CU
  // (mg_stage_a_workgroups_per_cu): a job whose pass has a long chain of small dependent kernels and host round
  // trips beside this kernel (the multi-GPU exchange) wants them to get issue slots — two per CU cost this kernel
  // ~12 % and took the exchange pass from 0.99 to 0.73 ms; a single-shard job leaves it at the LDS limit.
  if (c.a_side && c.is_stage_a(c.stream) && c.a_side_wg_per_cu && per_cu > c.a_side_wg_per_cu) per_cu = c.a_side_wg_per_cu;
  unsigned grid = grid_for(ntiles, kWavesPerBlock, (unsigned)c.num_cus * per_cu);
  ProfScope ps("sketch_reads");
  // (epoch != 0: d_tab is a resident index — the kernel takes "no filter words, a mask" as that and the mask as the epoch)
  const uint32_t* fb = filter && !epoch ? filter->bits.as<uint32_t>() : (const uint32_t*)nullptr;
  const uint64_t fm = epoch ? (uint64_t)epoch : (filter ? filter->mask : 0ull);
  if (c.hash_mode == kHashCmash)  // (instantiated in mg_sketch_cmash.hip)
    MG_TRY(launch_sketch_reads_cmash(K, grid, lds, c.stream, d_bases, d_offsets, nreads, hmax, d_cand, cap, d_counters, d_tab,
                                     bucket_shift, stage_bytes, fb, fm, stage_a_cs_word()));
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sketch_reads<K, kHashCanonical>), dim3(grid), dim3(kBlock), lds, c.stream, d_bases,
                       d_offsets, nreads, hmax, d_cand, cap, d_counters, d_tab, bucket_shift, stage_bytes, fb, fm, stage_a_cs_word());
  MG_HIP(hipGetLastError());
  return MG_OK;
}

// ---- partitioned counting table: plan, buffers, and the table -> sketch tail shared by reads and merges ----
struct TablePlan {
  uint64_t lo = 0;       // bucket = (hash - lo) >> shift
  unsigned shift = 0;
  uint64_t nbuckets = 0, slots = 0;
  Slot* tab = nullptr;        // [slots]: key = hash + 1 (0 = empty) and its counter
  uint32_t* nuniq = nullptr;
  uint64_t* offs = nullptr;
  uint32_t epoch = 0;         // != 0: tab is a resident index (seeded, never cleared; live slots carry this epoch)
  uint32_t* list = nullptr;   // ... and the kernel lists the SLOTS it touches here (listcap entries, kNoSlot = unused)
  uint64_t listcap = 0;
};

// Buckets of ~kBucketTarget expected distinct hashes over the key range [lo, hi]; false when the range or the
// estimate does not suit the table (caller takes the general path).
static bool plan_table(uint64_t lo, uint64_t hi, double distinct_est, TablePlan& tp) {
  if (hi < lo) return false;
  if (distinct_est < 4096.0) distinct_est = 4096.0;
  const uint64_t span = hi - lo;
  unsigned shift = bit_length(span);  // one bucket
  if (shift > 63) shift = 63;
  while (shift > 0 && distinct_est / (double)((span >> shift) + 1) > (double)kBucketTarget) --shift;
  tp.lo = lo;
  tp.shift = shift;
  tp.nbuckets = (span >> shift) + 1;
  if (tp.nbuckets > (1ull << 27)) return false;
  tp.slots = tp.nbuckets * kBucketSlots;
  return true;
}

// The table proper of k number `ki` of a fused launch (0 for everybody else): [16-byte slots x slots |
// 4 counter words], zeroed in one memset.
static int alloc_table_core(TablePlan& tp, unsigned long long** d_counters, int ki) {
  char name[24];
  snprintf(name, sizeof(name), ki ? "sk_table#%d" : "sk_table", ki);
  const uint64_t tab_bytes = tp.slots * sizeof(Slot);
  uint8_t* d_tab = (uint8_t*)scratch(name, tab_bytes + 4 * sizeof(unsigned long long));
  if (!d_tab) return MG_ERR_NOMEM;
  tp.tab = reinterpret_cast<Slot*>(d_tab);
  if (d_counters) *d_counters = reinterpret_cast<unsigned long long*>(d_tab + tab_bytes);
  ProfScope ps("table_clear");
  MG_HIP(hipMemsetAsync(d_tab, 0, tab_bytes + (d_counters ? 4 * sizeof(unsigned long long) : 0), ctx().stream));
  return MG_OK;
}

// Per-bucket counts and offsets of table_pack (shared by consecutive tables of a stream).
static int alloc_table_staging(TablePlan& tp) {
  tp.nuniq = (uint32_t*)scratch("sk_bucket_n", tp.nbuckets * sizeof(uint32_t));
  tp.offs = (uint64_t*)scratch("sk_bucket_off", (tp.nbuckets + 1) * sizeof(uint64_t));
  if (!tp.nuniq || !tp.offs) return MG_ERR_NOMEM;
  return MG_OK;
}

static int alloc_table(TablePlan& tp, unsigned long long** d_counters = nullptr) {
  MG_TRY(alloc_table_staging(tp));
  return alloc_table_core(tp, d_counters, 0);
}

// Sort every bucket and pack the buckets in order into the sketch's own buffers of `cap` entries (d_meta[0] = distinct
// hashes; what exceeds cap is dropped and must be detected by the caller: k_sketch_meta's cap).
static int table_pack(const TablePlan& tp, mg_sketch* sk, uint64_t* d_meta, uint64_t cap) {
  Context& c = ctx();
  hipStream_t st = c.stream;
  if (cap > tp.slots) cap = tp.slots;  // a sketch cannot outgrow the table
  if (tp.epoch && tp.list) {
    // the sketch = the listed slots' hashes in order, once each, with their counters: the work is the pass's distinct
    // hashes', not the table's (a sample covers a few per cent of a 200k-genome table: 1.7 against 15 ms for three k)
    MG_TRY(sk->hashes.alloc((cap + 1) * sizeof(uint64_t)));
    MG_TRY(sk->counts.alloc((cap + 1) * sizeof(uint32_t)));
    const uint64_t nl = tp.listcap, nblocks = (nl + 255) / 256;
    uint32_t* sorted = (uint32_t*)scratch("sk_rsorted", nl * sizeof(uint32_t));
    uint32_t* nuniq = (uint32_t*)scratch("sk_rlist_n", nblocks * sizeof(uint32_t));
    uint64_t* offs = (uint64_t*)scratch("sk_rlist_off", (nblocks + 1) * sizeof(uint64_t));
    uint32_t* before = (uint32_t*)scratch("sk_rlist_before", nl * sizeof(uint32_t));
    uint64_t* keys = (uint64_t*)scratch("sk_rlist_keys", nl * sizeof(uint64_t));
    uint32_t* cnts = (uint32_t*)scratch("sk_rlist_cnts", nl * sizeof(uint32_t));
    if (!sorted || !nuniq || !offs || !before || !keys || !cnts) return MG_ERR_NOMEM;
    {
      ProfScope ps("bucket_sort");
      MG_TRY(sort_keys_u32(tp.list, sorted, nl));
    }
    ProfScope ps("bucket_pack");
    hipLaunchKernelGGL(k_list_count, dim3((unsigned)nblocks), dim3(256), 0, st, sorted, nl, nuniq);
    hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, nuniq, nblocks, offs, d_meta);
    hipLaunchKernelGGL(k_list_keys, dim3((unsigned)nblocks), dim3(256), 0, st, sorted, nl, offs, tp.tab, tp.epoch, c.count_sat, before,
                       keys, cnts);
    hipLaunchKernelGGL(k_list_place, dim3((unsigned)nblocks), dim3(256), 0, st, sorted, nl, before, keys, cnts,
                       sk->hashes.as<uint64_t>(), sk->counts.as<uint32_t>(), cap);
    MG_HIP(hipGetLastError());
    return MG_OK;
  }
  if (tp.epoch) {  // (knob resident_scan: the same sketch from a walk over every slot of the index)
    const unsigned grid = grid_for(tp.nbuckets, 4, (unsigned)c.num_cus * 8);
    MG_TRY(sk->hashes.alloc((cap + 1) * sizeof(uint64_t)));
    MG_TRY(sk->counts.alloc((cap + 1) * sizeof(uint32_t)));
    {
      ProfScope ps("bucket_sort");
      hipLaunchKernelGGL(k_resident_count, dim3(grid), dim3(256), 0, st, tp.tab, tp.nbuckets, tp.epoch, tp.nuniq);
      hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, tp.nuniq, tp.nbuckets, tp.offs, d_meta);
      MG_HIP(hipGetLastError());
    }
    ProfScope ps("bucket_pack");
    hipLaunchKernelGGL(k_resident_sort_out, dim3(grid), dim3(256), 0, st, tp.tab, tp.nbuckets, tp.epoch, tp.offs, c.count_sat,
                       sk->hashes.as<uint64_t>(), sk->counts.as<uint32_t>(), cap);
    MG_HIP(hipGetLastError());
    return MG_OK;
  }
  {
    ProfScope ps("bucket_sort");
    hipLaunchKernelGGL(k_bucket_sort, dim3(grid_for(tp.nbuckets, 4, (unsigned)c.num_cus * 8)), dim3(256), 0, st, tp.tab,
                       tp.nbuckets, tp.nuniq, c.count_sat);
    MG_HIP(hipGetLastError());
  }
  MG_TRY(sk->hashes.alloc((cap + 1) * sizeof(uint64_t)));
  MG_TRY(sk->counts.alloc((cap + 1) * sizeof(uint32_t)));
  {
    ProfScope ps("bucket_pack");
    hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, tp.nuniq, tp.nbuckets, tp.offs, d_meta);
    hipLaunchKernelGGL(k_bucket_compact, dim3(grid_for(tp.nbuckets, 4, (unsigned)c.num_cus * 8)), dim3(256), 0, st,
                       tp.tab, tp.nuniq, tp.offs, tp.nbuckets, sk->hashes.as<uint64_t>(),
                       sk->counts.as<uint32_t>(), cap);
    MG_HIP(hipGetLastError());
  }
  return MG_OK;
}

// ... then apply bound / s and read back.
static int table_to_sketch(const TablePlan& tp, mg_sketch* sk, uint64_t* d_meta, uint64_t s, bool use_bound, uint64_t bound,
                           const unsigned long long* d_counters, uint64_t* h_counters, uint64_t cap) {
  MG_TRY(table_pack(tp, sk, d_meta, cap));
  return adopt_runs(sk, d_meta, s, use_bound, bound, d_counters, h_counters);
}

// List path: flat candidate list -> rocPRIM radix sort -> run-length encode (any size, any distribution).
static int sketch_via_list(mg_sketch* sk, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, int k,
                           uint64_t hmax, uint64_t s, uint64_t cap, unsigned stage, unsigned long long* d_counters,
                           const mg_filter* filter = nullptr) {
  hipStream_t st = ctx().stream;
  uint64_t* pin = host_words();
  uint64_t ncand = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    uint64_t* d_cand = (uint64_t*)scratch("sk_cand", cap * sizeof(uint64_t));
    if (!d_cand) return MG_ERR_NOMEM;
    MG_HIP(hipMemsetAsync(d_counters, 0, 4 * sizeof(unsigned long long), st));
    int rc = MG_ERR_ARG;
    dispatch_k(k, [&]<int K>() {
      rc = launch_sketch_reads<K>(d_bases, d_offsets, nreads, hmax, d_cand, cap, d_counters, nullptr, 64u, stage, filter);
    });
    if (rc) return rc;
    MG_HIP(hipMemcpyAsync(pin + 2, d_counters, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st));
    ncand = pin[2];
    sk->kmers_seen = pin[3];
    if (ncand <= cap) break;
    if (attempt == 1) return fail(MG_ERR_CAPACITY, "candidate list overflow after retry");
    cap = ncand + 64;  // exact size is known now; rerun once
  }
  uint64_t* d_cand = (uint64_t*)scratch("sk_cand", cap * sizeof(uint64_t));
  uint64_t* d_sorted = (uint64_t*)scratch("sk_sorted", (ncand + 1) * sizeof(uint64_t));
  if (!d_sorted) return MG_ERR_NOMEM;
  // the run-length pass writes straight into the sketch's own (pooled) buffers, sized for the worst case
  MG_TRY(sk->hashes.alloc((ncand + 1) * sizeof(uint64_t)));
  MG_TRY(sk->counts.alloc((ncand + 1) * sizeof(uint32_t)));
  uint64_t* d_meta = reinterpret_cast<uint64_t*>(d_counters) + 4;
  {
    ProfScope ps("sketch_sort");
    MG_TRY(sort_keys(d_cand, d_sorted, ncand, bit_length(hmax)));
  }
  {
    ProfScope ps("sketch_rle");
    MG_TRY(rle_keys(d_sorted, ncand, sk->hashes.as<uint64_t>(), sk->counts.as<uint32_t>(), d_meta));
  }
  return adopt_runs(sk, d_meta, s, false, 0);
}

// While alive, the library launches on `st` and uses that stream's own scratch buffers.
struct StreamGuard {
  hipStream_t saved;
  const char* saved_prefix;
  StreamGuard(hipStream_t st, const char* prefix) : saved(ctx().stream), saved_prefix(ctx().scratch_prefix) {
    ctx().stream = st;
    ctx().scratch_prefix = prefix;
  }
  ~StreamGuard() { ctx().stream = saved; ctx().scratch_prefix = saved_prefix; }
};

static hipEvent_t take_event() {
  Context& c = ctx();
  if (!c.ev_pool.empty()) { hipEvent_t e = c.ev_pool.back(); c.ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}

int sketch_wait(const mg_sketch* sk) {
  if (sk && sk->ev && sk->ev_stream != ctx().stream) MG_HIP(hipStreamWaitEvent(ctx().stream, sk->ev, 0));
  return MG_OK;
}

// distinct / expected candidates of the previous batch of the same k (x2): sizes the counting table
// The next table of the same k is sized for this many times the distinct hashes the last one held (a bucket then
// expects at most kBucketTarget / kHintSafety of them; 256 slots are 12 standard deviations away, and a sample twice
// as diverse as its predecessor is caught by the overflow counter and redone on the list path).
constexpr double kHintSafety = 1.25;
static double distinct_hint_k[MG_MAX_K + 1];
static bool distinct_hint_init = false;
static double& distinct_hint_for(int k) {
  if (!distinct_hint_init) {
    for (double& v : distinct_hint_k) v = 1.0;
    distinct_hint_init = true;
  }
  return distinct_hint_k[(k >= 0 && k <= MG_MAX_K) ? k : 0];
}

int sketch_resolve(mg_sketch* sk, int* rebuilt) {
  if (rebuilt) *rebuilt = 0;
  if (!sk || !sk->pending) return MG_OK;
  Context& c = ctx();
  if (sk->ev) MG_HIP(hipEventSynchronize(sk->ev));  // only what built this sketch, not whatever was queued after it
  else MG_HIP(hipStreamSynchronize(c.stream));
  const uint64_t* m = sk->h_meta;
  const uint64_t runs = m[0], candidates = m[4], overflows = m[6];
  sk->n = m[1];
  sk->last_hash = m[2];
  sk->truncated = m[3] ? 1 : 0;
  sk->kmers_seen = m[5];
  sk->pending = false;
  if (sk->pend_slot >= 0 && c.pend_owner[sk->pend_slot] == sk) c.pend_owner[sk->pend_slot] = nullptr;
  sk->pend_slot = -1;
  if (overflows == 0) {
    if (!sk->redo.is_merge) {
      const double r = kHintSafety * (double)runs / sk->expect;
      distinct_hint_for(sk->redo.k) = r < 0.02 ? 0.02 : (r > 1.0 ? 1.0 : r);
    }
    return MG_OK;
  }
  if (sk->redo.is_merge) {  // a pair outside the declared range, or a full bucket: the general (sorting) merge
    if (rebuilt) *rebuilt = 1;
    sk->index.release();
    sk->hashes.release();
    sk->counts.release();
    MG_TRY(::merge_via_sort(sk, sk->redo.m_hashes, sk->redo.m_counts, sk->redo.m_n, sk->redo.s, sk->redo.use_bound ? 1 : 0,
                          sk->redo.m_bound));
    return MG_OK;
  }
  // a bucket ran out of slots (more distinct hashes than the hint allowed for): size for the worst case next
  // time and redo this sketch on the list path, which has no such limit
  distinct_hint_for(sk->redo.k) = 1.0;
  if (!sk->redo.bases)  // a streamed sketch (mg_sketch_stream_*): its reads are gone; the caller streams them again
    return fail(MG_ERR_CAPACITY, "streamed read sketch (k = %d): a counting-table bucket overflowed (%llu candidates for %llu "
                "expected); the table hint is reset — stream the reads again", sk->redo.k, (unsigned long long)candidates,
                (unsigned long long)sk->expect);
  if (rebuilt) *rebuilt = 1;
  sk->index.release();
  sk->hashes.release();
  sk->counts.release();
  if (sk->redo.s == 0 && sk->redo.filter && sk->redo.filter->resident && !sk->redo.filter->resident_off) {
    // made against a resident index, whose sketch is the exact intersection (the list path's filter lets ~6 % of the
    // other hashes through): the list of touched hashes, or the sketch's buffers, were too small — again, with room for
    // every hash of the index (the hint has just been reset)
    StreamGuard guard(sk->ev_stream ? sk->ev_stream : c.stream, c.is_stage_a(sk->ev_stream) ? c.stage_a_prefix(sk->ev_stream) : c.scratch_prefix);
    return ::redo_resident(sk);
  }
  uint64_t cap = sk->redo.cap;
  if (candidates + 64 > cap) cap = candidates + 64;
  // (the rebuild is synchronous on the stream that built the sketch)
  StreamGuard guard(sk->ev_stream ? sk->ev_stream : c.stream, c.is_stage_a(sk->ev_stream) ? c.stage_a_prefix(sk->ev_stream) : c.scratch_prefix);
  unsigned long long* d_counters = (unsigned long long*)scratch("sk_counters", 8 * sizeof(unsigned long long));
  if (!d_counters) return MG_ERR_NOMEM;
  return sketch_via_list(sk, sk->redo.bases, sk->redo.offsets, sk->redo.nreads, sk->redo.k, sk->redo.hmax, sk->redo.s, cap,
                         sk->redo.stage, d_counters, sk->redo.filter);
}

}  // namespace mg

mg_filter::Resident::~Resident() {
  for (auto& cp : copies)
    if (cp.slots) (void)hipFree(cp.slots);  // (synchronises the device: nothing is still counting in them afterwards)
}

mg_sketch::~mg_sketch() {
  mg::Context& c = mg::ctx();
  if (ev) c.ev_pool.push_back(ev);
  if (pend_slot >= 0 && c.pend_owner[pend_slot] == this) c.pend_owner[pend_slot] = nullptr;
}

using namespace mg;

// Membership pre-filter: bit (h mod nbits) for every hash of the set.
__global__ void k_filter_set(const uint64_t* __restrict__ hashes, uint64_t n, uint32_t* __restrict__ bits, uint64_t mask) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const uint64_t b = hashes[i] & mask;
    atomicOr(&bits[b >> 5], 1u << (b & 31u));
  }
}

// Sizes derived from the read buffer: total bases (one read-back, cached per input buffer: a stale value only
// mis-sizes buffers, and every mis-sizing is detected and retried) and the LDS tile of a wavefront.
struct ReadPlan {
  uint64_t nbases = 0;
  unsigned stage = 0;
};
static int plan_reads(const uint64_t* d_offsets, uint64_t nreads, hipStream_t st, ReadPlan& rp) {
  uint64_t* pin = host_words();
  static const uint64_t* cached_off = nullptr;
  static uint64_t cached_nreads = 0, cached_nbases = 0;
  if (cached_off == d_offsets && cached_nreads == nreads) {
    rp.nbases = cached_nbases;
  } else {
    MG_HIP(hipMemcpyAsync(pin + 0, d_offsets, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MG_HIP(hipMemcpyAsync(pin + 1, d_offsets + nreads, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st));
    rp.nbases = pin[1] - pin[0];
    cached_off = d_offsets; cached_nreads = nreads; cached_nbases = rp.nbases;
  }
  // LDS tile: 64 reads of average length at half a byte per base, 12.5 % slack, 16-byte granules, at most
  // 8 KiB (16 k bases) per wavefront
  const uint64_t avg = (rp.nbases + nreads - 1) / nreads;
  uint64_t stage = (((64 * avg * 9 / 8 + 64) / 2 + 15) / 16) * 16;
  if (stage < 1024) stage = 1024;
  if (stage > 8192) stage = 8192;
  rp.stage = (unsigned)stage;
  return MG_OK;
}

struct KPlan {  // one k of a sketch call
  uint64_t hmax = 0, expect = 0, cap = 0;
  double distinct_est = 0.0;
  bool table = false;
  TablePlan tp;
};
// The share of all k-mers whose hash is <= hmax: hashes are uniform over 2^64 under definition 0 and over [0, 9999999999971)
// under definition 1 (mg_set_hash_mode) — where a table of k-prefixes (build_db --prefix_tables) has its largest key near
// the prime and EVERY k-mer is a candidate.
static double hash_fraction(uint64_t hmax) {
  const double range = ctx().hash_mode == kHashCmash ? (double)kCmashPrime : 18446744073709551616.0;
  const double f = ((double)hmax + 1.0) / range;
  return f < 1.0 ? f : 1.0;
}

static void plan_k(const ReadPlan& rp, int k, uint64_t hmax, KPlan& kp) {
  kp.hmax = hmax;
  const double frac = hash_fraction(hmax);
  kp.expect = (uint64_t)((double)rp.nbases * frac);
  kp.cap = kp.expect + kp.expect / 4 + (1u << 16);
  if (kp.cap > rp.nbases + 64) kp.cap = rp.nbases + 64;
  // ---- table path: counting hash table partitioned into hash-range buckets ----
  // Sized from the expected number of DISTINCT candidates: `expect` bounds it; the ratio observed on the
  // previous call of the same k (x kHintSafety) tightens it for steady-state batches.  Under-sizing is detected (a
  // bucket with no free slot) and handled by the list path.
  const bool force_list = dbg("force_list") != 0;
  kp.distinct_est = (double)kp.expect * mg::distinct_hint_for(k);
  if (dbg("distinct_hint_ppm") > 0) kp.distinct_est = (double)kp.expect * (double)dbg("distinct_hint_ppm") * 1e-6;  // tests: force overflow
  kp.table = !force_list && kp.expect >= 32768 && plan_table(0, hmax, kp.distinct_est, kp.tp);
}

// A call that sketches against the filter's RESIDENT INDEX (mg_internal.h: mg_filter::Resident): no table to size or
// clear — this stream's copy of the seeded slots (made on its first use: two passes in flight on two streams must not
// count in the same slots) and an epoch nobody has used on any copy.  The threshold is the table's own largest hash at
// most: nothing above it can be in the index, whose buckets end there.
static uint32_t next_resident_epoch() {
  Context& c = ctx();
  if (++c.resident_epoch == 0) ++c.resident_epoch;  // (0 = "not resident"; a wrap after 2^32 passes is not handled)
  return c.resident_epoch;
}
static int plan_resident(const mg_filter* f, const ReadPlan& rp, int k, uint64_t hmax, uint32_t epoch, KPlan& kp,
                         unsigned long long** t_counters, int ki, bool full = false) {
  mg_filter::Resident& R = *f->resident;
  Context& c = ctx();
  hipStream_t st = c.stream;
  void* slots = nullptr;
  for (auto& cp : R.copies)
    if (cp.stream == st) slots = cp.slots;
  if (!slots && R.copies[0].stream == nullptr) { R.copies[0].stream = st; slots = R.copies[0].slots; }
  if (!slots) {
    mg_filter::Resident::Copy cp;
    MG_HIP(hipDeviceSynchronize());  // (counters of passes in flight elsewhere may be copied torn: their epochs are never reused)
    if (hipMalloc(&cp.slots, R.slots * sizeof(Slot)) != hipSuccess) {
      pool_release_all();
      MG_HIP(hipMalloc(&cp.slots, R.slots * sizeof(Slot)));
    }
    cp.stream = st;
    R.copies.push_back(cp);
    MG_HIP(hipMemcpyAsync(cp.slots, R.copies[0].slots, R.slots * sizeof(Slot), hipMemcpyDeviceToDevice, st));
    // The SOURCE belongs to another stream, whose next pass — queued by the caller right after this call returns, with a
    // LARGER epoch — must not write slots the copy has not read yet: a slot copied with that later epoch in it would read
    // as "count 0" to this call's smaller one (atomicMax of an epoch below the stored one is a no-op).  Once per stream.
    MG_HIP(hipStreamSynchronize(st));
    slots = cp.slots;
  }
  if (hmax > R.hmax) hmax = R.hmax;
  kp.hmax = hmax;
  kp.expect = (uint64_t)((double)rp.nbases * hash_fraction(hmax));
  kp.cap = kp.expect + kp.expect / 4 + (1u << 16);
  if (kp.cap > rp.nbases + 64) kp.cap = rp.nbases + 64;
  // the sketch cannot hold a hash the index does not; fewer when the previous sketch of this k held fewer (an estimate: a
  // list or a sketch too small is reported like a table overflow, and the sketch made again with room for every hash)
  kp.distinct_est = (double)kp.expect * mg::distinct_hint_for(k);
  if (full || kp.distinct_est > (double)R.distinct || dbg("resident_scan")) kp.distinct_est = (double)R.distinct;
  const double tight = full ? 0.0 : (double)dbg("distinct_hint_ppm") * 1e-6;  // tests: force the overflow
  if (tight > 0.0) kp.distinct_est = (double)kp.expect * tight;
  kp.table = true;
  kp.tp.lo = 0;
  kp.tp.shift = R.shift;
  kp.tp.nbuckets = R.nbuckets;
  kp.tp.slots = R.slots;
  kp.tp.tab = reinterpret_cast<Slot*>(slots);
  kp.tp.epoch = epoch;
  char name[24];
  snprintf(name, sizeof(name), "sk_rcounters#%d", ki);
  *t_counters = (unsigned long long*)scratch(name, 4 * sizeof(unsigned long long));
  if (!*t_counters) return MG_ERR_NOMEM;
  MG_HIP(hipMemsetAsync(*t_counters, 0, 4 * sizeof(unsigned long long), st));
  if (!dbg("resident_scan")) {
    // room for the estimate, the few hashes two lanes list, and a last chunk per wavefront of the largest grid
    kp.tp.listcap = (uint64_t)kp.distinct_est + (uint64_t)kp.distinct_est / 16 + (uint64_t)c.num_cus * 8 * kWavesPerBlock * kListChunk + 4096;
    if (tight > 0.0) kp.tp.listcap = (uint64_t)kp.distinct_est + 16 * kListChunk;  // (tests: ... of the list too)
    snprintf(name, sizeof(name), "sk_rlist#%d", ki);
    kp.tp.list = (uint32_t*)scratch(name, kp.tp.listcap * sizeof(uint32_t));
    if (!kp.tp.list) return MG_ERR_NOMEM;
    MG_HIP(hipMemsetAsync(kp.tp.list, 0xff, kp.tp.listcap * sizeof(uint32_t), st));
  }
  return MG_OK;
}
// (a bottom-s sketch keeps the bit filter's definition — the s smallest of what passes the FILTER — so that the cut does
// not depend on whether the table has an index)
static bool use_resident(const mg_filter* f, uint64_t s) { return s == 0 && f && f->resident && !f->resident_off; }

// sketch_resolve's way out when the list of touched hashes (or the sketch's buffers) of a resident sketch was too small:
// the one-k kernel again, synchronously, sized for every hash of the index.
static int redo_resident(mg_sketch* sk) {
  ReadPlan rp;
  rp.nbases = sk->redo.nbases;
  rp.stage = sk->redo.stage;
  KPlan kp;
  unsigned long long* t_counters = nullptr;
  MG_TRY(plan_resident(sk->redo.filter, rp, sk->redo.k, sk->redo.hmax, next_resident_epoch(), kp, &t_counters, 0, true));
  MG_TRY(alloc_table_staging(kp.tp));
  int rc = MG_ERR_ARG;
  const bool ok = dispatch_k(sk->redo.k, [&]<int K>() {
    rc = launch_sketch_reads<K>(sk->redo.bases, sk->redo.offsets, sk->redo.nreads, kp.hmax, reinterpret_cast<uint64_t*>(kp.tp.list),
                                kp.tp.listcap, t_counters, kp.tp.tab, kp.tp.shift, rp.stage, sk->redo.filter, kp.tp.epoch);
  });
  if (!ok) return fail(MG_ERR_ARG, "unsupported k=%d", sk->redo.k);
  if (rc) return rc;
  MG_TRY(sk->meta.alloc(8 * sizeof(uint64_t)));
  uint64_t h_counters[3] = {0, 0, 0};
  MG_TRY(table_to_sketch(kp.tp, sk, sk->meta.as<uint64_t>(), sk->redo.s, false, 0, t_counters, h_counters,
                         sk->redo.filter->resident->distinct + 1024));
  if (h_counters[2]) return fail(MG_ERR_CAPACITY, "resident sketch (k = %d): the list of touched hashes overflowed at full size", sk->redo.k);
  return MG_OK;
}

// Deferred finalisation of a sketch whose counting table has just been filled: bucket sort / pack, then the sketch's
// size, last hash and the table-overflow counter stay on the device with a copy in flight to pinned memory;
// sketch_resolve() reads them at the first host-side use.
static int finish_pending(mg_sketch* sk, const KPlan& kp, const ReadPlan& rp, unsigned long long* t_counters, int k,
                          uint64_t s, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads,
                          const mg_filter* filter) {
  Context& cc = ctx();
  hipStream_t st = cc.stream;
  const unsigned slot = cc.pend_next++ % Context::kPendSlots;
  if (cc.pend_owner[slot]) MG_TRY(sketch_resolve(cc.pend_owner[slot], nullptr));  // ring full: settle the oldest
  MG_TRY(sk->meta.alloc(8 * sizeof(uint64_t)));
  uint64_t* sk_meta = sk->meta.as<uint64_t>();
  // the sketch's buffers are sized from the distinct-count estimate (which is the candidate count itself until a
  // first batch has been seen, and kHintSafety times the observed ratio afterwards), not from the table's slots
  // (2 to 4 per expected entry): 12 B instead of 32 to 64 per expected entry and sketch in flight
  uint64_t cap = (uint64_t)kp.distinct_est + 1024;
  if (cap > kp.tp.slots) cap = kp.tp.slots;
  MG_TRY(table_pack(kp.tp, sk, sk_meta, cap));
  sk->h_meta = cc.pend_pinned + 8 * slot;
  hipLaunchKernelGGL(k_sketch_meta, dim3(1), dim3(64), 0, st, sk->hashes.as<uint64_t>(), sk_meta, s, 0u, (uint64_t)0,
                     (const unsigned long long*)t_counters, sk->h_meta, cap);
  MG_HIP(hipGetLastError());
  sk->ev = take_event();
  if (sk->ev) { MG_HIP(hipEventRecord(sk->ev, st)); sk->ev_stream = st; }
  sk->pending = true;
  sk->pend_slot = (int)slot;
  cc.pend_owner[slot] = sk;
  sk->n_bound = cap;
  if (s > 0 && s < sk->n_bound) sk->n_bound = s;
  sk->hmax = kp.hmax;
  sk->expect = (double)(kp.expect ? kp.expect : 1);
  sk->redo.bases = d_bases; sk->redo.offsets = d_offsets; sk->redo.nreads = nreads; sk->redo.k = k;
  sk->redo.hmax = kp.hmax; sk->redo.s = s; sk->redo.cap = kp.cap; sk->redo.stage = rp.stage; sk->redo.nbases = rp.nbases;
  sk->redo.filter = filter;
  return MG_OK;
}

static int sketch_reads_async(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, int k, uint64_t hmax,
                              uint64_t s, const mg_filter* filter, mg_sketch** out) {
  MG_REQUIRE_READY();
  if (!out) return fail(MG_ERR_ARG, "null out handle");
  *out = nullptr;
  if (k < 1 || k > MG_MAX_K) return fail(MG_ERR_ARG, "k=%d outside [1,%d]", k, MG_MAX_K);
  if (nreads > 0 && (!d_bases || !d_offsets)) return fail(MG_ERR_ARG, "null device input");
  if (hmax == kReservedHash) hmax = kReservedHash - 1;
  Context& c = ctx();
  // with mg_stage_a_side_stream on, the whole sketch pipeline of this call goes to the stage-A stream
  hipStream_t side = c.a_side == 2 ? c.stream_a2 : c.stream_a;
  StreamGuard guard(c.a_side ? side : c.stream, c.a_side ? c.stage_a_prefix(side) : c.scratch_prefix);
  hipStream_t st = c.stream;
  std::unique_ptr<mg_sketch> sk(new mg_sketch());
  if (nreads == 0) {
    MG_TRY(sk->hashes.alloc(0));
    MG_TRY(sk->counts.alloc(0));
    *out = sk.release();
    return MG_OK;
  }
  ReadPlan rp;
  MG_TRY(plan_reads(d_offsets, nreads, st, rp));
  KPlan kp;
  unsigned long long* t_counters = nullptr;  // cleared together with the table
  if (use_resident(filter, s)) MG_TRY(plan_resident(filter, rp, k, hmax, next_resident_epoch(), kp, &t_counters, 0));
  else plan_k(rp, k, hmax, kp);
  unsigned long long* d_counters = (unsigned long long*)scratch("sk_counters", 8 * sizeof(unsigned long long));
  if (!d_counters) return MG_ERR_NOMEM;
  if (kp.table) {
    if (kp.tp.epoch) MG_TRY(alloc_table_staging(kp.tp));
    else MG_TRY(alloc_table(kp.tp, &t_counters));
    int rc = MG_ERR_ARG;
    bool ok = dispatch_k(k, [&]<int K>() {
      rc = launch_sketch_reads<K>(d_bases, d_offsets, nreads, kp.hmax, reinterpret_cast<uint64_t*>(kp.tp.list), kp.tp.listcap,
                                  t_counters, kp.tp.tab, kp.tp.shift, rp.stage, filter, kp.tp.epoch);
    });
    if (!ok) return fail(MG_ERR_ARG, "unsupported k=%d", k);
    if (rc) return rc;
    MG_TRY(finish_pending(sk.get(), kp, rp, t_counters, k, s, d_bases, d_offsets, nreads, filter));
    *out = sk.release();
    return MG_OK;
  }
  MG_TRY(sketch_via_list(sk.get(), d_bases, d_offsets, nreads, k, hmax, s, kp.cap, rp.stage, d_counters, filter));
  MG_HIP(hipStreamSynchronize(st));  // the list path reads back as it goes; nothing is left in flight
  *out = sk.release();
  return MG_OK;
}

// Every k of a multi-k query from ONE pass over the reads (mg_sketch_multi.hip) when the k set has a fused kernel and
// every k takes the counting-table path; otherwise one launch per k.  out[i] = the sketch of ks[i], pending.
static int sketch_reads_multi_async(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, int nk, const int* ks,
                                    const uint64_t* hmaxs, uint64_t s, const mg_filter* const* filters, mg_sketch** out) {
  MG_REQUIRE_READY();
  if (!out || !ks || !hmaxs || nk < 1) return fail(MG_ERR_ARG, "null argument");
  for (int i = 0; i < nk; ++i) out[i] = nullptr;
  auto per_k = [&]() -> int {
    for (int i = 0; i < nk; ++i) {
      int rc = sketch_reads_async(d_bases, d_offsets, nreads, ks[i], hmaxs[i], s, filters ? filters[i] : nullptr, &out[i]);
      if (rc != MG_OK) {
        for (int j = 0; j < i; ++j) { mg_sketch_free(out[j]); out[j] = nullptr; }
        return rc;
      }
    }
    return MG_OK;
  };
  if (nreads == 0 || nk == 1 || nk > 4 || !sketch_reads_multi_supported(ks, nk) || dbg("no_fused")) return per_k();
  if (!d_bases || !d_offsets) return fail(MG_ERR_ARG, "null device input");
  Context& c = ctx();
  hipStream_t side = c.a_side == 2 ? c.stream_a2 : c.stream_a;
  StreamGuard guard(c.a_side ? side : c.stream, c.a_side ? c.stage_a_prefix(side) : c.scratch_prefix);
  hipStream_t st = c.stream;
  ReadPlan rp;
  MG_TRY(plan_reads(d_offsets, nreads, st, rp));
  KPlan kp[4];
  uint64_t hm[4];
  bool resident = filters != nullptr;  // every k against its table's resident index, or none
  for (int i = 0; i < nk && resident; ++i) resident = use_resident(filters[i], s);
  MultiKTable tabs[4];
  unsigned long long* t_counters[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < nk; ++i) {
    hm[i] = hmaxs[i] == kReservedHash ? kReservedHash - 1 : hmaxs[i];
    if (resident) continue;
    plan_k(rp, ks[i], hm[i], kp[i]);
    if (!kp[i].table) return per_k();  // few candidates (list path) for some k: nothing to fuse
  }
  const uint32_t epoch = resident ? next_resident_epoch() : 0u;  // one for all k of the launch
  for (int i = 0; i < nk; ++i) {
    if (resident) {
      MG_TRY(plan_resident(filters[i], rp, ks[i], hm[i], epoch, kp[i], &t_counters[i], i));
    } else {
      MG_TRY(alloc_table_core(kp[i].tp, &t_counters[i], i));
    }
    tabs[i] = MultiKTable{kp[i].hmax, kp[i].tp.tab, t_counters[i], kp[i].tp.shift, filters ? filters[i] : nullptr, kp[i].tp.epoch,
                          kp[i].tp.list, kp[i].tp.listcap};
  }
  MG_TRY(launch_sketch_reads_multi(ks, nk, d_bases, d_offsets, nreads, tabs, rp.stage));
  for (int i = 0; i < nk; ++i) {
    std::unique_ptr<mg_sketch> sk(new mg_sketch());
    int rc = alloc_table_staging(kp[i].tp);
    if (rc == MG_OK)
      rc = finish_pending(sk.get(), kp[i], rp, t_counters[i], ks[i], s, d_bases, d_offsets, nreads, filters ? filters[i] : nullptr);
    if (rc != MG_OK) {
      for (int j = 0; j < i; ++j) { mg_sketch_free(out[j]); out[j] = nullptr; }
      return rc;
    }
    out[i] = sk.release();
  }
  return MG_OK;
}

// ---- streamed read sketch: one set of counting tables, any number of read batches ---------------------------------
// A sample that arrives in pieces (a file streamed through page-locked chunks while the next chunk is in flight, or a
// file larger than the device) hashes every piece into the SAME per-k counting tables — the tables are what dedupes and
// counts, so nothing is sketched per piece and nothing is merged (round 2 made a sketch per piece and merged pairs of
// them through the host).  Replaces: kmc reading the whole reads file (scripts/select_db.py:45-52).
struct mg_sketch_stream {
  int nk = 0;
  int ks[4] = {0, 0, 0, 0};
  uint64_t hmax[4] = {0, 0, 0, 0};
  uint64_t s = 0;
  const mg_filter* filters[4] = {nullptr, nullptr, nullptr, nullptr};
  KPlan kp[4];
  DevBuf tab[4];                           // [slots x 16 B | 4 counter words] per k, zeroed at begin
  unsigned long long* t_counters[4] = {nullptr, nullptr, nullptr, nullptr};
  bool fused = false;
  uint64_t nreads = 0, nbases = 0, expect_bases = 0;
  // mg_sketch_stream_begin_counts: the pieces are not sketched but COUNTED, by k-mer identity, into these (mg_kcount.hip)
  const mg_refdb* count_db = nullptr;
  mg_kcounts* count_kc = nullptr;
};

static unsigned stage_bytes_for(uint64_t nbases, uint64_t nreads) {
  const uint64_t avg = nreads ? (nbases + nreads - 1) / nreads : 0;
  uint64_t stage = (((64 * avg * 9 / 8 + 64) / 2 + 15) / 16) * 16;
  if (stage < 1024) stage = 1024;
  if (stage > 8192) stage = 8192;
  return (unsigned)stage;
}

static int stream_begin(int nk, const int* ks, const uint64_t* hmaxs, uint64_t s, const mg_filter* const* filters,
                        uint64_t expect_bases, mg_sketch_stream** out) {
  MG_REQUIRE_READY();
  if (!out || !ks || !hmaxs) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  if (nk < 1 || nk > 4) return fail(MG_ERR_ARG, "between 1 and 4 k per streamed sketch");
  std::unique_ptr<mg_sketch_stream> ss(new mg_sketch_stream());
  ss->nk = nk;
  ss->s = s;
  ss->expect_bases = expect_bases ? expect_bases : 1;
  ReadPlan rp;
  rp.nbases = ss->expect_bases;
  hipStream_t st = ctx().stream;
  for (int i = 0; i < nk; ++i) {
    if (ks[i] < 1 || ks[i] > MG_MAX_K) return fail(MG_ERR_ARG, "k=%d outside [1,%d]", ks[i], MG_MAX_K);
    if (i && ks[i] <= ks[i - 1]) return fail(MG_ERR_ARG, "the k of a streamed sketch must ascend");
    ss->ks[i] = ks[i];
    ss->hmax[i] = hmaxs[i] == kReservedHash ? kReservedHash - 1 : hmaxs[i];
    ss->filters[i] = filters ? filters[i] : nullptr;
    KPlan& kp = ss->kp[i];
    plan_k(rp, ks[i], ss->hmax[i], kp);
    // A table that proves too small costs a second pass over the whole FILE here (the reads are gone), not a redo from
    // resident reads: twice the room the one-shot path takes (the hint is the previous sample's ratio x 1.25; a sample
    // up to 2.5 x as diverse as its predecessor still fits), never more than one slot pair per expected candidate.
    kp.distinct_est = 2.0 * kp.distinct_est < (double)kp.expect ? 2.0 * kp.distinct_est : (double)kp.expect;
    kp.table = kp.table && plan_table(0, ss->hmax[i], kp.distinct_est, kp.tp);
    // always the table path here (the list path wants all candidates of the sample in one buffer): a sample too small
    // for plan_k's threshold gets the smallest table
    if (!kp.table && !plan_table(0, ss->hmax[i], kp.distinct_est, kp.tp))
      return fail(MG_ERR_ARG, "k=%d: no counting table for this threshold", ks[i]);
    kp.table = true;
    const uint64_t tab_bytes = kp.tp.slots * sizeof(Slot);
    MG_TRY(ss->tab[i].alloc(tab_bytes + 4 * sizeof(unsigned long long)));
    kp.tp.tab = ss->tab[i].as<Slot>();
    ss->t_counters[i] = reinterpret_cast<unsigned long long*>(ss->tab[i].as<uint8_t>() + tab_bytes);
    ProfScope ps("table_clear");
    MG_HIP(hipMemsetAsync(ss->tab[i].p, 0, tab_bytes + 4 * sizeof(unsigned long long), st));
  }
  ss->fused = nk > 1 && sketch_reads_multi_supported(ss->ks, nk) && !dbg("no_fused");
  *out = ss.release();
  return MG_OK;
}

static int stream_add(mg_sketch_stream* ss, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, uint64_t nbases) {
  MG_REQUIRE_READY();
  if (!ss) return fail(MG_ERR_ARG, "null stream");
  if (nreads == 0) return MG_OK;
  if (!d_bases || !d_offsets) return fail(MG_ERR_ARG, "null device input");
  if (nbases == 0) {  // not told: read the offsets' ends back (one small sync)
    ReadPlan rp;
    MG_TRY(plan_reads(d_offsets, nreads, ctx().stream, rp));
    nbases = rp.nbases;
  }
  if (ss->count_kc) {
    MG_TRY(mg_count_kmers_dev(d_bases, d_offsets, nreads, nbases, ss->count_db, ss->count_kc));
    ss->nreads += nreads;
    ss->nbases += nbases;
    return MG_OK;
  }
  const unsigned stage = stage_bytes_for(nbases, nreads);
  if (ss->fused) {
    MultiKTable tabs[4];
    for (int i = 0; i < ss->nk; ++i)
      tabs[i] = MultiKTable{ss->hmax[i], ss->kp[i].tp.tab, ss->t_counters[i], ss->kp[i].tp.shift, ss->filters[i]};
    MG_TRY(launch_sketch_reads_multi(ss->ks, ss->nk, d_bases, d_offsets, nreads, tabs, stage));
  } else {
    for (int i = 0; i < ss->nk; ++i) {
      int rc = MG_ERR_ARG;
      dispatch_k(ss->ks[i], [&]<int K>() {
        rc = launch_sketch_reads<K>(d_bases, d_offsets, nreads, ss->hmax[i], nullptr, 0, ss->t_counters[i], ss->kp[i].tp.tab,
                                    ss->kp[i].tp.shift, stage, ss->filters[i]);
      });
      if (rc) return rc;
    }
  }
  ss->nreads += nreads;
  ss->nbases += nbases;
  return MG_OK;
}

static int stream_finish(mg_sketch_stream* ss, mg_sketch** out) {
  MG_REQUIRE_READY();
  if (!ss || !out) return fail(MG_ERR_ARG, "null argument");
  for (int i = 0; i < ss->nk; ++i) out[i] = nullptr;
  ReadPlan rp;
  rp.nbases = ss->nbases;
  rp.stage = stage_bytes_for(ss->nbases, ss->nreads);
  for (int i = 0; i < ss->nk; ++i) {
    std::unique_ptr<mg_sketch> sk(new mg_sketch());
    KPlan& kp = ss->kp[i];
    // (the hint update at resolution divides the distinct hashes by the candidates EXPECTED: of what was streamed)
    const double frac = hash_fraction(kp.hmax);
    kp.expect = (uint64_t)((double)ss->nbases * frac);
    int rc = alloc_table_staging(kp.tp);
    if (rc == MG_OK) rc = finish_pending(sk.get(), kp, rp, ss->t_counters[i], ss->ks[i], ss->s, nullptr, nullptr, 0, ss->filters[i]);
    if (rc != MG_OK) {
      for (int j = 0; j < i; ++j) { mg_sketch_free(out[j]); out[j] = nullptr; }
      return rc;
    }
    out[i] = sk.release();
  }
  return MG_OK;
}

extern "C" {

int mg_sketch_stream_begin(int nk, const int* ks, const uint64_t* hmaxs, uint64_t s, const mg_filter* const* filters,
                           uint64_t expect_bases, mg_sketch_stream** out) {
  return stream_begin(nk, ks, hmaxs, s, filters, expect_bases, out);
}
int mg_sketch_stream_begin_counts(const mg_refdb* db, mg_kcounts* kc, mg_sketch_stream** out) {
  MG_REQUIRE_READY();
  if (!db || !kc || !out) return fail(MG_ERR_ARG, "null argument");
  std::unique_ptr<mg_sketch_stream> ss(new mg_sketch_stream());
  ss->count_db = db;
  ss->count_kc = kc;
  *out = ss.release();
  return MG_OK;
}
int mg_sketch_stream_add_dev(mg_sketch_stream* ss, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads,
                             uint64_t nbases) {
  return stream_add(ss, d_bases, d_offsets, nreads, nbases);
}
int mg_sketch_stream_finish(mg_sketch_stream* ss, mg_sketch** out) { return stream_finish(ss, out); }
uint64_t mg_sketch_stream_nreads(const mg_sketch_stream* ss) { return ss ? ss->nreads : 0; }
uint64_t mg_sketch_stream_nbases(const mg_sketch_stream* ss) { return ss ? ss->nbases : 0; }
void mg_sketch_stream_free(mg_sketch_stream* ss) { delete ss; }

int mg_sketch_reads_multi_dev_async(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, int nk, const int* ks,
                                    const uint64_t* hmaxs, uint64_t s, const mg_filter* const* filters, mg_sketch** out) {
  for (int i = 0; ks && i < nk; ++i)
    if (ks[i] < 1 || ks[i] > MG_MAX_K) return fail(MG_ERR_ARG, "k=%d outside [1,%d]", ks[i], MG_MAX_K);
  return sketch_reads_multi_async(d_bases, d_offsets, nreads, nk, ks, hmaxs, s, filters, out);
}

int mg_sketch_reads_dev_async(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, int k, uint64_t hmax,
                              uint64_t s, mg_sketch** out) {
  return sketch_reads_async(d_bases, d_offsets, nreads, k, hmax, s, nullptr, out);
}

int mg_sketch_reads_filtered_dev_async(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, int k,
                                       uint64_t hmax, uint64_t s, const mg_filter* filter, mg_sketch** out) {
  if (!filter) return fail(MG_ERR_ARG, "null filter");
  return sketch_reads_async(d_bases, d_offsets, nreads, k, hmax, s, filter, out);
}

int mg_sketch_reads_filtered_dev(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, int k, uint64_t hmax,
                                 uint64_t s, const mg_filter* filter, mg_sketch** out) {
  MG_TRY(mg_sketch_reads_filtered_dev_async(d_bases, d_offsets, nreads, k, hmax, s, filter, out));
  int rc = sketch_resolve(*out, nullptr);
  if (rc != MG_OK) { mg_sketch_free(*out); *out = nullptr; }
  return rc;
}

int mg_filter_build(const uint64_t* hashes, uint64_t n, mg_filter** out) {
  MG_REQUIRE_READY();
  if (!out || (n && !hashes)) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  std::unique_ptr<mg_filter> f(new mg_filter());
  unsigned lb = 16;  // 16 bits per hash, between 2^16 and 2^30 bits (128 MB: still inside the Infinity Cache)
  while (lb < 30 && (1ull << lb) < 16 * n) ++lb;
  f->log2_bits = lb;
  f->mask = (1ull << lb) - 1;
  const uint64_t bytes = (1ull << lb) / 8;
  MG_TRY(f->bits.alloc(bytes));
  hipStream_t st = ctx().stream;
  MG_HIP(hipMemsetAsync(f->bits.p, 0, bytes, st));
  if (n) {
    uint64_t* d_h = (uint64_t*)scratch("filter_h", n * sizeof(uint64_t));
    if (!d_h) return MG_ERR_NOMEM;
    MG_HIP(hipMemcpyAsync(d_h, hashes, n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_filter_set, dim3(grid_for(n, 256, (unsigned)ctx().num_cus * 8)), dim3(256), 0, st, d_h, n,
                       f->bits.as<uint32_t>(), f->mask);
    MG_HIP(hipGetLastError());
  }
  MG_HIP(hipStreamSynchronize(st));
  *out = f.release();
  return MG_OK;
}

// The filter's resident index (mg_internal.h: mg_filter::Resident): every one of the n hashes (duplicates welcome; all
// <= hmax) seeded into a counting table of buckets of <= kBucketTarget >> spread expected hashes over [0, hmax].  A table whose
// hashes crowd some range — a bucket without a free slot — gets MG_ERR_CAPACITY and stays a bit filter.
int mg_filter_make_resident(mg_filter* f, const uint64_t* hashes, uint64_t n, uint64_t hmax, unsigned spread) {
  MG_REQUIRE_READY();
  if (!f || (n && !hashes)) return fail(MG_ERR_ARG, "null argument");
  if (hmax == kReservedHash) hmax = kReservedHash - 1;
  f->resident.reset();
  TablePlan tp;
  if (spread > 3) return fail(MG_ERR_ARG, "resident index: spread %u outside [0,3]", spread);
  // (spread s: buckets planned for 2^s times the hashes — at s = 1 half as many hashes live away from their home slot and
  // half as many candidates go round again, 28.5 against 29.7 ms per 12.5M reads at 200k genomes, for twice the memory)
  if (n == 0 || !plan_table(0, hmax, (double)n * (double)(1u << spread), tp)) return fail(MG_ERR_ARG, "no resident index for %llu hashes up to %llu", (unsigned long long)n, (unsigned long long)hmax);
  if (tp.slots >= (1ull << 32))  // (a pass lists the slots it touches as 32-bit numbers)
    return fail(MG_ERR_CAPACITY, "resident index: %llu slots do not fit 32-bit slot numbers", (unsigned long long)tp.slots);
  std::unique_ptr<mg_filter::Resident> R(new mg_filter::Resident());
  R->shift = tp.shift; R->nbuckets = tp.nbuckets; R->slots = tp.slots; R->hmax = hmax;
  mg_filter::Resident::Copy cp;
  if (hipMalloc(&cp.slots, tp.slots * sizeof(Slot)) != hipSuccess) {
    pool_release_all();
    if (hipMalloc(&cp.slots, tp.slots * sizeof(Slot)) != hipSuccess)
      return fail(MG_ERR_NOMEM, "resident index: hipMalloc(%llu) failed", (unsigned long long)(tp.slots * sizeof(Slot)));
  }
  R->copies.push_back(cp);
  hipStream_t st = ctx().stream;
  MG_HIP(hipMemsetAsync(cp.slots, 0, tp.slots * sizeof(Slot), st));
  unsigned long long* d_counters = (unsigned long long*)scratch("sk_counters", 8 * sizeof(unsigned long long));
  if (!d_counters) return MG_ERR_NOMEM;
  MG_HIP(hipMemsetAsync(d_counters, 0, 2 * sizeof(unsigned long long), st));
  const uint64_t chunk = 1ull << 26;  // 512 MB of hashes at a time
  uint64_t* d_h = (uint64_t*)scratch("filter_h", (n < chunk ? n : chunk) * sizeof(uint64_t));
  if (!d_h) return MG_ERR_NOMEM;
  for (uint64_t at = 0; at < n; at += chunk) {
    const uint64_t m = n - at < chunk ? n - at : chunk;
    MG_HIP(hipMemcpyAsync(d_h, hashes + at, m * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_index_seed, dim3(grid_for(m, 256, (unsigned)ctx().num_cus * 8)), dim3(256), 0, st, d_h, m, tp.shift,
                       tp.nbuckets, reinterpret_cast<Slot*>(cp.slots), d_counters);
    MG_HIP(hipGetLastError());
    MG_HIP(hipStreamSynchronize(st));  // (pageable source: the copy is synchronous anyway; d_h is reused)
  }
  uint64_t* pin = host_words();
  MG_HIP(hipMemcpyAsync(pin, d_counters, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  R->distinct = pin[0];
  if (pin[1])
    return fail(MG_ERR_CAPACITY, "resident index: %llu of %llu hashes found no slot (above hmax, or a crowded hash range)",
                (unsigned long long)pin[1], (unsigned long long)n);
  f->resident = std::move(R);
  return MG_OK;
}

// The filter back to a bit filter: its resident index (every copy) is freed.
int mg_filter_use_resident(mg_filter* f, int on) {
  if (!f) return fail(MG_ERR_ARG, "null filter");
  f->resident_off = !on;
  return MG_OK;
}

int mg_filter_drop_resident(mg_filter* f) {
  MG_REQUIRE_READY();
  if (!f) return fail(MG_ERR_ARG, "null argument");
  if (f->resident) MG_HIP(hipDeviceSynchronize());  // (sketches in flight may still be counting in it)
  f->resident.reset();
  return MG_OK;
}

// Bytes of HBM the filter's resident index holds (every copy), 0 without one.
uint64_t mg_filter_resident_bytes(const mg_filter* f) {
  return f && f->resident ? f->resident->slots * sizeof(mg::Slot) * f->resident->copies.size() : 0;
}

int mg_set_count_saturation(uint32_t cs) {
  MG_REQUIRE_READY();
  if (cs > mg::kCsMask) return fail(MG_ERR_ARG, "count saturation %u above %u", cs, mg::kCsMask);
  ctx().count_sat = cs;
  return MG_OK;
}
uint32_t mg_count_saturation(void) { return ctx().count_sat; }

int mg_set_hash_mode(int mode) {
  MG_REQUIRE_READY();
  if (mode != mg::kHashCanonical && mode != mg::kHashCmash) return fail(MG_ERR_ARG, "hash mode %d: 0 (canonical k-mer) or 1 (CMash recollection)", mode);
  ctx().hash_mode = mode;
  return MG_OK;
}
int mg_hash_mode(void) { return ctx().hash_mode; }

int mg_filter_download(const mg_filter* f, uint32_t* bits, uint64_t nbytes) {
  MG_REQUIRE_READY();
  if (!f || !bits) return fail(MG_ERR_ARG, "null argument");
  if (nbytes != (1ull << f->log2_bits) / 8) return fail(MG_ERR_CAPACITY, "the filter holds %llu bytes", (unsigned long long)((1ull << f->log2_bits) / 8));
  return mg_memcpy_d2h(bits, f->bits.p, nbytes);
}

int mg_filter_from_bits(const uint32_t* bits, unsigned log2_bits, mg_filter** out) {
  MG_REQUIRE_READY();
  if (!bits || !out) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  if (log2_bits < 16 || log2_bits > 30) return fail(MG_ERR_ARG, "filter size 2^%u outside [2^16, 2^30] bits", log2_bits);
  std::unique_ptr<mg_filter> f(new mg_filter());
  f->log2_bits = log2_bits;
  f->mask = (1ull << log2_bits) - 1;
  const uint64_t bytes = (1ull << log2_bits) / 8;
  MG_TRY(f->bits.alloc(bytes));
  MG_TRY(mg_memcpy_h2d(f->bits.p, bits, bytes));
  *out = f.release();
  return MG_OK;
}

unsigned mg_filter_log2_bits(const mg_filter* f) { return f ? f->log2_bits : 0; }
void mg_filter_free(mg_filter* f) { delete f; }

int mg_sketch_reads_dev(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, int k, uint64_t hmax,
                        uint64_t s, mg_sketch** out) {
  MG_TRY(mg_sketch_reads_dev_async(d_bases, d_offsets, nreads, k, hmax, s, out));
  int rc = sketch_resolve(*out, nullptr);
  if (rc != MG_OK) { mg_sketch_free(*out); *out = nullptr; }
  return rc;
}

// General merge: sort the pairs by hash (rocPRIM) and add the counts of equal hashes.
static int merge_via_sort(mg_sketch* sk, const uint64_t* d_hashes, const uint32_t* d_counts, uint64_t n, uint64_t s,
                          int any_truncated, uint64_t bound) {
  uint64_t* d_ks = (uint64_t*)scratch("mp_keys", (n + 1) * sizeof(uint64_t));
  uint32_t* d_vs = (uint32_t*)scratch("mp_vals", (n + 1) * sizeof(uint32_t));
  uint64_t* d_meta = (uint64_t*)scratch("mp_meta", 8 * sizeof(uint64_t));
  if (!d_ks || !d_vs || !d_meta) return MG_ERR_NOMEM;
  MG_TRY(sk->hashes.alloc((n + 1) * sizeof(uint64_t)));
  MG_TRY(sk->counts.alloc((n + 1) * sizeof(uint32_t)));
  {
    ProfScope ps("merge_sort");
    MG_TRY(sort_pairs(d_hashes, d_ks, d_counts, d_vs, n));
    MG_TRY(reduce_pairs(d_ks, d_vs, n, sk->hashes.as<uint64_t>(), sk->counts.as<uint32_t>(), d_meta));
  }
  return adopt_runs(sk, d_meta, s, any_truncated != 0, bound);
}

int mg_sketch_merge_dev(const uint64_t* d_hashes, const uint32_t* d_counts, uint64_t n, uint64_t range_lo,
                        uint64_t range_hi, uint64_t s, int any_truncated, uint64_t bound, mg_sketch** out) {
  MG_REQUIRE_READY();
  if (!out) return fail(MG_ERR_ARG, "null out handle");
  *out = nullptr;
  if (n > 0 && (!d_hashes || !d_counts)) return fail(MG_ERR_ARG, "null device input");
  std::unique_ptr<mg_sketch> sk(new mg_sketch());
  hipStream_t st = ctx().stream;
  TablePlan tp;
  if (n >= 32768 && range_hi >= range_lo && !dbg("force_list") && plan_table(range_lo, range_hi, (double)n, tp)) {
    unsigned long long* d_counters = (unsigned long long*)scratch("mp_meta", 8 * sizeof(uint64_t));
    if (!d_counters) return MG_ERR_NOMEM;
    uint64_t* d_meta = reinterpret_cast<uint64_t*>(d_counters) + 4;
    MG_HIP(hipMemsetAsync(d_counters, 0, 4 * sizeof(unsigned long long), st));
    MG_TRY(alloc_table(tp));
    {
      ProfScope ps("merge_insert");
      hipLaunchKernelGGL(k_table_insert_pairs, dim3(grid_for(n, 256, (unsigned)ctx().num_cus * 8)), dim3(256), 0, st,
                         d_hashes, d_counts, n, tp.lo, tp.shift, tp.nbuckets, tp.tab, d_counters, ctx().count_sat);
      MG_HIP(hipGetLastError());
    }
    uint64_t h_counters[3] = {0, 0, 0};
    MG_TRY(table_to_sketch(tp, sk.get(), d_meta, s, any_truncated != 0, bound, d_counters, h_counters, n));  // a union of n pairs has <= n entries
    if (h_counters[2] == 0) {
      *out = sk.release();
      return MG_OK;
    }
    sk.reset(new mg_sketch());  // a pair fell outside the declared range: general path
  }
  MG_TRY(merge_via_sort(sk.get(), d_hashes, d_counts, n, s, any_truncated, bound));
  *out = sk.release();
  return MG_OK;
}

// The same merge without a host synchronisation: the table path is queued and the sketch is handed back pending
// (its size / last hash / overflow counter land in pinned words behind an event), exactly like a deferred read
// sketch.  The inputs must stay alive until mg_sketch_resolve (they are what a redo reads).  Inputs the table path
// does not take (few pairs, no declared range) are merged synchronously.
int mg_sketch_merge_dev_async(const uint64_t* d_hashes, const uint32_t* d_counts, uint64_t n, uint64_t range_lo,
                              uint64_t range_hi, uint64_t s, int any_truncated, uint64_t bound, mg_sketch** out) {
  MG_REQUIRE_READY();
  if (!out) return fail(MG_ERR_ARG, "null out handle");
  *out = nullptr;
  if (n > 0 && (!d_hashes || !d_counts)) return fail(MG_ERR_ARG, "null device input");
  TablePlan tp;
  if (!(n >= 32768 && range_hi >= range_lo && !dbg("force_list") && plan_table(range_lo, range_hi, (double)n, tp)))
    return mg_sketch_merge_dev(d_hashes, d_counts, n, range_lo, range_hi, s, any_truncated, bound, out);
  std::unique_ptr<mg_sketch> sk(new mg_sketch());
  Context& cc = ctx();
  hipStream_t st = cc.stream;
  unsigned long long* t_counters = nullptr;  // cleared together with the table
  MG_TRY(alloc_table(tp, &t_counters));
  {
    ProfScope ps("merge_insert");
    hipLaunchKernelGGL(k_table_insert_pairs, dim3(grid_for(n, 256, (unsigned)cc.num_cus * 8)), dim3(256), 0, st, d_hashes,
                       d_counts, n, tp.lo, tp.shift, tp.nbuckets, tp.tab, t_counters, cc.count_sat);
    MG_HIP(hipGetLastError());
  }
  const unsigned slot = cc.pend_next++ % Context::kPendSlots;
  if (cc.pend_owner[slot]) MG_TRY(sketch_resolve(cc.pend_owner[slot], nullptr));  // ring full: settle the oldest
  MG_TRY(sk->meta.alloc(8 * sizeof(uint64_t)));
  uint64_t* sk_meta = sk->meta.as<uint64_t>();
  MG_TRY(table_pack(tp, sk.get(), sk_meta, n));  // (the union has at most n entries: no capacity check needed)
  sk->h_meta = cc.pend_pinned + 8 * slot;
  hipLaunchKernelGGL(k_sketch_meta, dim3(1), dim3(64), 0, st, sk->hashes.as<uint64_t>(), sk_meta, s,
                     (uint32_t)(any_truncated ? 1 : 0), bound, (const unsigned long long*)t_counters, sk->h_meta, (uint64_t)0);
  MG_HIP(hipGetLastError());
  sk->ev = take_event();
  if (sk->ev) { MG_HIP(hipEventRecord(sk->ev, st)); sk->ev_stream = st; }
  sk->pending = true;
  sk->pend_slot = (int)slot;
  cc.pend_owner[slot] = sk.get();
  sk->n_bound = n < tp.slots ? n : tp.slots;
  if (s > 0 && s < sk->n_bound) sk->n_bound = s;
  sk->hmax = range_hi;
  sk->expect = (double)(n ? n : 1);
  sk->redo.is_merge = true;
  sk->redo.use_bound = any_truncated != 0;
  sk->redo.s = s;
  sk->redo.m_hashes = d_hashes; sk->redo.m_counts = d_counts; sk->redo.m_n = n; sk->redo.m_bound = bound;
  *out = sk.release();
  return MG_OK;
}

int mg_sketch_from_pairs_dev(const uint64_t* d_hashes, const uint32_t* d_counts, uint64_t n, uint64_t s,
                             int any_truncated, uint64_t bound, mg_sketch** out) {
  // no declared range: range_hi < range_lo selects the general (sorting) merge
  return mg_sketch_merge_dev(d_hashes, d_counts, n, 1, 0, s, any_truncated, bound, out);
}

int mg_sketch_split(const mg_sketch* sk, const uint64_t* bounds, uint32_t nbounds, uint64_t* out_idx) {
  MG_REQUIRE_READY();
  if (!sk || !bounds || !out_idx) return fail(MG_ERR_ARG, "null argument");
  MG_TRY(sketch_resolve(const_cast<mg_sketch*>(sk), nullptr));
  MG_TRY(sketch_wait(sk));
  if (nbounds == 0) return MG_OK;
  if (nbounds > 4096) return fail(MG_ERR_ARG, "too many slice bounds");
  hipStream_t st = ctx().stream;
  uint64_t* d_io = (uint64_t*)scratch("sk_split", 2 * (uint64_t)nbounds * sizeof(uint64_t));
  if (!d_io) return MG_ERR_NOMEM;
  MG_HIP(hipMemcpyAsync(d_io, bounds, nbounds * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_sketch_split, dim3((nbounds + 63) / 64), dim3(64), 0, st, sk->hashes.as<uint64_t>(), sk->n, d_io,
                     nbounds, d_io + nbounds);
  MG_HIP(hipGetLastError());
  MG_HIP(hipMemcpyAsync(out_idx, d_io + nbounds, nbounds * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  return MG_OK;
}

// Slice sizes of a sketch cut at ascending hash bounds, and its size / last hash / truncation / table-overflow
// count, written to device words — also for a sketch whose finalisation is still deferred (nothing is synchronised):
// out[0 .. nbounds] = entries per slice, then truncated, last hash, n, overflows.
__global__ __launch_bounds__(256) void k_slice_words(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ meta,
                                                     uint64_t n_host, uint64_t last_host, uint32_t trunc_host,
                                                     const uint64_t* __restrict__ bounds, uint32_t nbounds,
                                                     long long* __restrict__ out) {
  __shared__ uint64_t cuts[4098];
  const uint64_t n = meta ? meta[1] : n_host;
  for (uint32_t i = threadIdx.x; i < nbounds; i += blockDim.x) {
    const uint64_t key = bounds[i];
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (hashes[mid] < key) lo = mid + 1; else hi = mid;
    }
    cuts[i + 1] = lo;
  }
  if (threadIdx.x == 0) { cuts[0] = 0; cuts[nbounds + 1] = n; }
  __syncthreads();
  for (uint32_t q = threadIdx.x; q <= nbounds; q += blockDim.x) out[q] = (long long)(cuts[q + 1] - cuts[q]);
  if (threadIdx.x == 0) {
    out[nbounds + 1] = meta ? (long long)meta[3] : (long long)trunc_host;
    out[nbounds + 2] = (long long)(meta ? meta[2] : last_host);
    out[nbounds + 3] = (long long)n;
    out[nbounds + 4] = meta ? (long long)meta[6] : 0;
  }
}

int mg_sketch_slice_words_dev(const mg_sketch* sk, const uint64_t* d_bounds, uint32_t nbounds, int64_t* d_out) {
  MG_REQUIRE_READY();
  if (!sk || !d_out || (nbounds && !d_bounds)) return fail(MG_ERR_ARG, "null argument");
  if (nbounds > 4096) return fail(MG_ERR_ARG, "too many slice bounds");
  MG_TRY(sketch_wait(sk));  // built on another stream: this one waits for it on the device
  hipLaunchKernelGGL(k_slice_words, dim3(1), dim3(256), 0, ctx().stream, sk->hashes.as<uint64_t>(),
                     sk->pending ? sk->meta.as<uint64_t>() : (const uint64_t*)nullptr, sk->n, sk->n ? sk->last_hash : 0,
                     (uint32_t)sk->truncated, d_bounds, nbounds, reinterpret_cast<long long*>(d_out));
  MG_HIP(hipGetLastError());
  return MG_OK;
}

int mg_sketch_set_bound(mg_sketch* sk, int truncated, uint64_t bound) {
  if (!sk) return fail(MG_ERR_ARG, "null sketch");
  MG_TRY(sketch_resolve(sk, nullptr));
  sk->has_bound = true;
  sk->truncated = truncated ? 1 : 0;
  sk->bound = bound;
  return MG_OK;
}

static const mg_sketch* settled(const mg_sketch* sk) {  // accessors see a finalised sketch
  if (sk && sk->pending) (void)sketch_resolve(const_cast<mg_sketch*>(sk), nullptr);
  return sk;
}
uint64_t mg_sketch_size(const mg_sketch* sk) { return settled(sk) ? sk->n : 0; }
int mg_sketch_truncated(const mg_sketch* sk) { return settled(sk) ? sk->truncated : 0; }
uint64_t mg_sketch_last_hash(const mg_sketch* sk) { return (settled(sk) && sk->n) ? sk->last_hash : 0; }
uint64_t mg_sketch_kmers_seen(const mg_sketch* sk) { return settled(sk) ? sk->kmers_seen : 0; }
int mg_sketch_resolve(mg_sketch* sk, int* rebuilt) {
  MG_REQUIRE_READY();
  if (!sk) return fail(MG_ERR_ARG, "null sketch");
  return sketch_resolve(sk, rebuilt);
}

int mg_sketch_device_ptrs(const mg_sketch* sk, const uint64_t** d_hashes, const uint32_t** d_counts) {
  if (!sk) return fail(MG_ERR_ARG, "null sketch");
  MG_TRY(sketch_resolve(const_cast<mg_sketch*>(sk), nullptr));  // a rebuild would move the buffers
  MG_TRY(sketch_wait(sk));
  if (d_hashes) *d_hashes = sk->hashes.as<uint64_t>();
  if (d_counts) *d_counts = sk->counts.as<uint32_t>();
  return MG_OK;
}

int mg_sketch_download(const mg_sketch* sk, uint64_t* hashes, uint32_t* counts, uint64_t cap) {
  MG_REQUIRE_READY();
  if (!sk) return fail(MG_ERR_ARG, "null sketch");
  MG_TRY(sketch_resolve(const_cast<mg_sketch*>(sk), nullptr));
  MG_TRY(sketch_wait(sk));
  if (cap < sk->n) return fail(MG_ERR_CAPACITY, "sketch has %llu entries, buffer holds %llu", (unsigned long long)sk->n,
                               (unsigned long long)cap);
  if (sk->n == 0) return MG_OK;
  hipStream_t st = ctx().stream;
  if (hashes) MG_HIP(hipMemcpyAsync(hashes, sk->hashes.p, sk->n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  if (counts) MG_HIP(hipMemcpyAsync(counts, sk->counts.p, sk->n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  return MG_OK;
}

void mg_sketch_free(mg_sketch* sk) { delete sk; }

int mg_sketch_reads(const uint8_t* bases, const uint64_t* offsets, uint64_t nreads, int k, uint64_t hmax, uint64_t s,
                    uint64_t* out_hashes, uint32_t* out_counts, uint64_t out_cap, uint64_t* out_n, int* out_truncated,
                    uint64_t* out_kmers_seen) {
  MG_REQUIRE_READY();
  if (!offsets || !out_n) return fail(MG_ERR_ARG, "null argument");
  const uint64_t nbases = offsets[nreads];
  DevBuf d_bases, d_offsets;
  MG_TRY(d_bases.alloc(nbases + 16));
  MG_TRY(d_offsets.alloc((nreads + 1) * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_h2d(d_bases.p, bases, nbases));
  MG_TRY(mg_memcpy_h2d(d_offsets.p, offsets, (nreads + 1) * sizeof(uint64_t)));
  mg_sketch* sk = nullptr;
  MG_TRY(mg_sketch_reads_dev(d_bases.as<uint8_t>(), d_offsets.as<uint64_t>(), nreads, k, hmax, s, &sk));
  *out_n = sk->n;
  if (out_truncated) *out_truncated = sk->truncated;
  if (out_kmers_seen) *out_kmers_seen = sk->kmers_seen;
  int rc = mg_sketch_download(sk, out_hashes, out_counts, out_cap);
  mg_sketch_free(sk);
  return rc;
}

// ---- the k < k_max tables of hash mode 1: k-prefixes of the sketched k_max-mers (oracle: mgo_sketch_genomes_prefix) ----
}  // extern "C"

namespace mg {

constexpr uint64_t kTagBit = 1ull << 63;

// the tagged position hashes without their tag (what the sketch is made of); the reserved value stays
__global__ void k_clear_tags(const uint64_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i] == kReservedHash ? kReservedHash : (in[i] & ~kTagBit);
}

__global__ void k_fill_u64(uint64_t* __restrict__ v, uint64_t n, uint64_t value) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) v[i] = value;
}

// Every position p that ends a valid k_max-mer whose hash is in its genome's sketch: the key of that k_max-mer's k-prefix
// (in the kept strand's orientation) = the mode-1 hash of the k-mer ending at p - kmax + k (kept strand = forward) or at p
// (kept strand = reverse complement: the prefix of the reverse complement is the reverse complement of the SUFFIX, and the
// hash is symmetric in the strand).  keys[g * n + slot] = the smallest key among the k_max-mers that share the slot's hash.
__global__ __launch_bounds__(256) void k_prefix_keys(const uint64_t* __restrict__ tagged, const uint64_t* __restrict__ hk,
                                                     const uint64_t* __restrict__ offsets, uint64_t nseq, uint64_t nbases,
                                                     const uint64_t* __restrict__ sketch, const uint32_t* __restrict__ cnt, uint64_t n,
                                                     int kmax, int k, unsigned long long* __restrict__ keys) {
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; p < nbases; p += stride) {
    const uint64_t t = tagged[p];
    if (t == kReservedHash) continue;
    const uint64_t h = t & ~kTagBit;
    uint64_t lo = 0, hi = nseq;  // genome of p: offsets[lo] <= p < offsets[hi]
    while (hi - lo > 1) {
      const uint64_t mid = (lo + hi) >> 1;
      if (offsets[mid] <= p) lo = mid; else hi = mid;
    }
    const uint64_t g = lo;
    const uint64_t* sk = sketch + g * n;
    uint32_t a = 0, b = cnt[g];
    while (a < b) {
      const uint32_t mid = (a + b) >> 1;
      if (sk[mid] < h) a = mid + 1; else b = mid;
    }
    if (a >= cnt[g] || sk[a] != h) continue;
    const uint64_t e = (t & kTagBit) ? p : p - (uint64_t)kmax + (uint64_t)k;
    atomicMin(&keys[g * n + a], (unsigned long long)hk[e]);
  }
}

// ---- stage A' with the k-mers kept (mg_sketch_genomes_kmers; oracle: mgo_sketch_genomes_kmers) ----
// first[g * n + slot] = the smallest position p at which a k-mer with the slot's hash ENDS (CountEstimator.add keeps the first
// k-mer of a hash).  pos[p]: the position hashes (mode 1: tagged with the kept strand in bit 63).
__global__ __launch_bounds__(256) void k_first_positions(const uint64_t* __restrict__ pos, int tagged, const uint64_t* __restrict__ offsets,
                                                         uint64_t nseq, uint64_t nbases, const uint64_t* __restrict__ sketch,
                                                         const uint32_t* __restrict__ cnt, uint64_t n, unsigned long long* __restrict__ first) {
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; p < nbases; p += stride) {
    const uint64_t t = pos[p];
    if (t == kReservedHash) continue;
    const uint64_t h = tagged ? (t & ~kTagBit) : t;
    uint64_t lo = 0, hi = nseq;  // genome of p: offsets[lo] <= p < offsets[hi]
    while (hi - lo > 1) {
      const uint64_t mid = (lo + hi) >> 1;
      if (offsets[mid] <= p) lo = mid; else hi = mid;
    }
    const uint64_t g = lo;
    const uint64_t* sk = sketch + g * n;
    uint32_t a = 0, b = cnt[g];
    while (a < b) {
      const uint32_t mid = (a + b) >> 1;
      if (sk[mid] < h) a = mid + 1; else b = mid;
    }
    if (a >= cnt[g] || sk[a] != h) continue;
    atomicMin(&first[g * n + a], (unsigned long long)p);
  }
}

// The kept k-mer of every sketch entry, 2-bit packed (first base most significant) into (khi, klo): the window that ends at
// first[entry], as the table keeps it — mode 0: the lexicographically smaller strand (KMC's canonical k-mer); mode 1 (tagged
// position hashes): the strand CountEstimator.add keeps, bit 63 of the position's hash = the reverse complement.
__global__ __launch_bounds__(256) void k_entry_kmers(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ pos, int tagged,
                                                     const unsigned long long* __restrict__ first, const uint32_t* __restrict__ cnt,
                                                     uint64_t nseq, uint64_t n, int k, uint64_t* __restrict__ khi, uint64_t* __restrict__ klo) {
  uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; e < nseq * n; e += stride) {
    const uint64_t g = e / n, slot = e - g * n;
    if (slot >= cnt[g]) continue;
    const uint64_t p = first[e];
    uint64_t f_hi = 0, f_lo = 0, r_hi = 0, r_lo = 0;
    for (int i = 0; i < k; ++i) {
      uint32_t c;
      decode_base(bases[p + 1 - (uint64_t)k + (uint64_t)i], c);
      f_hi = (f_hi << 2) | (f_lo >> 62);
      f_lo = (f_lo << 2) | c;
      const uint64_t cc = 3u - c;
      if (2 * i < 64) r_lo |= cc << (2 * i); else r_hi |= cc << (2 * i - 64);
    }
    bool use_rc;
    if (tagged == 2) use_rc = false;  // (a sketch selected by the forward hash keeps the k-mer as it stands in the genome)
    else if (tagged) use_rc = (pos[p] & kTagBit) != 0;
    else use_rc = r_hi < f_hi || (r_hi == f_hi && r_lo < f_lo);
    khi[e] = use_rc ? r_hi : f_hi;
    klo[e] = use_rc ? r_lo : f_lo;
  }
}

// A sketch selected by the forward hash: what its entries MATCH by — the hash of the window that ends at first[entry] under the
// mode in force (ident[p], untagged) — in the place of the hash that selected them.
__global__ __launch_bounds__(256) void k_entry_identity(const uint64_t* __restrict__ ident, const unsigned long long* __restrict__ first,
                                                        const uint32_t* __restrict__ cnt, uint64_t nseq, uint64_t n, uint64_t* __restrict__ out) {
  uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; e < nseq * n; e += stride) {
    const uint64_t g = e / n, slot = e - g * n;
    if (slot < cnt[g]) out[e] = ident[first[e]];
  }
}

}  // namespace mg

extern "C" {

int mg_sketch_genomes_prefix(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int kmax, int k, uint64_t n,
                             uint64_t* out_hashes, uint64_t* out_offsets) {
  MG_REQUIRE_READY();
  if (!offsets || !out_offsets) return fail(MG_ERR_ARG, "null argument");
  if (kmax < 1 || kmax > MG_MAX_K || k < 1 || k > kmax) return fail(MG_ERR_ARG, "need 1 <= k=%d <= kmax=%d <= %d", k, kmax, MG_MAX_K);
  if (n == 0) return fail(MG_ERR_ARG, "n must be positive");
  Context& c = ctx();
  hipStream_t st = c.stream;
  out_offsets[0] = 0;
  const uint64_t kBatchBases = 1ull << 27;
  uint64_t g0 = 0, written = 0;
  std::vector<uint64_t> h_slots, rel, seg;
  std::vector<uint32_t> h_cnt;
  while (g0 < ngenomes) {
    uint64_t g1 = g0 + 1;
    while (g1 < ngenomes && offsets[g1 + 1] - offsets[g0] <= kBatchBases && g1 - g0 < (1u << 20)) ++g1;
    const uint64_t ng = g1 - g0, nb = offsets[g1] - offsets[g0];
    if (nb > 0xffffffffull) return fail(MG_ERR_ARG, "single genome of %llu bases exceeds 2^32-1", (unsigned long long)nb);
    rel.resize(ng + 1);
    seg.resize(ng + 1);
    for (uint64_t i = 0; i <= ng; ++i) { rel[i] = offsets[g0 + i] - offsets[g0]; seg[i] = i * n; }
    uint8_t* d_bases = (uint8_t*)scratch("g_bases", nb + 16);
    uint64_t* d_off = (uint64_t*)scratch("g_off", (ng + 1) * sizeof(uint64_t));
    uint64_t* d_seg = (uint64_t*)scratch("gp_seg", (ng + 1) * sizeof(uint64_t));
    uint64_t* d_tag = (uint64_t*)scratch("gp_tag", (nb + 1) * sizeof(uint64_t));
    uint64_t* d_hk = (uint64_t*)scratch("gp_hk", (nb + 1) * sizeof(uint64_t));
    uint64_t* d_pos = (uint64_t*)scratch("g_pos", (nb + 1) * sizeof(uint64_t));
    uint64_t* d_sorted = (uint64_t*)scratch("g_sorted", (nb + 1) * sizeof(uint64_t));
    uint64_t* d_out = (uint64_t*)scratch("g_out", ng * n * sizeof(uint64_t));
    uint32_t* d_cnt = (uint32_t*)scratch("g_cnt", ng * sizeof(uint32_t));
    unsigned long long* d_keys = (unsigned long long*)scratch("gp_keys", ng * n * sizeof(uint64_t));
    uint64_t* d_keys_sorted = (uint64_t*)scratch("gp_keys_sorted", ng * n * sizeof(uint64_t));
    uint64_t* d_out2 = (uint64_t*)scratch("gp_out", ng * n * sizeof(uint64_t));
    uint32_t* d_cnt2 = (uint32_t*)scratch("gp_cnt", ng * sizeof(uint32_t));
    if (!d_bases || !d_off || !d_seg || !d_tag || !d_hk || !d_pos || !d_sorted || !d_out || !d_cnt || !d_keys || !d_keys_sorted ||
        !d_out2 || !d_cnt2)
      return MG_ERR_NOMEM;
    if (nb) MG_HIP(hipMemcpyAsync(d_bases, bases + offsets[g0], nb, hipMemcpyHostToDevice, st));
    MG_HIP(hipMemcpyAsync(d_off, rel.data(), (ng + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    MG_HIP(hipMemcpyAsync(d_seg, seg.data(), (ng + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    const unsigned g256 = grid_for(nb ? nb : 1, 256, (unsigned)c.num_cus * 8);
    if (nb) {
      ProfScope ps("hash_positions");
      const uint64_t nchunks = (nb + kChunk - 1) / kChunk;
      const unsigned grid = grid_for(nchunks, 256, (unsigned)c.num_cus * 8);
      MG_TRY(launch_hash_positions_cmash(kmax, grid, st, d_bases, d_off, ng, nb, d_tag, true));
      MG_TRY(launch_hash_positions_cmash(k, grid, st, d_bases, d_off, ng, nb, d_hk, false));
      hipLaunchKernelGGL(k_clear_tags, dim3(g256), dim3(256), 0, st, d_tag, nb, d_pos);
      MG_HIP(hipGetLastError());
    }
    MG_TRY(segmented_sort_keys(d_pos, d_sorted, nb, d_off, ng));
    hipLaunchKernelGGL(k_take_bottom_n, dim3(grid_for(ng, 1, (unsigned)c.num_cus * 8)), dim3(256), 0, st, d_sorted, d_off, ng, n,
                       d_out, d_cnt);
    hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(ng * n, 256, (unsigned)c.num_cus * 8)), dim3(256), 0, st,
                       reinterpret_cast<uint64_t*>(d_keys), ng * n, kReservedHash);
    if (nb)
      hipLaunchKernelGGL(k_prefix_keys, dim3(g256), dim3(256), 0, st, d_tag, d_hk, d_off, ng, nb, d_out, d_cnt, n, kmax, k, d_keys);
    MG_HIP(hipGetLastError());
    // per genome: its keys ascending, distinct (unfilled slots hold the reserved value, which k_take_bottom_n skips)
    MG_TRY(segmented_sort_keys(reinterpret_cast<uint64_t*>(d_keys), d_keys_sorted, ng * n, d_seg, ng));
    hipLaunchKernelGGL(k_take_bottom_n, dim3(grid_for(ng, 1, (unsigned)c.num_cus * 8)), dim3(256), 0, st, d_keys_sorted, d_seg, ng,
                       n, d_out2, d_cnt2);
    MG_HIP(hipGetLastError());
    h_slots.resize(ng * n);
    h_cnt.resize(ng);
    MG_HIP(hipMemcpyAsync(h_slots.data(), d_out2, ng * n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MG_HIP(hipMemcpyAsync(h_cnt.data(), d_cnt2, ng * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st));
    for (uint64_t i = 0; i < ng; ++i) {
      for (uint32_t j = 0; j < h_cnt[i]; ++j) out_hashes[written + j] = h_slots[i * n + j];
      written += h_cnt[i];
      out_offsets[g0 + i + 1] = written;
    }
    g0 = g1;
  }
  return MG_OK;
}

}  // extern "C"

// mg_sketch_genomes, and (out_khi / out_klo given) mg_sketch_genomes_kmers
// forward_select (with the k-mers only): the n smallest MurmurHash3(k-mer as it stands) mod the prime select a genome's entries, the
// k-mer is kept as it stands, and out_hashes = what it matches by (the mode in force) — neither ascending nor necessarily distinct
// within a genome (a genome may hold a k-mer and its reverse complement).  oracle: mgo_sketch_genomes_kmers_forward
static int sketch_genomes_impl(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int k, uint64_t n,
                               uint64_t* out_hashes, uint64_t* out_khi, uint64_t* out_klo, uint64_t* out_offsets, bool forward_select = false) {
  MG_REQUIRE_READY();
  if (!offsets || !out_offsets) return fail(MG_ERR_ARG, "null argument");
  if (k < 1 || k > MG_MAX_K) return fail(MG_ERR_ARG, "k=%d outside [1,%d]", k, MG_MAX_K);
  if (n == 0) return fail(MG_ERR_ARG, "n must be positive");
  Context& c = ctx();
  hipStream_t st = c.stream;
  out_offsets[0] = 0;
  const uint64_t kBatchBases = 1ull << 28;  // 2 GiB of position hashes per batch
  uint64_t g0 = 0, written = 0;
  std::vector<uint64_t> h_slots, h_hi, h_lo;
  std::vector<uint32_t> h_cnt;
  std::vector<uint64_t> rel;
  while (g0 < ngenomes) {
    uint64_t g1 = g0 + 1;
    while (g1 < ngenomes && offsets[g1 + 1] - offsets[g0] <= kBatchBases && g1 - g0 < (1u << 20)) ++g1;
    const uint64_t ng = g1 - g0, nb = offsets[g1] - offsets[g0];
    if (nb > 0xffffffffull) return fail(MG_ERR_ARG, "single genome of %llu bases exceeds 2^32-1", (unsigned long long)nb);
    rel.resize(ng + 1);
    for (uint64_t i = 0; i <= ng; ++i) rel[i] = offsets[g0 + i] - offsets[g0];
    uint8_t* d_bases = (uint8_t*)scratch("g_bases", nb + 16);
    uint64_t* d_off = (uint64_t*)scratch("g_off", (ng + 1) * sizeof(uint64_t));
    uint64_t* d_pos = (uint64_t*)scratch("g_pos", (nb + 1) * sizeof(uint64_t));
    uint64_t* d_sorted = (uint64_t*)scratch("g_sorted", (nb + 1) * sizeof(uint64_t));
    uint64_t* d_out = (uint64_t*)scratch("g_out", ng * n * sizeof(uint64_t));
    uint32_t* d_cnt = (uint32_t*)scratch("g_cnt", ng * sizeof(uint32_t));
    if (!d_bases || !d_off || !d_pos || !d_sorted || !d_out || !d_cnt) return MG_ERR_NOMEM;
    const bool kmers = out_khi != nullptr;
    const bool tagged = kmers && !forward_select && ctx().hash_mode == kHashCmash;  // the kept strand rides in bit 63 of the position hashes
    uint64_t* d_key = d_pos;  // what is sorted (the untagged hashes)
    unsigned long long* d_first = nullptr;
    uint64_t *d_khi = nullptr, *d_klo = nullptr;
    if (tagged) {
      d_key = (uint64_t*)scratch("gk_key", (nb + 1) * sizeof(uint64_t));
      if (!d_key) return MG_ERR_NOMEM;
    }
    if (kmers) {
      d_first = (unsigned long long*)scratch("gk_first", ng * n * sizeof(uint64_t));
      d_khi = (uint64_t*)scratch("gk_hi", ng * n * sizeof(uint64_t));
      d_klo = (uint64_t*)scratch("gk_lo", ng * n * sizeof(uint64_t));
      if (!d_first || !d_khi || !d_klo) return MG_ERR_NOMEM;
    }
    if (nb) MG_HIP(hipMemcpyAsync(d_bases, bases + offsets[g0], nb, hipMemcpyHostToDevice, st));
    MG_HIP(hipMemcpyAsync(d_off, rel.data(), (ng + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    const unsigned g256 = grid_for(nb ? nb : 1, 256, (unsigned)c.num_cus * 8);
    if (nb) {
      ProfScope ps("hash_positions");
      const uint64_t nchunks = (nb + kChunk - 1) / kChunk;
      unsigned grid = grid_for(nchunks, 256, (unsigned)c.num_cus * 8);
      int crc = MG_OK;
      bool ok = dispatch_k(k, [&]<int K>() {
        if (forward_select)
          crc = launch_hash_positions_forward(K, grid, st, d_bases, d_off, ng, nb, d_pos);
        else if (ctx().hash_mode == kHashCmash)  // (instantiated in mg_sketch_cmash.hip, for the k of its list)
          crc = launch_hash_positions_cmash(K, grid, st, d_bases, d_off, ng, nb, d_pos, tagged);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_hash_positions<K, kHashCanonical>), dim3(grid), dim3(256), 0, st, d_bases, d_off, ng, nb,
                           d_pos);
      });
      if (!ok) return fail(MG_ERR_ARG, "unsupported k=%d", k);
      MG_TRY(crc);
      if (tagged) hipLaunchKernelGGL(k_clear_tags, dim3(g256), dim3(256), 0, st, d_pos, nb, d_key);
      MG_HIP(hipGetLastError());
    }
    MG_TRY(segmented_sort_keys(d_key, d_sorted, nb, d_off, ng));
    {
      ProfScope ps("take_bottom_n");
      hipLaunchKernelGGL(k_take_bottom_n, dim3(grid_for(ng, 1, (unsigned)c.num_cus * 8)), dim3(256), 0, st, d_sorted,
                         d_off, ng, n, d_out, d_cnt);
      MG_HIP(hipGetLastError());
    }
    h_slots.resize(ng * n);
    h_cnt.resize(ng);
    if (kmers) {
      // per sketch entry the first window of the genome that has its hash, then that window as the table keeps it
      hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(ng * n, 256, (unsigned)c.num_cus * 8)), dim3(256), 0, st,
                         reinterpret_cast<uint64_t*>(d_first), ng * n, kReservedHash);
      if (nb) {
        hipLaunchKernelGGL(k_first_positions, dim3(g256), dim3(256), 0, st, d_pos, tagged ? 1 : 0, d_off, ng, nb, d_out, d_cnt, n, d_first);
        hipLaunchKernelGGL(k_entry_kmers, dim3(grid_for(ng * n, 256, (unsigned)c.num_cus * 8)), dim3(256), 0, st, d_bases, d_pos,
                           forward_select ? 2 : tagged ? 1 : 0, d_first, d_cnt, ng, n, k, d_khi, d_klo);
        if (forward_select) {  // the entries' matching identity: the position hashes of the mode in force (d_sorted is free again)
          const uint64_t nchunks = (nb + kChunk - 1) / kChunk;
          unsigned grid = grid_for(nchunks, 256, (unsigned)c.num_cus * 8);
          int crc = MG_OK;
          dispatch_k(k, [&]<int K>() {
            if (ctx().hash_mode == kHashCmash)
              crc = launch_hash_positions_cmash(K, grid, st, d_bases, d_off, ng, nb, d_sorted, false);
            else
              hipLaunchKernelGGL(HIP_KERNEL_NAME(k_hash_positions<K, kHashCanonical>), dim3(grid), dim3(256), 0, st, d_bases, d_off, ng, nb, d_sorted);
          });
          MG_TRY(crc);
          hipLaunchKernelGGL(k_entry_identity, dim3(grid_for(ng * n, 256, (unsigned)c.num_cus * 8)), dim3(256), 0, st, d_sorted, d_first, d_cnt, ng, n, d_out);
        }
      }
      MG_HIP(hipGetLastError());
      h_hi.resize(ng * n);
      h_lo.resize(ng * n);
      MG_HIP(hipMemcpyAsync(h_hi.data(), d_khi, ng * n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      MG_HIP(hipMemcpyAsync(h_lo.data(), d_klo, ng * n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    }
    MG_HIP(hipMemcpyAsync(h_slots.data(), d_out, ng * n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MG_HIP(hipMemcpyAsync(h_cnt.data(), d_cnt, ng * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st));
    for (uint64_t i = 0; i < ng; ++i) {
      for (uint32_t j = 0; j < h_cnt[i]; ++j) {
        out_hashes[written + j] = h_slots[i * n + j];
        if (kmers) { out_khi[written + j] = h_hi[i * n + j]; out_klo[written + j] = h_lo[i * n + j]; }
      }
      written += h_cnt[i];
      out_offsets[g0 + i + 1] = written;
    }
    g0 = g1;
  }
  return MG_OK;
}

extern "C" {

int mg_sketch_genomes(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int k, uint64_t n,
                      uint64_t* out_hashes, uint64_t* out_offsets) {
  return sketch_genomes_impl(bases, offsets, ngenomes, k, n, out_hashes, nullptr, nullptr, out_offsets);
}

int mg_sketch_genomes_kmers(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int k, uint64_t n,
                            uint64_t* out_hashes, uint64_t* out_kmer_hi, uint64_t* out_kmer_lo, uint64_t* out_offsets) {
  if (!out_kmer_hi || !out_kmer_lo) return fail(MG_ERR_ARG, "null argument");
  return sketch_genomes_impl(bases, offsets, ngenomes, k, n, out_hashes, out_kmer_hi, out_kmer_lo, out_offsets);
}

int mg_sketch_genomes_kmers_forward(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int k, uint64_t n,
                                    uint64_t* out_hashes, uint64_t* out_kmer_hi, uint64_t* out_kmer_lo, uint64_t* out_offsets) {
  if (!out_kmer_hi || !out_kmer_lo) return fail(MG_ERR_ARG, "null argument");
  return sketch_genomes_impl(bases, offsets, ngenomes, k, n, out_hashes, out_kmer_hi, out_kmer_lo, out_offsets, true);
}

}  // extern "C"
