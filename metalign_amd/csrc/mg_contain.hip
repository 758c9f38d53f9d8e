// mg_contain.hip — Stage B: containment of every genome sketch in the read sketch.
//
// K2: both operands are sorted sets.  The table is held HASH-MAJOR (built once at upload): every (hash, genome)
// pair of every genome sketch, sorted by hash.  One streaming pass (`k_contain_pairs`):
//   * a workgroup takes a tile of 2048 consecutive pairs; their hashes span a narrow range, so the part of the
//     read sketch that can match them is a short contiguous run, found with two look-ups in the sketch's bucket
//     index and copied to LDS with coalesced loads;
//   * every pair looks its hash up in that LDS run (binary search) and, when it is present with count >= ci,
//     adds one to its genome's hit counter — atomics only for the PRESENT pairs, a small fraction of a
//     metagenome's table.
// HBM traffic: the table once (12 B per pair) + the read sketch once.  The previous layout (ascending union +
// per-genome positions: a presence bitmap, then one gather per genome hash) took 3.9 ms at 200 k genomes because
// the 200 M bit gathers each pulled a 64-byte line out of the Infinity Cache; see DESIGN.md.
//
// Replaces: kmc_tools simple ... intersect (scripts/select_db.py:54-56) and the
// containment index of StreamingQueryDNADatabase.py (scripts/select_db.py:73-76).
#include <memory>

#include "mg_internal.h"

namespace mg {

// One k of a stage-B call (every k of a pass goes through ONE launch of each kernel: at 10k genomes a launch is 5-50 us
// of mostly latency, and a pass had nine of them).
constexpr int kMaxSmallK = 3;  // k below the largest in a reference-pipeline table (mg_refdb)
struct ContainK {
  // the read sketch and its bucket index
  const uint64_t* q;
  const uint32_t* qc;
  uint64_t qn, q_last;
  const uint64_t* meta;   // a sketch whose finalisation is deferred: [1] size, [2] last hash still on the device
  uint32_t* idx;
  unsigned shift;
  uint64_t idx_buckets;
  uint32_t build_index;   // 1: this call builds idx
  // the table
  const uint64_t* ph;
  const uint32_t* pg;
  const uint32_t* gsize;
  uint64_t npairs, ngenomes;
  uint64_t tile0;         // first tile of this k among the tiles of the call
  // counters
  uint32_t* hits_part;
  uint32_t* sizes_part;   // null unless the sketch is truncated
  uint32_t* hits;
  uint32_t* sizes;
  // the reference pipeline (k_contain_pairs<true>, mg_refpipe_*): per k below the largest, the prefix numbers of every pair
  // (kept strand / other strand, 0xffffffff = none) and the bitmap over D_k that a matched pair marks
  // stage A by k-mer identity (mg_kcount.hip; k_match_pairs): where a pair's k-mer is counted, and the counters
  const uint32_t* head;
  const uint32_t* counts;
  int nsmall;
  const uint32_t* pa[kMaxSmallK];
  const uint32_t* pb[kMaxSmallK];
  uint32_t* marks[kMaxSmallK];
};
constexpr int kMaxContainK = 4;
struct ContainArgs {
  ContainK k[kMaxContainK];
  int nk;
  uint32_t ci, copies;
  uint32_t cs;            // (k_match_pairs) counters saturate here, 0 = never
  uint64_t ntiles;        // of all k together
  uint32_t* zero;         // the hit-counter copies of the call, zeroed by the index kernel (one launch less)
  uint64_t nzero;
  uint32_t* zero2;        // (reference pipeline: the prefix bitmaps, zeroed in the same launch)
  uint64_t nzero2;
};

// idx[b] = first position whose hash >= b << shift; idx[nbuckets] = n.  One thread per sketch entry i (and one
// past the end) writes i into every bucket in (bucket(q[i-1]), bucket(q[i])]: hashes are uniform and there is
// about one bucket per eight entries, so that is a store for one thread in eight, against a 20-step binary search per bucket.
__global__ void k_build_index(const ContainArgs a) {
  const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t z = i0; z < a.nzero; z += stride) a.zero[z] = 0;
  for (uint64_t z = i0; z < a.nzero2; z += stride) a.zero2[z] = 0;
  for (int ki = 0; ki < a.nk; ++ki) {
    const ContainK& K = a.k[ki];
    if (!K.build_index) continue;
    const uint64_t* __restrict__ q = K.q;
    const uint64_t n = K.meta ? K.meta[1] : K.qn;
    for (uint64_t i = i0; i <= n; i += stride) {
      const uint64_t first = i == 0 ? 0 : (q[i - 1] >> K.shift) + 1;
      // past the end: only the bucket right after the last hash's is ever read (look-ups stop at q[n-1])
      const uint64_t last = i == n ? first : (q[i] >> K.shift);
      for (uint64_t b = first; b <= last; ++b) K.idx[b] = (uint32_t)i;
    }
  }
}

constexpr int kCT = 256;                // threads per workgroup
constexpr int kCTile = 8 * kCT;         // pairs per tile
constexpr uint32_t kCCap = 2048;        // read-sketch entries staged in LDS per tile (24 KB: six workgroups per CU)
// (round 3, measured again and dropped: a 1024-entry stage and eight workgroups per CU whenever a tile's run is expected under
// 512 entries — the filtered sketch of a metagenome against any table: 0.140 against 0.127 ms per three-k pass at configs[2])

// hits[g] += 1 for every pair (h, g) with h in the read sketch at count >= ci; sizes (optional, truncated
// sketches only) counts every pair.  npairs = pairs with hash <= the sketch's completeness bound.
// A matched pair of a reference-pipeline table marks, for every k below the largest, the k-prefix of its k-mer and of the reverse
// complement (the streaming query tries both strands, scripts/select_db.py:73-76).
// (all of a thread's matched pairs at once, per k: their prefix numbers are requested together, then the bits are set — pair
// by pair inside the search's branches it was a chain of load -> wait -> atomic per pair and k, and the tile waited for the
// thread with the most matches: 131 us per 10M pairs against 41 for the same kernel without the marks)
template <int N>
__device__ __forceinline__ void mark_pairs(const ContainK& K, uint64_t i0, uint32_t stride, uint32_t matched) {
  if (!matched) return;
  for (int s = 0; s < K.nsmall; ++s) {
    const uint32_t* __restrict__ pa = K.pa[s];
    const uint32_t* __restrict__ pb = K.pb[s];
    uint32_t* __restrict__ marks = K.marks[s];
    uint32_t x[N], y[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      x[j] = y[j] = 0xffffffffu;
      if ((matched >> j) & 1u) { const uint64_t i = i0 + (uint64_t)j * stride; x[j] = pa[i]; y[j] = pb[i]; }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (x[j] != 0xffffffffu) atomicOr(&marks[x[j] >> 5], 1u << (x[j] & 31u));
      if (y[j] != 0xffffffffu) atomicOr(&marks[y[j] >> 5], 1u << (y[j] & 31u));
    }
  }
}

template <bool MARK>
__global__ __launch_bounds__(kCT) void k_contain_pairs(const ContainArgs a) {
  __shared__ uint64_t s_q[kCCap];
  __shared__ uint32_t s_c[kCCap];
  __shared__ uint64_t s_edge[2];
  constexpr int kPer = kCTile / kCT;
  const int tid = threadIdx.x;
  for (uint64_t gtile = blockIdx.x; gtile < a.ntiles; gtile += gridDim.x) {
    // which k this tile belongs to (tiles of the ks follow each other; block-uniform)
    int ki = 0;
#pragma unroll
    for (int j = 1; j < kMaxContainK; ++j) ki += (j < a.nk && gtile >= a.k[j].tile0) ? 1 : 0;
    const ContainK& K = a.k[ki];
    const uint64_t* __restrict__ q = K.q;
    const uint32_t* __restrict__ qc = K.qc;
    const uint32_t* __restrict__ idx = K.idx;
    const uint64_t* __restrict__ ph = K.ph;
    const uint32_t* __restrict__ pg = K.pg;
    const unsigned shift = K.shift;
    const uint32_t ci = a.ci;
    const uint64_t npairs = K.npairs;
    const uint64_t qn = K.meta ? K.meta[1] : K.qn;          // deferred sketch: size and last hash are still on the device
    const uint64_t q_last = K.meta ? K.meta[2] : K.q_last;
    // counters are replicated (copy = workgroup id modulo the number of copies, see mg_containment_dev): a few
    // abundant genomes collect most hits, and atomics on one address retire one at a time
    uint32_t* const hits = K.hits_part + (uint64_t)(blockIdx.x & (a.copies - 1)) * K.ngenomes;
    uint32_t* const sizes = K.sizes_part ? K.sizes_part + (uint64_t)(blockIdx.x & (a.copies - 1)) * K.ngenomes : nullptr;
    const uint64_t tile = gtile - K.tile0;
    const uint64_t t0 = tile * kCTile;
    const uint64_t t1 = t0 + kCTile < npairs ? t0 + kCTile : npairs;
    // this thread's pairs: issued first, they travel while the matching run of the read sketch is located
    uint64_t h[kPer];
    uint32_t g[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const uint64_t i = t0 + tid + (uint64_t)j * kCT;
      h[j] = ~0ull;  // never in a sketch (reserved value): inactive slot
      g[j] = 0;
      if (i < t1) { h[j] = ph[i]; g[j] = pg[i]; }
    }
    // first / last hash of the tile: already among the loaded pairs (no second round trip)
    if (tid == 0) s_edge[0] = h[0];
    {
      const uint64_t il = t1 - 1 - t0;  // tile-local index of the last pair
      if ((il % kCT) == (uint64_t)tid) {
        uint64_t v = h[0];
#pragma unroll
        for (int j = 1; j < kPer; ++j) v = (il / kCT) == (uint64_t)j ? h[j] : v;
        s_edge[1] = v;
      }
    }
    __syncthreads();
    const uint64_t h_first = s_edge[0], h_last = s_edge[1];
    uint32_t lo = 0, hi = 0;  // the run of the read sketch that can match this tile
    if (qn > 0 && h_first <= q_last) {
      const uint64_t top = h_last < q_last ? h_last : q_last;
      lo = idx[h_first >> shift];
      hi = idx[(top >> shift) + 1];
    }
    const uint32_t len = hi - lo;
    const bool staged = len < kCCap;
    uint32_t pow2 = 1;  // smallest power of two > len: the padded length of the staged run
    while (pow2 <= len) pow2 <<= 1;
    if (staged) {
      // the run's loads are issued four rounds at a time (a plain loop waits for every round trip in turn)
      for (uint32_t r0 = 0; r0 < len; r0 += 4 * kCT) {
        uint64_t vq[4];
        uint32_t vc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t i = r0 + r * kCT + tid;
          vq[r] = 0; vc[r] = 0;
          if (i < len) { vq[r] = q[lo + i]; vc[r] = qc[lo + i]; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t i = r0 + r * kCT + tid;
          if (i < len) { s_q[i] = vq[r]; s_c[i] = vc[r]; }
        }
      }
      for (uint32_t i = len + tid; i < pow2; i += kCT) { s_q[i] = ~0ull; s_c[i] = 0; }
    }
    __syncthreads();
    if (sizes) {
#pragma unroll
      for (int j = 0; j < kPer; ++j)
        if (h[j] != ~0ull) atomicAdd(&sizes[g[j]], 1u);
    }
    if (len && staged) {
      // Eight lower bounds in lock step over the LDS run, padded with +inf up to a power of two so that a probe
      // needs neither a bound check nor a branch: the eight loads of a step are independent and pipeline.
      uint32_t p[kPer];
#pragma unroll
      for (int j = 0; j < kPer; ++j) p[j] = 0;
      for (uint32_t step = pow2 >> 1; step; step >>= 1) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
          const uint32_t t = p[j] + step;
          p[j] = s_q[t - 1] < h[j] ? t : p[j];
        }
      }
      uint32_t matched = 0;
#pragma unroll
      for (int j = 0; j < kPer; ++j) {  // p <= len; s_q[len] is padding, and an inactive slot (h = +inf) stops there
        if (s_q[p[j]] == h[j] && h[j] != ~0ull && s_c[p[j]] >= ci) {
          atomicAdd(&hits[g[j]], 1u);
          matched |= 1u << j;
        }
      }
      if constexpr (MARK) mark_pairs<kPer>(K, t0 + tid, kCT, matched);
    } else if (len) {
      // A run longer than the LDS stage: either the read sketch is locally much denser than the table (the top of
      // the hash range, where few genome sketches reach) or simply huge.  Every pair goes through the bucket index
      // on its own (about eight sketch entries per bucket): three dependent loads, the eight pairs' chains independent.
      uint32_t a[kPer], b[kPer];
#pragma unroll
      for (int j = 0; j < kPer; ++j) {
        a[j] = b[j] = 0;
        if (h[j] <= q_last) { const uint64_t bk = h[j] >> shift; a[j] = idx[bk]; b[j] = idx[bk + 1]; }
      }
      uint32_t matched = 0;
#pragma unroll
      for (int j = 0; j < kPer; ++j) {
        uint32_t x = a[j], y = b[j];
        while (x < y) {  // lower_bound inside the bucket
          const uint32_t mid = (x + y) >> 1;
          if (q[mid] < h[j]) x = mid + 1; else y = mid;
        }
        if (x < b[j] && q[x] == h[j] && qc[x] >= ci) {
          atomicAdd(&hits[g[j]], 1u);
          matched |= 1u << j;
        }
      }
      if constexpr (MARK) mark_pairs<kPer>(K, t0 + tid, kCT, matched);
    }
    __syncthreads();
  }
}

// Stage B after stage A BY K-MER IDENTITY (mg_kcount.hip): a pair is matched when its k-mer — counted at the first pair of the
// hash-major table that holds it, so the gather runs nearly in step with the stream — occurred >= ci times in the reads
// (saturating at cs, as `kmc -cs<cs>` does; scripts/select_db.py:50-56).  No search: 4 B of genome id, 4 B of head and one
// counter per pair; the marks of the smaller k as in k_contain_pairs.
template <bool MARK>
__global__ __launch_bounds__(kCT) void k_match_pairs(const ContainArgs a) {
  constexpr int kPer = kCTile / kCT;
  const int tid = threadIdx.x;
  const ContainK& K = a.k[0];
  const uint32_t* __restrict__ pg = K.pg;
  const uint32_t* __restrict__ head = K.head;
  const uint32_t* __restrict__ counts = K.counts;
  uint32_t* const hits = K.hits_part + (uint64_t)(blockIdx.x & (a.copies - 1)) * K.ngenomes;
  for (uint64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const uint64_t t0 = tile * kCTile;
    uint32_t g[kPer], hd[kPer], c[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const uint64_t i = t0 + tid + (uint64_t)j * kCT;
      g[j] = 0; hd[j] = 0xffffffffu;
      if (i < K.npairs) { g[j] = pg[i]; hd[j] = head[i]; }
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) c[j] = hd[j] != 0xffffffffu ? counts[hd[j]] : 0u;
    uint32_t matched = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const uint32_t v = (a.cs && c[j] > a.cs) ? a.cs : c[j];
      if (hd[j] != 0xffffffffu && v >= a.ci) {
        atomicAdd(&hits[g[j]], 1u);
        matched |= 1u << j;
      }
    }
    if constexpr (MARK) mark_pairs<kPer>(K, t0 + tid, kCT, matched);
  }
}

__global__ void k_zero_u32(uint32_t* __restrict__ v, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) v[i] = 0;
}

// hits[g] = sum over the copies; sizes[g] likewise when they were counted (truncated sketch), else the stored size.
__global__ void k_contain_reduce(const ContainArgs a) {
  const uint64_t g0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (int ki = 0; ki < a.nk; ++ki) {
    const ContainK& K = a.k[ki];
    const uint32_t* __restrict__ hits_part = K.hits_part;
    const uint32_t* __restrict__ sizes_part = K.sizes_part;
    const uint64_t ngenomes = K.ngenomes;
    for (uint64_t g = g0; g < ngenomes; g += stride) {
      uint32_t h = 0, z = 0;
      for (uint32_t c0 = 0; c0 < a.copies; c0 += 8) {  // eight independent loads per round trip
        uint32_t x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t c = c0 + u;
          x[u] = c < a.copies ? hits_part[(uint64_t)c * ngenomes + g] : 0u;
          y[u] = (c < a.copies && sizes_part) ? sizes_part[(uint64_t)c * ngenomes + g] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { h += x[u]; z += y[u]; }
      }
      K.hits[g] = h;
      K.sizes[g] = sizes_part ? z : K.gsize[g];
    }
  }
}

// ---- the reference pipeline's smaller-k columns: count list x prefix bitmap ----
struct CountK {
  const uint32_t* cid;    // the distinct (prefix number, genome) combinations of this k, ascending by prefix number
  const uint32_t* cgen;
  uint64_t n, tile0, ngenomes;
  const uint32_t* marks;  // bit p: prefix p of D_k was marked by a matched pair (this rank's, or the OR over the ranks)
  const uint32_t* gsize;  // distinct prefixes per genome (within this list)
  uint32_t* hits_part;
  uint32_t* hits;
  uint32_t* sizes;
};
struct CountArgs {
  CountK k[kMaxSmallK];
  int nk;
  uint32_t copies;
  uint64_t ntiles;
};

// hits[g] += 1 for every (p, g) of the list whose prefix is marked.  The list is sorted by prefix number, so a tile reads a short
// run of the bitmap (coalesced, cached) and atomics go only to the marked entries; counters replicated as in k_contain_pairs.
__global__ __launch_bounds__(kCT) void k_refpipe_count(const CountArgs a) {
  constexpr int kPer = kCTile / kCT;
  const int tid = threadIdx.x;
  for (uint64_t gtile = blockIdx.x; gtile < a.ntiles; gtile += gridDim.x) {
    int ki = 0;
#pragma unroll
    for (int j = 1; j < kMaxSmallK; ++j) ki += (j < a.nk && gtile >= a.k[j].tile0) ? 1 : 0;
    const CountK& K = a.k[ki];
    const uint32_t* __restrict__ cid = K.cid;
    const uint32_t* __restrict__ cgen = K.cgen;
    const uint32_t* __restrict__ marks = K.marks;
    uint32_t* const hits = K.hits_part + (uint64_t)(blockIdx.x & (a.copies - 1)) * K.ngenomes;
    const uint64_t t0 = (gtile - K.tile0) * kCTile;
    uint32_t p[kPer], g[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const uint64_t i = t0 + tid + (uint64_t)j * kCT;
      p[j] = 0xffffffffu; g[j] = 0;
      if (i < K.n) { p[j] = cid[i]; g[j] = cgen[i]; }
    }
    uint32_t w[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) w[j] = p[j] != 0xffffffffu ? marks[p[j] >> 5] : 0u;
#pragma unroll
    for (int j = 0; j < kPer; ++j)
      if ((w[j] >> (p[j] & 31u)) & 1u) atomicAdd(&hits[g[j]], 1u);
  }
}

__global__ void k_refpipe_reduce(const CountArgs a) {
  const uint64_t g0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (int ki = 0; ki < a.nk; ++ki) {
    const CountK& K = a.k[ki];
    for (uint64_t g = g0; g < K.ngenomes; g += stride) {
      uint32_t h = 0;
      for (uint32_t c = 0; c < a.copies; ++c) h += K.hits_part[(uint64_t)c * K.ngenomes + g];
      K.hits[g] = h;
      K.sizes[g] = K.gsize[g];
    }
  }
}

// ---- table upload helpers ----
__global__ void k_iota_u32(uint32_t* v, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) v[i] = (uint32_t)i;
}

// pair_gen[j] = genome whose sketch holds original entry orig[j]: last g with offsets[g] <= orig[j].
__global__ void k_pair_genomes(const uint32_t* __restrict__ orig, uint64_t n, const uint64_t* __restrict__ offsets,
                               uint64_t ngenomes, uint32_t* __restrict__ pair_gen) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const uint64_t o = orig[i];
    uint64_t lo = 0, hi = ngenomes;  // invariant: offsets[lo] <= o < offsets[hi]
    while (hi - lo > 1) {
      const uint64_t mid = (lo + hi) >> 1;
      if (offsets[mid] <= o) lo = mid; else hi = mid;
    }
    pair_gen[i] = (uint32_t)lo;
  }
}

__global__ void k_genome_sizes(const uint64_t* __restrict__ offsets, uint64_t ngenomes, uint32_t* __restrict__ gsize) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; g < ngenomes; g += stride) gsize[g] = (uint32_t)(offsets[g + 1] - offsets[g]);
}

__global__ void k_upper_bound_one(const uint64_t* __restrict__ uniq, uint64_t n, uint64_t bound, uint64_t* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (uniq[mid] <= bound) lo = mid + 1; else hi = mid;
  }
  *out = lo;
}

// Counts the containment calls: the counter copies one of them zeroed for the count step that follows it are good only
// until the next call takes the same scratch buffer.
static uint64_t g_contain_gen = 0;

// Sizes and allocates the sketch's bucket index if it has none; *build = the index kernel of this call must fill it.
static int plan_index(mg_sketch* sk, bool* build) {
  *build = false;
  if (sk->index.p) return MG_OK;
  const bool pending = sk->pending;
  const uint64_t n = pending ? sk->n_bound : sk->n;  // pending: an estimate sizes the index, the kernel reads the true n
  if (n == 0) return MG_OK;
  if (n > 0xfffffff0ull) return fail(MG_ERR_ARG, "read sketch too large for 32-bit index");
  unsigned bits = 0;
  for (uint64_t v = pending ? sk->hmax : sk->last_hash; v; v >>= 1) ++bits;
  if (bits == 0) bits = 1;
  // log2(buckets): about eight sketch entries per bucket.  A tile reads two words of the index (its run has up to eight entries of
  // slack at either end), the rare unstaged pair does three more search steps — and the build is a pass over the sketch in which
  // one thread in eight stores (one bucket per entry was 0.66 ms for the 9.7M-entry sketch at configs[3]: 67 MB of scattered stores)
  unsigned lb = 0;
  while ((1ull << lb) < (n + 7) / 8 && lb < 27) ++lb;
  if (lb < 1) lb = 1;  // keeps the shift below 64
  if (lb > bits) lb = bits;
  sk->index_shift = bits - lb;
  sk->index_buckets = 1ull << lb;
  MG_TRY(sk->index.alloc((sk->index_buckets + 1) * sizeof(uint32_t)));
  *build = true;
  return MG_OK;
}

}  // namespace mg

using namespace mg;

extern "C" {

int mg_db_upload(const uint64_t* hashes, const uint64_t* offsets, uint64_t ngenomes, mg_db** out) {
  MG_REQUIRE_READY();
  if (!out || !offsets) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  if (offsets[0] != 0) return fail(MG_ERR_ARG, "db offsets must start at 0");
  std::unique_ptr<mg_db> db(new mg_db());
  Context& c = ctx();
  hipStream_t st = c.stream;
  const uint64_t total = offsets[ngenomes];
  if (total > 0xfffffff0ull) return fail(MG_ERR_ARG, "sketch table of %llu hashes exceeds the 32-bit position range",
                                         (unsigned long long)total);
  if (ngenomes > 0xfffffff0ull) return fail(MG_ERR_ARG, "too many genomes");
  db->ngenomes = ngenomes;
  db->total = total;
  MG_TRY(db->offsets.alloc((ngenomes + 1) * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_h2d(db->offsets.p, offsets, (ngenomes + 1) * sizeof(uint64_t)));
  uint64_t mx = 0;  // each genome sketch is ascending: its maximum is its last entry
  for (uint64_t g = 0; g < ngenomes; ++g) {
    if (offsets[g + 1] < offsets[g]) return fail(MG_ERR_ARG, "db offsets must be non-decreasing");
    if (offsets[g + 1] > offsets[g] && hashes[offsets[g + 1] - 1] > mx) mx = hashes[offsets[g + 1] - 1];
  }
  db->max_hash = mx;
  MG_TRY(db->pair_hash.alloc((total + 1) * sizeof(uint64_t)));
  MG_TRY(db->pair_gen.alloc((total + 1) * sizeof(uint32_t)));
  MG_TRY(db->gsize.alloc((ngenomes + 1) * sizeof(uint32_t)));
  if (ngenomes)
    hipLaunchKernelGGL(k_genome_sizes, dim3(grid_for(ngenomes, 256, (unsigned)c.num_cus * 4)), dim3(256), 0, st,
                       db->offsets.as<uint64_t>(), ngenomes, db->gsize.as<uint32_t>());
  if (total) {
    // one-time inversion on the device: sort (hash, original index), then original index -> genome
    uint64_t* d_h = (uint64_t*)scratch("db_h", total * sizeof(uint64_t));
    uint32_t* d_i = (uint32_t*)scratch("db_i", total * sizeof(uint32_t));
    uint32_t* d_is = (uint32_t*)scratch("db_is", total * sizeof(uint32_t));
    if (!d_h || !d_i || !d_is) return MG_ERR_NOMEM;
    MG_HIP(hipMemcpyAsync(d_h, hashes, total * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    const unsigned grid = grid_for(total, 256, (unsigned)c.num_cus * 8);
    hipLaunchKernelGGL(k_iota_u32, dim3(grid), dim3(256), 0, st, d_i, total);
    MG_TRY(sort_pairs(d_h, db->pair_hash.as<uint64_t>(), d_i, d_is, total));
    hipLaunchKernelGGL(k_pair_genomes, dim3(grid), dim3(256), 0, st, d_is, total, db->offsets.as<uint64_t>(), ngenomes,
                       db->pair_gen.as<uint32_t>());
  }
  MG_HIP(hipGetLastError());
  MG_HIP(hipStreamSynchronize(st));
  *out = db.release();
  return MG_OK;
}

// A stored hash-major table is trusted for nothing: one streaming pass over what was uploaded counts the pairs that are
// out of order, above max_hash, or name a genome that does not exist (stage B adds into hits[genome] with atomics — a
// corrupt or mismatched k*.pair_gen.u32 would write out of bounds).  bad[0] order, bad[1] range, bad[2] genome id.
__global__ void k_check_pairs(const uint64_t* __restrict__ pair_hash, const uint32_t* __restrict__ pair_gen, uint64_t npairs,
                              uint64_t ngenomes, uint64_t max_hash, unsigned long long* __restrict__ bad) {
  unsigned o = 0, r = 0, g = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t h = pair_hash[i];
    o += (i > 0 && pair_hash[i - 1] > h) ? 1u : 0u;
    r += h > max_hash ? 1u : 0u;
    g += pair_gen[i] >= ngenomes ? 1u : 0u;
  }
  if (o) atomicAdd(bad + 0, (unsigned long long)o);
  if (r) atomicAdd(bad + 1, (unsigned long long)r);
  if (g) atomicAdd(bad + 2, (unsigned long long)g);
}

}  // extern "C"
namespace mg {
// k_check_pairs over an uploaded table (stream-ordered on the library stream; waits for the answer)
int check_pairs_dev(const mg_db& db) {
  if (!db.total) return MG_OK;
  hipStream_t st = ctx().stream;
  unsigned long long* d_bad = (unsigned long long*)scratch("db_check", 4 * sizeof(unsigned long long));
  if (!d_bad) return MG_ERR_NOMEM;
  MG_HIP(hipMemsetAsync(d_bad, 0, 4 * sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_check_pairs, dim3(grid_for(db.total, 256, (unsigned)ctx().num_cus * 8)), dim3(256), 0, st,
                     db.pair_hash.as<uint64_t>(), db.pair_gen.as<uint32_t>(), db.total, db.ngenomes, db.max_hash, d_bad);
  MG_HIP(hipGetLastError());
  uint64_t* pin = host_words();
  MG_HIP(hipMemcpyAsync(pin, d_bad, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  if (pin[0] || pin[1] || pin[2])
    return fail(MG_ERR_ARG, "hash-major sketch table is corrupt: %llu pairs out of order, %llu above max_hash, %llu with a genome id >= %llu",
                (unsigned long long)pin[0], (unsigned long long)pin[1], (unsigned long long)pin[2], (unsigned long long)db.ngenomes);
  return MG_OK;
}
}  // namespace mg
extern "C" {

int mg_db_upload_sorted(const uint64_t* pair_hash, const uint32_t* pair_gen, uint64_t npairs, const uint32_t* gsize,
                        uint64_t ngenomes, uint64_t max_hash, mg_db** out) {
  MG_REQUIRE_READY();
  if (!out || !gsize || (npairs && (!pair_hash || !pair_gen))) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  if (npairs > 0xfffffff0ull) return fail(MG_ERR_ARG, "sketch table of %llu hashes exceeds the 32-bit position range",
                                          (unsigned long long)npairs);
  if (ngenomes > 0xfffffff0ull) return fail(MG_ERR_ARG, "too many genomes");
  if (npairs && (pair_hash[0] > pair_hash[npairs - 1] || pair_hash[npairs - 1] > max_hash))
    return fail(MG_ERR_ARG, "pair list is not ascending within [0, max_hash]");
  std::unique_ptr<mg_db> db(new mg_db());
  hipStream_t st = ctx().stream;
  db->ngenomes = ngenomes;
  db->total = npairs;
  db->max_hash = max_hash;
  MG_TRY(db->pair_hash.alloc((npairs + 1) * sizeof(uint64_t)));
  MG_TRY(db->pair_gen.alloc((npairs + 1) * sizeof(uint32_t)));
  MG_TRY(db->gsize.alloc((ngenomes + 1) * sizeof(uint32_t)));
  if (npairs) {
    MG_TRY(upload_ranges({{pair_hash, {db->pair_hash.p, npairs * sizeof(uint64_t)}}, {pair_gen, {db->pair_gen.p, npairs * sizeof(uint32_t)}}}, st));
  }
  if (ngenomes) MG_HIP(hipMemcpyAsync(db->gsize.p, gsize, ngenomes * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  MG_TRY(check_pairs_dev(*db));
  MG_HIP(hipStreamSynchronize(st));
  *out = db.release();
  return MG_OK;
}

uint64_t mg_db_ngenomes(const mg_db* db) { return db ? db->ngenomes : 0; }
uint64_t mg_db_max_hash(const mg_db* db) { return db ? db->max_hash : 0; }
void mg_db_free(mg_db* db) { delete db; }

}  // extern "C"

// Stage B for every k of a pass: ONE index launch, ONE pairs launch over the tiles of all tables, ONE reduction.
// rp (the reference pipeline): nk == 1, qs[0] = the read sketch of the table's largest k against rp->kmax; a matched pair also
// marks its prefixes in rp's bitmaps.
static int containment_launch(int nk, const mg_sketch* const* qs, const mg_db* const* dbs, uint32_t ci, uint32_t* const* d_hits,
                              uint32_t* const* d_sizes, const mg_refdb* rp) {
  MG_REQUIRE_READY();
  if (nk < 1 || nk > kMaxContainK) return fail(MG_ERR_ARG, "between 1 and %d k per stage-B call", kMaxContainK);
  if (!qs || !dbs || !d_hits || !d_sizes) return fail(MG_ERR_ARG, "null argument");
  if (ctx().count_sat && ci > ctx().count_sat)
    return fail(MG_ERR_ARG, "count threshold ci=%u above the counters' saturation cs=%u: nothing could ever match", ci,
                ctx().count_sat);
  Context& c = ctx();
  hipStream_t st = c.stream;
  ContainArgs a{};
  a.ci = ci;
  uint64_t gmax = 0, part_total = 0, index_work = 0, reduce_work = 0;
  int m = 0;
  bool any_index = false;
  struct Pend { mg_sketch* sk; const mg_db* db; bool count_sizes; uint64_t bound; };
  Pend pend[kMaxContainK];
  for (int i = 0; i < nk; ++i) {
    if (!qs[i] || !dbs[i] || !d_hits[i] || !d_sizes[i]) return fail(MG_ERR_ARG, "null argument");
    if (dbs[i]->ngenomes == 0) continue;
    mg_sketch* sk = const_cast<mg_sketch*>(qs[i]);  // the look-up index is a cache inside the handle
    // A sketch whose finalisation is deferred is consumed as it is when no completeness bound can apply (s = 0):
    // the kernels read its size and last hash from the device, nothing is synchronised here.
    if (sk->pending && (sk->redo.s > 0 || sk->has_bound || sk->redo.use_bound)) MG_TRY(sketch_resolve(sk, nullptr));
    MG_TRY(sketch_wait(sk));  // built on another stream: this one waits for it on the device
    uint64_t bound = (sk->truncated && sk->n > 0) ? sk->last_hash : ~0ull;
    if (sk->has_bound) bound = sk->truncated ? sk->bound : ~0ull;
    if (rp && bound != ~0ull)
      return fail(MG_ERR_ARG, "the reference pipeline counts every k_max-mer of the reads (kmc, scripts/select_db.py:50-52): a bottom-s "
                              "sketch cannot be its query");
    pend[m] = Pend{sk, dbs[i], bound != ~0ull, bound};  // truncated sketch: only table hashes <= bound take part, in the sizes too
    ContainK& K = a.k[m];
    K.hits = d_hits[i];
    K.sizes = d_sizes[i];
    if (dbs[i]->ngenomes > gmax) gmax = dbs[i]->ngenomes;
    ++m;
  }
  if (m == 0) return MG_OK;
  a.nk = m;
  // counter copies: enough to spread a skewed sample's hits, few enough to zero and sum in microseconds
  uint32_t copies = 1;
  while (copies < 64 && (uint64_t)copies * 2 * gmax <= 65536) copies *= 2;
  a.copies = copies;
  for (int i = 0; i < m; ++i) part_total += (pend[i].count_sizes ? 2 : 1) * (uint64_t)copies * pend[i].db->ngenomes;
  // (the reference pipeline: the count step's counter copies ride in the same buffer and are zeroed by the same launch)
  const uint64_t count_total = rp ? (uint64_t)(rp->nk - 1) * copies * rp->kmax.ngenomes : 0;
  uint32_t* d_part = (uint32_t*)scratch("contain_part", (part_total + count_total) * sizeof(uint32_t));
  if (!d_part) return MG_ERR_NOMEM;
  a.zero = d_part;
  a.nzero = part_total + count_total;
  ++g_contain_gen;
  if (rp) { rp->count_part = d_part + part_total; rp->count_copies = copies; rp->count_gen = g_contain_gen; }
  uint64_t at = 0, tiles = 0;
  for (int i = 0; i < m; ++i) {
    mg_sketch* sk = pend[i].sk;
    const mg_db* db = pend[i].db;
    ContainK& K = a.k[i];
    bool build = false;
    MG_TRY(plan_index(sk, &build));
    any_index = any_index || build;
    K.q = sk->hashes.as<uint64_t>();
    K.qc = sk->counts.as<uint32_t>();
    K.qn = sk->n;
    K.q_last = sk->last_hash;
    K.meta = sk->pending ? sk->meta.as<uint64_t>() : nullptr;
    K.idx = sk->index.as<uint32_t>();
    K.shift = sk->index_shift;
    K.idx_buckets = sk->index_buckets;
    K.build_index = build ? 1u : 0u;
    if (build) { const uint64_t n = sk->pending ? sk->n_bound : sk->n; if (n + 1 > index_work) index_work = n + 1; }
    K.ph = db->pair_hash.as<uint64_t>();
    K.pg = db->pair_gen.as<uint32_t>();
    K.gsize = db->gsize.as<uint32_t>();
    K.ngenomes = db->ngenomes;
    K.hits_part = d_part + at;
    at += (uint64_t)copies * db->ngenomes;
    K.sizes_part = nullptr;
    if (pend[i].count_sizes) { K.sizes_part = d_part + at; at += (uint64_t)copies * db->ngenomes; }
    K.npairs = db->total;
    if (pend[i].count_sizes) {  // rare path: one read-back
      uint64_t* d_bpos = (uint64_t*)scratch("contain_bpos", sizeof(uint64_t));
      if (!d_bpos) return MG_ERR_NOMEM;
      hipLaunchKernelGGL(k_upper_bound_one, dim3(1), dim3(64), 0, st, db->pair_hash.as<uint64_t>(), db->total, pend[i].bound, d_bpos);
      uint64_t* pin = host_words();
      MG_HIP(hipMemcpyAsync(pin + 20, d_bpos, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      MG_HIP(hipStreamSynchronize(st));
      K.npairs = pin[20];
    }
    K.tile0 = tiles;
    tiles += (K.npairs + kCTile - 1) / kCTile;
    if (db->ngenomes > reduce_work) reduce_work = db->ngenomes;
    K.nsmall = 0;
    if (rp) {
      K.nsmall = rp->nk - 1;
      for (int s = 0; s < K.nsmall; ++s) {
        K.pa[s] = rp->small[s].pa.as<uint32_t>();
        K.pb[s] = rp->small[s].pb.as<uint32_t>();
        K.marks[s] = rp->marks.as<uint32_t>() + rp->small[s].marks_at;
      }
    }
  }
  a.ntiles = tiles;
  if (rp) { a.zero2 = rp->marks.as<uint32_t>(); a.nzero2 = rp->marks_words; }
  {
    ProfScope ps("contain_index");
    const uint64_t work = index_work > a.nzero ? index_work : a.nzero;
    hipLaunchKernelGGL(k_build_index, dim3(grid_for(work ? work : 1, 256, (unsigned)c.num_cus * 16)), dim3(256), 0, st, a);
    MG_HIP(hipGetLastError());
  }
  ProfScope ps("containment");
  if (tiles) {
    if (rp)
      hipLaunchKernelGGL(k_contain_pairs<true>, dim3(grid_for(tiles, 1, (unsigned)c.num_cus * 6)), dim3(kCT), 0, st, a);
    else
      hipLaunchKernelGGL(k_contain_pairs<false>, dim3(grid_for(tiles, 1, (unsigned)c.num_cus * 6)), dim3(kCT), 0, st, a);
  }
  hipLaunchKernelGGL(k_contain_reduce, dim3(grid_for(reduce_work, 256, (unsigned)c.num_cus * 4)), dim3(256), 0, st, a);
  MG_HIP(hipGetLastError());
  return MG_OK;
}

// Stage B of the reference pipeline from the k-mer counters of stage A by identity: the largest k's column + the prefix marks.
static int match_launch(const uint32_t* d_counts, const mg_refdb* rp, uint32_t ci, uint32_t* d_hits, uint32_t* d_sizes) {
  Context& c = ctx();
  hipStream_t st = c.stream;
  if (c.count_sat && ci > c.count_sat)
    return fail(MG_ERR_ARG, "count threshold ci=%u above the counters' saturation cs=%u: nothing could ever match", ci, c.count_sat);
  const mg_db* db = &rp->kmax;
  if (db->ngenomes == 0) return MG_OK;
  if (ci == 0) return fail(MG_ERR_ARG, "ci = 0 would match every pair");
  ContainArgs a{};
  a.ci = ci;
  a.cs = c.count_sat;
  a.nk = 1;
  uint32_t copies = 1;
  while (copies < 64 && (uint64_t)copies * 2 * db->ngenomes <= 65536) copies *= 2;
  a.copies = copies;
  const uint64_t part_total = (uint64_t)copies * db->ngenomes;
  const uint64_t count_total = (uint64_t)(rp->nk - 1) * copies * db->ngenomes;
  uint32_t* d_part = (uint32_t*)scratch("contain_part", (part_total + count_total) * sizeof(uint32_t));
  if (!d_part) return MG_ERR_NOMEM;
  a.zero = d_part;
  a.nzero = part_total + count_total;
  ++g_contain_gen;
  rp->count_part = d_part + part_total; rp->count_copies = copies; rp->count_gen = g_contain_gen;
  ContainK& K = a.k[0];
  K.hits = d_hits;
  K.sizes = d_sizes;
  K.pg = db->pair_gen.as<uint32_t>();
  K.gsize = db->gsize.as<uint32_t>();
  K.ngenomes = db->ngenomes;
  K.npairs = db->total;
  K.hits_part = d_part;
  K.sizes_part = nullptr;
  K.head = rp->kidx->head.as<uint32_t>();
  K.counts = d_counts;
  K.nsmall = rp->nk - 1;
  for (int s = 0; s < K.nsmall; ++s) {
    K.pa[s] = rp->small[s].pa.as<uint32_t>();
    K.pb[s] = rp->small[s].pb.as<uint32_t>();
    K.marks[s] = rp->marks.as<uint32_t>() + rp->small[s].marks_at;
  }
  a.ntiles = (K.npairs + kCTile - 1) / kCTile;
  a.zero2 = rp->marks.as<uint32_t>();
  a.nzero2 = rp->marks_words;
  ProfScope ps("containment");
  const uint64_t work = a.nzero > a.nzero2 ? a.nzero : a.nzero2;
  hipLaunchKernelGGL(k_build_index, dim3(grid_for(work ? work : 1, 256, (unsigned)c.num_cus * 16)), dim3(256), 0, st, a);  // (zeroes; no index here)
  if (a.ntiles) hipLaunchKernelGGL(k_match_pairs<true>, dim3(grid_for(a.ntiles, 1, (unsigned)c.num_cus * 8)), dim3(kCT), 0, st, a);
  hipLaunchKernelGGL(k_contain_reduce, dim3(grid_for(db->ngenomes, 256, (unsigned)c.num_cus * 4)), dim3(256), 0, st, a);
  MG_HIP(hipGetLastError());
  return MG_OK;
}

// The smaller-k columns of the reference pipeline from the prefix bitmaps (rp's own after a mark, or d_marks: the OR over the ranks
// of a multi-GPU job): ONE launch over the count lists of all k, one reduction.  d_hits / d_sizes: nk - 1 of them.
static int refpipe_count_launch(const mg_refdb* rp, const uint32_t* const* d_marks, uint32_t* const* d_hits, uint32_t* const* d_sizes) {
  Context& c = ctx();
  hipStream_t st = c.stream;
  const int m = rp->nk - 1;
  if (m == 0 || rp->kmax.ngenomes == 0) return MG_OK;
  CountArgs a{};
  a.nk = m;
  const uint64_t G = rp->kmax.ngenomes;
  uint32_t copies = 1;
  while (copies < 64 && (uint64_t)copies * 2 * G <= 65536) copies *= 2;
  a.copies = copies;
  // zeroed by the mark call that came before (same copies: same G), else here
  const bool prezeroed = rp->count_part != nullptr && rp->count_copies == copies && rp->count_gen == g_contain_gen;
  uint32_t* d_part = prezeroed ? rp->count_part : (uint32_t*)scratch("refpipe_part", (uint64_t)m * copies * G * sizeof(uint32_t));
  if (!d_part) return MG_ERR_NOMEM;
  rp->count_part = nullptr;  // (used up: a second count without a mark in between zeroes for itself)
  uint64_t tiles = 0;
  for (int s = 0; s < m; ++s) {
    CountK& K = a.k[s];
    const uint64_t nc = rp->small[s].ncount, W = rp->share_world ? rp->share_world : 1, r = rp->share_rank;
    const uint64_t lo = nc / W * r + (nc % W) * r / W, hi = r + 1 == W ? nc : nc / W * (r + 1) + (nc % W) * (r + 1) / W;
    K.cid = rp->small[s].cid.as<uint32_t>() + lo;
    K.cgen = rp->small[s].cgen.as<uint32_t>() + lo;
    K.n = hi - lo;
    K.tile0 = tiles;
    tiles += (K.n + kCTile - 1) / kCTile;
    K.ngenomes = G;
    K.marks = d_marks ? d_marks[s] : rp->marks.as<uint32_t>() + rp->small[s].marks_at;
    K.gsize = rp->small[s].gsize.as<uint32_t>();
    K.hits_part = d_part + (uint64_t)s * copies * G;
    K.hits = d_hits[s];
    K.sizes = d_sizes[s];
  }
  a.ntiles = tiles;
  ProfScope ps("refpipe_count");
  if (!prezeroed)
    hipLaunchKernelGGL(k_zero_u32, dim3(grid_for((uint64_t)m * copies * G, 256, (unsigned)c.num_cus * 8)), dim3(256), 0, st, d_part,
                       (uint64_t)m * copies * G);
  if (tiles) hipLaunchKernelGGL(k_refpipe_count, dim3(grid_for(tiles, 1, (unsigned)c.num_cus * 8)), dim3(kCT), 0, st, a);
  hipLaunchKernelGGL(k_refpipe_reduce, dim3(grid_for(G, 256, (unsigned)c.num_cus * 4)), dim3(256), 0, st, a);
  MG_HIP(hipGetLastError());
  return MG_OK;
}

extern "C" {

int mg_containment_multi_dev(int nk, const mg_sketch* const* qs, const mg_db* const* dbs, uint32_t ci, uint32_t* const* d_hits,
                             uint32_t* const* d_sizes) {
  return containment_launch(nk, qs, dbs, ci, d_hits, d_sizes, nullptr);
}

int mg_refpipe_mark_dev(const mg_sketch* q, const mg_refdb* db, uint32_t ci, uint32_t* d_hits_kmax, uint32_t* d_sizes_kmax) {
  if (!q || !db) return fail(MG_ERR_ARG, "null argument");
  MG_TRY(refdb_ready(db));
  const mg_db* kdb = &db->kmax;
  return containment_launch(1, &q, &kdb, ci, &d_hits_kmax, &d_sizes_kmax, db);
}

int mg_refdb_set_count_share(const mg_refdb* db, uint32_t rank, uint32_t world) {
  if (!db || world == 0 || rank >= world) return fail(MG_ERR_ARG, "rank %u of %u", rank, world);
  db->share_rank = rank;
  db->share_world = world;
  return MG_OK;
}

int mg_refpipe_count_dev(const mg_refdb* db, const uint32_t* const* d_marks, uint32_t* const* d_hits, uint32_t* const* d_sizes) {
  MG_REQUIRE_READY();
  if (!db || !d_hits || !d_sizes) return fail(MG_ERR_ARG, "null argument");
  MG_TRY(refdb_ready(db));
  return refpipe_count_launch(db, d_marks, d_hits, d_sizes);
}

int mg_refpipe_containment_dev(const mg_sketch* q, const mg_refdb* db, uint32_t ci, uint32_t* const* d_hits, uint32_t* const* d_sizes) {
  if (!q || !db || !d_hits || !d_sizes) return fail(MG_ERR_ARG, "null argument");
  const int last = db->nk - 1;
  MG_TRY(mg_refpipe_mark_dev(q, db, ci, d_hits[last], d_sizes[last]));
  return refpipe_count_launch(db, nullptr, d_hits, d_sizes);
}

int mg_refpipe_mark_counts_dev(const mg_kcounts* kc, const mg_refdb* db, uint32_t ci, uint32_t* d_hits_kmax, uint32_t* d_sizes_kmax) {
  MG_REQUIRE_READY();
  if (!kc || !db || !d_hits_kmax || !d_sizes_kmax) return fail(MG_ERR_ARG, "null argument");
  MG_TRY(refdb_ready(db));
  if (!db->kidx) return fail(MG_ERR_STATE, "the table has no k-mer index (mg_refdb_index_kmers)");
  uint32_t* d_counts = nullptr;
  uint64_t n = 0;
  MG_TRY(mg_kcounts_device(kc, &d_counts, &n));
  if (n != db->kmax.total) return fail(MG_ERR_ARG, "these counters belong to another table");
  MG_TRY(kcounts_wait(kc));
  return match_launch(d_counts, db, ci, d_hits_kmax, d_sizes_kmax);
}

int mg_refpipe_mark_counts_ptr_dev(const uint32_t* d_counts, const mg_refdb* db, uint32_t ci, uint32_t* d_hits_kmax, uint32_t* d_sizes_kmax) {
  MG_REQUIRE_READY();
  if (!d_counts || !db || !d_hits_kmax || !d_sizes_kmax) return fail(MG_ERR_ARG, "null argument");
  MG_TRY(refdb_ready(db));
  if (!db->kidx) return fail(MG_ERR_STATE, "the table has no k-mer index (mg_refdb_index_kmers)");
  return match_launch(d_counts, db, ci, d_hits_kmax, d_sizes_kmax);
}

int mg_refpipe_containment_counts_dev(const mg_kcounts* kc, const mg_refdb* db, uint32_t ci, uint32_t* const* d_hits, uint32_t* const* d_sizes) {
  if (!kc || !db || !d_hits || !d_sizes) return fail(MG_ERR_ARG, "null argument");
  const int last = db->nk - 1;
  MG_TRY(mg_refpipe_mark_counts_dev(kc, db, ci, d_hits[last], d_sizes[last]));
  return refpipe_count_launch(db, nullptr, d_hits, d_sizes);
}

int mg_containment_dev(const mg_sketch* q, const mg_db* db, uint32_t ci, uint32_t* d_hits, uint32_t* d_sizes) {
  return mg_containment_multi_dev(1, &q, &db, ci, &d_hits, &d_sizes);
}

int mg_containment(const uint64_t* q_hashes, const uint32_t* q_counts, uint64_t qn, int q_truncated, uint32_t ci,
                   const uint64_t* db_hashes, const uint64_t* db_offsets, uint64_t ngenomes, uint32_t* out_hits,
                   uint32_t* out_sizes) {
  MG_REQUIRE_READY();
  std::unique_ptr<mg_sketch> sk(new mg_sketch());
  MG_TRY(sk->hashes.alloc(qn * sizeof(uint64_t)));
  MG_TRY(sk->counts.alloc(qn * sizeof(uint32_t)));
  MG_TRY(mg_memcpy_h2d(sk->hashes.p, q_hashes, qn * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_h2d(sk->counts.p, q_counts, qn * sizeof(uint32_t)));
  sk->n = qn;
  sk->truncated = q_truncated;
  sk->last_hash = qn ? q_hashes[qn - 1] : 0;
  mg_db* db = nullptr;
  MG_TRY(mg_db_upload(db_hashes, db_offsets, ngenomes, &db));
  std::unique_ptr<mg_db> dbg(db);
  DevBuf d_hits, d_sizes;
  MG_TRY(d_hits.alloc(ngenomes * sizeof(uint32_t)));
  MG_TRY(d_sizes.alloc(ngenomes * sizeof(uint32_t)));
  MG_TRY(mg_containment_dev(sk.get(), db, ci, d_hits.as<uint32_t>(), d_sizes.as<uint32_t>()));
  MG_TRY(mg_memcpy_d2h(out_hits, d_hits.p, ngenomes * sizeof(uint32_t)));
  MG_TRY(mg_memcpy_d2h(out_sizes, d_sizes.p, ngenomes * sizeof(uint32_t)));
  return MG_OK;
}

}  // extern "C"
