// mg_contain.hip — Stage B: containment of every genome sketch in the read sketch.
//
// K2 `k_containment`: one wavefront per genome.  Both operands are sorted sets
// (genome sketch ascending, read sketch ascending); the intersection is taken
// by looking each genome hash up in the read sketch through a bucket index
// over the hash's leading bits (expected bucket population ~1), 64 genome
// hashes per step, and counting matches with a wavefront ballot + popcount.
// The genome table is streamed once, 512 B per wavefront-step, fully coalesced;
// the read sketch and its index are L2 / Infinity-Cache resident.
//
// Replaces: kmc_tools simple ... intersect (scripts/select_db.py:54-56) and the
// containment index of StreamingQueryDNADatabase.py (scripts/select_db.py:73-76).
#include <memory>

#include "mg_internal.h"

namespace mg {

// idx[b] = first position whose hash >= b << shift; idx[nbuckets] = n.
__global__ void k_build_index(const uint64_t* __restrict__ q, uint64_t n, unsigned shift, uint64_t nbuckets,
                              uint32_t* __restrict__ idx) {
  uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nbuckets) return;
  if (b == nbuckets) { idx[b] = (uint32_t)n; return; }
  const uint64_t key = b << shift;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (q[mid] < key) lo = mid + 1; else hi = mid;
  }
  idx[b] = (uint32_t)lo;
}

__global__ __launch_bounds__(256) void k_containment(const uint64_t* __restrict__ q, const uint32_t* __restrict__ qc,
                                                     uint64_t qn, uint64_t q_last, const uint32_t* __restrict__ idx,
                                                     unsigned shift, uint64_t bound, uint32_t ci,
                                                     const uint64_t* __restrict__ db, const uint64_t* __restrict__ offs,
                                                     uint64_t ngenomes, uint32_t* __restrict__ hits,
                                                     uint32_t* __restrict__ sizes) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t g = wave; g < ngenomes; g += nwaves) {
    const uint64_t beg = offs[g], end = offs[g + 1];
    uint32_t nh = 0, ns = 0;
    for (uint64_t base = beg; base < end; base += 64) {
      const uint64_t i = base + lane;
      bool inb = false, found = false;
      if (i < end) {
        const uint64_t h = db[i];
        inb = h <= bound;
        if (inb && qn > 0 && h <= q_last) {
          const uint64_t b = h >> shift;
          uint32_t lo = idx[b], hi = idx[b + 1];
          while (lo < hi) {  // lower_bound inside the bucket
            uint32_t mid = (lo + hi) >> 1;
            if (q[mid] < h) lo = mid + 1; else hi = mid;
          }
          found = lo < qn && q[lo] == h && qc[lo] >= ci;
        }
      }
      nh += __popcll(__ballot(found));
      ns += __popcll(__ballot(inb));
    }
    if (lane == 0) { hits[g] = nh; sizes[g] = ns; }
  }
}

static int ensure_index(mg_sketch* sk) {
  if (sk->index.p || sk->n == 0) return MG_OK;
  if (sk->n > 0xfffffff0ull) return fail(MG_ERR_ARG, "read sketch too large for 32-bit index");
  unsigned bits = 0;
  for (uint64_t v = sk->last_hash; v; v >>= 1) ++bits;
  if (bits == 0) bits = 1;
  unsigned lb = 0;  // log2(buckets): about one sketch entry per bucket, at most 2^27 buckets
  while ((1ull << lb) < sk->n && lb < 27) ++lb;
  if (lb < 1) lb = 1;  // keeps the shift below 64
  if (lb > bits) lb = bits;
  sk->index_shift = bits - lb;
  sk->index_buckets = 1ull << lb;
  MG_TRY(sk->index.alloc((sk->index_buckets + 1) * sizeof(uint32_t)));
  ProfScope ps("contain_index");
  hipLaunchKernelGGL(k_build_index, dim3((unsigned)((sk->index_buckets + 1 + 255) / 256)), dim3(256), 0, ctx().stream,
                     sk->hashes.as<uint64_t>(), sk->n, sk->index_shift, sk->index_buckets, sk->index.as<uint32_t>());
  MG_HIP(hipGetLastError());
  return MG_OK;
}

}  // namespace mg

using namespace mg;

extern "C" {

int mg_db_upload(const uint64_t* hashes, const uint64_t* offsets, uint64_t ngenomes, mg_db** out) {
  MG_REQUIRE_READY();
  if (!out || !offsets) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  std::unique_ptr<mg_db> db(new mg_db());
  db->ngenomes = ngenomes;
  db->total = offsets[ngenomes] - offsets[0];
  if (offsets[0] != 0) return fail(MG_ERR_ARG, "db offsets must start at 0");
  MG_TRY(db->hashes.alloc(db->total * sizeof(uint64_t)));
  MG_TRY(db->offsets.alloc((ngenomes + 1) * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_h2d(db->hashes.p, hashes, db->total * sizeof(uint64_t)));
  MG_TRY(mg_memcpy_h2d(db->offsets.p, offsets, (ngenomes + 1) * sizeof(uint64_t)));
  uint64_t mx = 0;  // each genome sketch is ascending: its maximum is its last entry
  for (uint64_t g = 0; g < ngenomes; ++g)
    if (offsets[g + 1] > offsets[g] && hashes[offsets[g + 1] - 1] > mx) mx = hashes[offsets[g + 1] - 1];
  db->max_hash = mx;
  *out = db.release();
  return MG_OK;
}

uint64_t mg_db_ngenomes(const mg_db* db) { return db ? db->ngenomes : 0; }
uint64_t mg_db_max_hash(const mg_db* db) { return db ? db->max_hash : 0; }
void mg_db_free(mg_db* db) { delete db; }

int mg_containment_dev(const mg_sketch* q, const mg_db* db, uint32_t ci, uint32_t* d_hits, uint32_t* d_sizes) {
  MG_REQUIRE_READY();
  if (!q || !db || !d_hits || !d_sizes) return fail(MG_ERR_ARG, "null argument");
  if (db->ngenomes == 0) return MG_OK;
  mg_sketch* sk = const_cast<mg_sketch*>(q);  // the look-up index is a cache inside the handle
  MG_TRY(ensure_index(sk));
  uint64_t bound = (sk->truncated && sk->n > 0) ? sk->last_hash : ~0ull;
  if (sk->has_bound) bound = sk->truncated ? sk->bound : ~0ull;
  Context& c = ctx();
  ProfScope ps("containment");
  unsigned grid = grid_for(db->ngenomes, 4, (unsigned)c.num_cus * 8);
  hipLaunchKernelGGL(k_containment, dim3(grid), dim3(256), 0, c.stream, sk->hashes.as<uint64_t>(),
                     sk->counts.as<uint32_t>(), sk->n, sk->last_hash, sk->index.as<uint32_t>(), sk->index_shift, bound,
                     ci, db->hashes.as<uint64_t>(), db->offsets.as<uint64_t>(), db->ngenomes, d_hits, d_sizes);
  MG_HIP(hipGetLastError());
  return MG_OK;
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
