// mg_refpipe.hip — the table of the REFERENCE PIPELINE: stage A/B wired the way scripts/select_db.py wires KMC and CMash.
//
// The reference counts only k_max-mers of the reads (`kmc -k60 -ci2 -cs3`, scripts/select_db.py:50-52), intersects them with the
// sketches' k_max-mers (:54-59) and lets CMash's streaming query derive every smaller-k column from the k-PREFIXES of the surviving
// k_max-mers and of their reverse complements (:73-76, k range 30-60-10).  So the read side hashes ONE k, and what a smaller k's
// column needs is a function of which sketched k_max-mers matched — prepared here, once per table, on the device:
//   pairs            the hash-major table of k_max (as mg_db)
//   pa / pb [pair]   per k < k_max: the number (rank in ascending order among the table's distinct k-prefixes D_k) of the k-prefix
//                    of the pair's kept k-mer, and of its reverse complement's k-prefix if that string is in D_k
//   count list       per k < k_max: the distinct (prefix number, genome) combinations, ascending; gsize_k[g] of them per genome
// Stage B of a pass (mg_contain.hip): k_contain_pairs<true> finds the matched pairs and sets their prefixes' bits,
// k_refpipe_count streams the count lists against the bitmaps.  Normative statement: oracle/mg_oracle.c (mgo_refpipe_*).
//
// Sorting is rocPRIM's (mg_sort.hip): 128-bit prefix keys as two stable 64-bit passes.
#include <memory>

#include "mg_internal.h"

namespace mg {

namespace {

__global__ void k_rp_iota(uint32_t* v, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) v[i] = (uint32_t)i;
}

// pair_gen[j] = genome whose sketch holds entry perm[j]; the entry's k-mer moves to pair order
__global__ void k_rp_pairs(const uint32_t* __restrict__ perm, uint64_t n, const uint64_t* __restrict__ offsets, uint64_t ngenomes,
                           const uint64_t* __restrict__ khi, const uint64_t* __restrict__ klo, uint32_t* __restrict__ pair_gen,
                           uint64_t* __restrict__ phi, uint64_t* __restrict__ plo) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const uint64_t o = perm[i];
    uint64_t lo = 0, hi = ngenomes;  // invariant: offsets[lo] <= o < offsets[hi]
    while (hi - lo > 1) {
      const uint64_t mid = (lo + hi) >> 1;
      if (offsets[mid] <= o) lo = mid; else hi = mid;
    }
    pair_gen[i] = (uint32_t)lo;
    phi[i] = khi[o];
    plo[i] = klo[o];
  }
}

__global__ void k_rp_gsizes(const uint64_t* __restrict__ offsets, uint64_t ngenomes, uint32_t* __restrict__ gsize) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; g < ngenomes; g += stride) gsize[g] = (uint32_t)(offsets[g + 1] - offsets[g]);
}

struct U128 { uint64_t hi, lo; };
__device__ __forceinline__ U128 shr128(U128 v, int s) {  // 0 <= s < 128
  if (s == 0) return v;
  if (s >= 64) return U128{0, s == 64 ? v.hi : (v.hi >> (s - 64))};
  return U128{v.hi >> s, (v.lo >> s) | (v.hi << (64 - s))};
}
// the order of the 2-bit groups of a 64-bit word reversed
__device__ __forceinline__ uint64_t rev2(uint64_t x) {
  x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
  x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
  x = ((x >> 8) & 0x00ff00ff00ff00ffull) | ((x & 0x00ff00ff00ff00ffull) << 8);
  x = ((x >> 16) & 0x0000ffff0000ffffull) | ((x & 0x0000ffff0000ffffull) << 16);
  return (x >> 32) | (x << 32);
}

// Per pair: A = the first k bases of the kept kmax-mer, B = the first k bases of its reverse complement (= the reverse
// complement of its LAST k bases), both 2-bit packed, first base most significant, right-aligned.
__global__ void k_rp_prefix_keys(const uint64_t* __restrict__ khi, const uint64_t* __restrict__ klo, uint64_t n, int kmax, int k,
                                 uint64_t* __restrict__ a_hi, uint64_t* __restrict__ a_lo, uint64_t* __restrict__ b_hi,
                                 uint64_t* __restrict__ b_lo) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const U128 y{khi[i], klo[i]};
    const U128 a = shr128(y, 2 * (kmax - k));
    // the last k bases = the low 2k bits; complemented, their order reversed: reverse all 64 groups of the 128-bit value and
    // shift the wanted k groups (now on top) down
    U128 low = y;
    if (k < 32) { low.hi = 0; low.lo &= (1ull << (2 * k)) - 1ull; }
    else if (k == 32) low.hi = 0;
    else if (k < 64) low.hi &= (1ull << (2 * (k - 32))) - 1ull;
    const U128 comp{~low.hi, ~low.lo};                    // complement: code -> 3 - code (the unused high groups turn to 3s ...)
    const U128 rev{rev2(comp.lo), rev2(comp.hi)};        // ... and land at the bottom after the reversal, where the shift drops them
    const U128 b = shr128(rev, 2 * (64 - k));
    a_hi[i] = a.hi; a_lo[i] = a.lo;
    b_hi[i] = b.hi; b_lo[i] = b.lo;
  }
}

__global__ void k_rp_gather_u64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n, uint64_t* __restrict__ dst) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[idx[i]];
}

// flag[j] = 1 where sorted key j differs from its predecessor
__global__ void k_rp_heads128(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint64_t n, uint32_t* __restrict__ flag) {
  uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; j < n; j += stride) flag[j] = (j == 0 || lo[j] != lo[j - 1] || (hi && hi[j] != hi[j - 1])) ? 1u : 0u;
}

// pa[order[j]] = number of sorted key j; D[number] = the key
__global__ void k_rp_number(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, const uint32_t* __restrict__ order,
                            const uint32_t* __restrict__ flag, const uint64_t* __restrict__ before, uint64_t n, uint32_t* __restrict__ pa,
                            uint64_t* __restrict__ d_hi, uint64_t* __restrict__ d_lo) {
  uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; j < n; j += stride) {
    const uint64_t id = before[j] + flag[j] - 1;  // heads before j, j's own included
    pa[order[j]] = (uint32_t)id;
    if (flag[j]) { d_hi[id] = hi ? hi[j] : 0ull; d_lo[id] = lo[j]; }
  }
}

// pb[i] = number of key B[i] in D (ascending), 0xffffffff when absent
__global__ void k_rp_lookup(const uint64_t* __restrict__ b_hi, const uint64_t* __restrict__ b_lo, uint64_t n,
                            const uint64_t* __restrict__ d_hi, const uint64_t* __restrict__ d_lo, uint64_t nd, uint32_t* __restrict__ pb) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const uint64_t h = b_hi[i], l = b_lo[i];
    uint64_t lo = 0, hi = nd;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      const bool less = d_hi[mid] < h || (d_hi[mid] == h && d_lo[mid] < l);
      if (less) lo = mid + 1; else hi = mid;
    }
    pb[i] = (lo < nd && d_hi[lo] == h && d_lo[lo] == l) ? (uint32_t)lo : 0xffffffffu;
  }
}

__global__ void k_rp_count_keys(const uint32_t* __restrict__ pa, const uint32_t* __restrict__ gen, uint64_t n, uint64_t* __restrict__ key) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) key[i] = ((uint64_t)pa[i] << 32) | gen[i];
}

__global__ void k_rp_count_emit(const uint64_t* __restrict__ key, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ before,
                                uint64_t n, uint32_t* __restrict__ cid, uint32_t* __restrict__ cgen, uint32_t* __restrict__ gsize) {
  uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; j < n; j += stride) {
    if (!flag[j]) continue;
    const uint64_t at = before[j];
    cid[at] = (uint32_t)(key[j] >> 32);
    cgen[at] = (uint32_t)key[j];
    atomicAdd(&gsize[(uint32_t)key[j]], 1u);
  }
}

// what an uploaded table must satisfy for stage B's atomics and bit sets to stay in bounds
__global__ void k_rp_check(const uint32_t* __restrict__ pa, const uint32_t* __restrict__ pb, uint64_t npairs, const uint32_t* __restrict__ cid,
                           const uint32_t* __restrict__ cgen, uint64_t ncount, uint64_t nprefix, uint64_t ngenomes,
                           unsigned long long* __restrict__ bad) {
  unsigned b = 0;
  const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = i0; i < npairs; i += stride) b += (pa[i] >= nprefix || (pb[i] != 0xffffffffu && pb[i] >= nprefix)) ? 1u : 0u;
  for (uint64_t i = i0; i < ncount; i += stride) b += (cid[i] >= nprefix || cgen[i] >= ngenomes || (i > 0 && cid[i - 1] > cid[i])) ? 1u : 0u;
  if (b) atomicAdd(bad, (unsigned long long)b);
}

inline unsigned g256(uint64_t n) { return grid_for(n ? n : 1, 256, (unsigned)ctx().num_cus * 8); }

int alloc_marks(mg_refdb* db) {
  uint64_t words = 0;
  for (int s = 0; s < db->nk - 1; ++s) {
    db->small[s].marks_at = words;
    db->small[s].marks_n = (db->small[s].nprefix + 31) / 32;
    words += db->small[s].marks_n;
  }
  db->marks_words = words;
  MG_TRY(db->marks.alloc((words + 1) * sizeof(uint32_t)));
  MG_HIP(hipMemsetAsync(db->marks.p, 0, (words + 1) * sizeof(uint32_t), ctx().stream));
  return MG_OK;
}

}  // namespace

// A table whose arrays are still on their way up (mg_refdb_upload_begin): wait for them, then trust them for nothing — the pairs
// (k_check_pairs) and the prefix structures (k_rp_check) are checked before anything reads them.  An error stays with the handle.
int refdb_ready(const mg_refdb* db) {
  if (!db) return fail(MG_ERR_ARG, "null argument");
  if (db->failed) return fail(db->failed, "this table's upload failed earlier (the handle holds no table: free it)");
  hipStream_t st = ctx().stream;
  if (db->pending) {
    UploadJob* job = db->pending;
    db->pending = nullptr;
    const int rc = upload_ranges_end(job, st);
    if (rc != MG_OK) { db->failed = rc; return rc; }  // (its buffers are partly filled: nothing may read them, ever)
  }
  if (!db->unchecked) return MG_OK;
  {
    const int rc = check_pairs_dev(db->kmax);
    if (rc != MG_OK) { db->failed = rc; return rc; }
  }
  unsigned long long* d_bad = (unsigned long long*)scratch("db_check", 4 * sizeof(unsigned long long));
  if (!d_bad) return MG_ERR_NOMEM;
  MG_HIP(hipMemsetAsync(d_bad, 0, 4 * sizeof(unsigned long long), st));
  const uint64_t npairs = db->kmax.total;
  for (int s = 0; s < db->nk - 1; ++s) {
    const mg_refdb::Small& S = db->small[s];
    hipLaunchKernelGGL(k_rp_check, dim3(g256(npairs > S.ncount ? npairs : S.ncount)), dim3(256), 0, st, S.pa.as<uint32_t>(), S.pb.as<uint32_t>(),
                       npairs, S.cid.as<uint32_t>(), S.cgen.as<uint32_t>(), S.ncount, S.nprefix, db->kmax.ngenomes, d_bad);
    MG_HIP(hipGetLastError());
  }
  uint64_t* pin = host_words();
  MG_HIP(hipMemcpyAsync(pin, d_bad, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  if (pin[0]) {
    db->failed = MG_ERR_ARG;
    return fail(MG_ERR_ARG, "reference-pipeline table is corrupt: %llu prefix numbers / genome ids out of range or out of order",
                (unsigned long long)pin[0]);
  }
  db->unchecked = false;
  return MG_OK;
}

}  // namespace mg

mg_refdb::~mg_refdb() {
  if (pending) mg::upload_ranges_abort(pending);  // (joins the uploader's threads before the buffers go)
  pending = nullptr;
}

using namespace mg;

extern "C" {

int mg_refdb_build(const uint64_t* hashes, const uint64_t* kmer_hi, const uint64_t* kmer_lo, const uint64_t* offsets, uint64_t ngenomes,
                   int nk, const int* ks, mg_refdb** out) {
  MG_REQUIRE_READY();
  if (!out || !offsets || !ks) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  if (nk < 1 || nk > 4) return fail(MG_ERR_ARG, "between 1 and 4 k per table");
  for (int i = 0; i < nk; ++i)
    if (ks[i] < 1 || ks[i] > MG_MAX_K || (i > 0 && ks[i] <= ks[i - 1])) return fail(MG_ERR_ARG, "ks must ascend within [1, %d]", MG_MAX_K);
  if (offsets[0] != 0) return fail(MG_ERR_ARG, "offsets must start at 0");
  const uint64_t E = offsets[ngenomes];
  if (E > 0xfffffff0ull || ngenomes > 0xfffffff0ull) return fail(MG_ERR_ARG, "table too large for 32-bit positions");
  if (E && (!hashes || !kmer_hi || !kmer_lo)) return fail(MG_ERR_ARG, "null argument");
  const int kmax = ks[nk - 1];
  Context& c = ctx();
  hipStream_t st = c.stream;
  std::unique_ptr<mg_refdb> db(new mg_refdb());
  db->nk = nk;
  for (int i = 0; i < nk; ++i) db->ks[i] = ks[i];
  mg_db& T = db->kmax;
  T.ngenomes = ngenomes;
  T.total = E;
  uint64_t mx = 0;
  for (uint64_t g = 0; g < ngenomes; ++g) {
    if (offsets[g + 1] < offsets[g]) return fail(MG_ERR_ARG, "offsets must be non-decreasing");
  }
  // (every entry: the entries of a genome selected by the forward hash, mg_sketch_genomes_kmers_forward, do not ascend)
  for (uint64_t e = 0; e < E; ++e)
    if (hashes[e] > mx) mx = hashes[e];
  T.max_hash = mx;
  MG_TRY(T.offsets.alloc((ngenomes + 1) * sizeof(uint64_t)));
  MG_TRY(T.pair_hash.alloc((E + 1) * sizeof(uint64_t)));
  MG_TRY(T.pair_gen.alloc((E + 1) * sizeof(uint32_t)));
  MG_TRY(T.gsize.alloc((ngenomes + 1) * sizeof(uint32_t)));
  MG_TRY(db->kmer_hi.alloc((E + 1) * sizeof(uint64_t)));
  MG_TRY(db->kmer_lo.alloc((E + 1) * sizeof(uint64_t)));
  MG_HIP(hipMemcpyAsync(T.offsets.p, offsets, (ngenomes + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  if (ngenomes) hipLaunchKernelGGL(k_rp_gsizes, dim3(g256(ngenomes)), dim3(256), 0, st, T.offsets.as<uint64_t>(), ngenomes, T.gsize.as<uint32_t>());
  DevBuf h_in, khi_in, klo_in, iota, perm, a_hi, a_lo, b_hi, b_lo, s_lo, s_hi, ord1, ord2, g_hi, flag, before, d_hi, d_lo, ckey, cks;
  if (E) {
    MG_TRY(h_in.alloc(E * 8)); MG_TRY(khi_in.alloc(E * 8)); MG_TRY(klo_in.alloc(E * 8));
    MG_TRY(iota.alloc(E * 4)); MG_TRY(perm.alloc(E * 4));
    MG_HIP(hipMemcpyAsync(h_in.p, hashes, E * 8, hipMemcpyHostToDevice, st));
    MG_HIP(hipMemcpyAsync(khi_in.p, kmer_hi, E * 8, hipMemcpyHostToDevice, st));
    MG_HIP(hipMemcpyAsync(klo_in.p, kmer_lo, E * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_rp_iota, dim3(g256(E)), dim3(256), 0, st, iota.as<uint32_t>(), E);
    // hash-major: a stable sort by hash keeps equal hashes in genome order
    MG_TRY(sort_pairs(h_in.as<uint64_t>(), T.pair_hash.as<uint64_t>(), iota.as<uint32_t>(), perm.as<uint32_t>(), E));
    hipLaunchKernelGGL(k_rp_pairs, dim3(g256(E)), dim3(256), 0, st, perm.as<uint32_t>(), E, T.offsets.as<uint64_t>(), ngenomes,
                       khi_in.as<uint64_t>(), klo_in.as<uint64_t>(), T.pair_gen.as<uint32_t>(), db->kmer_hi.as<uint64_t>(),
                       db->kmer_lo.as<uint64_t>());
    MG_HIP(hipGetLastError());
    h_in.release(); khi_in.release(); klo_in.release(); perm.release();
    MG_TRY(a_hi.alloc(E * 8)); MG_TRY(a_lo.alloc(E * 8)); MG_TRY(b_hi.alloc(E * 8)); MG_TRY(b_lo.alloc(E * 8));
    MG_TRY(s_lo.alloc(E * 8)); MG_TRY(s_hi.alloc(E * 8)); MG_TRY(g_hi.alloc(E * 8));
    MG_TRY(ord1.alloc(E * 4)); MG_TRY(ord2.alloc(E * 4)); MG_TRY(flag.alloc(E * 4)); MG_TRY(before.alloc((E + 2) * 8));
    MG_TRY(d_hi.alloc(E * 8)); MG_TRY(d_lo.alloc(E * 8)); MG_TRY(ckey.alloc(E * 8)); MG_TRY(cks.alloc(E * 8));
  }
  for (int s = 0; s < nk - 1; ++s) {
    mg_refdb::Small& S = db->small[s];
    const int k = ks[s];
    MG_TRY(S.pa.alloc((E + 1) * 4));
    MG_TRY(S.pb.alloc((E + 1) * 4));
    MG_TRY(S.gsize.alloc((ngenomes + 1) * 4));
    MG_HIP(hipMemsetAsync(S.gsize.p, 0, (ngenomes + 1) * 4, st));
    if (!E) { MG_TRY(S.cid.alloc(4)); MG_TRY(S.cgen.alloc(4)); continue; }
    hipLaunchKernelGGL(k_rp_prefix_keys, dim3(g256(E)), dim3(256), 0, st, db->kmer_hi.as<uint64_t>(), db->kmer_lo.as<uint64_t>(), E, kmax, k,
                       a_hi.as<uint64_t>(), a_lo.as<uint64_t>(), b_hi.as<uint64_t>(), b_lo.as<uint64_t>());
    // D_k: the A keys ascending — low word first, then (k > 32) a stable pass over the high word
    MG_TRY(sort_pairs(a_lo.as<uint64_t>(), s_lo.as<uint64_t>(), iota.as<uint32_t>(), ord1.as<uint32_t>(), E));
    const uint64_t* srt_hi = nullptr;
    const uint32_t* order = ord1.as<uint32_t>();
    if (k > 32) {
      hipLaunchKernelGGL(k_rp_gather_u64, dim3(g256(E)), dim3(256), 0, st, a_hi.as<uint64_t>(), ord1.as<uint32_t>(), E, g_hi.as<uint64_t>());
      MG_TRY(sort_pairs(g_hi.as<uint64_t>(), s_hi.as<uint64_t>(), ord1.as<uint32_t>(), ord2.as<uint32_t>(), E));
      hipLaunchKernelGGL(k_rp_gather_u64, dim3(g256(E)), dim3(256), 0, st, a_lo.as<uint64_t>(), ord2.as<uint32_t>(), E, s_lo.as<uint64_t>());
      srt_hi = s_hi.as<uint64_t>();
      order = ord2.as<uint32_t>();
    }
    hipLaunchKernelGGL(k_rp_heads128, dim3(g256(E)), dim3(256), 0, st, srt_hi, s_lo.as<uint64_t>(), E, flag.as<uint32_t>());
    MG_HIP(hipGetLastError());
    uint64_t nd = 0;
    MG_TRY(exclusive_sum_u32_to_u64(flag.as<uint32_t>(), before.as<uint64_t>(), E, &nd));
    S.nprefix = nd;
    hipLaunchKernelGGL(k_rp_number, dim3(g256(E)), dim3(256), 0, st, srt_hi, s_lo.as<uint64_t>(), order, flag.as<uint32_t>(),
                       before.as<uint64_t>(), E, S.pa.as<uint32_t>(), d_hi.as<uint64_t>(), d_lo.as<uint64_t>());
    hipLaunchKernelGGL(k_rp_lookup, dim3(g256(E)), dim3(256), 0, st, b_hi.as<uint64_t>(), b_lo.as<uint64_t>(), E, d_hi.as<uint64_t>(),
                       d_lo.as<uint64_t>(), nd, S.pb.as<uint32_t>());
    // the count list: distinct (prefix number, genome), ascending
    hipLaunchKernelGGL(k_rp_count_keys, dim3(g256(E)), dim3(256), 0, st, S.pa.as<uint32_t>(), T.pair_gen.as<uint32_t>(), E, ckey.as<uint64_t>());
    MG_TRY(sort_keys(ckey.as<uint64_t>(), cks.as<uint64_t>(), E, 64));
    hipLaunchKernelGGL(k_rp_heads128, dim3(g256(E)), dim3(256), 0, st, (const uint64_t*)nullptr, cks.as<uint64_t>(), E, flag.as<uint32_t>());
    MG_HIP(hipGetLastError());
    uint64_t nc = 0;
    MG_TRY(exclusive_sum_u32_to_u64(flag.as<uint32_t>(), before.as<uint64_t>(), E, &nc));
    S.ncount = nc;
    MG_TRY(S.cid.alloc((nc + 1) * 4));
    MG_TRY(S.cgen.alloc((nc + 1) * 4));
    hipLaunchKernelGGL(k_rp_count_emit, dim3(g256(E)), dim3(256), 0, st, cks.as<uint64_t>(), flag.as<uint32_t>(), before.as<uint64_t>(), E,
                       S.cid.as<uint32_t>(), S.cgen.as<uint32_t>(), S.gsize.as<uint32_t>());
    MG_HIP(hipGetLastError());
  }
  MG_TRY(alloc_marks(db.get()));
  MG_HIP(hipStreamSynchronize(st));
  *out = db.release();
  return MG_OK;
}

int mg_refdb_upload_begin(uint64_t ngenomes, int nk, const int* ks, uint64_t npairs, const uint64_t* pair_hash, const uint32_t* pair_gen,
                          const uint32_t* gsize_kmax, uint64_t max_hash, const uint32_t* const* pa, const uint32_t* const* pb,
                          const uint64_t* nprefix, const uint32_t* const* cid, const uint32_t* const* cgen, const uint64_t* ncount,
                          const uint32_t* const* gsize, mg_refdb** out) {
  MG_REQUIRE_READY();
  if (!out || !ks || !gsize_kmax || (npairs && (!pair_hash || !pair_gen))) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  if (nk < 1 || nk > 4) return fail(MG_ERR_ARG, "between 1 and 4 k per table");
  if (nk > 1 && (!pa || !pb || !nprefix || !cid || !cgen || !ncount || !gsize)) return fail(MG_ERR_ARG, "null argument");
  if (npairs > 0xfffffff0ull) return fail(MG_ERR_ARG, "sketch table of %llu hashes exceeds the 32-bit position range", (unsigned long long)npairs);
  if (ngenomes > 0xfffffff0ull) return fail(MG_ERR_ARG, "too many genomes");
  if (npairs && (pair_hash[0] > pair_hash[npairs - 1] || pair_hash[npairs - 1] > max_hash))
    return fail(MG_ERR_ARG, "pair list is not ascending within [0, max_hash]");
  std::unique_ptr<mg_refdb> db(new mg_refdb());
  db->nk = nk;
  for (int i = 0; i < nk; ++i) db->ks[i] = ks[i];
  db->kmax.ngenomes = ngenomes;
  db->kmax.total = npairs;
  db->kmax.max_hash = max_hash;
  MG_TRY(db->kmax.pair_hash.alloc((npairs + 1) * sizeof(uint64_t)));
  MG_TRY(db->kmax.pair_gen.alloc((npairs + 1) * sizeof(uint32_t)));
  MG_TRY(db->kmax.gsize.alloc((ngenomes + 1) * sizeof(uint32_t)));
  std::vector<std::pair<const void*, std::pair<void*, uint64_t>>> up;
  if (npairs) { up.push_back({pair_hash, {db->kmax.pair_hash.p, npairs * 8}}); up.push_back({pair_gen, {db->kmax.pair_gen.p, npairs * 4}}); }
  if (ngenomes) up.push_back({gsize_kmax, {db->kmax.gsize.p, ngenomes * 4}});
  for (int s = 0; s < nk - 1; ++s) {
    mg_refdb::Small& S = db->small[s];
    if (nprefix[s] > 0xfffffff0ull || ncount[s] > 0xfffffff0ull) return fail(MG_ERR_ARG, "table too large for 32-bit positions");
    if ((npairs && (!pa[s] || !pb[s])) || (ncount[s] && (!cid[s] || !cgen[s])) || !gsize[s]) return fail(MG_ERR_ARG, "null argument");
    S.nprefix = nprefix[s];
    S.ncount = ncount[s];
    MG_TRY(S.pa.alloc((npairs + 1) * 4)); MG_TRY(S.pb.alloc((npairs + 1) * 4));
    MG_TRY(S.cid.alloc((ncount[s] + 1) * 4)); MG_TRY(S.cgen.alloc((ncount[s] + 1) * 4));
    MG_TRY(S.gsize.alloc((ngenomes + 1) * 4));
    if (npairs) { up.push_back({pa[s], {S.pa.p, npairs * 4}}); up.push_back({pb[s], {S.pb.p, npairs * 4}}); }
    if (ncount[s]) { up.push_back({cid[s], {S.cid.p, ncount[s] * 4}}); up.push_back({cgen[s], {S.cgen.p, ncount[s] * 4}}); }
    if (ngenomes) up.push_back({gsize[s], {S.gsize.p, ngenomes * 4}});
  }
  MG_TRY(alloc_marks(db.get()));
  // (the buffers may have been other kernels' a moment ago: nothing of the copy stream may land before the library stream got here)
  MG_HIP(hipStreamSynchronize(ctx().stream));
  db->unchecked = true;
  MG_TRY(upload_ranges_begin(up, &db->pending));
  *out = db.release();
  return MG_OK;
}

int mg_refdb_upload(uint64_t ngenomes, int nk, const int* ks, uint64_t npairs, const uint64_t* pair_hash, const uint32_t* pair_gen,
                    const uint32_t* gsize_kmax, uint64_t max_hash, const uint32_t* const* pa, const uint32_t* const* pb,
                    const uint64_t* nprefix, const uint32_t* const* cid, const uint32_t* const* cgen, const uint64_t* ncount,
                    const uint32_t* const* gsize, mg_refdb** out) {
  mg_refdb* db = nullptr;
  MG_TRY(mg_refdb_upload_begin(ngenomes, nk, ks, npairs, pair_hash, pair_gen, gsize_kmax, max_hash, pa, pb, nprefix, cid, cgen, ncount, gsize, &db));
  const int rc = refdb_ready(db);
  if (rc != MG_OK) { delete db; return rc; }
  *out = db;
  return MG_OK;
}

int mg_refdb_sizes(const mg_refdb* db, uint64_t* npairs, uint64_t* nprefix, uint64_t* ncount) {
  if (!db) return fail(MG_ERR_ARG, "null argument");
  if (npairs) *npairs = db->kmax.total;
  for (int s = 0; s < db->nk - 1; ++s) {
    if (nprefix) nprefix[s] = db->small[s].nprefix;
    if (ncount) ncount[s] = db->small[s].ncount;
  }
  return MG_OK;
}

int mg_refdb_download_kmax(const mg_refdb* db, uint64_t* pair_hash, uint32_t* pair_gen, uint32_t* gsize, uint64_t* kmer_hi, uint64_t* kmer_lo) {
  MG_REQUIRE_READY();
  if (!db) return fail(MG_ERR_ARG, "null argument");
  MG_TRY(refdb_ready(db));
  const uint64_t n = db->kmax.total, G = db->kmax.ngenomes;
  if (pair_hash && n) MG_TRY(mg_memcpy_d2h(pair_hash, db->kmax.pair_hash.p, n * 8));
  if (pair_gen && n) MG_TRY(mg_memcpy_d2h(pair_gen, db->kmax.pair_gen.p, n * 4));
  if (gsize && G) MG_TRY(mg_memcpy_d2h(gsize, db->kmax.gsize.p, G * 4));
  if ((kmer_hi || kmer_lo) && n && !db->kmer_hi.p) return fail(MG_ERR_STATE, "an uploaded table does not hold its k-mers");
  if (kmer_hi && n) MG_TRY(mg_memcpy_d2h(kmer_hi, db->kmer_hi.p, n * 8));
  if (kmer_lo && n) MG_TRY(mg_memcpy_d2h(kmer_lo, db->kmer_lo.p, n * 8));
  return MG_OK;
}

int mg_refdb_download_k(const mg_refdb* db, int ki, uint32_t* pa, uint32_t* pb, uint32_t* cid, uint32_t* cgen, uint32_t* gsize) {
  MG_REQUIRE_READY();
  if (!db || ki < 0 || ki >= db->nk - 1) return fail(MG_ERR_ARG, "no such k below the largest");
  MG_TRY(refdb_ready(db));
  const mg_refdb::Small& S = db->small[ki];
  const uint64_t n = db->kmax.total, G = db->kmax.ngenomes;
  if (pa && n) MG_TRY(mg_memcpy_d2h(pa, S.pa.p, n * 4));
  if (pb && n) MG_TRY(mg_memcpy_d2h(pb, S.pb.p, n * 4));
  if (cid && S.ncount) MG_TRY(mg_memcpy_d2h(cid, S.cid.p, S.ncount * 4));
  if (cgen && S.ncount) MG_TRY(mg_memcpy_d2h(cgen, S.cgen.p, S.ncount * 4));
  if (gsize && G) MG_TRY(mg_memcpy_d2h(gsize, S.gsize.p, G * 4));
  return MG_OK;
}

int mg_refdb_marks(const mg_refdb* db, int ki, uint32_t** d_marks, uint64_t* nwords) {
  if (!db || ki < 0 || ki >= db->nk - 1 || !d_marks || !nwords) return fail(MG_ERR_ARG, "no such k below the largest");
  *d_marks = db->marks.as<uint32_t>() + db->small[ki].marks_at;
  *nwords = db->small[ki].marks_n;
  return MG_OK;
}

int mg_refdb_nk(const mg_refdb* db) { return db ? db->nk : 0; }
uint64_t mg_refdb_ngenomes(const mg_refdb* db) { return db ? db->kmax.ngenomes : 0; }
uint64_t mg_refdb_max_hash(const mg_refdb* db) { return db ? db->kmax.max_hash : 0; }
const mg_db* mg_refdb_kmax_table(const mg_refdb* db) {  // (what the caller does with it reads the pairs: they have to be there, and checked)
  if (!db || refdb_ready(db) != MG_OK) return nullptr;
  return &db->kmax;
}
void mg_refdb_free(mg_refdb* db) { delete db; }

}  // extern "C"
