// mg_profile.hip — Stage C: per-read taxon assignment + abundance histogram.
//
// Replaces the loop of map_and_process (scripts/map_and_profile.py:193-264).
// The reference loop carries one bit across read boundaries: when a read is
// judged Ambiguous the `continue` at :232 skips the append at :257-259, so the
// first line of the NEXT read is dropped.  Whether read g is Ambiguous depends
// on whether its own first line was dropped, so read g is a map
//     A_g : {first line kept, first line dropped} -> {next kept, next dropped}
// and the true state of every read is the prefix composition of these maps
// (associative, not commutative) started from "dropped" (the phantom first
// boundary, :155-156).  Pipeline:
//   k_profile_maps     per record: leaders evaluate A_g(0), A_g(1); per-block
//                      ordered composition (wave shuffles + LDS)
//   k_profile_scan     one block: exclusive composition over block aggregates
//   k_profile_commit   per record: in-block ordered scan gives each leader its
//                      true state; the read is classified once; unique reads go
//                      to an LDS-privatised histogram (count, bases, first
//                      seen) flushed with global atomics; multimapped reads
//                      record their list length
//   k_profile_scan_mm  one block: exclusive sums of the per-tile multimapped totals
//   k_profile_fill_mm  writes the multimapped CSR (lists in SAM order)
// Records are 16 B and are streamed with coalesced 16-byte loads.
#include <memory>

#include "mg_internal.h"

namespace mg {

constexpr int kPB = 256;  // threads per block == records per tile

struct Verdict {
  uint32_t kind;  // 0 Ambiguous, 1 unique, 2 multimapped
  uint32_t tax;
  uint64_t hitlen;
  uint32_t nmm;
};

__device__ __forceinline__ bool rec_pair1(uint32_t flag) { return (flag & 1u) && (flag & 64u); }   // :106
__device__ __forceinline__ bool rec_pair2(uint32_t flag) { return (flag & 1u) && (flag & 128u); }  // :107

// filter_line (:86-100) or chimeric (:108,135)
__device__ __forceinline__ bool rec_rejected(const mg_aln_rec& r, double pct_id) {
  return ((double)r.matched / (double)r.total < pct_id) || (r.flag_len & 2048u);
}

// process_read (:152-176) over the lines [s, e); next_flag = FLAG of the line that closes the read
// (its pair bits are the `pair1, pair2` passed at :225-226).  mm_out: where to write the taxon list
// of a multimapped read (nullptr = only count it).
__device__ Verdict eval_group(const mg_aln_rec* __restrict__ recs, uint64_t s, uint64_t e, uint32_t next_flag,
                              const uint32_t* __restrict__ ref2tax, double pct_id, uint32_t* mm_out) {
  Verdict v{0, 0, 0, 0};
  long p1 = 0, p2 = 0;
  uint64_t nk = 0;
  for (uint64_t i = s; i < e; ++i) {  // appends (:257-258) then clean_read_hits (:130-147)
    const mg_aln_rec r = recs[i];
    const uint32_t fl = r.flag_len & MG_REC_FLAG_MASK;
    const bool a = rec_pair1(fl), b = rec_pair2(fl);
    p1 += (a || !(a || b)) ? 1 : 0;
    p2 += b ? 1 : 0;
    if (rec_rejected(r, pct_id)) {
      if (a) p1 -= 1; else if (b) p2 -= 1;
    } else {
      if (nk == 0) v.tax = ref2tax[r.ref_new & MG_REC_REF_MASK];
      ++nk;
    }
    v.hitlen += r.flag_len >> MG_REC_LEN_SHIFT;
  }
  if (nk == 0) return v;  // :155-156
  auto kept = [&](uint64_t i) { return !rec_rejected(recs[i], pct_id); };
  auto tax = [&](uint64_t i) { return ref2tax[recs[i].ref_new & MG_REC_REF_MASK]; };
  auto emit = [&](uint32_t t) {
    if (mm_out) mm_out[v.nmm] = t;
    ++v.nmm;
  };
  if (rec_pair1(next_flag) || rec_pair2(next_flag)) {  // :157
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

// State maps: bit0 = f(0), bit1 = f(1); x = 1 means "first line dropped".
constexpr uint32_t kIdentity = 2u;
__device__ __forceinline__ uint32_t map_then(uint32_t first, uint32_t second) {
  return ((second >> (first & 1u)) & 1u) | (((second >> ((first >> 1) & 1u)) & 1u) << 1);
}

// End of the group led by record i: index of the next record with the new-read bit, or ntotal.
__device__ __forceinline__ uint64_t group_end(const mg_aln_rec* __restrict__ recs, uint64_t i, uint64_t ntotal) {
  uint64_t j = i + 1;
  while (j < ntotal && !(recs[j].ref_new & MG_REC_NEW_BIT)) ++j;
  return j;
}

// Ordered inclusive scan of maps across the block; returns the EXCLUSIVE prefix for this thread
// and the block aggregate in *total.  lds: 4 words.
__device__ __forceinline__ uint32_t block_scan_maps(uint32_t m, uint32_t* lds, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = m;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t prev = __shfl_up(inc, o, 64);
    if (lane >= o) inc = map_then(prev, inc);
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t before = kIdentity, all = kIdentity;
#pragma unroll
  for (int w = 0; w < kPB / 64; ++w) {
    if (w < wave) before = map_then(before, lds[w]);
    all = map_then(all, lds[w]);
  }
  uint32_t excl = __shfl_up(inc, 1, 64);
  if (lane == 0) excl = kIdentity;
  __syncthreads();
  *total = all;
  return map_then(before, excl);
}

// Exclusive count of set flags before this thread within the block, and the block total.
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t* lds, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  if (lane == 0) lds[wave] = __popcll(m);
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kPB / 64; ++w) {
    if (w < wave) before += lds[w];
    all += lds[w];
  }
  __syncthreads();
  *total = all;
  return before + __popcll(m & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(kPB) void k_profile_maps(const mg_aln_rec* __restrict__ recs, uint64_t nrecs,
                                                      uint64_t ntotal, const uint32_t* __restrict__ ref2tax,
                                                      double pct_id, uint8_t* __restrict__ maps,
                                                      uint8_t* __restrict__ blk_map, uint32_t* __restrict__ blk_groups) {
  __shared__ uint32_t lds[8];
  const uint64_t i = (uint64_t)blockIdx.x * kPB + threadIdx.x;
  uint32_t m = kIdentity;
  bool leader = false;
  if (i < nrecs && (recs[i].ref_new & MG_REC_NEW_BIT)) {
    leader = true;
    const uint64_t e = group_end(recs, i, ntotal);
    if (e < ntotal) {  // a following boundary exists: this read IS processed (:225-226)
      const uint32_t nf = recs[e].flag_len & MG_REC_FLAG_MASK;
      const uint32_t a0 = eval_group(recs, i, e, nf, ref2tax, pct_id, nullptr).kind == 0;
      const uint32_t a1 = eval_group(recs, i + 1, e, nf, ref2tax, pct_id, nullptr).kind == 0;
      m = a0 | (a1 << 1);
    }
  }
  if (i < nrecs) maps[i] = (uint8_t)m;
  uint32_t total, ngroups;
  (void)block_scan_maps(m, lds, &total);
  (void)block_rank(leader, lds + 4, &ngroups);
  if (threadIdx.x == 0) { blk_map[blockIdx.x] = (uint8_t)total; blk_groups[blockIdx.x] = ngroups; }
}

// ---- helpers for the single-block (1024 threads) scans over per-tile aggregates ----
template <int NT>
__device__ __forceinline__ uint64_t blockN_excl_sum(uint64_t v, uint64_t* lds, uint64_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint64_t prev = __shfl_up(inc, o, 64);
    if (lane >= o) inc += prev;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint64_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const uint64_t t = lds[w];
    if (w < wave) before += t;
    all += t;
  }
  __syncthreads();
  *total = all;
  return before + inc - v;
}

template <int NT>
__device__ __forceinline__ uint32_t blockN_excl_map(uint32_t m, uint32_t* lds, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = m;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t prev = __shfl_up(inc, o, 64);
    if (lane >= o) inc = map_then(prev, inc);
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t before = kIdentity, all = kIdentity;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const uint32_t t = lds[w];
    if (w < wave) before = map_then(before, t);
    all = map_then(all, t);
  }
  uint32_t excl = __shfl_up(inc, 1, 64);
  if (lane == 0) excl = kIdentity;
  __syncthreads();
  *total = all;
  return map_then(before, excl);
}

// One block.  In: per-tile aggregates.  Out: exclusive prefixes per tile, totals in out_tot[0..1].
__global__ __launch_bounds__(1024) void k_profile_scan(const uint8_t* __restrict__ blk_map,
                                                       const uint32_t* __restrict__ blk_groups, uint64_t nblocks,
                                                       uint8_t* __restrict__ pre_map, uint64_t* __restrict__ pre_groups,
                                                       uint64_t* __restrict__ out_tot) {
  __shared__ uint32_t s_map[16];
  __shared__ uint64_t s_grp[16];
  const uint64_t per = (nblocks + 1023) / 1024;
  const uint64_t b0 = (uint64_t)threadIdx.x * per;
  const uint64_t b1 = b0 + per < nblocks ? b0 + per : nblocks;
  uint32_t m = kIdentity;
  uint64_t g = 0;
  for (uint64_t b = b0; b < b1; ++b) { m = map_then(m, blk_map[b]); g += blk_groups[b]; }
  uint32_t tot_m;
  uint64_t tot_g;
  m = blockN_excl_map<1024>(m, s_map, &tot_m);
  g = blockN_excl_sum<1024>(g, s_grp, &tot_g);
  if (threadIdx.x == 0) { out_tot[0] = tot_m; out_tot[1] = tot_g; }
  for (uint64_t b = b0; b < b1; ++b) {
    pre_map[b] = (uint8_t)m;
    pre_groups[b] = g;
    m = map_then(m, blk_map[b]);
    g += blk_groups[b];
  }
}

// One block: exclusive sums of the per-tile multimapped entry / read totals (in place), grand totals out.
__global__ __launch_bounds__(1024) void k_profile_scan_mm(uint64_t* __restrict__ tile_ent,
                                                          uint64_t* __restrict__ tile_reads, uint64_t ntiles,
                                                          uint64_t* __restrict__ out_tot,
                                                          uint64_t* __restrict__ mm_offsets) {
  __shared__ uint64_t s_a[16];
  __shared__ uint64_t s_b[16];
  const uint64_t per = (ntiles + 1023) / 1024;
  const uint64_t b0 = (uint64_t)threadIdx.x * per;
  const uint64_t b1 = b0 + per < ntiles ? b0 + per : ntiles;
  uint64_t e = 0, r = 0;
  for (uint64_t b = b0; b < b1; ++b) { e += tile_ent[b]; r += tile_reads[b]; }
  uint64_t tot_e, tot_r;
  e = blockN_excl_sum<1024>(e, s_a, &tot_e);
  r = blockN_excl_sum<1024>(r, s_b, &tot_r);
  if (threadIdx.x == 0) { out_tot[2] = tot_e; out_tot[3] = tot_r; mm_offsets[tot_r] = tot_e; }
  for (uint64_t b = b0; b < b1; ++b) {
    const uint64_t te = tile_ent[b], tr = tile_reads[b];
    tile_ent[b] = e;
    tile_reads[b] = r;
    e += te;
    r += tr;
  }
}

// Per-workgroup privatised histogram in LDS, flushed with global atomics at the end:
//   use_lds_hist == 1  direct:  ntax <= 2048 bins (dynamic LDS: 3 * ntax u64)
//   use_lds_hist == 2  hashed:  any ntax; kHashSlots open-addressed bins keyed by taxon id (a sample hits far
//                      fewer taxa than the table lists); a taxon that finds no bin within kHashProbe steps
//                      goes to global atomics directly.  Without this a skewed sample serialises millions of
//                      global atomics on a few hundred addresses (8.6 ms -> see DESIGN.md at 12.5 M records).
constexpr uint32_t kHashSlots = 2048, kHashProbe = 32;
__global__ __launch_bounds__(kPB) void k_profile_commit(
    const mg_aln_rec* __restrict__ recs, uint64_t nrecs, uint64_t ntotal, const uint32_t* __restrict__ ref2tax,
    double pct_id, uint8_t* __restrict__ maps, const uint8_t* __restrict__ pre_map,
    const uint64_t* __restrict__ pre_groups, uint32_t incoming, uint32_t first_shard, uint64_t group_base,
    uint32_t ntax, uint32_t use_lds_hist, unsigned long long* __restrict__ g_count,
    unsigned long long* __restrict__ g_bases, unsigned long long* __restrict__ g_first,
    unsigned long long* __restrict__ g_scalars, uint32_t* __restrict__ mm_cnt, uint64_t* __restrict__ tile_ent,
    uint64_t* __restrict__ tile_reads, uint64_t ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long hist[];
  __shared__ uint32_t lds[8];
  __shared__ uint64_t lds64[4];
  __shared__ unsigned long long s_ambig, s_groups;
  const uint32_t nbins = use_lds_hist == 2 ? kHashSlots : ntax;
  unsigned long long* h_count = hist;
  unsigned long long* h_bases = hist + nbins;
  unsigned long long* h_first = hist + 2 * (size_t)nbins;
  uint32_t* h_key = reinterpret_cast<uint32_t*>(hist + 3 * (size_t)nbins);  // hashed mode only
  if (use_lds_hist) {
    for (uint32_t t = threadIdx.x; t < nbins; t += kPB) {
      h_count[t] = 0; h_bases[t] = 0; h_first[t] = ~0ull;
      if (use_lds_hist == 2) h_key[t] = 0xffffffffu;
    }
  }
  if (threadIdx.x == 0) { s_ambig = (blockIdx.x == 0 && first_shard) ? 1ull : 0ull; s_groups = 0; }
  __syncthreads();
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint64_t i = tile * kPB + threadIdx.x;
    uint32_t m = kIdentity;
    bool leader = false;
    if (i < nrecs) {
      m = maps[i];
      leader = (recs[i].ref_new & MG_REC_NEW_BIT) != 0;
    }
    uint32_t blk_total, ngroups;
    const uint32_t excl = block_scan_maps(m, lds, &blk_total);
    const uint32_t rank = block_rank(leader, lds + 4, &ngroups);
    const uint32_t x_b = (pre_map[tile] >> incoming) & 1u;  // state at the start of this tile
    uint32_t my_cnt = 0;
    if (leader) {
      const uint32_t d = (excl >> x_b) & 1u;  // 1: this read's first line was dropped (:232)
      const uint64_t e = group_end(recs, i, ntotal);
      maps[i] = (uint8_t)d;
      if (e < ntotal) {
        const uint32_t nf = recs[e].flag_len & MG_REC_FLAG_MASK;
        const Verdict v = eval_group(recs, i + d, e, nf, ref2tax, pct_id, nullptr);
        if (v.kind == 0) {
          atomicAdd(&s_ambig, 1ull);
        } else if (v.kind == 1) {
          const unsigned long long gidx = group_base + pre_groups[tile] + rank;
          uint32_t bin = v.tax;
          bool in_lds = use_lds_hist == 1;
          if (use_lds_hist == 2) {
            uint32_t p = (v.tax * 2654435761u) >> 21;  // 11 bits: kHashSlots
            for (uint32_t step = 0; step < kHashProbe; ++step) {
              const uint32_t old = atomicCAS(&h_key[p], 0xffffffffu, v.tax);
              if (old == 0xffffffffu || old == v.tax) { bin = p; in_lds = true; break; }
              p = (p + 1) & (kHashSlots - 1);
            }
          }
          if (in_lds) {
            atomicAdd(&h_count[bin], 1ull);
            atomicAdd(&h_bases[bin], (unsigned long long)v.hitlen);
            atomicMin(&h_first[bin], gidx);
          } else {
            atomicAdd(&g_count[v.tax], 1ull);
            atomicAdd(&g_bases[v.tax], (unsigned long long)v.hitlen);
            atomicMin(&g_first[v.tax], gidx);
          }
        } else {
          my_cnt = v.nmm;
        }
      }
    }
    if (i < nrecs) mm_cnt[i] = my_cnt;
    uint64_t t_ent;
    uint32_t t_reads;
    (void)blockN_excl_sum<kPB>((uint64_t)my_cnt, lds64, &t_ent);
    (void)block_rank(my_cnt != 0, lds + 4, &t_reads);
    if (threadIdx.x == 0) { s_groups += ngroups; tile_ent[tile] = t_ent; tile_reads[tile] = t_reads; }
  }
  __syncthreads();
  if (use_lds_hist) {
    for (uint32_t t = threadIdx.x; t < nbins; t += kPB) {
      if (h_count[t]) {
        const uint32_t tax = use_lds_hist == 2 ? h_key[t] : t;
        atomicAdd(&g_count[tax], h_count[t]);
        atomicAdd(&g_bases[tax], h_bases[t]);
        atomicMin(&g_first[tax], h_first[t]);
      }
    }
  }
  if (threadIdx.x == 0) {
    if (s_groups) atomicAdd(&g_scalars[0], s_groups);
    if (s_ambig) atomicAdd(&g_scalars[1], s_ambig);
  }
}

__global__ __launch_bounds__(kPB) void k_profile_fill_mm(
    const mg_aln_rec* __restrict__ recs, uint64_t nrecs, uint64_t ntotal, const uint32_t* __restrict__ ref2tax,
    double pct_id, const uint8_t* __restrict__ dropped, const uint64_t* __restrict__ pre_groups, uint64_t group_base,
    const uint32_t* __restrict__ mm_cnt, const uint64_t* __restrict__ tile_ent, const uint64_t* __restrict__ tile_reads,
    uint64_t* __restrict__ mm_offsets, uint32_t* __restrict__ mm_tax, uint64_t* __restrict__ mm_hitlen,
    uint64_t* __restrict__ mm_read, uint64_t ntiles) {
  __shared__ uint32_t lds[8];
  __shared__ uint64_t lds64[4];
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint64_t i = tile * kPB + threadIdx.x;
    const bool leader = i < nrecs && (recs[i].ref_new & MG_REC_NEW_BIT);
    const uint32_t cnt = i < nrecs ? mm_cnt[i] : 0;
    uint32_t ngroups, nreads_t;
    uint64_t nent_t;
    const uint32_t rank = block_rank(leader, lds, &ngroups);
    const uint32_t slot_in = block_rank(cnt != 0, lds + 4, &nreads_t);
    const uint64_t ent_in = blockN_excl_sum<kPB>((uint64_t)cnt, lds64, &nent_t);
    if (cnt != 0) {
      const uint64_t e = group_end(recs, i, ntotal);
      const uint32_t nf = recs[e].flag_len & MG_REC_FLAG_MASK;
      const uint64_t eo = tile_ent[tile] + ent_in, so = tile_reads[tile] + slot_in;
      const Verdict v = eval_group(recs, i + dropped[i], e, nf, ref2tax, pct_id, mm_tax + eo);
      mm_offsets[so] = eo;
      mm_hitlen[so] = v.hitlen;
      mm_read[so] = group_base + pre_groups[tile] + rank;
    }
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
  uint64_t nblocks = 0;
  DevBuf maps, blk_map, blk_groups, pre_map, pre_groups, tot;
  uint8_t map[2] = {0, 1};
  uint64_t ngroups = 0;
  bool have_map = false;     // totals of pass A read back (lazily: a single shard never needs them)
  bool have_mm = false;      // multimapped totals read back (lazily)
  bool committed = false;
  // multimapped CSR (device)
  DevBuf mm_cnt, tile_ent, tile_reads, mm_offsets, mm_tax, mm_hitlen, mm_read;
  uint64_t mm_nreads = 0, mm_nentries = 0;
};

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
  p->nblocks = (nrecs + kPB - 1) / kPB;
  if (nrecs == 0) { p->map[0] = 0; p->map[1] = 1; *out = p.release(); return MG_OK; }
  Context& c = ctx();
  hipStream_t st = c.stream;
  MG_TRY(p->maps.alloc(nrecs));
  MG_TRY(p->blk_map.alloc(p->nblocks));
  MG_TRY(p->blk_groups.alloc(p->nblocks * sizeof(uint32_t)));
  MG_TRY(p->pre_map.alloc(p->nblocks));
  MG_TRY(p->pre_groups.alloc(p->nblocks * sizeof(uint64_t)));
  MG_TRY(p->tot.alloc(4 * sizeof(uint64_t)));
  {
    ProfScope ps("profile_maps");
    hipLaunchKernelGGL(k_profile_maps, dim3((unsigned)p->nblocks), dim3(kPB), 0, st, d_recs, nrecs, p->ntotal,
                       d_ref2tax, pct_id, p->maps.as<uint8_t>(), p->blk_map.as<uint8_t>(),
                       p->blk_groups.as<uint32_t>());
    MG_HIP(hipGetLastError());
  }
  {
    ProfScope ps("profile_scan");
    hipLaunchKernelGGL(k_profile_scan, dim3(1), dim3(1024), 0, st, p->blk_map.as<uint8_t>(),
                       p->blk_groups.as<uint32_t>(), p->nblocks, p->pre_map.as<uint8_t>(),
                       p->pre_groups.as<uint64_t>(), p->tot.as<uint64_t>());
    MG_HIP(hipGetLastError());
  }
  *out = p.release();
  return MG_OK;
}

// Pass-A totals (composed state map, read count) are fetched on first use.
static int fetch_map(mg_profile* p) {
  if (p->have_map || p->nrecs == 0) { p->have_map = true; return MG_OK; }
  hipStream_t st = ctx().stream;
  uint64_t* h_tot = host_words() + 16;
  MG_HIP(hipMemcpyAsync(h_tot, p->tot.p, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  p->map[0] = (uint8_t)(h_tot[0] & 1u);
  p->map[1] = (uint8_t)((h_tot[0] >> 1) & 1u);
  p->ngroups = h_tot[1];
  p->have_map = true;
  return MG_OK;
}

static int fetch_mm(mg_profile* p) {
  if (p->have_mm || p->nrecs == 0) { p->have_mm = true; return MG_OK; }
  hipStream_t st = ctx().stream;
  uint64_t* h_tot = host_words() + 16;
  MG_HIP(hipMemcpyAsync(h_tot + 2, p->tot.as<uint64_t>() + 2, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  p->mm_nentries = h_tot[2];
  p->mm_nreads = h_tot[3];
  p->have_mm = true;
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

int mg_profile_commit_dev(mg_profile* p, int incoming_dropped, int first_shard, uint64_t group_base, uint64_t* d_count,
                          uint64_t* d_bases, uint64_t* d_first_seen, uint64_t* d_scalars) {
  MG_REQUIRE_READY();
  if (!p || !d_count || !d_bases || !d_first_seen || !d_scalars) return fail(MG_ERR_ARG, "null argument");
  if (p->committed) return fail(MG_ERR_STATE, "profile shard already committed");
  Context& c = ctx();
  hipStream_t st = c.stream;
  p->committed = true;
  if (p->nrecs == 0) {
    if (first_shard) {
      // the phantom boundary never happens without a first line: nothing to add
    }
    return MG_OK;
  }
  MG_TRY(p->mm_cnt.alloc(p->nrecs * sizeof(uint32_t)));
  MG_TRY(p->tile_ent.alloc(p->nblocks * sizeof(uint64_t)));
  MG_TRY(p->tile_reads.alloc(p->nblocks * sizeof(uint64_t)));
  const uint32_t use_lds = p->ntax <= 2048 ? 1u : 2u;
  const size_t lds = use_lds == 1 ? 3 * (size_t)p->ntax * sizeof(unsigned long long)
                                  : kHashSlots * (3 * sizeof(unsigned long long) + sizeof(uint32_t));
  {
    ProfScope ps("profile_commit");
    unsigned grid = grid_for(p->nblocks, 1, (unsigned)c.num_cus * 4);
    hipLaunchKernelGGL(k_profile_commit, dim3(grid), dim3(kPB), lds, st, p->d_recs, p->nrecs, p->ntotal, p->d_ref2tax,
                       p->pct_id, p->maps.as<uint8_t>(), p->pre_map.as<uint8_t>(), p->pre_groups.as<uint64_t>(),
                       (uint32_t)(incoming_dropped ? 1 : 0), (uint32_t)(first_shard ? 1 : 0), group_base, p->ntax,
                       use_lds, (unsigned long long*)d_count, (unsigned long long*)d_bases,
                       (unsigned long long*)d_first_seen, (unsigned long long*)d_scalars, p->mm_cnt.as<uint32_t>(),
                       p->tile_ent.as<uint64_t>(), p->tile_reads.as<uint64_t>(), p->nblocks);
    MG_HIP(hipGetLastError());
  }
  // multimapped CSR buffers sized by their upper bounds (pooled), so that nothing has to be read back here:
  // entries <= records, multimapped reads <= records
  MG_TRY(p->mm_offsets.alloc((p->nrecs + 2) * sizeof(uint64_t)));
  MG_TRY(p->mm_tax.alloc((p->nrecs + 1) * sizeof(uint32_t)));
  MG_TRY(p->mm_hitlen.alloc((p->nrecs + 1) * sizeof(uint64_t)));
  MG_TRY(p->mm_read.alloc((p->nrecs + 1) * sizeof(uint64_t)));
  {
    ProfScope ps("profile_scan_mm");
    hipLaunchKernelGGL(k_profile_scan_mm, dim3(1), dim3(1024), 0, st, p->tile_ent.as<uint64_t>(),
                       p->tile_reads.as<uint64_t>(), p->nblocks, p->tot.as<uint64_t>(), p->mm_offsets.as<uint64_t>());
    MG_HIP(hipGetLastError());
  }
  {
    ProfScope ps("profile_fill_mm");
    unsigned grid = grid_for(p->nblocks, 1, (unsigned)c.num_cus * 8);
    hipLaunchKernelGGL(k_profile_fill_mm, dim3(grid), dim3(kPB), 0, st, p->d_recs, p->nrecs, p->ntotal, p->d_ref2tax,
                       p->pct_id, p->maps.as<uint8_t>(), p->pre_groups.as<uint64_t>(), group_base,
                       p->mm_cnt.as<uint32_t>(), p->tile_ent.as<uint64_t>(), p->tile_reads.as<uint64_t>(),
                       p->mm_offsets.as<uint64_t>(), p->mm_tax.as<uint32_t>(), p->mm_hitlen.as<uint64_t>(),
                       p->mm_read.as<uint64_t>(), p->nblocks);
    MG_HIP(hipGetLastError());
  }
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
  hipStream_t st = ctx().stream;
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
