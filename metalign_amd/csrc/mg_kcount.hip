// mg_kcount.hip — stage A of the reference pipeline by K-MER IDENTITY: k_count_kmers<K> and the table-side index it reads.
//
// Replaces `kmc -k<kmax> -ci2 -cs3` + `kmc_tools simple ... intersect` (scripts/select_db.py:50-59) without hashing a single read
// position: the design, the definitions and every per-lane piece are in mg_kcount_core.h; here are
//   * the index over the table's distinct canonical k_max-mers (mg_refdb_index_kmers): an entry per k-mer and hash it is filed
//     under (the 19 bases around a smallest-rank candidate, hashed: nearly always one), buckets of four entries = one 128-byte line
//     by the hash's low bits with an overflow list, a gate bitmap over the hash's leading bits (and the bitmap of the bits two hashes
//     share), and per pair of the hash-major table the pair that counts for its k-mer (`head`: equal k-mers are adjacent in hash
//     order, so stage B reads counts[head[i]] nearly sequentially);
//   * the kernel: one wavefront per tile of 64 reads, one lane per read.  The tile's bases go HBM -> LDS with 16-byte coalesced
//     loads, packed to 2 bits per base on the way (a big-endian stream: 32 bits at any base offset are sixteen bases as k-mers
//     compare); every lane slides the minimizer over its read (kc_walk) and leaves the runs it closes in a list of its own in
//     LDS; kc_drain — ONE copy of the cold code, called — takes the lists through the gate, compacts the runs that pass
//     (ballot + popcount) and matches them 64 at a time against their buckets;
//   * the handle that holds a sample's counters (mg_kcounts) and its way into stage B (mg_contain.hip: k_match_pairs).
// Normative statement: oracle/mg_oracle.c, mgo_refpipe_count_kmers.
#include <algorithm>
#include <memory>

#include "mg_internal.h"
#include "mg_kcount_core.h"
#include "mg_sketch_dev.h"

namespace mg {

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// the index
// ---------------------------------------------------------------------------------------------------------------------
inline unsigned g256(uint64_t n) { return grid_for(n ? n : 1, 256, (unsigned)ctx().num_cus * 8); }
#define KC_FOR(i, n) for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (uint64_t)gridDim.x * blockDim.x)

// pair i: its kept k-mer (right-aligned) -> the canonical strand, left-aligned, as two words that compare like the k-mer
__global__ void k_kc_canon(const uint64_t* __restrict__ khi, const uint64_t* __restrict__ klo, uint64_t n, int k,
                           uint64_t* __restrict__ chi, uint64_t* __restrict__ clo, uint32_t* __restrict__ iota) {
  KC_FOR(i, n) {
    const KcWin c = kc_canonical(kc_from_right(khi[i], klo[i], k), k);
    chi[i] = ((uint64_t)c.w[0] << 32) | c.w[1];
    clo[i] = ((uint64_t)c.w[2] << 32) | c.w[3];
    iota[i] = (uint32_t)i;
  }
}
__global__ void k_kc_gather(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n, uint64_t* __restrict__ dst) {
  KC_FOR(i, n) dst[i] = src[idx[i]];
}
__global__ void k_kc_heads(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint64_t n, uint32_t* __restrict__ flag) {
  KC_FOR(j, n) flag[j] = (j == 0 || hi[j] != hi[j - 1] || lo[j] != lo[j - 1]) ? 1u : 0u;
}
// sorted position j opens distinct k-mer number before[j]: its words, the pair that counts for it (the sort is stable and
// started from pair order: the first of a group is its lowest pair), how many hashes it is filed under
__global__ void k_kc_distinct(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, const uint32_t* __restrict__ order,
                              const uint32_t* __restrict__ flag, const uint64_t* __restrict__ before, uint64_t n, int k,
                              uint64_t* __restrict__ dhi, uint64_t* __restrict__ dlo, uint32_t* __restrict__ dhead,
                              uint32_t* __restrict__ nkeys) {
  KC_FOR(j, n) {
    if (!flag[j]) continue;
    const uint64_t id = before[j];
    const KcWin x{{(uint32_t)(hi[j] >> 32), (uint32_t)hi[j], (uint32_t)(lo[j] >> 32), (uint32_t)lo[j]}};
    dhi[id] = hi[j];
    dlo[id] = lo[j];
    dhead[id] = order[j];
    uint32_t keys[kKcMaxCands];
    nkeys[id] = (uint32_t)kc_table_keys(x, k, keys);  // (entries it will have: one per hash it is filed under)
  }
}
// distinct k-mer id -> its entries' sort keys (bucket = the hash's low bits, above the hash) and itself, from kbefore[id] on
__global__ void k_kc_emit_keys(const uint64_t* __restrict__ dhi, const uint64_t* __restrict__ dlo, const uint64_t* __restrict__ kbefore,
                               uint64_t nd, int k, uint32_t bmask, uint64_t* __restrict__ ekey, uint32_t* __restrict__ eid) {
  KC_FOR(id, nd) {
    const KcWin x{{(uint32_t)(dhi[id] >> 32), (uint32_t)dhi[id], (uint32_t)(dlo[id] >> 32), (uint32_t)dlo[id]}};
    uint32_t keys[kKcMaxCands];
    const int n = kc_table_keys(x, k, keys);
    for (int t = 0; t < n; ++t) {
      ekey[kbefore[id] + (uint64_t)t] = ((uint64_t)(keys[t] & bmask) << 32) | keys[t];
      eid[kbefore[id] + (uint64_t)t] = (uint32_t)id;
    }
  }
}
__global__ void k_kc_pair_heads(const uint32_t* __restrict__ order, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ before,
                                const uint32_t* __restrict__ dhead, uint64_t n, uint32_t* __restrict__ head) {
  KC_FOR(j, n) head[order[j]] = dhead[before[j] + flag[j] - 1];
}
// entry j = distinct k-mer perm[j] under the hash in skey[j] (ascending by bucket, then hash); the hash's gate bit — and, when a
// DIFFERENT hash has set it before, its `shared` bit (equal hashes are adjacent: the first of them speaks for all)
__global__ void k_kc_entries(const uint64_t* __restrict__ skey, const uint32_t* __restrict__ perm, const uint64_t* __restrict__ dhi,
                             const uint64_t* __restrict__ dlo, const uint32_t* __restrict__ dhead, uint64_t nd, int k, uint32_t gshift,
                             KcEntry* __restrict__ ent, uint32_t* __restrict__ gate, uint32_t* __restrict__ shared) {
  KC_FOR(j, nd) {
    const uint32_t id = perm[j];
    const uint64_t h = dhi[id], l = dlo[id];
    KcEntry e;
    e.w[0] = (uint32_t)(h >> 32); e.w[1] = (uint32_t)h; e.w[2] = (uint32_t)(l >> 32); e.w[3] = (uint32_t)l;
    e.head = dhead[id];
    e.key = (uint32_t)skey[j];  // (the low word of the sort key: the hash)
    e.off = kc_table_key_offset(KcWin{{e.w[0], e.w[1], e.w[2], e.w[3]}}, k, e.key);
    e.pad = 0;
    ent[j] = e;
    if (j == 0 || skey[j - 1] != skey[j]) {
      const uint32_t g = e.key >> gshift, bit = 1u << (g & 31u);
      if (atomicOr(&gate[g >> 5], bit) & bit) atomicOr(&shared[g >> 5], bit);
    }
  }
}
// offs[b] = first entry whose bucket (the high word of its sort key) is >= b; offs[nb] = nd
__global__ void k_kc_offsets(const uint64_t* __restrict__ skey, uint64_t nd, uint64_t nb, uint32_t* __restrict__ offs) {
  KC_FOR(j, nd + 1) {
    const uint64_t first = j == 0 ? 0 : (skey[j - 1] >> 32) + 1;
    const uint64_t last = j == nd ? nb : (skey[j] >> 32);
    for (uint64_t b = first; b <= last; ++b) offs[b] = (uint32_t)j;
  }
}
// bucket b (entries offs[b] .. offs[b + 1]) has this many entries beyond its kKcSlots slots in prim
__global__ void k_kc_extra(const uint32_t* __restrict__ offs, uint64_t nb, uint32_t* __restrict__ extra) {
  KC_FOR(b, nb) {
    const uint32_t n = offs[b + 1] - offs[b];
    extra[b] = n > kKcSlots ? n - kKcSlots : 0u;
  }
}
// ... its first kKcSlots go to prim (an unused slot: key kKcNone), the rest to ovf from before[b] on; the bucket's first slot
// says how many there are, its second where
__global__ void k_kc_place(const KcEntry* __restrict__ ent, const uint32_t* __restrict__ offs, const uint64_t* __restrict__ before,
                           uint64_t nb, KcEntry* __restrict__ prim, KcEntry* __restrict__ ovf) {
  KC_FOR(b, nb) {
    const uint32_t lo = offs[b], n = offs[b + 1] - lo;
    KcEntry none;
    none.w[0] = none.w[1] = none.w[2] = none.w[3] = 0; none.head = 0; none.key = kKcNone; none.off = 0; none.pad = 0;
    for (uint32_t t = 0; t < kKcSlots; ++t) {
      KcEntry e = t < n ? ent[lo + t] : none;
      e.pad = t == 0 ? (n > kKcSlots ? n - kKcSlots : 0u) : (t == 1 ? (uint32_t)before[b] : 0u);
      prim[kKcSlots * b + t] = e;
    }
    for (uint32_t t = kKcSlots; t < n; ++t) {
      KcEntry e = ent[lo + t];
      e.pad = 0;
      ovf[before[b] + (t - kKcSlots)] = e;
    }
  }
}
__global__ void k_kc_per_pair(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ head, uint64_t n, uint32_t cs,
                              uint32_t* __restrict__ out) {
  KC_FOR(i, n) {
    const uint32_t c = counts[head[i]];
    out[i] = (cs && c > cs) ? cs : c;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MG_KC_WAVES  // (A/B builds: tools/kcount_variants.sh)
#define MG_KC_WAVES 4
#endif
#ifndef MG_KC_LIST_CAP
#define MG_KC_LIST_CAP 12
#endif
constexpr int kKcWaves = MG_KC_WAVES;    // wavefronts per workgroup (each works alone)
constexpr uint32_t kKcListCap = MG_KC_LIST_CAP;  // closed runs a lane can hold before the lists are emptied (150 bp, k = 51: six per read)
#ifndef MG_KC_DIRECT
#define MG_KC_DIRECT 7
#endif
constexpr uint32_t kKcDirect = MG_KC_DIRECT;  // slots of a lane's list that the drain takes lane by lane; the rest are gathered (kc_drain)
constexpr uint32_t kKcHitCap = 128;      // runs past the gate waiting for their look-up (hitq, hitl): it comes when 64 or more wait
constexpr uint32_t kKcSlack = 8;         // dwords a k-mer taken at the end of the stream may read past it

// LDS of one wavefront (bytes), for a stage of sd dwords (sd a multiple of 64)
struct KcLds {
  uint32_t fwd, inv, p0s, stat, lists, hitq, hitl, total;
  __host__ __device__ explicit KcLds(uint32_t sd) {
    fwd = 0;
    inv = fwd + 4u * (sd + kKcSlack);
    p0s = inv + 4u * (sd / 2 + kKcSlack);
    stat = p0s + 4u * 64u;  // [0] runs, [1] runs past the gate, [2] matches of this wavefront so far; MG_KC_CLOCKS: [4..7] cycles / 64
    lists = stat + 4u * 12u;
    hitq = lists + 4u * 64u * (kKcListCap + 1u);
    hitl = hitq + 8u * kKcHitCap;
    total = hitl + 4u * kKcHitCap;
  }
};

struct KcArgs {
  const uint8_t* bases;
  const uint64_t* offsets;
  uint64_t nreads;
  uint32_t* live;
  const uint32_t* shared;
  const KcEntry* prim;
  const KcEntry* ovf;
  uint32_t* counts;
  uint32_t* csat;
  unsigned long long* stats;  // [0] k-mers of the reads, [1] runs, [2] runs past the sample's gate, [3] matches counted
  uint32_t gshift, bmask, sd, cs, ablate, stagger;  // ablate (knob kc_ablate, measurements only): 1 = the lists are dropped, 2 = ... after the gate
};

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#ifndef MG_KC_WAVES_PER_EU
#define MG_KC_WAVES_PER_EU 3
#endif
// The drain of the usual tile's walk is INLINED since the end of round 6 (the chunked path of long reads calls the one copy): around a
// call the kernel's live registers went to scratch and back — 0.7 GB of the launch's 1.17 GB written and 0.3 GB of its 5.8 GB fetched
// (WRITE_SIZE / FETCH_SIZE, 10M reads: 1.17 -> 0.44 GB, 5.82 -> 5.53 GB), 1.2 % of its time — for 20 KB more code per k.
#ifdef MG_KC_CALL_DRAIN  // (A/B builds)
constexpr bool kKcInlineDrain = false;
#else
constexpr bool kKcInlineDrain = true;
#endif

// The lists of a wavefront, in two phases that each keep all 64 lanes on one kind of work and wait for memory ONCE:
//   gate    every closed run: the 19 bases around its candidate hashed, ONE bit of the sample's gate (the table's gate minus the
//           hashes whose k-mers are all saturated); the runs that pass are compacted (ballot + popcount) into `hitq`;
//   lookup  64 runs at a time, one to a lane: the run's bucket — one 128-byte line, four entries, all of it — and the four
//           entries' own counters, all requested together.  An entry filed under the run's hash whose counter is below the
//           saturation value is matched where it stands (kc_match_windows: the entry says which candidate of its k-mer the hash
//           is of, so the k-mer can only be at two windows of the run) and counted with two adds that nothing waits for.  The
//           rare rest — a bucket's entries beyond its line (one bucket in a thousand has any), entries whose hash is of several
//           candidates (repeats) — go through one rolled loop afterwards (an entry read per turn).  A run all of whose entries
//           are saturated clears its hash's bit in the sample's gate: at a metagenome's coverage most runs of an abundant
//           genome stop at the gate from then on.
// So a tile costs two round trips to memory beyond its own bases.  One copy of this code per translation unit, CALLED where
// nothing of the walk is live (cfg: log2(buckets) | k << 8 | bad << 16 | ablate << 17 (three bits) | cs << 20).
__device__ __forceinline__ void kc_drain_body(MG_GLB uint32_t* live, const MG_GLB uint32_t* shared, const MG_GLB KcEntry* prim,
                                              const MG_GLB KcEntry* ovf, MG_GLB uint32_t* counts, MG_GLB uint32_t* csat, uint32_t gshift,
                                              uint32_t cfg, uint32_t lds, uint32_t sd, uint32_t cnt, uint32_t from, uint32_t limit) {
  const int lane = (int)(threadIdx.x & 63u);
  const KcLds L(sd);
  const MG_LDS uint32_t* fwd = (const MG_LDS uint32_t*)(size_t)(lds + L.fwd);
  const MG_LDS uint32_t* inv = (const MG_LDS uint32_t*)(size_t)(lds + L.inv);
  const MG_LDS uint32_t* p0s = (const MG_LDS uint32_t*)(size_t)(lds + L.p0s);
  const MG_LDS uint32_t* lists = (const MG_LDS uint32_t*)(size_t)(lds + L.lists);
  MG_LDS unsigned long long* hitq = (MG_LDS unsigned long long*)(size_t)(lds + L.hitq);  // hash | event << 32
  MG_LDS uint32_t* hitl = (MG_LDS uint32_t*)(size_t)(lds + L.hitl);                      // ... and whose it is
  MG_LDS uint32_t* stat = (MG_LDS uint32_t*)(size_t)(lds + L.stat);
  const uint32_t ablate = (cfg >> 17) & 7u;
  const uint32_t epoch = (gshift >> 8) & 0xffu;  // (the pass number of the entry counters rides above the gate's shift)
  gshift &= 63u;
  const KcIndexView ix{live, shared, prim, ovf, counts, csat, (1u << (cfg & 0xffu)) - 1u, gshift, cfg >> 20, ablate, epoch};
  const uint32_t nprim = kKcSlots * (ix.bmask + 1u);
  const int k = (int)((cfg >> 8) & 0xffu);
  const bool bad = (cfg >> 16) & 1u;
  if (ablate == 1u || ablate == 7u) return;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t hn = 0, nev = 0, npass = 0, found = 0;
#ifdef MG_KC_CLOCKS
  const uint64_t clk0 = __builtin_readcyclecounter();
  uint64_t clk_hits = 0;
#endif

  auto lookup_batch = [&](uint32_t n) {
#pragma unroll 1
    for (uint32_t at0 = 0; at0 < n; at0 += 64u) {
      const uint32_t at = at0 + (uint32_t)lane;
      const bool active = at < n;
      uint32_t key = kKcNone, ev = 0, p0 = 0, num0 = 0, cv[kKcSlots];
      kc_u32x4 ea[kKcSlots], eb[kKcSlots];
#pragma unroll
      for (uint32_t t = 0; t < kKcSlots; ++t) { ea[t] = kc_u32x4{0u, 0u, 0u, 0u}; eb[t] = kc_u32x4{0u, kKcNone, 0u, 0u}; cv[t] = 0; }
      if (active) {
        const unsigned long long hq = hitq[at];
        key = (uint32_t)hq;
        ev = (uint32_t)(hq >> 32);
        p0 = p0s[hitl[at]];
        num0 = kKcSlots * (key & ix.bmask);
        const MG_GLB kc_u32x4* p = reinterpret_cast<const MG_GLB kc_u32x4*>(ix.prim + num0);
#pragma unroll
        for (uint32_t t = 0; t < kKcSlots; ++t) { ea[t] = p[2u * t]; eb[t] = p[2u * t + 1u]; }  // (words | head, hash, off, pad)
        if (ix.cs != 0u) {
#pragma unroll
          for (uint32_t t = 0; t < kKcSlots; ++t) cv[t] = MG_KC_LOAD(&ix.csat[num0 + t]);
        }
      }
      const uint32_t i1 = ev & 1023u, i2 = (ev >> 10) & 1023u, pos = kc_event_pos(ev);
      bool any = false, open = false;  // an entry filed under the run's hash; one of them not saturated
      uint32_t slow = 0, fastm = 0;    // entries of the line to match: whose hash is of several candidates / of one: bit t
#pragma unroll
      for (uint32_t t = 0; t < kKcSlots; ++t) {
        const bool mine = active && eb[t].y == key, want = mine && !(ix.cs != 0u && kc_entry_count(cv[t], ix.epoch) >= ix.cs);
        any = any || mine;
        open = open || want;
        fastm |= (want && eb[t].z != kKcSeveral && ablate != 3u) ? 1u << t : 0u;
        slow |= (want && eb[t].z == kKcSeveral && ablate != 3u) ? 1u << t : 0u;
      }
#pragma unroll 1
      for (uint32_t t = 0; t < kKcSlots; ++t) {  // (rolled: one copy of the comparison; the entry's words are selected)
        const bool fast = (fastm >> t) & 1u;
        if (__ballot(fast) == 0ull) continue;
        const kc_u32x4 a = t == 0 ? ea[0] : (t == 1 ? ea[1] : (t == 2 ? ea[2] : ea[3]));
        const kc_u32x4 b = t == 0 ? eb[0] : (t == 1 ? eb[1] : (t == 2 ? eb[2] : eb[3]));
        const uint32_t seen = t == 0 ? cv[0] : (t == 1 ? cv[1] : (t == 2 ? cv[2] : cv[3]));
        if (fast) {
          const uint32_t f = kc_match_windows(fwd, inv, k, bad, a.x, a.y, a.z, a.w, b.z, p0, pos, i1, i2);
          kc_count_entry(ix, num0 + t, b.x, f, seen);
          found += f;
        }
      }
      // the rare rest, an entry read per turn: the line's entries of several candidates, then the bucket's entries beyond its line
      uint32_t nv = active ? eb[0].w : 0u, tv = 0;
      const uint32_t oa = eb[1].w;
      while (__ballot(slow != 0u || tv < nv) != 0ull) {
        if (slow != 0u || tv < nv) {
          const bool line = slow != 0u;
          uint32_t num;
          if (line) { num = num0 + (uint32_t)__builtin_ctz(slow); slow &= slow - 1u; }
          else { num = nprim + oa + tv; ++tv; }
          const KcEntry E = kc_entry(ix, num);
          const uint32_t seen = ix.cs != 0u ? MG_KC_LOAD(&ix.csat[num]) : 0u;
          bool want = line;
          if (!line && E.key == key) {
            any = true;
            want = !(ix.cs != 0u && kc_entry_count(seen, ix.epoch) >= ix.cs);
            open = open || want;
          }
          if (want && ablate != 3u) {
            const uint32_t f = E.off == kKcSeveral ? kc_scan_run(fwd, inv, k, bad, E.w, p0, i1, i2)
                                                   : kc_match_windows(fwd, inv, k, bad, E.w[0], E.w[1], E.w[2], E.w[3], E.off, p0, pos, i1, i2);
            kc_count_entry(ix, num, E.head, f, seen);
            found += f;
          }
        }
      }
      // every k-mer filed under this hash is saturated: its runs stop at the gate from now on — unless another hash of the table
      // has the same bit (a run that such a bit let through found no entry of its own: not its bit to clear)
      if (ix.cs != 0u && any && !open) {
        const uint32_t g = key >> ix.gshift;
        if (!((ix.shared[g >> 5] >> (g & 31u)) & 1u)) MG_KC_AND(&ix.live[g >> 5], ~(1u << (g & 31u)));
      }
    }
  };

  const uint32_t maxc = MG_UNIFORM(wave_max_u32(cnt));  // (in a scalar register: the slots beyond it are skipped by scalar branches)
  const uint32_t myp0 = p0s[lane];
  limit = MG_UNIFORM(limit);
  const bool clip = limit < 1024u;  // (the usual call: the walk went to the end of the tile, nothing to cut)
  // All of a tile's runs in one round trip to memory, then ONE look-up for the runs that passed (a single call: a copy of the
  // look-up per slot of the unrolled loop was 80 KB of code).  A 150 bp read closes seven runs, the tile's longest list has eleven
  // or twelve: the first kKcDirect slots of every lane are taken lane by lane, and what the lanes hold BEYOND them — some thirty
  // runs of a tile — is gathered into one more slot, a run to a lane (tokens through LDS: whose, which slot), when it fits 64;
  // slot by slot, most lanes idle, those tail slots were 7 % of the kernel.  hitq holds 128: when a round's runs past the gate do
  // not fit (a tile of reads from unsaturated k-mers), the slots that did not get in are taken again next round.
  constexpr int G = (int)kKcDirect;
  MG_LDS uint32_t* tokens = hitl;  // (the tokens are read before the first run is queued)
  uint32_t done = G, first = from;  // (first: the first window of the lane's next run — the walk leaves it out of the events)
  for (uint32_t s0 = 0; s0 < maxc; s0 += done) {
    uint32_t ev[G + 1], hk[G + 1], gw[G + 1], src = (uint32_t)lane, srcp0 = myp0;
    // ---- the tail beyond this round's direct slots: gathered when it fits one slot
    bool gathered = false;
    if (s0 + (uint32_t)G < maxc) {  // (uniform)
      const uint32_t ex = cnt > s0 + (uint32_t)G ? cnt - (s0 + (uint32_t)G) : 0u;
      uint32_t total = 0;
#pragma unroll 1
      for (uint32_t e = 0; s0 + (uint32_t)G + e < maxc; ++e) total += (uint32_t)__popcll(__ballot(ex > e));
      if (total <= 64u) {  // (uniform)
        uint32_t off = 0;
#pragma unroll 1
        for (uint32_t e = 0; s0 + (uint32_t)G + e < maxc; ++e) {
          const unsigned long long m = __ballot(ex > e);
          if (ex > e) tokens[off + (uint32_t)__popcll(m & below)] = (uint32_t)lane | ((s0 + (uint32_t)G + e) << 8);
          off += (uint32_t)__popcll(m);
        }
        wave_lds_sync();
        ev[G] = ~0u;
        if ((uint32_t)lane < total) {
          const uint32_t tok = tokens[lane], slot = tok >> 8;
          src = tok & 63u;
          srcp0 = p0s[src];
          uint32_t nxt = ((lists[(slot - 1u) * 64u + src] >> 10) & 1023u) + 1u;  // (the run before it in its lane's list ends one window earlier)
          ev[G] = kc_event_first(lists[slot * 64u + src], nxt);
        }
        wave_lds_sync();
        gathered = true;
      }
    }
#pragma unroll
    for (int j = 0; j < G + 1; ++j) {
      hk[j] = 0; gw[j] = 0;
      if (j < G) {
        ev[j] = ~0u;
        if (s0 + j >= maxc) continue;  // (uniform)
        // (the slot is there whether the lane has filled it or not — kKcListCap + 1 of them: read, then selected, no branch)
        uint32_t nxt = first;
        const uint32_t got = kc_event_first(lists[(s0 + j) * 64u + (uint32_t)lane], nxt);
        const bool mine = s0 + j < cnt;
        ev[j] = mine ? got : ~0u;
        first = mine ? nxt : first;
      } else if (!gathered) {
        ev[j] = ~0u;
        continue;  // (uniform)
      }
      if (clip) {  // what lies at or beyond `limit` is walked again (mg_kcount_core.h: kc_walk): not now
        const uint32_t i1 = ev[j] & 1023u, i2 = (ev[j] >> 10) & 1023u;
        if (i1 >= limit) ev[j] = ~0u;
        else if (i2 >= limit) ev[j] = (ev[j] & ~(1023u << 10)) | ((limit - 1u) << 10);
      }
      // a run's event says where its candidate starts: the bases around it, hashed, are what the table files k-mers under
      hk[j] = kc_run_hash(fwd, j < G ? myp0 : srcp0, kc_event_pos(ev[j]), k);
      if (!kc_event_none(ev[j])) gw[j] = MG_KC_LOAD(&live[(hk[j] >> gshift) >> 5]);
    }
    done = gathered ? maxc - s0 : (uint32_t)G;  // (the gathered slot stands for every slot beyond the direct ones)
    hn = 0;
#pragma unroll
    for (int j = 0; j < G + 1; ++j) {
      if (j < G ? (s0 + j >= maxc || (uint32_t)j >= done) : (!gathered || done < maxc - s0)) continue;  // (uniform)
      const bool pass = (gw[j] >> ((hk[j] >> gshift) & 31u)) & 1u;
      const unsigned long long m = __ballot(pass);
      if (hn + (uint32_t)__popcll(m) > kKcHitCap) {  // no room: this slot and what follows it come again (the gathered one: its slots, lane by lane or gathered anew)
        done = (uint32_t)j;
        if (j < G) first = ev[j] & 1023u;
        continue;
      }
      nev += kc_event_none(ev[j]) ? 0u : 1u;
      if (m == 0ull) continue;
      if (pass) {
        const uint32_t at = hn + (uint32_t)__popcll(m & below);
        hitq[at] = (unsigned long long)hk[j] | ((unsigned long long)ev[j] << 32);
        hitl[at] = j < G ? (uint32_t)lane : src;
      }
      hn += (uint32_t)__popcll(m);
      npass += pass ? 1u : 0u;
    }
    if (ablate != 2u && hn) {
#ifdef MG_KC_CLOCKS
      const uint64_t h0 = __builtin_readcyclecounter();
#endif
      wave_lds_sync();
      lookup_batch(hn);
      wave_lds_sync();
#ifdef MG_KC_CLOCKS
      clk_hits += __builtin_readcyclecounter() - h0;
#endif
    }
  }
  // (the wavefront's totals stay in LDS until the kernel ends: an atomic per call on three global words that every wavefront
  // shares was most of the kernel's time — 31 k calls per 2M reads, each queueing behind the others at the memory side)
  nev = wave_sum_u32(nev); npass = wave_sum_u32(npass); found = wave_sum_u32(found);
  if (lane == 0) { stat[0] += nev; stat[1] += npass; stat[2] += found; }
#ifdef MG_KC_CLOCKS
  if (lane == 0) { stat[4] += (uint32_t)((__builtin_readcyclecounter() - clk0) >> 6); stat[5] += (uint32_t)(clk_hits >> 6); stat[11] += 1; }
#endif
}

__device__ __attribute__((noinline)) void kc_drain(MG_GLB uint32_t* live, const MG_GLB uint32_t* shared, const MG_GLB KcEntry* prim,
                                                   const MG_GLB KcEntry* ovf, MG_GLB uint32_t* counts, MG_GLB uint32_t* csat, uint32_t gshift,
                                                   uint32_t cfg, uint32_t lds, uint32_t sd, uint32_t cnt, uint32_t from, uint32_t limit) {
  kc_drain_body(live, shared, prim, ovf, counts, csat, gshift, cfg, lds, sd, cnt, from, limit);
}

// what kc_walk writes through
struct KcDevOut {
  MG_LDS uint32_t* mine;  // this lane's column of the lists (slot s at mine[s * 64]): kc_event
  uint32_t* live;
  const uint32_t* shared;
  const KcEntry* prim;
  const KcEntry* ovf;
  uint32_t* counts;
  uint32_t* csat;
  uint32_t gshift, cfg, lds, sd;
  __device__ __forceinline__ void put(uint32_t slot, uint32_t word, uint32_t info) { mine[slot * 64u] = kc_event(word, info); }
  template <bool INL>
  __device__ __forceinline__ void drain(uint32_t cnt, uint32_t from, uint32_t limit) {
#ifdef MG_KC_NO_DRAIN  // (ISA inspection: what the walks need by themselves)
    return;
#endif
    wave_lds_sync();
    if constexpr (INL)
      kc_drain_body((MG_GLB uint32_t*)live, (const MG_GLB uint32_t*)shared, (const MG_GLB KcEntry*)prim, (const MG_GLB KcEntry*)ovf,
                    (MG_GLB uint32_t*)counts, (MG_GLB uint32_t*)csat, gshift, cfg, lds, sd, cnt, from, limit);
    else
    kc_drain((MG_GLB uint32_t*)live, (const MG_GLB uint32_t*)shared, (const MG_GLB KcEntry*)prim, (const MG_GLB KcEntry*)ovf,
             (MG_GLB uint32_t*)counts, (MG_GLB uint32_t*)csat, gshift, cfg, lds, sd, cnt, from, limit);
    wave_lds_sync();
  }
  static constexpr uint32_t kCap = kKcListCap;
  __device__ __forceinline__ bool any_full(uint32_t cnt) const { return __builtin_amdgcn_ballot_w64(cnt >= kKcListCap) != 0ull; }
  __device__ __forceinline__ uint32_t last_window(uint32_t slot) const { return (mine[slot * 64u] >> 10) & 1023u; }
  __device__ __forceinline__ uint32_t wave_min(uint32_t v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t t = __shfl_xor(v, o, 64);
      v = t < v ? t : v;
    }
    return v;
  }
};

// a staged tile through the walk, as often as the lists fill up
template <int K, bool INL>
__device__ __forceinline__ void kc_tile(const MG_LDS uint32_t* fwd, const MG_LDS uint32_t* inv, uint32_t p0, uint32_t len, uint32_t maxlen,
                                        int mode, KcDevOut& out) {
  if (maxlen < (uint32_t)K) return;
  const uint32_t nwmax = maxlen - (uint32_t)K + 1u;
  uint32_t w0 = 0;
  do {
    uint32_t cnt = 0;
    const uint32_t from = w0;
    if (mode == 0) w0 = kc_walk<K, 0>(fwd, inv, p0, len, maxlen, w0, out, cnt);
    else if (mode == 1) w0 = kc_walk<K, 1>(fwd, inv, p0, len, maxlen, w0, out, cnt);
    else w0 = kc_walk<K, 2>(fwd, inv, p0, len, maxlen, w0, out, cnt);
    out.template drain<INL>(cnt, from, w0 < nwmax ? w0 : 1024u);
  } while (w0 < nwmax);
}

template <int K>
__global__ __launch_bounds__(64 * kKcWaves) __attribute__((amdgpu_waves_per_eu(MG_KC_WAVES_PER_EU))) void k_count_kmers(const KcArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
  const KcLds L(a.sd);
  MG_LDS uint8_t* base = (MG_LDS uint8_t*)smem + (size_t)wave * L.total;
  const uint32_t lds = (uint32_t)(size_t)base;
  MG_LDS uint32_t* fwd = (MG_LDS uint32_t*)(base + L.fwd);
  MG_LDS uint32_t* inv = (MG_LDS uint32_t*)(base + L.inv);
  MG_LDS uint16_t* inv16 = (MG_LDS uint16_t*)(base + L.inv);
  MG_LDS uint32_t* p0s = (MG_LDS uint32_t*)(base + L.p0s);
  MG_LDS uint32_t* lists = (MG_LDS uint32_t*)(base + L.lists);
  MG_LDS uint32_t* stat = (MG_LDS uint32_t*)(base + L.stat);
  const uint32_t cfg0 = (uint32_t)__builtin_popcount(a.bmask) | ((uint32_t)K << 8) | ((a.ablate & 7u) << 17) | ((a.cs > 4095u ? 0u : a.cs) << 20);
  uint32_t kmers = 0;
  if (lane < 12) stat[lane] = 0;
  wave_lds_sync();
  // The wavefronts of a SIMD start together and every tile costs them the same: left alone they walk at the same time (the
  // vector units busy, memory idle) and empty their lists at the same time (the reverse).  Each waits, once, for its slot's share
  // of a tile's duration: from then on one of them probes while the others walk.
  if (a.stagger) {
    const uint32_t slot = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 4) % 3u;  // HW_ID.WAVE_ID: the wavefront's slot on its SIMD
    for (uint32_t i = 0; i < slot * a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
#ifdef MG_KC_CLOCKS
  const uint64_t kclk0 = __builtin_readcyclecounter();
#endif
  const uint64_t ntiles = (a.nreads + 63) / 64;
  // (a tile's offsets are requested while the tile before it is walked: two round trips to memory less at the head of every tile)
  const uint64_t tstep = (uint64_t)gridDim.x * kKcWaves;
  uint64_t nbeg = 0, nend = 0;
  {
    const uint64_t rd0 = ((uint64_t)blockIdx.x * kKcWaves + wave) * 64 + lane;
    if (rd0 < a.nreads) { nbeg = a.offsets[rd0]; nend = a.offsets[rd0 + 1]; }
  }
  for (uint64_t tile = (uint64_t)blockIdx.x * kKcWaves + wave; tile < ntiles; tile += tstep) {
    const uint64_t rd = tile * 64 + lane;
    const uint64_t beg = nbeg, end = nend;
    {
      const uint64_t rdn = (tile + tstep) * 64 + lane;
      nbeg = nend = 0;
      if (rdn < a.nreads) { nbeg = a.offsets[rdn]; nend = a.offsets[rdn + 1]; }
    }
    const uint64_t len64 = end - beg;
    const uint64_t maxlen64 = wave_max_u64(len64);
    const uint64_t t_beg = __shfl(beg, 0, 64);
    const uint64_t t_end = wave_max_u64(end);
    const uintptr_t a_first = reinterpret_cast<uintptr_t>(a.bases) + t_beg;
    const uintptr_t a0 = a_first & ~(uintptr_t)15;
    const uint64_t shift = a_first - a0;
    const uint64_t nbytes = shift + (t_end - t_beg);
    if (maxlen64 < (uint64_t)K) continue;  // no k-mer in the tile
    if (nbytes <= 16ull * a.sd && maxlen64 <= kKcMaxRead) {
      // ---- the usual tile: coalesced HBM -> LDS copy of its whole span, 16 bases per lane and step -> one dword of the stream
      const uint32_t nd = (uint32_t)((nbytes + 15) / 16);
      const uint4* g = reinterpret_cast<const uint4*>(a0);
      uint32_t notbase = 0;
      // (six loads in flight per lane, then their packing: as a plain loop every 16 bytes were a round trip of their own, eleven per
      // tile of 150 bp reads — most of what a wavefront waited for)
      for (uint32_t i0 = lane; i0 < nd; i0 += 64 * 6) {
        uint4 v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const uint32_t i = i0 + 64u * (uint32_t)j;
          v[j] = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
          if (i < nd) v[j] = a.ablate != 7u ? g[i] : reinterpret_cast<const uint4*>(a.bases)[i & 511u];  // (7, measurements only: the walk alone, over 8 KB of text that stay in the caches)
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const uint32_t i = i0 + 64u * (uint32_t)j;
          const uint32_t vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
          uint32_t nb;
          const uint32_t packed = kc_pack16(vv, nb);
          if (i < nd) fwd[i] = packed;
          notbase |= nb;
        }
      }
      const bool bad = __ballot(notbase != 0) != 0ull;
      if (bad) {  // (rare: N runs, the slop of the neighbouring tiles at a buffer's edge) the same bytes again, for the bit per base
        for (uint32_t i = lane; i < nd + 4; i += 64) {
          uint32_t bits = 0;
          if (i < nd) {
            const uint4 v = g[i];
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
            bits = kc_notbase16(vv);
          }
          inv16[i ^ 1u] = (uint16_t)bits;  // (dword i >> 1 holds group i in its HIGH half when i is even)
        }
      }
      const uint32_t p0 = rd < a.nreads ? (uint32_t)(shift + (beg - t_beg)) : 0u;  // (a lane without a read walks the tile's first bases, masked)
      p0s[lane] = p0;
      wave_lds_sync();
      KcDevOut out{lists + lane, a.live, a.shared, a.prim, a.ovf, a.counts, a.csat, a.gshift, cfg0 | (bad ? 1u << 16 : 0u), lds, a.sd};
      const uint32_t len = (uint32_t)len64, maxlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)maxlen64);
      kmers += bad ? kc_clean_windows(inv, p0, len, maxlen, K) : (len >= (uint32_t)K ? len - (uint32_t)K + 1u : 0u);
      if (a.ablate != 6u)  // (6, measurements only: the tiles staged and nothing else)
        kc_tile<K, kKcInlineDrain>(fwd, inv, p0, len, maxlen, bad ? 0 : (__ballot(len != maxlen) == 0ull ? 1 : 2), out);
    } else {
      // ---- a tile that does not fit (a long read, or a span above the stage): every lane takes its read through in chunks of
      // ch bases that overlap by K - 1, in a slot of its own — byte loads, the general walk; no window is seen twice
      const uint32_t per = (a.sd * 16u / 64u) & ~15u;
      const uint32_t ch = per < 1008u ? per : 1008u;  // (>= K + 15: the launcher sizes the stage so)
      const uint32_t stride = ch - (uint32_t)K + 1u;
      const uint32_t p0 = (uint32_t)lane * ch;
      p0s[lane] = p0;
      const uint64_t nwin = len64 >= (uint64_t)K ? len64 - (uint64_t)K + 1 : 0;
      const uint64_t nchunks = (wave_max_u64(nwin) + stride - 1) / stride;
      for (uint64_t c = 0; c < nchunks; ++c) {
        const uint64_t cs = c * stride;
        const uint32_t clen = cs < nwin ? (uint32_t)(len64 - cs < ch ? len64 - cs : ch) : 0u;
        const uint8_t* src = a.bases + beg + cs;
        for (uint32_t gi = 0; gi < ch / 16u; ++gi) {
          uint32_t vv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint32_t v = 0;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
              const uint32_t at = gi * 16u + (uint32_t)(q * 4 + bb);
              v |= (at < clen ? (uint32_t)src[at] : (uint32_t)'A') << (8 * bb);
            }
            vv[q] = v;
          }
          uint32_t nb;
          const uint32_t gidx = p0 / 16u + gi;
          fwd[gidx] = kc_pack16(vv, nb);
          inv16[gidx ^ 1u] = (uint16_t)kc_notbase16(vv);
        }
        wave_lds_sync();
        KcDevOut out{lists + lane, a.live, a.shared, a.prim, a.ovf, a.counts, a.csat, a.gshift, cfg0 | (1u << 16), lds, a.sd};
        const uint32_t cmax = wave_max_u32(clen);
        kmers += kc_clean_windows(inv, p0, clen, cmax, K);
        kc_tile<K, false>(fwd, inv, p0, clen, cmax, 0, out);
      }
    }
  }
  const uint64_t total = wave_sum_u64((uint64_t)kmers);
  wave_lds_sync();
  if (lane == 0 && total) atomicAdd(a.stats, (unsigned long long)total);
  if (lane >= 1 && lane < 4 && stat[lane - 1]) atomicAdd(a.stats + lane, (unsigned long long)stat[lane - 1]);
#ifdef MG_KC_CLOCKS
  if (lane == 0) stat[6] = (uint32_t)((__builtin_readcyclecounter() - kclk0) >> 6);
  wave_lds_sync();
  if (lane >= 4 && lane < 12) atomicAdd(a.stats + lane, (unsigned long long)stat[lane]);
#endif
}

template <int K>
int launch_k(const KcArgs& a, unsigned grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_count_kmers<K>), dim3(grid), dim3(64 * kKcWaves), lds, st, a);
  return MG_OK;
}
#ifndef MG_KC_ONLY_K  // (A/B builds and ISA inspection: one k instead of fifty)
#define MG_KC_ONLY_K 0
#endif
template <int K = (MG_KC_ONLY_K ? MG_KC_ONLY_K : kKcMinK)>
int dispatch_kc(int k, const KcArgs& a, unsigned grid, size_t lds, hipStream_t st) {
  if constexpr (K > (MG_KC_ONLY_K ? MG_KC_ONLY_K : kKcMaxK)) {
    return fail(MG_ERR_ARG, "k = %d is outside [%d, %d]: no k-mer index for it", k, kKcMinK, kKcMaxK);
  } else {
    if (k == K) return launch_k<K>(a, grid, lds, st);
    return dispatch_kc<K + 1>(k, a, grid, lds, st);
  }
}

}  // namespace

}  // namespace mg

struct mg_kcounts {
  mg::DevBuf counts;  // u32[npairs + 1]
  mg::DevBuf live;    // the sample's gate: the table's, minus the hashes all of whose k-mers are saturated (mg_kcount_core.h)
  mg::DevBuf sat;     // a counter per entry number: what has been found under it (at the saturation value the entry is skipped)
  mg::DevBuf stats;   // u64[4]
  uint64_t n = 0, live_words = 0, sat_words = 0;
  uint32_t epoch = 0;  // the pass number the entry counters' words are tagged with (mg_kcount_core.h: kc_entry_count); 0: never reset yet
  const void* gate = nullptr;  // the table's gate bitmap the live one is reset from (owned by the table: it outlives the counters)
  // Who wrote last, and when: with mg_stage_a_side_stream on, a reset and the counting go to the stage-A stream (so that a
  // rank's collectives on the main stream do not hold the next pass's counting back); whoever touches the counters on another
  // stream waits for this event first.
  hipEvent_t ev = nullptr;
  hipStream_t ev_stream = nullptr;
  ~mg_kcounts() {
    if (ev) {
      if (ev_stream) (void)hipEventSynchronize(ev);  // (its buffers go back to the pool: nothing may still be writing them)
      (void)hipEventDestroy(ev);
    }
  }
};

using namespace mg;

namespace mg {
// `st` is about to read or write the counters: after whatever was queued on them last
int kcounts_order(const mg_kcounts* kc, hipStream_t st) {
  if (kc && kc->ev && kc->ev_stream && kc->ev_stream != st) MG_HIP(hipStreamWaitEvent(st, kc->ev, 0));
  return MG_OK;
}
int kcounts_wait(const mg_kcounts* kc) { return kcounts_order(kc, ctx().stream); }
static int kcounts_wrote(mg_kcounts* kc, hipStream_t st) {
  MG_HIP(hipEventRecord(kc->ev, st));
  kc->ev_stream = st;
  return MG_OK;
}
static hipStream_t kcounts_stage_stream() {
  Context& c = ctx();
  return c.a_side ? (c.a_side == 2 ? c.stream_a2 : c.stream_a) : c.stream;
}

// a sample's gate := the table's (a kernel: hipMemcpyAsync device-to-device goes through a DMA engine at ~100 GB/s — 1.2 ms for the
// 128 MB of a ten-million-k-mer table, where these loads and stores take 0.08)
__global__ void k_kc_copy_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t nwords) {
  const uint64_t n4 = nwords / 4;
  const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src);
  uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst);
  KC_FOR(i, n4) d4[i] = s4[i];
  KC_FOR(i, nwords - 4 * n4) dst[4 * n4 + i] = src[4 * n4 + i];
}

// a sample's counters for the other ranks: min(counter, 3) in two bits, pair i in bits 2 (i & 15) of dword i >> 4
__global__ void k_kc_pack2(const uint32_t* __restrict__ counts, uint64_t n, uint32_t* __restrict__ out) {
  KC_FOR(w, (n + 15) / 16) {
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t i = w * 16 + (uint64_t)j;
      const uint32_t c = i < n ? counts[i] : 0u;
      v |= (c > 3u ? 3u : c) << (2 * j);
    }
    out[w] = v;
  }
}
// ... and the sum of nranks such arrays (rank r's at all + r * stride dwords) back into counters
__global__ void k_kc_merge2(const uint32_t* __restrict__ all, uint32_t nranks, uint64_t stride, uint64_t n, uint32_t* __restrict__ counts) {
  KC_FOR(w, (n + 15) / 16) {
    uint32_t s[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = 0;
    for (uint32_t r = 0; r < nranks; ++r) {
      const uint32_t v = all[(uint64_t)r * stride + w];
#pragma unroll
      for (int j = 0; j < 16; ++j) s[j] += (v >> (2 * j)) & 3u;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t i = w * 16 + (uint64_t)j;
      if (i < n) counts[i] = s[j];
    }
  }
}
}  // namespace mg

extern "C" {

int mg_refdb_index_kmers(mg_refdb* db, const uint64_t* kmer_hi, const uint64_t* kmer_lo) {
  MG_REQUIRE_READY();
  if (!db) return fail(MG_ERR_ARG, "null argument");
  MG_TRY(refdb_ready(db));
  const int k = db->ks[db->nk - 1];
  if (k < kKcMinK || k > kKcMaxK)
    return fail(MG_ERR_ARG, "k = %d is outside [%d, %d]: no k-mer index for it (the hash path serves it)", k, kKcMinK, kKcMaxK);
  const uint64_t n = db->kmax.total;
  hipStream_t st = ctx().stream;
  if (n && (kmer_hi || kmer_lo)) {
    if (!kmer_hi || !kmer_lo) return fail(MG_ERR_ARG, "both words of the k-mers, or neither");
    MG_TRY(db->kmer_hi.alloc((n + 1) * 8));
    MG_TRY(db->kmer_lo.alloc((n + 1) * 8));
    MG_HIP(hipStreamSynchronize(st));
    MG_TRY(upload_ranges({{kmer_hi, {db->kmer_hi.p, n * 8}}, {kmer_lo, {db->kmer_lo.p, n * 8}}}, st));
  }
  if (n && !db->kmer_hi.p) return fail(MG_ERR_STATE, "the table does not hold its k-mers: pass them (format 3: k<K>.kmer_hi.u64 / .kmer_lo.u64)");
  std::unique_ptr<KmerIndex> ix(new KmerIndex());
  ix->k = k;

  MG_TRY(ix->head.alloc((n + 1) * 4));
  DevBuf chi, clo, iota, ord1, ord2, s_lo, s_hi, g_hi, flag, before, dhi, dlo, dhead, nkeys, kbefore, ekey, eid, skey, perm;
  uint64_t nd = 0, ne = 0;  // distinct k-mers; entries (a k-mer filed under several hashes has one for each)
  if (n) {
    MG_TRY(chi.alloc(n * 8)); MG_TRY(clo.alloc(n * 8)); MG_TRY(iota.alloc(n * 4)); MG_TRY(ord1.alloc(n * 4)); MG_TRY(ord2.alloc(n * 4));
    MG_TRY(s_lo.alloc(n * 8)); MG_TRY(s_hi.alloc(n * 8)); MG_TRY(g_hi.alloc(n * 8)); MG_TRY(flag.alloc(n * 4)); MG_TRY(before.alloc((n + 2) * 8));
    hipLaunchKernelGGL(k_kc_canon, dim3(g256(n)), dim3(256), 0, st, db->kmer_hi.as<uint64_t>(), db->kmer_lo.as<uint64_t>(), n, k,
                       chi.as<uint64_t>(), clo.as<uint64_t>(), iota.as<uint32_t>());
    // ascending by (chi, clo): low word first, then a stable pass over the high word
    MG_TRY(sort_pairs(clo.as<uint64_t>(), s_lo.as<uint64_t>(), iota.as<uint32_t>(), ord1.as<uint32_t>(), n));
    hipLaunchKernelGGL(k_kc_gather, dim3(g256(n)), dim3(256), 0, st, chi.as<uint64_t>(), ord1.as<uint32_t>(), n, g_hi.as<uint64_t>());
    MG_TRY(sort_pairs(g_hi.as<uint64_t>(), s_hi.as<uint64_t>(), ord1.as<uint32_t>(), ord2.as<uint32_t>(), n));
    hipLaunchKernelGGL(k_kc_gather, dim3(g256(n)), dim3(256), 0, st, clo.as<uint64_t>(), ord2.as<uint32_t>(), n, s_lo.as<uint64_t>());
    hipLaunchKernelGGL(k_kc_heads, dim3(g256(n)), dim3(256), 0, st, s_hi.as<uint64_t>(), s_lo.as<uint64_t>(), n, flag.as<uint32_t>());
    MG_HIP(hipGetLastError());
    MG_TRY(exclusive_sum_u32_to_u64(flag.as<uint32_t>(), before.as<uint64_t>(), n, &nd));
    MG_TRY(dhi.alloc(nd * 8)); MG_TRY(dlo.alloc(nd * 8)); MG_TRY(dhead.alloc(nd * 4)); MG_TRY(nkeys.alloc((nd + 1) * 4));
    MG_TRY(kbefore.alloc((nd + 2) * 8));
    hipLaunchKernelGGL(k_kc_distinct, dim3(g256(n)), dim3(256), 0, st, s_hi.as<uint64_t>(), s_lo.as<uint64_t>(), ord2.as<uint32_t>(),
                       flag.as<uint32_t>(), before.as<uint64_t>(), n, k, dhi.as<uint64_t>(), dlo.as<uint64_t>(), dhead.as<uint32_t>(),
                       nkeys.as<uint32_t>());
    hipLaunchKernelGGL(k_kc_pair_heads, dim3(g256(n)), dim3(256), 0, st, ord2.as<uint32_t>(), flag.as<uint32_t>(), before.as<uint64_t>(),
                       dhead.as<uint32_t>(), n, ix->head.as<uint32_t>());
    MG_HIP(hipGetLastError());
  }
  ix->ndistinct = nd;
  unsigned bb = 8;  // log2(buckets): about one k-mer per bucket
  while (bb < 28 && (1ull << bb) < nd) ++bb;
  ix->bmask = (1u << bb) - 1u;
  ix->nbuckets = 1ull << bb;
  const int64_t gx = dbg("kc_gate_extra");
  ix->gbits = kc_gate_bits(nd, gx > 0 ? (uint32_t)gx : kKcGateExtra);
  ix->gate_words = (1ull << ix->gbits) / 32 + 2;
  MG_TRY(ix->gate.alloc(ix->gate_words * 4));
  MG_TRY(ix->shared.alloc(ix->gate_words * 4));
  MG_HIP(hipMemsetAsync(ix->gate.p, 0, ix->gate_words * 4, st));
  MG_HIP(hipMemsetAsync(ix->shared.p, 0, ix->gate_words * 4, st));
  if (nd) {
    MG_TRY(exclusive_sum_u32_to_u64(nkeys.as<uint32_t>(), kbefore.as<uint64_t>(), nd, &ne));
    MG_TRY(ekey.alloc((ne + 1) * 8)); MG_TRY(eid.alloc((ne + 1) * 4)); MG_TRY(skey.alloc((ne + 1) * 8)); MG_TRY(perm.alloc((ne + 1) * 4));
    hipLaunchKernelGGL(k_kc_emit_keys, dim3(g256(nd)), dim3(256), 0, st, dhi.as<uint64_t>(), dlo.as<uint64_t>(), kbefore.as<uint64_t>(), nd, k,
                       ix->bmask, ekey.as<uint64_t>(), eid.as<uint32_t>());
    MG_HIP(hipGetLastError());
    MG_TRY(sort_pairs(ekey.as<uint64_t>(), skey.as<uint64_t>(), eid.as<uint32_t>(), perm.as<uint32_t>(), ne));
  }
  ix->nentries = ne;
  // the entries in (bucket, hash) order, the buckets' bounds, then their places: four per bucket in prim, the rest in ovf
  DevBuf offs, ent, extra, obefore;
  MG_TRY(offs.alloc((ix->nbuckets + 2) * 4));
  MG_TRY(ent.alloc((ne + 1) * sizeof(KcEntry)));
  MG_TRY(extra.alloc((ix->nbuckets + 1) * 4));
  MG_TRY(obefore.alloc((ix->nbuckets + 2) * 8));
  MG_TRY(ix->prim.alloc(kKcSlots * ix->nbuckets * sizeof(KcEntry)));
  if (ne) {
    hipLaunchKernelGGL(k_kc_entries, dim3(g256(ne)), dim3(256), 0, st, skey.as<uint64_t>(), perm.as<uint32_t>(), dhi.as<uint64_t>(),
                       dlo.as<uint64_t>(), dhead.as<uint32_t>(), ne, k, 32u - ix->gbits, ent.as<KcEntry>(), ix->gate.as<uint32_t>(),
                       ix->shared.as<uint32_t>());
    hipLaunchKernelGGL(k_kc_offsets, dim3(g256(ne + 1)), dim3(256), 0, st, skey.as<uint64_t>(), ne, ix->nbuckets, offs.as<uint32_t>());
  } else {
    MG_HIP(hipMemsetAsync(offs.p, 0, (ix->nbuckets + 2) * 4, st));
  }
  hipLaunchKernelGGL(k_kc_extra, dim3(g256(ix->nbuckets)), dim3(256), 0, st, offs.as<uint32_t>(), ix->nbuckets, extra.as<uint32_t>());
  MG_HIP(hipGetLastError());
  uint64_t novf = 0;
  MG_TRY(exclusive_sum_u32_to_u64(extra.as<uint32_t>(), obefore.as<uint64_t>(), ix->nbuckets, &novf));
  ix->novf = novf;
  MG_TRY(ix->ovf.alloc((novf + 1) * sizeof(KcEntry)));
  hipLaunchKernelGGL(k_kc_place, dim3(g256(ix->nbuckets)), dim3(256), 0, st, ent.as<KcEntry>(), offs.as<uint32_t>(), obefore.as<uint64_t>(),
                     ix->nbuckets, ix->prim.as<KcEntry>(), ix->ovf.as<KcEntry>());
  MG_HIP(hipGetLastError());
  MG_HIP(hipStreamSynchronize(st));
  db->kidx = std::move(ix);
  return MG_OK;
}

int mg_refdb_has_kmer_index(const mg_refdb* db) { return db && db->kidx ? 1 : 0; }
uint64_t mg_refdb_distinct_kmers(const mg_refdb* db) { return db && db->kidx ? db->kidx->ndistinct : 0; }

int mg_refdb_kmer_heads(const mg_refdb* db, uint32_t* head) {
  MG_REQUIRE_READY();
  if (!db || !db->kidx || !head) return fail(MG_ERR_ARG, "null argument, or a table without a k-mer index");
  return db->kmax.total ? mg_memcpy_d2h(head, db->kidx->head.p, db->kmax.total * 4) : MG_OK;
}

int mg_kcounts_new(const mg_refdb* db, mg_kcounts** out) {
  MG_REQUIRE_READY();
  if (!db || !out) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  if (!db->kidx) return fail(MG_ERR_STATE, "the table has no k-mer index (mg_refdb_index_kmers)");
  std::unique_ptr<mg_kcounts> kc(new mg_kcounts());
  kc->n = db->kmax.total;
  kc->live_words = db->kidx->gate_words;
  kc->sat_words = kKcSlots * db->kidx->nbuckets + db->kidx->novf + 4;  // (a counter per entry number)
  kc->gate = db->kidx->gate.p;
  MG_TRY(kc->counts.alloc((kc->n + 1) * 4));
  MG_TRY(kc->live.alloc(kc->live_words * 4));
  MG_TRY(kc->sat.alloc(kc->sat_words * 4));
  MG_TRY(kc->stats.alloc(12 * 8));
  MG_HIP(hipEventCreateWithFlags(&kc->ev, hipEventDisableTiming));
  MG_TRY(mg_kcounts_reset(kc.get()));
  *out = kc.release();
  return MG_OK;
}

int mg_kcounts_reset(mg_kcounts* kc) {
  MG_REQUIRE_READY();
  if (!kc) return fail(MG_ERR_ARG, "null argument");
  // (on a stream of its own: a caller that zeroes a set of counters when it is done with it — not in front of the next counting
  // kernel — gets the 0.2 ms of copies and fills beside that kernel instead of before it, and behind nothing: on the main stream
  // they queued behind the next pass's stage B, which waits for ITS counting kernel, and two passes' counting no longer overlapped)
  Context& c = ctx();
  if (!c.stream_r) MG_HIP(hipStreamCreateWithFlags(&c.stream_r, hipStreamNonBlocking));
  hipStream_t st = c.stream_r;
  MG_TRY(kcounts_order(kc, st));
  hipLaunchKernelGGL(k_kc_copy_words, dim3(g256(kc->live_words / 4 + 1)), dim3(256), 0, st, static_cast<const uint32_t*>(kc->gate),
                     kc->live.as<uint32_t>(), kc->live_words);
  MG_HIP(hipGetLastError());
  // (the entry counters are not zeroed: their words carry the pass number — only when that wraps, every 255 passes)
  kc->epoch = kc->epoch % 255u + 1u;
  if (kc->epoch == 1u) MG_HIP(hipMemsetAsync(kc->sat.p, 0, kc->sat_words * 4, st));
  MG_HIP(hipMemsetAsync(kc->counts.p, 0, (kc->n + 1) * 4, st));
  MG_HIP(hipMemsetAsync(kc->stats.p, 0, 12 * 8, st));
  return kcounts_wrote(kc, st);
}

int mg_kcounts_wait(const mg_kcounts* kc) {
  MG_REQUIRE_READY();
  if (!kc) return fail(MG_ERR_ARG, "null argument");
  return kcounts_wait(kc);
}

uint64_t mg_kcounts_pack2_bytes(const mg_kcounts* kc) { return kc ? ((kc->n + 15) / 16) * 4 : 0; }

int mg_kcounts_pack2_dev(const mg_kcounts* kc, uint32_t* d_out) {
  MG_REQUIRE_READY();
  if (!kc || !d_out) return fail(MG_ERR_ARG, "null argument");
  if (ctx().count_sat == 0 || ctx().count_sat > 3)
    return fail(MG_ERR_STATE, "two bits hold a counter that saturates at 3 or below (it saturates at %u): exchange the counters themselves", ctx().count_sat);
  MG_TRY(kcounts_wait(kc));
  if (kc->n) hipLaunchKernelGGL(k_kc_pack2, dim3(g256((kc->n + 15) / 16)), dim3(256), 0, ctx().stream, kc->counts.as<uint32_t>(), kc->n, d_out);
  MG_HIP(hipGetLastError());
  return MG_OK;
}

int mg_kcounts_merge2_dev(mg_kcounts* kc, const uint32_t* d_all, uint32_t nranks, uint64_t stride_dwords) {
  MG_REQUIRE_READY();
  if (!kc || !d_all || !nranks) return fail(MG_ERR_ARG, "null argument");
  if (stride_dwords < (kc->n + 15) / 16) return fail(MG_ERR_ARG, "a rank's array is shorter than the counters");
  MG_TRY(kcounts_wait(kc));
  if (kc->n) hipLaunchKernelGGL(k_kc_merge2, dim3(g256((kc->n + 15) / 16)), dim3(256), 0, ctx().stream, d_all, nranks, stride_dwords, kc->n,
                                kc->counts.as<uint32_t>());
  MG_HIP(hipGetLastError());
  return kcounts_wrote(kc, ctx().stream);
}

int mg_count_kmers_dev(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, uint64_t nbases, const mg_refdb* db,
                       mg_kcounts* kc) {
  MG_REQUIRE_READY();
  if (!db || !kc || (nreads && (!d_bases || !d_offsets))) return fail(MG_ERR_ARG, "null argument");
  if (!db->kidx) return fail(MG_ERR_STATE, "the table has no k-mer index (mg_refdb_index_kmers)");
  if (kc->n != db->kmax.total || kc->gate != db->kidx->gate.p)
    return fail(MG_ERR_ARG, "these counters belong to another table");
  if (nreads == 0) return MG_OK;
  Context& c = ctx();
  hipStream_t st = kcounts_stage_stream();
  MG_TRY(kcounts_order(kc, st));
  const KmerIndex& ix = *db->kidx;
  // the stage of a wavefront: 64 reads of average length + 12.5 %, in dwords of sixteen bases, a multiple of 64; never below what
  // a chunk of a long read needs (k + 15 bases per lane)
  const uint64_t avg = nbases ? (nbases + nreads - 1) / nreads : 150;
  uint64_t sd = ((64 * avg * 9 / 8 + 64 + 15) / 16 + 63) / 64 * 64;
  const uint64_t floor_sd = 64ull * (((uint64_t)ix.k + 15 + 15) / 16 + 1);
  if (sd < floor_sd) sd = floor_sd;
  if (sd > 2048) sd = 2048;  // 32 k bases
  const KcLds L((uint32_t)sd);
  const size_t lds = (size_t)kKcWaves * L.total;
  unsigned per_cu = (unsigned)(160 * 1024 / lds);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  // (what the registers allow: MG_KC_WAVES_PER_EU wavefronts on each of a CU's four SIMDs — a workgroup more than fit would start
  // when the first has finished, and the persistent grid's share of the tiles is cut by the workgroups LAUNCHED)
  if (per_cu > (unsigned)(MG_KC_WAVES_PER_EU * 4 / kKcWaves)) per_cu = (unsigned)(MG_KC_WAVES_PER_EU * 4 / kKcWaves);
  if (dbg("kc_wg_per_cu") > 0) per_cu = (unsigned)dbg("kc_wg_per_cu");
  if (c.a_side && c.a_side_wg_per_cu && per_cu > c.a_side_wg_per_cu) per_cu = c.a_side_wg_per_cu;
  const uint64_t ntiles = (nreads + 63) / 64;
  const unsigned grid = grid_for(ntiles, kKcWaves, (unsigned)c.num_cus * per_cu);
  KcArgs a{d_bases, d_offsets, nreads, kc->live.as<uint32_t>(), ix.shared.as<uint32_t>(), ix.prim.as<KcEntry>(), ix.ovf.as<KcEntry>(), kc->counts.as<uint32_t>(),
           kc->sat.as<uint32_t>(), kc->stats.as<unsigned long long>(), (32u - ix.gbits) | (kc->epoch << 8), ix.bmask, (uint32_t)sd, c.count_sat, (uint32_t)dbg("kc_ablate"), (uint32_t)dbg("kc_stagger")};
  {
    ProfScope ps("count_kmers", st);
    MG_TRY(dispatch_kc(ix.k, a, grid, lds, st));
    MG_HIP(hipGetLastError());
  }
  return kcounts_wrote(kc, st);
}

int mg_kcounts_stats(const mg_kcounts* kc, uint64_t* out4) {
  MG_REQUIRE_READY();
  if (!kc || !out4) return fail(MG_ERR_ARG, "null argument");
  MG_TRY(kcounts_wait(kc));
#ifdef MG_KC_CLOCKS  // (a build for tools/kcount_clocks.sh: four more words — wave-cycles / 64 in the drain, its matching part, the kernel)
  return mg_memcpy_d2h(out4, kc->stats.p, 12 * 8);
#else
  return mg_memcpy_d2h(out4, kc->stats.p, 4 * 8);
#endif
}

int mg_kcounts_download(const mg_kcounts* kc, const mg_refdb* db, uint32_t* per_pair) {
  MG_REQUIRE_READY();
  if (!kc || !db || !db->kidx || !per_pair) return fail(MG_ERR_ARG, "null argument");
  if (kc->n != db->kmax.total) return fail(MG_ERR_ARG, "these counters belong to another table");
  if (!kc->n) return MG_OK;
  MG_TRY(kcounts_wait(kc));
  DevBuf out;
  MG_TRY(out.alloc(kc->n * 4));
  hipLaunchKernelGGL(k_kc_per_pair, dim3(g256(kc->n)), dim3(256), 0, ctx().stream, kc->counts.as<uint32_t>(), db->kidx->head.as<uint32_t>(),
                     kc->n, ctx().count_sat, out.as<uint32_t>());
  MG_HIP(hipGetLastError());
  return mg_memcpy_d2h(per_pair, out.p, kc->n * 4);
}

int mg_kcounts_device(const mg_kcounts* kc, uint32_t** d_counts, uint64_t* n) {
  if (!kc || !d_counts || !n) return fail(MG_ERR_ARG, "null argument");
  *d_counts = kc->counts.as<uint32_t>();
  *n = kc->n;
  return MG_OK;
}

void mg_kcounts_free(mg_kcounts* kc) { delete kc; }

}  // extern "C"
