// mg_sketch_multi.hip — Stage A for EVERY k of a multi-k query in one launch.
//
// The reference's containment query is multi-k (`30-60-10`, scripts/select_db.py:75; BASELINE configs use
// {21,31,51}).  One launch per k streams, stages, encodes and rolls the reads |K| times; here a tile is staged and
// encoded once, each lane rolls ONE window at the largest k (mg_kmer.h: Roller<KMAX>) and every smaller k's
// canonical k-mer is read out of it (hash_suffix: its forward ASCII is a byte-granular funnel shift of the big
// window's, its reverse complement is the first K bytes of the big reverse window, the 2-bit forms are bit fields
// of the big ones) — what remains per k is its MurmurHash3 and its threshold test.  Candidates of all k share one
// LDS buffer per wavefront (entries tagged with the k's index) and go to per-k counting tables
// (mg_sketch_dev.h: table_add) exactly as in the single-k kernel; the host side turns every table into that k's
// sketch with the same tail (mg_sketch.hip: table_pack), so the sketches are bit-identical to single-k launches.
//
// Replaces: kmc -k60 -ci2 -cs3 (scripts/select_db.py:50-52) + k-mer hashing in CMash's streaming multi-k query
// (scripts/select_db.py:73-76).
#include <cstdlib>

#include "mg_sketch_dev.h"

namespace mg {

constexpr int kMaxMultiK = 4;

struct MultiArgs {
  uint64_t hmax[kMaxMultiK];
  Slot* tab[kMaxMultiK];                 // counting table of k number i: [nbuckets][kBucketSlots] slots
  unsigned long long* counters[kMaxMultiK];  // [0] candidates produced, [1] k-mers hashed, [2] table overflows
  const uint32_t* fbits[kMaxMultiK];     // optional membership pre-filter of k number i
  uint64_t fmask[kMaxMultiK];
  unsigned shift[kMaxMultiK];
  uint32_t cs;     // saturation value (low 30 bits of the cs word)
  uint32_t order;  // its top two bits
  uint32_t* list[kMaxMultiK];            // resident indexes: the list of slots the pass has touched in index i (or null)
  uint64_t listcap[kMaxMultiK];
  uint32_t epoch;  // != 0: the tables are resident indexes (mg_sketch_dev.h: resident_count) and this the pass's epoch
  uint32_t ablate; // resident kernel only, knob resident_ablate (tools/k1_probe.py): 1 = a flush drops its candidates (what the kernel costs without any look-up)
};

// Wave-level candidate sink shared by all k: (hash, k index) pairs staged in LDS, flushed into the per-k tables.
template <bool RESIDENT>  // the tables are resident indexes: a kernel of its own, so that the usual one carries none of it
struct MultiSink {
  uint64_t* lds_h;   // this wave's kCandBuf hashes
  uint8_t* lds_k;    // ... and the index of the k each belongs to
  const MultiArgs* A;  // in LDS
  int n;             // entries staged (wave-uniform)
  unsigned long long produced[kMaxMultiK];  // candidates inserted per k by this WAVEFRONT (wave-uniform: scalar registers)

  // A flush costs its candidates RANDOM accesses (one 128-byte line each): a word of the membership filter and the
  // home slot of the counting table.  Which of the two is looked at first does not change the result — whatever is in
  // the table has passed the filter — but it decides how many lines a candidate costs: filter first, 1 + (share that
  // passes); slot first, 1 + (share whose home slot does not hold it yet).  A sample that covers its genomes many
  // times over (or a dense table: BASELINE configs[3], 20 % of all k-mers are candidates) finds nearly every candidate
  // in its home slot; a sample of mostly unknown organisms has nearly every candidate rejected by the filter.  The
  // wavefront keeps to the order that was cheaper for its previous flush.
  bool slot_first;  // wave-uniform
  uint32_t order;   // 0: adapt; 1 / 2: pinned (tests; see kCsMask)
  uint64_t* lst = nullptr;  // RESIDENT, in LDS (registers are the hashing loop's): this wavefront's chunk of A->list[i], i < 4, then how full each is

  __device__ __forceinline__ void flush(int lane) {
    if (n == 0) return;
    wave_lds_sync();
    constexpr int J = kCandBuf / 64;
    uint64_t hh[J];
    uint32_t kk[J];
    Slot* home[J];
    uint4 sv[J];
    bool go[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int i = lane + 64 * j;
      hh[j] = i < n ? lds_h[i] : kReservedHash;
      kk[j] = i < n ? lds_k[i] : 0u;
      if constexpr (!RESIDENT)
        home[j] = A->tab[kk[j]] + (hh[j] >> A->shift[kk[j]]) * kBucketSlots + ((uint32_t)hh[j] & (kBucketSlots - 1));
      sv[j] = make_uint4(0, 0, 0, 0);
    }
    if constexpr (RESIDENT) {
      // resident indexes (mg_sketch_dev.h): one 16-byte load per candidate, all in flight together — one round trip; a
      // candidate that has to look one slot on goes back into the buffer (its k byte counts the slots above bit 1)
      if (A->ablate == 1u) { wave_lds_sync(); n = 0; return; }
      Slot* bucket[J];
      uint32_t hop[J];
      bool hit[J], fresh[J], on[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const uint32_t ki = kk[j] & 3u;
        hop[j] = kk[j] >> 2;
        bucket[j] = A->tab[ki] + (hh[j] != kReservedHash ? hh[j] >> A->shift[ki] : 0ull) * kBucketSlots;
      }
      uint32_t pos[J];
      resident_lookup<J>(bucket, hh, hop, A->epoch, A->cs, hit, fresh, on, pos, A->ablate < 2u);  // (2, 3: nothing is counted)
      if (A->ablate == 2u) { wave_lds_sync(); n = 0; return; }                               // (2: ... and nothing goes round again)
      wave_lds_sync();  // (every lane holds its entries in registers: the buffer's front is free for what goes round again)
      int back = 0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int q = 0; q < kMaxMultiK; ++q) produced[q] += (unsigned)__popcll(__ballot(hit[j] && (kk[j] & 3u) == (uint32_t)q));
        if (A->ablate == 0u && __ballot(fresh[j]) != 0ull) {
#pragma unroll
          for (int q = 0; q < kMaxMultiK; ++q) {
            if (!A->list[q]) continue;
            uint32_t* fills = reinterpret_cast<uint32_t*>(lst + kMaxMultiK);
            uint32_t* lbase = reinterpret_cast<uint32_t*>(lst[q]);
            uint32_t lfill = fills[q];
            resident_list_append(fresh[j] && (kk[j] & 3u) == (uint32_t)q, (uint32_t)(hh[j] >> A->shift[q]) * kBucketSlots + pos[j],
                                 A->list[q], A->listcap[q], A->counters[q], lbase, lfill, lane);
            lst[q] = reinterpret_cast<uint64_t>(lbase);  // (every lane the same value)
            fills[q] = lfill;
          }
        }
        const bool again = on[j] && hop[j] < kMaxHops;
        const unsigned long long m = __ballot(again);
        if (m == 0) continue;
        if (again) {
          const int at = back + __popcll(m & ((1ull << lane) - 1ull));
          lds_h[at] = hh[j];
          lds_k[at] = (uint8_t)(kk[j] + 4u);
        }
        back += __popcll(m);
      }
      wave_lds_sync();
      n = back;
      return;
    }
    if (slot_first) {
      // every candidate's home slot (key and counter in one 16-byte access), all in flight together; then the filter
      // words of those it does not hold
#pragma unroll
      for (int j = 0; j < J; ++j)
        if (hh[j] != kReservedHash) sv[j] = *reinterpret_cast<const uint4*>(home[j]);
      uint32_t fw[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const unsigned long long key = (unsigned long long)sv[j].x | ((unsigned long long)sv[j].y << 32);
        fw[j] = 0xffffffffu;
        if (hh[j] != kReservedHash && key != hh[j] + 1) {
          const uint32_t* fb = A->fbits[kk[j]];
          if (fb) fw[j] = fb[(hh[j] & A->fmask[kk[j]]) >> 5];
        }
      }
#pragma unroll
      for (int j = 0; j < J; ++j) go[j] = hh[j] != kReservedHash && ((fw[j] >> (hh[j] & 31u)) & 1u);
    } else {
      // every candidate's filter word, all in flight together; then the home slots of the survivors
      uint32_t fw[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        fw[j] = 0xffffffffu;
        if (hh[j] != kReservedHash) {
          const uint32_t* fb = A->fbits[kk[j]];
          if (fb) fw[j] = fb[(hh[j] & A->fmask[kk[j]]) >> 5];
        }
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        go[j] = hh[j] != kReservedHash && ((fw[j] >> (hh[j] & 31u)) & 1u);
        if (go[j]) sv[j] = *reinterpret_cast<const uint4*>(home[j]);
      }
    }
    int found = 0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const uint32_t ki = kk[j];
#pragma unroll
      for (int q = 0; q < kMaxMultiK; ++q) produced[q] += (unsigned)__popcll(__ballot(go[j] && ki == (uint32_t)q));
      const unsigned long long v = hh[j] + 1, key = (unsigned long long)sv[j].x | ((unsigned long long)sv[j].y << 32);
      found += __popcll(__ballot(go[j] && key == v));
      if (!go[j]) continue;
      if (key == v) {  // a repeat: nothing to do once its counter is saturated
        if (!(A->cs && sv[j].z >= A->cs)) atomicAdd(&home[j]->cnt, 1u);
      } else if (!table_add(A->tab[ki], hh[j] >> A->shift[ki], hh[j], 1u, A->cs, key == 0ull ? 0u : 1u)) {
        atomicAdd(A->counters[ki] + 2, 1ull);
      }
    }
    slot_first = order ? order == 2u : 2 * found > n;
    wave_lds_sync();
    n = 0;
  }

  template <int KI>
  __device__ __forceinline__ void offer(bool hit, uint64_t h, int lane) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    if (m == 0) return;
    if (hit) {
      const int at = n + __popcll(m & ((1ull << lane) - 1ull));
      lds_h[at] = h;
      lds_k[at] = (uint8_t)KI;
    }
    n += __popcll(m);
    if (n > kCandBuf - 64) flush(lane);
  }
  template <int KI>
  __device__ __forceinline__ void offer2(bool a, bool b, uint64_t h, int lane) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(a) & __builtin_amdgcn_ballot_w64(b);
    if (m == 0) return;
    if (a && b) {
      const int at = n + __popcll(m & ((1ull << lane) - 1ull));
      lds_h[at] = h;
      lds_k[at] = (uint8_t)KI;
    }
    n += __popcll(m);
    if (n > kCandBuf - 64) flush(lane);
  }
};

template <int... KS>
struct KList {
  static constexpr int N = sizeof...(KS);
  static constexpr int v[sizeof...(KS)] = {KS...};
  static constexpr int kmax = v[N - 1];
  static constexpr int kmin = v[0];
};

// One lane walks its read; per base it pushes into the roller at the largest k and hashes the k-mer of every k that
// can be complete at this position.  MODE as in mg_sketch.hip: 0 any tile, 1 clean tile of equally long reads,
// 2 clean tile of ragged reads.
// CODES: src is the wavefront's nibble-packed LDS stage and `start` the nibble index of this lane's read; otherwise
// src points at the read's ASCII bases in HBM (a tile that does not fit the stage; MODE 0 only).
template <class KL, bool CODES, int MODE, int HM, class SINK>
__device__ __forceinline__ void walk_reads_multi(const uint8_t* src, uint32_t start, uint32_t len, uint32_t maxlen_v,
                                                 const MultiArgs& A, SINK& sink, uint64_t (&kmers)[kMaxMultiK], int lane,
                                                 const uint64_t* htab) {
  constexpr int KMAX = KL::kmax;
  static_assert(CODES || MODE == 0, "the clean walks read the LDS stage");
  Roller<KMAX> roll;
  roll.reset();
  CodeStream cs;
  if constexpr (CODES) cs.open(src, start);
  const uint32_t maxlen = __builtin_amdgcn_readfirstlane(maxlen_v);
  uint64_t hmax[KL::N];  // (A is the kernel argument itself: these are scalar registers)
#pragma unroll
  for (int i = 0; i < KL::N; ++i) hmax[i] = A.hmax[i];
  // positions [0, kmin - 1): no k-mer of any k is complete — rolled in without hashing (wave-uniform)
  constexpr uint32_t kWarm = (uint32_t)(KL::kmin - 1);
  const uint32_t warm = kWarm < maxlen ? kWarm : maxlen;
  if constexpr (MODE != 0) {
    constexpr bool RAGGED = MODE == 2;
    for (uint32_t pos = 0; pos < warm; ++pos) roll.push_clean(cs.at(pos) & 3u);
    // positions [kmin - 1, kmax - 1): only the smaller k are complete — a scalar condition per k (the ks ascend, so the
    // tests nest)
    // (the hash is computed ONCE, into a variable: written twice — as the threshold test and as the value offered —
    // the four-k kernel ended up with 1.6 x the multiplies, the optimiser no longer merging the two)
#ifdef MG_K1_SEQUENTIAL  // (A/B builds: every position through the per-k conditional form)
    constexpr uint32_t kAll = 0xffffffffu;
#else
    constexpr uint32_t kAll = (uint32_t)(KMAX - 1);
#endif
    const uint32_t some_end = kAll < maxlen ? kAll : maxlen;
    for (uint32_t pos = warm; pos < some_end; ++pos) {
      roll.push_clean(cs.at(pos) & 3u);
      auto one = [&]<int I>() {
        const uint64_t h = hash_suffix<KL::v[I], KMAX, HM>(roll, htab);
        if constexpr (RAGGED) sink.template offer2<I>(pos < len, h <= hmax[I], h, lane);
        else sink.template offer<I>(h <= hmax[I], h, lane);
      };
      [&]<int... I>(std::integer_sequence<int, I...>) {
        ((pos + 1 >= (uint32_t)KL::v[I] ? one.template operator()<I>() : (void)0), ...);
      }(std::make_integer_sequence<int, KL::N>{});
    }
    // positions [kmax - 1, maxlen): EVERY k is complete (100 of a 150-base read's positions at {21,31,51}).  Two hashes
    // of the position at a time in one straight line — their table look-ups (mg_kmer.h) are issued together and the two
    // independent MurmurHash3 chains interleave, instead of each k's look-ups being waited for just before its own
    // multiplies — then their threshold tests and the (branchy) candidate hand-over
    for (uint32_t pos = some_end; pos < maxlen; ++pos) {
      roll.push_clean(cs.at(pos) & 3u);
#ifndef MG_K1_ALL  // the hashes two at a time (MG_K1_ALL, A/B builds: all of them at once)
      auto pair = [&]<int I0>() {
        constexpr int I1 = I0 + 1 < KL::N ? I0 + 1 : I0;
        const uint64_t ha = hash_suffix<KL::v[I0], KMAX, HM>(roll, htab);
        const uint64_t hb = I1 != I0 ? hash_suffix<KL::v[I1], KMAX, HM>(roll, htab) : 0ull;
        if constexpr (RAGGED) sink.template offer2<I0>(pos < len, ha <= hmax[I0], ha, lane); else sink.template offer<I0>(ha <= hmax[I0], ha, lane);
        if constexpr (I1 != I0) {
          if constexpr (RAGGED) sink.template offer2<I1>(pos < len, hb <= hmax[I1], hb, lane); else sink.template offer<I1>(hb <= hmax[I1], hb, lane);
        }
      };
      pair.template operator()<0>();
      if constexpr (KL::N > 2) pair.template operator()<2>();
#else
      uint64_t h[KL::N];
      [&]<int... I>(std::integer_sequence<int, I...>) {
        ((h[I] = hash_suffix<KL::v[I], KMAX, HM>(roll, htab)), ...);
      }(std::make_integer_sequence<int, KL::N>{});
      [&]<int... I>(std::integer_sequence<int, I...>) {
        ((RAGGED ? sink.template offer2<I>(pos < len, h[I] <= hmax[I], h[I], lane) : sink.template offer<I>(h[I] <= hmax[I], h[I], lane)), ...);
      }(std::make_integer_sequence<int, KL::N>{});
#endif
    }
#pragma unroll
    for (int i = 0; i < KL::N; ++i)  // the tile's k-mers in closed form, summed over the wavefront into a scalar
      kmers[i] += wave_sum_u64(len >= (uint32_t)KL::v[i] ? len - (uint32_t)KL::v[i] + 1u : 0u);
    return;
  }
  uint32_t nk[KL::N];
#pragma unroll
  for (int i = 0; i < KL::N; ++i) nk[i] = 0;
  for (uint32_t pos = 0; pos < maxlen; ++pos) {
    uint32_t c;
    if constexpr (CODES) { c = cs.at(pos); c = pos < len ? c : 4u; }
    else c = pos < len ? encode1(src[pos]) : 4u;
    roll.push(c);
    roll.run = c < 4u ? roll.run : 0;
    if (pos < warm) continue;  // (scalar)
    auto one = [&]<int I>() {
      const uint64_t h = hash_suffix<KL::v[I], KMAX, HM>(roll, htab);
      nk[I] += roll.run >= KL::v[I] ? 1u : 0u;
      sink.template offer2<I>(roll.run >= KL::v[I], h <= hmax[I], h, lane);
    };
    [&]<int... I>(std::integer_sequence<int, I...>) {
      ((pos + 1 >= (uint32_t)KL::v[I] ? one.template operator()<I>() : (void)0), ...);
    }(std::make_integer_sequence<int, KL::N>{});
  }
#pragma unroll
  for (int i = 0; i < KL::N; ++i) kmers[i] += wave_sum_u64(nk[i]);
}

template <class KL, int HM, bool RESIDENT>
__device__ __forceinline__ void sketch_reads_multi_body(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
                                                        uint64_t nreads, const MultiArgs& args, unsigned stage_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ MultiArgs s_args;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) s_args = args;
  const uint64_t* htab = fill_hash_tables();  // MurmurHash3's first multiplies (mg_kmer.h); ends in the workgroup barrier
  uint8_t* stage = smem + (size_t)wave * stage_bytes;
  uint8_t* cand = smem + (size_t)kWavesPerBlock * stage_bytes;
  uint64_t* cbuf = reinterpret_cast<uint64_t*>(cand) + wave * kCandBuf;
  uint8_t* kbuf = cand + (size_t)kWavesPerBlock * kCandBuf * sizeof(uint64_t) + wave * kCandBuf;
  MultiSink<RESIDENT> sink{cbuf, kbuf, &s_args, 0, {0, 0, 0, 0}, args.order == 2u, args.order};
  if constexpr (RESIDENT) {
    __shared__ uint64_t s_lst[kWavesPerBlock][kMaxMultiK + kMaxMultiK / 2];
    sink.lst = s_lst[wave];
    if (lane < kMaxMultiK + kMaxMultiK / 2) sink.lst[lane] = 0;  // no chunk yet
    wave_lds_sync();
  }
  uint64_t kmers[kMaxMultiK] = {0, 0, 0, 0};
  const uint64_t ntiles = (nreads + 63) / 64;
  for (uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave; tile < ntiles;
       tile += (uint64_t)gridDim.x * kWavesPerBlock) {
    const uint64_t r = tile * 64 + lane;
    uint64_t beg = 0, end = 0;
    if (r < nreads) { beg = offsets[r]; end = offsets[r + 1]; }
    const uint64_t len = end - beg;
    const uint64_t maxlen = wave_max_u64(len);
    const uint64_t t_beg = __shfl(beg, 0, 64);
    const uint64_t t_end = wave_max_u64(end);
    const uintptr_t a_first = reinterpret_cast<uintptr_t>(bases) + t_beg;
    const uintptr_t a0 = a_first & ~(uintptr_t)15;
    const uint64_t shift = a_first - a0;
    const uint64_t nbytes = shift + (t_end - t_beg);
    if (nbytes <= 2ull * stage_bytes) {
      // coalesced HBM -> LDS copy of the whole tile (16 B per lane per step), bases -> 4-bit codes on the way
      const uint4* g = reinterpret_cast<const uint4*>(a0);
      uint2* s = reinterpret_cast<uint2*>(stage);
      uint32_t bad = 0;
      for (uint64_t i = lane; i * 16 < nbytes; i += 64) {
        const uint4 v = g[i];
        const uint2 p{pack4(encode4(v.x)) | (pack4(encode4(v.y)) << 16), pack4(encode4(v.z)) | (pack4(encode4(v.w)) << 16)};
        bad |= (p.x | p.y) & 0x44444444u;
        s[i] = p;
      }
      wave_lds_sync();
      const uint32_t nstart = (uint32_t)(shift + (beg - t_beg));
      if (__ballot(bad != 0) != 0ull)
        walk_reads_multi<KL, true, 0, HM, MultiSink<RESIDENT>>(stage, nstart, (uint32_t)len, (uint32_t)maxlen, args, sink, kmers, lane, htab);
      else if (__ballot(len != maxlen) == 0ull)
        walk_reads_multi<KL, true, 1, HM, MultiSink<RESIDENT>>(stage, nstart, (uint32_t)len, (uint32_t)maxlen, args, sink, kmers, lane, htab);
      else
        walk_reads_multi<KL, true, 2, HM, MultiSink<RESIDENT>>(stage, nstart, (uint32_t)len, (uint32_t)maxlen, args, sink, kmers, lane, htab);
      wave_lds_sync();
    } else {
      walk_reads_multi<KL, false, 0, HM, MultiSink<RESIDENT>>(bases + beg, 0u, (uint32_t)len, (uint32_t)maxlen, args, sink, kmers, lane, htab);
    }
  }
  sink.flush(lane);
  if constexpr (RESIDENT)
    while (sink.n) sink.flush(lane);  // (candidates that went round again: at most kMaxHops times)
#pragma unroll
  for (int i = 0; i < KL::N; ++i) {  // (wave totals already: every lane holds the same value)
    if (lane == 0 && kmers[i]) atomicAdd(args.counters[i] + 1, (unsigned long long)kmers[i]);
    if (lane == 0 && sink.produced[i]) atomicAdd(args.counters[i], (unsigned long long)sink.produced[i]);
  }
}

// 157 VGPRs for {21,31,51}: three wavefronts per SIMD (held to 128 — four per SIMD, fifty spilled registers — 16.9
// against 15.9 ms per 10M reads at configs[2], 49 against 39 ms in the dense regime).  {30,40,50,60} wants 179: held to
// 168 (eleven spilled) for three wavefronts per SIMD, 22.6 against 23.9 ms.
// Measured on configs[2] (10M reads; tools/ab_k1.sh, profiles/r03/k1_variants.txt): per-k conditional form at 127 VGPRs
// 14.60 ms alone / 21.72 ms for {30,40,50,60} / 12.91 ms per pipelined pass; all hashes of a position in one straight
// line (157 VGPRs, three wavefronts per SIMD) 14.59 / 20.78 / 13.61 — faster alone, slower beside the next pass's
// launch; two hashes at a time held to 128 VGPRs (11 and 41 spilled registers) 14.52 / 20.79 / 12.90: the default.
#if defined(MG_K1_SEQUENTIAL) || defined(MG_K1_ALL)
#define MG_K1_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#else
#define MG_K1_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
template <class KL, int HM>
__global__ __launch_bounds__(kBlock) MG_K1_WAVES_ATTR void k_sketch_reads_multi(const uint8_t* __restrict__ bases,
                                                               const uint64_t* __restrict__ offsets, uint64_t nreads,
                                                               const MultiArgs args, unsigned stage_bytes) {
  sketch_reads_multi_body<KL, HM, false>(bases, offsets, nreads, args, stage_bytes);
}
// ... against resident indexes (args.epoch != 0)
template <class KL, int HM>
__global__ __launch_bounds__(kBlock) MG_K1_WAVES_ATTR void k_sketch_reads_multi_resident(const uint8_t* __restrict__ bases,
                                                               const uint64_t* __restrict__ offsets, uint64_t nreads,
                                                               const MultiArgs args, unsigned stage_bytes) {
  sketch_reads_multi_body<KL, HM, true>(bases, offsets, nreads, args, stage_bytes);
}

template <class KL>
static int launch_multi(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, const MultiArgs& a,
                        unsigned stage_bytes) {
  Context& c = ctx();
  const size_t lds = (size_t)kWavesPerBlock * (stage_bytes + kCandBuf * (sizeof(uint64_t) + 1));
  const uint64_t ntiles = (nreads + 63) / 64;
  unsigned per_cu = (unsigned)(160 * 1024 / (lds + sizeof(MultiArgs) + 64 + 256 + kHashTabEntries * sizeof(uint64_t)));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  if (c.a_side && c.is_stage_a(c.stream) && c.a_side_wg_per_cu && per_cu > c.a_side_wg_per_cu) per_cu = c.a_side_wg_per_cu;
  const unsigned grid = grid_for(ntiles, kWavesPerBlock, (unsigned)c.num_cus * per_cu);
  ProfScope ps("sketch_reads");
  if (a.epoch && c.hash_mode == kHashCmash)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sketch_reads_multi_resident<KL, kHashCmash>), dim3(grid), dim3(kBlock), lds, c.stream, d_bases,
                       d_offsets, nreads, a, stage_bytes);
  else if (a.epoch)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sketch_reads_multi_resident<KL, kHashCanonical>), dim3(grid), dim3(kBlock), lds, c.stream, d_bases,
                       d_offsets, nreads, a, stage_bytes);
  else if (c.hash_mode == kHashCmash)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sketch_reads_multi<KL, kHashCmash>), dim3(grid), dim3(kBlock), lds, c.stream, d_bases,
                       d_offsets, nreads, a, stage_bytes);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sketch_reads_multi<KL, kHashCanonical>), dim3(grid), dim3(kBlock), lds, c.stream, d_bases,
                       d_offsets, nreads, a, stage_bytes);
  MG_HIP(hipGetLastError());
  return MG_OK;
}

bool sketch_reads_multi_supported(const int* ks, int nk) {
  auto is = [&](std::initializer_list<int> want) {
    if ((int)want.size() != nk) return false;
    int i = 0;
    for (int k : want)
      if (ks[i++] != k) return false;
    return true;
  };
  return is({21, 31, 51}) || is({30, 40, 50, 60});
}

// One launch for all k; `tables` etc. are per k (ascending k).  MG_ERR_ARG when the k set has no instantiation.
int launch_sketch_reads_multi(const int* ks, int nk, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads,
                              const MultiKTable* tabs, unsigned stage_bytes) {
  if (nk < 1 || nk > kMaxMultiK) return fail(MG_ERR_ARG, "between 1 and %d k per fused launch", kMaxMultiK);
  MultiArgs a{};
  for (int i = 0; i < nk; ++i) {
    a.hmax[i] = tabs[i].hmax; a.tab[i] = tabs[i].tab; a.counters[i] = tabs[i].counters;
    a.fbits[i] = tabs[i].filter ? tabs[i].filter->bits.as<uint32_t>() : nullptr;
    a.fmask[i] = tabs[i].filter ? tabs[i].filter->mask : 0ull;
    a.shift[i] = tabs[i].shift;
    if (tabs[i].epoch) {  // (all of a call's k or none)
      a.fbits[i] = nullptr; a.fmask[i] = 0; a.epoch = tabs[i].epoch;
      a.list[i] = tabs[i].list; a.listcap[i] = tabs[i].listcap;
    }
  }
  a.ablate = (uint32_t)dbg("resident_ablate");
  a.cs = stage_a_cs_word() & kCsMask;
  a.order = stage_a_cs_word() >> 30;
  if (nk == 3 && ks[0] == 21 && ks[1] == 31 && ks[2] == 51)
    return launch_multi<KList<21, 31, 51>>(d_bases, d_offsets, nreads, a, stage_bytes);
  if (nk == 4 && ks[0] == 30 && ks[1] == 40 && ks[2] == 50 && ks[3] == 60)
    return launch_multi<KList<30, 40, 50, 60>>(d_bases, d_offsets, nreads, a, stage_bytes);
  return fail(MG_ERR_ARG, "no fused stage-A kernel for this k set");
}

}  // namespace mg
