// mg_kmer.h — device-side k-mer roller + MurmurHash3_x64_128 specialised on k.
//
// One lane walks one sequence and keeps, in registers, both strands of the current window 2-bit packed (first base most
// significant): the canonical choice "lexicographically smaller of k-mer and reverse complement" is one integer
// compare, and the canonical k-mer is the selected packed word.
//
// The hash is MurmurHash3_x64_128(seed 0), first 64 bits, of the canonical k-mer's ASCII bytes (normative statement:
// oracle/mg_oracle.c, canonical_hash).  Its ASCII is never materialised: the only thing MurmurHash3 does with a key
// word (8 bytes) before anything non-linear is multiply it by a constant — k1 *= C1, k2 *= C2 — and a dword of four
// ASCII bases takes 256 values, so
//     (lo + hi * 2^32) * C  mod 2^64  =  T[g_lo]  +  (low32(T[g_hi]) << 32),        T[g] = ASCII4(g) * C mod 2^64,
// g = the 8-bit packed code of the four bases: one 8-byte and one 4-byte LDS read and one 32-bit add replace a 64-bit
// multiply (two v_mul_lo_u32 + v_mad_u64_u32 + v_add3_u32 = 17 issue cycles on gfx950, profiles/r03/valu_classes.json)
// AND the two rolling ASCII windows, the per-k byte funnel shifts and the per-dword strand selects that fed it (round
// 2's design, kept under tools/experiments/mg_kmer_ascii_windows.h.txt: 26 + 6 v_alignbyte_b32 + 27 v_cndmask_b32 per
// step of the fused {21,31,51} kernel).  14 of that kernel's 40 multiplies per position are such first multiplies.
// Key words that end inside a group of four (K mod 4 != 0) use tables of 4, 16 and 64 entries for groups of 1, 2 and 3
// bases.  A workgroup builds the tables (2 constants x 340 entries x 8 B = 5.4 KB) when it starts: fill_hash_tables().
#pragma once
#ifndef MG_HOST_CHECK
#include <hip/hip_runtime.h>
#else
// tests/host_kmer_check.cpp compiles this header with g++ (no GPU in the build container): the roller and the
// table-driven MurmurHash3 below then run on the host, against the oracle, for every k (tests/test_kmer_header_host.py)
#define __device__
#define __host__
#define __forceinline__ inline
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned s) {
  return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (s & 31u));
}
#endif

#include <cstdint>
#include <utility>

namespace mg {

// rotl of a 64-bit value by a constant as two funnel shifts over its dwords (v_alignbit_b32: 2 x 4.2 issue cycles).  Written as
// (x << r) | (x >> (64 - r)) the compiler made some of the hash's rotates a v_lshlrev_b64 + v_lshrrev_b64 + two v_or_b32 (12.9).
// MG_ROTL_PLAIN (A/B builds) keeps the shifts.
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) {
#ifndef MG_ROTL_PLAIN
  uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  r &= 63;
  if (r >= 32) { const uint32_t t = lo; lo = hi; hi = t; r -= 32; }
  if (r == 0) return (uint64_t)lo | ((uint64_t)hi << 32);
  const uint32_t nh = __builtin_amdgcn_alignbit(hi, lo, 32u - (uint32_t)r);
  const uint32_t nl = __builtin_amdgcn_alignbit(lo, hi, 32u - (uint32_t)r);
  return (uint64_t)nl | ((uint64_t)nh << 32);
#else
  return (x << r) | (x >> (64 - r));
#endif
}

// x * C mod 2^64 for a 64-bit constant as three chained v_mad_u64_u32 (lo x lo in full; + lo x hi; + hi x lo): 13.7 issue
// cycles + two register moves (the chain's 64-bit addends want register pairs) against 17.3 for the compiler's form — v_mad_u64_u32 + 2 x v_mul_lo_u32 (the cross terms) + v_add3_u32
// (profiles/r03/valu_classes.json).  One dependent chain instead of three independent multiplies: rounds 1 and 2 measured
// no gain from it; with the ASCII windows gone and two hashes of a position interleaved (mg_sketch_multi.hip) the fused
// kernel alone takes 14.28 against 14.39 ms per 10M reads and the pipelined pass 12.24 against 13.04 ms (round 3).
// MG_MUL64_COMPILER (A/B builds) keeps the compiler's form.
template <uint64_t C>
__device__ __forceinline__ uint64_t mul64c(uint64_t x) {
#if !defined(MG_MUL64_COMPILER) && !defined(MG_HOST_CHECK)
  const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
  constexpr uint32_t cl = (uint32_t)C, ch = (uint32_t)(C >> 32);
  uint64_t p, t, u, c0, c1, c2;  // (c*: the carry-outs nobody reads; any SGPR pair)
#ifdef MG_MUL64_SPLIT
  // Round 4, measured and NOT shipped (A/B build: -DMG_MUL64_SPLIT -DMG_MUL5_LSHL): the cross terms chained on their own (lo x hi,
  // + hi x lo: the sum's low dword is what counts) and ONE 32-bit add onto the full product's high dword, in place — three
  // v_mad_u64_u32 + v_add_u32 and no register moves, where the chained form below moves the full product's high dword into a
  // register pair for the second multiply and the last result's low dword up (two v_mov_b32 per multiply).  With mul5_add's
  // v_lshl_add_u64 form the one-k kernel's hot loop went from 432 to 400 VALU instructions per pair of positions (53 v_mov_b32 -> 1,
  // 94 v_mad_u64_u32 -> 70, 15 v_lshl_add_u64 -> 39) and took exactly as long: 6.18 ms per 10M reads alone, 5.67 per pipelined pass
  // (mode 1: 10.8 / 9.8).  The moves are not what the SIMDs wait for; the count of 64-bit operations (109 per pair either way) is.
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(c0) : "v"(xl), "s"(cl));
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(t), "=s"(c1) : "v"(xl), "s"(ch));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(u), "=s"(c2) : "v"(xh), "s"(cl), "v"(t));
  uint32_t hi;  // (opaque on purpose: as C the compiler makes it a 64-bit sum again — a move into a register pair + v_lshl_add_u64)
  asm("v_add_u32 %0, %1, %2" : "=v"(hi) : "v"((uint32_t)(p >> 32)), "v"((uint32_t)u));
  return (uint64_t)(uint32_t)p | ((uint64_t)hi << 32);
#else
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(c0) : "v"(xl), "s"(cl));
  const uint64_t hi0 = p >> 32;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t), "=s"(c1) : "v"(xl), "s"(ch), "v"(hi0));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(u), "=s"(c2) : "v"(xh), "s"(cl), "v"(t));
  return (uint64_t)(uint32_t)p | (u << 32);
#endif
#else
  return x * C;
#endif
}

// h * 5 + C (the end of a body block's h1 / h2 step).  The compiler's form: two v_mad_u64_u32 (low dword x 5 + C, high dword x 5
// + the carry) and two register moves between their pairs; six per hash of a 51-mer.  MG_MUL5_LSHL (A/B builds): two
// v_lshl_add_u64 — (h << 2) + h, then + C out of an SGPR pair.  Measured on the kernel with the alignbit rotates, each change alone,
// same run, pipelined pass per 10M reads: shipped 5.22 ms; this 5.32-5.36; MG_MUL64_SPLIT 5.44-5.52; low dword x 5 + C as one
// v_mad_u64_u32 and the high dword's x 5 in 32-bit operations 5.39-5.45; the hash's 64-bit additions as v_add_co_u32 +
// v_addc_co_u32 instead of v_lshl_add_u64 5.32-5.33.  The compiler's forms stay.
template <uint64_t C>
__device__ __forceinline__ uint64_t mul5_add(uint64_t h) {
#if defined(MG_MUL5_LSHL) && !defined(MG_HOST_CHECK)
  uint64_t a, r;
  constexpr uint64_t c = C;
  asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(a) : "v"(h));
  asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "s"(c));
  return r;
#else
  return h * 5 + C;
#endif
}

__device__ __forceinline__ uint64_t fmix64(uint64_t v) {
  v ^= v >> 33;
  v = mul64c<0xff51afd7ed558ccdULL>(v);
  v ^= v >> 33;
  v = mul64c<0xc4ceb9fe1a85ec53ULL>(v);
  v ^= v >> 33;
  return v;
}

constexpr uint64_t kMurmurC1 = 0x87c37b91114253d5ULL, kMurmurC2 = 0x4cf5ad432745937fULL;

// Which definition of a k-mer's hash a kernel is instantiated for (mg_set_hash_mode; oracle/mg_oracle.c: g_hash_mode):
//   kHashCanonical  MurmurHash3 of the lexicographically smaller strand, the full 64 bits (KMC's canonical k-mer; default)
//   kHashCmash      min(MurmurHash3(kmer), MurmurHash3(revcomp)) mod 9999999999971 — CMash's CountEstimator as SURVEY.md
//                   §8(c) recollects it (unverified: its source is not under /root/reference); two hashes per k-mer
constexpr int kHashCanonical = 0, kHashCmash = 1;
// (table builder only: kHashCmash with the kept strand in bit 63 — set when the reverse complement's hash is the smaller or
// equal one, the strand CMash's CountEstimator.add keeps; mode-1 hashes are below 2^44)
constexpr int kHashCmashTagged = 2;
// (table builder only, `build_db --sketch_hash forward`: MurmurHash3 of the k-mer AS IT STANDS in the genome mod the prime — what
// selects a genome's sketch when CMash trains without reverse complements, as recollected; what a k-mer MATCHES by stays
// kHashCanonical / kHashCmash)
constexpr int kHashForward = 3;
constexpr uint64_t kCmashPrime = 9999999999971ULL;

// Base decode: A,C,G,T (either case) -> 0..3 (lexicographic order), anything else -> invalid.
// idx = (b & 0xDF) - 'A'; valid letters sit at idx 0 (A), 2 (C), 6 (G), 19 (T).
__device__ __forceinline__ bool decode_base(uint32_t b, uint32_t& code) {
  uint32_t idx = (b & 0xDFu) - 0x41u;
  bool ok = idx < 20u && ((0x80045u >> idx) & 1u);
  uint32_t x = (b >> 1) & 3u;  // A:0 C:1 T:2 G:3
  code = x ^ (x >> 1);         // A:0 C:1 G:2 T:3
  return ok;
}

// ---- the first-multiply tables -------------------------------------------------------------------------------------
// Per constant: [0, 256) groups of four bases, then groups of 1 (4 entries), 2 (16) and 3 (64) bases.
constexpr int kHashTabPer = 256 + 4 + 16 + 64;
constexpr int kHashTabEntries = 2 * kHashTabPer;  // C1's tables, then C2's
__host__ __device__ constexpr int hash_tab_part(int nbases) { return nbases == 4 ? 0 : nbases == 1 ? 256 : nbases == 2 ? 260 : 276; }

// Entry e of the tables: the ASCII of its group (first base = byte 0 = the group's most significant code) times its constant.
__device__ __forceinline__ uint64_t hash_tab_entry(int e) {
  const uint64_t c = e >= kHashTabPer ? kMurmurC2 : kMurmurC1;
  e = e >= kHashTabPer ? e - kHashTabPer : e;
  int m = 4, g = e;
  if (e >= 276) { m = 3; g = e - 276; }
  else if (e >= 260) { m = 2; g = e - 260; }
  else if (e >= 256) { m = 1; g = e - 256; }
  uint64_t ascii = 0;
  for (int b = 0; b < m; ++b) {
    const uint32_t code = (uint32_t)(g >> (2 * (m - 1 - b))) & 3u;
    ascii |= (uint64_t)((0x54474341u >> (8 * code)) & 0xffu) << (8 * b);  // "ACGT"[code]
  }
  return ascii * c;
}

#ifndef MG_HOST_CHECK
// The workgroup's tables (static LDS of whatever kernel calls this: the address space stays visible to the optimiser,
// so every look-up is a ds_read with the table's offset as its immediate).
__device__ __forceinline__ const uint64_t* hash_tables() {
  __shared__ __attribute__((aligned(16))) uint64_t s_hash_tab[kHashTabEntries];
  return s_hash_tab;
}

// Every thread of the workgroup calls this once before its first hash (ends in a workgroup barrier).
__device__ __forceinline__ const uint64_t* fill_hash_tables() {
  uint64_t* t = const_cast<uint64_t*>(hash_tables());
  for (int e = (int)threadIdx.x; e < kHashTabEntries; e += (int)blockDim.x) t[e] = hash_tab_entry(e);
  __syncthreads();
  return t;
}
#endif

// A canonical k-mer, 2-bit packed (first base most significant), as up to four dwords (d[0] least significant).
struct Packed {
  uint32_t d[4];
};
__device__ __forceinline__ Packed make_packed(uint64_t lo, uint64_t hi) {
  return Packed{{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)}};
}

// 8 x the W-bit field at bit LB of p: the byte offset of an 8-byte table entry.
template <int LB, int W>
__device__ __forceinline__ uint32_t field_x8(const Packed& p) {
  static_assert(W >= 2 && W <= 8 && LB >= 0 && LB + W <= 128, "field out of range");
  constexpr uint32_t M = ((1u << W) - 1u) << 3;
  constexpr int S = LB - 3;  // the shift that leaves the field at bit 3
  if constexpr (S < 0) {
    return (p.d[0] << (-S)) & M;
  } else {
    constexpr int Q = S / 32, R = S % 32;
    if constexpr (R == 0) return p.d[Q] & M;
    else if constexpr (R + W + 3 <= 32) return (p.d[Q] >> R) & M;
    else return __builtin_amdgcn_alignbit(Q + 1 < 4 ? p.d[Q + 1] : 0u, p.d[Q], R) & M;
  }
}

// Key word number WORD (bytes 8*WORD .. of the K-byte key) times C1 (CI = 0) or C2 (CI = 1), mod 2^64.
template <int K, int WORD, int CI>
__device__ __forceinline__ uint64_t key_word_times_c(const Packed& p, const uint64_t* tab) {
  constexpr int S = 8 * WORD, E = K < S + 8 ? K : S + 8;
  static_assert(E > S, "no such key word");
  constexpr int NLO = E - S < 4 ? E - S : 4, NHI = E - S - NLO;
  const uint8_t* base = reinterpret_cast<const uint8_t*>(tab + CI * kHashTabPer);
  uint64_t v = *reinterpret_cast<const uint64_t*>(base + 8 * hash_tab_part(NLO) + field_x8<2 * (K - S - NLO), 2 * NLO>(p));
  if constexpr (NHI > 0) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(base + 8 * hash_tab_part(NHI) + field_x8<2 * (K - S - NLO - NHI), 2 * NHI>(p));
    v = (uint64_t)(uint32_t)v | ((uint64_t)((uint32_t)(v >> 32) + u) << 32);  // one 32-bit add on the high dword
  }
  return v;
}

// MurmurHash3_x64_128(ASCII of the packed K-mer p, seed 0) -> first 64 bits.
template <int K>
__device__ __forceinline__ uint64_t murmur3_h1_packed(const Packed& p, const uint64_t* tab) {
  constexpr uint64_t C1 = kMurmurC1, C2 = kMurmurC2;
  constexpr int NBLK = K / 16, TAIL = K & 15;
  uint64_t h1 = 0, h2 = 0;
  // 16-byte body blocks, expanded with compile-time indices
  auto body = [&]<int B>() {
    uint64_t k1 = key_word_times_c<K, 2 * B, 0>(p, tab);
    uint64_t k2 = key_word_times_c<K, 2 * B + 1, 1>(p, tab);
    k1 = rotl64(k1, 31); k1 = mul64c<C2>(k1); h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = mul5_add<0x52dce729ULL>(h1);
    k2 = rotl64(k2, 33); k2 = mul64c<C1>(k2); h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = mul5_add<0x38495ab5ULL>(h2);
  };
  [&]<int... B>(std::integer_sequence<int, B...>) { (body.template operator()<B>(), ...); }(std::make_integer_sequence<int, NBLK>{});
  if constexpr (TAIL > 8) {
    uint64_t k2 = key_word_times_c<K, 2 * NBLK + 1, 1>(p, tab);
    k2 = rotl64(k2, 33); k2 = mul64c<C1>(k2); h2 ^= k2;
  }
  if constexpr (TAIL > 0) {
    uint64_t k1 = key_word_times_c<K, 2 * NBLK, 0>(p, tab);
    k1 = rotl64(k1, 31); k1 = mul64c<C2>(k1); h1 ^= k1;
  }
  h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2;
  return h1;
}

template <int K>
struct Roller {
  static_assert(K >= 1 && K <= 64, "k out of range");
  static constexpr int NW = (K + 31) / 32;  // 64-bit words of the 2-bit form

  uint64_t pf_lo, pf_hi, pr_lo, pr_hi;  // 2-bit packed strands (hi unused when K <= 32)
  int run;                              // consecutive valid bases seen

  __device__ __forceinline__ void reset() {
    pf_lo = pf_hi = pr_lo = pr_hi = 0;
    run = 0;
  }

  // Append one base with code c (0..3; anything else only in bits that `run` keeps from being used).
  __device__ __forceinline__ void push(uint32_t c) {
    c &= 3u;
    const uint64_t cc = 3u - c;
    if constexpr (NW == 1) {
      constexpr uint64_t M = K == 32 ? ~0ull : ((1ull << (2 * K)) - 1ull);
      pf_lo = ((pf_lo << 2) | c) & M;
      pr_lo = (pr_lo >> 2) | (cc << (2 * (K - 1)));
    } else {
      constexpr int HB = 2 * (K - 32);  // bits used in the high word
      constexpr uint64_t MH = HB == 64 ? ~0ull : ((1ull << HB) - 1ull);
      pf_hi = ((pf_hi << 2) | (pf_lo >> 62)) & MH;
      pf_lo = (pf_lo << 2) | c;
      pr_lo = (pr_lo >> 2) | (pr_hi << 62);
      pr_hi = (pr_hi >> 2) | (cc << (HB - 2));
    }
    ++run;
  }

  // The same for a base known to be one of ACGT, in a stretch where nobody looks at `run` (a tile of equally long
  // reads without a single invalid base: whether a k-mer is complete is then a function of the position alone).
  __device__ __forceinline__ void push_clean(uint32_t c) {
    const int keep = run;
    push(c);      // (the code is 0..3: the compiler drops the mask; the counter is restored, i.e. never computed)
    run = keep;
  }

  __device__ __forceinline__ bool full() const { return run >= K; }

  __device__ __forceinline__ bool forward_is_canonical() const {
    if constexpr (NW == 1) return pf_lo <= pr_lo;
    return pf_hi < pr_hi || (pf_hi == pr_hi && pf_lo <= pr_lo);
  }

  // The k-mer's hash under definition HM.  tab: fill_hash_tables().
  template <int HM = kHashCanonical>
  __device__ __forceinline__ uint64_t hash(const uint64_t* tab) const {
    if constexpr (HM == kHashForward) return murmur3_h1_packed<K>(make_packed(pf_lo, NW == 1 ? 0ull : pf_hi), tab) % kCmashPrime;
    if constexpr (HM == kHashCmash || HM == kHashCmashTagged) {
      const uint64_t a = murmur3_h1_packed<K>(make_packed(pf_lo, NW == 1 ? 0ull : pf_hi), tab);
      const uint64_t b = murmur3_h1_packed<K>(make_packed(pr_lo, NW == 1 ? 0ull : pr_hi), tab);
      const uint64_t h = (a < b ? a : b) % kCmashPrime;
      if constexpr (HM == kHashCmashTagged) return h | (b <= a ? 1ull << 63 : 0ull);
      return h;
    }
    const bool fw = forward_is_canonical();
    const uint64_t lo = fw ? pf_lo : pr_lo, hi = NW == 1 ? 0ull : (fw ? pf_hi : pr_hi);
    return murmur3_h1_packed<K>(make_packed(lo, hi), tab);
  }
};

// Hash of the canonical K-mer that ENDS at the newest base of a Roller<KMAX>, K <= KMAX: one roller at the largest
// k of a multi-k query serves every smaller k.  The K-mer is the SUFFIX of the forward window (the low 2K bits of the
// forward 2-bit form) and its reverse complement is the PREFIX of the reverse-complement window (the top 2K bits of
// the reverse 2-bit form).
template <int K, int KMAX, int HM = kHashCanonical>
__device__ __forceinline__ uint64_t hash_suffix(const Roller<KMAX>& R, const uint64_t* tab) {
  static_assert(K >= 1 && K <= KMAX, "sub-k must not exceed the roller's k");
  if constexpr (K == KMAX) {
    return R.template hash<HM>(tab);
  } else {
    constexpr int S = 2 * (KMAX - K);  // the reverse form's K-mer sits S bits up
    uint64_t f_lo = R.pf_lo, f_hi = KMAX > 32 ? R.pf_hi : 0ull;
    uint64_t r_lo, r_hi;
    {
      const uint64_t p_lo = R.pr_lo, p_hi = KMAX > 32 ? R.pr_hi : 0ull;
      if constexpr (S >= 64) { r_lo = S == 64 ? p_hi : (p_hi >> (S - 64)); r_hi = 0; }
      else { r_lo = (p_lo >> S) | (p_hi << (64 - S)); r_hi = p_hi >> S; }  // (S > 0 here: K < KMAX)
    }
    if constexpr (K <= 32) {
      constexpr uint64_t M = K == 32 ? ~0ull : ((1ull << (2 * K)) - 1ull);
      f_lo &= M; r_lo &= M; f_hi = 0; r_hi = 0;
    } else {
      constexpr uint64_t MH = (1ull << (2 * (K - 32))) - 1ull;  // K < KMAX <= 64: at most 62 bits
      f_hi &= MH; r_hi &= MH;
    }
    if constexpr (HM == kHashCmash) {
      const uint64_t a = murmur3_h1_packed<K>(make_packed(f_lo, f_hi), tab);
      const uint64_t b = murmur3_h1_packed<K>(make_packed(r_lo, r_hi), tab);
      return (a < b ? a : b) % kCmashPrime;
    }
    const bool fw = K <= 32 ? (f_lo <= r_lo) : (f_hi < r_hi || (f_hi == r_hi && f_lo <= r_lo));
    return murmur3_h1_packed<K>(make_packed(fw ? f_lo : r_lo, fw ? f_hi : r_hi), tab);
  }
}

#ifndef MG_MAX_K
#define MG_MAX_K 64
#endif
// Calls F.template operator()<K>() for the runtime k; false when k is unsupported.
template <class F, int K = 1>
inline bool dispatch_k(int k, F&& fn) {
  if constexpr (K > MG_MAX_K) {
    return false;
  } else {
    if (k == K) { fn.template operator()<K>(); return true; }
    return dispatch_k<F, K + 1>(k, static_cast<F&&>(fn));
  }
}

}  // namespace mg
