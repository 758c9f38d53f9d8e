// mg_kmer.h — device-side k-mer roller + MurmurHash3_x64_128 specialised on k.
//
// One lane walks one sequence and keeps, in registers,
//   * the forward k-mer and its reverse complement as ASCII bytes (what
//     MurmurHash3 consumes), rolled one byte per base with v_alignbyte, and
//   * both strands 2-bit packed (first base most significant) so that the
//     canonical choice "lexicographically smaller of k-mer and revcomp" is one
//     integer compare.
// The hash is MurmurHash3_x64_128(seed 0), first 64 bits, of the canonical
// ASCII k-mer: block/tail structure is resolved at compile time from K.
// Normative statement: oracle/mg_oracle.c (canonical_hash).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <utility>

namespace mg {

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

__device__ __forceinline__ uint64_t fmix64(uint64_t v) {
  v ^= v >> 33;
  v *= 0xff51afd7ed558ccdULL;
  v ^= v >> 33;
  v *= 0xc4ceb9fe1a85ec53ULL;
  v ^= v >> 33;
  return v;
}

// Base decode: A,C,G,T (either case) -> 0..3 (lexicographic order), anything else -> invalid.
// idx = (b & 0xDF) - 'A'; valid letters sit at idx 0 (A), 2 (C), 6 (G), 19 (T).
__device__ __forceinline__ bool decode_base(uint32_t b, uint32_t& code) {
  uint32_t idx = (b & 0xDFu) - 0x41u;
  bool ok = idx < 20u && ((0x80045u >> idx) & 1u);
  uint32_t x = (b >> 1) & 3u;  // A:0 C:1 T:2 G:3
  code = x ^ (x >> 1);         // A:0 C:1 G:2 T:3
  return ok;
}

template <int K>
struct Roller {
  static_assert(K >= 1 && K <= 64, "k out of range");
  static constexpr int ND = (K + 3) / 4;       // dwords holding K ASCII bytes
  static constexpr int NB = K - 4 * (ND - 1);  // valid bytes in the last dword, 1..4
  static constexpr int NW = (K + 31) / 32;     // 64-bit words of the 2-bit form
  static constexpr uint32_t LAST_MASK = NB == 4 ? 0xffffffffu : ((1u << (8 * NB)) - 1u);

  uint32_t f[ND];  // forward strand, ASCII, byte j of the string at bits 8*(j%4) of f[j/4]
  uint32_t r[ND];  // reverse complement, ASCII
  uint64_t pf_lo, pf_hi, pr_lo, pr_hi;  // 2-bit packed strands (hi unused when K <= 32)
  int run;                              // consecutive valid bases seen

  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int j = 0; j < ND; ++j) { f[j] = 0; r[j] = 0; }
    pf_lo = pf_hi = pr_lo = pr_hi = 0;
    run = 0;
  }

  // Append one base with code c (0..3; anything else only in bits that `run` keeps from being used).
  // ASCII of the base and of its complement come from one v_perm_b32 each (byte select out of "ACGT" / "TGCA").
  __device__ __forceinline__ void push(uint32_t c) {
    const uint32_t sel = c | 0x0c0c0c00u;  // byte 0 <- table[c]; bytes 1..3 <- 0x00
    const uint32_t up = __builtin_amdgcn_perm(0u, 0x54474341u, sel);  // "ACGT"[c]
    const uint32_t cu = __builtin_amdgcn_perm(0u, 0x41434754u, sel);  // complement: "TGCA"[c]
    c &= 3u;
    // forward ASCII window: drop byte 0, append `up` as byte K-1
#pragma unroll
    for (int j = 0; j + 1 < ND; ++j) f[j] = __builtin_amdgcn_alignbyte(f[j + 1], f[j], 1);
    if constexpr (NB == 1) f[ND - 1] = up;  // the last dword holds one byte: nothing to keep of it
    else f[ND - 1] = (f[ND - 1] >> 8) | (up << (8 * (NB - 1)));
    // reverse-complement ASCII window: prepend `cu` as byte 0, drop byte K-1
#pragma unroll
    for (int j = ND - 1; j >= 1; --j) r[j] = __builtin_amdgcn_alignbyte(r[j], r[j - 1], 3);
    r[0] = (r[0] << 8) | cu;
    r[ND - 1] &= LAST_MASK;
    // 2-bit packed strands
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

  // MurmurHash3_x64_128(canonical ASCII k-mer, seed 0) -> first 64 bits.
  __device__ __forceinline__ uint64_t hash() const {
    const bool fw = forward_is_canonical();
    uint32_t w[ND + 4];
    if constexpr (ND <= 8) {
#pragma unroll
      for (int j = 0; j < ND; ++j) w[j] = fw ? f[j] : r[j];
    } else {
      // For longer windows the optimiser turns the element-wise select into "select the ARRAY, then load",
      // which forces f[] and r[] into scratch memory; a bit-field blend (v_bfi_b32) keeps them in registers.
      const uint32_t m = fw ? 0xffffffffu : 0u;
#pragma unroll
      for (int j = 0; j < ND; ++j) w[j] = (f[j] & m) | (r[j] & ~m);
    }
#pragma unroll
    for (int j = ND; j < ND + 4; ++j) w[j] = 0;
    constexpr uint64_t C1 = 0x87c37b91114253d5ULL, C2 = 0x4cf5ad432745937fULL;
    constexpr int NBLK = K / 16, TAIL = K & 15;
    uint64_t h1 = 0, h2 = 0;
    // 16-byte body blocks, expanded with compile-time indices (a `for` over NBLK >= 2 is unrolled too late
    // for the register promotion of w[], f[] and r[]: they would live in scratch memory)
    auto body = [&]<int B>() {
      uint64_t k1 = (uint64_t)w[4 * B] | ((uint64_t)w[4 * B + 1] << 32);
      uint64_t k2 = (uint64_t)w[4 * B + 2] | ((uint64_t)w[4 * B + 3] << 32);
      k1 *= C1; k1 = rotl64(k1, 31); k1 *= C2; h1 ^= k1;
      h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ULL;
      k2 *= C2; k2 = rotl64(k2, 33); k2 *= C1; h2 ^= k2;
      h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ULL;
    };
    [&]<int... B>(std::integer_sequence<int, B...>) { (body.template operator()<B>(), ...); }(
        std::make_integer_sequence<int, NBLK>{});
    if constexpr (TAIL > 8) {
      uint64_t k2 = (uint64_t)w[4 * NBLK + 2] | ((uint64_t)w[4 * NBLK + 3] << 32);
      k2 *= C2; k2 = rotl64(k2, 33); k2 *= C1; h2 ^= k2;
    }
    if constexpr (TAIL > 0) {
      uint64_t k1 = (uint64_t)w[4 * NBLK] | ((uint64_t)w[4 * NBLK + 1] << 32);
      k1 *= C1; k1 = rotl64(k1, 31); k1 *= C2; h1 ^= k1;
    }
    h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
  }
};

// MurmurHash3_x64_128(the K bytes held in w[0 .. (K+3)/4), zero padded up to NDW dwords; seed 0) -> first 64 bits.
template <int K, int NDW>
__device__ __forceinline__ uint64_t murmur3_h1_words(const uint32_t (&w)[NDW]) {
  static_assert(NDW >= (K + 3) / 4 + 4, "the tail reads up to four dwords past the last byte");
  constexpr uint64_t C1 = 0x87c37b91114253d5ULL, C2 = 0x4cf5ad432745937fULL;
  constexpr int NBLK = K / 16, TAIL = K & 15;
  uint64_t h1 = 0, h2 = 0;
  auto body = [&]<int B>() {
    uint64_t k1 = (uint64_t)w[4 * B] | ((uint64_t)w[4 * B + 1] << 32);
    uint64_t k2 = (uint64_t)w[4 * B + 2] | ((uint64_t)w[4 * B + 3] << 32);
    k1 *= C1; k1 = rotl64(k1, 31); k1 *= C2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ULL;
    k2 *= C2; k2 = rotl64(k2, 33); k2 *= C1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ULL;
  };
  [&]<int... B>(std::integer_sequence<int, B...>) { (body.template operator()<B>(), ...); }(
      std::make_integer_sequence<int, NBLK>{});
  if constexpr (TAIL > 8) {
    uint64_t k2 = (uint64_t)w[4 * NBLK + 2] | ((uint64_t)w[4 * NBLK + 3] << 32);
    k2 *= C2; k2 = rotl64(k2, 33); k2 *= C1; h2 ^= k2;
  }
  if constexpr (TAIL > 0) {
    uint64_t k1 = (uint64_t)w[4 * NBLK] | ((uint64_t)w[4 * NBLK + 1] << 32);
    k1 *= C1; k1 = rotl64(k1, 31); k1 *= C2; h1 ^= k1;
  }
  h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2;
  return h1;
}

// Hash of the canonical K-mer that ENDS at the newest base of a Roller<KMAX>, K <= KMAX: one roller at the largest
// k of a multi-k query serves every smaller k.  The K-mer is the SUFFIX of the forward window (bytes KMAX-K ..
// KMAX-1: a byte-granular funnel shift of the forward ASCII words; the low 2K bits of the forward 2-bit form) and
// its reverse complement is the PREFIX of the reverse-complement window (its first K bytes as they are; the top 2K
// bits of the reverse 2-bit form).
template <int K, int KMAX>
__device__ __forceinline__ uint64_t hash_suffix(const Roller<KMAX>& R) {
  static_assert(K >= 1 && K <= KMAX, "sub-k must not exceed the roller's k");
  if constexpr (K == KMAX) {
    return R.hash();
  } else {
    constexpr int ND = (K + 3) / 4, NB = K - 4 * (ND - 1);
    constexpr uint32_t LAST = NB == 4 ? 0xffffffffu : ((1u << (8 * NB)) - 1u);
    constexpr int OFF = KMAX - K, OD = OFF / 4, OB = OFF % 4;
    constexpr int NDM = Roller<KMAX>::ND;
    // canonical choice on the 2-bit forms (as 128-bit values hi:lo)
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
    const bool fw = K <= 32 ? (f_lo <= r_lo) : (f_hi < r_hi || (f_hi == r_hi && f_lo <= r_lo));
    const uint32_t m = fw ? 0xffffffffu : 0u;
    uint32_t w[ND + 4];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const uint32_t a = R.f[OD + j];
      const uint32_t b = (OD + j + 1 < NDM) ? R.f[OD + j + 1] : 0u;
      uint32_t fwd = OB == 0 ? a : __builtin_amdgcn_alignbyte(b, a, OB);
      uint32_t rev = R.r[j];
      if (j == ND - 1) { fwd &= LAST; rev &= LAST; }
      w[j] = (fwd & m) | (rev & ~m);
    }
#pragma unroll
    for (int j = ND; j < ND + 4; ++j) w[j] = 0;
    return murmur3_h1_words<K, ND + 4>(w);
  }
}

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
