/*
 * mg_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain, sequential C restatement of the algorithms on Metalign's hot path,
 * used only as the checker in tests/, by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py.  The product path (metalign_amd/) never loads
 * this file; it fails loudly when libmetalign_hip.so is missing.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   Stage C  (mgo_profile_assign)    PINNED — follows scripts/map_and_profile.py
 *            line by line and is checked against golden vectors produced by
 *            importing that file in the build container
 *            (tests/golden/make_golden.py).
 *   MurmurHash3_x64_128              PINNED — public known-answer vectors
 *            (tests/golden/murmur3_kat.json).
 *   Stage A/B (sketch / containment) PARITY UNPINNED — the arithmetic lives in
 *            KMC 3 and CMash (scripts/select_db.py:50-59,73-76), neither
 *            vendored nor version-pinned by the reference and absent from this
 *            image.  This file is the normative statement of what the build
 *            computes for those stages; the GPU must match it bit for bit.
 *            That includes the "reference pipeline" below (mgo_refpipe_*): the
 *            reference's own wiring of those tools (k_max-mers only on the read
 *            side, smaller-k columns from prefixes of the matched k_max-mers)
 *            and mgo_refpipe_count_kmers (round 6): the read side of that wiring
 *            by k-mer IDENTITY — canonical k_max-mers of the reads counted among
 *            the table's, no hash — which is what kmc + kmc_tools intersect
 *            compute and what the product's default stage A (mg_kcount.hip) is
 *            held to.  tools/verify_against_cmash.sh is how to pin all of this
 *            on a machine that has the two tools.
 *
 * All paths below are relative to /root/reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/metalign_hip.h"

/* ---------------------------------------------------------------------- *
 * MurmurHash3_x64_128 (Austin Appleby, public domain algorithm), restated.
 * CMash hashes k-mers with the first 64 bits of this function, seed 0
 * [UPSTREAM-RECOLLECTION, SURVEY.md §8c].
 * ---------------------------------------------------------------------- */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t v) {
  v ^= v >> 33;
  v *= 0xff51afd7ed558ccdULL;
  v ^= v >> 33;
  v *= 0xc4ceb9fe1a85ec53ULL;
  v ^= v >> 33;
  return v;
}

static inline uint64_t load_le64(const uint8_t* p, int nbytes) {
  uint64_t v = 0;
  for (int i = 0; i < nbytes; ++i) v |= (uint64_t)p[i] << (8 * i);
  return v;
}

void mgo_murmur3_x64_128(const void* key, int len, uint32_t seed, uint64_t out[2]) {
  const uint64_t C1 = 0x87c37b91114253d5ULL, C2 = 0x4cf5ad432745937fULL;
  const uint8_t* p = (const uint8_t*)key;
  uint64_t h1 = seed, h2 = seed;
  int done = 0;
  while (len - done >= 16) {
    uint64_t k1 = load_le64(p + done, 8), k2 = load_le64(p + done + 8, 8);
    k1 *= C1; k1 = rotl64(k1, 31); k1 *= C2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ULL;
    k2 *= C2; k2 = rotl64(k2, 33); k2 *= C1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ULL;
    done += 16;
  }
  int rem = len - done;
  if (rem > 8) {
    uint64_t k2 = load_le64(p + done + 8, rem - 8);
    k2 *= C2; k2 = rotl64(k2, 33); k2 *= C1; h2 ^= k2;
  }
  if (rem > 0) {
    uint64_t k1 = load_le64(p + done, rem > 8 ? 8 : rem);
    k1 *= C1; k1 = rotl64(k1, 31); k1 *= C2; h1 ^= k1;
  }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2; h2 += h1;
  out[0] = h1; out[1] = h2;
}

/* ---------------------------------------------------------------------- *
 * k-mer enumeration.  KMC counts CANONICAL k-mers (lexicographic min of the
 * k-mer and its reverse complement; scripts/select_db.py:50-52 runs it with
 * default canonical mode) over windows free of non-ACGT symbols.
 * ---------------------------------------------------------------------- */
static inline int base_code(uint8_t b) { /* A,C,G,T (either case) -> 0..3, else -1 */
  switch (b) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

static const char kUpper[4] = {'A', 'C', 'G', 'T'};

/* Which definition of a k-mer's hash is in force (the library's mg_set_hash_mode):
 *   0  hash(min(kmer, revcomp)): MurmurHash3 of the lexicographically smaller strand — KMC's canonical k-mer, ONE hash,
 *      the full 64 bits.  The default.
 *   1  min(hash(kmer), hash(revcomp)) mod p, p = 9999999999971: CMash's MinHash.CountEstimator AS SURVEY.md §8(c) RECOLLECTS
 *      it (khmer's hash_no_rc_murmur3 on both strands, the smaller value kept, `% p` with p = get_prime_lt_x(9999999999971.)
 *      = 9999999999971, itself prime) — UNVERIFIED: CMash's source is not under /root/reference, the reference pins no
 *      version and holds no vectors at this seam (scripts/select_db.py:69-76; local_tests/dump_kmers.py:2-7).  It exists
 *      so that a table built by this definition and CMash's own could be compared at all. */
static int g_hash_mode = 0;
#define MGO_CMASH_PRIME 9999999999971ULL
void mgo_set_hash_mode(int mode) { g_hash_mode = mode ? 1 : 0; }
int mgo_hash_mode(void) { return g_hash_mode; }

/* Hash of the k-mer seq[0..k) (all symbols valid) under the mode in force. */
static uint64_t canonical_hash(const uint8_t* seq, int k) {
  char fwd[MG_MAX_K], rc[MG_MAX_K];
  for (int i = 0; i < k; ++i) {
    int c = base_code(seq[i]);
    fwd[i] = kUpper[c];
    rc[k - 1 - i] = kUpper[3 - c];
  }
  uint64_t out[2];
  if (g_hash_mode == 1) {
    uint64_t a, b;
    mgo_murmur3_x64_128(fwd, k, 0, out);
    a = out[0];
    mgo_murmur3_x64_128(rc, k, 0, out);
    b = out[0];
    return (a < b ? a : b) % MGO_CMASH_PRIME;
  }
  const char* pick = memcmp(fwd, rc, (size_t)k) <= 0 ? fwd : rc;
  mgo_murmur3_x64_128(pick, k, 0, out);
  return out[0];
}

/* Per-position canonical hash of one sequence: out[i] for window starting at i,
 * UINT64_MAX-with-flag semantics avoided: valid[i] says whether the window is a
 * k-mer.  Returns number of valid windows.  (Helper for tests.) */
uint64_t mgo_kmer_hashes(const uint8_t* seq, uint64_t len, int k, uint64_t* out,
                         uint8_t* valid) {
  uint64_t nvalid = 0;
  if (len < (uint64_t)k) return 0;
  uint64_t run = 0; /* consecutive valid symbols ending at position j */
  for (uint64_t j = 0; j < len; ++j) {
    run = base_code(seq[j]) >= 0 ? run + 1 : 0;
    if (j + 1 >= (uint64_t)k) {
      uint64_t i = j + 1 - (uint64_t)k;
      if (run >= (uint64_t)k) {
        out[i] = canonical_hash(seq + i, k);
        valid[i] = 1;
        ++nvalid;
      } else {
        out[i] = 0;
        valid[i] = 0;
      }
    }
  }
  return nvalid;
}

static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

typedef struct { uint64_t* v; uint64_t n, cap; } u64vec;

static int vec_push(u64vec* a, uint64_t x) {
  if (a->n == a->cap) {
    uint64_t nc = a->cap ? a->cap * 2 : 1024;
    uint64_t* nv = (uint64_t*)realloc(a->v, nc * sizeof(uint64_t));
    if (!nv) return -1;
    a->v = nv; a->cap = nc;
  }
  a->v[a->n++] = x;
  return 0;
}

static int collect_hashes(const uint8_t* seq, uint64_t len, int k, uint64_t hmax,
                          u64vec* acc, uint64_t* kmers_seen) {
  if (len < (uint64_t)k) return 0;
  uint64_t run = 0;
  for (uint64_t j = 0; j < len; ++j) {
    run = base_code(seq[j]) >= 0 ? run + 1 : 0;
    if (run >= (uint64_t)k) {
      uint64_t h = canonical_hash(seq + (j + 1 - (uint64_t)k), k);
      ++*kmers_seen;
      /* 2^64-1 is reserved (the GPU uses it as the "no k-mer here" marker): never a sketch member */
      if (h <= hmax && h != UINT64_MAX && vec_push(acc, h)) return -1;
    }
  }
  return 0;
}

/* Stage A.  See include/metalign_hip.h (mg_sketch_reads) for the definition.
 * cs: occurrence counters saturate at cs (kmc -cs3, scripts/select_db.py:50); 0 = exact counts. */
int mgo_sketch_reads(const uint8_t* bases, const uint64_t* offsets, uint64_t nreads,
                     int k, uint64_t hmax, uint64_t s, uint32_t cs, uint64_t* out_hashes,
                     uint32_t* out_counts, uint64_t out_cap, uint64_t* out_n,
                     int* out_truncated, uint64_t* out_kmers_seen) {
  if (k < 1 || k > MG_MAX_K) return MG_ERR_ARG;
  u64vec acc = {0, 0, 0};
  uint64_t seen = 0;
  for (uint64_t r = 0; r < nreads; ++r)
    if (collect_hashes(bases + offsets[r], offsets[r + 1] - offsets[r], k, hmax, &acc, &seen)) {
      free(acc.v);
      return MG_ERR_NOMEM;
    }
  if (acc.n) qsort(acc.v, acc.n, sizeof(uint64_t), cmp_u64);
  uint64_t n = 0;
  int truncated = 0, rc = MG_OK;
  for (uint64_t i = 0; i < acc.n;) {
    uint64_t j = i;
    while (j < acc.n && acc.v[j] == acc.v[i]) ++j;
    if (s > 0 && n == s) { truncated = 1; break; }
    if (n == out_cap) { rc = MG_ERR_CAPACITY; break; }
    uint64_t c = j - i;
    if (cs && c > cs) c = cs;
    out_hashes[n] = acc.v[i];
    out_counts[n] = c > 0xffffffffULL ? 0xffffffffu : (uint32_t)c;
    ++n;
    i = j;
  }
  free(acc.v);
  *out_n = n;
  if (out_truncated) *out_truncated = truncated;
  if (out_kmers_seen) *out_kmers_seen = seen;
  return rc;
}

/* Stage A'.  Per-genome bottom-n distinct hashes, ascending. */
int mgo_sketch_genomes(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes,
                       int k, uint64_t n, uint64_t* out_hashes, uint64_t* out_offsets) {
  if (k < 1 || k > MG_MAX_K) return MG_ERR_ARG;
  uint64_t w = 0;
  out_offsets[0] = 0;
  for (uint64_t g = 0; g < ngenomes; ++g) {
    u64vec acc = {0, 0, 0};
    uint64_t seen = 0;
    if (collect_hashes(bases + offsets[g], offsets[g + 1] - offsets[g], k, UINT64_MAX, &acc, &seen)) {
      free(acc.v);
      return MG_ERR_NOMEM;
    }
    if (acc.n) qsort(acc.v, acc.n, sizeof(uint64_t), cmp_u64);
    uint64_t kept = 0;
    for (uint64_t i = 0; i < acc.n && kept < n; ++i)
      if (i == 0 || acc.v[i] != acc.v[i - 1]) { out_hashes[w++] = acc.v[i]; ++kept; }
    free(acc.v);
    out_offsets[g + 1] = w;
  }
  return MG_OK;
}

/* The k < k_max tables of hash mode 1 (CMash as SURVEY.md §8(c) recollects it: the smaller-k columns are containments of
 * the k-PREFIXES of the sketched k_max-mers, stored in a ternary search tree; UNVERIFIED like the rest of the mode).  Per
 * genome: the bottom-n sketch at k_max under mode 1; every sketched k_max-mer oriented as CMash's CountEstimator.add keeps
 * it (the strand whose hash is the smaller one; the reverse complement on a tie); its first k bases; the key of that
 * k-prefix = its own mode-1 hash (symmetric in the strand, so a read k-mer matches the prefix or its reverse complement, as
 * the streaming query does); the genome's table entry for k = its DISTINCT keys, ascending.  Two different k_max-mers with
 * the same hash value (a collision modulo the prime) share one sketch slot: the smaller key is kept.  The mode in force
 * does not matter to this function (it is mode 1 by definition). */
typedef struct { uint64_t h, key; } hk_pair;
static int cmp_hk(const void* a, const void* b) {
  const hk_pair* x = (const hk_pair*)a; const hk_pair* y = (const hk_pair*)b;
  if (x->h != y->h) return x->h < y->h ? -1 : 1;
  return x->key < y->key ? -1 : (x->key > y->key ? 1 : 0);
}
int mgo_sketch_genomes_prefix(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int kmax, int k, uint64_t n,
                              uint64_t* out_hashes, uint64_t* out_offsets) {
  if (k < 1 || k > kmax || kmax > MG_MAX_K) return MG_ERR_ARG;
  const int saved = g_hash_mode;
  uint64_t w = 0;
  out_offsets[0] = 0;
  for (uint64_t g = 0; g < ngenomes; ++g) {
    const uint8_t* seq = bases + offsets[g];
    const uint64_t len = offsets[g + 1] - offsets[g];
    hk_pair* v = (hk_pair*)malloc((len + 1) * sizeof(hk_pair));
    if (!v) return MG_ERR_NOMEM;
    uint64_t nv = 0, run = 0;
    for (uint64_t j = 0; j < len; ++j) {
      run = base_code(seq[j]) >= 0 ? run + 1 : 0;
      if (run < (uint64_t)kmax) continue;
      const uint8_t* win = seq + (j + 1 - (uint64_t)kmax);
      char fwd[MG_MAX_K], rc[MG_MAX_K];
      for (int i = 0; i < kmax; ++i) {
        int c = base_code(win[i]);
        fwd[i] = kUpper[c];
        rc[kmax - 1 - i] = kUpper[3 - c];
      }
      uint64_t o[2], hf, hr;
      mgo_murmur3_x64_128(fwd, kmax, 0, o); hf = o[0];
      mgo_murmur3_x64_128(rc, kmax, 0, o); hr = o[0];
      const char* kept = hr <= hf ? rc : fwd;  /* CountEstimator.add: `if h == h2: kmer = rc` */
      g_hash_mode = 1;
      v[nv].h = (hf < hr ? hf : hr) % MGO_CMASH_PRIME;
      v[nv].key = canonical_hash((const uint8_t*)kept, k);  /* the k-prefix of the kept strand, hashed under mode 1 */
      g_hash_mode = saved;
      ++nv;
    }
    if (nv) qsort(v, nv, sizeof(hk_pair), cmp_hk);
    uint64_t* keys = (uint64_t*)malloc((n + 1) * sizeof(uint64_t));
    if (!keys) { free(v); return MG_ERR_NOMEM; }
    uint64_t kept_n = 0;
    for (uint64_t i = 0; i < nv && kept_n < n; ++i)
      if (i == 0 || v[i].h != v[i - 1].h) keys[kept_n++] = v[i].key;  /* first of a run of equal h: the smallest key */
    free(v);
    if (kept_n) qsort(keys, kept_n, sizeof(uint64_t), cmp_u64);
    for (uint64_t i = 0; i < kept_n; ++i)
      if (i == 0 || keys[i] != keys[i - 1]) out_hashes[w++] = keys[i];
    free(keys);
    out_offsets[g + 1] = w;
  }
  return MG_OK;
}

/* ---------------------------------------------------------------------- *
 * THE REFERENCE'S OWN WIRING OF STAGE A/B ("reference pipeline").
 *
 * scripts/select_db.py counts ONLY k_max-mers of the reads (`kmc -k60 -ci2 -cs3`, :50-52), intersects them with the k_max-mers
 * of all genome sketches (`kmc_tools simple ... intersect`, :54-56), dumps the survivors as FASTA (:58-65) and hands THAT to
 * CMash's streaming query with the k range 30-60-10 (:73-76).  The query [UPSTREAM-RECOLLECTION, SURVEY.md §8c: CMash is not
 * under /root/reference] walks every k_max-mer x of its input and the reverse complement of x, and for every k of the range
 * looks the k-PREFIX up in a prefix tree of the sketched k_max-mers (each stored in the orientation CountEstimator.add kept);
 * a genome's column for k is (distinct k-prefixes of its sketched k_max-mers that were found) / (distinct k-prefixes of its
 * sketched k_max-mers).  So the read side is ONE k, and every smaller-k column is a function of WHICH SKETCHED k_max-MERS
 * MATCHED — derived here on the table side:
 *
 *   table (mgo_sketch_genomes_kmers + mgo_refpipe_build_k)
 *     entry  = one sketched k_max-mer of one genome: its hash under the mode in force (the identity the read side matches
 *              on) and the k-mer as the table keeps it, 2-bit packed (first base most significant): mode 0 the
 *              lexicographically smaller strand, mode 1 the strand with the smaller MurmurHash3 (the reverse complement on a
 *              tie) — of the FIRST window of the genome that has the hash (CountEstimator.add keeps the first k-mer of a hash)
 *     pairs  = all entries sorted by (hash, genome): the hash-major table of k_max as everywhere else in this build
 *     for k < k_max:  D_k = the distinct k-prefixes of all entries' kept k-mers, numbered in ascending (lexicographic) order
 *              pa[pair] = number of the kept k-mer's k-prefix;  pb[pair] = number of its REVERSE COMPLEMENT's k-prefix when
 *              that string is in D_k, else 0xffffffff (the query tries both strands, :73-76)
 *              count list = the distinct (prefix number, genome) combinations, ascending; gsize_k[g] = how many genome g has
 *   query (mgo_refpipe_matched + mgo_refpipe_hits_k)
 *     matched[pair] = the pair's hash is in the read sketch of k_max with count >= ci                      (:50-56)
 *     k_max column  = matched pairs of the genome / sketch size                         (what mgo_containment computes)
 *     marked_k      = { pa[pair], pb[pair] : matched[pair] }
 *     k column      = #{(p, g) in the count list : p in marked_k} / gsize_k[g]                              (:73-76)
 *
 * PARITY UNPINNED like the rest of stage A/B; tests/indep_sketch.py states the same on STRINGS (dicts of k-mers and
 * prefixes, no hash on the query side at all) and tests/test_oracle_independent.py holds the two together.
 * ---------------------------------------------------------------------- */

/* codes c[0..k) (0..3 = ACGT) -> the 2k-bit number with the first base most significant, as (hi, lo) */
static void pack_codes(const uint8_t* c, int k, uint64_t* hi, uint64_t* lo) {
  uint64_t h = 0, l = 0;
  for (int i = 0; i < k; ++i) {
    h = (h << 2) | (l >> 62);
    l = (l << 2) | (uint64_t)(c[i] & 3);
  }
  *hi = h; *lo = l;
}

static void unpack_codes(uint64_t hi, uint64_t lo, int k, uint8_t* c) {
  for (int i = k - 1; i >= 0; --i) {
    c[i] = (uint8_t)(lo & 3);
    lo = (lo >> 2) | (hi << 62);
    hi >>= 2;
  }
}

typedef struct { uint64_t h, pos; int use_rc; } hp_entry;
static int cmp_hp(const void* a, const void* b) {
  const hp_entry* x = (const hp_entry*)a; const hp_entry* y = (const hp_entry*)b;
  if (x->h != y->h) return x->h < y->h ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}

/* Stage A' with the k-mers kept (CMash's database holds _mins AND _kmers per genome: local_tests/dump_kmers.py:7-14).
 * As mgo_sketch_genomes under the mode in force; out_khi/out_klo[i] = the kept k-mer of out_hashes[i], packed. */
int mgo_sketch_genomes_kmers(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int k, uint64_t n,
                             uint64_t* out_hashes, uint64_t* out_khi, uint64_t* out_klo, uint64_t* out_offsets) {
  if (k < 1 || k > MG_MAX_K) return MG_ERR_ARG;
  uint64_t w = 0;
  out_offsets[0] = 0;
  for (uint64_t g = 0; g < ngenomes; ++g) {
    const uint8_t* seq = bases + offsets[g];
    const uint64_t len = offsets[g + 1] - offsets[g];
    hp_entry* v = (hp_entry*)malloc((len + 1) * sizeof(hp_entry));
    if (!v) return MG_ERR_NOMEM;
    uint64_t nv = 0, run = 0;
    for (uint64_t j = 0; j < len; ++j) {
      run = base_code(seq[j]) >= 0 ? run + 1 : 0;
      if (run < (uint64_t)k) continue;
      const uint8_t* win = seq + (j + 1 - (uint64_t)k);
      char fwd[MG_MAX_K], rc[MG_MAX_K];
      for (int i = 0; i < k; ++i) {
        int c = base_code(win[i]);
        fwd[i] = kUpper[c];
        rc[k - 1 - i] = kUpper[3 - c];
      }
      uint64_t o[2];
      if (g_hash_mode == 1) {
        uint64_t hf, hr;
        mgo_murmur3_x64_128(fwd, k, 0, o); hf = o[0];
        mgo_murmur3_x64_128(rc, k, 0, o); hr = o[0];
        v[nv].h = (hf < hr ? hf : hr) % MGO_CMASH_PRIME;
        v[nv].use_rc = hr <= hf;  /* CountEstimator.add: the strand with the smaller hash, the reverse complement on a tie */
      } else {
        v[nv].use_rc = memcmp(fwd, rc, (size_t)k) > 0;  /* KMC's canonical k-mer */
        mgo_murmur3_x64_128(v[nv].use_rc ? rc : fwd, k, 0, o);
        v[nv].h = o[0];
      }
      if (v[nv].h == UINT64_MAX) continue;  /* reserved: never a sketch member */
      v[nv].pos = j + 1 - (uint64_t)k;
      ++nv;
    }
    if (nv) qsort(v, nv, sizeof(hp_entry), cmp_hp);
    uint64_t kept = 0;
    for (uint64_t i = 0; i < nv && kept < n; ++i) {
      if (i > 0 && v[i].h == v[i - 1].h) continue;  /* a hash seen before: its first window stays */
      uint8_t codes[MG_MAX_K];
      const uint8_t* win = seq + v[i].pos;
      for (int t = 0; t < k; ++t) {
        int c = base_code(win[t]);
        if (v[i].use_rc) codes[k - 1 - t] = (uint8_t)(3 - c); else codes[t] = (uint8_t)c;
      }
      out_hashes[w] = v[i].h;
      pack_codes(codes, k, &out_khi[w], &out_klo[w]);
      ++w; ++kept;
    }
    free(v);
    out_offsets[g + 1] = w;
  }
  return MG_OK;
}

/* `build_db --sketch_hash forward`: CMash's TRAINING without reverse complements, as recollected (nothing under /root/reference
 * shows it: parity unpinned like the rest of stage A/B).  A genome's entries: its k-mers with the n smallest distinct values of
 * MurmurHash3(k-mer as it stands, seed 0).h1 % 9999999999971, the first window of a value, kept as they stand.  out_hashes[i] =
 * what entry i MATCHES by: the hash of its k-mer under the mode in force (as mgo_sketch_genomes_kmers computes it) — in the order of the
 * selecting values, so neither ascending nor necessarily distinct within a genome. */
int mgo_sketch_genomes_kmers_forward(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int k, uint64_t n,
                                     uint64_t* out_hashes, uint64_t* out_khi, uint64_t* out_klo, uint64_t* out_offsets) {
  if (k < 1 || k > MG_MAX_K) return MG_ERR_ARG;
  uint64_t w = 0;
  out_offsets[0] = 0;
  for (uint64_t g = 0; g < ngenomes; ++g) {
    const uint8_t* seq = bases + offsets[g];
    const uint64_t len = offsets[g + 1] - offsets[g];
    hp_entry* v = (hp_entry*)malloc((len + 1) * sizeof(hp_entry));
    if (!v) return MG_ERR_NOMEM;
    uint64_t nv = 0, run = 0;
    for (uint64_t j = 0; j < len; ++j) {
      run = base_code(seq[j]) >= 0 ? run + 1 : 0;
      if (run < (uint64_t)k) continue;
      const uint8_t* win = seq + (j + 1 - (uint64_t)k);
      char fwd[MG_MAX_K];
      for (int i = 0; i < k; ++i) fwd[i] = kUpper[base_code(win[i])];
      uint64_t o[2];
      mgo_murmur3_x64_128(fwd, k, 0, o);
      v[nv].h = o[0] % MGO_CMASH_PRIME;
      v[nv].use_rc = 0;
      v[nv].pos = j + 1 - (uint64_t)k;
      ++nv;
    }
    if (nv) qsort(v, nv, sizeof(hp_entry), cmp_hp);
    uint64_t kept = 0;
    for (uint64_t i = 0; i < nv && kept < n; ++i) {
      if (i > 0 && v[i].h == v[i - 1].h) continue;
      const uint8_t* win = seq + v[i].pos;
      uint8_t codes[MG_MAX_K];
      char fwd[MG_MAX_K], rc[MG_MAX_K];
      for (int t = 0; t < k; ++t) {
        int c = base_code(win[t]);
        codes[t] = (uint8_t)c;
        fwd[t] = kUpper[c];
        rc[k - 1 - t] = kUpper[3 - c];
      }
      uint64_t o[2], id;
      if (g_hash_mode == 1) {
        uint64_t hf, hr;
        mgo_murmur3_x64_128(fwd, k, 0, o); hf = o[0];
        mgo_murmur3_x64_128(rc, k, 0, o); hr = o[0];
        id = (hf < hr ? hf : hr) % MGO_CMASH_PRIME;
      } else {
        mgo_murmur3_x64_128(memcmp(fwd, rc, (size_t)k) > 0 ? rc : fwd, k, 0, o);
        id = o[0];
      }
      out_hashes[w] = id;
      pack_codes(codes, k, &out_khi[w], &out_klo[w]);
      ++w; ++kept;
    }
    free(v);
    out_offsets[g + 1] = w;
  }
  return MG_OK;
}

typedef struct { uint64_t h; uint32_t g; uint64_t e; } pair_ent;   /* e: index of the entry in the genome-major arrays */
static int cmp_pair_ent(const void* a, const void* b) {
  const pair_ent* x = (const pair_ent*)a; const pair_ent* y = (const pair_ent*)b;
  if (x->h != y->h) return x->h < y->h ? -1 : 1;
  if (x->g != y->g) return x->g < y->g ? -1 : 1;
  return x->e < y->e ? -1 : (x->e > y->e ? 1 : 0);
}
typedef struct { uint64_t hi, lo; } u128;
static int cmp_u128(const void* a, const void* b) {
  const u128* x = (const u128*)a; const u128* y = (const u128*)b;
  if (x->hi != y->hi) return x->hi < y->hi ? -1 : 1;
  return x->lo < y->lo ? -1 : (x->lo > y->lo ? 1 : 0);
}

/* The hash-major pairs of the k_max table: entries sorted by (hash, genome).  perm[i] = genome-major index of pair i. */
int mgo_refpipe_pairs(const uint64_t* hashes, const uint64_t* offsets, uint64_t ngenomes, uint64_t* pair_hash,
                      uint32_t* pair_gen, uint64_t* perm) {
  const uint64_t E = offsets[ngenomes];
  pair_ent* p = (pair_ent*)malloc((E + 1) * sizeof(pair_ent));
  if (!p) return MG_ERR_NOMEM;
  for (uint64_t g = 0; g < ngenomes; ++g)
    for (uint64_t e = offsets[g]; e < offsets[g + 1]; ++e) { p[e].h = hashes[e]; p[e].g = (uint32_t)g; p[e].e = e; }
  if (E) qsort(p, E, sizeof(pair_ent), cmp_pair_ent);
  for (uint64_t i = 0; i < E; ++i) { pair_hash[i] = p[i].h; pair_gen[i] = p[i].g; perm[i] = p[i].e; }
  free(p);
  return MG_OK;
}

/* One k < kmax of the table, from the entries in PAIR order (khi/klo/gen[npairs]: the kept kmax-mers and genomes of the
 * pairs).  pa, pb: [npairs]; cid, cgen: [npairs] capacity, *ncount filled; gsize: [ngenomes]; *nprefix = |D_k|. */
int mgo_refpipe_build_k(const uint64_t* khi, const uint64_t* klo, const uint32_t* gen, uint64_t npairs, uint64_t ngenomes,
                        int kmax, int k, uint32_t* pa, uint32_t* pb, uint32_t* cid, uint32_t* cgen, uint64_t* ncount,
                        uint32_t* gsize, uint64_t* nprefix) {
  if (k < 1 || k >= kmax || kmax > MG_MAX_K) return MG_ERR_ARG;
  u128* a = (u128*)malloc((npairs + 1) * sizeof(u128));  /* prefix of the kept strand */
  u128* b = (u128*)malloc((npairs + 1) * sizeof(u128));  /* prefix of the other strand */
  u128* d = (u128*)malloc((npairs + 1) * sizeof(u128));
  uint64_t* cg = (uint64_t*)malloc((npairs + 1) * sizeof(uint64_t));
  if (!a || !b || !d || !cg) { free(a); free(b); free(d); free(cg); return MG_ERR_NOMEM; }
  for (uint64_t i = 0; i < npairs; ++i) {
    uint8_t c[MG_MAX_K], r[MG_MAX_K];
    unpack_codes(khi[i], klo[i], kmax, c);
    for (int t = 0; t < kmax; ++t) r[kmax - 1 - t] = (uint8_t)(3 - c[t]);
    pack_codes(c, k, &a[i].hi, &a[i].lo);
    pack_codes(r, k, &b[i].hi, &b[i].lo);
    d[i] = a[i];
  }
  if (npairs) qsort(d, npairs, sizeof(u128), cmp_u128);
  uint64_t nd = 0;
  for (uint64_t i = 0; i < npairs; ++i)
    if (i == 0 || cmp_u128(&d[i], &d[i - 1]) != 0) d[nd++] = d[i];
  *nprefix = nd;
  for (uint64_t i = 0; i < npairs; ++i) {
    for (int which = 0; which < 2; ++which) {
      const u128* key = which ? &b[i] : &a[i];
      uint64_t lo = 0, hi = nd;
      while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (cmp_u128(&d[mid], key) < 0) lo = mid + 1; else hi = mid;
      }
      const int found = lo < nd && cmp_u128(&d[lo], key) == 0;
      if (which) pb[i] = found ? (uint32_t)lo : 0xffffffffu; else pa[i] = (uint32_t)lo;
    }
    cg[i] = ((uint64_t)pa[i] << 32) | gen[i];
  }
  if (npairs) qsort(cg, npairs, sizeof(uint64_t), cmp_u64);
  for (uint64_t g = 0; g < ngenomes; ++g) gsize[g] = 0;
  uint64_t nc = 0;
  for (uint64_t i = 0; i < npairs; ++i) {
    if (i > 0 && cg[i] == cg[i - 1]) continue;
    cid[nc] = (uint32_t)(cg[i] >> 32);
    cgen[nc] = (uint32_t)cg[i];
    gsize[cgen[nc]] += 1;
    ++nc;
  }
  *ncount = nc;
  free(a); free(b); free(d); free(cg);
  return MG_OK;
}

/* matched[i] = 1 when pair i's hash is in the read sketch with count >= ci (kmc -ci2 + kmc_tools intersect, :50-56) */
int mgo_refpipe_matched(const uint64_t* q_hashes, const uint32_t* q_counts, uint64_t qn, uint32_t ci,
                        const uint64_t* pair_hash, uint64_t npairs, uint8_t* matched) {
  for (uint64_t i = 0; i < npairs; ++i) {
    const uint64_t h = pair_hash[i];
    uint64_t lo = 0, hi = qn;
    while (lo < hi) {
      uint64_t mid = lo + (hi - lo) / 2;
      if (q_hashes[mid] < h) lo = mid + 1; else hi = mid;
    }
    matched[i] = (uint8_t)(lo < qn && q_hashes[lo] == h && q_counts[lo] >= ci);
  }
  return MG_OK;
}

/* The column of one k < kmax: hits[g] = #{(p, g) in the count list : p marked by a matched pair's pa / pb} (:73-76) */
int mgo_refpipe_hits_k(const uint8_t* matched, const uint32_t* pa, const uint32_t* pb, uint64_t npairs, uint64_t nprefix,
                       const uint32_t* cid, const uint32_t* cgen, uint64_t ncount, uint64_t ngenomes, uint32_t* out_hits) {
  uint8_t* marked = (uint8_t*)calloc(nprefix + 1, 1);
  if (!marked) return MG_ERR_NOMEM;
  for (uint64_t i = 0; i < npairs; ++i) {
    if (!matched[i]) continue;
    marked[pa[i]] = 1;
    if (pb[i] != 0xffffffffu) marked[pb[i]] = 1;
  }
  for (uint64_t g = 0; g < ngenomes; ++g) out_hits[g] = 0;
  for (uint64_t i = 0; i < ncount; ++i)
    if (marked[cid[i]]) out_hits[cgen[i]] += 1;
  free(marked);
  return MG_OK;
}

/* ---------------------------------------------------------------------- *
 * Stage A of the reference pipeline BY K-MER IDENTITY (the build's default since round 6).
 *
 * `kmc -k<kmax> -ci<ci> -cs<cs>` counts the reads' CANONICAL k-mers — the lexicographically smaller strand — and `kmc_tools simple
 * ... intersect` keeps those that are k-mers of the genome sketches (scripts/select_db.py:50-59): k-mers are compared as what they
 * ARE, nothing on the read side is hashed (MurmurHash3 only SELECTS a genome's sketch, when the table is built).  So:
 *   counts[i] = occurrences, over all reads, of windows (free of non-ACGT symbols) whose canonical k-mer equals the canonical
 *               form of pair i's kept k-mer (khi / klo: 2-bit packed, first base most significant, as mgo_sketch_genomes_kmers
 *               leaves them, in any orientation), saturating at cs when cs > 0;
 *   a pair is MATCHED when counts[i] >= ci.
 * mgo_refpipe_matched (by hash value) gives the same pairs unless two different k-mers share a hash.  tests/indep_sketch.py:
 * refpipe_query states this on strings.  PARITY UNPINNED like the rest of stage A/B.
 * ---------------------------------------------------------------------- */
/* the smaller of codes c[0..k) and their reverse complement, packed */
static void canonical_pack(const uint8_t* c, int k, uint64_t* hi, uint64_t* lo) {
  uint8_t r[MG_MAX_K];
  for (int t = 0; t < k; ++t) r[k - 1 - t] = (uint8_t)(3 - c[t]);
  pack_codes(memcmp(c, r, (size_t)k) <= 0 ? c : r, k, hi, lo);
}

/* The table's canonical k-mers as a set one can look a k-mer up in — KMC's database of the sketches' k-mers
 * (local_tests/retrain_and_test_metalign.sh:59-66).  An open-addressed table keyed by the packed canonical k-mer (the slot a
 * key starts from is a function of the oracle's own choosing: it decides nothing but where to look); head[i] = the first pair
 * that holds pair i's canonical k-mer — where that k-mer is counted.  Read-only once built: mgo_kmer_table_count may run on
 * several threads at once, each with counters of its own (bench.py's cpu_baseline). */
typedef struct mgo_kmer_table {
  int k;
  uint64_t npairs, nslots;
  uint64_t* hi; uint64_t* lo;
  uint32_t* slot_head;   /* 0xffffffff = empty */
  uint32_t* head;        /* [npairs] */
} mgo_kmer_table;

static inline uint64_t kt_start(uint64_t hi, uint64_t lo, uint64_t mask) {
  uint64_t x = lo * 0x9e3779b97f4a7c15ULL ^ (hi + 0x632be59bd9b4e019ULL) * 0xff51afd7ed558ccdULL;
  x ^= x >> 29;
  return x & mask;
}

void mgo_kmer_table_free(mgo_kmer_table* t) {
  if (!t) return;
  free(t->hi); free(t->lo); free(t->slot_head); free(t->head); free(t);
}

mgo_kmer_table* mgo_kmer_table_new(const uint64_t* khi, const uint64_t* klo, uint64_t npairs, int k) {
  if (k < 1 || k > MG_MAX_K || npairs > 0xfffffff0ULL) return NULL;
  mgo_kmer_table* t = (mgo_kmer_table*)calloc(1, sizeof(mgo_kmer_table));
  if (!t) return NULL;
  t->k = k; t->npairs = npairs;
  t->nslots = 16;
  while (t->nslots < 2 * npairs + 2) t->nslots <<= 1;
  t->hi = (uint64_t*)malloc(t->nslots * sizeof(uint64_t));
  t->lo = (uint64_t*)malloc(t->nslots * sizeof(uint64_t));
  t->slot_head = (uint32_t*)malloc(t->nslots * sizeof(uint32_t));
  t->head = (uint32_t*)malloc((npairs + 1) * sizeof(uint32_t));
  if (!t->hi || !t->lo || !t->slot_head || !t->head) { mgo_kmer_table_free(t); return NULL; }
  memset(t->slot_head, 0xff, t->nslots * sizeof(uint32_t));
  const uint64_t mask = t->nslots - 1;
  for (uint64_t i = 0; i < npairs; ++i) {   /* ascending i: the first pair of a k-mer claims its slot */
    uint8_t c[MG_MAX_K];
    uint64_t hi, lo;
    unpack_codes(khi[i], klo[i], k, c);
    canonical_pack(c, k, &hi, &lo);
    uint64_t s = kt_start(hi, lo, mask);
    while (t->slot_head[s] != 0xffffffffu && !(t->hi[s] == hi && t->lo[s] == lo)) s = (s + 1) & mask;
    if (t->slot_head[s] == 0xffffffffu) { t->hi[s] = hi; t->lo[s] = lo; t->slot_head[s] = (uint32_t)i; }
    t->head[i] = t->slot_head[s];
  }
  return t;
}

/* counts[head] += occurrences among the reads' canonical k-mers (counts: uint64[npairs], the caller's, zeroed by it) */
int mgo_kmer_table_count(const mgo_kmer_table* t, const uint8_t* bases, const uint64_t* offsets, uint64_t nreads, uint64_t* counts,
                         uint64_t* out_kmers_seen) {
  if (!t || !counts) return MG_ERR_ARG;
  const int k = t->k;
  const uint64_t mask = t->nslots - 1;
  uint64_t seen = 0;
  for (uint64_t r = 0; r < nreads; ++r) {
    const uint8_t* seq = bases + offsets[r];
    const uint64_t len = offsets[r + 1] - offsets[r];
    uint64_t run = 0;
    for (uint64_t j = 0; j < len; ++j) {
      run = base_code(seq[j]) >= 0 ? run + 1 : 0;
      if (run < (uint64_t)k) continue;
      ++seen;
      uint8_t c[MG_MAX_K];
      for (int u = 0; u < k; ++u) c[u] = (uint8_t)base_code(seq[j + 1 - (uint64_t)k + (uint64_t)u]);
      uint64_t hi, lo;
      canonical_pack(c, k, &hi, &lo);
      uint64_t s = kt_start(hi, lo, mask);
      while (t->slot_head[s] != 0xffffffffu && !(t->hi[s] == hi && t->lo[s] == lo)) s = (s + 1) & mask;
      if (t->slot_head[s] != 0xffffffffu) ++counts[t->slot_head[s]];
    }
  }
  if (out_kmers_seen) *out_kmers_seen = seen;
  return MG_OK;
}

/* per pair: min(occurrences of its k-mer, cs) (cs = 0: exact, clamped to 32 bits) */
int mgo_kmer_table_per_pair(const mgo_kmer_table* t, const uint64_t* counts, uint32_t cs, uint32_t* out_counts) {
  if (!t || !counts || !out_counts) return MG_ERR_ARG;
  for (uint64_t i = 0; i < t->npairs; ++i) {
    uint64_t c = counts[t->head[i]];
    if (cs && c > cs) c = cs;
    out_counts[i] = c > 0xffffffffULL ? 0xffffffffu : (uint32_t)c;
  }
  return MG_OK;
}

int mgo_refpipe_count_kmers(const uint8_t* bases, const uint64_t* offsets, uint64_t nreads, int k, uint32_t cs,
                            const uint64_t* khi, const uint64_t* klo, uint64_t npairs, uint32_t* out_counts,
                            uint64_t* out_kmers_seen) {
  if (k < 1 || k > MG_MAX_K) return MG_ERR_ARG;
  mgo_kmer_table* t = mgo_kmer_table_new(khi, klo, npairs, k);
  uint64_t* cnt = (uint64_t*)calloc(npairs + 1, sizeof(uint64_t));
  if (!t || !cnt) { mgo_kmer_table_free(t); free(cnt); return MG_ERR_NOMEM; }
  int rc = mgo_kmer_table_count(t, bases, offsets, nreads, cnt, out_kmers_seen);
  if (rc == MG_OK) rc = mgo_kmer_table_per_pair(t, cnt, cs, out_counts);
  mgo_kmer_table_free(t);
  free(cnt);
  return rc;
}

/* Stage B.  See include/metalign_hip.h (mg_containment). */
int mgo_containment(const uint64_t* q_hashes, const uint32_t* q_counts, uint64_t qn,
                    int q_truncated, uint32_t ci, const uint64_t* db_hashes,
                    const uint64_t* db_offsets, uint64_t ngenomes, uint32_t* out_hits,
                    uint32_t* out_sizes) {
  uint64_t bound = (q_truncated && qn > 0) ? q_hashes[qn - 1] : UINT64_MAX;
  for (uint64_t g = 0; g < ngenomes; ++g) {
    uint32_t hits = 0, size = 0;
    for (uint64_t i = db_offsets[g]; i < db_offsets[g + 1]; ++i) {
      uint64_t h = db_hashes[i];
      if (h > bound) continue;
      ++size;
      uint64_t lo = 0, hi = qn; /* lower_bound */
      while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (q_hashes[mid] < h) lo = mid + 1; else hi = mid;
      }
      if (lo < qn && q_hashes[lo] == h && q_counts[lo] >= ci) ++hits;
    }
    out_hits[g] = hits;
    out_sizes[g] = size;
  }
  return MG_OK;
}

/* ---------------------------------------------------------------------- *
 * Stage C.  Sequential restatement of map_and_process
 * (scripts/map_and_profile.py:193-264) over pre-tokenised records.
 * The variable names mirror the reference so the two can be read side by side.
 * ---------------------------------------------------------------------- */
typedef struct { int pair1, pair2, chimeric; } flagbits;

static inline flagbits parse_flag(uint32_t flag) { /* :104-111 (is_bad handled at ingest) */
  flagbits f;
  f.pair1 = (flag & 1u) && (flag & 64u);
  f.pair2 = (flag & 1u) && (flag & 128u);
  f.chimeric = (flag & 2048u) != 0;
  return f;
}

static inline int filter_line(const mg_aln_rec* r, double pct_id) { /* :86-100 */
  return (double)r->matched / (double)r->total < pct_id;
}

typedef struct {
  int kind;          /* 0 = Ambiguous, 1 = unique, 2 = multimapped */
  uint32_t taxid;    /* unique: dense taxon id */
  uint64_t hitlen;
} read_verdict;

/* process_read (:152-176) on lines hits[0..nh) (indices into recs); when the
 * verdict is multimapped the taxon list is appended to mm_tax. */
static read_verdict process_read(const mg_aln_rec* recs, const uint64_t* hits, uint64_t nh,
                                 int pair1, int pair2, long pair1maps, long pair2maps,
                                 const uint32_t* ref2tax, double pct_id, uint64_t* kept,
                                 uint32_t* mm_tax, uint64_t* mm_n, uint64_t mm_cap, int* overflow) {
  read_verdict v = {0, 0, 0};
  /* clean_read_hits (:130-147) */
  uint64_t nk = 0, hitlen = 0;
  for (uint64_t i = 0; i < nh; ++i) {
    const mg_aln_rec* r = &recs[hits[i]];
    flagbits f = parse_flag(r->flag_len & MG_REC_FLAG_MASK);
    if (filter_line(r, pct_id) || f.chimeric) {
      if (f.pair1) pair1maps -= 1;
      else if (f.pair2) pair2maps -= 1;
    } else {
      kept[nk++] = hits[i];
    }
    hitlen += r->flag_len >> MG_REC_LEN_SHIFT; /* len(SEQ), 0 when SEQ == '*' (:142-144) */
  }
  v.hitlen = hitlen;
  if (nk == 0) return v; /* :155-156 */
#define TAX(i) (ref2tax[recs[kept[(i)]].ref_new & MG_REC_REF_MASK])
  if (pair1 || pair2) { /* :157 */
    if (pair1maps + pair2maps == 1) { v.kind = 1; v.taxid = TAX(0); return v; } /* :158-160 */
    /* intersect_read_hits (:115-125) */
    if (pair1maps == 0 || pair2maps == 0) return v; /* -> len(intersect)==0 -> Ambiguous */
    uint64_t split = (uint64_t)(pair1maps < 0 ? 0 : pair1maps);
    if (split > nk) split = nk;
    /* distinct taxa of the first `split` hits that also occur among the rest */
    uint64_t ndistinct = 0;
    for (uint64_t i = 0; i < split; ++i) {
      uint32_t t = TAX(i);
      int in2 = 0, dup = 0;
      for (uint64_t j = split; j < nk && !in2; ++j) in2 = TAX(j) == t;
      if (!in2) continue;
      for (uint64_t j = 0; j < i && !dup; ++j) dup = TAX(j) == t;
      if (!dup) ++ndistinct;
    }
    if (ndistinct == 0) return v;                                   /* :164-165 */
    if (ndistinct == 1) { v.kind = 1; v.taxid = TAX(0); return v; } /* :166-167 */
    v.kind = 2;                                                     /* :168-169 */
    for (uint64_t h = 0; h < nk; ++h) {
      uint32_t t = TAX(h);
      int in1 = 0, in2 = 0;
      for (uint64_t j = 0; j < split && !in1; ++j) in1 = TAX(j) == t;
      for (uint64_t j = split; j < nk && !in2; ++j) in2 = TAX(j) == t;
      if (in1 && in2) {
        if (*mm_n < mm_cap) mm_tax[(*mm_n)++] = t; else *overflow = 1;
      }
    }
    return v;
  }
  if (pair1maps > 1) { /* single end, multimapped (:172-173) */
    v.kind = 2;
    for (uint64_t h = 0; h < nk; ++h) {
      if (*mm_n < mm_cap) mm_tax[(*mm_n)++] = TAX(h); else *overflow = 1;
    }
    return v;
  }
  v.kind = 1; v.taxid = TAX(0); /* :174-176 */
  return v;
#undef TAX
}

int mgo_profile_assign(const mg_aln_rec* recs, uint64_t nrecs, const uint32_t* ref2tax,
                       uint32_t nref, uint32_t ntax, double pct_id, uint64_t* out_count,
                       uint64_t* out_bases, uint64_t* out_first_seen, uint64_t* out_tot_rds,
                       uint64_t* out_n_ambig, uint64_t* mm_offsets, uint32_t* mm_tax,
                       uint64_t* mm_hitlen, uint64_t* mm_read, uint64_t mm_cap_reads,
                       uint64_t mm_cap_entries, uint64_t* mm_nreads, uint64_t* mm_nentries) {
  (void)nref;
  for (uint32_t t = 0; t < ntax; ++t) { out_count[t] = 0; out_bases[t] = 0; out_first_seen[t] = UINT64_MAX; }
  uint64_t cap = 16, nh = 0;
  uint64_t* read_hits = (uint64_t*)malloc(cap * sizeof(uint64_t));
  uint64_t* kept = (uint64_t*)malloc(cap * sizeof(uint64_t));
  if (!read_hits || !kept) { free(read_hits); free(kept); return MG_ERR_NOMEM; }
  long pair1maps = 0, pair2maps = 0;
  uint64_t tot_rds = 0, n_ambig = 0, n_mm = 0, n_ent = 0;
  int overflow = 0;
  if (mm_cap_reads > 0 || mm_offsets) mm_offsets[0] = 0;
  for (uint64_t i = 0; i < nrecs; ++i) {
    const mg_aln_rec* r = &recs[i];
    flagbits f = parse_flag(r->flag_len & MG_REC_FLAG_MASK);
    if (r->ref_new & MG_REC_NEW_BIT) { /* read != prev_read (:220) */
      tot_rds += 1;
      uint64_t ent_before = n_ent;
      read_verdict v = process_read(recs, read_hits, nh, f.pair1, f.pair2, pair1maps, pair2maps,
                                    ref2tax, pct_id, kept, mm_tax, &n_ent, mm_cap_entries, &overflow);
      nh = 0; pair1maps = 0; pair2maps = 0; /* :227 */
      if (v.kind == 0) { /* Ambiguous: count and DROP this line (:229-232) */
        n_ambig += 1;
        continue;
      }
      uint64_t read_index = tot_rds - 2; /* index of the read just decided (0-based) */
      if (v.kind == 1) { /* :235-240 */
        out_count[v.taxid] += 1;
        out_bases[v.taxid] += v.hitlen;
        if (out_first_seen[v.taxid] == UINT64_MAX) out_first_seen[v.taxid] = read_index;
      } else { /* :245-248 */
        if (n_mm < mm_cap_reads) {
          mm_hitlen[n_mm] = v.hitlen;
          mm_read[n_mm] = read_index;
          mm_offsets[n_mm + 1] = n_ent;
          ++n_mm;
        } else {
          overflow = 1;
          n_ent = ent_before;
        }
      }
    }
    pair1maps += (f.pair1 || !(f.pair1 || f.pair2)) ? 1 : 0; /* :257 */
    pair2maps += f.pair2 ? 1 : 0;                            /* :258 */
    if (nh == cap) {
      cap *= 2;
      uint64_t* a = (uint64_t*)realloc(read_hits, cap * sizeof(uint64_t));
      if (!a) { free(read_hits); free(kept); return MG_ERR_NOMEM; }
      read_hits = a;
      uint64_t* b = (uint64_t*)realloc(kept, cap * sizeof(uint64_t));
      if (!b) { free(read_hits); free(kept); return MG_ERR_NOMEM; }
      kept = b;
    }
    read_hits[nh++] = i; /* :259 */
  }
  free(read_hits); free(kept);
  *out_tot_rds = tot_rds;
  *out_n_ambig = n_ambig;
  *mm_nreads = n_mm;
  *mm_nentries = n_ent;
  return overflow ? MG_ERR_CAPACITY : MG_OK;
}
