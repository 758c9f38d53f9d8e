/* selftest.c — sanitizer driver for the CPU oracle (test infrastructure).
 * Built with -fsanitize=address,undefined by tests/test_oracle_sanitizers.py; exercises every oracle entry
 * point on small adversarial inputs (empty, ragged, N-only, k = 1 / 64, truncated sketches, odd record
 * streams) so that out-of-bounds reads and undefined shifts in the restatement cannot hide. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/metalign_hip.h"

void mgo_murmur3_x64_128(const void*, int, uint32_t, uint64_t[2]);
uint64_t mgo_kmer_hashes(const uint8_t*, uint64_t, int, uint64_t*, uint8_t*);
int mgo_sketch_reads(const uint8_t*, const uint64_t*, uint64_t, int, uint64_t, uint64_t, uint32_t, uint64_t*, uint32_t*, uint64_t,
                     uint64_t*, int*, uint64_t*);
int mgo_sketch_genomes(const uint8_t*, const uint64_t*, uint64_t, int, uint64_t, uint64_t*, uint64_t*);
int mgo_containment(const uint64_t*, const uint32_t*, uint64_t, int, uint32_t, const uint64_t*, const uint64_t*, uint64_t,
                    uint32_t*, uint32_t*);
int mgo_profile_assign(const mg_aln_rec*, uint64_t, const uint32_t*, uint32_t, uint32_t, double, uint64_t*, uint64_t*,
                       uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint32_t*, uint64_t*, uint64_t*, uint64_t, uint64_t,
                       uint64_t*, uint64_t*);

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(void) {
  /* murmur over every length 0..70 */
  uint8_t buf[80];
  for (int i = 0; i < 80; ++i) buf[i] = (uint8_t)rnd();
  uint64_t acc = 0, out[2];
  for (int len = 0; len <= 70; ++len) { mgo_murmur3_x64_128(buf, len, (uint32_t)len, out); acc ^= out[0] ^ out[1]; }
  /* reads: empty, short, N-rich, exact k, long */
  const int nreads = 40;
  uint64_t offs[41];
  size_t total = 0;
  int lens[40];
  for (int r = 0; r < nreads; ++r) { lens[r] = r == 0 ? 0 : (r == 1 ? 3 : (int)(rnd() % 300)); total += (size_t)lens[r]; }
  uint8_t* bases = (uint8_t*)malloc(total + 1);
  size_t w = 0;
  for (int r = 0; r < nreads; ++r) { offs[r] = w; for (int i = 0; i < lens[r]; ++i) bases[w++] = (uint8_t)"ACGTNacgtn"[rnd() % 10]; }
  offs[nreads] = w;
  const int ks[] = {1, 2, 15, 16, 17, 21, 31, 32, 33, 48, 51, 60, 63, 64};
  for (unsigned ki = 0; ki < sizeof(ks) / sizeof(ks[0]); ++ki) {
    const int k = ks[ki];
    uint64_t* h = (uint64_t*)malloc((total + 1) * 8);
    uint32_t* c = (uint32_t*)malloc((total + 1) * 4);
    uint64_t n = 0, seen = 0;
    int trunc = 0;
    if (mgo_sketch_reads(bases, offs, nreads, k, UINT64_MAX, 0, 0, h, c, total + 1, &n, &trunc, &seen)) return 1;
    uint64_t n2 = 0;
    if (mgo_sketch_reads(bases, offs, nreads, k, UINT64_MAX / 3, 7, 3, h, c, total + 1, &n2, &trunc, &seen)) return 2;
    if (n2 > 7) return 3;
    uint64_t* gh = (uint64_t*)malloc((size_t)nreads * 5 * 8 + 8);
    uint64_t go[41];
    if (mgo_sketch_genomes(bases, offs, nreads, k, 5, gh, go)) return 4;
    uint32_t hits[40], sizes[40];
    if (mgo_containment(h, c, n2, trunc, 1, gh, go, nreads, hits, sizes)) return 5;
    for (int g = 0; g < nreads; ++g) { if (hits[g] > sizes[g] || sizes[g] > 5) return 6; acc += hits[g]; }
    if (lens[5] >= k) {
      uint64_t* ph = (uint64_t*)malloc((size_t)lens[5] * 8 + 8);
      uint8_t* pv = (uint8_t*)malloc((size_t)lens[5] + 1);
      acc += mgo_kmer_hashes(bases + offs[5], (uint64_t)lens[5], k, ph, pv);
      free(ph); free(pv);
    }
    free(h); free(c); free(gh);
  }
  /* stage C on random record streams, including none and one */
  for (int trial = 0; trial < 50; ++trial) {
    const uint64_t n = trial < 2 ? (uint64_t)trial : rnd() % 400;
    mg_aln_rec* recs = (mg_aln_rec*)malloc((n + 1) * sizeof(mg_aln_rec));
    uint32_t ref2tax[13];
    for (int i = 0; i < 13; ++i) ref2tax[i] = (uint32_t)(rnd() % 5);
    const uint32_t flags[] = {0, 16, 256, 272, 2048, 99, 147, 355, 403, 65, 129, 73, 137, 1};
    for (uint64_t i = 0; i < n; ++i) {
      recs[i].ref_new = (uint32_t)(rnd() % 13) | ((i == 0 || rnd() % 3) ? MG_REC_NEW_BIT : 0);
      recs[i].total = 1 + (uint32_t)(rnd() % 150);
      recs[i].matched = (uint32_t)(rnd() % (recs[i].total + 1));
      recs[i].flag_len = flags[rnd() % 14] | ((uint32_t)(rnd() % 2 ? recs[i].total : 0) << MG_REC_LEN_SHIFT);
    }
    uint64_t count[5], bs[5], first[5], tot, amb, nmm, nent;
    uint64_t* mo = (uint64_t*)malloc((n + 2) * 8);
    uint32_t* mt = (uint32_t*)malloc((n + 1) * 4);
    uint64_t* ml = (uint64_t*)malloc((n + 1) * 8);
    uint64_t* mr = (uint64_t*)malloc((n + 1) * 8);
    if (mgo_profile_assign(recs, n, ref2tax, 13, 5, 0.5, count, bs, first, &tot, &amb, mo, mt, ml, mr, n + 1, n + 1, &nmm, &nent)) return 7;
    uint64_t uniq = 0;
    for (int t = 0; t < 5; ++t) uniq += count[t];
    if (tot > 0 && uniq + nmm + (amb - 1) != tot - 1) return 8; /* every processed read is classified exactly once */
    free(recs); free(mo); free(mt); free(ml); free(mr);
  }
  free(bases);
  printf("oracle selftest ok %llu\n", (unsigned long long)acc);
  return 0;
}
