// mg_sketch_cmash.hip — the one-k stage-A / A' kernels for hash definition 1: min(MurmurHash3(k-mer), MurmurHash3(reverse
// complement)) mod 9999999999971, CMash's MinHash.CountEstimator as SURVEY.md §8(c) recollects it (unverified; see
// include/metalign_hip.h: mg_set_hash_mode).  The templates are mg_sketch_kernel.h's; this translation unit only
// instantiates them so that the two definitions compile side by side — for the k of kCmashKs (the reference's 30 / 40 / 50 / 60,
// bench.py's 21 / 31 / 51, every fifth k, the word boundaries): a quarter of the 2 x 64 kernels rounds 2-4 carried in a 23 MB library.
// Any other k in this mode is refused with the list (mode 0 takes every k from 1 to 64).
// Replaces (in that mode): k-mer hashing inside CMash's MakeStreamingDNADatabase.py / StreamingQueryDNADatabase.py
// (local_tests/retrain_and_test_metalign.sh:49-54, scripts/select_db.py:73-76).
#include "mg_sketch_kernel.h"

namespace mg {

template <int... Ks> struct KSet {};
using CmashKs = KSet<1, 5, 10, 15, 16, 20, 21, 25, 30, 31, 32, 33, 35, 40, 45, 50, 51, 55, 60, 63, 64>;
static const char kCmashKsText[] = "1, 5, 10, 15, 16, 20, 21, 25, 30, 31, 32, 33, 35, 40, 45, 50, 51, 55, 60, 63, 64";
template <class F, int... Ks>
static bool dispatch_listed(int k, F&& fn, KSet<Ks...>) {
  return ((k == Ks ? (fn.template operator()<Ks>(), true) : false) || ...);
}
static int refuse(int k) { return fail(MG_ERR_ARG, "hash mode 1 (the CMash recollection) is built for k in {%s}, not k = %d", kCmashKsText, k); }

int launch_sketch_reads_cmash(int k, unsigned grid, size_t lds, hipStream_t st, const uint8_t* d_bases, const uint64_t* d_offsets,
                              uint64_t nreads, uint64_t hmax, uint64_t* d_cand, uint64_t cap, unsigned long long* d_counters,
                              Slot* d_tab, unsigned bucket_shift, unsigned stage_bytes, const uint32_t* fbits, uint64_t fmask,
                              uint32_t cs_word) {
  const bool ok = dispatch_listed(k, [&]<int K>() {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sketch_reads<K, kHashCmash>), dim3(grid), dim3(kBlock), lds, st, d_bases, d_offsets, nreads,
                       hmax, d_cand, cap, d_counters, d_tab, bucket_shift, stage_bytes, fbits, fmask, cs_word);
  }, CmashKs());
  return ok ? MG_OK : refuse(k);
}

int launch_hash_positions_cmash(int k, unsigned grid, hipStream_t st, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nseq,
                                uint64_t nbases, uint64_t* d_out, bool tagged) {
  const bool ok = dispatch_listed(k, [&]<int K>() {
    if (tagged)  // the kept strand in bit 63 (the prefix-table builder, mg_sketch_genomes_prefix)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_hash_positions<K, kHashCmashTagged>), dim3(grid), dim3(256), 0, st, d_bases, d_offsets, nseq,
                         nbases, d_out);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_hash_positions<K, kHashCmash>), dim3(grid), dim3(256), 0, st, d_bases, d_offsets, nseq, nbases,
                         d_out);
  }, CmashKs());
  return ok ? MG_OK : refuse(k);
}

// `build_db --sketch_hash forward`: the position hashes that SELECT a genome's sketch (mg_kmer.h, kHashForward), for the same list of k
int launch_hash_positions_forward(int k, unsigned grid, hipStream_t st, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nseq,
                                  uint64_t nbases, uint64_t* d_out) {
  const bool ok = dispatch_listed(k, [&]<int K>() {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_hash_positions<K, kHashForward>), dim3(grid), dim3(256), 0, st, d_bases, d_offsets, nseq, nbases, d_out);
  }, CmashKs());
  return ok ? MG_OK : fail(MG_ERR_ARG, "sketches selected by the forward hash are built for k in {%s}, not k = %d", kCmashKsText, k);
}

}  // namespace mg

// the k hash mode 1 is built for, as text ("1, 5, 10, ..."): a command line checks its k list against it before any genome is read
extern "C" const char* mg_hash_mode1_ks(void) { return mg::kCmashKsText; }
