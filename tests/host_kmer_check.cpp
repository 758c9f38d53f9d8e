// host_kmer_check.cpp — the kernels' k-mer header (metalign_amd/csrc/mg_kmer.h) compiled for the HOST: the 2-bit roller,
// the first-multiply tables and the table-driven MurmurHash3 run here exactly as the device code is written (the one
// device builtin, v_alignbit, is restated in the header's MG_HOST_CHECK section), so that the build container — which has
// no GPU — can hold them against the oracle for every k.  Test infrastructure: tests/test_kmer_header_host.py.
//
// stdin: one sequence of [ACGTacgtN...] per line.  stdout, per line, per hash definition ("mode 0" / "mode 1" lines) and
// per k in 1..64: "k" then the hash of the k-mer ENDING at every position (hex; '-' where there is none), as
// Roller<k>::hash gives it; then, for the fused kernels' k sets, "s k kmax" lines with hash_suffix<k, kmax> of a Roller<kmax>.
#define MG_HOST_CHECK 1
#include "../metalign_amd/csrc/mg_kmer.h"

#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

static std::vector<uint64_t> tab;

template <int K, int HM>
static void one_k(const std::string& seq) {
  mg::Roller<K> r;
  r.reset();
  std::printf("%d", K);
  for (char ch : seq) {
    uint32_t c;
    if (mg::decode_base((unsigned char)ch, c)) {
      r.push(c);
      if (r.full()) { std::printf(" %016llx", (unsigned long long)r.template hash<HM>(tab.data())); continue; }
    } else {
      r.run = 0;
    }
    std::printf(" -");
  }
  std::printf("\n");
}

template <int K, int KMAX, int HM>
static void one_suffix(const std::string& seq) {
  mg::Roller<KMAX> r;
  r.reset();
  std::printf("s %d %d", K, KMAX);
  for (char ch : seq) {
    uint32_t c;
    if (mg::decode_base((unsigned char)ch, c)) {
      r.push(c);
      if (r.run >= K) { std::printf(" %016llx", (unsigned long long)mg::hash_suffix<K, KMAX, HM>(r, tab.data())); continue; }
    } else {
      r.run = 0;
    }
    std::printf(" -");
  }
  std::printf("\n");
}

template <int HM, int... K>
static void all_k(const std::string& seq, std::integer_sequence<int, K...>) { (one_k<K + 1, HM>(seq), ...); }

template <int HM>
static void one_mode(const std::string& line) {
  std::printf("mode %d\n", HM);
  all_k<HM>(line, std::make_integer_sequence<int, 64>{});
  one_suffix<21, 51, HM>(line); one_suffix<31, 51, HM>(line); one_suffix<51, 51, HM>(line);
  one_suffix<30, 60, HM>(line); one_suffix<40, 60, HM>(line); one_suffix<50, 60, HM>(line); one_suffix<60, 60, HM>(line);
  one_suffix<1, 64, HM>(line); one_suffix<32, 64, HM>(line); one_suffix<33, 64, HM>(line); one_suffix<17, 33, HM>(line); one_suffix<4, 5, HM>(line);
}

int main() {
  tab.resize(mg::kHashTabEntries);
  for (int e = 0; e < mg::kHashTabEntries; ++e) tab[e] = mg::hash_tab_entry(e);
  std::string line;
  while (std::getline(std::cin, line)) {
    std::printf("seq %zu\n", line.size());
    one_mode<mg::kHashCanonical>(line);
    one_mode<mg::kHashCmash>(line);
  }
  return 0;
}
