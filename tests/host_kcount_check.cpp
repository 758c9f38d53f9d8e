// host_kcount_check.cpp — the per-lane code of stage A by k-mer identity (metalign_amd/csrc/mg_kcount_core.h) compiled for the
// HOST: the base stream and its packing, the sliding minimizer (kc_walk, every k and all three modes), the restart after a
// full list, the minimizer of a table k-mer, the gate, the buckets, the signature and the exact comparison run here as the
// device code is written, one lane after the other, so that the build container — which has no GPU — can hold them against
// the oracle (tests/test_kcount_core_host.py).  What is NOT covered here is the wavefront glue of mg_kcount.hip (LDS addresses,
// ballots, the drain's batches, atomics): the GPU tests hold that to the same oracle.
//
// stdin:  "k cap ntable nreads lead cs bb\n", then ntable lines with a table k-mer each (ACGT, any strand), then nreads lines with a
//         read each (anything; may be empty).  cap = slots of a lane's event list (small values force restarts); lead = bytes
//         in front of the first read in the buffer (the tile then starts off a 16-byte boundary); cs = the counters' saturation
//         value (0: exact counts; else a k-mer seen cs times is skipped from then on and its minimizer marked done); bb > 0:
//         2^bb buckets (a handful of buckets: nearly every k-mer lives in the overflow list).
// stdout: one line per table k-mer: the number of windows of the reads whose canonical k-mer equals its canonical form;
//         then "kmers N runs R passed P restarts S".
#define MG_HOST_CHECK 1
#include "../metalign_amd/csrc/mg_kcount_core.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>

using namespace mg;

static int g_cap = 12;
static uint32_t g_cs = 0;  // counters saturate here (0: exact); > 0 exercises the "done" marks
static uint32_t g_gate_extra = kKcGateExtra;  // (bb > 0 also takes this to 0)
static int g_bb = 0;       // > 0: this many bucket bits instead of about one bucket per k-mer (few buckets: most overflow their four slots)

struct HostOut {
  static constexpr uint32_t kCap = 0;  // (not used: the capacity is a run-time value here, see below)
};
// kc_walk reads Out::kCap as a compile-time constant: one instantiation per capacity the test uses
template <uint32_t CAP>
struct HostOutN {
  static constexpr uint32_t kCap = CAP;
  std::vector<uint32_t> ev;  // CAP + 1 slots
  HostOutN() : ev(CAP + 1) {}
  void put(uint32_t slot, uint32_t word, uint32_t info) { ev.at(slot) = kc_event(word, info); }
  bool any_full(uint32_t) const { return false; }  // (the device leaves a block early when some lane is full: fewer wasted steps, same events)
  uint32_t last_window(uint32_t slot) const { return (ev.at(slot) >> 10) & 1023u; }
  uint32_t wave_min(uint32_t v) const { return v; }  // (the caller takes the minimum over the lanes)
};

static KcWin win_from_string(const std::string& s) {
  KcWin x{{0, 0, 0, 0}};
  for (size_t i = 0; i < s.size(); ++i) {
    uint32_t c = s[i] == 'A' ? 0 : s[i] == 'C' ? 1 : s[i] == 'G' ? 2 : 3;
    x.w[i >> 4] |= c << (30 - 2 * (i & 15));
  }
  return x;
}

struct Index {
  std::vector<uint32_t> live, shared, counts, sat;
  std::vector<KcEntry> ent, prim, ovf;
  uint32_t bmask = 0, gshift = 0, cs = 0;
  KcIndexView view() { return KcIndexView{live.data(), shared.data(), prim.data(), ovf.data(), counts.data(), sat.data(), bmask, gshift, cs, 0u, 1u}; }
};

template <int K, uint32_t CAP>
static void run(const std::vector<std::string>& table, const std::vector<std::string>& reads, size_t lead) {
  // ---- the index, as mg_refdb_index_kmers builds it (here with a map) ----
  Index ix;
  ix.counts.assign(table.size() + 1, 0u);
  std::map<std::array<uint32_t, 4>, uint32_t> first;  // canonical k-mer -> the first pair that holds it
  std::vector<uint32_t> head(table.size());
  for (size_t i = 0; i < table.size(); ++i) {
    // through the table's right-aligned packing, as the device does
    uint64_t hi = 0, lo = 0;
    for (char ch : table[i]) {
      const uint64_t c = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
      hi = (hi << 2) | (lo >> 62);
      lo = (lo << 2) | c;
    }
    const KcWin direct = win_from_string(table[i]);
    const KcWin viaright = kc_from_right(hi, lo, K);
    if (!kc_equal(direct, viaright)) { std::fprintf(stderr, "kc_from_right differs from the direct packing\n"); std::exit(2); }
    const KcWin c = kc_canonical(viaright, K);
    const std::array<uint32_t, 4> key{c.w[0], c.w[1], c.w[2], c.w[3]};
    auto it = first.find(key);
    if (it == first.end()) it = first.emplace(key, (uint32_t)i).first;
    head[i] = it->second;
  }
  for (auto& kv : first) {
    KcEntry e;
    std::memcpy(e.w, kv.first.data(), 16);
    const KcWin x{{e.w[0], e.w[1], e.w[2], e.w[3]}};
    e.head = kv.second;
    e.pad = 0;
    uint32_t keys[kKcMaxCands];
    const int nk = kc_table_keys(x, K, keys);  // (one entry per hash the k-mer is filed under)
    if (nk < 1) { std::fprintf(stderr, "a k-mer without a key\n"); std::exit(2); }
    for (int t = 0; t < nk; ++t) { e.key = keys[t]; e.off = kc_table_key_offset(x, K, keys[t]); ix.ent.push_back(e); }
  }
  unsigned bb = 8;
  while (bb < 28 && (1ull << bb) < ix.ent.size()) ++bb;
  if (g_bb > 0) bb = (unsigned)g_bb;
  ix.bmask = (1u << bb) - 1u;
  const uint32_t bm = ix.bmask;
  std::stable_sort(ix.ent.begin(), ix.ent.end(), [bm](const KcEntry& a, const KcEntry& b) {
    return (a.key & bm) != (b.key & bm) ? (a.key & bm) < (b.key & bm) : a.key < b.key;
  });
  {  // the first two entries of a bucket in prim, the rest in ovf (mg_kcount.hip: k_kc_place)
    KcEntry none{};
    none.key = kKcNone;
    ix.prim.assign((size_t)kKcSlots << bb, none);
    size_t j = 0;
    for (uint32_t b = 0; b < (1u << bb); ++b) {
      size_t j0 = j;
      while (j < ix.ent.size() && (ix.ent[j].key & bm) == b) ++j;
      const size_t n = j - j0;
      for (size_t t = 0; t < n && t < kKcSlots; ++t) { ix.prim[kKcSlots * b + t] = ix.ent[j0 + t]; ix.prim[kKcSlots * b + t].pad = 0; }
      ix.prim[kKcSlots * b].pad = n > kKcSlots ? (uint32_t)(n - kKcSlots) : 0u;
      ix.prim[kKcSlots * b + 1].pad = (uint32_t)ix.ovf.size();
      for (size_t t = kKcSlots; t < n; ++t) ix.ovf.push_back(ix.ent[j0 + t]);
    }
    ix.ovf.push_back(none);
  }
  {  // the gate: few bits here (g_gate_extra = 0: about one per k-mer), so that hashes SHARE bits and the shared ones stay set
    const uint32_t gbits = kc_gate_bits(first.size(), g_gate_extra);
    ix.gshift = 32u - gbits;
    ix.live.assign(((size_t)1 << gbits) / 32 + 1, 0u);
    ix.shared.assign(ix.live.size(), 0u);
    for (size_t j = 0; j < ix.ent.size(); ++j) {
      if (j && ix.ent[j].key == ix.ent[j - 1].key) continue;  // (sorted by bucket, then hash: equal hashes are adjacent)
      const uint32_t g = ix.ent[j].key >> ix.gshift;
      if ((ix.live[g >> 5] >> (g & 31u)) & 1u) ix.shared[g >> 5] |= 1u << (g & 31u);
      ix.live[g >> 5] |= 1u << (g & 31u);
    }
  }
  ix.sat.assign(((size_t)kKcSlots << bb) + ix.ovf.size() + 2, 0u);  // (a counter per entry number)
  ix.cs = g_cs;
  const KcIndexView view = ix.view();

  // ---- the reads, one buffer, tiles of 64 ----
  std::vector<uint8_t> bases(lead, (uint8_t)'#');  // (bytes in front: whatever another tile or another buffer left there)
  std::vector<uint64_t> offsets{lead};
  for (auto& r : reads) { bases.insert(bases.end(), r.begin(), r.end()); offsets.push_back(bases.size()); }
  bases.resize(bases.size() + 64, 0);  // (the device's loads run up to 15 bytes past the tile)
  const uint64_t nreads = reads.size();
  uint64_t kmers = 0, runs = 0, passed = 0, restarts = 0;
  const uint32_t sd = 64 * 64;  // stage, dwords: generous — the chunked path is entered by read length here
  std::vector<uint32_t> fwd(sd + 16), inv(sd / 2 + 16);
  for (uint64_t tile = 0; tile * 64 < nreads; ++tile) {
    uint64_t beg[64], end[64];
    uint64_t maxlen = 0, t_end = 0;
    for (int l = 0; l < 64; ++l) {
      const uint64_t rd = tile * 64 + l;
      beg[l] = end[l] = 0;
      if (rd < nreads) { beg[l] = offsets[rd]; end[l] = offsets[rd + 1]; }
      maxlen = std::max(maxlen, end[l] - beg[l]);
      t_end = std::max(t_end, end[l]);
    }
    const uint64_t t_beg = beg[0];
    if (maxlen < (uint64_t)K) continue;
    auto drain_and_walk = [&](const uint32_t* p0, const uint32_t* len, uint32_t mlen, int mode) {
      const uint32_t nwmax = mlen - K + 1;
      uint32_t w0 = 0;
      bool first_call = true;
      do {
        std::vector<HostOutN<CAP>> out(64);
        uint32_t cnt[64], next = nwmax;
        for (int l = 0; l < 64; ++l) {
          cnt[l] = 0;
          uint32_t nx;
          if (mode == 0) nx = kc_walk<K, 0>(fwd.data(), inv.data(), p0[l], len[l], mlen, w0, out[l], cnt[l]);
          else if (mode == 1) nx = kc_walk<K, 1>(fwd.data(), inv.data(), p0[l], len[l], mlen, w0, out[l], cnt[l]);
          else nx = kc_walk<K, 2>(fwd.data(), inv.data(), p0[l], len[l], mlen, w0, out[l], cnt[l]);
          next = std::min(next, nx);
        }
        if (!first_call) ++restarts;
        first_call = false;
        if (next <= w0) { std::fprintf(stderr, "the walk made no progress\n"); std::exit(2); }
        for (int l = 0; l < 64; ++l) {
          uint32_t first = w0;  // (the walk leaves a run's first window out: the one after the run before it)
          for (uint32_t s = 0; s < cnt[l]; ++s) {
            const uint32_t ev = kc_event_first(out[l].ev[s], first);
            uint32_t i1 = ev & 1023u, i2 = (ev >> 10) & 1023u;
            if (kc_event_none(ev) || i1 >= next) continue;
            if (i2 >= next) i2 = next - 1;
            ++runs;
            const uint32_t key = kc_run_hash(fwd.data(), p0[l], kc_event_pos(ev), K);
            if (!kc_gate(view, key)) continue;
            ++passed;
            kc_match_run(view, fwd.data(), inv.data(), K, mode == 0, key, p0[l], kc_event_pos(ev), i1, i2);
          }
        }
        w0 = next;
      } while (w0 < nwmax);
    };
    if (maxlen <= kKcMaxRead) {
      const uintptr_t a_first = (uintptr_t)t_beg, a0 = a_first & ~(uintptr_t)15;  // (the buffer itself is taken as 16-byte aligned)
      const uint64_t shift = a_first - a0, nbytes = shift + (t_end - t_beg);
      const uint32_t nd = (uint32_t)((nbytes + 15) / 16);
      if (nd > sd) { std::fprintf(stderr, "tile above the harness's stage\n"); std::exit(2); }
      uint32_t notbase = 0;
      for (uint32_t i = 0; i < nd; ++i) {
        uint32_t vv[4], nb;
        std::memcpy(vv, bases.data() + a0 + 16 * (size_t)i, 16);
        fwd[i] = kc_pack16(vv, nb);
        notbase |= nb;
      }
      const bool bad = notbase != 0;
      if (bad) {
        uint16_t* inv16 = reinterpret_cast<uint16_t*>(inv.data());
        for (uint32_t i = 0; i < nd + 4; ++i) {
          uint32_t bits = 0;
          if (i < nd) { uint32_t vv[4]; std::memcpy(vv, bases.data() + a0 + 16 * (size_t)i, 16); bits = kc_notbase16(vv); }
          inv16[i ^ 1u] = (uint16_t)bits;
        }
      }
      uint32_t p0[64], len[64];
      bool ragged = false;
      for (int l = 0; l < 64; ++l) {
        len[l] = (uint32_t)(end[l] - beg[l]);
        p0[l] = tile * 64 + l < nreads ? (uint32_t)(shift + (beg[l] - t_beg)) : 0u;  // (a lane without a read walks the tile's first bases, masked)
        ragged = ragged || len[l] != maxlen;
        kmers += bad ? kc_clean_windows(inv.data(), p0[l], len[l], (uint32_t)maxlen, K) : (len[l] >= (uint32_t)K ? len[l] - K + 1 : 0);
      }
      drain_and_walk(p0, len, (uint32_t)maxlen, bad ? 0 : (ragged ? 2 : 1));
    } else {
      const uint32_t per = (sd * 16u / 64u) & ~15u;
      const uint32_t ch = per < 1008u ? per : 1008u, stride = ch - K + 1;
      uint64_t nwin[64], maxwin = 0;
      for (int l = 0; l < 64; ++l) { const uint64_t n = end[l] - beg[l]; nwin[l] = n >= (uint64_t)K ? n - K + 1 : 0; maxwin = std::max(maxwin, nwin[l]); }
      const uint64_t nchunks = (maxwin + stride - 1) / stride;
      uint16_t* inv16 = reinterpret_cast<uint16_t*>(inv.data());
      for (uint64_t c = 0; c < nchunks; ++c) {
        uint32_t p0[64], clen[64], cmax = 0;
        for (int l = 0; l < 64; ++l) {
          const uint64_t cs = c * stride, n = end[l] - beg[l];
          p0[l] = (uint32_t)l * ch;
          clen[l] = cs < nwin[l] ? (uint32_t)std::min<uint64_t>(n - cs, ch) : 0u;
          cmax = std::max(cmax, clen[l]);
          const uint8_t* src = bases.data() + beg[l] + cs;
          for (uint32_t gi = 0; gi < ch / 16u; ++gi) {
            uint32_t vv[4];
            for (int q = 0; q < 4; ++q) {
              uint32_t v = 0;
              for (int bb2 = 0; bb2 < 4; ++bb2) {
                const uint32_t at = gi * 16u + (uint32_t)(q * 4 + bb2);
                v |= (at < clen[l] ? (uint32_t)src[at] : (uint32_t)'A') << (8 * bb2);
              }
              vv[q] = v;
            }
            uint32_t nb;
            const uint32_t gidx = p0[l] / 16u + gi;
            fwd[gidx] = kc_pack16(vv, nb);
            inv16[gidx ^ 1u] = (uint16_t)kc_notbase16(vv);
          }
        }
        for (int l = 0; l < 64; ++l) kmers += kc_clean_windows(inv.data(), p0[l], clen[l], cmax, K);
        if (cmax >= (uint32_t)K) drain_and_walk(p0, clen, cmax, 0);
      }
    }
  }
  for (size_t i = 0; i < table.size(); ++i) std::printf("%u\n", g_cs && ix.counts[head[i]] > g_cs ? g_cs : ix.counts[head[i]]);
  std::printf("kmers %llu runs %llu passed %llu restarts %llu\n", (unsigned long long)kmers, (unsigned long long)runs,
              (unsigned long long)passed, (unsigned long long)restarts);
}

#ifndef KC_HOST_ONLY_K  // (debug builds: one k)
#define KC_HOST_ONLY_K 0
#endif
template <uint32_t CAP, int K = (KC_HOST_ONLY_K ? KC_HOST_ONLY_K : kKcMinK)>
static bool dispatch(int k, const std::vector<std::string>& table, const std::vector<std::string>& reads, size_t lead) {
  if constexpr (K > (KC_HOST_ONLY_K ? KC_HOST_ONLY_K : kKcMaxK)) {
    return false;
  } else {
    if (k == K) { run<K, CAP>(table, reads, lead); return true; }
    return dispatch<CAP, K + 1>(k, table, reads, lead);
  }
}

int main() {
  int k = 0, ntable = 0, nreads = 0, lead = 0;
  std::string line;
  std::getline(std::cin, line);
  int cs = 0;
  if (std::sscanf(line.c_str(), "%d %d %d %d %d %d %d", &k, &g_cap, &ntable, &nreads, &lead, &cs, &g_bb) != 7) return 2;
  g_cs = (uint32_t)cs;
  if (g_bb > 0) g_gate_extra = 0;
  std::vector<std::string> table(ntable), reads(nreads);
  for (auto& t : table) std::getline(std::cin, t);
  for (auto& r : reads) std::getline(std::cin, r);
  bool ok = false;
  if (g_cap == 3) ok = dispatch<3>(k, table, reads, (size_t)lead);
  else if (g_cap == 12) ok = dispatch<12>(k, table, reads, (size_t)lead);
  if (!ok) { std::fprintf(stderr, "k or cap not instantiated\n"); return 2; }
  return 0;
}
