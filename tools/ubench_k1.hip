// ubench_k1.hip — ablation of the per-base work of k_sketch_reads<21>.
// hipcc --offload-arch=gfx950 -O3 -std=c++20 -I metalign_amd/csrc tools/ubench_k1.hip -o tools/ubench_k1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define MG_MAX_K 64
#include "mg_kmer.h"
using namespace mg;

constexpr int K = 21, READLEN = 150, TILES = 16;  // tiles per wave

// MODE 0: roll + hash + ballot/compaction   1: roll + hash   2: roll only   3: hash only (words from a cheap LCG)
template <int MODE>
__global__ __launch_bounds__(256) void kern(const uint8_t* __restrict__ bases, uint64_t hmax, uint64_t* out) {
  extern __shared__ uint8_t smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t* stage = smem + wave * (64 * READLEN + 64);
  uint64_t* cbuf = reinterpret_cast<uint64_t*>(smem + 4 * (64 * READLEN + 64)) + wave * 256;
  uint64_t acc = 0;
  int nc = 0;
  for (int t = 0; t < TILES; ++t) {
    const uint8_t* g = bases + ((size_t)(blockIdx.x * 4 + wave) * TILES + t) % 1024 * (64 * READLEN);
    for (int i = lane; i < 64 * READLEN / 16; i += 64) ((uint4*)stage)[i] = ((const uint4*)g)[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint8_t* src = stage + lane * READLEN;
    Roller<K> roll; roll.reset();
    uint64_t lcg = lane * 0x9E3779B97F4A7C15ull + t;
    for (int pos = 0; pos < READLEN; pos += 2) {
      uint32_t b0 = src[pos], b1 = src[pos + 1];
      uint32_t c0, c1;
      uint64_t h0 = 0, h1 = 0; bool f0 = false, f1 = false;
      if (MODE != 3) {
        bool ok0 = decode_base(b0, c0), ok1 = decode_base(b1, c1);
        roll.push(c0); roll.run = ok0 ? roll.run : 0;
        if (MODE != 2) h0 = roll.hash(); else h0 = roll.f[0] ^ roll.r[1] ^ roll.pf_lo;
        f0 = roll.run >= K;
        roll.push(c1); roll.run = ok1 ? roll.run : 0;
        if (MODE != 2) h1 = roll.hash(); else h1 = roll.f[2] ^ roll.r[3] ^ roll.pr_lo;
        f1 = roll.run >= K;
      } else {
        lcg = lcg * 6364136223846793005ull + b0;  // cheap, data dependent
        roll.f[0] = (uint32_t)lcg; roll.f[1] = (uint32_t)(lcg >> 32); roll.f[2] = roll.f[0] ^ 0x5555; roll.f[3] = roll.f[1] + 7; roll.f[4] = roll.f[0] + b1; roll.f[5] = b1;
        roll.pf_lo = 0; roll.pr_lo = 1;
        h0 = roll.hash();
        roll.f[0] += 0x01010101u; roll.f[3] ^= b0;
        h1 = roll.hash();
        f0 = f1 = true;
      }
      if (MODE == 0) {
        bool hit = f0 && h0 <= hmax;
        unsigned long long m = __ballot(hit);
        if (m) { if (hit) cbuf[(nc + __popcll(m & ((1ull << lane) - 1))) & 255] = h0; nc += __popcll(m); }
        hit = f1 && h1 <= hmax;
        m = __ballot(hit);
        if (m) { if (hit) cbuf[(nc + __popcll(m & ((1ull << lane) - 1))) & 255] = h1; nc += __popcll(m); }
      } else {
        acc += (f0 ? h0 : 0) ^ (f1 ? h1 : 0);
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + nc + cbuf[lane];
}

template <int MODE> void run(const char* name, const uint8_t* d_bases, uint64_t* d_out, int blocks) {
  size_t lds = 4 * (64 * READLEN + 64) + 4 * 256 * 8;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  kern<MODE><<<blocks, 256, lds>>>(d_bases, 0x0590000000000000ull, d_out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  kern<MODE><<<blocks, 256, lds>>>(d_bases, 0x0590000000000000ull, d_out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double steps_per_simd = (double)blocks * 4 * TILES * READLEN / 1024.0;  // wave-steps per SIMD
  printf("%-28s %.3f ms  -> %.1f ns per wave-step per SIMD\n", name, ms, ms * 1e6 / steps_per_simd);
}

int main() {
  const size_t nb = 1024ull * 64 * READLEN;
  std::vector<uint8_t> h(nb);
  uint64_t s = 88172645463325252ull;
  for (auto& c : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; c = "ACGT"[s & 3]; }
  uint8_t* d; hipMalloc(&d, nb); hipMemcpy(d, h.data(), nb, hipMemcpyHostToDevice);
  int blocks = 768;
  uint64_t* out; hipMalloc(&out, blocks * 256 * 8);
  run<0>("roll+hash+compaction", d, out, blocks);
  run<1>("roll+hash", d, out, blocks);
  run<2>("roll only", d, out, blocks);
  run<3>("hash only (2 per iter)", d, out, blocks);
  return 0;
}
