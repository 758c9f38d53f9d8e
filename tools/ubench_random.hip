// The part's rate for RANDOM 16-byte reads out of a table much larger than its caches (what a candidate of stage A
// costs in the dense regime): G loads/s, for several table sizes and loads in flight per lane.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_random.hip -o tools/ubench_random.bin && tools/ubench_random.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
  return x;
}

template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_random(const uint4* __restrict__ tab, uint64_t nslots, uint32_t rounds, uint32_t* out) {
  uint64_t s = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
  uint32_t acc = 0;
  for (uint32_t r = 0; r < rounds; ++r) {
    uint4 v[INFLIGHT];
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) {
      s = mix(s + j);
      v[j] = tab[(uint64_t)(((unsigned __int128)s * nslots) >> 64)];
    }
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) acc ^= v[j].x + v[j].z;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// LANES consecutive lanes share ONE random 128-byte line (16 bytes each at consecutive offsets): LANES = 1 is a random
// 16-byte load per lane (64 lines per wave instruction), 4 a random 64-byte half line per four lanes, 8 a whole line
// per eight lanes.  If the memory side moves whole 128-byte lines whatever is asked of them, lines/s is the same for
// all three; if a 16-byte load moves only its 32- or 64-byte sector, the wider forms run at a fraction of it.
template <int LANES>
__global__ __launch_bounds__(256) void k_random_line(const uint4* __restrict__ tab, uint64_t nlines, uint32_t rounds, uint32_t* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t s = (uint64_t)(t / LANES) * 0x9E3779B97F4A7C15ull + 1;
  uint32_t acc = 0;
  for (uint32_t r = 0; r < rounds; ++r) {
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s = mix(s + j);
      const uint64_t line = (uint64_t)(((unsigned __int128)s * nlines) >> 64);
      v[j] = tab[line * 8 + (LANES == 1 ? (s & 7) : (t % LANES))];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc ^= v[j].x + v[j].z;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int LANES>
static void run_line(const uint4* tab, uint64_t nslots, uint32_t* out, int cus) {
  const uint32_t rounds = 256;
  const int grid = cus * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_random_line<LANES>, dim3(grid), dim3(256), 0, 0, tab, nslots / 8, rounds, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double lines = (double)grid * 256 * rounds * 4 / LANES;
    if (rep == 2)
      printf("table %6.0f MB, %d lanes per random line: %.1f G lines/s, %.0f GB/s asked for (%.2f ms for %.0f M lines)\n",
             nslots * 16 / 1e6, LANES, lines / ms / 1e6, lines * LANES * 16 / ms / 1e6, ms, lines / 1e6);
  }
}

template <int INFLIGHT>
static void run(const uint4* tab, uint64_t nslots, uint32_t* out, int cus) {
  const uint32_t rounds = 256 / INFLIGHT * 4;
  const int grid = cus * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_random<INFLIGHT>, dim3(grid), dim3(256), 0, 0, tab, nslots, rounds, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double loads = (double)grid * 256 * rounds * INFLIGHT;
    if (rep == 2)
      printf("table %6.0f MB, %d loads in flight per lane: %.1f G loads/s (%.2f ms for %.0f M loads)\n",
             nslots * 16 / 1e6, INFLIGHT, loads / ms / 1e6, ms, loads / 1e6);
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  uint32_t* out;
  hipMalloc(&out, 64);
  if (argc > 1) {  // "lines": the sector question above, one table size (for a --pmc pass: one kernel name per form)
    const uint64_t nslots = 4096ull * 1000000ull / 16;
    uint4* tab;
    hipMalloc(&tab, nslots * 16);
    hipMemset(tab, 1, nslots * 16);
    run_line<1>(tab, nslots, out, p.multiProcessorCount);
    run_line<2>(tab, nslots, out, p.multiProcessorCount);
    run_line<4>(tab, nslots, out, p.multiProcessorCount);
    run_line<8>(tab, nslots, out, p.multiProcessorCount);
    hipFree(tab);
    return 0;
  }
  for (uint64_t mb : {64ull, 256ull, 1024ull, 4096ull}) {
    const uint64_t nslots = mb * 1000000ull / 16;
    uint4* tab;
    hipMalloc(&tab, nslots * 16);
    hipMemset(tab, 1, nslots * 16);
    run<1>(tab, nslots, out, p.multiProcessorCount);
    run<4>(tab, nslots, out, p.multiProcessorCount);
    run<8>(tab, nslots, out, p.multiProcessorCount);
    hipFree(tab);
  }
  return 0;
}
