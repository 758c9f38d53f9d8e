// How fast does a 16-bit (two-digit) partition of N 64-bit hashes run?  (rocPRIM radix sort restricted to the top bits
// of the hash range) — the cost a sort-based stage A would pay per k in the dense regime.  See tools/experiments/README.md.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_sort.hip -o /tmp/ubench_sort && /tmp/ubench_sort [N millions] [bits]
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void fill(uint64_t* p, uint64_t n, uint64_t hmax) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
    p[i] = x % hmax;
  }
}

int main(int argc, char** argv) {
  const uint64_t n = (uint64_t)(argc > 1 ? atof(argv[1]) : 325.0) * 1000000ull;
  const unsigned bits = argc > 2 ? (unsigned)atoi(argv[2]) : 16u;
  const uint64_t hmax = (uint64_t)(0.2 * 18446744073709551616.0);
  unsigned top = 64;
  while (top > 1 && !((hmax >> (top - 1)) & 1ull)) --top;
  uint64_t *a, *b;
  hipMalloc(&a, n * 8);
  hipMalloc(&b, n * 8);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, a, n, hmax);
  size_t tmp = 0;
  rocprim::radix_sort_keys(nullptr, tmp, a, b, n, top - bits, top, 0);
  void* t;
  hipMalloc(&t, tmp);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    rocprim::radix_sort_keys(t, tmp, a, b, n, top - bits, top, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("n = %llu M keys, bits [%u, %u): %.3f ms  (%.1f GB/s of keys in+out per digit pass pair)\n",
           (unsigned long long)(n / 1000000ull), top - bits, top, ms, (double)n * 8 * 2 * ((bits + 7) / 8) / ms / 1e6);
  }
  return 0;
}
