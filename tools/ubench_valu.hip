// ubench_valu.hip — issue cost of the integer VALU ops MurmurHash3 is made of, on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu ; run on the GPU box.
// Each kernel runs ITER x 32 dependent-free ops per lane (4 independent chains), 8 waves/SIMD on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77u, a3 = a1 * 3u;
  uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
  const uint32_t c = seed | 0x87c37b91u;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (OP == 0) {  // v_mul_lo_u32
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(c));
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a1) : "v"(c));
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a2) : "v"(c));
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a3) : "v"(c));
      } else if constexpr (OP == 1) {  // v_mul_hi_u32
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a0) : "v"(c));
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a1) : "v"(c));
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a2) : "v"(c));
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a3) : "v"(c));
      } else if constexpr (OP == 2) {  // v_mad_u64_u32
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q0) : "v"(a0), "v"(c) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q1) : "v"(a1), "v"(c) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q2) : "v"(a2), "v"(c) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q3) : "v"(a3), "v"(c) : "vcc");
      } else if constexpr (OP == 3) {  // v_mul_u32_u24
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a0) : "v"(c));
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a1) : "v"(c));
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a2) : "v"(c));
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a3) : "v"(c));
      } else if constexpr (OP == 4) {  // v_xor_b32 (full-rate reference)
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(c));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a1) : "v"(c));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a2) : "v"(c));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a3) : "v"(c));
      } else if constexpr (OP == 5) {  // v_lshl_add_u64
        asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q0) : "v"(q1));
        asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q1) : "v"(q2));
        asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q2) : "v"(q3));
        asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q3) : "v"(q0));
      } else if constexpr (OP == 6) {  // v_alignbit_b32
        asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a0) : "v"(a1));
        asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a1) : "v"(a2));
        asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a2) : "v"(a3));
        asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a3) : "v"(a0));
      } else if constexpr (OP == 7) {  // v_add3_u32
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(c));
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a1) : "v"(a2), "v"(c));
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "v"(c));
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a3) : "v"(a0), "v"(c));
      } else if constexpr (OP == 8) {  // v_mad_u32_u24
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(a1));
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a1) : "v"(c), "v"(a2));
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a2) : "v"(c), "v"(a3));
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a3) : "v"(c), "v"(a0));
      } else if constexpr (OP == 9) {  // v_lshlrev_b64
        asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q0));
        asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q1));
        asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q2));
        asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q3));
      } else if constexpr (OP == 10) {  // v_mul_hi_u32_u24
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a0) : "v"(c));
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a1) : "v"(c));
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a2) : "v"(c));
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a3) : "v"(c));
      } else if constexpr (OP == 11) {  // dependent chain v_mul_lo_u32 (latency)
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(c));
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(c));
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(c));
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(c));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3) >> 32);
}

template <int OP>
void run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block == 1 wave/SIMD per block)
  uint32_t* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(out, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<OP><<<blocks, 256>>>(out, 2);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops_per_wave = (double)ITER * 32;
  double wave_instr_per_simd = ops_per_wave * waves_per_simd;  // each SIMD hosts waves_per_simd waves
  double ns_per_instr = ms * 1e6 / wave_instr_per_simd;
  printf("%-22s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instruction per SIMD (= %.1f cycles @2.4GHz)\n", name,
         waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<4>("v_xor_b32", w);
    run<0>("v_mul_lo_u32", w);
    run<1>("v_mul_hi_u32", w);
    run<2>("v_mad_u64_u32", w);
    run<3>("v_mul_u32_u24", w);
    run<8>("v_mad_u32_u24", w);
    run<10>("v_mul_hi_u32_u24", w);
    run<5>("v_lshl_add_u64", w);
    run<9>("v_lshlrev_b64", w);
    run<6>("v_alignbit_b32", w);
    run<7>("v_add3_u32", w);
    run<11>("v_mul_lo_u32 dep", w);
  }
  return 0;
}
