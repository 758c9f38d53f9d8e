// mg_sort.hip — sort / run-length / reduce-by-key / scan services.
// These are plain library operations on small derived lists (sketch candidates,
// multimapped offsets), delegated to rocPRIM; the hot kernels are hand-written
// in mg_sketch.hip, mg_contain.hip and mg_profile.hip.
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "mg_internal.h"

namespace mg {

namespace {

struct sat_add_u32 {
  __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const {
    uint32_t s = a + b;
    return s < a ? 0xffffffffu : s;
  }
};

__global__ void k_widen_u32(const uint32_t* in, uint64_t* out, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

}  // namespace

int sort_keys(const uint64_t* d_in, uint64_t* d_out, uint64_t n, unsigned end_bit) {
  if (n == 0) return MG_OK;
  if (end_bit == 0 || end_bit > 64) end_bit = 64;
  hipStream_t st = ctx().stream;
  size_t tmp = 0;
  MG_HIP(rocprim::radix_sort_keys(nullptr, tmp, d_in, d_out, n, 0u, end_bit, st));
  void* t = scratch("sort_tmp", tmp);
  if (!t) return MG_ERR_NOMEM;
  MG_HIP(rocprim::radix_sort_keys(t, tmp, d_in, d_out, n, 0u, end_bit, st));
  return MG_OK;
}

int sort_keys_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t n) {
  if (n == 0) return MG_OK;
  hipStream_t st = ctx().stream;
  size_t tmp = 0;
  MG_HIP(rocprim::radix_sort_keys(nullptr, tmp, d_in, d_out, n, 0u, 32u, st));
  void* t = scratch("sort_tmp", tmp);
  if (!t) return MG_ERR_NOMEM;
  MG_HIP(rocprim::radix_sort_keys(t, tmp, d_in, d_out, n, 0u, 32u, st));
  return MG_OK;
}

int sort_pairs(const uint64_t* d_kin, uint64_t* d_kout, const uint32_t* d_vin, uint32_t* d_vout, uint64_t n) {
  if (n == 0) return MG_OK;
  hipStream_t st = ctx().stream;
  size_t tmp = 0;
  MG_HIP(rocprim::radix_sort_pairs(nullptr, tmp, d_kin, d_kout, d_vin, d_vout, n, 0u, 64u, st));
  void* t = scratch("sort_tmp", tmp);
  if (!t) return MG_ERR_NOMEM;
  MG_HIP(rocprim::radix_sort_pairs(t, tmp, d_kin, d_kout, d_vin, d_vout, n, 0u, 64u, st));
  return MG_OK;
}

int rle_keys(const uint64_t* d_sorted, uint64_t n, uint64_t* d_unique, uint32_t* d_counts, uint64_t* d_runs) {
  hipStream_t st = ctx().stream;
  if (n == 0) { MG_HIP(hipMemsetAsync(d_runs, 0, sizeof(uint64_t), st)); return MG_OK; }
  if (n > 0xffffffffull) return fail(MG_ERR_ARG, "candidate list of %llu entries exceeds 2^32-1", (unsigned long long)n);
  size_t tmp = 0;
  MG_HIP(rocprim::run_length_encode(nullptr, tmp, d_sorted, (unsigned)n, d_unique, d_counts, d_runs, st));
  void* t = scratch("rle_tmp", tmp);
  if (!t) return MG_ERR_NOMEM;
  MG_HIP(rocprim::run_length_encode(t, tmp, d_sorted, (unsigned)n, d_unique, d_counts, d_runs, st));
  return MG_OK;
}

int reduce_pairs(const uint64_t* d_keys, const uint32_t* d_vals, uint64_t n, uint64_t* d_unique, uint32_t* d_sums,
                 uint64_t* d_runs) {
  hipStream_t st = ctx().stream;
  if (n == 0) { MG_HIP(hipMemsetAsync(d_runs, 0, sizeof(uint64_t), st)); return MG_OK; }
  size_t tmp = 0;
  MG_HIP(rocprim::reduce_by_key(nullptr, tmp, d_keys, d_vals, (size_t)n, d_unique, d_sums, d_runs, sat_add_u32(),
                                rocprim::equal_to<uint64_t>(), st));
  void* t = scratch("rle_tmp", tmp);
  if (!t) return MG_ERR_NOMEM;
  MG_HIP(rocprim::reduce_by_key(t, tmp, d_keys, d_vals, (size_t)n, d_unique, d_sums, d_runs, sat_add_u32(),
                                rocprim::equal_to<uint64_t>(), st));
  return MG_OK;
}

int segmented_sort_keys(const uint64_t* d_in, uint64_t* d_out, uint64_t n, const uint64_t* d_offsets, uint64_t nseg) {
  if (n == 0 || nseg == 0) return MG_OK;
  if (n > 0xffffffffull || nseg > 0xffffffffull) return fail(MG_ERR_ARG, "segmented sort batch too large");
  hipStream_t st = ctx().stream;
  size_t tmp = 0;
  MG_HIP(rocprim::segmented_radix_sort_keys(nullptr, tmp, d_in, d_out, (unsigned)n, (unsigned)nseg, d_offsets,
                                            d_offsets + 1, 0u, 64u, st));
  void* t = scratch("sort_tmp", tmp);
  if (!t) return MG_ERR_NOMEM;
  MG_HIP(rocprim::segmented_radix_sort_keys(t, tmp, d_in, d_out, (unsigned)n, (unsigned)nseg, d_offsets,
                                            d_offsets + 1, 0u, 64u, st));
  return MG_OK;
}

int exclusive_sum_u32_to_u64(const uint32_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* h_total) {
  *h_total = 0;
  if (n == 0) return MG_OK;
  hipStream_t st = ctx().stream;
  // widen first so the scan accumulates in 64 bits, then scan in place (n+1 outputs: last = total)
  uint64_t* d_wide = (uint64_t*)scratch("scan_wide", (n + 1) * sizeof(uint64_t));
  if (!d_wide) return MG_ERR_NOMEM;
  hipLaunchKernelGGL(k_widen_u32, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, d_in, d_wide, n);
  MG_HIP(hipMemsetAsync(d_wide + n, 0, sizeof(uint64_t), st));
  size_t tmp = 0;
  MG_HIP(rocprim::exclusive_scan(nullptr, tmp, d_wide, d_out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), st));
  void* t = scratch("scan_tmp", tmp);
  if (!t) return MG_ERR_NOMEM;
  MG_HIP(rocprim::exclusive_scan(t, tmp, d_wide, d_out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), st));
  uint64_t* pin = host_words();
  MG_HIP(hipMemcpyAsync(pin + 8, d_out + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MG_HIP(hipStreamSynchronize(st));
  *h_total = pin[8];
  return MG_OK;
}

}  // namespace mg
