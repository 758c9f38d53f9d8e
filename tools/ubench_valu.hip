// ubench_valu.hip — issue cost on gfx950 of every VALU opcode in the hot loop of k_sketch_reads_multi (stage A), one
// opcode per kernel, in the operand forms the compiler emits there (SGPR / literal second sources, VOP3 carry-out pairs).
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu ; run on the GPU box:
//     tools/ubench_valu [json-out]
// Each kernel runs ITER x 32 instructions per lane in 4 independent chains, W waves per SIMD on every CU
// (W = 1, 2, 3, 4, 8; the fused kernel runs at 3).  Reported: ns per wave-instruction per SIMD, and cycles at the clock
// measured in the same launch (s_memtime ticks / s_memrealtime at 100 MHz).  tools/valu_roofline.py multiplies these by the
// opcode counts of the hot loop (tools/isa_histogram.py) to price the kernel's VALU roofline.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define ITER 8192


enum Op {
  XOR_E32, OR_E32, AND_E32, ADD_U32, LSHR_B32, LSHL_B32, MOV_B32, MOV_B64, BCNT, BITOP3,
  MUL_LO_S, MUL_HI_S, MAD64_S0, MAD64_ACC, MAD64_5,
  ADD3, ALIGNBYTE, ALIGNBIT, PERM, BFI, LSHL_OR, LSHL_ADD_U32, BFE, AND_OR, MUL_U24, MAD_U24,
  CNDMASK_VCC, CNDMASK_SGPR, CMP_U32, CMP_U64, ADDC_PAIR,
  LSHL_B64, LSHR_B64, LSHL_ADD_U64,
  MUL64_2MUL, MUL64_3MAD,
  XOR_VV, ADD_VV, MUL_LO_V, LSHL1, LSHR31, CND_E32_CMP, CND_E32_SET, CND_E64_VCC, SUB_U32, XOR_LIT, CND_E32_INDEP, CND_E32_SRC0CONST, NOPS
};

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint64_t* clk, uint32_t seed, uint32_t sc, uint32_t sd) {
  uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77u, a3 = a1 * 3u;
  uint64_t q0 = a0 | ((uint64_t)a1 << 32), q1 = a1 | ((uint64_t)a2 << 32), q2 = a2 | ((uint64_t)a3 << 32), q3 = a3 | ((uint64_t)a0 << 32);
  uint32_t b0 = a0 + 1, b1 = a1 + 2, b2 = a2 + 3, b3 = a3 + 4;
  const uint32_t c = __builtin_amdgcn_readfirstlane(sc | 0x87c37b91u);  // SGPR operands
  const uint32_t d = __builtin_amdgcn_readfirstlane(sd | 0x114253d5u);
  asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n\ts_mov_b64 s[4:5], vcc\n\ts_mov_b64 s[6:7], 0x7fffffff" : : "v"(b0), "v"(b1) : "vcc", "s4", "s5", "s6", "s7");
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (OP == XOR_E32) {
        asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a0) : "s"(c)); asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a1) : "s"(c));
        asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a2) : "s"(c)); asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a3) : "s"(c));
      } else if constexpr (OP == OR_E32) {
        asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(a1) : "v"(b1));
        asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(a2) : "v"(b2)); asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(a3) : "v"(b3));
      } else if constexpr (OP == AND_E32) {
        asm volatile("v_and_b32_e32 %0, 0x3ff0ff3f, %0" : "+v"(a0)); asm volatile("v_and_b32_e32 %0, 0x3ff0ff3f, %0" : "+v"(a1));
        asm volatile("v_and_b32_e32 %0, 0x3ff0ff3f, %0" : "+v"(a2)); asm volatile("v_and_b32_e32 %0, 0x3ff0ff3f, %0" : "+v"(a3));
      } else if constexpr (OP == ADD_U32) {
        asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a0) : "s"(c)); asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a1) : "s"(c));
        asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a2) : "s"(c)); asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a3) : "s"(c));
      } else if constexpr (OP == LSHR_B32) {
        asm volatile("v_lshrrev_b32_e32 %0, 1, %1" : "=v"(a0) : "v"(b0)); asm volatile("v_lshrrev_b32_e32 %0, 1, %1" : "=v"(a1) : "v"(b1));
        asm volatile("v_lshrrev_b32_e32 %0, 1, %1" : "=v"(a2) : "v"(b2)); asm volatile("v_lshrrev_b32_e32 %0, 1, %1" : "=v"(a3) : "v"(b3));
      } else if constexpr (OP == LSHL_B32) {
        asm volatile("v_lshlrev_b32_e32 %0, 31, %1" : "=v"(a0) : "v"(b0)); asm volatile("v_lshlrev_b32_e32 %0, 31, %1" : "=v"(a1) : "v"(b1));
        asm volatile("v_lshlrev_b32_e32 %0, 31, %1" : "=v"(a2) : "v"(b2)); asm volatile("v_lshlrev_b32_e32 %0, 31, %1" : "=v"(a3) : "v"(b3));
      } else if constexpr (OP == MOV_B32) {
        asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a0) : "v"(b0)); asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a1) : "v"(b1));
        asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a2) : "v"(b2)); asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a3) : "v"(b3));
      } else if constexpr (OP == MOV_B64) {
        asm volatile("v_mov_b64_e32 %0, %1" : "=v"(q0) : "v"(q1)); asm volatile("v_mov_b64_e32 %0, %1" : "=v"(q1) : "v"(q2));
        asm volatile("v_mov_b64_e32 %0, %1" : "=v"(q2) : "v"(q3)); asm volatile("v_mov_b64_e32 %0, %1" : "=v"(q3) : "v"(q0));
      } else if constexpr (OP == BCNT) {
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a0) : "v"(b0)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a1) : "v"(b1));
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a2) : "v"(b2)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a3) : "v"(b3));
      } else if constexpr (OP == BITOP3) {
        asm volatile("v_bitop3_b32 %0, %0, 3, %1 bitop3:0xc" : "+v"(a0) : "v"(b0)); asm volatile("v_bitop3_b32 %0, %0, 3, %1 bitop3:0xc" : "+v"(a1) : "v"(b1));
        asm volatile("v_bitop3_b32 %0, %0, 3, %1 bitop3:0xc" : "+v"(a2) : "v"(b2)); asm volatile("v_bitop3_b32 %0, %0, 3, %1 bitop3:0xc" : "+v"(a3) : "v"(b3));
      } else if constexpr (OP == MUL_LO_S) {
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "s"(c)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a1) : "s"(c));
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a2) : "s"(c)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a3) : "s"(c));
      } else if constexpr (OP == MUL_HI_S) {
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a0) : "s"(c)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a1) : "s"(c));
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a2) : "s"(c)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a3) : "s"(c));
      } else if constexpr (OP == MAD64_S0) {  // v_mad_u64_u32 v[..], s[..], v, s, 0  (the low x low product)
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, 0" : "=v"(q0) : "v"(a0), "s"(c) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, 0" : "=v"(q1) : "v"(a1), "s"(c) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, 0" : "=v"(q2) : "v"(a2), "s"(c) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, 0" : "=v"(q3) : "v"(a3), "s"(c) : "s2", "s3");
      } else if constexpr (OP == MAD64_ACC) {  // with a 64-bit VGPR addend
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(q0) : "v"(a0), "s"(c) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(q1) : "v"(a1), "s"(c) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(q2) : "v"(a2), "s"(c) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(q3) : "v"(a3), "s"(c) : "s2", "s3");
      } else if constexpr (OP == MAD64_5) {  // h * 5 + c
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, 5, %0" : "+v"(q0) : "v"(a0) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, 5, %0" : "+v"(q1) : "v"(a1) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, 5, %0" : "+v"(q2) : "v"(a2) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, 5, %0" : "+v"(q3) : "v"(a3) : "s2", "s3");
      } else if constexpr (OP == ADD3) {
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a0) : "v"(b1), "v"(b2)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a1) : "v"(b2), "v"(b3));
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a2) : "v"(b3), "v"(b0)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a3) : "v"(b0), "v"(b1));
      } else if constexpr (OP == ALIGNBYTE) {
        asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(a0) : "v"(b1)); asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(a1) : "v"(b2));
        asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(a2) : "v"(b3)); asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(a3) : "v"(b0));
      } else if constexpr (OP == ALIGNBIT) {
        asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a0) : "v"(b1)); asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a1) : "v"(b2));
        asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a2) : "v"(b3)); asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a3) : "v"(b0));
      } else if constexpr (OP == PERM) {
        asm volatile("v_perm_b32 %0, 0, %1, %0" : "+v"(a0) : "s"(c)); asm volatile("v_perm_b32 %0, 0, %1, %0" : "+v"(a1) : "s"(c));
        asm volatile("v_perm_b32 %0, 0, %1, %0" : "+v"(a2) : "s"(c)); asm volatile("v_perm_b32 %0, 0, %1, %0" : "+v"(a3) : "s"(c));
      } else if constexpr (OP == BFI) {
        asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a0) : "v"(b1), "v"(b2)); asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a1) : "v"(b2), "v"(b3));
        asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a2) : "v"(b3), "v"(b0)); asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a3) : "v"(b0), "v"(b1));
      } else if constexpr (OP == LSHL_OR) {
        asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a0) : "v"(b1)); asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a1) : "v"(b2));
        asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a2) : "v"(b3)); asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a3) : "v"(b0));
      } else if constexpr (OP == LSHL_ADD_U32) {
        asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a0) : "v"(b1)); asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a1) : "v"(b2));
        asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a2) : "v"(b3)); asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a3) : "v"(b0));
      } else if constexpr (OP == BFE) {
        asm volatile("v_bfe_u32 %0, %1, 4, 2" : "=v"(a0) : "v"(b0)); asm volatile("v_bfe_u32 %0, %1, 4, 2" : "=v"(a1) : "v"(b1));
        asm volatile("v_bfe_u32 %0, %1, 4, 2" : "=v"(a2) : "v"(b2)); asm volatile("v_bfe_u32 %0, %1, 4, 2" : "=v"(a3) : "v"(b3));
      } else if constexpr (OP == AND_OR) {
        asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a0) : "s"(c), "v"(b1)); asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a1) : "s"(c), "v"(b2));
        asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a2) : "s"(c), "v"(b3)); asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a3) : "s"(c), "v"(b0));
      } else if constexpr (OP == MUL_U24) {
        asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(a0) : "s"(c)); asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(a1) : "s"(c));
        asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(a2) : "s"(c)); asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(a3) : "s"(c));
      } else if constexpr (OP == MAD_U24) {
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a0) : "s"(c), "v"(b1)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a1) : "s"(c), "v"(b2));
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a2) : "s"(c), "v"(b3)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a3) : "s"(c), "v"(b0));
      } else if constexpr (OP == CNDMASK_VCC) {
        asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b1) : ); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b2));
        asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b3)); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b0));
      } else if constexpr (OP == CNDMASK_SGPR) {
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[4:5]" : "+v"(a0) : "v"(b1) : ); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[4:5]" : "+v"(a1) : "v"(b2));
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[4:5]" : "+v"(a2) : "v"(b3)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[4:5]" : "+v"(a3) : "v"(b0));
      } else if constexpr (OP == CMP_U32) {
        asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "s"(c), "v"(a0) : "vcc"); asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "s"(c), "v"(a1) : "vcc");
        asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "s"(c), "v"(a2) : "vcc"); asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "s"(c), "v"(a3) : "vcc");
      } else if constexpr (OP == CMP_U64) {
        asm volatile("v_cmp_ge_u64_e64 s[2:3], s[6:7], %0" : : "v"(q0) : "s2", "s3"); asm volatile("v_cmp_ge_u64_e64 s[2:3], s[6:7], %0" : : "v"(q1) : "s2", "s3");
        asm volatile("v_cmp_ge_u64_e64 s[2:3], s[6:7], %0" : : "v"(q2) : "s2", "s3"); asm volatile("v_cmp_ge_u64_e64 s[2:3], s[6:7], %0" : : "v"(q3) : "s2", "s3");
      } else if constexpr (OP == ADDC_PAIR) {  // a 64-bit add as the carry pair (two instructions per "op": reported per instruction)
        asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a0) : "v"(b1) : "vcc"); asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(a1) : "v"(b2) : "vcc");
        asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a2) : "v"(b3) : "vcc"); asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(a3) : "v"(b0) : "vcc");
      } else if constexpr (OP == LSHL_B64) {
        asm volatile("v_lshlrev_b64 %0, 27, %0" : "+v"(q0)); asm volatile("v_lshlrev_b64 %0, 27, %0" : "+v"(q1));
        asm volatile("v_lshlrev_b64 %0, 27, %0" : "+v"(q2)); asm volatile("v_lshlrev_b64 %0, 27, %0" : "+v"(q3));
      } else if constexpr (OP == LSHR_B64) {
        asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(q0)); asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(q1));
        asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(q2)); asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(q3));
      } else if constexpr (OP == LSHL_ADD_U64) {
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q0) : "v"(q1)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q1) : "v"(q2));
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q2) : "v"(q3)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q3) : "v"(q0));
      } else if constexpr (OP == XOR_VV) {
        asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a0) : "v"(b0)); asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a1) : "v"(b1));
        asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a2) : "v"(b2)); asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a3) : "v"(b3));
      } else if constexpr (OP == XOR_LIT) {
        asm volatile("v_xor_b32_e32 %0, 21, %0" : "+v"(a0)); asm volatile("v_xor_b32_e32 %0, 21, %0" : "+v"(a1));
        asm volatile("v_xor_b32_e32 %0, 21, %0" : "+v"(a2)); asm volatile("v_xor_b32_e32 %0, 21, %0" : "+v"(a3));
      } else if constexpr (OP == ADD_VV) {
        asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a0) : "v"(b0)); asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a1) : "v"(b1));
        asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a2) : "v"(b2)); asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a3) : "v"(b3));
      } else if constexpr (OP == SUB_U32) {
        asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a1) : "v"(b1));
        asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a2) : "v"(b2)); asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a3) : "v"(b3));
      } else if constexpr (OP == MUL_LO_V) {
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a1) : "v"(b1));
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a2) : "v"(b2)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a3) : "v"(b3));
      } else if constexpr (OP == LSHL1) {
        asm volatile("v_lshlrev_b32_e32 %0, 1, %1" : "=v"(a0) : "v"(b0)); asm volatile("v_lshlrev_b32_e32 %0, 1, %1" : "=v"(a1) : "v"(b1));
        asm volatile("v_lshlrev_b32_e32 %0, 1, %1" : "=v"(a2) : "v"(b2)); asm volatile("v_lshlrev_b32_e32 %0, 1, %1" : "=v"(a3) : "v"(b3));
      } else if constexpr (OP == LSHR31) {
        asm volatile("v_lshrrev_b32_e32 %0, 31, %1" : "=v"(a0) : "v"(b0)); asm volatile("v_lshrrev_b32_e32 %0, 31, %1" : "=v"(a1) : "v"(b1));
        asm volatile("v_lshrrev_b32_e32 %0, 31, %1" : "=v"(a2) : "v"(b2)); asm volatile("v_lshrrev_b32_e32 %0, 31, %1" : "=v"(a3) : "v"(b3));
      } else if constexpr (OP == CND_E32_CMP) {  // the form of the hot loop: one compare into VCC, then a run of selects (5 instructions: reported per 5)
        asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(b0), "v"(b1) : "vcc");
        asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b1)); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b2));
        asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b3)); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b0));
      } else if constexpr (OP == CND_E32_SET) {  // VCC written once before the loop (defined), selects only
        asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b1)); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b2));
        asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b3)); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b0));
      } else if constexpr (OP == CND_E64_VCC) {
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a0) : "v"(b1)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a1) : "v"(b2));
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a2) : "v"(b3)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a3) : "v"(b0));
      } else if constexpr (OP == CND_E32_INDEP) {  // no dependence between consecutive selects
        asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a0) : "v"(b0), "v"(b1)); asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a1) : "v"(b1), "v"(b2));
        asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a2) : "v"(b2), "v"(b3)); asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a3) : "v"(b3), "v"(b0));
      } else if constexpr (OP == CND_E32_SRC0CONST) {
        asm volatile("v_cndmask_b32_e32 %0, 0, %1, vcc" : "=v"(a0) : "v"(b0)); asm volatile("v_cndmask_b32_e32 %0, 0, %1, vcc" : "=v"(a1) : "v"(b1));
        asm volatile("v_cndmask_b32_e32 %0, 0, %1, vcc" : "=v"(a2) : "v"(b2)); asm volatile("v_cndmask_b32_e32 %0, 0, %1, vcc" : "=v"(a3) : "v"(b3));
      } else if constexpr (OP == MUL64_2MUL) {  // a 64-bit x 64-bit -> low 64 as the compiler emits it: 4 instructions, 2 chains x 2
        uint32_t t, u;
        asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(t) : "v"(a1), "s"(c));
        asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u) : "v"(a0), "s"(d));
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, 0" : "=v"(q0) : "v"(a0), "s"(c) : "s2", "s3");
        asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a1) : "v"((uint32_t)(q0 >> 32)), "v"(u), "v"(t));
        a0 = (uint32_t)q0;
        asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(t) : "v"(a3), "s"(c));
        asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u) : "v"(a2), "s"(d));
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, 0" : "=v"(q2) : "v"(a2), "s"(c) : "s2", "s3");
        asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a3) : "v"((uint32_t)(q2 >> 32)), "v"(u), "v"(t));
        a2 = (uint32_t)q2;
      } else if constexpr (OP == MUL64_3MAD) {  // the same product as three v_mad_u64_u32: 3 instructions, 2 chains x 2 (reported per 4!)
        uint64_t t;
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, 0" : "=v"(q0) : "v"(a0), "s"(c) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %3" : "=v"(t) : "v"(a0), "s"(d), "v"(q0 >> 32) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %3" : "=v"(t) : "v"(a1), "s"(c), "v"(t) : "s2", "s3");
        a0 = (uint32_t)q0; a1 = (uint32_t)t;
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, 0" : "=v"(q2) : "v"(a2), "s"(c) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %3" : "=v"(t) : "v"(a2), "s"(d), "v"(q2 >> 32) : "s2", "s3");
        asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %3" : "=v"(t) : "v"(a3), "s"(c), "v"(t) : "s2", "s3");
        a2 = (uint32_t)q2; a3 = (uint32_t)t;
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3) >> 32);
}

struct Row { std::string name; int waves; double ms, ns, cyc, ghz; };
static std::vector<Row> rows;

template <int OP>
void run(const char* name, int waves_per_simd, double instr_per_group = 32.0) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block == 1 wave/SIMD per block)
  uint32_t* out;
  uint64_t* clk;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&clk, 16);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) k<OP><<<blocks, 256>>>(out, clk, 1, 0, 0);  // clocks up
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<OP><<<blocks, 256>>>(out, clk, 2, 0, 0);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  uint64_t h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);  // memrealtime ticks at 100 MHz
  const double per_simd = (double)ITER * instr_per_group * waves_per_simd;
  const double ns = ms * 1e6 / per_simd;
  rows.push_back({name, waves_per_simd, ms, ns, ns * ghz, ghz});
  printf("%-34s waves/SIMD=%d  %8.3f ms  %6.3f ns per wave-instruction per SIMD = %5.2f cycles at %.2f GHz\n", name, waves_per_simd, ms, ns,
         ns * ghz, ghz);
  hipFree(out); hipFree(clk);
}

int main(int argc, char** argv) {
  for (int w : {3, 1, 2, 4, 8}) {
    run<XOR_E32>("v_xor_b32_e32 (sgpr src)", w);
    run<OR_E32>("v_or_b32_e32", w);
    run<AND_E32>("v_and_b32_e32 (literal)", w);
    run<ADD_U32>("v_add_u32_e32", w);
    run<LSHR_B32>("v_lshrrev_b32_e32", w);
    run<LSHL_B32>("v_lshlrev_b32_e32", w);
    run<MOV_B32>("v_mov_b32_e32", w);
    run<MOV_B64>("v_mov_b64_e32", w);
    run<BCNT>("v_bcnt_u32_b32", w);
    run<BITOP3>("v_bitop3_b32", w);
    run<MUL_LO_S>("v_mul_lo_u32 (sgpr src)", w);
    run<MUL_HI_S>("v_mul_hi_u32 (sgpr src)", w);
    run<MAD64_S0>("v_mad_u64_u32 (v, s, 0)", w);
    run<MAD64_ACC>("v_mad_u64_u32 (v, s, v[2])", w);
    run<MAD64_5>("v_mad_u64_u32 (v, 5, v[2])", w);
    run<ADD3>("v_add3_u32", w);
    run<ALIGNBYTE>("v_alignbyte_b32", w);
    run<ALIGNBIT>("v_alignbit_b32", w);
    run<PERM>("v_perm_b32", w);
    run<BFI>("v_bfi_b32", w);
    run<LSHL_OR>("v_lshl_or_b32", w);
    run<LSHL_ADD_U32>("v_lshl_add_u32", w);
    run<BFE>("v_bfe_u32", w);
    run<AND_OR>("v_and_or_b32", w);
    run<MUL_U24>("v_mul_u32_u24_e32", w);
    run<MAD_U24>("v_mad_u32_u24", w);
    run<CNDMASK_VCC>("v_cndmask_b32_e32 (vcc)", w);
    run<CNDMASK_SGPR>("v_cndmask_b32_e64 (sgpr pair)", w);
    run<CMP_U32>("v_cmp_lt_u32_e32", w);
    run<CMP_U64>("v_cmp_ge_u64_e64", w);
    run<ADDC_PAIR>("v_add_co_u32 + v_addc_co_u32", w);
    run<LSHL_B64>("v_lshlrev_b64", w);
    run<LSHR_B64>("v_lshrrev_b64", w);
    run<LSHL_ADD_U64>("v_lshl_add_u64", w);
    run<XOR_VV>("v_xor_b32_e32 (vgpr src)", w);
    run<XOR_LIT>("v_xor_b32_e32 (inline const)", w);
    run<ADD_VV>("v_add_u32_e32 (vgpr src)", w);
    run<SUB_U32>("v_sub_u32_e32 (vgpr src)", w);
    run<MUL_LO_V>("v_mul_lo_u32 (vgpr src)", w);
    run<LSHL1>("v_lshlrev_b32_e32 by 1", w);
    run<LSHR31>("v_lshrrev_b32_e32 by 31", w);
    run<CND_E32_CMP>("v_cmp_lt_u32 + 4 v_cndmask_e32 /5", w, 40.0);
    run<CND_E32_SET>("v_cndmask_b32_e32 (vcc set before loop)", w);
    run<CND_E64_VCC>("v_cndmask_b32_e64 (vcc)", w);
    run<CND_E32_INDEP>("v_cndmask_b32_e32 (independent)", w);
    run<CND_E32_SRC0CONST>("v_cndmask_b32_e32 (0, v, vcc)", w);
    run<MUL64_2MUL>("mul64: 2 mul_lo + mad64 + add3 /4", w, 64.0);
    run<MUL64_3MAD>("mul64: 3 mad64 /3", w, 48.0);
  }
  if (argc > 1) {
    FILE* f = fopen(argv[1], "w");
    fprintf(f, "{\"iter\": %d, \"note\": \"ns and cycles per wave-instruction per SIMD; W waves per SIMD on all 256 CUs; 4 independent chains\", \"rows\": [\n", ITER);
    for (size_t i = 0; i < rows.size(); ++i)
      fprintf(f, " {\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ns\": %.4f, \"cycles\": %.3f, \"ghz\": %.3f}%s\n", rows[i].name.c_str(),
              rows[i].waves, rows[i].ms, rows[i].ns, rows[i].cyc, rows[i].ghz, i + 1 < rows.size() ? "," : "");
    fprintf(f, "]}\n");
    fclose(f);
  }
  return 0;
}
