// Micro-benchmark: issue rate of the integer VALU instructions the scan kernel is made of (gfx950).
// Each wave runs ITER x 32 independent instructions of one kind; 8 waves per SIMD; reports cycles per
// wave-instruction per SIMD (2 = 32 lanes/clk, 4 = 16 lanes/clk).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7 + i;
  uint32_t c1 = seed ^ 0x0A0A0A0A, c2 = seed | 0x01010101;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(c1));
        if (KIND == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(c1));
        if (KIND == 2) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(seed), "v"(c2));
        if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (KIND == 4) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "s"(seed), "v"(c2));
        if (KIND == 5) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
        if (KIND == 6) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(c2));
        if (KIND == 7) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        if (KIND == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (KIND == 9) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (KIND == 10) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (KIND == 11) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c1));
        if (KIND == 13) asm volatile("v_cmp_eq_u32 vcc, %0, %1" :: "v"(a[i]), "v"(c1) : "vcc");
        if (KIND == 14) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
        if (KIND == 15) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
        if (KIND == 16) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
        if (KIND == 17) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i]));
        if (KIND == 18) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (KIND == 19) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (KIND == 20) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (KIND == 21) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (KIND == 22) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (KIND == 23) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (KIND == 24) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(c1));
        if (KIND == 25) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c1));
        if (KIND == 26) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a[i]) : "v"(c1));
        if (KIND == 27) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca" : "+v"(a[i]) : "s"(seed), "v"(c2));
        if (KIND == 28) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(c1));
        if (KIND == 29) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
        if (KIND == 30) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(c1) : "vcc");
        if (KIND == 31) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
        if (KIND == 32) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(c1));
        if (KIND == 33) asm volatile("v_pk_lshrrev_b16 %0, %1, %0" : "+v"(a[i]) : "v"(c2));
        if (KIND == 35) asm volatile("v_add_u32 %0, 0x7f7f7f7f, %0" : "+v"(a[i]));
        if (KIND == 36) asm volatile("v_xor_b32 %0, 0x0a0a0a0a, %0" : "+v"(a[i]));
        if (KIND == 37) asm volatile("v_and_b32 %0, 15, %0" : "+v"(a[i]));
        if (KIND == 38) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(c2));
        if (KIND == 39) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c2));
        if (KIND == 40) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
        if (KIND == 41) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(*(unsigned long long*)&a[i & 6]));
        if (KIND == 42) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
        if (KIND == 43) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(c2) : "vcc");
        if (KIND == 44) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
        if (KIND == 34) asm volatile("v_bitop3_b16 %0, %0, %1, %2 bitop3:0xca" : "+v"(a[i]) : "v"(c1), "v"(c2));
      }
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> double run(const char* name, uint32_t* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)blocks * 4 / (256.0 * 4) * ITER * 32.0;
  printf("%-16s %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz, %.2f at 2.0 GHz)\n", name, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, ms * 1e6 / instr_per_simd * 2.0);
  return ms;
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<8>("v_fma_f32", d); run<0>("v_and_b32", d); run<1>("v_add_u32", d); run<2>("v_xad_u32", d); run<3>("v_perm_b32", d);
  run<4>("v_bfi_b32", d); run<5>("v_bcnt_u32_b32", d); run<6>("v_lshl_add_u32", d); run<7>("v_add_u32_dpp", d);
  run<9>("v_bitop3_b32", d); run<10>("v_pk_add_u16", d); run<11>("v_or3_b32", d);
  run<12>("v_cndmask_b32", d); run<13>("v_cmp_eq_u32", d); run<14>("v_lshrrev_b32", d); run<15>("v_not_b32", d);
  run<16>("v_ffbl_b32", d); run<17>("v_bfrev_b32", d); run<18>("v_and_or_b32", d); run<19>("v_add3_u32", d);
  run<20>("v_mul_u32_u24", d); run<21>("v_mad_u32_u24", d); run<22>("v_sad_u8", d); run<23>("v_dot4_u32_u8", d);
  run<24>("v_and_b32_sdwa", d); run<25>("v_mov_b32_dpp", d); run<26>("v_alignbit_b32", d); run<27>("v_bitop3 s,v", d);
  run<28>("v_lshl_or_b32", d); run<29>("v_xor_b32 sgpr", d); run<30>("v_sub_co_u32", d); run<31>("v_bfe_u32", d);
  run<35>("v_add_u32_literal", d); run<36>("v_xor_b32_literal", d); run<37>("v_and_b32_inline", d); run<38>("v_lshrrev_vgprshift", d);
  run<39>("v_mov_b32", d); run<40>("v_sub_u32", d); run<41>("v_lshlrev_b64", d); run<42>("v_max_u32", d); run<43>("v_add_co_u32", d); run<44>("v_or_b32", d);
  run<32>("v_mbcnt_lo", d); run<33>("v_pk_lshrrev_b16", d); run<34>("v_bitop3_b16", d);
  return 0;
}
