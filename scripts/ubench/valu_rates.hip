// Micro-benchmark behind DESIGN 3.2: issue cost (shader cycles per wave64 instruction) of the vector instructions the
// small-window kernel's epilogue is made of, for ONE wave per SIMD, alone and beside a stream of MFMAs.
// hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)

template <int V>
__global__ __launch_bounds__(256) void k(unsigned long long *out, float *sink, int iters) {
  float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c2 = c;
  half8 wa = {1, 2, 3, 4, 5, 6, 7, 8}, wb = {1, 1, 1, 1, 1, 1, 1, 1};
  unsigned u0 = threadIdx.x, u1 = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (V == 0) {   // 32 independent v_fma_f32
      asm volatile(R4("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                      "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (V == 1) {   // 32 independent v_pk_fma_f32
      asm volatile(R4("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                      "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
    } else if constexpr (V == 2) {   // v_exp_f32
      asm volatile(R4("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                      "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (V == 3) {   // v_rcp_f32
      asm volatile(R4("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                      "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (V == 4) {   // 1 trans : 3 packed, independent (does the transcendental unit run beside the main pipe?)
      asm volatile(R4("v_exp_f32 %0, %0\n v_pk_fma_f32 %8, %8, %8, %8\n v_pk_fma_f32 %9, %9, %9, %9\n v_pk_fma_f32 %10, %10, %10, %10\n"
                      "v_exp_f32 %1, %1\n v_pk_fma_f32 %11, %11, %11, %11\n v_pk_fma_f32 %8, %8, %8, %8\n v_pk_fma_f32 %9, %9, %9, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    } else if constexpr (V == 5) {   // 1 trans : 1 packed
      asm volatile(R4("v_exp_f32 %0, %0\n v_pk_fma_f32 %8, %8, %8, %8\n v_exp_f32 %1, %1\n v_pk_fma_f32 %9, %9, %9, %9\n"
                      "v_exp_f32 %2, %2\n v_pk_fma_f32 %10, %10, %10, %10\n v_exp_f32 %3, %3\n v_pk_fma_f32 %11, %11, %11, %11\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    } else if constexpr (V == 6) {   // v_cvt_pk_f16_f32 + v_fma_mix_f32 + v_cndmask + v_max3 (the store path's mix)
      asm volatile(R4("v_cvt_pk_f16_f32 %8, %0, %1\n v_fma_mix_f32 %2, %8, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"
                      "v_fma_mix_f32 %3, %8, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_cvt_pk_f16_f32 %9, %2, %3\n"
                      "v_max3_f32 %4, %4, |%0|, |%1|\n v_and_b32 %8, %8, %9\n v_and_b32 %9, %9, %8\n v_max3_f32 %5, %5, |%6|, |%7|\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(u0), "+v"(u1));
    } else if constexpr (V == 7) {   // accumulator-file moves
      asm volatile(R4("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n"
                      "v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    } else if constexpr (V == 8) {   // 4 dependent MFMAs (one accumulator)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n") : "+v"(c) : "v"(wa), "v"(wb));
    } else if constexpr (V == 9) {   // 4 x (1 MFMA + 7 independent packed FMAs): does the vector work hide under the MFMA?
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n"
                      "v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n v_pk_fma_f32 %8, %8, %8, %8\n v_pk_fma_f32 %9, %9, %9, %9\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6));
    } else if constexpr (V == 10) {  // 4 x (1 MFMA + 12 independent vector instructions incl. two transcendentals)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_exp_f32 %10, %10\n v_pk_fma_f32 %5, %5, %5, %5\n"
                      "v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n v_rcp_f32 %11, %11\n v_pk_fma_f32 %8, %8, %8, %8\n v_pk_fma_f32 %9, %9, %9, %9\n"
                      "v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(a0), "v"(a1));
    } else if constexpr (V == 11) {  // the same 12 vector instructions without the MFMA
      asm volatile(R4("v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_exp_f32 %10, %10\n v_pk_fma_f32 %5, %5, %5, %5\n"
                      "v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n v_rcp_f32 %11, %11\n v_pk_fma_f32 %8, %8, %8, %8\n v_pk_fma_f32 %9, %9, %9, %9\n"
                      "v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(a0), "v"(a1));
    } else if constexpr (V == 12) {  // two independent MFMA chains
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n") : "+v"(c), "+v"(c2) : "v"(wa), "v"(wb));
    } else if constexpr (V == 13) {  // v_pk_mul_f32 with dependent chain (result feeds the next): forwarding stalls?
      asm volatile(R32("v_pk_mul_f32 %0, %0, %1\n") : "+v"(p0) : "v"(p1));
    } else if constexpr (V == 14) {  // v_mul_f32 dependent chain
      asm volatile(R32("v_mul_f32 %0, %0, %1\n") : "+v"(a0) : "v"(a1));
    } else if constexpr (V == 16) {  // 4 x (1 MFMA + 14 plain FMAs): unpacked vector work in the MFMA's shadow
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n"
                      "v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n"
                      "v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n"
                      "v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 17) {  // 4 x (1 MFMA + 4 transcendentals)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_exp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_exp_f32 %5, %5\n v_rcp_f32 %6, %6\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 18) {  // 4 x (1 MFMA + 3 v_fma_f32): is a short vector slice free?
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 19) {  // 4 x (1 MFMA + 3 v_pk_fma_f32)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6));
    } else if constexpr (V == 20) {  // is an s_nop a full issue slot for a lone wave?
      asm volatile(R16("v_fma_f32 %0, %0, %0, %0\n s_nop 0\n") : "+v"(a0));
    } else if constexpr (V == 21) {  // scalar instructions between vector instructions
      asm volatile(R16("v_fma_f32 %0, %0, %0, %0\n s_add_u32 s20, s20, 1\n") : "+v"(a0) : : "s20");
    } else if constexpr (V == 22) {  // dependent packed chain with the hazard nop the compiler inserts
      asm volatile(R16("v_pk_mul_f32 %0, %0, %1\n s_nop 0\n") : "+v"(p0) : "v"(p1));
    } else if constexpr (V == 23) {  // two interleaved packed chains (no nop needed)
      asm volatile(R16("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n") : "+v"(p0), "+v"(p3) : "v"(p1));
    } else if constexpr (V == 24) {  // 4 x (1 MFMA + cvt_pk, fma_mix, fma_mix: the store path's instructions)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_cvt_pk_f16_f32 %7, %3, %4\n v_fma_mix_f32 %5, %7, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"
                      "v_fma_mix_f32 %6, %7, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(u0));
    } else if constexpr (V == 25) {  // 4 x (1 MFMA + 5 plain)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 26) {  // 4 x (1 MFMA + 7 plain)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 27) {  // 2 independent MFMA chains with 3 plain behind each MFMA
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n"
                      "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n")
                   : "+v"(c), "+v"(c2) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 28) {  // 4 x (MFMA 16x16x32: 16 cycles + 2 plain)
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      asm volatile(R4("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n")
                   : "+v"(*reinterpret_cast<f32x4 *>(&c)) : "v"(wa), "v"(wb), "v"(a0), "v"(a1));
    } else if constexpr (V == 29) {  // 4 x (1 MFMA + 2 transcendentals + 4 plain)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %3, %3\n v_exp_f32 %7, %7\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_rcp_f32 %8, %8\n v_fma_f32 %6, %6, %6, %6\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 30) {  // 4 x (1 MFMA + 1 transcendental + 5 plain)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %3, %3\n v_exp_f32 %7, %7\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %6, %6, %6, %6\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 31) {  // 4 x (1 MFMA + 2 transcendentals + 2 plain)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %3, %3\n v_exp_f32 %7, %7\n v_fma_f32 %4, %4, %4, %4\n v_rcp_f32 %8, %8\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 32) {  // 4 x (1 MFMA + 2 transcendentals)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_exp_f32 %7, %7\n v_rcp_f32 %8, %8\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 33) {  // 4 x (1 MFMA + 3 transcendentals + 3 plain)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_exp_f32 %7, %7\n v_fma_f32 %3, %3, %3, %3\n v_exp_f32 %9, %9\n v_fma_f32 %4, %4, %4, %4\n v_rcp_f32 %8, %8\n v_fma_f32 %5, %5, %5, %5\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 34) {  // 4 x (1 MFMA + 4 transcendentals + 4 plain)
      asm volatile(R4("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_exp_f32 %7, %7\n v_fma_f32 %3, %3, %3, %3\n v_exp_f32 %9, %9\n v_fma_f32 %4, %4, %4, %4\n v_rcp_f32 %8, %8\n v_fma_f32 %5, %5, %5, %5\n v_rcp_f32 %6, %6\n v_fma_f32 %3, %3, %3, %3\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 35) {  // 4 transcendentals + 4 plain, no MFMA
      asm volatile(R4("v_exp_f32 %7, %7\n v_fma_f32 %3, %3, %3, %3\n v_exp_f32 %9, %9\n v_fma_f32 %4, %4, %4, %4\n v_rcp_f32 %8, %8\n v_fma_f32 %5, %5, %5, %5\n v_rcp_f32 %6, %6\n v_fma_f32 %3, %3, %3, %3\n")
                   : "+v"(c) : "v"(wa), "v"(wb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));
    } else if constexpr (V == 15) {  // exp -> dependent add -> dependent rcp chain (the GELU tail), one element
      asm volatile(R8("v_exp_f32 %0, %0\n v_add_f32 %0, 1.0, %0\n v_rcp_f32 %0, %0\n v_mul_f32 %0, %0, %1\n") : "+v"(a0) : "v"(a1));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[0] + p2[0] + p3[0] + p4[1] + p5[1] + p6[1] + p7[1] + c[0] + c2[3] + (float)(u0 + u1);
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
}

#define RUN(V, NINS, label)                                                                       \
  do {                                                                                            \
    hipMemset(d, 0, 8);                                                                           \
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, d, sink, iters);                         \
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, d, sink, iters);                         \
    hipMemset(d, 0, 8);                                                                           \
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, d, sink, iters);                         \
    hipDeviceSynchronize();                                                                       \
    unsigned long long h;                                                                         \
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);                                                   \
    printf("%-64s %7.2f cycles per instruction group of %d -> %.2f per instruction\n", label,      \
           (double)h / 1024.0 / iters, NINS, (double)h / 1024.0 / iters / NINS);                   \
  } while (0)

int main() {
  unsigned long long *d;
  float *sink;
  hipMalloc(&d, 8);
  hipMalloc(&sink, 4);
  const int iters = 2000;
  RUN(0, 32, "v_fma_f32 x32 independent");
  RUN(1, 32, "v_pk_fma_f32 x32 independent");
  RUN(2, 32, "v_exp_f32 x32 independent");
  RUN(3, 32, "v_rcp_f32 x32 independent");
  RUN(4, 32, "(1 v_exp_f32 + 3 v_pk_fma_f32) x8");
  RUN(5, 32, "(1 v_exp_f32 + 1 v_pk_fma_f32) x16");
  RUN(6, 32, "store path mix (cvt_pk, fma_mix, max3, and) x32");
  RUN(7, 32, "v_accvgpr_write x16 + v_accvgpr_read x16");
  RUN(8, 4, "4 dependent v_mfma_f32_32x32x16_f16");
  RUN(9, 4, "4 x (1 MFMA + 7 v_pk_fma_f32)");
  RUN(10, 4, "4 x (1 MFMA + 10 v_pk_fma_f32 + exp + rcp)");
  RUN(11, 4, "4 x (10 v_pk_fma_f32 + exp + rcp), no MFMA");
  RUN(12, 8, "2 independent MFMA chains x4");
  RUN(13, 32, "v_pk_mul_f32 x32 dependent chain");
  RUN(14, 32, "v_mul_f32 x32 dependent chain");
  RUN(15, 32, "(exp, add, rcp, mul) dependent x8");
  RUN(16, 4, "4 x (1 MFMA + 14 v_fma_f32)");
  RUN(17, 4, "4 x (1 MFMA + 2 v_exp_f32 + 2 v_rcp_f32)");
  RUN(18, 4, "4 x (1 MFMA + 3 v_fma_f32)");
  RUN(19, 4, "4 x (1 MFMA + 3 v_pk_fma_f32)");
  RUN(29, 4, "4 x (1 MFMA + 2 trans + 4 v_fma_f32)");
  RUN(30, 4, "4 x (1 MFMA + 1 trans + 5 v_fma_f32)");
  RUN(31, 4, "4 x (1 MFMA + 2 trans + 2 v_fma_f32)");
  RUN(32, 4, "4 x (1 MFMA + 2 trans)");
  RUN(33, 4, "4 x (1 MFMA + 3 trans + 3 v_fma_f32)");
  RUN(34, 4, "4 x (1 MFMA + 4 trans + 4 v_fma_f32)");
  RUN(35, 4, "4 x (4 trans + 4 v_fma_f32), no MFMA");
  RUN(20, 32, "(v_fma_f32 + s_nop 0) x16");
  RUN(22, 32, "(v_pk_mul_f32 dependent + s_nop 0) x16");
  RUN(23, 32, "two interleaved dependent v_pk_mul_f32 chains x16");
  RUN(24, 4, "4 x (1 MFMA + cvt_pk + 2 fma_mix)");
  RUN(25, 4, "4 x (1 MFMA + 5 v_fma_f32)");
  RUN(26, 4, "4 x (1 MFMA + 7 v_fma_f32)");
  RUN(27, 8, "2 MFMA chains x4, 3 v_fma_f32 behind each MFMA");
  RUN(28, 4, "4 x (1 MFMA 16x16x32 + 2 v_fma_f32)");
  return 0;
}
