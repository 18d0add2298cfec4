// Issue cost of a few vector instructions on gfx950: one wavefront per SIMD, 64 independent copies of the instruction
// between two s_memtime reads (developer tool; numbers quoted in DESIGN.md 7.2).
// build: hipcc -O2 --offload-arch=gfx950 valu_cost.hip -o valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define KERNEL(name, body)                                                                   \
  __global__ void name(long* out, unsigned seed) {                                           \
    unsigned a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 7 + 1;           \
    unsigned d0 = a, d1 = b, d2 = c, d3 = a ^ b;                                             \
    long t0, t1;                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)); \
    for (int it = 0; it < 16; ++it) {                                                        \
      asm volatile(REP16(body) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b), "v"(c), "s"(0xffff0001u)); \
    }                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)); \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                         \
    if (d0 + d1 + d2 + d3 == 0x12345) out[1000] = 1;                                         \
  }

#define KERNEL64(name, body)                                                                 \
  __global__ void name(long* out, unsigned seed) {                                           \
    unsigned long long a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 7 + 1; \
    unsigned long long d0 = a, d1 = b, d2 = c, d3 = a ^ b;                                   \
    long t0, t1;                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)); \
    for (int it = 0; it < 16; ++it) {                                                        \
      asm volatile(REP16(body) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b), "v"(c)); \
    }                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)); \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                         \
    if (d0 + d1 + d2 + d3 == 0x12345) out[1000] = 1;                                         \
  }

// each body = 4 independent instructions (d0..d3), so a REP16 = 64 instructions, x16 iterations = 1024
KERNEL(k_add, "v_add_u32 %0, %4, %0\n v_add_u32 %1, %5, %1\n v_add_u32 %2, %6, %2\n v_add_u32 %3, %4, %3\n")
KERNEL(k_dot2, "v_dot2_i32_i16 %0, %4, %5, 0\n v_dot2_i32_i16 %1, %5, %6, 0\n v_dot2_i32_i16 %2, %6, %4, 0\n v_dot2_i32_i16 %3, %4, %4, 0\n")
KERNEL(k_dot2c, "v_dot2c_i32_i16 %0, %4, %5\n v_dot2c_i32_i16 %1, %5, %6\n v_dot2c_i32_i16 %2, %6, %4\n v_dot2c_i32_i16 %3, %4, %4\n")
KERNEL64(k_pkmulf32, "v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %5, %6\n v_pk_mul_f32 %2, %6, %4\n v_pk_mul_f32 %3, %4, %4\n")
KERNEL(k_mulf32, "v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %5, %6\n v_mul_f32 %2, %6, %4\n v_mul_f32 %3, %4, %4\n")
KERNEL(k_cvtf32i32, "v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %4\n")
KERNEL(k_cvtpkf16, "v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %5, %6\n v_cvt_pk_f16_f32 %2, %6, %4\n v_cvt_pk_f16_f32 %3, %4, %4\n")
KERNEL(k_pkmullo, "v_pk_mul_lo_u16 %0, %4, %7 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_mul_lo_u16 %1, %5, %7 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_mul_lo_u16 %2, %6, %7 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_mul_lo_u16 %3, %4, %7 op_sel:[1,0] op_sel_hi:[0,1]\n")
KERNEL(k_cvtf16i16, "v_cvt_f16_i16 %0, %4\n v_cvt_f16_i16 %1, %5\n v_cvt_f16_i16 %2, %6\n v_cvt_f16_i16 %3, %4\n")
KERNEL(k_cvtf16i16hi, "v_cvt_f16_i16_sdwa %0, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_cvt_f16_i16_sdwa %1, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_cvt_f16_i16_sdwa %2, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_cvt_f16_i16_sdwa %3, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n")
KERNEL(k_perm, "v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %5, %6, %4\n v_perm_b32 %2, %6, %4, %5\n v_perm_b32 %3, %4, %4, %5\n")
KERNEL(k_madi16, "v_mad_i32_i16 %0, %4, %5, %0\n v_mad_i32_i16 %1, %5, %6, %1\n v_mad_i32_i16 %2, %6, %4, %2\n v_mad_i32_i16 %3, %4, %4, %3\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %4, %5, 16\n v_alignbit_b32 %1, %5, %6, 16\n v_alignbit_b32 %2, %6, %4, 16\n v_alignbit_b32 %3, %4, %4, 16\n")
KERNEL(k_pkfmaf16, "v_pk_fma_f16 %0, %4, %5, %0\n v_pk_fma_f16 %1, %5, %6, %1\n v_pk_fma_f16 %2, %6, %4, %2\n v_pk_fma_f16 %3, %4, %4, %3\n")
KERNEL(k_dot2f32f16, "v_dot2_f32_f16 %0, %4, %5, %0\n v_dot2_f32_f16 %1, %5, %6, %1\n v_dot2_f32_f16 %2, %6, %4, %2\n v_dot2_f32_f16 %3, %4, %4, %3\n")
KERNEL(k_cvtf32f16, "v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %5\n v_cvt_f32_f16 %2, %6\n v_cvt_f32_f16 %3, %4\n")
KERNEL(k_pkmulf16, "v_pk_mul_f16 %0, %4, %5\n v_pk_mul_f16 %1, %5, %6\n v_pk_mul_f16 %2, %6, %4\n v_pk_mul_f16 %3, %4, %4\n")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 15, %4\n v_ashrrev_i32 %1, 15, %5\n v_ashrrev_i32 %2, 15, %6\n v_ashrrev_i32 %3, 15, %4\n")

int main() {
  long* out;
  hipMalloc(&out, 2048 * sizeof(long));
  struct K { const char* name; void (*fn)(long*, unsigned); };
  K ks[] = {{"v_add_u32", k_add}, {"v_dot2_i32_i16", k_dot2}, {"v_dot2c_i32_i16", k_dot2c}, {"v_pk_mul_f32", k_pkmulf32},
            {"v_mul_f32", k_mulf32}, {"v_cvt_f32_i32", k_cvtf32i32}, {"v_cvt_pk_f16_f32", k_cvtpkf16},
            {"v_pk_mul_lo_u16", k_pkmullo}, {"v_cvt_f16_i16", k_cvtf16i16}, {"v_cvt_f16_i16 sdwa hi", k_cvtf16i16hi},
            {"v_perm_b32", k_perm}, {"v_mad_i32_i16", k_madi16}, {"v_alignbit_b32", k_alignbit},
            {"v_pk_fma_f16", k_pkfmaf16}, {"v_dot2_f32_f16", k_dot2f32f16}, {"v_cvt_f32_f16", k_cvtf32f16},
            {"v_pk_mul_f16", k_pkmulf16}, {"v_ashrrev_i32", k_ashr}};
  for (int waves = 1; waves <= 2; ++waves) {
    printf("-- %d wavefront(s) per SIMD\n", waves);
    for (auto& k : ks) {
      std::vector<long> h(4);
      for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(k.fn, dim3(1), dim3(256 * waves), 0, 0, out, 12345u + r);
        hipDeviceSynchronize();
      }
      hipMemcpy(h.data(), out, sizeof(long), hipMemcpyDeviceToHost);
      printf("%-20s %6.2f cycles per instruction (wavefront 0; 1024 instructions)\n", k.name, h[0] / 1024.0);
    }
  }
  return 0;
}
