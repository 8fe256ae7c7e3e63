// Micro-benchmark: issue cost (cycles per wave64 instruction, one wave per
// SIMD and 2 waves per SIMD) of the VALU instructions that dominate the
// fused HMC kernel on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 tools/instr_bench.hip -o build/instr_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define KERNEL(name, SETUP, BODY)                                         \
  __global__ void name(uint64_t* out, float* sink, int iters) {           \
    SETUP                                                                  \
    uint64_t t0 = __builtin_readcyclecounter();                            \
    for (int i = 0; i < iters; ++i) { REP64(BODY) }                        \
    uint64_t t1 = __builtin_readcyclecounter();                            \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                       \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = fa + fb + fc + fd + (float)(ua + ub + uc + ud) + (float)(la + lb); \
  }

#define SETUP_ALL                                                          \
  float fa = threadIdx.x * 1e-3f + 0.1f, fb = fa + 1.f, fc = fa + 2.f, fd = fa + 3.f; \
  unsigned ua = threadIdx.x + 1, ub = ua * 3, uc = ua * 5, ud = ua * 7;    \
  uint64_t la = ua, lb = ub;                                               \
  typedef float f2 __attribute__((ext_vector_type(2)));                   \
  f2 pa = {fa, fb}, pb = {fc, fd}, pc = {fb, fc}, pd = {fd, fa};

KERNEL(k_fma, SETUP_ALL, asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_pk_fma, SETUP_ALL, asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd)); fa = pa[0];)
KERNEL(k_xor, SETUP_ALL, asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_add_u32, SETUP_ALL, asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_alignbit, SETUP_ALL, asm volatile("v_alignbit_b32 %0, %0, %0, 13\n v_alignbit_b32 %1, %1, %1, 7\n v_alignbit_b32 %2, %2, %2, 5\n v_alignbit_b32 %3, %3, %3, 21" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_mul_lo, SETUP_ALL, asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_mul_hi, SETUP_ALL, asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_mad_u64, SETUP_ALL, asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_mad_u64_u32 %1, vcc, %3, %2, 0\n v_mad_u64_u32 %0, vcc, %3, %3, 0\n v_mad_u64_u32 %1, vcc, %2, %2, 0" : "+v"(la), "+v"(lb) : "v"(ua), "v"(ub) : "vcc");)
KERNEL(k_mul_u24, SETUP_ALL, asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_hi_u32_u24 %2, %2, %3\n v_mul_hi_u32_u24 %3, %3, %0" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_log, SETUP_ALL, asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_sin, SETUP_ALL, asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_sqrt, SETUP_ALL, asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_exp, SETUP_ALL, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_rcp, SETUP_ALL, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_cvt, SETUP_ALL, asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %6\n v_cvt_f32_u32 %3, %7" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(ua), "v"(ub), "v"(uc), "v"(ud));)
#define SETUP_F64 SETUP_ALL double da = fa; double db = fb;
KERNEL(k_fma_f64, SETUP_F64, asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %0, %0\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %0, %0" : "+v"(da), "+v"(db)); fa = (float)da;)
KERNEL(k_mul_i32_i24_pk, SETUP_ALL, asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)

KERNEL(k_xor_dep, SETUP_ALL, asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %2\n v_xor_b32 %0, %0, %3\n v_xor_b32 %0, %0, %1" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_xor_sgpr, SETUP_ALL, asm volatile("v_xor_b32 %0, s4, %0\n v_xor_b32 %1, s5, %1\n v_xor_b32 %2, s6, %2\n v_xor_b32 %3, s7, %3" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_xor_mad_mix, SETUP_ALL, asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %5\n v_mad_u64_u32 %1, vcc, %3, %2, 0\n v_xor_b32 %4, %4, %2\n v_xor_b32 %5, %5, %3" : "+v"(la), "+v"(lb), "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud) : : "vcc");)
KERNEL(k_pk_xor_mix, SETUP_ALL, asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_xor_b32 %4, %4, %5\n v_pk_fma_f32 %1, %1, %2, %3\n v_xor_b32 %5, %5, %6\n v_pk_fma_f32 %2, %2, %3, %0\n v_xor_b32 %6, %6, %7\n v_pk_fma_f32 %3, %3, %0, %1\n v_xor_b32 %7, %7, %4" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd), "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud)); fa = pa[0];)
KERNEL(k_mov, SETUP_ALL, asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_mul_f32, SETUP_ALL, asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_add_dpp, SETUP_ALL, asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_cndmask, SETUP_ALL, asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)

KERNEL(k_cndmask_vccset, SETUP_ALL, asm volatile("s_mov_b64 vcc, 0x5555\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud) : : "vcc");)
KERNEL(k_cndmask_e64, SETUP_ALL, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[8:9]\n v_cndmask_b32_e64 %1, %1, %2, s[8:9]\n v_cndmask_b32_e64 %2, %2, %3, s[8:9]\n v_cndmask_b32_e64 %3, %3, %0, s[8:9]" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_cmp, SETUP_ALL, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : : "vcc");)
KERNEL(k_fma_sgpr, SETUP_ALL, asm volatile("v_fma_f32 %0, s4, %1, %0\n v_fma_f32 %1, s4, %2, %1\n v_fma_f32 %2, s4, %3, %2\n v_fma_f32 %3, s4, %0, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_pk_fma_sgpr, SETUP_ALL, asm volatile("v_pk_fma_f32 %0, s[4:5], %1, %0\n v_pk_fma_f32 %1, s[4:5], %2, %1\n v_pk_fma_f32 %2, s[4:5], %3, %2\n v_pk_fma_f32 %3, s[4:5], %0, %3" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd)); fa = pa[0];)
KERNEL(k_readlane, SETUP_ALL, asm volatile("v_readlane_b32 s8, %0, 63\n v_readlane_b32 s9, %1, 63\n v_readlane_b32 s10, %2, 63\n v_readlane_b32 s11, %3, 63" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud) : : "s8", "s9", "s10", "s11");)
KERNEL(k_mov_dpp, SETUP_ALL, asm volatile("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_mov_sgpr, SETUP_ALL, asm volatile("v_mov_b32 %0, s4\n v_mov_b32 %1, s5\n v_mov_b32 %2, s6\n v_mov_b32 %3, s7" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_add_lit, SETUP_ALL, asm volatile("v_add_f32 %0, 0x3fb8aa3b, %0\n v_add_f32 %1, 0x3fb8aa3b, %1\n v_add_f32 %2, 0x3fb8aa3b, %2\n v_add_f32 %3, 0x3fb8aa3b, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_xor_inl, SETUP_ALL, asm volatile("v_xor_b32 %0, 17, %0\n v_xor_b32 %1, 17, %1\n v_xor_b32 %2, 17, %2\n v_xor_b32 %3, 17, %3" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_mad_u64_sgpr, SETUP_ALL, asm volatile("v_mad_u64_u32 %0, vcc, s4, %3, 0\n v_mad_u64_u32 %1, vcc, s5, %2, 0\n v_mad_u64_u32 %0, vcc, s4, %3, 0\n v_mad_u64_u32 %1, vcc, s5, %2, 0" : "+v"(la), "+v"(lb) : "v"(ua), "v"(ub) : "vcc");)
KERNEL(k_sub_f32, SETUP_ALL, asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_sub_f32 %3, %3, %0" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_lshr, SETUP_ALL, asm volatile("v_lshrrev_b32 %0, 8, %0\n v_lshrrev_b32 %1, 8, %1\n v_lshrrev_b32 %2, 8, %2\n v_lshrrev_b32 %3, 8, %3" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_sin2, SETUP_ALL, asm volatile("v_sin_f32 %0, %0\n v_cos_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_log_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_pk_mul, SETUP_ALL, asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd)); fa = pa[0];)

KERNEL(k_bitop3, SETUP_ALL, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_bitop3_sgpr, SETUP_ALL, asm volatile("v_bitop3_b32 %0, %0, %1, s4 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, s5 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, s6 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, s7 bitop3:0x96" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_fmamk, SETUP_ALL, asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1\n v_fmamk_f32 %1, %1, 0x2f800000, %2\n v_fmamk_f32 %2, %2, 0x2f800000, %3\n v_fmamk_f32 %3, %3, 0x2f800000, %0" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
KERNEL(k_or_inl, SETUP_ALL, asm volatile("v_or_b32 %0, 1.0, %0\n v_or_b32 %1, 1.0, %1\n v_or_b32 %2, 1.0, %2\n v_or_b32 %3, 1.0, %3" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));)
KERNEL(k_add_f64, SETUP_F64, asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %0\n v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %0" : "+v"(da), "+v"(db)); fa = (float)da;)

template <typename K>
void run(const char* name, K kern, int waves_per_simd) {
  const int blocks = 256 * 4;  // 4 single-wave blocks per CU -> one per SIMD
  const int nb = blocks * waves_per_simd;
  uint64_t* out; float* sink;
  hipMalloc(&out, nb * sizeof(uint64_t));
  hipMalloc(&sink, nb * 64 * sizeof(float));
  const int iters = 2000;
  hipLaunchKernelGGL(kern, dim3(nb), dim3(64), 0, 0, out, sink, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(64), 0, 0, out, sink, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[8]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const double n_inst = (double)iters * 64 * 4;
  // wall-clock based: total wave-instructions per SIMD / time
  const double inst_per_simd = n_inst * waves_per_simd;
  printf("%-18s waves/SIMD=%d  clk/inst(counter)=%.2f  ns/inst/SIMD(wall)=%.3f  (%.3f ms)\n", name,
         waves_per_simd, (double)h[0] / n_inst, ms * 1e6 / inst_per_simd, ms);
  hipFree(out); hipFree(sink);
}

int main() {
  for (int w : {4}) {
#define R(k) run(#k, k, w)
    R(k_fma); R(k_pk_fma); R(k_xor); R(k_add_u32); R(k_mad_u64); R(k_log); R(k_cvt);
    R(k_xor_dep); R(k_xor_sgpr); R(k_xor_mad_mix); R(k_pk_xor_mix); R(k_mov); R(k_mul_f32); R(k_add_dpp); R(k_cndmask);
    R(k_cndmask_vccset); R(k_cndmask_e64); R(k_cmp); R(k_fma_sgpr); R(k_pk_fma_sgpr); R(k_readlane); R(k_mov_dpp); R(k_mov_sgpr); R(k_add_lit); R(k_xor_inl); R(k_mad_u64_sgpr); R(k_sub_f32); R(k_lshr); R(k_sin2); R(k_pk_mul);
#if 0
    R(k_mul_u24);
#endif
  }
  return 0;
}
