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

template <typename K>
void run(const char* name, K kern, int waves_per_simd) {
  const int blocks = 256 * 4;  // 4 single-wave blocks per CU -> one per SIMD
  const int nb = blocks * waves_per_simd;
  uint64_t* out; float* sink;
  hipMalloc(&out, nb * sizeof(uint64_t));
  hipMalloc(&sink, nb * 64 * sizeof(float));
  const int iters = 200;
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
  for (int w : {3, 4, 8}) {
#define R(k) run(#k, k, w)
    R(k_fma); R(k_pk_fma); R(k_xor); R(k_add_u32); R(k_mad_u64); R(k_log); R(k_cvt);
#if 0
    R(k_mul_u24);
#endif
  }
  return 0;
}
