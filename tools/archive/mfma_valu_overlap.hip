// Probe: do VALU / transcendental instructions of a wave execute under its own
// in-flight MFMAs on gfx950?  One wave per SIMD.  Three kernels with the same
// loop: 8 independent-accumulator v_mfma_f32_32x32x2_f32 only; 32 VALU + 8
// transcendental ops only; both interleaved (1 MFMA : 4 VALU : 1 trans).
// overlap  => t(both) ~ max(t_mfma, t_valu);  none => t(both) ~ sum.
// Build: hipcc --offload-arch=gfx950 -O3 tools/archive/mfma_valu_overlap.hip -o build/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
  f16v acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = threadIdx.x * 1e-3f + 0.5f, b = a + 0.25f;
  float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE & 1)
        acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j & 3], 0, 0, 0);
      if (MODE & 2) {
        asm volatile(
            "v_fma_f32 %0, %0, %1, %2\n\t"
            "v_fma_f32 %1, %1, %2, %3\n\t"
            "v_fma_f32 %2, %2, %3, %0\n\t"
            "v_fma_f32 %3, %3, %0, %1\n\t"
            "v_exp_f32 %0, %0"
            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      }
    }
  }
  float s = v0 + v1 + v2 + v3;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][15];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
float run(const char* name, float* out) {
  const int blocks = 256 * 4, iters = 4000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.3f ms  (%.1f ns per 8-MFMA / 40-VALU group)\n", name, ms,
         ms * 1e6 / iters);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 64 * sizeof(float));
  const float m = run<1>("MFMA only", out);
  const float v = run<2>("VALU+trans only", out);
  const float b = run<3>("interleaved", out);
  printf("sum %.3f  max %.3f  measured %.3f -> %s\n", m + v, m > v ? m : v, b,
         b < 0.5f * (m + v + (m > v ? m : v)) ? "overlap" : "NO overlap");
  return 0;
}
