// Probe: what does it cost when every workgroup of a 256-workgroup launch adds
// a row of 2 048 doubles into ONE of 8 rows (its XCD's), with atomics that
// execute in the XCD's own L2 (no scope bits) or at the device's coherence
// point (agent scope)?  The question behind a two-level reduction of the mass
// statistics (docs/LABNOTES.md section 10).
// Build: hipcc --offload-arch=gfx950 -O3 tools/archive/l2_atomic_probe.hip -o build/l2_atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k(double* rows, int n_cols, int spin) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7;
  // equal work first: the workgroups arrive together, the worst case for the adds
  float v = threadIdx.x * 1e-3f;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  double* row = rows + (size_t)xcc * n_cols;
  for (int c = threadIdx.x; c < n_cols; c += blockDim.x) {
    const double x = (double)v + c;
    if (MODE == 1)
      __hip_atomic_fetch_add(&row[c], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 2)
      __hip_atomic_fetch_add(&row[c], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 3)
      row[(size_t)(blockIdx.x) * 0 + c] = x;   // plain store to the XCD row (races: timing only)
  }
  if (MODE == 0 && v == 12345.f) rows[0] = v;
}

static int g_spin = 0;
template <int MODE>
float run(const char* name, double* rows, int n_cols) {
  hipMemset(rows, 0, 8 * n_cols * sizeof(double));
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, rows, n_cols, g_spin);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 50;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, rows, n_cols, g_spin);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.2f us per launch\n", name, ms * 1e3 / reps);
  return ms;
}

int main(int argc, char** argv) {
  if (argc > 1) g_spin = atoi(argv[1]);   // equal work per workgroup before the adds
  printf("spin %d\n", g_spin);
  const int n_cols = 2048;
  double* rows;
  hipMalloc(&rows, 8 * n_cols * sizeof(double));
  run<0>("no memory traffic", rows, n_cols);
  run<3>("plain stores", rows, n_cols);
  run<1>("f64 atomics, no scope bits (XCD-local L2)", rows, n_cols);
  run<2>("f64 atomics, agent scope", rows, n_cols);
  double h[8];
  for (int x = 0; x < 8; ++x) hipMemcpy(&h[x], rows + (size_t)x * n_cols, 8, hipMemcpyDeviceToHost);
  printf("row heads after the agent-scope run: %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f\n",
         h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  return 0;
}
