// Probe for DESIGN 3.1 (round 2): does the distance between the read of a row
// and its write-back decide whether an in-place update runs at the
// read-modify-write rate (7.0 TB/s) or at the copy rate (5.5-5.9 TB/s)?
// Each wave keeps P rows in registers: it reads row i + P - 1 and then writes
// row i (P = 1: immediate RMW).  Optionally it sleeps between read and write
// to stretch the distance further (SLEEP x 64 clocks per row).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(256) void k(float* q, long rows, int sleep) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  f4 buf[P][4];
  long r = wave;
#pragma unroll
  for (int j = 0; j < P - 1; ++j) {
    const long rr = r + j * nw < rows ? r + j * nw : wave;
    f4* p = (f4*)(q + rr * 1024) + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) buf[j][c] = p[64 * c];
  }
  for (; r < rows; r += nw) {
    {
      const long rr = r + (P - 1) * nw < rows ? r + (P - 1) * nw : wave;
      f4* p = (f4*)(q + rr * 1024) + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c) buf[P - 1][c] = p[64 * c];
    }
    for (int s = 0; s < sleep; ++s) __builtin_amdgcn_s_sleep(1);
    f4* o = (f4*)(q + r * 1024) + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[64 * c] = buf[0][c] + 1.f;
#pragma unroll
    for (int j = 0; j + 1 < P; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) buf[j][c] = buf[j + 1][c];
  }
}

template <int P>
void run(int blocks, float* q, long rows, int sleep) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(256), 0, 0, q, rows, sleep);
  hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(256), 0, 0, q, rows, sleep);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
  printf("rows in flight per wave P=%d blocks=%5d sleep=%3d  %.4f ms  %.0f GB/s\n", P, blocks, sleep, ms,
         (double)rows * 8192 / ms / 1e6);
}

int main() {
  const long rows = 65536;
  float* q; hipMalloc(&q, rows * 4096); hipMemset(q, 0, rows * 4096);
  for (int blocks : {1024, 4096}) {
    run<1>(blocks, q, rows, 0);
    run<2>(blocks, q, rows, 0);
    run<3>(blocks, q, rows, 0);
    run<3>(blocks, q, rows, 20);
    run<3>(blocks, q, rows, 100);
    run<1>(blocks, q, rows, 100);
  }
  return 0;
}
