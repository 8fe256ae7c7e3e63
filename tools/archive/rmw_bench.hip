// Ceiling probe: in-place read-modify-write of a [C, 1024] float32 array
// (the access pattern of the fused HMC kernel without any arithmetic), one
// 4 KiB row per wave-trip, grid-stride; plus copy and read-only for reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0 = rmw in place, 1 = copy src->dst, 2 = read only
__global__ __launch_bounds__(256) void k(float* q, float* dst, long rows, float* sink) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  f4 acc = {0, 0, 0, 0};
  for (long r = wave; r < rows; r += nw) {
    f4* p = (f4*)(q + r * 1024) + lane;
    f4 a = p[0], b = p[64], c = p[128], d = p[192];
    if (MODE == 2) { acc += a + b + c + d; continue; }
    f4* o = MODE == 0 ? p : (f4*)(dst + r * 1024) + lane;
    o[0] = a + 1.f; o[64] = b + 1.f; o[128] = c + 1.f; o[192] = d + 1.f;
  }
  if (MODE == 2 && acc[0] == 123.f) sink[0] = acc[1];
}

template <int MODE>
void run(const char* name, int blocks, float* q, float* dst, long rows, float* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, q, dst, rows, sink);
  hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, q, dst, rows, sink);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
  double bytes = (double)rows * 4096 * (MODE == 2 ? 1 : 2);
  printf("%-10s blocks=%5d  %.4f ms  %.0f GB/s\n", name, blocks, ms, bytes / ms / 1e6);
}

int main() {
  const long rows = 65536;
  float *q, *dst, *sink;
  hipMalloc(&q, rows * 4096); hipMalloc(&dst, rows * 4096); hipMalloc(&sink, 64);
  hipMemset(q, 0, rows * 4096);
  for (int blocks : {256 * 2, 256 * 4, 256 * 8, 16384}) {
    run<0>("rmw", blocks, q, dst, rows, sink);
    run<1>("copy", blocks, q, dst, rows, sink);
    run<2>("read", blocks, q, dst, rows, sink);
  }
  return 0;
}
