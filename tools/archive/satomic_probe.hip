// Probe: do scalar memory atomics (s_atomic_add ... glc) work on gfx950 as a
// device-wide ticket counter?  Every wave draws T tickets; all returned values
// must be a permutation of 0 .. n_waves*T-1.  Also times one draw (s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ void k(unsigned* counter, unsigned* out, unsigned long long* cyc, int T) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = clock64();
  for (int t = 0; t < T; ++t) {
    unsigned v;
    asm volatile(
        "s_mov_b32 %0, 1\n\t"
        "s_atomic_add %0, %1, 0x0 glc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(v)
        : "s"(counter)
        : "memory");
    if (lane == 0) out[wave * T + t] = v;
  }
  unsigned long long t1 = clock64();
  if (lane == 0) cyc[wave] = t1 - t0;
}

int main() {
  const int blocks = 1024, threads = 256, T = 16;
  const int n_waves = blocks * threads / 64, n = n_waves * T;
  unsigned *counter, *out; unsigned long long* cyc;
  hipMalloc(&counter, 4); hipMalloc(&out, n * 4); hipMalloc(&cyc, n_waves * 8);
  hipMemset(counter, 0, 4);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, counter, out, cyc, T);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  std::vector<unsigned> h(n); std::vector<unsigned long long> c(n_waves);
  hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), cyc, n_waves * 8, hipMemcpyDeviceToHost);
  unsigned fin; hipMemcpy(&fin, counter, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  int bad = 0; for (int i = 0; i < n; ++i) bad += h[i] != (unsigned)i;
  double mean = 0; for (auto v : c) mean += (double)v; mean /= n_waves;
  printf("final counter %u (expect %d); permutation errors %d; mean cycles per draw %.0f\n", fin, n, bad, mean / T);
  return 0;
}
