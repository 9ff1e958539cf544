// Microbenchmark: LDS atomic-add throughput on gfx950 for float / uint32 / uint64 / double operands,
// 40 active lanes per wave to consecutive addresses (the dcomp pattern of k_mix_bwd_nm), one
// 1024-thread block per CU.   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_atomics.hip -o lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T> __global__ __launch_bounds__(1024) void k(int iters, int rows, T *out) {
  extern __shared__ __align__(16) unsigned char smem[];
  T *s = reinterpret_cast<T *>(smem);
  for (int t = threadIdx.x; t < rows * 40; t += blockDim.x) s[t] = T(0);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned r = (wv * 7919u + blockIdx.x * 104729u) % rows;
  if (lane < 40)
    for (int i = 0; i < iters; ++i) {
      atomicAdd(&s[r * 40 + lane], T(1));
      r = (r * 1664525u + 1013904223u) % rows;
    }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[0];
}

template <typename T> void run(const char *name) {
  T *out;
  hipMalloc(&out, 256 * sizeof(T));
  const int iters = 20000, rows = 267;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<T><<<256, 1024, rows * 40 * sizeof(T)>>>(10, rows, out);
  hipEventRecord(e0);
  k<T><<<256, 1024, rows * 40 * sizeof(T)>>>(iters, rows, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = 256.0 * 16 * iters;  // wave-level atomic instructions
  printf("%-8s %8.3f ms  %.1f ns per wave-atomic per CU  (%.2f cycles/lane at 2.4 GHz)\n", name, ms,
         ms * 1e6 / (16.0 * iters), ms * 1e6 / (16.0 * iters) * 2.4 / 40.0);
  (void)instr;
  hipFree(out);
}

int main() {
  run<float>("float");
  run<unsigned int>("uint32");
  run<unsigned long long>("uint64");
  run<double>("double");
  return 0;
}
