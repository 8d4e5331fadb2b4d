// Microbenchmark: cycles per VALU instruction for ONE wave on a SIMD, dependent chain vs k independent chains.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o valu_latency valu_latency.hip && ./valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(float* out, unsigned long long* cyc, float a, float b) {
  float x[CHAINS];
  for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 0.001f + c;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int it = 0; it < 256; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) x[c] = __builtin_fmaf(x[c], a, b);
  }
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int c = 0; c < CHAINS; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS> void run(int blocks, int threads) {
  float* out; unsigned long long* cyc; hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
  k<CHAINS><<<blocks, threads>>>(out, cyc, 0.999f, 0.001f); hipDeviceSynchronize();
  k<CHAINS><<<blocks, threads>>>(out, cyc, 0.999f, 0.001f); hipDeviceSynchronize();
  unsigned long long h[4]; hipMemcpy(h, cyc, 8 * (blocks < 4 ? blocks : 4), hipMemcpyDeviceToHost);
  printf("chains %d blocks %d threads %d: %.2f s_memtime ticks per FMA instruction (block 0)\n", CHAINS, blocks, threads, (double)h[0] / (256.0 * 16 * CHAINS));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1>(256, 64); run<2>(256, 64); run<4>(256, 64); run<8>(256, 64);
  run<1>(1024, 64); run<4>(1024, 64); run<1>(256, 32); run<4>(256, 32);
  run<1>(4096, 64); run<4>(4096, 64);
  return 0;
}
