// What a wave-uniform decision costs a lone wave: K dependent-free VALU ops, then ballot(x) != 0 -> s_cbranch.
// Prints cycles per iteration for K = 8, 32 with and without the decision, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K, bool DECIDE>
__global__ void __launch_bounds__(64) kern(float* out, unsigned long long* cyc, int iters, float thr) {
  float a[8]; for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 0.001f + k;
  float acc = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < K; k++) a[k & 7] = __builtin_fmaf(a[k & 7], 1.0001f, 0.5f);
    if (DECIDE) {
      if (__builtin_amdgcn_ballot_w64(a[0] > thr) != 0ull) acc += a[1];      // never taken (thr huge), but the wave must decide
      if (__builtin_amdgcn_ballot_w64(a[2] < -thr) != 0ull) acc += a[3];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc; for (int k = 0; k < 8; k++) s += a[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int K, bool D> double run(float* out, unsigned long long* cyc, int iters) {
  hipLaunchKernelGGL((kern<K, D>), dim3(1024), dim3(64), 0, 0, out, cyc, iters, 1e30f);
  hipDeviceSynchronize();
  unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; i++) s += h[i]; return s / 1024 / iters;
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 1024 * 8);
  const int iters = 20000;
  run<8, false>(out, cyc, iters);
  printf("K=8  : %.1f cycles/iter plain, %.1f with two ballot decisions\n", run<8, false>(out, cyc, iters), run<8, true>(out, cyc, iters));
  printf("K=32 : %.1f cycles/iter plain, %.1f with two ballot decisions\n", run<32, false>(out, cyc, iters), run<32, true>(out, cyc, iters));
  return 0;
}
