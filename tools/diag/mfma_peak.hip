// Diagnostic: peak rate of v_mfma_f32_32x32x2_f32 on this chip (pure register loop), for 1/2/3 waves per SIMD
// and 1 or 2 independent accumulators per wave.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
  float* out; hipMalloc(&out, 256 * 256 * 8 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int grid = 256 * blocks_per_cu;
  k<NACC><<<grid, 256>>>(out, 10);
  hipEventRecord(a);
  k<NACC><<<grid, 256>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double flops = (double)grid * 4 * iters * 16 * NACC * 4096.0;
  printf("acc=%d blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(1, 20000); run<1>(2, 10000); run<1>(3, 7000); run<2>(1, 10000); run<4>(1, 5000); run<4>(2, 2500);
  return 0;
}
