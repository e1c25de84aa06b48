// Practical fp32 matrix peak of v_mfma_f32_16x16x4_f32 on this part (the roofline's 157.3 TFLOP/s assumes 2.4 GHz under load):
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// CHAINS independent accumulators per wave (1 = every MFMA depends on the previous one), 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x4 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64 / CHAINS; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678f) out[0] = s;
}

template <int CHAINS>
void run(int waves_per_simd, float* d) {
  const int threads = 256 * waves_per_simd, blocks = 256 * 4, iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  mfma_loop<CHAINS><<<blocks, threads>>>(d, 10, 1.0f, 1.0f);
  hipEventRecord(e0);
  mfma_loop<CHAINS><<<blocks, threads>>>(d, iters, 1.0f, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * (threads / 64) * iters * 64 * 2048.0;
  printf("chains %2d  waves/SIMD %d  %.3f ms  %.1f TFLOP/s\n", CHAINS, waves_per_simd, ms, flops / ms / 1e9);
}

int main() {
  float* d;
  hipMalloc(&d, 4);
  for (int w = 1; w <= 2; ++w) {
    run<1>(w, d);
    run<2>(w, d);
    run<4>(w, d);
    run<16>(w, d);
  }
  return 0;
}
