// Micro-benchmark: throughput of LDS / global atomic adds on gfx950 (float vs integer, dense vs sparse lanes).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomics.hip -o lds_atomics && ./lds_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE, int ACTIVE>
__global__ __launch_bounds__(256) void k(float* out, int iters, uint32_t seed) {
  extern __shared__ float lds[];  // 32768 floats = 128 KiB
  for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = 0.f;
  __syncthreads();
  uint32_t s = seed + blockIdx.x * 977 + threadIdx.x * 31;
  const int lane = threadIdx.x & 63;
  const bool act = (lane % (64 / ACTIVE)) == 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const uint32_t a = (s >> 8) & 32767u;
    if (act) {
      if (MODE == 0) atomicAdd(&lds[a], 1.0f);                                      // ds_add_f32
      if (MODE == 1) atomicAdd((unsigned*)&lds[a], 1u);                             // ds_add_u32
      if (MODE == 2) atomicAdd((unsigned long long*)&lds[a & 32766u], 1ull);        // ds_add_u64
      if (MODE == 3) lds[a] += 1.0f;                                                // plain RMW (racy)
      if (MODE == 4) { volatile float* p = lds; p[a] = 1.0f; }                       // plain store
    }
  }
  __syncthreads();
  float acc = 0;
  for (int i = threadIdx.x; i < 32768; i += 256) acc += lds[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
__global__ void g(float* buf, uint32_t mask, int iters, uint32_t seed) {
  uint32_t s = seed + (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const uint32_t a = (s >> 4) & mask;
    if (MODE == 0) unsafeAtomicAdd(&buf[a], 1.0f);
    if (MODE == 1) atomicAdd((unsigned*)&buf[a], 1u);
    if (MODE == 2) atomicAdd((unsigned long long*)&buf[a & ~1u], 1ull);
  }
}

template <class F>
float timeit(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * 4 * 4);
  const int iters = 20000;
  const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw", "plain store"};
#define RUN(MODE, ACT)                                                                                          \
  {                                                                                                             \
    hipFuncSetAttribute((const void*)k<MODE, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);          \
    float ms = timeit([&] { k<MODE, ACT><<<256, 256, 131072>>>(out, iters, 1234u); });                          \
    double ops = 256.0 * 256 * iters * ACT / 64.0;                                                              \
    printf("LDS %-12s active %2d/64: %8.3f ms  %8.1f G lane-ops/s chip  (%.2f cycles/wave-instr @2.1GHz/4 waves)\n", names[MODE], \
           ACT, ms, ops / ms / 1e6, ms * 1e-3 * 2.1e9 / iters);                                                 \
  }
  RUN(0, 64) RUN(0, 16) RUN(0, 2) RUN(1, 64) RUN(1, 16) RUN(1, 2) RUN(2, 64) RUN(2, 2) RUN(3, 64) RUN(4, 64)
  float* buf;
  hipMalloc(&buf, (64u << 20));
  hipMemset(buf, 0, 64u << 20);
  const char* gn[] = {"global_atomic_add_f32", "global_atomic_add_u32", "global_atomic_add_u64"};
#define RUNG(MODE, BITS)                                                                              \
  {                                                                                                   \
    float ms = timeit([&] { g<MODE><<<4096, 256>>>(buf, (1u << BITS) - 1, 256, 99u); });              \
    double ops = 4096.0 * 256 * 256;                                                                  \
    printf("%-24s over 2^%d words: %8.3f ms  %8.1f G atomics/s\n", gn[MODE], BITS, ms, ops / ms / 1e6); \
  }
  RUNG(0, 24) RUNG(1, 24) RUNG(2, 24) RUNG(0, 16) RUNG(1, 16)
  return 0;
}
