// Streaming ceiling of the dense Adam update (4 streams in, 3 out, 28 B per parameter) on MI355X, for the 940 M parameters of the
// production tile:  hipcc --offload-arch=gfx950 -O3 adam_stream.hip -o adam_stream && ./adam_stream
// Variants: V per thread (16-byte vectors per array and thread), nontemporal loads / stores, grid-stride vs one tile per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, bool NT, bool STRIDE>
__global__ __launch_bounds__(256) void adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                            float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s) {
  const int64_t tile = 256LL * 4 * V;
  for (int64_t base = blockIdx.x * tile; base < n; base += STRIDE ? (int64_t)gridDim.x * tile : n) {
    f32x4 P[V], G[V], M[V], Vv[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int64_t i = base + (q * 256 + threadIdx.x) * 4;
      if (i < n) {
        if (NT) {
          P[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i));
          G[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
          M[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + i));
          Vv[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + i));
        } else {
          P[q] = *reinterpret_cast<const f32x4*>(p + i);
          G[q] = *reinterpret_cast<const f32x4*>(g + i);
          M[q] = *reinterpret_cast<const f32x4*>(m + i);
          Vv[q] = *reinterpret_cast<const f32x4*>(v + i);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int64_t i = base + (q * 256 + threadIdx.x) * 4;
      if (i < n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float gk = G[q][k] + wd * P[q][k];
          M[q][k] = b1 * M[q][k] + (1.0f - b1) * gk;
          Vv[q][k] = b2 * Vv[q][k] + (1.0f - b2) * gk * gk;
          P[q][k] = P[q][k] - (lr / bc1) * (M[q][k] / (sqrtf(Vv[q][k]) / bc2s + eps));
        }
        if (NT) {
          __builtin_nontemporal_store(P[q], reinterpret_cast<f32x4*>(p + i));
          __builtin_nontemporal_store(M[q], reinterpret_cast<f32x4*>(m + i));
          __builtin_nontemporal_store(Vv[q], reinterpret_cast<f32x4*>(v + i));
        } else {
          *reinterpret_cast<f32x4*>(p + i) = P[q];
          *reinterpret_cast<f32x4*>(m + i) = M[q];
          *reinterpret_cast<f32x4*>(v + i) = Vv[q];
        }
      }
    }
  }
}

template <int V, bool NT, bool STRIDE>
void run(const char* name, float* p, float* g, float* m, float* v, int64_t n, int blocks_per_cu) {
  const int64_t tile = 256LL * 4 * V;
  const unsigned grid = STRIDE ? 256u * blocks_per_cu : (unsigned)((n + tile - 1) / tile);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) adam<V, NT, STRIDE><<<grid, 256>>>(p, g, m, v, n, 1e-2f, 0.9f, 0.999f, 1e-15f, 1e-5f, 0.1f, 0.03f);
  hipEventRecord(a);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) adam<V, NT, STRIDE><<<grid, 256>>>(p, g, m, v, n, 1e-2f, 0.9f, 0.999f, 1e-15f, 1e-5f, 0.1f, 0.03f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  ms /= reps;
  printf("%-44s grid %8u  %.3f ms  %.2f TB/s\n", name, grid, ms, 28.0 * n / ms / 1e9);
}

int main() {
  const int64_t n = 940LL * 1000 * 1000 / 4096 * 4096;
  float *p, *g, *m, *v;
  hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4);
  hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
  run<4, false, false>("V=4 plain, one tile per workgroup (product)", p, g, m, v, n, 0);
  run<2, false, false>("V=2 plain, one tile per workgroup", p, g, m, v, n, 0);
  run<1, false, false>("V=1 plain, one tile per workgroup", p, g, m, v, n, 0);
  run<8, false, false>("V=8 plain, one tile per workgroup", p, g, m, v, n, 0);
  run<4, true, false>("V=4 nontemporal, one tile per workgroup", p, g, m, v, n, 0);
  run<2, true, false>("V=2 nontemporal, one tile per workgroup", p, g, m, v, n, 0);
  run<4, false, true>("V=4 plain, grid-stride 8 wg/CU", p, g, m, v, n, 8);
  run<4, false, true>("V=4 plain, grid-stride 16 wg/CU", p, g, m, v, n, 16);
  run<2, false, true>("V=2 plain, grid-stride 16 wg/CU", p, g, m, v, n, 16);
  run<4, true, true>("V=4 nontemporal, grid-stride 8 wg/CU", p, g, m, v, n, 8);
  run<1, false, true>("V=1 plain, grid-stride 32 wg/CU", p, g, m, v, n, 32);
  return 0;
}
