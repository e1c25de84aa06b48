// Random-gather ceiling of MI355X for the hash-grid encode's access pattern (tools/microbench, run on the GPU box):
// every lane loads ROWB-byte rows at pseudo-random positions of a table of `bytes` bytes, U independent loads in flight per
// lane, PAIR = lanes 2i / 2i+1 read adjacent rows (the lane-paired encode: 32 distinct lines per wave instruction).
// Prints G rows/s and G distinct-64B-lines/s.   hipcc --offload-arch=gfx950 -O3 gather.hip -o gather.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int ROWB> struct Row;
template <> struct Row<4> { typedef float T; };
template <> struct Row<8> { typedef float2 T; };
template <> struct Row<16> { typedef float4 T; };
__device__ inline float sum(float v) { return v; }
__device__ inline float sum(float2 v) { return v.x + v.y; }
__device__ inline float sum(float4 v) { return v.x + v.y + v.z + v.w; }

template <int ROWB, int U, bool PAIR>
__global__ __launch_bounds__(256) void gather_kernel(const char* __restrict__ table, uint32_t row_mask, int iters, float* __restrict__ out) {
  typedef typename Row<ROWB>::T T;
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  if (PAIR) s = ((blockIdx.x * 256u + threadIdx.x) >> 1) * 2654435761u + 12345u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = s * 1664525u + 1013904223u;
      uint32_t r = (s >> 4) & row_mask;
      if (PAIR) r = (r & ~1u) | (threadIdx.x & 1u);
      v[u] = *reinterpret_cast<const T*>(table + (size_t)r * ROWB);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += sum(v[u]);
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int ROWB, int U, bool PAIR>
void run(const char* table, size_t bytes, float* out, int blocks, const char* tag) {
  const uint32_t rows = (uint32_t)(bytes / ROWB);
  const int iters = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  gather_kernel<ROWB, U, PAIR><<<blocks, 256>>>(table, rows - 1, 8, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  gather_kernel<ROWB, U, PAIR><<<blocks, 256>>>(table, rows - 1, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double n = (double)blocks * 256 * iters * U;
  const double lines = PAIR ? n / 2 : n;  // distinct 64-byte lines per row load (adjacent pair rows share one)
  printf("%-28s table %8.1f MiB  row %2d B  U=%2d  blocks %5d : %7.1f G rows/s  %7.1f G lines/s  (%6.2f TB/s of 64-B lines)\n", tag,
         bytes / 1048576.0, ROWB, U, blocks, n / ms / 1e6, lines / ms / 1e6, lines * 64 / ms / 1e9);
}

int main() {
  const size_t maxb = (size_t)2 << 30;
  char* table; float* out;
  hipMalloc(&table, maxb); hipMalloc(&out, 256);
  hipMemset(table, 1, maxb);
  for (size_t bytes : {(size_t)32 << 10, (size_t)4 << 20, (size_t)16 << 20, (size_t)160 << 20, (size_t)2 << 30}) {
    for (int blocks : {2048, 8192}) {
      run<8, 4, false>(table, bytes, out, blocks, "scattered");
      run<8, 8, false>(table, bytes, out, blocks, "scattered");
      run<8, 4, true>(table, bytes, out, blocks, "lane pairs");
      run<8, 8, true>(table, bytes, out, blocks, "lane pairs");
      run<4, 8, true>(table, bytes, out, blocks, "lane pairs");
      run<16, 4, true>(table, bytes, out, blocks, "lane pairs");
    }
  }
  return 0;
}
