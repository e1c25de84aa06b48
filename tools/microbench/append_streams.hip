// Write-side ceiling of the binned table backward (tools/microbench, run on the GPU box): many workgroups append short RUNS of
// records to 64 streams per "level" through atomic cursors -- exactly what bin_kernel does -- and nothing else.
//   layout SoA: 4 planes of 4-byte words (idx | ox | q0 | q1), AoS: 16-byte records
//   run   : records a workgroup appends to one stream per visit (bin_kernel: ~32)
//   align : reservations rounded up to a multiple of `align` records (1 = packed; 8 AoS records / 32 SoA words = one 128-B line)
// Prints useful GB/s (pad bytes not counted).    hipcc --offload-arch=gfx950 -O3 append_streams.hip -o append_streams.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kStreams = 64;

template <bool AOS>
__global__ __launch_bounds__(256) void append_kernel(unsigned* __restrict__ cursors, unsigned* __restrict__ idx, float* __restrict__ val,
                                                     int64_t n_rec_max, int run, int align, int L, int jitter) {
  __shared__ unsigned base[kStreams], cnt[kStreams];
  const int level = blockIdx.x % L;
  unsigned s = blockIdx.x * 2654435761u + 99u;
  if (threadIdx.x < kStreams) {
    s = (s + threadIdx.x) * 1664525u + 1013904223u;
    const unsigned c = run + (jitter ? (int)((s >> 8) % (unsigned)(2 * jitter + 1)) - jitter : 0);  // bin_kernel's buckets vary
    cnt[threadIdx.x] = c;
    const unsigned padded = (c + align - 1) / align * align;
    base[threadIdx.x] = atomicAdd(&cursors[level * kStreams + threadIdx.x], padded);
  }
  __syncthreads();
  // thread p writes record p of the workgroup's sorted staging area (here: synthetic), as bin_kernel's output loop does
  unsigned off = 0;
  for (int st = 0; st < kStreams; ++st) {
    const unsigned c = cnt[st];
    for (unsigned r = threadIdx.x; r < c; r += 256) {
      const int64_t dst = (int64_t)base[st] + r;
      if (AOS) {
        reinterpret_cast<float4*>(val)[dst] = make_float4(1.f, 2.f, 3.f, (float)r);
      } else {
        idx[dst] = r;
        val[dst] = 1.f;
        val[n_rec_max + dst] = 2.f;
        val[2 * n_rec_max + dst] = 3.f;
      }
    }
    off += c;
  }
}

// the same bytes, but every thread handles a flat position of the WORKGROUP's record list (coalesced across stream boundaries,
// 4 records in flight per thread), like bin_kernel's real output loop
template <bool AOS>
__global__ __launch_bounds__(256) void append_flat_kernel(unsigned* __restrict__ cursors, unsigned* __restrict__ idx, float* __restrict__ val,
                                                          int64_t n_rec_max, int run, int align, int L) {
  __shared__ unsigned base[kStreams];
  const int level = blockIdx.x % L;
  if (threadIdx.x < kStreams) base[threadIdx.x] = atomicAdd(&cursors[level * kStreams + threadIdx.x], (unsigned)((run + align - 1) / align * align));
  __syncthreads();
  const unsigned total = run * kStreams;
  for (unsigned p = threadIdx.x; p < total; p += 256) {
    const unsigned st = p / run, r = p % run;
    const int64_t dst = (int64_t)base[st] + r;
    if (AOS) {
      reinterpret_cast<float4*>(val)[dst] = make_float4(1.f, 2.f, 3.f, (float)r);
    } else {
      idx[dst] = r;
      val[dst] = 1.f;
      val[n_rec_max + dst] = 2.f;
      val[2 * n_rec_max + dst] = 3.f;
    }
  }
}

int main() {
  const int L = 16;
  const int64_t n_rec = 268435456;  // cfg-2 main field: 4.19 M points x 16 levels x 4 x-pairs
  const int64_t cap = 2 * n_rec + (1 << 20);
  unsigned *cursors, *idx;
  float* val;
  hipMalloc(&cursors, L * kStreams * 4);
  hipMalloc(&idx, cap * 4);
  hipMalloc(&val, cap * 16);
  std::vector<unsigned> starts(L * kStreams);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  printf("%-6s %-5s %5s %6s %7s %10s %10s\n", "layout", "loop", "run", "align", "jitter", "ms", "useful GB/s");
  for (int aos = 0; aos < 2; ++aos)
    for (int flat = 0; flat < 2; ++flat)
      for (int run : {16, 32, 64, 128})
        for (int align : {1, 8, 32}) {
          for (int jitter : {0, 8}) {
            if (flat && jitter) continue;
            if (jitter >= run || (run + jitter + align - 1) / align * align > 2 * run) continue;
            const int64_t per_stream = cap / (L * kStreams);
            for (int i = 0; i < L * kStreams; ++i) starts[i] = (unsigned)(i * per_stream / 32 * 32);
            const int64_t blocks = n_rec / ((int64_t)run * kStreams);
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
              hipMemcpy(cursors, starts.data(), starts.size() * 4, hipMemcpyHostToDevice);
              hipEventRecord(a);
              if (aos) {
                if (flat) append_flat_kernel<true><<<(unsigned)blocks, 256>>>(cursors, idx, val, cap, run, align, L);
                else append_kernel<true><<<(unsigned)blocks, 256>>>(cursors, idx, val, cap, run, align, L, jitter);
              } else {
                if (flat) append_flat_kernel<false><<<(unsigned)blocks, 256>>>(cursors, idx, val, cap, run, align, L);
                else append_kernel<false><<<(unsigned)blocks, 256>>>(cursors, idx, val, cap, run, align, L, jitter);
              }
              hipEventRecord(b);
              hipEventSynchronize(b);
              float ms;
              hipEventElapsedTime(&ms, a, b);
              if (ms < best) best = ms;
            }
            printf("%-6s %-5s %5d %6d %7d %10.3f %10.0f\n", aos ? "AoS" : "SoA", flat ? "flat" : "strm", run, align, jitter, best,
                   n_rec * 16.0 / best / 1e6);
          }
        }
  return 0;
}
