// Sub-field router of the *MS fields (ns/fields/PreSight/ingp_field_ms.py:97-126, prop_density_field_ms.py:90-102,
// sky_field_ms.py:97-114): cluster = argmin_k ||p - centroid_k||, then every sub-field evaluates its own points.  The
// reference does this with K boolean-mask gathers / scatters and K host synchronisations (torch.any); here the points are
// stably sorted by sub-field into the padded chunk layout of ms_core.hpp entirely on the device:
//   ms_route_kernel    nearest centroid per point + per-workgroup histogram            (4096 points per workgroup)
//   ms_scan_kernel     one workgroup per sub-field: exclusive scan of its histogram column, group size
//   ms_scatter_kernel  padded group starts from the K sizes; stable rank of every point inside its workgroup
//                      (wave ballots + LDS) -> perm[slot] = point; workgroup 0 also publishes field_start / chunk_field
// and ms_field_points_kernel normalises / contracts every sorted slot with ITS sub-field's AABB
// (ns/fields/PreSight/ingp_field.py:169-177).
#include "common.hpp"
#include "ms_core.hpp"
#include "pointwise_core.hpp"

namespace {

using namespace ps;

constexpr int kRouteThreads = 1024;
constexpr int kRoutePts = 4096;  // points per workgroup: 4 rounds of 1024

struct PointSrc {
  const float* pos;      // [N,3] or null
  const float* origins;  // rays: point n = ray n / S, sample n % S
  const float* dirs;
  const float* ebins;  // [R, S+1]
  int S;
};

__device__ __forceinline__ void load_point(const PointSrc& s, int64_t n, float (&p)[3]) {
#pragma clang fp contract(off)  // o + d*t rounded like torch (mul, then add), as in field_points_kernel
  if (s.pos != nullptr) {
    p[0] = s.pos[n * 3];
    p[1] = s.pos[n * 3 + 1];
    p[2] = s.pos[n * 3 + 2];
  } else {
    const int64_t r = n / s.S;
    const int i = (int)(n % s.S);
    const float mid = (s.ebins[r * (s.S + 1) + i] + s.ebins[r * (s.S + 1) + i + 1]) / 2.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = s.origins[r * 3 + k] + s.dirs[r * 3 + k] * mid;
  }
}

__global__ __launch_bounds__(kRouteThreads) void ms_route_kernel(PointSrc src, int64_t N, const float* __restrict__ centroids, int K,
                                                                 int* __restrict__ assign, int* __restrict__ hist) {
  __shared__ float c[kMsMaxFields * 3];
  __shared__ int h[kMsMaxFields];
  for (int i = threadIdx.x; i < K * 3; i += kRouteThreads) c[i] = centroids[i];
  if (threadIdx.x < K) h[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kRoutePts / kRouteThreads; ++q) {
    const int64_t n = (int64_t)blockIdx.x * kRoutePts + q * kRouteThreads + threadIdx.x;
    if (n < N) {
      float p[3];
      load_point(src, n, p);
      const int k = nearest_centroid(p[0], p[1], p[2], c, K);
      assign[n] = k;
      atomicAdd(&h[k], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < K) hist[(int64_t)blockIdx.x * K + threadIdx.x] = h[threadIdx.x];
}

// workgroup k: blk_off[b][k] = sum_{b' < b} hist[b'][k] (in place), totals[k] = group size
__global__ __launch_bounds__(1024) void ms_scan_kernel(int* __restrict__ hist, int nblk, int K, int* __restrict__ totals) {
  __shared__ int part[1024];
  const int k = blockIdx.x;
  const int per = (nblk + 1023) / 1024;
  const int b0 = threadIdx.x * per;
  int sum = 0;
  for (int i = 0; i < per; ++i)
    if (b0 + i < nblk) sum += hist[(int64_t)(b0 + i) * K + k];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = (threadIdx.x >= (unsigned)d) ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - sum;
  for (int i = 0; i < per; ++i)
    if (b0 + i < nblk) {
      const int v = hist[(int64_t)(b0 + i) * K + k];
      hist[(int64_t)(b0 + i) * K + k] = run;
      run += v;
    }
  if (threadIdx.x == 1023) totals[k] = part[1023];
}

__global__ __launch_bounds__(kRouteThreads) void ms_scatter_kernel(const int* __restrict__ assign, const int* __restrict__ blk_off,
                                                                   const int* __restrict__ totals, int64_t N, int K, int max_chunks,
                                                                   int* __restrict__ field_start, int* __restrict__ chunk_field,
                                                                   int* __restrict__ perm) {
  __shared__ int start_chunk[kMsMaxFields + 1];
  __shared__ int base[kMsMaxFields];                            // slot of this workgroup's first point of sub-field k
  __shared__ int cnt[kRouteThreads / 64][kMsMaxFields];         // per-wave counts of the current round
  if (threadIdx.x == 0) {
    int s = 0;
    for (int k = 0; k < K; ++k) {
      start_chunk[k] = s;
      s += (totals[k] + kMsChunk - 1) / kMsChunk;
    }
    start_chunk[K] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) base[threadIdx.x] = start_chunk[threadIdx.x] * kMsChunk + blk_off[(int64_t)blockIdx.x * K + threadIdx.x];
  if (blockIdx.x == 0) {  // publish the layout for the kernels that follow on the stream
    if (threadIdx.x <= K) field_start[threadIdx.x] = start_chunk[threadIdx.x];
    for (int c = threadIdx.x; c < max_chunks; c += kRouteThreads) {
      int f = -1;
      for (int k = 0; k < K; ++k)
        if (c >= start_chunk[k] && c < start_chunk[k + 1]) f = k;
      chunk_field[c] = f;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = 0; q < kRoutePts / kRouteThreads; ++q) {
    for (int i = threadIdx.x; i < (kRouteThreads / 64) * kMsMaxFields; i += kRouteThreads) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int64_t n = (int64_t)blockIdx.x * kRoutePts + q * kRouteThreads + threadIdx.x;
    const bool valid = n < N;
    const int myk = valid ? assign[n] : -1;
    // lanes of the wave that go to the same sub-field, by rounds of ballots (1-2 rounds for ray-coherent points)
    unsigned long long mine = 0ull;
    bool done = !valid;
    while (true) {
      const unsigned long long active = __ballot(!done);
      if (active == 0ull) break;
      const int leader = __ffsll((long long)active) - 1;
      const int k0 = __shfl(myk, leader, 64);
      const unsigned long long m = __ballot(!done && myk == k0);
      if (!done && myk == k0) {
        mine = m;
        done = true;
      }
    }
    const int rank = __popcll(mine & ((1ull << lane) - 1ull));
    if (valid && rank == 0) cnt[wave][myk] = __popcll(mine);
    __syncthreads();
    if (valid) {
      int before = 0;
      for (int w = 0; w < wave; ++w) before += cnt[w][myk];
      perm[base[myk] + before + rank] = (int)n;
    }
    __syncthreads();
    if (threadIdx.x < K) {
      int s = 0;
      for (int w = 0; w < kRouteThreads / 64; ++w) s += cnt[w][threadIdx.x];
      base[threadIdx.x] += s;
    }
    __syncthreads();
  }
}

// every slot of the sorted layout: u = contract(normalise(position, AABB of the slot's sub-field)), sel; padding slots
// (perm < 0) get u = 0, sel = 0; slots of unused chunks are not written
__global__ void ms_field_points_kernel(PointSrc src, const int* __restrict__ perm, const int* __restrict__ chunk_field,
                                       const float* __restrict__ aabbs, int contract, int64_t n_slots, float* __restrict__ u,
                                       float* __restrict__ sel) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  const int k = chunk_field[i / kMsChunk];
  if (k < 0) return;
  const int n = perm[i];
  float q[3] = {0.f, 0.f, 0.f};
  bool s = false;
  if (n >= 0) {
    float p[3];
    load_point(src, n, p);
    s = normalize_contract(p[0], p[1], p[2], aabbs + k * 6, contract != 0, q);
  }
  u[i * 3] = q[0];
  u[i * 3 + 1] = q[1];
  u[i * 3 + 2] = q[2];
  sel[i] = s ? 1.0f : 0.0f;
}

// out[perm[i]] = in[i] (rows of `width` floats) for the occupied slots: un-sort of a per-point result
__global__ void ms_unsort_kernel(const float* __restrict__ in, const int* __restrict__ perm, int64_t n_slots, int width,
                                 float* __restrict__ out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t i = t / width;
  if (i >= n_slots) return;
  const int n = perm[i];
  if (n >= 0) out[(int64_t)n * width + (t - i * width)] = in[t];
}

struct Layout {
  int64_t chunks, slots, off_fs, off_cf, off_hist, off_tot, off_assign, ints;
  int nblk;
};
Layout ms_layout(int64_t N, int K) {
  Layout l;
  l.chunks = (N + kMsChunk - 1) / kMsChunk + K;
  l.slots = l.chunks * kMsChunk;
  l.nblk = (int)((N + kRoutePts - 1) / kRoutePts);
  l.off_fs = 0;
  l.off_cf = 80;
  l.off_tot = l.off_cf + ((l.chunks + 15) & ~(int64_t)15);
  l.off_hist = l.off_tot + 64;
  l.off_assign = l.off_hist + (((int64_t)l.nblk * K + 15) & ~(int64_t)15);
  l.ints = l.off_assign + ((N + 15) & ~(int64_t)15);
  return l;
}

}  // namespace

extern "C" int ps_ms_chunk(void) { return ps::kMsChunk; }

extern "C" int ps_ms_layout(int64_t N, int K, int64_t* out) {
  PS_REQUIRE(K >= 1 && K <= ps::kMsMaxFields, "ps_ms_layout: 1..64 sub-fields");
  const Layout l = ms_layout(N, K);
  out[0] = l.ints;
  out[1] = l.off_fs;
  out[2] = l.off_cf;
  out[3] = l.slots;
  out[4] = l.chunks;
  return 0;
}

extern "C" int ps_ms_route(const float* pos, const float* origins, const float* dirs, const float* ebins, int S, int64_t N,
                           const float* centroids, int K, int32_t* plan, int32_t* perm, void* stream) {
  PS_REQUIRE(K >= 1 && K <= ps::kMsMaxFields, "ps_ms_route: 1..64 sub-fields");
  PS_REQUIRE(N == 0 || pos != nullptr || (origins && dirs && ebins && S > 0), "ps_ms_route: need positions or rays");
  PS_REQUIRE(N < ((int64_t)1 << 31) - ps::kMsChunk * (int64_t)(K + 1), "ps_ms_route: point count must fit 32-bit slots");
  hipStream_t s = (hipStream_t)stream;
  const Layout l = ms_layout(N, K);
  hipError_t e = hipMemsetAsync(perm, 0xFF, (size_t)l.slots * 4, s);
  if (e != hipSuccess) { ps_set_error(hipGetErrorString(e)); return (int)e; }
  int* field_start = plan + l.off_fs;
  int* chunk_field = plan + l.off_cf;
  int* totals = plan + l.off_tot;
  int* hist = plan + l.off_hist;
  int* assign = plan + l.off_assign;
  if (N == 0) {
    e = hipMemsetAsync(field_start, 0, (size_t)(K + 1) * 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(chunk_field, 0xFF, (size_t)l.chunks * 4, s);
    if (e != hipSuccess) { ps_set_error(hipGetErrorString(e)); return (int)e; }
    return 0;
  }
  const PointSrc src{pos, origins, dirs, ebins, S};
  ms_route_kernel<<<l.nblk, kRouteThreads, 0, s>>>(src, N, centroids, K, assign, hist);
  ms_scan_kernel<<<K, 1024, 0, s>>>(hist, l.nblk, K, totals);
  ms_scatter_kernel<<<l.nblk, kRouteThreads, 0, s>>>(assign, hist, totals, N, K, (int)l.chunks, field_start, chunk_field, perm);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_ms_field_points(const float* pos, const float* origins, const float* dirs, const float* ebins, int S,
                                  const float* aabbs, int contract, int64_t N, int K, const int32_t* plan, const int32_t* perm, float* u,
                                  float* sel, void* stream) {
  PS_REQUIRE(N == 0 || pos != nullptr || (origins && dirs && ebins && S > 0), "ps_ms_field_points: need positions or rays");
  const Layout l = ms_layout(N, K);
  const PointSrc src{pos, origins, dirs, ebins, S};
  ms_field_points_kernel<<<(unsigned)((l.slots + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, perm, plan + l.off_cf, aabbs, contract,
                                                                                          l.slots, u, sel);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_ms_unsort(const float* sorted, const int32_t* perm, int64_t n_slots, int width, float* out, void* stream) {
  if (n_slots == 0) return 0;
  ms_unsort_kernel<<<(unsigned)((n_slots * width + 255) / 256), 256, 0, (hipStream_t)stream>>>(sorted, perm, n_slots, width, out);
  PS_CHECK_LAUNCH();
}
