// Field-level hash-grid path: sample positions -> normalise/contract -> multires encode, and the
// table backward.  Data layout in HBM is "level planes": feat[l][n][f] (and dfeat likewise), so that
//   * the encode kernel (one workgroup = 256 points of ONE level) writes coalesced,
//   * the MFMA field kernels read 16 consecutive points of a plane per lane group,
//   * the backward scatter streams exactly one plane.
//
// Backward (ps_grid_scatter).  MI355X sustains only ~13-18 G fp32 global atomics/s (measured,
// profiles/r01_microbench.txt), 30x too slow for the ~10^9 corner contributions of one step, so the
// table gradient is NOT built with HBM atomics.  Instead every workgroup OWNS one slice of one
// level's table (<=128 KiB, resident in LDS), streams all points of that level, recomputes the 8
// corner hashes and accumulates only the corners that fall into its slice with LDS atomics
// (ds_add_f32); the slice is then written back with plain stores.  Work items are dealt to XCDs
// level-major, so the ~32 slice owners of a level share one XCD's L2 while they stream the same
// u / dfeat plane (HBM sees each plane once, L2 serves the other 31 readers).
//
// Reference semantics: ns/cameras/rays.py:49-58, ns/fields/PreSight/ingp_field.py:169-177,
// ns/field_components/encodings.py:343-384 (forward) and its autograd (index_put_ scatter-add).
#include "common.hpp"
#include "hashgrid_core.hpp"
#include "pointwise_core.hpp"

namespace {

__device__ __forceinline__ void xcd_item(int64_t n_items, int64_t& item, bool& valid) {
  const int64_t b = blockIdx.x;
  const int64_t per = (n_items + 7) / 8;
  const int64_t x = b & 7, i = b >> 3;
  item = x * per + i;
  valid = (i < per) && (item < n_items);
}

// u[n] = contract(normalise(position n)), sel[n] in {0,1}.  Positions are either given (pos != null)
// or generated from rays: point n = ray n/S, sample n%S.
__global__ void field_points_kernel(const float* __restrict__ pos, const float* __restrict__ origins,
                                    const float* __restrict__ dirs, const float* __restrict__ ebins, int S,
                                    const float* __restrict__ aabb, int contract, int64_t N, float* __restrict__ u,
                                    float* __restrict__ sel) {
#pragma clang fp contract(off)  // o + d*t rounded like torch (mul, then add): at 16384^3 resolution one ulp of u matters
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= N) return;
  float p[3];
  if (pos != nullptr) {
    p[0] = pos[n * 3];
    p[1] = pos[n * 3 + 1];
    p[2] = pos[n * 3 + 2];
  } else {
    const int64_t r = n / S;
    const int s = (int)(n % S);
    const float mid = (ebins[r * (S + 1) + s] + ebins[r * (S + 1) + s + 1]) / 2.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = origins[r * 3 + k] + dirs[r * 3 + k] * mid;
  }
  float q[3];
  const bool s_ = ps::normalize_contract(p[0], p[1], p[2], aabb, contract != 0, q);
  u[n * 3] = q[0];
  u[n * 3 + 1] = q[1];
  u[n * 3 + 2] = q[2];
  sel[n] = s_ ? 1.0f : 0.0f;
}

template <int F>
__global__ __launch_bounds__(256) void grid_encode_kernel(const float* __restrict__ u, const float* __restrict__ table,
                                                          const float* __restrict__ scalings, int L, int log2T, int64_t N,
                                                          int64_t plane_stride, float* __restrict__ feat) {
  const int64_t chunks = (N + 255) / 256;
  int64_t item;
  bool valid;
  xcd_item(chunks * L, item, valid);
  if (!valid) return;
  const int level = (int)(item / chunks);
  const int64_t n = (item % chunks) * 256 + threadIdx.x;
  if (n >= N) return;
  const uint32_t mask = (1u << log2T) - 1u;
  ps::Cell c = ps::make_cell(u[n * 3], u[n * 3 + 1], u[n * 3 + 2], scalings[level]);
  float v[F];
  ps::encode_level<F>(table + ((int64_t)level << log2T) * F, c, mask, v);
  float* o = feat + level * plane_stride + n * F;
  if constexpr (F == 1) o[0] = v[0];
  if constexpr (F == 2) *reinterpret_cast<f32x2*>(o) = (f32x2){v[0], v[1]};
  if constexpr (F == 4) *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
}

// ---- slice-owner scatter ---------------------------------------------------------------------
constexpr int kSliceBytes = 128 * 1024;
constexpr int kScatterThreads = 1024;

template <int F>
__global__ __launch_bounds__(kScatterThreads) void grid_scatter_kernel(const float* __restrict__ u,
                                                                       const float* __restrict__ dfeat,
                                                                       const float* __restrict__ scalings, int L, int log2T,
                                                                       int log2_slice, int64_t N, int64_t plane_stride,
                                                                       float* __restrict__ dtable, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float slice[];  // [entries][F]
  const int entries = 1 << log2_slice;
  const int n_slices = 1 << (log2T - log2_slice);
  int64_t item;
  bool valid;
  xcd_item((int64_t)L * n_slices, item, valid);
  if (!valid) return;
  const int level = (int)(item / n_slices);
  const uint32_t my_slice = (uint32_t)(item % n_slices);
  for (int i = threadIdx.x; i < entries * F; i += kScatterThreads) slice[i] = 0.0f;
  __syncthreads();
  const float s = scalings[level];
  const uint32_t mask = (1u << log2T) - 1u, low = (uint32_t)entries - 1u;
  const float* g_plane = dfeat + level * plane_stride;
  for (int64_t n = threadIdx.x; n < N; n += kScatterThreads) {
    ps::Cell c = ps::make_cell(u[n * 3], u[n * 3 + 1], u[n * 3 + 2], s);
    uint32_t h[8];
    ps::corner_hashes(c, mask, h);
    bool any = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) any |= ((h[k] >> log2_slice) == my_slice);
    if (!any) continue;
    float g[F];
    if constexpr (F == 1) g[0] = g_plane[n];
    if constexpr (F == 2) {
      const f32x2 t = *reinterpret_cast<const f32x2*>(g_plane + n * 2);
      g[0] = t.x;
      g[1] = t.y;
    }
    if constexpr (F == 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(g_plane + n * 4);
      g[0] = t.x;
      g[1] = t.y;
      g[2] = t.z;
      g[3] = t.w;
    }
    const float ox = c.ox, oy = c.oy, oz = c.oz, ux = 1.0f - ox, uy = 1.0f - oy, uz = 1.0f - oz;
    const float w[8] = {ox * oy * oz, ox * uy * oz, ux * uy * oz, ux * oy * oz,
                        ox * oy * uz, ox * uy * uz, ux * uy * uz, ux * oy * uz};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if ((h[k] >> log2_slice) == my_slice && w[k] != 0.0f) {
        float* p = slice + (h[k] & low) * F;
#pragma unroll
        for (int f = 0; f < F; ++f) atomicAdd(p + f, w[k] * g[f]);  // ds_add_f32
      }
    }
  }
  __syncthreads();
  float* out = dtable + (((int64_t)level << log2T) + ((int64_t)my_slice << log2_slice)) * F;
  if (accumulate) {
    for (int i = threadIdx.x; i < entries * F; i += kScatterThreads) out[i] += slice[i];
  } else {
    for (int i = threadIdx.x; i < entries * F; i += kScatterThreads) out[i] = slice[i];
  }
}

}  // namespace

extern "C" int ps_field_points(const float* pos, const float* origins, const float* dirs, const float* ebins, int S,
                               const float* aabb, int contract, int64_t N, float* u, float* sel, void* stream) {
  if (N == 0) return 0;
  PS_REQUIRE(pos != nullptr || (origins && dirs && ebins && S > 0), "ps_field_points: need positions or rays");
  field_points_kernel<<<(unsigned)((N + 255) / 256), 256, 0, (hipStream_t)stream>>>(pos, origins, dirs, ebins, S, aabb, contract,
                                                                                   N, u, sel);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_grid_encode(const float* u, const float* table, const float* scalings, int L, int F, int log2T, int64_t N,
                              int64_t plane_stride, float* feat, void* stream) {
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_grid_encode: features_per_level must be 1, 2 or 4");
  if (N == 0) return 0;
  const int64_t chunks = (N + 255) / 256;
  const int64_t per = (chunks * L + 7) / 8;
  dim3 grid((unsigned)(per * 8)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (F == 1) grid_encode_kernel<1><<<grid, block, 0, s>>>(u, table, scalings, L, log2T, N, plane_stride, feat);
  if (F == 2) grid_encode_kernel<2><<<grid, block, 0, s>>>(u, table, scalings, L, log2T, N, plane_stride, feat);
  if (F == 4) grid_encode_kernel<4><<<grid, block, 0, s>>>(u, table, scalings, L, log2T, N, plane_stride, feat);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_grid_scatter(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t N,
                               int64_t plane_stride, float* dtable, int accumulate, void* stream) {
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_grid_scatter: features_per_level must be 1, 2 or 4");
  int log2_slice = 0;
  while ((1 << (log2_slice + 1)) * F * 4 <= kSliceBytes) ++log2_slice;
  if (log2_slice > log2T) log2_slice = log2T;
  const int n_slices = 1 << (log2T - log2_slice);
  const int64_t items = (int64_t)L * n_slices;
  const int64_t per = (items + 7) / 8;
  const size_t lds = (size_t)(1 << log2_slice) * F * 4;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(per * 8)), block(kScatterThreads);
#define PS_LAUNCH_SCATTER(FF)                                                                                         \
  {                                                                                                                   \
    static bool attr_set = false;                                                                                     \
    if (!attr_set) {                                                                                                  \
      hipFuncSetAttribute((const void*)grid_scatter_kernel<FF>, hipFuncAttributeMaxDynamicSharedMemorySize, kSliceBytes); \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    grid_scatter_kernel<FF><<<grid, block, lds, s>>>(u, dfeat, scalings, L, log2T, log2_slice, N, plane_stride, dtable, \
                                                     accumulate);                                                    \
  }
  if (F == 1) PS_LAUNCH_SCATTER(1)
  if (F == 2) PS_LAUNCH_SCATTER(2)
  if (F == 4) PS_LAUNCH_SCATTER(4)
#undef PS_LAUNCH_SCATTER
  PS_CHECK_LAUNCH();
}
