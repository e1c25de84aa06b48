// Prior-extraction helpers (BASELINE config 5):
//   ps_voxel_index   : Open3D voxel_down_sample_and_trace index rule used by ns/scripts/extract_priors.py:216-245,
//                      idx = floor((p - (min_bound - voxel/2)) / voxel) per axis, evaluated in fp64 -> int64 (bit exact)
//   ps_lattice_points: the dense res^3 query lattice over a tile AABB (z fastest), cell centres
//   ps_mean_density  : mean of the proposal-net and main-field densities (extract_priors.py:133-137)
#include "common.hpp"

namespace {

__global__ void voxel_index_kernel(const float* __restrict__ pts, int64_t n, double voxel, double mx, double my, double mz,
                                   int64_t* __restrict__ idx) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double h = voxel * 0.5;
  idx[i * 3 + 0] = (int64_t)floor(((double)pts[i * 3 + 0] - (mx - h)) / voxel);
  idx[i * 3 + 1] = (int64_t)floor(((double)pts[i * 3 + 1] - (my - h)) / voxel);
  idx[i * 3 + 2] = (int64_t)floor(((double)pts[i * 3 + 2] - (mz - h)) / voxel);
}

__global__ void lattice_points_kernel(float lox, float loy, float loz, float hix, float hiy, float hiz, int res, int64_t start,
                                      int64_t count, float* __restrict__ pts) {
#pragma clang fp contract(off)
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int64_t n = start + i;
  const int iz = (int)(n % res), iy = (int)((n / res) % res), ix = (int)(n / ((int64_t)res * res));
  const float inv = 1.0f / (float)res;
  pts[i * 3 + 0] = lox + (hix - lox) * (((float)ix + 0.5f) * inv);
  pts[i * 3 + 1] = loy + (hiy - loy) * (((float)iy + 0.5f) * inv);
  pts[i * 3 + 2] = loz + (hiz - loz) * (((float)iz + 0.5f) * inv);
}

__global__ void mean3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int64_t n,
                             float* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = ((a[i] + b[i]) + c[i]) / 3.0f;  // torch.stack([p0, p1, main]).mean(0): sequential sum then divide
}

}  // namespace

extern "C" int ps_voxel_index(const float* pts, int64_t n, double voxel, const double* min_bound /*host[3]*/, int64_t* idx,
                              void* stream) {
  if (n == 0) return 0;
  PS_REQUIRE(voxel > 0.0, "ps_voxel_index: voxel size must be positive");
  voxel_index_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(pts, n, voxel, min_bound[0], min_bound[1],
                                                                                  min_bound[2], idx);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_lattice_points(const float* aabb /*host[6]: min xyz, max xyz*/, int res, int64_t start, int64_t count,
                                 float* pts, void* stream) {
  if (count == 0) return 0;
  lattice_points_kernel<<<(unsigned)((count + 255) / 256), 256, 0, (hipStream_t)stream>>>(aabb[0], aabb[1], aabb[2], aabb[3],
                                                                                         aabb[4], aabb[5], res, start, count, pts);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_mean_density(const float* a, const float* b, const float* c, int64_t n, float* out, void* stream) {
  if (n == 0) return 0;
  mean3_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(a, b, c, n, out);
  PS_CHECK_LAUNCH();
}
