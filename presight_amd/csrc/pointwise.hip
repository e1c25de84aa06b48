// Point-wise operators: AABB normalisation + L-inf scene contraction + selector (a6), degree-4
// spherical harmonics on (d+1)/2 (a10), nearest-centroid router (a5), sample positions (a4).
//   ns/fields/PreSight/utils.py:6-10, ns/field_components/spatial_distortions.py:66-69,
//   ns/fields/PreSight/ingp_field.py:169-177, ns/utils/math.py:27-79, ns/fields/base_field.py:136-142,
//   ns/fields/PreSight/ingp_field_ms.py:97, ns/cameras/rays.py:49-58
#include "common.hpp"
#include "pointwise_core.hpp"

namespace {

__global__ void contract_kernel(const float* __restrict__ p, const float* __restrict__ aabb, int64_t M, int contract,
                                float* __restrict__ u, uint8_t* __restrict__ sel) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= M) return;
  float q[3];
  const bool s = ps::normalize_contract(p[i * 3], p[i * 3 + 1], p[i * 3 + 2], aabb, contract != 0, q);
  u[i * 3 + 0] = q[0];
  u[i * 3 + 1] = q[1];
  u[i * 3 + 2] = q[2];
  if (sel) sel[i] = s ? 1 : 0;
}

__global__ void sh4_kernel(const float* __restrict__ d, int64_t M, float* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= M) return;
  float sh[16];
  ps::sh4((d[i * 3] + 1.0f) / 2.0f, (d[i * 3 + 1] + 1.0f) / 2.0f, (d[i * 3 + 2] + 1.0f) / 2.0f, sh);
  f32x4* o = reinterpret_cast<f32x4*>(out + i * 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (f32x4){sh[4 * k], sh[4 * k + 1], sh[4 * k + 2], sh[4 * k + 3]};
}

// SH of the input AS GIVEN: SHEncoding.pytorch_fwd evaluates components_from_spherical_harmonics on whatever it is handed
// (ns/field_components/encodings.py:711-714); the fields hand it (d+1)/2
__global__ void sh4_raw_kernel(const float* __restrict__ x, int64_t M, int n_out, float* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= M) return;
  float sh[16];
  ps::sh4(x[i * 3], x[i * 3 + 1], x[i * 3 + 2], sh);
  for (int k = 0; k < n_out; ++k) out[i * n_out + k] = sh[k];
}

__global__ void route_kernel(const float* __restrict__ p, int64_t M, const float* __restrict__ centroids, int K,
                             int32_t* __restrict__ assign) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= M) return;
  assign[i] = ps::nearest_centroid(p[i * 3], p[i * 3 + 1], p[i * 3 + 2], centroids, K);
}

__global__ void positions_kernel(const float* __restrict__ origins, const float* __restrict__ dirs,
                                 const float* __restrict__ ebins, int64_t R, int S, float* __restrict__ pos) {
#pragma clang fp contract(off)
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= R * S) return;
  const int64_t r = i / S;
  const int s = (int)(i % S);
  const float mid = (ebins[r * (S + 1) + s] + ebins[r * (S + 1) + s + 1]) / 2.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) pos[i * 3 + k] = origins[r * 3 + k] + dirs[r * 3 + k] * mid;
}

}  // namespace

extern "C" int ps_contract(const float* p, const float* aabb, int64_t M, int contract, float* u, uint8_t* sel, void* stream) {
  if (M == 0) return 0;
  contract_kernel<<<(unsigned)((M + 255) / 256), 256, 0, (hipStream_t)stream>>>(p, aabb, M, contract, u, sel);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_sh4(const float* dirs, int64_t M, float* out, void* stream) {
  if (M == 0) return 0;
  sh4_kernel<<<(unsigned)((M + 255) / 256), 256, 0, (hipStream_t)stream>>>(dirs, M, out);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_sh_encode(const float* x, int64_t M, int levels, float* out, void* stream) {
  if (M == 0) return 0;
  PS_REQUIRE(levels >= 1 && levels <= 4, "ps_sh_encode: 1..4 spherical-harmonic levels");
  sh4_raw_kernel<<<(unsigned)((M + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, M, levels * levels, out);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_route(const float* p, int64_t M, const float* centroids, int K, int32_t* assign, void* stream) {
  if (M == 0) return 0;
  route_kernel<<<(unsigned)((M + 255) / 256), 256, 0, (hipStream_t)stream>>>(p, M, centroids, K, assign);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_sample_positions(const float* origins, const float* dirs, const float* ebins, int64_t R, int S, float* pos,
                                   void* stream) {
  if (R == 0) return 0;
  positions_kernel<<<(unsigned)((R * S + 255) / 256), 256, 0, (hipStream_t)stream>>>(origins, dirs, ebins, R, S, pos);
  PS_CHECK_LAUNCH();
}
