// Device functions for the point-wise parts of a field evaluation, shared with the fused kernels.
#pragma once
#include "common.hpp"

namespace ps {

// (p - aabb_min)/(aabb_max - aabb_min)*2-1 -> L-inf contraction -> (x+2)/4 -> selector -> x*selector
// aabb = {min.xyz, max.xyz}.  Returns the selector.
__device__ __forceinline__ bool normalize_contract(float px, float py, float pz, const float* __restrict__ aabb,
                                                   bool contract, float (&u)[3]) {
#pragma clang fp contract(off)  // keep the reference's separately rounded mul/add so that u is bit-identical to torch
  float q[3] = {(px - aabb[0]) / (aabb[3] - aabb[0]), (py - aabb[1]) / (aabb[4] - aabb[1]),
                (pz - aabb[2]) / (aabb[5] - aabb[2])};
  if (contract) {
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = q[k] * 2.0f - 1.0f;
    const float mag = fmaxf(fabsf(q[0]), fmaxf(fabsf(q[1]), fabsf(q[2])));
    if (!(mag < 1.0f)) {
      const float sc = 2.0f - (1.0f / mag);
#pragma unroll
      for (int k = 0; k < 3; ++k) q[k] = sc * (q[k] / mag);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = (q[k] + 2.0f) / 4.0f;
  }
  const bool sel = (q[0] > 0.0f) && (q[0] < 1.0f) && (q[1] > 0.0f) && (q[1] < 1.0f) && (q[2] > 0.0f) && (q[2] < 1.0f);
#pragma unroll
  for (int k = 0; k < 3; ++k) u[k] = sel ? q[k] : 0.0f;
  return sel;
}

// torch.nan_to_num of a float (nan -> 0, +-inf -> +-FLT_MAX)
__device__ __forceinline__ float nan_to_num(float v) {
  if (isnan(v)) return 0.0f;
  if (isinf(v)) return v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  return v;
}

// real spherical harmonics, 4 levels (16 components), ns/utils/math.py:53-79
__device__ __forceinline__ void sh4(float x, float y, float z, float (&o)[16]) {
  const float xx = x * x, yy = y * y, zz = z * z;
  o[0] = 0.28209479177387814f;
  o[1] = 0.4886025119029199f * y;
  o[2] = 0.4886025119029199f * z;
  o[3] = 0.4886025119029199f * x;
  o[4] = 1.0925484305920792f * x * y;
  o[5] = 1.0925484305920792f * y * z;
  o[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
  o[7] = 1.0925484305920792f * x * z;
  o[8] = 0.5462742152960396f * (xx - yy);
  o[9] = 0.5900435899266435f * y * (3.0f * xx - yy);
  o[10] = 2.890611442640554f * x * y * z;
  o[11] = 0.4570457994644658f * y * (5.0f * zz - 1.0f);
  o[12] = 0.3731763325901154f * z * (5.0f * zz - 3.0f);
  o[13] = 0.4570457994644658f * x * (5.0f * zz - 1.0f);
  o[14] = 1.445305721320277f * z * (xx - yy);
  o[15] = 0.5900435899266435f * x * (xx - 3.0f * yy);
}

// argmin_k ||p - c_k||_2 (first minimum wins, like torch.argmin)
__device__ __forceinline__ int nearest_centroid(float px, float py, float pz, const float* __restrict__ c, int K) {
  int best = 0;
  float bd = 3.4e38f;
  for (int k = 0; k < K; ++k) {
    const float dx = px - c[k * 3], dy = py - c[k * 3 + 1], dz = pz - c[k * 3 + 2];
    const float d = dx * dx + dy * dy + dz * dz;
    if (d < bd) {
      bd = d;
      best = k;
    }
  }
  return best;
}

}  // namespace ps
