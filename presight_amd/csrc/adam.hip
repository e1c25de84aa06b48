// Fused Adam step over one parameter tensor (torch.optim.Adam semantics as configured by the
// reference: ns/engine/optimizers.py:133-140, ns/configs/method_configs.py:158-168 — lr 1e-2, betas (0.9, 0.999),
// eps 1e-15, L2 weight decay 1e-5 added to the gradient, no amsgrad).  One pass: 4 streams in (p, g, m, v), 3 out.
#include "common.hpp"

namespace {

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  const int64_t i0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4;
  if (i0 >= n) return;
  if (i0 + 3 < n) {
    f32x4 P = *reinterpret_cast<f32x4*>(p + i0), G = *reinterpret_cast<const f32x4*>(g + i0);
    f32x4 M = *reinterpret_cast<f32x4*>(m + i0), V = *reinterpret_cast<f32x4*>(v + i0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = G[k] + wd * P[k];
      M[k] = b1 * M[k] + (1.0f - b1) * gk;
      V[k] = b2 * V[k] + (1.0f - b2) * gk * gk;
      const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
      P[k] = P[k] - (lr / bc1) * (M[k] / denom);
    }
    *reinterpret_cast<f32x4*>(p + i0) = P;
    *reinterpret_cast<f32x4*>(m + i0) = M;
    *reinterpret_cast<f32x4*>(v + i0) = V;
  } else {
    for (int64_t i = i0; i < n; ++i) {
      const float gk = g[i] + wd * p[i];
      m[i] = b1 * m[i] + (1.0f - b1) * gk;
      v[i] = b2 * v[i] + (1.0f - b2) * gk * gk;
      const float denom = sqrtf(v[i]) / bc2_sqrt + eps;
      p[i] = p[i] - (lr / bc1) * (m[i] / denom);
    }
  }
}

}  // namespace

// step >= 1; bias corrections bc1 = 1 - b1^step, bc2 = 1 - b2^step are evaluated on the host in double precision
extern "C" int ps_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, void* stream) {
  if (n == 0) return 0;
  PS_REQUIRE(step >= 1, "ps_adam_step: step counts from 1");
  PS_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
             "ps_adam_step: buffers must be 16-byte aligned (presight_amd.dist.FlatGrads pads its views accordingly)");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const int64_t threads = (n + 3) / 4;
  adam_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps,
                                                                                 weight_decay, (float)bc1, (float)sqrt(bc2));
  PS_CHECK_LAUNCH();
}
