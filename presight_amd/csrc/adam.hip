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

// The same update over up to kMaxRanges disjoint ranges of one flat buffer in ONE launch, each range with its own bias
// corrections (torch.optim.Adam keeps state["step"] per parameter and advances it only when that parameter has a gradient:
// ns/engine/optimizers.py:133-140 with zero_grad(set_to_none=True), ns/engine/trainer.py:470).  The table travels as a
// kernel argument, so there is no host->device copy; a block finds its range by a linear search over the block prefix.
constexpr int kMaxRanges = 32;
constexpr int kRangeBlockElems = 256 * 4 * 4;  // elements per workgroup: 256 threads x 4 vectors of 4
struct AdamRanges {
  int n;
  int64_t start[kMaxRanges], count[kMaxRanges];
  unsigned blk0[kMaxRanges + 1];
  float bc1[kMaxRanges], bc2_sqrt[kMaxRanges];
};

__global__ __launch_bounds__(256) void adam_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, AdamRanges R, float lr, float b1, float b2, float eps,
                                                          float wd) {
  int r = 0;
  while (r + 1 < R.n && blockIdx.x >= R.blk0[r + 1]) ++r;
  const int64_t base = R.start[r], n = R.count[r];
  const float bc1 = R.bc1[r], bc2_sqrt = R.bc2_sqrt[r];
  const int64_t first = (int64_t)(blockIdx.x - R.blk0[r]) * kRangeBlockElems;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t i0 = first + (q * 256 + threadIdx.x) * 4;
    if (i0 >= n) continue;
    float* pp = p + base + i0;
    const float* gg = g + base + i0;
    float* mm = m + base + i0;
    float* vv = v + base + i0;
    if (i0 + 3 < n) {
      f32x4 P = *reinterpret_cast<f32x4*>(pp), G = *reinterpret_cast<const f32x4*>(gg);
      f32x4 M = *reinterpret_cast<f32x4*>(mm), V = *reinterpret_cast<f32x4*>(vv);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gk = G[k] + wd * P[k];
        M[k] = b1 * M[k] + (1.0f - b1) * gk;
        V[k] = b2 * V[k] + (1.0f - b2) * gk * gk;
        const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
        P[k] = P[k] - (lr / bc1) * (M[k] / denom);
      }
      *reinterpret_cast<f32x4*>(pp) = P;
      *reinterpret_cast<f32x4*>(mm) = M;
      *reinterpret_cast<f32x4*>(vv) = V;
    } else {
      for (int64_t i = 0; i0 + i < n; ++i) {
        const float gk = gg[i] + wd * pp[i];
        mm[i] = b1 * mm[i] + (1.0f - b1) * gk;
        vv[i] = b2 * vv[i] + (1.0f - b2) * gk * gk;
        const float denom = sqrtf(vv[i]) / bc2_sqrt + eps;
        pp[i] = pp[i] - (lr / bc1) * (mm[i] / denom);
      }
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

// n_ranges disjoint ranges [start[i], start[i]+count[i]) (floats, start a multiple of 4) of the flat buffers, range i at
// its own step count step[i] >= 1; start / count / step are HOST arrays.  ceil(n_ranges / 32) launches.
extern "C" int ps_adam_step_ranges(float* p, const float* g, float* m, float* v, int n_ranges, const int64_t* start,
                                   const int64_t* count, const int* step, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, void* stream) {
  PS_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "ps_adam_step_ranges: buffers must be 16-byte aligned");
  for (int r0 = 0; r0 < n_ranges; r0 += kMaxRanges) {
    AdamRanges R;
    R.n = 0;
    unsigned blocks = 0;
    for (int i = r0; i < n_ranges && R.n < kMaxRanges; ++i) {
      if (count[i] <= 0) continue;
      PS_REQUIRE(step[i] >= 1, "ps_adam_step_ranges: step counts from 1");
      PS_REQUIRE((start[i] & 3) == 0, "ps_adam_step_ranges: range starts must be multiples of 4 floats");
      const double bc1 = 1.0 - pow((double)beta1, (double)step[i]), bc2 = 1.0 - pow((double)beta2, (double)step[i]);
      R.start[R.n] = start[i];
      R.count[R.n] = count[i];
      R.blk0[R.n] = blocks;
      R.bc1[R.n] = (float)bc1;
      R.bc2_sqrt[R.n] = (float)sqrt(bc2);
      blocks += (unsigned)((count[i] + kRangeBlockElems - 1) / kRangeBlockElems);
      ++R.n;
    }
    if (R.n == 0) continue;
    R.blk0[R.n] = blocks;
    adam_ranges_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, R, lr, beta1, beta2, eps, weight_decay);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ps_set_error(hipGetErrorString(e)); return (int)e; }
  }
  return 0;
}
