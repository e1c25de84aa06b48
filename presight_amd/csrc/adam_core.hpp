// The Adam update of ONE element, shared by adam.hip (the optimizer kernels) and encode.hip (the table backward that applies it in
// its flush, ps_adam_fuse): one definition, compiled without fma contraction, so that both paths produce the same bits.
// torch.optim.Adam as the reference configures it (ns/engine/optimizers.py:73-170, ns/configs/method_configs.py:158-168): L2 weight
// decay added to the gradient, no amsgrad; bc1 = 1 - beta1^step, bc2_sqrt = sqrt(1 - beta2^step).
#pragma once

namespace ps {

struct AdamHyper {
  float lr, b1, b2, eps, wd, gs;  // gs: factor on the gradient before the weight decay is added (1 = the reference's default path)
};

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const AdamHyper& h, float bc1, float bc2_sqrt) {
#pragma clang fp contract(off)
  const float gk = g * h.gs + h.wd * p;
  m = h.b1 * m + (1.0f - h.b1) * gk;
  v = h.b2 * v + (1.0f - h.b2) * gk * gk;
  const float denom = sqrtf(v) / bc2_sqrt + h.eps;
  p = p - (h.lr / bc1) * (m / denom);
}

// bias corrections of a device-decided group (routed sub-field) at step count `step` (>= 1), evaluated like the host does
__device__ __forceinline__ void adam_bias_corrections(float b1, float b2, int step, float& bc1, float& bc2_sqrt) {
  const double st = (double)step;
  bc1 = (float)(1.0 - pow((double)b1, st));
  bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, st));
}

}  // namespace ps
