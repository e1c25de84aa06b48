// Fused per-ray kernels around the factored training render node (round 6).  Between the main field's forward kernel and its
// backward kernels a cfg-2 step used to run ~45 small launches (compositing, the per-ray output layer of the semantic head, sky
// blending, three mean losses with their finish / chain-rule launches, the weights' backward); a dependent launch costs ~6 us of
// timeline on MI355X whatever its size, and most of these kernels stream the same [R, <= 64] rows.  Here:
//
//   ps_ray_out_fwd       compositing of the colour branch (rgb, accumulation) + semantic output layer, one wavefront per ray
//                        = ps_composite_fwd (rgb, acc) + ps_sem_out_fwd, same arithmetic in the same order
//                        (ns/model_components/renderers.py:70-117,286-314; ns/fields/PreSight/ingp_field.py:143-151)
//   ps_ray_dsigma_bwd    d(weights) of the compositing + RaySamples.get_weights' backward -> d(density), one wavefront per ray
//                        = ps_composite_bwd (weights only, S <= 64) + ps_weights_bwd; d(weights) never reaches memory
//                        (ns/cameras/rays.py:128-150)
//   ps_blend_losses      sky blending + rgb MSE + sky BCE + semantic MSE, values AND the gradients w.r.t. every input of the blend,
//                        in ONE launch: the three terms are linear in the seed of the backward pass, which the trainer knows
//                        before the forward runs (ns/models/PreSight/nerfacto_nusc_ms.py:512-533,558-575;
//                        ns/model_components/PreSight/losses.py:106-125)
//   ps_finish_losses     the scalar values of ALL loss terms of a step (scale * sum / count each, as ps_loss_finish forms them) and
//                        their sum (functools.reduce(torch.add, loss_dict.values()), ns/engine/trainer.py:478) in one launch
//
// (Measured and dropped on the way: finishing each loss inside its own kernel -- "the last workgroup to take a ticket sums the
//  terms".  The device-scope fence every workgroup needs before its ticket writes back its XCD's L2 each time: interlevel 0.12 ->
//  1.5 ms with 16 384 workgroups.  Kernel boundaries are the cheap fences; the finish is one small launch behind all loss kernels.)
#include "common.hpp"

namespace {

constexpr int kC = 64;

__global__ __launch_bounds__(256) void ray_out_fwd_kernel(const float* __restrict__ weights, const float* __restrict__ rgb_s,
                                                          const float* __restrict__ H, const float* __restrict__ W,
                                                          const float* __restrict__ b, int64_t R, int S, float* __restrict__ rgb,
                                                          float* __restrict__ acc, float* __restrict__ sem) {
  __shared__ float Wt[kC][kC + 1];  // Wt[k][c] = W[c][k]
  for (int i = threadIdx.x; i < kC * kC; i += 256) Wt[i % kC][i / kC] = W[i];
  __syncthreads();
  const int lane = ps_lane(), wave = threadIdx.x >> 6;
  const float bias = b[lane];
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < R; r += (int64_t)gridDim.x * 4) {
    const float* w = weights + r * S;
    const float h = H[r * kC + lane];
    // colour: lane = sample, one wave reduction per channel (composite_ray<false, 1> in render.hip)
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int sr = lane; sr < S; sr += 64) {
      const float* pr = rgb_s + (r * S + sr) * 3;
      const float wv = w[sr];
      c0 += wv * pr[0];
      c1 += wv * pr[1];
      c2 += wv * pr[2];
    }
    c0 = ps_wave_sum(c0);
    c1 = ps_wave_sum(c1);
    c2 = ps_wave_sum(c2);
    if (lane < 3) rgb[r * 3 + lane] = lane == 0 ? c0 : (lane == 1 ? c1 : c2);
    // accumulation: the last lane of the inclusive scan, as the compositing kernel forms it
    const float wl = lane < S ? w[lane] : 0.0f;
    const float total = __shfl(ps_wave_incl_scan(wl), 63, 64);
    if (lane == 0) acc[r] = total;
    // semantic output layer on the composited hidden activations (sem_out_fwd_kernel in factored.hip)
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kC; ++k) s = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, h), k)), Wt[k][lane], s);
    sem[r * kC + lane] = s + bias * total;
  }
}

// lane = sample (S <= 64).  d(weights)[s] = <rgb_s[s], d_rgb> + d_acc + d_acc2 + add0[s] + add1[s] in the order of
// composite_bwd_w_kernel<false>; then weights_bwd_kernel<1>.
__global__ __launch_bounds__(256) void ray_dsigma_bwd_kernel(const float* __restrict__ ebins, const float* __restrict__ sigma,
                                                             const float* __restrict__ rgb_s, const float* __restrict__ d_rgb,
                                                             const float* __restrict__ d_acc, const float* __restrict__ d_acc2,
                                                             const float* __restrict__ add0, const float* __restrict__ add1,
                                                             int64_t R, int S, float* __restrict__ dsigma) {
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int s = ps_lane();
  const bool ok = s < S;
  const float* e = ebins + ray * (S + 1);
  float gr = 0.0f;
  if (ok && rgb_s != nullptr && d_rgb != nullptr) {
    const float* pr = rgb_s + (ray * S + s) * 3;
    gr += pr[0] * d_rgb[ray * 3] + pr[1] * d_rgb[ray * 3 + 1] + pr[2] * d_rgb[ray * 3 + 2];
  }
  float ga = d_acc ? d_acc[ray] : 0.0f;
  if (d_acc2 != nullptr) ga = d_acc ? ga + d_acc2[ray] : d_acc2[ray];
  gr += ga;
  if (ok) {
    if (add0 != nullptr) gr += add0[ray * S + s];
    if (add1 != nullptr) gr += add1[ray * S + s];
  }
  const float gw = ok ? gr : 0.0f;
  const float delta = ok ? (e[s + 1] - e[s]) : 0.0f;
  const float dd = ok ? delta * sigma[ray * S + s] : 0.0f;
  const float excl = ps_wave_incl_scan(dd) - dd;
  const float T = expf(-excl), ed = expf(-dd);
  const float w = (1.0f - ed) * T;
  const bool fin = isfinite(w);
  const float a = fin ? gw * w : 0.0f;
  const float tr_after = fin ? gw * ed * T : 0.0f;  // d w_k / d dd_k
  const float incl = ps_wave_incl_scan(a);
  const float later = __shfl(incl, 63, 64) - incl;  // sum of a over all later samples
  if (ok) dsigma[ray * S + s] = (tr_after - later) * delta;
}

struct BlendLossArgs {
  const float *rgb_f, *acc_raw, *sem_f, *sky_rgb, *sky_sem;  // [R,3] [R] [R,C] [R,3] [R,C]; sem_f / sky_* nullable
  const float *rgb_t, *sky_t, *sem_t;                        // targets [R,3] [R] [R,C], each nullable (term absent)
  int64_t R;
  int C, clip_sem_target;
  float bce_eps;
  float k_rgb, k_sky, k_sem;  // gradient factors: loss_mult * seed * (2 / (3R) | 1 / R | 2 / (RC))
  float *rgb, *acc, *sem;     // blended outputs
  float *d_rgb, *d_sem, *d_acc_raw, *d_sky_rgb, *d_sky_sem;  // d_rgb = d(rgb_f) = d(rgb), d_sem likewise (the blend passes them through)
  float* partial;             // [3][gridDim.x]: per-workgroup sums of squared errors / cross entropies, summed by ps_finish_losses
};

// One wavefront per ray, lane = channel: sky_blend_fwd_kernel, mse_kernel (x2), sky_bce_kernel and sky_blend_bwd_kernel of tail.hip.
__global__ __launch_bounds__(256) void blend_losses_kernel(BlendLossArgs a) {
  __shared__ float red[4][3];
  const int lane = ps_lane(), wave = threadIdx.x >> 6;
  const int C = a.C;
  float s_rgb = 0.f, s_sky = 0.f, s_sem = 0.f;
  for (int64_t ray = (int64_t)blockIdx.x * 4 + wave; ray < a.R; ray += (int64_t)gridDim.x * 4) {
    const float ar = a.acc_raw[ray];
    const float ac = fminf(fmaxf(ar, 0.0f), 1.0f);
    float dot = 0.f;
    if (lane < 3) {
      const float sk = a.sky_rgb ? a.sky_rgb[ray * 3 + lane] : 0.0f;
      const float x = a.rgb_f[ray * 3 + lane] + (a.sky_rgb ? (1.0f - ac) * sk : 0.0f);
      a.rgb[ray * 3 + lane] = x;
      float g = 0.f;
      if (a.rgb_t != nullptr) {
        const float e = x - a.rgb_t[ray * 3 + lane];
        s_rgb += e * e;
        g = a.k_rgb * e;
      }
      a.d_rgb[ray * 3 + lane] = g;
      if (a.sky_rgb != nullptr) {
        dot += g * sk;
        a.d_sky_rgb[ray * 3 + lane] = (1.0f - ac) * g;
      }
    }
    if (a.sem_f != nullptr && lane < C) {
      const float sk = a.sky_sem ? a.sky_sem[ray * C + lane] : 0.0f;
      const float x = a.sem_f[ray * C + lane] + (a.sky_sem ? (1.0f - ac) * sk : 0.0f);
      a.sem[ray * C + lane] = x;
      float g = 0.f;
      if (a.sem_t != nullptr) {
        float t = a.sem_t[ray * C + lane];
        if (a.clip_sem_target) t = fminf(fmaxf(t, 0.0f), 1.0f);
        const float e = x - t;
        s_sem += e * e;
        g = a.k_sem * e;
      }
      a.d_sem[ray * C + lane] = g;
      if (a.sky_sem != nullptr) {
        dot += g * sk;
        a.d_sky_sem[ray * C + lane] = (1.0f - ac) * g;
      }
    }
    dot = ps_wave_sum(dot);
    if (lane == 0) {
      a.acc[ray] = ac;
      float dacc = 0.f;
      if (a.sky_t != nullptr) {  // binary cross entropy of clip(acc, eps, 1 - eps) against 1 - sky_mask (sky_bce_kernel)
        const float t = 1.0f - a.sky_t[ray];
        const float c = fminf(fmaxf(ac, a.bce_eps), 1.0f - a.bce_eps);
        const float la = fmaxf(logf(c), -100.0f), lb = fmaxf(logf(1.0f - c), -100.0f);
        s_sky -= t * la + (1.0f - t) * lb;
        dacc = (ac >= a.bce_eps && ac <= 1.0f - a.bce_eps) ? a.k_sky * (c - t) / fmaxf(c * (1.0f - c), 1e-12f) : 0.0f;
      }
      a.d_acc_raw[ray] = (ar >= 0.0f && ar <= 1.0f) ? dacc - dot : 0.0f;  // torch.clamp passes the gradient at the bounds
    }
  }
  s_rgb = ps_wave_sum(s_rgb);
  s_sem = ps_wave_sum(s_sem);
  if (lane == 0) {
    red[wave][0] = s_rgb;
    red[wave][1] = s_sky;
    red[wave][2] = s_sem;
  }
  __syncthreads();
  if (threadIdx.x < 3) a.partial[threadIdx.x * gridDim.x + blockIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- finish: value_o = sum over the descriptors d of output o (in order) of scale_d * (sum(terms_d[0..n_d)) / denom_d) ---------------
// One 1024-thread workgroup per output; sum() is loss_finish_kernel's (tail.hip): thread t adds the 16-byte groups t, t + 1024, ... as
// (x0 + x1) + (x2 + x3), then the scalar tail, a wave sum, 16 sequential adds -- the same bits as ps_loss_finish.  The workgroup that
// draws the last ticket adds the outputs in order, ((v0 + v1) + v2) + ...: at most 16 device-scope fences per launch.
constexpr int kMaxFin = 16;
struct FinishArgs {
  const float* terms[kMaxFin];
  int64_t n[kMaxFin];
  float denom[kMaxFin], scale[kMaxFin];
  int out_of[kMaxFin];   // output index of descriptor d (non-decreasing)
  float* out[kMaxFin];   // one device scalar per output
  int n_desc, n_out;
  float* total;          // nullable
  unsigned* ticket;      // zero before the launch, zero again after it
};
__global__ __launch_bounds__(1024) void finish_losses_kernel(FinishArgs a) {
  __shared__ float red[16];
  __shared__ int last;
  const int o = blockIdx.x, lane = ps_lane(), w = threadIdx.x >> 6;
  float value = 0.f;
  bool first = true;
  for (int d = 0; d < a.n_desc; ++d) {
    if (a.out_of[d] != o) continue;
    const float* v = a.terms[d];
    const int64_t n = a.n[d];
    float s = 0.f;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(v) & 15) == 0 ? n / 4 : 0;
#pragma unroll 16
    for (int64_t i = threadIdx.x; i < n4; i += 1024) {
      const f32x4 x = reinterpret_cast<const f32x4*>(v)[i];
      s += (x[0] + x[1]) + (x[2] + x[3]);
    }
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 1024) s += v[i];
    s = ps_wave_sum(s);
    __syncthreads();  // (the previous descriptor's readers of red[] are done)
    if (lane == 0) red[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int q = 0; q < 16; ++q) t += red[q];
      const float term = a.scale[d] * (t / a.denom[d]);  // denom == 0: NaN, like torch.mean of an empty selection
      value = first ? term : value + term;
      first = false;
    }
  }
  if (threadIdx.x == 0) a.out[o][0] = value;
  if (a.total == nullptr) return;
  __threadfence();
  if (threadIdx.x == 0) last = atomicAdd(a.ticket, 1u) == gridDim.x - 1u;
  __syncthreads();
  if (!last || threadIdx.x != 0) return;
  __threadfence();
  float t = __hip_atomic_load(a.out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int q = 1; q < a.n_out; ++q) t += __hip_atomic_load(a.out[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  a.total[0] = t;
  *a.ticket = 0u;
}

}  // namespace

extern "C" int ps_ray_out_fwd(const float* weights, const float* rgb_s, const float* sem_hidden_ray, const float* W, const float* b, int64_t R,
                              int S, int C, float* rgb, float* acc, float* sem, void* stream) {
  PS_REQUIRE(C == kC && S > 0 && S <= 64, "ps_ray_out_fwd: 64 semantic channels, at most 64 samples per ray");
  PS_REQUIRE(weights && rgb_s && sem_hidden_ray && W && b && rgb && acc && sem, "ps_ray_out_fwd: null argument");
  if (R == 0) return 0;
  int grid = (int)((R + 3) / 4);
  if (grid > 2048) grid = 2048;
  ray_out_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(weights, rgb_s, sem_hidden_ray, W, b, R, S, rgb, acc, sem);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_ray_dsigma_bwd(const float* ebins, const float* sigma, const float* rgb_s, const float* d_rgb, const float* d_acc,
                                 const float* d_acc2, const float* d_weights_add0, const float* d_weights_add1, int64_t R, int S,
                                 float* dsigma, void* stream) {
  PS_REQUIRE(S > 0 && S <= 64, "ps_ray_dsigma_bwd: at most 64 samples per ray");
  PS_REQUIRE(ebins && sigma && dsigma, "ps_ray_dsigma_bwd: null argument");
  if (R == 0) return 0;
  ray_dsigma_bwd_kernel<<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(ebins, sigma, rgb_s, d_rgb, d_acc, d_acc2, d_weights_add0,
                                                                                 d_weights_add1, R, S, dsigma);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_blend_losses_partials(int64_t R) {
  const int64_t b = (R + 3) / 4;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int ps_blend_losses(const float* rgb_f, const float* acc_raw, const float* sem_f, const float* sky_rgb, const float* sky_sem,
                               const float* rgb_t, const float* sky_t, const float* sem_t, int64_t R, int C, int clip_sem_target,
                               float bce_eps, float k_rgb, float k_sky, float k_sem, float* rgb, float* acc, float* sem, float* d_rgb,
                               float* d_sem, float* d_acc_raw, float* d_sky_rgb, float* d_sky_sem, float* partial, void* stream) {
  PS_REQUIRE(R > 0 && C >= 0 && C <= 64, "ps_blend_losses: empty batch or more than 64 semantic channels");
  PS_REQUIRE(rgb_f && acc_raw && rgb && acc && d_rgb && d_acc_raw && partial, "ps_blend_losses: null argument");
  PS_REQUIRE(sem_f == nullptr || (sem && d_sem), "ps_blend_losses: semantics without output buffers");
  PS_REQUIRE((sky_rgb == nullptr || d_sky_rgb) && (sky_sem == nullptr || (d_sky_sem && sem_f)), "ps_blend_losses: sky inputs without gradient buffers");
  BlendLossArgs a;
  a.rgb_f = rgb_f; a.acc_raw = acc_raw; a.sem_f = sem_f; a.sky_rgb = sky_rgb; a.sky_sem = sem_f ? sky_sem : nullptr;
  a.rgb_t = rgb_t; a.sky_t = sky_t; a.sem_t = sem_f ? sem_t : nullptr;
  a.R = R; a.C = C; a.clip_sem_target = clip_sem_target; a.bce_eps = bce_eps;
  a.k_rgb = k_rgb; a.k_sky = k_sky; a.k_sem = k_sem;
  a.rgb = rgb; a.acc = acc; a.sem = sem; a.d_rgb = d_rgb; a.d_sem = d_sem; a.d_acc_raw = d_acc_raw; a.d_sky_rgb = d_sky_rgb; a.d_sky_sem = d_sky_sem;
  a.partial = partial;
  blend_losses_kernel<<<ps_blend_losses_partials(R), 256, 0, (hipStream_t)stream>>>(a);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_finish_losses(const float* const* terms, const int64_t* n, const float* denom, const float* scale, const int* out_of,
                                int n_desc, float* const* out, int n_out, float* total, uint32_t* ticket, void* stream) {
  PS_REQUIRE(terms && n && denom && scale && out_of && out && n_desc >= 1 && n_desc <= kMaxFin && n_out >= 1 && n_out <= n_desc,
             "ps_finish_losses: 1..16 descriptors, 1..n_desc outputs");
  PS_REQUIRE(total == nullptr || ticket != nullptr, "ps_finish_losses: the total needs a ticket");
  FinishArgs a;
  for (int d = 0; d < kMaxFin; ++d) {
    const bool on = d < n_desc;
    a.terms[d] = on ? terms[d] : nullptr;
    a.n[d] = on ? n[d] : 0;
    a.denom[d] = on ? denom[d] : 1.0f;
    a.scale[d] = on ? scale[d] : 0.0f;
    a.out_of[d] = on ? out_of[d] : -1;
    a.out[d] = d < n_out ? out[d] : nullptr;
    PS_REQUIRE(!on || (terms[d] != nullptr && n[d] >= 1 && out_of[d] >= 0 && out_of[d] < n_out && (d == 0 || out_of[d] >= out_of[d - 1])),
               "ps_finish_losses: bad descriptor (null terms, empty, or output indices not non-decreasing)");
  }
  a.n_desc = n_desc; a.n_out = n_out; a.total = total; a.ticket = ticket;
  finish_losses_kernel<<<n_out, 1024, 0, (hipStream_t)stream>>>(a);
  PS_CHECK_LAUNCH();
}
