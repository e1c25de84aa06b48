// Per-ray kernels: pinhole ray generation, piecewise spaced sampler, density->weights scan,
// PDF resampling and the alpha-composite renderers.  One 64-lane wavefront owns one ray; prefix
// sums along the ray are wavefront shuffles (no LDS round trip), a lane holds CH = ceil(S/64)
// consecutive samples.
//
// Reference semantics (file:line in nerfstudio-0.3.3/nerfstudio):
//   rays      cameras/cameras.py:614-616,773-778,841-870 ; model_components/ray_generators.py:43-61
//   sampler   model_components/ray_samplers.py:78-128 ; models/PreSight/nerfacto_nusc_ms.py:312-317
//   weights   cameras/rays.py:128-150
//   pdf       model_components/ray_samplers.py:305-372
//   renderers model_components/renderers.py:70-117,286-383 ; nerfacto_nusc_ms.py:503-533
#include <algorithm>
#include <type_traits>
#include "common.hpp"
#include "pointwise_core.hpp"

namespace {

constexpr int kMaxCh = 4;  // up to 256 samples per ray

__device__ __forceinline__ float spacing_fn(float x, float thr) { return x < thr ? x / (2.0f * thr) : 1.0f - 1.0f / (2.0f * x / thr); }
__device__ __forceinline__ float spacing_inv(float x, float thr) { return x < 0.5f ? x * (2.0f * thr) : thr / (2.0f - 2.0f * x); }
__device__ __forceinline__ float s_to_euclid(float s, float s_near, float s_far, float thr) {
  return spacing_inv(s * s_far + (1.0f - s) * s_near, thr);
}
using ps::nan_to_num;

// ------------------------------------------------------------------------------------------ rays
__global__ void generate_rays_kernel(const int64_t* __restrict__ ray_indices, const float* __restrict__ c2w,
                                     const float* __restrict__ fx, const float* __restrict__ fy, const float* __restrict__ cx,
                                     const float* __restrict__ cy, int64_t R, float* __restrict__ origins,
                                     float* __restrict__ dirs, float* __restrict__ pixel_area, float* __restrict__ dir_norm) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= R) return;
  const int64_t cam = ray_indices[i * 3 + 0];
  const float y = (float)ray_indices[i * 3 + 1] + 0.5f;
  const float x = (float)ray_indices[i * 3 + 2] + 0.5f;
  const float fx_ = fx[cam], fy_ = fy[cam], cx_ = cx[cam], cy_ = cy[cam];
  const float* M = c2w + cam * 12;
  const float px[3] = {(x - cx_) / fx_, (x - cx_ + 1.0f) / fx_, (x - cx_) / fx_};
  const float py[3] = {-(y - cy_) / fy_, -(y - cy_) / fy_, -(y - cy_ + 1.0f) / fy_};
  float d[3][3];
  float n0 = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = px[k] * M[r * 4 + 0] + py[k] * M[r * 4 + 1] + (-1.0f) * M[r * 4 + 2];
    float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    nrm = fmaxf(nrm, 1e-8f);
    if (k == 0) n0 = nrm;
#pragma unroll
    for (int r = 0; r < 3; ++r) d[k][r] = v[r] / nrm;
  }
  float dx = 0.f, dy = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    dx += (d[0][r] - d[1][r]) * (d[0][r] - d[1][r]);
    dy += (d[0][r] - d[2][r]) * (d[0][r] - d[2][r]);
    origins[i * 3 + r] = M[r * 4 + 3];
    dirs[i * 3 + r] = d[0][r];
  }
  pixel_area[i] = sqrtf(dx) * sqrtf(dy);
  if (dir_norm) dir_norm[i] = n0;
}

// ------------------------------------------------------------------------------------------ spaced sampler
// The next field's points from the bin edges a sampling kernel has just formed (round 6): point (ray, s) = o + d * (e_s + e_{s+1}) / 2,
// normalised / contracted into the field's box -- field_points_kernel's arithmetic (encode.hip: mul, then add, separately rounded),
// without its launch and without re-reading the edges.
struct RayPoints {
  const float *origins, *dirs, *aabb;  // [R,3] [R,3] [6]; u == nullptr: no points wanted
  int contract;
  float *u, *sel;                       // [R*S,3], [R*S]
};
__device__ __forceinline__ void ray_point_store(const RayPoints& P, int64_t ray, int64_t n, float e0, float e1) {
#pragma clang fp contract(off)
  const float mid = (e0 + e1) / 2.0f;
  float p[3], q[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = P.origins[ray * 3 + k] + P.dirs[ray * 3 + k] * mid;
  const bool s_ = ps::normalize_contract(p[0], p[1], p[2], P.aabb, P.contract != 0, q);
  P.u[n * 3] = q[0];
  P.u[n * 3 + 1] = q[1];
  P.u[n * 3 + 2] = q[2];
  P.sel[n] = s_ ? 1.0f : 0.0f;
}

__global__ void spaced_bins_kernel(const float* __restrict__ jitter, int64_t R, int S, float near, float far, float thr,
                                   float* __restrict__ sbins, float* __restrict__ ebins, RayPoints P) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int nb = S + 1;
  if (i >= R * nb) return;
  const int64_t r = i / nb;
  const int k = (int)(i % nb);
  // torch.linspace(0,1,S+1): symmetric evaluation around the midpoint
  const float step = 1.0f / (float)S;
  auto lin = [&](int idx) { return idx < nb / 2 ? step * (float)idx : 1.0f - step * (float)(nb - idx - 1); };
  const float s_near = spacing_fn(near, thr), s_far = spacing_fn(far, thr);
  auto edge = [&](int kk) {
    float b = lin(kk);
    if (jitter != nullptr) {
      const float lo = (kk == 0) ? lin(0) : (lin(kk) + lin(kk - 1)) / 2.0f;
      const float hi = (kk == S) ? lin(S) : (lin(kk + 1) + lin(kk)) / 2.0f;
      b = lo + (hi - lo) * jitter[r];
    }
    return b;
  };
  const float b = edge(k);
  const float e = s_to_euclid(b, s_near, s_far, thr);
  sbins[i] = b;
  ebins[i] = e;
  if (P.u != nullptr && k < S) ray_point_store(P, r, r * S + k, e, s_to_euclid(edge(k + 1), s_near, s_far, thr));  // (the neighbour's edge: same function, same bits)
}

// ------------------------------------------------------------------------------------------ weights scan
// lane owns samples [lane*CH, lane*CH+CH)
template <int CH>
__global__ __launch_bounds__(256) void weights_fwd_kernel(const float* __restrict__ ebins, const float* __restrict__ sigma,
                                                          int64_t R, int S, float* __restrict__ weights) {
  const int64_t ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int lane = ps_lane();
  const float* e = ebins + ray * (S + 1);
  float dd[CH], local = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int s = lane * CH + c;
    dd[c] = (s < S) ? (e[s + 1] - e[s]) * sigma[ray * S + s] : 0.0f;
    local += dd[c];
  }
  float excl = ps_wave_incl_scan(local) - local;  // sum of dd over all earlier lanes
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int s = lane * CH + c;
    const float w = (1.0f - expf(-dd[c])) * expf(-excl);
    if (s < S) weights[ray * S + s] = nan_to_num(w);
    excl += dd[c];
  }
}

template <int CH>
__global__ __launch_bounds__(256) void weights_bwd_kernel(const float* __restrict__ ebins, const float* __restrict__ sigma,
                                                          const float* __restrict__ dweights, int64_t R, int S,
                                                          float* __restrict__ dsigma) {
  const int64_t ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int lane = ps_lane();
  const float* e = ebins + ray * (S + 1);
  float dd[CH], delta[CH], gw[CH], local = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int s = lane * CH + c;
    delta[c] = (s < S) ? (e[s + 1] - e[s]) : 0.0f;
    dd[c] = (s < S) ? delta[c] * sigma[ray * S + s] : 0.0f;
    gw[c] = (s < S) ? dweights[ray * S + s] : 0.0f;
    local += dd[c];
  }
  float excl = ps_wave_incl_scan(local) - local;
  // a_k = gw_k * w_k (zero where the raw weight is not finite, as torch.nan_to_num's backward does)
  float a[CH], tr_after[CH], asum = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const float T = expf(-excl), ed = expf(-dd[c]);
    const float w = (1.0f - ed) * T;
    const bool fin = isfinite(w);
    a[c] = fin ? gw[c] * w : 0.0f;
    tr_after[c] = fin ? gw[c] * ed * T : 0.0f;  // d w_k / d dd_k
    asum += a[c];
    excl += dd[c];
  }
  // suffix sum over later samples: total - inclusive prefix
  const float incl = ps_wave_incl_scan(asum);
  const float total = __shfl(incl, 63, 64);
  float later = total - incl;  // sum of a over all later lanes
#pragma unroll
  for (int c = CH - 1; c >= 0; --c) {
    const int s = lane * CH + c;
    if (s < S) dsigma[ray * S + s] = (tr_after[c] - later) * delta[c];
    later += a[c];
  }
}

// ------------------------------------------------------------------------------------------ pdf resampling
// one wave per ray; cdf staged in LDS, each lane binary-searches its new bin edges
__global__ __launch_bounds__(256) void pdf_resample_kernel(const float* __restrict__ weights, const float* __restrict__ sbins,
                                                           const float* __restrict__ jitter, int64_t R, int S, int n_new,
                                                           float anneal, float pad, float eps, float near, float far,
                                                           float thr, float* __restrict__ new_sbins,
                                                           float* __restrict__ new_ebins) {
  __shared__ float lds[4][2][kMaxCh * 64 + 1];
  const int wv = threadIdx.x >> 6;
  const int64_t ray = blockIdx.x * 4 + wv;
  if (ray >= R) return;
  const int lane = ps_lane();
  float* cdf = lds[wv][0];
  float* eb = lds[wv][1];
  const int CH = (S + 63) / 64;
  // w = pow(weights, anneal) + pad, summed over the ray
  float wloc[kMaxCh], local = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxCh; ++c) {
    const int s = lane * CH + c;
    float w = 0.f;
    if (c < CH && s < S) {
      w = weights[ray * S + s];
      if (anneal != 1.0f) w = powf(w, anneal);
      w += pad;
    }
    wloc[c] = w;
    local += w;
  }
  float wsum = ps_wave_sum(local);
  const float padding = fmaxf(eps - wsum, 0.0f);
  const float padw = padding / (float)S;
  wsum += padding;
  local = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxCh; ++c) {
    const int s = lane * CH + c;
    wloc[c] = (c < CH && s < S) ? (wloc[c] + padw) / wsum : 0.0f;
    local += wloc[c];
  }
  float run = ps_wave_incl_scan(local) - local;
  if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
  for (int c = 0; c < kMaxCh; ++c) {
    const int s = lane * CH + c;
    run += wloc[c];
    if (c < CH && s < S) cdf[s + 1] = fminf(1.0f, run);
  }
  for (int s = lane; s <= S; s += 64) eb[s] = sbins[ray * (S + 1) + s];
  __builtin_amdgcn_wave_barrier();
  const int nb = n_new + 1;
  const float end = (float)(1.0 - 1.0 / (double)nb);
  const float step = end / (float)(nb - 1);
  const float s_near = spacing_fn(near, thr), s_far = spacing_fn(far, thr);
  for (int i = lane; i < nb; i += 64) {
    float u = (i < nb / 2) ? step * (float)i : end - step * (float)(nb - i - 1);
    if (jitter != nullptr)
      u = u + jitter[ray] / (float)nb;
    else
      u = u + (float)(1.0 / (2.0 * (double)nb));
    // searchsorted(cdf, u, right): number of entries <= u
    int lo = 0, hi = S + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = min(max(lo - 1, 0), S), above = min(max(lo, 0), S);
    const float c0 = cdf[below], c1 = cdf[above], b0 = eb[below], b1 = eb[above];
    float t = (u - c0) / (c1 - c0);
    if (isnan(t)) t = 0.0f;
    t = fminf(fmaxf(nan_to_num(t), 0.0f), 1.0f);
    const float b = b0 + t * (b1 - b0);
    new_sbins[ray * nb + i] = b;
    new_ebins[ray * nb + i] = s_to_euclid(b, s_near, s_far, thr);
  }
}

// ------------------------------------------------------------------------------------------ weights + resampling (+ next points)
// One launch per proposal level instead of three (round 6): RaySamples.get_weights (weights_fwd_kernel), the PDF resampling of
// the next level's bin edges (pdf_resample_kernel) and the next field's points (field_points_kernel) all walk one ray per
// wavefront with a lane owning CH consecutive samples -- the weights stay in registers between the first two, the new edges in LDS
// between the last two.  Same arithmetic in the same order as the three kernels (the weights are written out: the interlevel
// loss and the backward read them).
template <int CH>
__global__ __launch_bounds__(256) void weights_resample_kernel(const float* __restrict__ ebins, const float* __restrict__ sigma,
                                                               const float* __restrict__ sbins, const float* __restrict__ jitter,
                                                               int64_t R, int S, int n_new, float anneal, float pad, float eps,
                                                               float near, float far, float thr, float* __restrict__ weights,
                                                               float* __restrict__ new_sbins, float* __restrict__ new_ebins, RayPoints P) {
  __shared__ float lds[4][3][kMaxCh * 64 + 1];
  const int wv = threadIdx.x >> 6;
  const int64_t ray = blockIdx.x * 4 + wv;
  if (ray >= R) return;
  const int lane = ps_lane();
  float* cdf = lds[wv][0];
  float* eb = lds[wv][1];
  float* ne = lds[wv][2];
  const float* e = ebins + ray * (S + 1);
  // weights (weights_fwd_kernel<CH>)
  float dd[CH], wl[CH], local = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int s = lane * CH + c;
    dd[c] = (s < S) ? (e[s + 1] - e[s]) * sigma[ray * S + s] : 0.0f;
    local += dd[c];
  }
  float excl = ps_wave_incl_scan(local) - local;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int s = lane * CH + c;
    wl[c] = nan_to_num((1.0f - expf(-dd[c])) * expf(-excl));
    if (s < S) weights[ray * S + s] = wl[c];
    excl += dd[c];
  }
  // resampling (pdf_resample_kernel with its weights in registers)
  float wloc[CH];
  local = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int s = lane * CH + c;
    float w = 0.f;
    if (s < S) {
      w = wl[c];
      if (anneal != 1.0f) w = powf(w, anneal);
      w += pad;
    }
    wloc[c] = w;
    local += w;
  }
  float wsum = ps_wave_sum(local);
  const float padding = fmaxf(eps - wsum, 0.0f);
  const float padw = padding / (float)S;
  wsum += padding;
  local = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int s = lane * CH + c;
    wloc[c] = (s < S) ? (wloc[c] + padw) / wsum : 0.0f;
    local += wloc[c];
  }
  float run = ps_wave_incl_scan(local) - local;
  if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int s = lane * CH + c;
    run += wloc[c];
    if (s < S) cdf[s + 1] = fminf(1.0f, run);
  }
  for (int s = lane; s <= S; s += 64) eb[s] = sbins[ray * (S + 1) + s];
  __builtin_amdgcn_wave_barrier();
  const int nb = n_new + 1;
  const float end = (float)(1.0 - 1.0 / (double)nb);
  const float step = end / (float)(nb - 1);
  const float s_near = spacing_fn(near, thr), s_far = spacing_fn(far, thr);
  for (int i = lane; i < nb; i += 64) {
    float u = (i < nb / 2) ? step * (float)i : end - step * (float)(nb - i - 1);
    if (jitter != nullptr)
      u = u + jitter[ray] / (float)nb;
    else
      u = u + (float)(1.0 / (2.0 * (double)nb));
    int lo = 0, hi = S + 1;  // searchsorted(cdf, u, right): number of entries <= u
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = min(max(lo - 1, 0), S), above = min(max(lo, 0), S);
    const float c0 = cdf[below], c1 = cdf[above], b0 = eb[below], b1 = eb[above];
    float t = (u - c0) / (c1 - c0);
    if (isnan(t)) t = 0.0f;
    t = fminf(fmaxf(nan_to_num(t), 0.0f), 1.0f);
    const float b = b0 + t * (b1 - b0);
    const float en = s_to_euclid(b, s_near, s_far, thr);
    new_sbins[ray * nb + i] = b;
    new_ebins[ray * nb + i] = en;
    ne[i] = en;
  }
  if (P.u == nullptr) return;
  __builtin_amdgcn_wave_barrier();
  for (int s2 = lane; s2 < n_new; s2 += 64) ray_point_store(P, ray, ray * n_new + s2, ne[s2], ne[s2 + 1]);
}

// ------------------------------------------------------------------------------------------ composite
// One wave per ray.  Lane c accumulates channel c of the C-dim features (C <= 64); lanes 0..2 also
// the rgb channels; sample-indexed scalars (acc, depths) are done with CH samples per lane + shuffles.
// SEM = false / MAXCH = 1: the factored training node's call (no per-sample semantics, S <= 64) -- the same arithmetic without the
// semantic branch's 16 row registers and with one sample per lane: 146 -> ~30 registers, 3 -> 8 resident waves per SIMD.
template <bool SEM = true, int MAXCH = kMaxCh>
__device__ __forceinline__ void composite_ray(const float* __restrict__ weights, const float* __restrict__ ebins,
                                              const float* __restrict__ rgb_s, const float* __restrict__ sem_s, int64_t ray, int S, int C,
                                              float threshold, float* __restrict__ rgb, float* __restrict__ acc,
                                              float* __restrict__ depth, float* __restrict__ exp_depth, float* __restrict__ sem) {
  const int lane = ps_lane();
  const float* w = weights + ray * S;
  const float* e = ebins + ray * (S + 1);
  // channel-parallel accumulations
  float a_sem = 0.f, a_rgb = 0.f;
  const bool vec = (C & 3) == 0;  // 16 lanes x 4 channels cover a row with one 16-byte load per lane: four rows per load instruction
  if (vec && (sem_s != nullptr || rgb_s != nullptr)) {
    const int q = lane & 15, g = lane >> 4;
    if (SEM && sem_s != nullptr) {
      const bool on = 4 * q < C;
      const float* ps = sem_s + ray * S * C + 4 * q;
      f32x4 a4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (S <= 64) {  // the usual ray: all 16 row loads of a lane in flight at once (inside a step: 0.50 -> 0.43 ms per step)
        f32x4 v[16];
        float ws[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int sr = 4 * u + g;
          const bool ok = on && sr < S;
          v[u] = ok ? *reinterpret_cast<const f32x4*>(ps + (int64_t)sr * C) : (f32x4){0.f, 0.f, 0.f, 0.f};
          ws[u] = ok ? w[sr] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) a4 += v[u] * ws[u];
      } else
#pragma unroll 4
      for (int s0 = 0; s0 < S; s0 += 16) {
        f32x4 v[4];
        float ws[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int sr = s0 + 4 * u + g;
          const bool ok = on && sr < S;
          v[u] = ok ? *reinterpret_cast<const f32x4*>(ps + (int64_t)sr * C) : (f32x4){0.f, 0.f, 0.f, 0.f};
          ws[u] = ok ? w[sr] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) a4 += v[u] * ws[u];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a4[r] += __shfl_xor(a4[r], 16, 64);
        a4[r] += __shfl_xor(a4[r], 32, 64);
      }
      if (sem != nullptr && g == 0 && on) *reinterpret_cast<f32x4*>(sem + ray * C + 4 * q) = a4;
    }
    if (rgb_s != nullptr) {  // lane = sample, three channels each, one wave reduction per channel
      float c0 = 0.f, c1 = 0.f, c2 = 0.f;
      for (int sr = lane; sr < S; sr += 64) {
        const float* pr = rgb_s + (ray * S + sr) * 3;
        const float wv = w[sr];
        c0 += wv * pr[0];
        c1 += wv * pr[1];
        c2 += wv * pr[2];
      }
      c0 = ps_wave_sum(c0);
      c1 = ps_wave_sum(c1);
      c2 = ps_wave_sum(c2);
      if (rgb != nullptr && lane < 3) rgb[ray * 3 + lane] = lane == 0 ? c0 : (lane == 1 ? c1 : c2);
    }
  } else if (sem_s != nullptr || rgb_s != nullptr) {
    const bool sem_on = SEM && sem_s != nullptr && lane < C, rgb_on = rgb_s != nullptr && lane < 3;
    const float* ps = sem_s + ray * S * C + lane;
    const float* pr = rgb_s + ray * S * 3 + lane;
    // 8 sample rows in flight per lane (the rows are independent 256-byte streams; the FMA chain is the only dependency)
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      float v[8], c3[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = sem_on ? ps[(int64_t)(s + k) * C] : 0.0f;
        c3[k] = rgb_on ? pr[(int64_t)(s + k) * 3] : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float ws = w[s + k];
        a_sem += ws * v[k];
        a_rgb += ws * c3[k];
      }
    }
    for (; s < S; ++s) {
      const float ws = w[s];
      if (sem_on) a_sem += ws * ps[(int64_t)s * C];
      if (rgb_on) a_rgb += ws * pr[(int64_t)s * 3];
    }
  }
  if (!vec) {
    if (sem != nullptr && lane < C) sem[ray * C + lane] = a_sem;
    if (rgb != nullptr && lane < 3) rgb[ray * 3 + lane] = a_rgb;
  }
  // sample-parallel part
  const int CH = (S + 63) / 64;
  float wl[MAXCH], st[MAXCH], local = 0.f, wt = 0.f;
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) {
    const int s = lane * CH + c;
    const bool ok = (c < CH && s < S);
    wl[c] = ok ? w[s] : 0.0f;
    st[c] = ok ? (e[s] + e[s + 1]) / 2.0f : 0.0f;
    local += wl[c];
    wt += wl[c] * st[c];
  }
  const float incl = ps_wave_incl_scan(local);
  const float total = __shfl(incl, 63, 64);
  const float wtsum = ps_wave_sum(wt);
  // threshold depth: first sample whose inclusive cumsum >= threshold (searchsorted left), clamped to S-1
  float run = incl - local;
  int first = S;  // sentinel
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) {
    const int s = lane * CH + c;
    run += wl[c];
    if (c < CH && s < S && run >= threshold && first == S) first = s;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) first = min(first, __shfl_xor(first, m, 64));
  const int idx = min(first, S - 1);
  if (lane == 0) {
    if (acc) acc[ray] = total;
    if (depth) depth[ray] = (e[idx] + e[idx + 1]) / 2.0f;
    if (exp_depth) exp_depth[ray] = wtsum / (total + 1e-10f);
  }
}

template <bool SEM, int MAXCH>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ weights, const float* __restrict__ ebins,
                                                            const float* __restrict__ rgb_s, const float* __restrict__ sem_s,
                                                            int64_t R, int S, int C, float threshold, float* __restrict__ rgb,
                                                            float* __restrict__ acc, float* __restrict__ depth,
                                                            float* __restrict__ exp_depth, float* __restrict__ sem) {
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray < R) composite_ray<SEM, MAXCH>(weights, ebins, rgb_s, sem_s, ray, S, C, threshold, rgb, acc, depth, exp_depth, sem);
}

// Extrema of the sample mid-points (e[s] + e[s+1]) / 2 over the whole batch = the clip range of the expected depth
// (ns/model_components/renderers.py:374-389: clip to [steps.min(), steps.max()]).  They depend on the bin edges only, so they are
// a flat reduction over ebins [R, S + 1] by a FIXED number of workgroups, one atomic pair each: same-address global atomics
// serialise at ~12 ns apiece, and when the compositing kernel published one pair per ray (or, later, per workgroup of a
// grid-stride version) 65536 rays cost between 0.05 and 1.5 ms depending on how short the kernel was.
__global__ __launch_bounds__(256) void midpoint_minmax_kernel(const float* __restrict__ ebins, int64_t R, int S, float* __restrict__ minmax) {
  __shared__ float s_min[4], s_max[4];
  float smin = 3.4e38f, smax = -3.4e38f;
  const int64_t n = R * (S + 1);
  const uint32_t row = (uint32_t)(S + 1);
  // four strided elements per thread and pass: the loop is a chain load -> min / max, and 256 workgroups of such chains needed 40 us
  // for 17 MB (min / max are exact whatever the order)
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) {
    float a[4], b[4];
    bool use[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * stride;
      const bool in = i < n;
      const bool edge = n < (int64_t(1) << 32) ? ((uint32_t)i % row) == (uint32_t)S : (i % (S + 1)) == S;  // the last edge of a ray starts no sample
      use[k] = in && !edge;
      a[k] = use[k] ? ebins[i] : 0.0f;
      b[k] = use[k] ? ebins[i + 1] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (use[k]) {
        const float m = (a[k] + b[k]) / 2.0f;
        smin = fminf(smin, m);
        smax = fmaxf(smax, m);
      }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    smin = fminf(smin, __shfl_xor(smin, m, 64));
    smax = fmaxf(smax, __shfl_xor(smax, m, 64));
  }
  if (ps_lane() == 0) {
    s_min[threadIdx.x >> 6] = smin;
    s_max[threadIdx.x >> 6] = smax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    smin = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
    smax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    // steps are positive: the int ordering of the bit patterns equals the float ordering
    int* mm = reinterpret_cast<int*>(minmax);
    if (smin < 3.0e38f) atomicMin(mm, __float_as_int(smin));
    if (smax > 0.0f) atomicMax(mm + 1, __float_as_int(smax));
  }
}

__global__ void clip_kernel(float* __restrict__ v, int64_t n, const float* __restrict__ minmax, float* __restrict__ keep) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) {
    const float x = v[i];
    const float c = fminf(fmaxf(x, minmax[0]), minmax[1]);
    v[i] = c;
    if (keep != nullptr) keep[i] = (c == x) ? 1.0f : 0.0f;  // the clip's derivative (what `raw == clipped` gives)
  }
}

// d_w[s] = sum_c rgb_s*d_rgb + sum_c sem_s*d_sem + d_acc + d_expdepth-terms ; d_rgb_s = w*d_rgb ; d_sem_s = w*d_sem
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ weights, const float* __restrict__ ebins,
                                                            const float* __restrict__ rgb_s, const float* __restrict__ sem_s,
                                                            const float* __restrict__ d_rgb, const float* __restrict__ d_acc,
                                                            const float* __restrict__ d_sem, const float* __restrict__ d_exp,
                                                            int64_t R, int S, int C, float* __restrict__ d_weights,
                                                            float* __restrict__ d_rgb_s, float* __restrict__ d_sem_s,
                                                            const float* __restrict__ add0, const float* __restrict__ add1) {
  const int64_t ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int lane = ps_lane();
  const float* w = weights + ray * S;
  const float* e = ebins + ray * (S + 1);
  const float gsem = (d_sem != nullptr && lane < C) ? d_sem[ray * C + lane] : 0.0f;
  const float grgb = (d_rgb != nullptr && lane < 3) ? d_rgb[ray * 3 + lane] : 0.0f;
  const float gacc = d_acc ? d_acc[ray] : 0.0f;
  // expected depth D = A/(B+eps), A = sum w t, B = sum w  ->  dD/dw_s = t_s/(B+eps) - A/(B+eps)^2
  float ge = 0.f, A = 0.f, B = 0.f;
  if (d_exp != nullptr) {
    ge = d_exp[ray];
    float la = 0.f, lb = 0.f;
    for (int s = lane; s < S; s += 64) {
      la += w[s] * (e[s] + e[s + 1]) / 2.0f;
      lb += w[s];
    }
    A = ps_wave_sum(la);
    B = ps_wave_sum(lb) + 1e-10f;
  }
  for (int s = 0; s < S; ++s) {
    const float ws = w[s];
    float part = 0.f;
    if (sem_s != nullptr && lane < C) {
      part += sem_s[(ray * S + s) * C + lane] * gsem;
      if (d_sem_s) d_sem_s[(ray * S + s) * C + lane] = ws * gsem;
    }
    if (rgb_s != nullptr && lane < 3) {
      part += rgb_s[(ray * S + s) * 3 + lane] * grgb;
      if (d_rgb_s) d_rgb_s[(ray * S + s) * 3 + lane] = ws * grgb;
    }
    part = ps_wave_sum(part);
    if (lane == 0) {
      float g = part + gacc;
      if (d_exp != nullptr) g += ge * (((e[s] + e[s + 1]) / 2.0f) / B - A / (B * B));
      if (add0 != nullptr) g += add0[ray * S + s];
      if (add1 != nullptr) g += add1[ray * S + s];
      d_weights[ray * S + s] = g;
    }
  }
}

// d_weights ONLY, for S <= 64 (the per-sample gradients d_rgb_s / d_sem_s are then formed inside the field backward from
// the per-ray gradients, ps_main_field_bwd with `weights`).  One wavefront per ray; while streaming the sample rows the
// lane is the channel (coalesced 256-byte rows, 64 independent loads in flight), then a recursive-halving
// transpose-reduction (63 cross-lane exchanges instead of 64 six-step wave sums) leaves lane s with the dot product of
// sample s.
template <bool SEM>
__global__ __launch_bounds__(256) void composite_bwd_w_kernel(const float* __restrict__ weights, const float* __restrict__ ebins,
                                                              const float* __restrict__ rgb_s, const float* __restrict__ sem_s,
                                                              const float* __restrict__ d_rgb, const float* __restrict__ d_acc,
                                                              const float* __restrict__ d_sem, const float* __restrict__ d_exp,
                                                              int64_t R, int S, int C, float* __restrict__ d_weights,
                                                              const float* __restrict__ add0, const float* __restrict__ add1) {
  const int64_t ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int lane = ps_lane();
  const float* w = weights + ray * S;
  const float* e = ebins + ray * (S + 1);
  if (!SEM || (C & 3) == 0) {
    // 16 lanes x 4 channels per row, lane group g takes the rows 4u + g: 16 sixteen-byte loads per lane instead of 64 dword
    // loads, then a recursive halving over the 16 lanes of a row leaves lane (q, g) with the dot product of sample 4q + g
    const int q = lane & 15, g = lane >> 4;
    float v[SEM ? 16 : 1];
    v[0] = 0.0f;  // (SEM = false, the factored node's call: no semantic rows -- the sum of sixteen zeros)
    if constexpr (SEM) {
    const bool on = sem_s != nullptr && d_sem != nullptr && 4 * q < C;
    const f32x4 gs = on ? *reinterpret_cast<const f32x4*>(d_sem + ray * C + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int sr = 4 * u + g;
      f32x4 x = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (on && sr < S) x = *reinterpret_cast<const f32x4*>(sem_s + (ray * S + sr) * C + 4 * q);
      v[u] = (x[0] * gs[0] + x[1] * gs[1]) + (x[2] * gs[2] + x[3] * gs[3]);
    }
#define PS_HALVE16(M)                                   \
  _Pragma("unroll") for (int i = 0; i < M; ++i) {       \
    const bool up = (q & M) != 0;                       \
    const float keep = up ? v[i + M] : v[i];            \
    const float send = up ? v[i] : v[i + M];            \
    v[i] = keep + __shfl_xor(send, M, 64);              \
  }
    PS_HALVE16(8) PS_HALVE16(4) PS_HALVE16(2) PS_HALVE16(1)
#undef PS_HALVE16
    }
    const int sr = 4 * q + g;
    const bool ok = sr < S;
    float gr = v[0];
    if (ok && rgb_s != nullptr && d_rgb != nullptr) {
      const float* pr = rgb_s + (ray * S + sr) * 3;
      gr += pr[0] * d_rgb[ray * 3] + pr[1] * d_rgb[ray * 3 + 1] + pr[2] * d_rgb[ray * 3 + 2];
    }
    const float ws = ok ? w[sr] : 0.0f;
    const float mid = ok ? (e[sr] + e[sr + 1]) / 2.0f : 0.0f;
    gr += d_acc ? d_acc[ray] : 0.0f;
    if (d_exp != nullptr) {
      const float A = ps_wave_sum(ws * mid), B = ps_wave_sum(ws) + 1e-10f;
      gr += d_exp[ray] * (mid / B - A / (B * B));
    }
    if (ok) {  // (+ the addends, in this order: what `dw += add0; dw += add1` launches used to do)
      if (add0 != nullptr) gr += add0[ray * S + sr];
      if (add1 != nullptr) gr += add1[ray * S + sr];
      d_weights[ray * S + sr] = gr;
    }
    return;
  }
  if constexpr (!SEM) return;  // (never reached: the branch above took every SEM = false call)
  const bool sem_on = sem_s != nullptr && d_sem != nullptr && lane < C;
  const bool rgb_on = rgb_s != nullptr && d_rgb != nullptr && lane < 3;
  const float gsem = sem_on ? d_sem[ray * C + lane] : 0.0f;
  const float grgb = rgb_on ? d_rgb[ray * 3 + lane] : 0.0f;
  float v[SEM ? 64 : 1];
#pragma unroll
  for (int s = 0; s < 64; ++s) {
    float part = 0.f;
    if (s < S) {
      if (sem_on) part = sem_s[(ray * S + s) * C + lane] * gsem;
      if (rgb_on) part += rgb_s[(ray * S + s) * 3 + lane] * grgb;
    }
    v[s] = part;
  }
#define PS_HALVE(M)                                     \
  _Pragma("unroll") for (int i = 0; i < M; ++i) {       \
    const bool up = (lane & M) != 0;                    \
    const float keep = up ? v[i + M] : v[i];            \
    const float send = up ? v[i] : v[i + M];            \
    v[i] = keep + __shfl_xor(send, M, 64);              \
  }
  PS_HALVE(32) PS_HALVE(16) PS_HALVE(8) PS_HALVE(4) PS_HALVE(2) PS_HALVE(1)
#undef PS_HALVE
  const float ws = lane < S ? w[lane] : 0.0f;
  const float mid = lane < S ? (e[lane] + e[lane + 1]) / 2.0f : 0.0f;
  float g = v[0] + (d_acc ? d_acc[ray] : 0.0f);
  if (d_exp != nullptr) {  // expected depth D = A/(B+eps): dD/dw_s = t_s/(B+eps) - A/(B+eps)^2
    const float A = ps_wave_sum(ws * mid), B = ps_wave_sum(ws) + 1e-10f;
    g += d_exp[ray] * (mid / B - A / (B * B));
  }
  if (lane < S) {
    if (add0 != nullptr) g += add0[ray * S + lane];
    if (add1 != nullptr) g += add1[ray * S + lane];
    d_weights[ray * S + lane] = g;
  }
}

template <class F>
int by_chunk(int S, F f) {
  const int ch = (S + 63) / 64;
  if (ch <= 1) return f(std::integral_constant<int, 1>());
  if (ch == 2) return f(std::integral_constant<int, 2>());
  if (ch <= 4) return f(std::integral_constant<int, 4>());
  ps_set_error("samples per ray must be <= 256");
  return -1;
}

}  // namespace

extern "C" int ps_generate_rays(const int64_t* ray_indices, const float* c2w, const float* fx, const float* fy,
                                const float* cx, const float* cy, int64_t R, float* origins, float* dirs, float* pixel_area,
                                float* dir_norm, void* stream) {
  if (R == 0) return 0;
  generate_rays_kernel<<<(unsigned)((R + 255) / 256), 256, 0, (hipStream_t)stream>>>(ray_indices, c2w, fx, fy, cx, cy, R,
                                                                                    origins, dirs, pixel_area, dir_norm);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_spaced_bins(const float* jitter, int64_t R, int S, float near, float far, float thr, float* sbins,
                              float* ebins, void* stream) {
  if (R == 0) return 0;
  const int64_t n = R * (S + 1);
  spaced_bins_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(jitter, R, S, near, far, thr, sbins, ebins, RayPoints{});
  PS_CHECK_LAUNCH();
}

// ps_spaced_bins + ps_field_points (rays + these bin edges) in one launch
extern "C" int ps_spaced_bins_points(const float* jitter, int64_t R, int S, float near, float far, float thr, float* sbins, float* ebins,
                                     const float* origins, const float* dirs, const float* aabb, int contract, float* u, float* sel,
                                     void* stream) {
  PS_REQUIRE(origins && dirs && aabb && u && sel, "ps_spaced_bins_points: null argument");
  if (R == 0) return 0;
  const int64_t n = R * (S + 1);
  spaced_bins_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(jitter, R, S, near, far, thr, sbins, ebins,
                                                                                   RayPoints{origins, dirs, aabb, contract, u, sel});
  PS_CHECK_LAUNCH();
}

// ps_weights_fwd + ps_pdf_resample (+ ps_field_points on the new edges when u != NULL) in one launch
extern "C" int ps_weights_resample(const float* ebins, const float* sigma, const float* sbins, const float* jitter, int64_t R, int S,
                                   int n_new, float anneal, float pad, float eps, float near, float far, float thr, float* weights,
                                   float* new_sbins, float* new_ebins, const float* origins, const float* dirs, const float* aabb,
                                   int contract, float* u, float* sel, void* stream) {
  PS_REQUIRE(S <= kMaxCh * 64 && n_new <= kMaxCh * 64 - 1, "ps_weights_resample: samples per ray must be <= 256 (255 new ones)");
  PS_REQUIRE(ebins && sigma && sbins && weights && new_sbins && new_ebins, "ps_weights_resample: null argument");
  PS_REQUIRE(u == nullptr || (origins && dirs && aabb && sel), "ps_weights_resample: points without rays / box");
  if (R == 0) return 0;
  const unsigned grid = (unsigned)((R + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  const RayPoints P{origins, dirs, aabb, contract, u, sel};
  int rc = by_chunk(S, [&](auto ch) {
    weights_resample_kernel<decltype(ch)::value><<<grid, 256, 0, s>>>(ebins, sigma, sbins, jitter, R, S, n_new, anneal, pad, eps, near, far,
                                                                      thr, weights, new_sbins, new_ebins, P);
    return 0;
  });
  if (rc) return rc;
  PS_CHECK_LAUNCH();
}

extern "C" int ps_weights_fwd(const float* ebins, const float* sigma, int64_t R, int S, float* weights, void* stream) {
  if (R == 0) return 0;
  const unsigned grid = (unsigned)((R + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  int rc = by_chunk(S, [&](auto ch) {
    weights_fwd_kernel<decltype(ch)::value><<<grid, 256, 0, s>>>(ebins, sigma, R, S, weights);
    return 0;
  });
  if (rc) return rc;
  PS_CHECK_LAUNCH();
}

extern "C" int ps_weights_bwd(const float* ebins, const float* sigma, const float* dweights, int64_t R, int S,
                              float* dsigma, void* stream) {
  if (R == 0) return 0;
  const unsigned grid = (unsigned)((R + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  int rc = by_chunk(S, [&](auto ch) {
    weights_bwd_kernel<decltype(ch)::value><<<grid, 256, 0, s>>>(ebins, sigma, dweights, R, S, dsigma);
    return 0;
  });
  if (rc) return rc;
  PS_CHECK_LAUNCH();
}

extern "C" int ps_pdf_resample(const float* weights, const float* sbins, const float* jitter, int64_t R, int S, int n_new,
                               float anneal, float pad, float eps, float near, float far, float thr, float* new_sbins,
                               float* new_ebins, void* stream) {
  PS_REQUIRE(S <= kMaxCh * 64, "ps_pdf_resample: samples per ray must be <= 256");
  if (R == 0) return 0;
  pdf_resample_kernel<<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(weights, sbins, jitter, R, S, n_new, anneal,
                                                                               pad, eps, near, far, thr, new_sbins, new_ebins);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_composite_fwd(const float* weights, const float* ebins, const float* rgb_s, const float* sem_s, int64_t R,
                                int S, int C, float threshold, float* rgb, float* acc, float* depth, float* exp_depth,
                                float* sem, float* minmax, void* stream) {
  PS_REQUIRE(S <= kMaxCh * 64 && C <= 64, "ps_composite_fwd: S must be <= 256 and C <= 64");
  if (R == 0) return 0;
  if (minmax != nullptr) {
    const unsigned grid = (unsigned)std::min<int64_t>((R * (S + 1) + 1023) / 1024, 256);
    midpoint_minmax_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(ebins, R, S, minmax);
  }
  if (sem_s == nullptr && S <= 64)
    composite_fwd_kernel<false, 1><<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(weights, ebins, rgb_s, sem_s, R, S, C, threshold,
                                                                                            rgb, acc, depth, exp_depth, sem);
  else
    composite_fwd_kernel<true, kMaxCh><<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(weights, ebins, rgb_s, sem_s, R, S, C,
                                                                                                threshold, rgb, acc, depth, exp_depth, sem);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_clip(float* v, int64_t n, const float* minmax, float* keep, void* stream) {
  if (n == 0) return 0;
  clip_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(v, n, minmax, keep);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_composite_bwd(const float* weights, const float* ebins, const float* rgb_s, const float* sem_s,
                                const float* d_rgb, const float* d_acc, const float* d_sem, const float* d_exp, int64_t R,
                                int S, int C, float* d_weights, float* d_rgb_s, float* d_sem_s, const float* d_weights_add0,
                                const float* d_weights_add1, void* stream) {
  PS_REQUIRE(S <= kMaxCh * 64 && C <= 64, "ps_composite_bwd: S must be <= 256 and C <= 64");
  if (R == 0) return 0;
  if (d_rgb_s == nullptr && d_sem_s == nullptr && S <= 64) {
    if (sem_s == nullptr || d_sem == nullptr)
      composite_bwd_w_kernel<false><<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(weights, ebins, rgb_s, sem_s, d_rgb, d_acc,
                                                                                             d_sem, d_exp, R, S, C, d_weights, d_weights_add0, d_weights_add1);
    else
      composite_bwd_w_kernel<true><<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(weights, ebins, rgb_s, sem_s, d_rgb, d_acc,
                                                                                            d_sem, d_exp, R, S, C, d_weights, d_weights_add0, d_weights_add1);
    PS_CHECK_LAUNCH();
  }
  composite_bwd_kernel<<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(weights, ebins, rgb_s, sem_s, d_rgb, d_acc,
                                                                                d_sem, d_exp, R, S, C, d_weights, d_rgb_s,
                                                                                d_sem_s, d_weights_add0, d_weights_add1);
  PS_CHECK_LAUNCH();
}
