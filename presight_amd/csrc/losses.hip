// Per-ray regularisation losses, forward + gradient in one pass, one 64-lane wavefront per ray:
//   distortion loss                 ns/model_components/losses.py:130-149 (mip-NeRF 360)
//   z-anti-aliased interlevel loss  ns/model_components/PreSight/losses.py:127-206 (Zip-NeRF)
// The reference materialises [R,64,64] and [R,130,129] temporaries for these (30 % of its CPU step time);
// here every ray lives in registers + ~3 KiB of LDS.
#include "common.hpp"

namespace {

constexpr int kMaxS = 128;           // samples of the final level
constexpr int kMaxM = 2 * (kMaxS + 1);
constexpr int kMaxSp = 256;          // samples of a proposal level

// in-place inclusive scan of an LDS array a[0..len) by one wavefront (lane owns a contiguous chunk)
__device__ __forceinline__ void wave_scan_lds(float* a, int len) {
  const int lane = ps_lane();
  const int ch = (len + 63) / 64;
  const int b = lane * ch;
  float local = 0.f;
  for (int i = 0; i < ch; ++i)
    if (b + i < len) local += a[b + i];
  float run = ps_wave_incl_scan(local) - local;
  for (int i = 0; i < ch; ++i)
    if (b + i < len) {
      run += a[b + i];
      a[b + i] = run;
    }
  __builtin_amdgcn_wave_barrier();
}

// gscale: factor on the stored gradient (1 = the plain derivative of the per-ray value; ps_*_loss_scaled passes
// loss_mult / R * the backward seed, so that the loss node's backward has nothing left to launch)
__global__ __launch_bounds__(256) void distortion_kernel(const float* __restrict__ sbins, const float* __restrict__ w,
                                                         int64_t R, int S, float* __restrict__ per_ray,
                                                         float* __restrict__ dw, float gscale) {
  __shared__ float lds[4][2][kMaxSp];
  const int wv = threadIdx.x >> 6, lane = ps_lane();
  const int64_t ray = blockIdx.x * 4 + wv;
  if (ray < R) {
    float* sw = lds[wv][0];
    float* su = lds[wv][1];
    const float* b = sbins + ray * (S + 1);
    for (int s = lane; s < S; s += 64) {
      sw[s] = w[ray * S + s];
      su[s] = (b[s + 1] + b[s]) / 2.0f;
    }
    __builtin_amdgcn_wave_barrier();
    float total = 0.f;
    for (int k = lane; k < S; k += 64) {
      const float wk = sw[k], uk = su[k];
      float acc = 0.f;
      for (int j = 0; j < S; ++j) acc += sw[j] * fabsf(uk - su[j]);
      const float delta = b[k + 1] - b[k];
      total += wk * acc + wk * wk * delta / 3.0f;
      const float g = 2.0f * acc + 2.0f * wk * delta / 3.0f;
      dw[ray * S + k] = g * gscale;
    }
    total = ps_wave_sum(total);
    if (lane == 0) per_ray[ray] = total;
  }
}

__device__ __forceinline__ float nan_to_num0(float v) {
  if (isnan(v)) return 0.0f;
  if (isinf(v)) return v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  return v;
}

__host__ __device__ constexpr int interlevel_floats_per_ray(int S, int Sp) { return 2 * (S + 1) + 4 * 2 * (S + 1) + Sp + 1; }

__global__ __launch_bounds__(256) void interlevel_kernel(const float* __restrict__ c_, const float* __restrict__ w_,
                                                         const float* __restrict__ cp_, const float* __restrict__ wp_,
                                                         int64_t R, int S, int Sp, float r, float* __restrict__ per_ray,
                                                         float* __restrict__ dwp, float gscale) {
  // per wave: A,B [n] | xr [m] | v2 [m] | yr [m] | cdf [m] | ret [Sp+1] -- sized for THIS call's S / Sp (dynamic LDS): the kernel is
  // a chain of dependent LDS look-ups (binary searches, scans), so what it needs is resident waves, and at the maximal sizes
  // (S 128, Sp 256: 6.2 KB per ray) a CU held 24 instead of 32
  extern __shared__ float lds_dyn[];
  const int wv = threadIdx.x >> 6, lane = ps_lane();
  const int64_t ray = blockIdx.x * 4 + wv;
  if (ray < R) {
  const int n = S + 1, m = 2 * n;
  float* A = lds_dyn + (size_t)wv * interlevel_floats_per_ray(S, Sp);
  float* B = A + n;
  float* xr = B + n;
  float* v2 = xr + m;
  float* yr = v2 + m;
  float* cdf = yr + m;
  float* ret = cdf + m;
  const float* c = c_ + ray * n;
  const float* w = w_ + ray * S;
  // a/b: shifted edges and the derivative pulses y1 (blur_stepfun, PreSight/losses.py:127-133)
  for (int i = lane; i < n; i += 64) {
    A[i] = c[i] - r;
    B[i] = c[i] + r;
  }
  __builtin_amdgcn_wave_barrier();
  // c: merge by rank (both sequences ascending; ties keep the concatenation order [A | B])
  for (int i = lane; i < n; i += 64) {
    const float wn_i = (i < S) ? w[i] / (c[i + 1] - c[i]) : 0.0f;
    const float wn_p = (i > 0) ? w[i - 1] / (c[i] - c[i - 1]) : 0.0f;
    const float y1 = (wn_i - wn_p) / (2.0f * r);
    // rank of A_i: i + #{j : B_j < A_i}
    int lo = 0, hi = n;
    const float a = A[i];
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (B[mid] < a) lo = mid + 1; else hi = mid;
    }
    xr[i + lo] = a;
    v2[i + lo] = y1;
    // rank of B_i: i + #{j : A_j <= B_i}
    lo = 0;
    hi = n;
    const float bb = B[i];
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (A[mid] <= bb) lo = mid + 1; else hi = mid;
    }
    xr[i + lo] = bb;
    v2[i + lo] = -y1;
  }
  __builtin_amdgcn_wave_barrier();
  // d: yr = [0, clamp_min(cumsum(dx * cumsum(y2)), 0)]
  wave_scan_lds(v2, m - 1);                              // cs_k
  for (int k = lane; k < m - 1; k += 64) yr[k + 1] = (xr[k + 1] - xr[k]) * v2[k];
  if (lane == 0) yr[0] = 0.0f;
  __builtin_amdgcn_wave_barrier();
  wave_scan_lds(yr + 1, m - 1);
  for (int k = lane; k < m - 1; k += 64) yr[k + 1] = fmaxf(yr[k + 1], 0.0f);
  __builtin_amdgcn_wave_barrier();
  // e: trapezoid areas -> cdf
  for (int k = lane; k < m - 1; k += 64) cdf[k + 1] = 0.5f * (yr[k + 1] + yr[k]) * (xr[k + 1] - xr[k]);
  if (lane == 0) cdf[0] = 0.0f;
  __builtin_amdgcn_wave_barrier();
  wave_scan_lds(cdf + 1, m - 1);
  // f: sorted_interp_quad at the proposal bin edges (PreSight/losses.py:141-164)
  const float* cp = cp_ + ray * (Sp + 1);
  for (int q = lane; q <= Sp; q += 64) {
    const float x = cp[q];
    int lo = 0, hi = m;  // cnt = #{i : xr_i <= x}
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (xr[mid] <= x) lo = mid + 1; else hi = mid;
    }
    const int cnt = lo;
    const float xp0 = cnt > 0 ? xr[cnt - 1] : xr[0];
    const float xp1 = cnt < m ? xr[cnt] : xr[m - 1];
    // torch.max / torch.min return the FIRST index of the extreme value
    int i0 = 0;
    float c0 = cdf[0];
    if (cnt > 0) {
      c0 = cdf[cnt - 1];
      int l2 = 0, h2 = cnt - 1;  // first i in [0,cnt) with cdf_i == c0 (cdf is non-decreasing)
      while (l2 < h2) {
        const int mid = (l2 + h2) >> 1;
        if (cdf[mid] < c0) l2 = mid + 1; else h2 = mid;
      }
      i0 = l2;
    }
    int i1 = 0;
    float c1 = cdf[m - 1];
    if (cnt < m) {
      c1 = cdf[cnt];
      i1 = (cnt > 0 && cdf[cnt] == cdf[m - 1]) ? 0 : cnt;
    }
    const float p0 = yr[i0], p1 = yr[i1];
    float off = nan_to_num0((x - xp0) / (xp1 - xp0));
    off = fminf(fmaxf(off, 0.0f), 1.0f);
    ret[q] = c0 + (x - xp0) * (p0 + p1 * off + p0 * (1.0f - off)) / 2.0f;
    (void)c1;
  }
  __builtin_amdgcn_wave_barrier();
  // g: loss and its gradient w.r.t. the proposal weights
  float total = 0.f;
  for (int k = lane; k < Sp; k += 64) {
    const float wp = wp_[ray * Sp + k];
    const float ws = ret[k + 1] - ret[k];
    const float e = fmaxf(ws - wp, 0.0f);
    const float den = wp + 1e-5f;
    total += e * e / den;
    const float g = -2.0f * e / den - e * e / (den * den);
    dwp[ray * Sp + k] = g * gscale;
  }
  total = ps_wave_sum(total);
  if (lane == 0) per_ray[ray] = total;
  }
}

// ---- depth supervision (lidar / monodepth configs): ns/model_components/PreSight/losses.py:28-103 -----------------
// One wavefront per ray.  steps = sample midpoints / pose_scale (metres); keep[ray] = 1 < depth < upper_bound (and not
// sky); per_ray and dw are zero for rays that are not kept, so the caller's mean is sum(per_ray) / sum(keep).
__global__ __launch_bounds__(256) void line_of_sight_kernel(const float* __restrict__ w, const float* __restrict__ ebins,
                                                            const float* __restrict__ depth, const float* __restrict__ sky,
                                                            int64_t R, int S, float sigma, float upper_bound, float pose_scale,
                                                            float var2, float log_norm, float* __restrict__ per_ray,
                                                            float* __restrict__ dw, float* __restrict__ keep) {
  const int wv = threadIdx.x >> 6, lane = ps_lane();
  const int64_t ray = blockIdx.x * 4 + wv;
  if (ray >= R) return;
  const float d = depth[ray];
  const bool k = d > 1.0f && d < upper_bound && (sky == nullptr || sky[ray] == 0.0f);
  const float* b = ebins + ray * (S + 1);
  float total = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float step = ((b[s] + b[s + 1]) / 2.0f) / pose_scale;
    const float wk = w[ray * S + s];
    const float x = step - d;
    const bool near = step <= d + sigma && step >= d - sigma;
    const bool empty = step < d - sigma;
    const float diff = wk - expf(-(x * x) / var2 - log_norm);
    float l = 0.f, g = 0.f;
    if (near) {
      l += diff * diff;
      g += 2.0f * diff;
    }
    if (empty) {
      l += wk * wk;
      g += 2.0f * wk;
    }
    total += l;
    dw[ray * S + s] = k ? g : 0.0f;
  }
  total = ps_wave_sum(total);
  if (lane == 0) {
    per_ray[ray] = k ? total : 0.0f;
    keep[ray] = k ? 1.0f : 0.0f;
  }
}

__global__ void expected_depth_kernel(const float* __restrict__ depth, const float* __restrict__ pred, const float* __restrict__ sky,
                                      int64_t R, float upper_bound, int inverse, float pose_scale, float* __restrict__ per_ray,
                                      float* __restrict__ dpred, float* __restrict__ keep) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= R) return;
  const float d = depth[i];
  const bool k = d > 1.0f && d < upper_bound && (sky == nullptr || sky[i] == 0.0f);
  const float p_m = pred[i] / pose_scale;  // metres
  float t, p, dp;
  if (inverse) {
    t = 1.0f / (d + 5.0f);
    p = 1.0f / (p_m + 5.0f);
    dp = -(p * p);
  } else {
    const float tn = d / upper_bound, pn = p_m / upper_bound;
    t = fminf(fmaxf(tn, 0.0f), 1.0f);
    p = fminf(fmaxf(pn, 0.0f), 1.0f);
    dp = (pn >= 0.0f && pn <= 1.0f) ? 1.0f / upper_bound : 0.0f;
  }
  const float e = t - p;
  per_ray[i] = k ? e * e : 0.0f;
  dpred[i] = k ? -2.0f * e * dp / pose_scale : 0.0f;
  keep[i] = k ? 1.0f : 0.0f;
}

}  // namespace

extern "C" int ps_distortion_loss(const float* sbins, const float* w, int64_t R, int S, float* per_ray, float* dw,
                                  void* stream) {
  PS_REQUIRE(S <= kMaxSp, "ps_distortion_loss: samples per ray must be <= 256");
  if (R == 0) return 0;
  distortion_kernel<<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(sbins, w, R, S, per_ray, dw, 1.0f);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_interlevel_loss(const float* c, const float* w, const float* cp, const float* wp, int64_t R, int S, int Sp,
                                  float pulse_width, float* per_ray, float* dwp, void* stream) {
  PS_REQUIRE(S <= kMaxS && Sp <= kMaxSp, "ps_interlevel_loss: S must be <= 128 and Sp <= 256");
  if (R == 0) return 0;
  interlevel_kernel<<<(unsigned)((R + 3) / 4), 256, 4 * interlevel_floats_per_ray(S, Sp) * sizeof(float), (hipStream_t)stream>>>(
      c, w, cp, wp, R, S, Sp, pulse_width, per_ray, dwp, 1.0f);
  PS_CHECK_LAUNCH();
}

// The same two losses with the chain rule folded in (round 6): the stored gradient is grad_scale * d(per-ray value); the caller passes
// loss_mult / count * <the seed of the backward pass>, so the loss node's backward launches nothing (ps_scale_grad's product, same bits).
extern "C" int ps_distortion_loss_scaled(const float* sbins, const float* w, int64_t R, int S, float* per_ray, float* dw, float grad_scale,
                                         void* stream) {
  PS_REQUIRE(S <= kMaxSp, "ps_distortion_loss_scaled: samples per ray must be <= 256");
  if (R == 0) return 0;
  distortion_kernel<<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(sbins, w, R, S, per_ray, dw, grad_scale);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_interlevel_loss_scaled(const float* c, const float* w, const float* cp, const float* wp, int64_t R, int S, int Sp,
                                         float pulse_width, float* per_ray, float* dwp, float grad_scale, void* stream) {
  PS_REQUIRE(S <= kMaxS && Sp <= kMaxSp, "ps_interlevel_loss_scaled: S must be <= 128 and Sp <= 256");
  if (R == 0) return 0;
  interlevel_kernel<<<(unsigned)((R + 3) / 4), 256, 4 * interlevel_floats_per_ray(S, Sp) * sizeof(float), (hipStream_t)stream>>>(
      c, w, cp, wp, R, S, Sp, pulse_width, per_ray, dwp, grad_scale);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_line_of_sight_loss(const float* w, const float* ebins, const float* depth, const float* sky, int64_t R, int S,
                                     float sigma, float upper_bound, float pose_scale, float* per_ray, float* dw, float* keep,
                                     void* stream) {
  PS_REQUIRE(sigma > 0.f && pose_scale > 0.f, "ps_line_of_sight_loss: sigma and pose_scale must be positive");
  if (R == 0) return 0;
  const float sd = sigma / 3.0f;  // URF_SIGMA_SCALE_FACTOR
  const float log_norm = (float)(log((double)sd) + 0.5 * log(2.0 * 3.14159265358979323846));
  line_of_sight_kernel<<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(w, ebins, depth, sky, R, S, sigma, upper_bound,
                                                                               pose_scale, 2.0f * (sd * sd), log_norm, per_ray,
                                                                               dw, keep);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_expected_depth_loss(const float* depth, const float* pred, const float* sky, int64_t R, float upper_bound,
                                      int inverse, float pose_scale, float* per_ray, float* dpred, float* keep, void* stream) {
  PS_REQUIRE(upper_bound > 0.f && pose_scale > 0.f, "ps_expected_depth_loss: upper_bound and pose_scale must be positive");
  if (R == 0) return 0;
  expected_depth_kernel<<<(unsigned)((R + 255) / 256), 256, 0, (hipStream_t)stream>>>(depth, pred, sky, R, upper_bound, inverse,
                                                                                    pose_scale, per_ray, dpred, keep);
  PS_CHECK_LAUNCH();
}
