// Per-ray "tail" of the training step: embedding lookup, sky blending and the scalar per-ray losses.  Each of these is a
// chain of 5-15 element-wise torch kernels in the reference (a few microseconds of work each, but one dependent launch
// each: ~10 us of GPU timeline apiece on MI355X); here every operator is one launch per direction.
//   embedding      ns/field_components/embedding.py:27-55 (nn.Embedding lookup; backward = scatter-add of the rows)
//   sky blend      ns/models/PreSight/nerfacto_nusc_ms.py:512-533  acc = clamp(acc,0,1); out += (1-acc) * sky_out
//   rgb / semantic MSE, sky BCE   nerfacto_nusc_ms.py:568, ns/model_components/PreSight/losses.py:106-125
#include "common.hpp"

namespace {

// out[r, col0 + d] = table[idx[r], d]
__global__ void embedding_fwd_kernel(const int64_t* __restrict__ idx, const float* __restrict__ table, int64_t R, int D,
                                     int out_stride, int col0, float* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= R * D) return;
  const int64_t r = i / D;
  const int d = (int)(i - r * D);
  out[r * out_stride + col0 + d] = table[idx[r] * D + d];
}

// dtable[idx[r], d] += dout[r, col0 + d].  Small tables (rows*D <= kEmbLds floats: the per-camera / per-video appearance
// codes) are first reduced in LDS per workgroup, so that the global atomics are one per touched table entry and
// workgroup instead of one per ray and column; large tables go straight to global atomics.
constexpr int kEmbLds = 12288;
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ idx, const float* __restrict__ dout, int64_t R,
                                                            int D, int rows, int out_stride, int col0, float* __restrict__ dtable) {
  __shared__ float acc[kEmbLds];
  const bool use_lds = (int64_t)rows * D <= kEmbLds;
  const int n_tab = rows * D;
  if (use_lds) {
    for (int i = threadIdx.x; i < n_tab; i += 256) acc[i] = 0.0f;
    __syncthreads();
  }
  const int64_t per = (R * D + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per, hi = min(R * D, lo + per);
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    const float g = dout[r * out_stride + col0 + d];
    const int64_t row = idx[r];
    if (use_lds)
      atomicAdd(&acc[row * D + d], g);
    else
      unsafeAtomicAdd(dtable + row * D + d, g);
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < n_tab; i += 256)
      if (acc[i] != 0.0f) unsafeAtomicAdd(dtable + i, acc[i]);
  }
}

// Two tables in one launch each way (the model's per-camera appearance code + per-video code, nerfacto_nusc_ms.py:472-485), indices
// read through an element stride (camera index = column 0 of ray_indices [R,3]: no contiguous copy): blockIdx.y = table.
struct EmbedPair {
  const int64_t* idx[2];
  int64_t stride[2];
  const float* table[2];
  float* dtable[2];
  int D[2], rows[2], col0[2];
};
__global__ void embedding_pair_fwd_kernel(EmbedPair e, int64_t R, int out_stride, float* __restrict__ out) {
  const int t = blockIdx.y, D = e.D[t];
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= R * D) return;
  const int64_t r = i / D;
  const int d = (int)(i - r * D);
  out[r * out_stride + e.col0[t] + d] = e.table[t][e.idx[t][r * e.stride[t]] * D + d];
}
__global__ __launch_bounds__(256) void embedding_pair_bwd_kernel(EmbedPair e, const float* __restrict__ dout, int64_t R, int out_stride) {
  __shared__ float acc[kEmbLds];
  const int t = blockIdx.y, D = e.D[t], rows = e.rows[t], col0 = e.col0[t];
  const int64_t* __restrict__ idx = e.idx[t];
  const int64_t st = e.stride[t];
  float* __restrict__ dtable = e.dtable[t];
  const bool use_lds = (int64_t)rows * D <= kEmbLds;
  const int n_tab = rows * D;
  if (use_lds) {
    for (int i = threadIdx.x; i < n_tab; i += 256) acc[i] = 0.0f;
    __syncthreads();
  }
  const int64_t per = (R * D + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per, hi = min(R * D, lo + per);
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    const float g = dout[r * out_stride + col0 + d];
    const int64_t row = idx[r * st];
    if (use_lds)
      atomicAdd(&acc[row * D + d], g);
    else
      unsafeAtomicAdd(dtable + row * D + d, g);
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < n_tab; i += 256)
      if (acc[i] != 0.0f) unsafeAtomicAdd(dtable + i, acc[i]);
  }
}

// One wavefront per ray, lane = channel.  sky_* may be null (no sky model): plain clamp of the accumulation.
__global__ __launch_bounds__(256) void sky_blend_fwd_kernel(const float* __restrict__ rgb_f, const float* __restrict__ acc_raw,
                                                            const float* __restrict__ sem_f, const float* __restrict__ sky_rgb,
                                                            const float* __restrict__ sky_sem, int64_t R, int C,
                                                            float* __restrict__ rgb, float* __restrict__ acc, float* __restrict__ sem) {
  const int64_t ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int lane = ps_lane();
  const float a = fminf(fmaxf(acc_raw[ray], 0.0f), 1.0f);
  if (lane == 0) acc[ray] = a;
  if (lane < 3) rgb[ray * 3 + lane] = rgb_f[ray * 3 + lane] + (sky_rgb ? (1.0f - a) * sky_rgb[ray * 3 + lane] : 0.0f);
  if (sem != nullptr && lane < C) sem[ray * C + lane] = sem_f[ray * C + lane] + (sky_sem ? (1.0f - a) * sky_sem[ray * C + lane] : 0.0f);
}

// d(rgb_f) = d(rgb), d(sem_f) = d(sem) (identity: the caller passes the same tensors on); here
//   d(sky_rgb) = (1-acc) d(rgb), d(sky_sem) = (1-acc) d(sem),
//   d(acc_raw) = [0 <= acc_raw <= 1] * (d(acc) - sum_c d(rgb) sky_rgb - sum_c d(sem) sky_sem)      (torch.clamp passes the bounds)
__global__ __launch_bounds__(256) void sky_blend_bwd_kernel(const float* __restrict__ acc_raw, const float* __restrict__ sky_rgb,
                                                            const float* __restrict__ sky_sem, const float* __restrict__ d_rgb,
                                                            const float* __restrict__ d_acc, const float* __restrict__ d_sem,
                                                            int64_t R, int C, float* __restrict__ d_acc_raw,
                                                            float* __restrict__ d_sky_rgb, float* __restrict__ d_sky_sem) {
  const int64_t ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int lane = ps_lane();
  const float ar = acc_raw[ray];
  const float a = fminf(fmaxf(ar, 0.0f), 1.0f);
  float dot = 0.f;
  if (sky_rgb != nullptr && d_rgb != nullptr && lane < 3) {
    const float g = d_rgb[ray * 3 + lane];
    dot += g * sky_rgb[ray * 3 + lane];
    d_sky_rgb[ray * 3 + lane] = (1.0f - a) * g;
  } else if (d_sky_rgb != nullptr && lane < 3) {
    d_sky_rgb[ray * 3 + lane] = 0.0f;
  }
  if (sky_sem != nullptr && d_sem != nullptr && lane < C) {
    const float g = d_sem[ray * C + lane];
    dot += g * sky_sem[ray * C + lane];
    d_sky_sem[ray * C + lane] = (1.0f - a) * g;
  } else if (d_sky_sem != nullptr && lane < C) {
    d_sky_sem[ray * C + lane] = 0.0f;
  }
  dot = ps_wave_sum(dot);
  if (lane == 0) d_acc_raw[ray] = (ar >= 0.0f && ar <= 1.0f) ? (d_acc ? d_acc[ray] : 0.0f) - dot : 0.0f;
}

// sum of squared errors of a block of elements (one partial per workgroup) + d/dpred of the MEAN over n elements
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n,
                                                  int clip_target, float* __restrict__ partial, float* __restrict__ dpred) {
  __shared__ float red[4];
  float s = 0.f;
  const float k = 2.0f / (float)n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float t = target[i];
    if (clip_target) t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float e = pred[i] - t;
    s += e * e;
    dpred[i] = k * e;
  }
  s = ps_wave_sum(s);
  if (ps_lane() == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// binary cross entropy of clip(acc, eps, 1-eps) against (1 - sky_mask), torch semantics (log clamped at -100)
__global__ __launch_bounds__(256) void sky_bce_kernel(const float* __restrict__ acc, const float* __restrict__ sky_mask, int64_t R,
                                                      float eps, float* __restrict__ partial, float* __restrict__ dacc) {
  __shared__ float red[4];
  float s = 0.f;
  const float inv = 1.0f / (float)R;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < R; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = acc[i], t = 1.0f - sky_mask[i];
    const float a = fminf(fmaxf(x, eps), 1.0f - eps);
    const float la = fmaxf(logf(a), -100.0f), lb = fmaxf(logf(1.0f - a), -100.0f);
    s -= t * la + (1.0f - t) * lb;
    // d/da = (a - t) / (a (1 - a)), through the clip only where eps <= acc <= 1-eps
    dacc[i] = (x >= eps && x <= 1.0f - eps) ? inv * (a - t) / fmaxf(a * (1.0f - a), 1e-12f) : 0.0f;
  }
  s = ps_wave_sum(s);
  if (ps_lane() == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}


// ---- data feed: one training batch out of a device-resident pixel chunk ---------------------------
// ns/data/PreSight/my_dataset.py:52-73 (ImageChunk.__getitem__) + the DataLoader's collate: for every drawn pixel id
// gather rgb / sky / depth / features / video id and form ray_index = (image, pixel // width, pixel % width).
// One wavefront per ray: lane = feature channel (C <= 64), so the 256-byte feature row is one coalesced load.
__global__ __launch_bounds__(256) void gather_batch_kernel(const int64_t* __restrict__ pick, int64_t R, const float* __restrict__ rgbs,
                                                           const float* __restrict__ skies, const float* __restrict__ depths,
                                                           const float* __restrict__ features, int C,
                                                           const int64_t* __restrict__ pixel_indices,
                                                           const int64_t* __restrict__ image_indices,
                                                           const int64_t* __restrict__ video_ids, const int64_t* __restrict__ widths,
                                                           int64_t* __restrict__ ray_indices, float* __restrict__ o_rgb,
                                                           float* __restrict__ o_sky, float* __restrict__ o_depth,
                                                           float* __restrict__ o_feat, int64_t* __restrict__ o_video) {
  const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = ps_lane();
  const int64_t i = pick[r];
  if (features != nullptr && lane < C) o_feat[r * C + lane] = features[i * C + lane];
  if (lane < 3) o_rgb[r * 3 + lane] = rgbs[i * 3 + lane];
  if (lane == 0) {
    const int64_t px = pixel_indices[i], w = widths[i];
    ray_indices[r * 3] = image_indices[i];
    ray_indices[r * 3 + 1] = px / w;
    ray_indices[r * 3 + 2] = px % w;
    if (skies != nullptr) o_sky[r] = skies[i];
    if (depths != nullptr) o_depth[r] = depths[i];
    o_video[r] = video_ids[i];
  }
}

}  // namespace

extern "C" int ps_embedding_fwd(const int64_t* idx, const float* table, int64_t R, int D, int out_stride, int col0, float* out,
                                void* stream) {
  if (R == 0 || D == 0) return 0;
  embedding_fwd_kernel<<<(unsigned)((R * D + 255) / 256), 256, 0, (hipStream_t)stream>>>(idx, table, R, D, out_stride, col0, out);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_embedding_bwd(const int64_t* idx, const float* dout, int64_t R, int D, int rows, int out_stride, int col0,
                                float* dtable, void* stream) {
  if (R == 0 || D == 0) return 0;
  const int64_t want = (R * D + 256 * 32 - 1) / (256 * 32);
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 128 ? 128 : want));
  embedding_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(idx, dout, R, D, rows, out_stride, col0, dtable);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_embedding_pair_fwd(const int64_t* idx0, int64_t stride0, const float* table0, int D0, const int64_t* idx1, int64_t stride1,
                                     const float* table1, int D1, int64_t R, float* out, void* stream) {
  PS_REQUIRE(idx0 && table0 && idx1 && table1 && out && D0 > 0 && D1 > 0 && stride0 >= 1 && stride1 >= 1, "ps_embedding_pair_fwd: null argument");
  if (R == 0) return 0;
  EmbedPair e{{idx0, idx1}, {stride0, stride1}, {table0, table1}, {nullptr, nullptr}, {D0, D1}, {0, 0}, {0, D0}};
  const int Dm = D0 > D1 ? D0 : D1;
  embedding_pair_fwd_kernel<<<dim3((unsigned)((R * Dm + 255) / 256), 2), 256, 0, (hipStream_t)stream>>>(e, R, D0 + D1, out);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_embedding_pair_bwd(const int64_t* idx0, int64_t stride0, int rows0, int D0, float* dtable0, const int64_t* idx1,
                                     int64_t stride1, int rows1, int D1, float* dtable1, const float* dout, int64_t R, void* stream) {
  PS_REQUIRE(idx0 && dtable0 && idx1 && dtable1 && dout && D0 > 0 && D1 > 0 && stride0 >= 1 && stride1 >= 1, "ps_embedding_pair_bwd: null argument");
  if (R == 0) return 0;
  EmbedPair e{{idx0, idx1}, {stride0, stride1}, {nullptr, nullptr}, {dtable0, dtable1}, {D0, D1}, {rows0, rows1}, {0, D0}};
  const int Dm = D0 > D1 ? D0 : D1;
  const int64_t want = (R * Dm + 256 * 32 - 1) / (256 * 32);
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 128 ? 128 : want));
  embedding_pair_bwd_kernel<<<dim3(grid, 2), 256, 0, (hipStream_t)stream>>>(e, dout, R, D0 + D1);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_sky_blend_fwd(const float* rgb_f, const float* acc_raw, const float* sem_f, const float* sky_rgb,
                                const float* sky_sem, int64_t R, int C, float* rgb, float* acc, float* sem, void* stream) {
  PS_REQUIRE(C <= 64, "ps_sky_blend_fwd: at most 64 semantic channels");
  if (R == 0) return 0;
  sky_blend_fwd_kernel<<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(rgb_f, acc_raw, sem_f, sky_rgb, sky_sem, R, C, rgb,
                                                                                acc, sem);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_sky_blend_bwd(const float* acc_raw, const float* sky_rgb, const float* sky_sem, const float* d_rgb,
                                const float* d_acc, const float* d_sem, int64_t R, int C, float* d_acc_raw, float* d_sky_rgb,
                                float* d_sky_sem, void* stream) {
  PS_REQUIRE(C <= 64, "ps_sky_blend_bwd: at most 64 semantic channels");
  if (R == 0) return 0;
  sky_blend_bwd_kernel<<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(acc_raw, sky_rgb, sky_sem, d_rgb, d_acc, d_sem, R, C,
                                                                                d_acc_raw, d_sky_rgb, d_sky_sem);
  PS_CHECK_LAUNCH();
}

// value of a mean-reduced loss from its terms in ONE launch: out = scale * sum(terms) / D with D = sum(keep) when a validity
// mask is given (the depth losses average over the qualifying rays only), else `denom`; inv = scale / D is what the backward
// multiplies the stored per-element gradient with.  One workgroup of 1024 threads (n is a ray count or a partial-sum count:
// a 65 536-ray batch is 16 independent 16-byte loads per thread, all in flight at once).
__device__ __forceinline__ float sum_strided(const float* __restrict__ v, int64_t n) {
  float s = 0.f;
  const int64_t n4 = (reinterpret_cast<uintptr_t>(v) & 15) == 0 ? n / 4 : 0;
#pragma unroll 16
  for (int64_t i = threadIdx.x; i < n4; i += 1024) {
    const f32x4 x = reinterpret_cast<const f32x4*>(v)[i];
    s += (x[0] + x[1]) + (x[2] + x[3]);
  }
  for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 1024) s += v[i];
  return s;
}
__global__ __launch_bounds__(1024) void loss_finish_kernel(const float* __restrict__ terms, int64_t n, const float* __restrict__ keep,
                                                           int64_t n_keep, float denom, float scale, float* __restrict__ out,
                                                           float* __restrict__ inv) {
  __shared__ float red[32];
  float s = sum_strided(terms, n), k = keep != nullptr ? sum_strided(keep, n_keep) : 0.f;
  s = ps_wave_sum(s);
  k = ps_wave_sum(k);
  if (ps_lane() == 0) {
    red[threadIdx.x >> 6] = s;
    red[16 + (threadIdx.x >> 6)] = k;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = 0.f, kt = 0.f;
    for (int w = 0; w < 16; ++w) {
      total += red[w];
      kt += red[16 + w];
    }
    const float d = keep != nullptr ? kt : denom;
    out[0] = scale * (total / d);  // d == 0: NaN, like torch.mean of an empty selection
    if (inv != nullptr) inv[0] = scale / d;
  }
}

// out = grad * g[0] * (factor ? factor[0] : 1) * host_scale: the chain rule of a scalar loss in one launch (g is the device
// scalar autograd hands over)
__global__ __launch_bounds__(256) void scale_grad_kernel(const float* __restrict__ grad, int64_t n, const float* __restrict__ g,
                                                         const float* __restrict__ factor, float host_scale, float* __restrict__ out,
                                                         int vec /* both arrays 16-byte aligned */) {
  const float f = g[0] * (factor != nullptr ? factor[0] : 1.0f) * host_scale;
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    if (vec && i + 4 <= n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(grad + i);
      *reinterpret_cast<f32x4*>(out + i) = v * f;
    } else {
      for (int64_t q = i; q < n && q < i + 4; ++q) out[q] = grad[q] * f;
    }
  }
}

// number of partial sums ps_mse_loss / ps_sky_bce_loss write for n elements
extern "C" int ps_loss_partials(int64_t n) {
  const int64_t b = (n + 256 * 8 - 1) / (256 * 8);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int ps_mse_loss(const float* pred, const float* target, int64_t n, int clip_target, float* partial, float* dpred,
                           void* stream) {
  PS_REQUIRE(n > 0, "ps_mse_loss: empty input");
  mse_kernel<<<ps_loss_partials(n), 256, 0, (hipStream_t)stream>>>(pred, target, n, clip_target, partial, dpred);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_sky_bce_loss(const float* acc, const float* sky_mask, int64_t R, float eps, float* partial, float* dacc,
                               void* stream) {
  PS_REQUIRE(R > 0, "ps_sky_bce_loss: empty input");
  sky_bce_kernel<<<ps_loss_partials(R), 256, 0, (hipStream_t)stream>>>(acc, sky_mask, R, eps, partial, dacc);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_loss_finish(const float* terms, int64_t n, const float* keep, int64_t n_keep, float denom, float scale, float* out,
                              float* inv, void* stream) {
  PS_REQUIRE(n > 0 && out != nullptr, "ps_loss_finish: no terms");
  loss_finish_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(terms, n, keep, keep != nullptr ? n_keep : 0, denom, scale, out, inv);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_scale_grad(const float* grad, int64_t n, const float* g, const float* factor, float host_scale, float* out,
                             void* stream) {
  if (n == 0) return 0;
  PS_REQUIRE(g != nullptr, "ps_scale_grad: the upstream gradient is a device scalar");
  const int64_t b = (n + 1023) / 1024;
  const int vec = (((uintptr_t)grad | (uintptr_t)out) & 15) == 0;
  scale_grad_kernel<<<(unsigned)(b > 2048 ? 2048 : b), 256, 0, (hipStream_t)stream>>>(grad, n, g, factor, host_scale, out, vec);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_gather_batch(const int64_t* pick, int64_t R, const float* rgbs, const float* skies, const float* depths,
                               const float* features, int C, const int64_t* pixel_indices, const int64_t* image_indices,
                               const int64_t* video_ids, const int64_t* widths, int64_t* ray_indices, float* o_rgb, float* o_sky,
                               float* o_depth, float* o_feat, int64_t* o_video, void* stream) {
  PS_REQUIRE(C <= 64, "ps_gather_batch: at most 64 feature channels");
  if (R == 0) return 0;
  gather_batch_kernel<<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(pick, R, rgbs, skies, depths, features, C,
                                                                               pixel_indices, image_indices, video_ids, widths,
                                                                               ray_indices, o_rgb, o_sky, o_depth, o_feat, o_video);
  PS_CHECK_LAUNCH();
}
