// Prior-extraction helpers (BASELINE config 5):
//   ps_voxel_index   : Open3D voxel_down_sample_and_trace index rule used by ns/scripts/extract_priors.py:216-245,
//                      idx = floor((p - (min_bound - voxel/2)) / voxel) per axis, evaluated in fp64 -> int64 (bit exact)
//   ps_lattice_points: the dense res^3 query lattice over a tile AABB (z fastest), cell centres
//   ps_mean_density  : mean of the proposal-net and main-field densities (extract_priors.py:133-137)
//   ps_gather_clip_f16: the kept points' features, clipped to [0, 1] and stored as fp16 (extract_priors.py:136-138), in one pass
#include <hip/hip_fp16.h>
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

namespace {
// rows idx[i] of src [n, C] (fp32) -> clamp to [lo, hi] -> fp16.  C = 64 (the feature width of the prior extraction): 16 lanes per row,
// 16 bytes in / 8 bytes out per lane, four rows per wavefront instruction; other widths: one lane per channel
__device__ __forceinline__ float clampf_nan(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }  // torch.clamp: a NaN stays a NaN
__global__ __launch_bounds__(256) void gather_clip_f16_c64_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, int64_t m, float lo,
                                                                  float hi, __half* __restrict__ out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t i = t >> 4;
  if (i >= m) return;
  const int q = (int)(t & 15);
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + idx[i] * 64 + 4 * q));
  const __half2 a = __floats2half2_rn(clampf_nan(v.x, lo, hi), clampf_nan(v.y, lo, hi));
  const __half2 b = __floats2half2_rn(clampf_nan(v.z, lo, hi), clampf_nan(v.w, lo, hi));
  uint2 o;
  o.x = *reinterpret_cast<const unsigned*>(&a);
  o.y = *reinterpret_cast<const unsigned*>(&b);
  *reinterpret_cast<uint2*>(out + i * 64 + 4 * q) = o;
}
__global__ __launch_bounds__(256) void gather_clip_f16_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, int64_t m, int C,
                                                              float lo, float hi, __half* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= m || lane >= C) return;
  out[i * C + lane] = __float2half_rn(clampf_nan(src[idx[i] * C + lane], lo, hi));
}
}  // namespace

extern "C" int ps_gather_clip_f16(const float* src, const int64_t* idx, int64_t m, int C, float lo, float hi, void* out_f16, void* stream) {
  if (m == 0) return 0;
  PS_REQUIRE(C >= 1 && C <= 64, "ps_gather_clip_f16: 1..64 channels");
  if (C == 64)
    gather_clip_f16_c64_kernel<<<(unsigned)((m * 16 + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, idx, m, lo, hi, (__half*)out_f16);
  else
    gather_clip_f16_kernel<<<(unsigned)((m + 3) / 4), 256, 0, (hipStream_t)stream>>>(src, idx, m, C, lo, hi, (__half*)out_f16);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_mean_density(const float* a, const float* b, const float* c, int64_t n, float* out, void* stream) {
  if (n == 0) return 0;
  mean3_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(a, b, c, n, out);
  PS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------------------
// Voxel down-sampling of the extracted points (ns/scripts/extract_priors.py:151-191, 216-245): Open3D's
// voxel_down_sample_and_trace groups the points by integer voxel index and returns, per voxel, the mean point and the member
// lists; the script then averages the members' colours (fp32) and features (fp16 -> fp64 mean -> fp16) and counts the hits.
// Here:  ps_voxel_keys      one int64 key per point (x-major linearisation of the bit-exact voxel index of ps_voxel_index),
//        [caller: stable sort of the keys, run lengths of equal keys]
//        ps_voxel_reduce    one 64-lane wavefront per voxel walks the run: lane c sums feature channel c of every member in
//                           fp64 (a member's 64 fp16 channels are one coalesced 128-byte row), lanes 0-2 / 3-5 also carry
//                           the point / colour sums; means are written in the reference's output types.
// Members are visited in input order (stable sort), so the fp64 sums are reproducible.
namespace {

__global__ void voxel_keys_kernel(const float* __restrict__ pts, int64_t n, double voxel, double mx, double my, double mz, int64_t ny,
                                  int64_t nz, int64_t* __restrict__ keys) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double h = voxel * 0.5;
  const int64_t ix = (int64_t)floor(((double)pts[i * 3 + 0] - (mx - h)) / voxel);
  const int64_t iy = (int64_t)floor(((double)pts[i * 3 + 1] - (my - h)) / voxel);
  const int64_t iz = (int64_t)floor(((double)pts[i * 3 + 2] - (mz - h)) / voxel);
  keys[i] = (ix * ny + iy) * nz + iz;
}

// order [n]: point index of sorted position; starts / counts [V]: run of every voxel in the sorted order.
// sums != null: also write the raw fp64 sums ([V, 3 + 3 + 64]: point, colour, feature) for a later merge of partial results
__global__ __launch_bounds__(256) void voxel_reduce_kernel(const int64_t* __restrict__ order, const int64_t* __restrict__ starts,
                                                           const int64_t* __restrict__ counts, int64_t V, const float* __restrict__ pts,
                                                           const __half* __restrict__ feats, const float* __restrict__ colors, int C,
                                                           float* __restrict__ o_pts, __half* __restrict__ o_feat,
                                                           float* __restrict__ o_col, double* __restrict__ sums) {
  const int64_t v = blockIdx.x * (int64_t)(blockDim.x / 64) + (threadIdx.x >> 6);
  if (v >= V) return;
  const int lane = threadIdx.x & 63;
  const int64_t s = starts[v], c = counts[v];
  double fsum = 0.0, psum = 0.0;
  // A voxel holds ~4 points: walked one by one, every member cost two dependent round trips (its index, then its row).  The indices of
  // up to 64 members are fetched at once (one per lane) and the rows of four members are requested together; the sums still run in
  // member order (same fp64 results).
  for (int64_t m0 = 0; m0 < c; m0 += 64) {
    const int64_t nm = (c - m0) < 64 ? (c - m0) : 64;
    const int64_t mine = lane < nm ? order[s + m0 + lane] : 0;
    for (int64_t m = 0; m < nm; m += 4) {
      int64_t p[4];
      float f[4], q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) p[k] = __shfl(mine, (int)((m + k) < nm ? (m + k) : m), 64);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f[k] = lane < C ? __half2float(feats[p[k] * C + lane]) : 0.0f;
        q[k] = lane < 3 ? pts[p[k] * 3 + lane] : ((lane < 6 && colors != nullptr) ? colors[p[k] * 3 + (lane - 3)] : 0.0f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (m + k < nm) {
          fsum += (double)f[k];
          psum += (double)q[k];
        }
    }
  }
  const double inv = 1.0 / (double)c;
  if (lane < C) o_feat[v * C + lane] = __float2half_rn((float)(fsum * inv));  // fp64 mean -> fp16, like .astype(np.float16)
  if (lane < 3) o_pts[v * 3 + lane] = (float)(psum * inv);
  else if (lane < 6 && colors != nullptr) o_col[v * 3 + (lane - 3)] = (float)(psum * inv);
  if (sums != nullptr) {
    double* o = sums + v * (6 + C);
    if (lane < 6) o[lane] = (lane < 3 || colors != nullptr) ? psum : 0.0;
    if (lane < C) o[6 + lane] = fsum;
  }
}

// C even: HALF a wavefront per voxel, a lane carries two channels (one 4-byte load per member row instead of 2 bytes); same sums in the
// same (member) order per channel
__global__ __launch_bounds__(256) void voxel_reduce_h2_kernel(const int64_t* __restrict__ order, const int64_t* __restrict__ starts,
                                                              const int64_t* __restrict__ counts, int64_t V, const float* __restrict__ pts,
                                                              const __half* __restrict__ feats, const float* __restrict__ colors, int C,
                                                              float* __restrict__ o_pts, __half* __restrict__ o_feat,
                                                              float* __restrict__ o_col, double* __restrict__ sums) {
  const int64_t v = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 5;
  if (v >= V) return;
  const int lane = threadIdx.x & 31, C2 = C / 2;
  const int64_t s = starts[v], c = counts[v];
  double f0 = 0.0, f1 = 0.0, psum = 0.0;
  for (int64_t m0 = 0; m0 < c; m0 += 32) {
    const int64_t nm = (c - m0) < 32 ? (c - m0) : 32;
    const int64_t mine = lane < nm ? order[s + m0 + lane] : 0;
    for (int64_t m = 0; m < nm; m += 4) {
      int64_t p[4];
      __half2 f[4];
      float q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) p[k] = __shfl(mine, (int)((m + k) < nm ? (m + k) : m), 32);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f[k] = lane < C2 ? *reinterpret_cast<const __half2*>(feats + p[k] * C + 2 * lane) : __half2();
        q[k] = lane < 3 ? pts[p[k] * 3 + lane] : ((lane < 6 && colors != nullptr) ? colors[p[k] * 3 + (lane - 3)] : 0.0f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (m + k < nm) {
          f0 += (double)__low2float(f[k]);
          f1 += (double)__high2float(f[k]);
          psum += (double)q[k];
        }
    }
  }
  const double inv = 1.0 / (double)c;
  if (lane < C2) *reinterpret_cast<__half2*>(o_feat + v * C + 2 * lane) = __halves2half2(__float2half_rn((float)(f0 * inv)), __float2half_rn((float)(f1 * inv)));
  if (lane < 3) o_pts[v * 3 + lane] = (float)(psum * inv);
  else if (lane < 6 && colors != nullptr) o_col[v * 3 + (lane - 3)] = (float)(psum * inv);
  if (sums != nullptr) {
    double* o = sums + v * (6 + C);
    if (lane < 6) o[lane] = (lane < 3 || colors != nullptr) ? psum : 0.0;
    if (lane < C2) {
      o[6 + 2 * lane] = f0;
      o[6 + 2 * lane + 1] = f1;
    }
  }
}

}  // namespace

// keys[i] = (ix * ny + iy) * nz + iz with (ix, iy, iz) = ps_voxel_index(p_i); ny, nz = voxel counts along y / z (any bound >=
// the largest index + 1); min_bound is a HOST array
extern "C" int ps_voxel_keys(const float* pts, int64_t n, double voxel, const double* min_bound, int64_t ny, int64_t nz, int64_t* keys,
                             void* stream) {
  if (n == 0) return 0;
  PS_REQUIRE(voxel > 0.0 && ny > 0 && nz > 0, "ps_voxel_keys: voxel size and grid extents must be positive");
  voxel_keys_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(pts, n, voxel, min_bound[0], min_bound[1], min_bound[2],
                                                                                 ny, nz, keys);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_voxel_reduce(const int64_t* order, const int64_t* starts, const int64_t* counts, int64_t V, const float* pts,
                               const void* feats_f16, const float* colors, int C, float* o_pts, void* o_feat_f16, float* o_col,
                               double* sums, void* stream) {
  if (V == 0) return 0;
  PS_REQUIRE(C >= 1 && C <= 64, "ps_voxel_reduce: 1..64 feature channels (one lane each)");
  if (C % 2 == 0)
    voxel_reduce_h2_kernel<<<(unsigned)((V + 7) / 8), 256, 0, (hipStream_t)stream>>>(order, starts, counts, V, pts, (const __half*)feats_f16,
                                                                                     colors, C, o_pts, (__half*)o_feat_f16, o_col, sums);
  else
    voxel_reduce_kernel<<<(unsigned)((V + 3) / 4), 256, 0, (hipStream_t)stream>>>(order, starts, counts, V, pts, (const __half*)feats_f16, colors,
                                                                                  C, o_pts, (__half*)o_feat_f16, o_col, sums);
  PS_CHECK_LAUNCH();
}

// ---- kept rows of a lattice chunk without a host round trip (round 6) --------------------------------------------------------------
// ns/scripts/extract_priors.py:133-152 keeps the points whose mean density exceeds the threshold; as torch code that is
// nonzero(dens > thr) (a host synchronisation for the output size, once per 8 M-point chunk) followed by index_select / gather /
// clamp / convert / divide / voxel-index launches on the selected rows.  Here a chunk's selection is three launches and NO host sync:
//   count   kept points per 256-point workgroup
//   scan    one workgroup: exclusive prefix over the workgroup counts, shifted by the running total of the EARLIER chunks, which lives
//           in device memory (cursor[0]; cursor[1] = rows that did not fit the caller's capacity)
//   emit    every workgroup ranks its kept points (ballot + popcount), and writes their rows -- point / pose_scale, density,
//           clamp(features, 0, 1) as fp16, integer voxel index, global lattice row -- at their final position of the tile-wide arrays
// Rows come out in ascending lattice order, exactly as nonzero() lists them; the host reads cursor[] once, after the last chunk.
namespace {
constexpr int kKeptBlock = 256;

__global__ __launch_bounds__(kKeptBlock) void kept_count_kernel(const float* __restrict__ dens, int64_t n, float thr, unsigned* __restrict__ counts) {
  __shared__ unsigned wsum[kKeptBlock / 64];
  const int64_t i = (int64_t)blockIdx.x * kKeptBlock + threadIdx.x;
  const bool keep = i < n && dens[i] > thr;
  const unsigned c = (unsigned)__popcll(__ballot(keep));
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// counts[b] -> first output row of workgroup b (int64, in `offsets`); cursor[0] += the chunk's total.  One workgroup: thread t owns the
// contiguous run of ceil(nb / 1024) counts starting at t * per (two passes over its own run around ONE block-wide scan; the first
// version walked the counts in 1024-wide tiles with a block scan per tile: 47 us for the 32 k counts of an 8 M-point chunk).
__global__ __launch_bounds__(1024) void kept_scan_kernel(const unsigned* __restrict__ counts, int nb, int64_t* __restrict__ offsets,
                                                         int64_t* __restrict__ cursor, int64_t capacity) {
  __shared__ unsigned long long wsum[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = cursor[0];
  const int per = (nb + 1023) / 1024;
  const int b0 = (int)threadIdx.x * per, b1 = min(nb, b0 + per);
  unsigned long long mine = 0;
  for (int b = b0; b < b1; ++b) mine += counts[b];
  unsigned long long incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned long long before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    before += w < wave ? wsum[w] : 0ull;
    total += wsum[w];
  }
  unsigned long long run = before + (incl - mine);
  for (int b = b0; b < b1; ++b) {
    offsets[b] = base + (int64_t)run;
    run += counts[b];
  }
  if (threadIdx.x == 0) {
    const int64_t end = base + (int64_t)total;
    cursor[0] = end;
    if (end > capacity) cursor[1] = end - capacity;  // (rows beyond the capacity are dropped by the emit pass: the caller retries)
  }
}

struct KeptOut {
  float* pts;          // [cap, 3] points / pose_scale
  float* dens;         // [cap]
  __half* feat;        // [cap, 64] clamp(sem, 0, 1)
  int64_t* vox;        // [cap, 3]
  int64_t* row;        // [cap] global lattice row (row0 + i), nullable
};
__global__ __launch_bounds__(kKeptBlock) void kept_emit_kernel(const float* __restrict__ dens, int64_t n, float thr, const float* __restrict__ pts,
                                                               const float* __restrict__ sem, float inv_scale, double voxel, double mx, double my,
                                                               double mz, int64_t row0, const int64_t* __restrict__ offsets, int64_t capacity,
                                                               KeptOut o) {
  __shared__ unsigned wsum[kKeptBlock / 64];
  __shared__ int src_of[kKeptBlock];  // local index of the k-th kept point of this workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * kKeptBlock + threadIdx.x;
  const bool keep = i < n && dens[i] > thr;
  const unsigned long long m = __ballot(keep);
  if (lane == 0) wsum[wave] = (unsigned)__popcll(m);
  __syncthreads();
  unsigned before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kKeptBlock / 64; ++w) {
    before += w < wave ? wsum[w] : 0u;
    total += wsum[w];
  }
  const int rank = (int)(before + (unsigned)__popcll(m & ((1ull << lane) - 1ull)));
  if (keep) src_of[rank] = threadIdx.x;
  __syncthreads();
  const int64_t out0 = offsets[blockIdx.x];
  // scalar columns: thread k < total takes the k-th kept point
  if ((unsigned)threadIdx.x < total && out0 + threadIdx.x < capacity) {
    const int64_t s = (int64_t)blockIdx.x * kKeptBlock + src_of[threadIdx.x], d = out0 + threadIdx.x;
    // torch divides a float tensor by a python scalar as a multiplication with the fp32 reciprocal (BinaryDivTrueKernel: cpu-scalar path)
    const float px = pts[s * 3] * inv_scale, py = pts[s * 3 + 1] * inv_scale, pz = pts[s * 3 + 2] * inv_scale;
    o.pts[d * 3] = px;
    o.pts[d * 3 + 1] = py;
    o.pts[d * 3 + 2] = pz;
    o.dens[d] = dens[s];
    const double h = voxel * 0.5;
    o.vox[d * 3] = (int64_t)floor(((double)px - (mx - h)) / voxel);
    o.vox[d * 3 + 1] = (int64_t)floor(((double)py - (my - h)) / voxel);
    o.vox[d * 3 + 2] = (int64_t)floor(((double)pz - (mz - h)) / voxel);
    if (o.row != nullptr) o.row[d] = row0 + s;
  }
  // feature rows: 16 lanes per row, 16 bytes in / 8 bytes out per lane (gather_clip_f16_c64_kernel)
  const int q = threadIdx.x & 15;
  for (unsigned k = threadIdx.x >> 4; k < total; k += kKeptBlock / 16) {
    const int64_t d = out0 + k;
    if (d >= capacity) break;
    const int64_t s = (int64_t)blockIdx.x * kKeptBlock + src_of[k];
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sem + s * 64 + 4 * q));
    const __half2 a = __floats2half2_rn(clampf_nan(v.x, 0.0f, 1.0f), clampf_nan(v.y, 0.0f, 1.0f));
    const __half2 b = __floats2half2_rn(clampf_nan(v.z, 0.0f, 1.0f), clampf_nan(v.w, 0.0f, 1.0f));
    uint2 w2;
    w2.x = *reinterpret_cast<const unsigned*>(&a);
    w2.y = *reinterpret_cast<const unsigned*>(&b);
    *reinterpret_cast<uint2*>(o.feat + d * 64 + 4 * q) = w2;
  }
}
}  // namespace

// bytes of scratch ps_emit_kept needs for a chunk of n points
extern "C" int64_t ps_emit_kept_workspace(int64_t n) {
  const int64_t nb = (n + kKeptBlock - 1) / kKeptBlock;
  return nb * 4 + 16 + nb * 8;
}

// Append the rows of this chunk whose density exceeds `threshold` to the tile-wide arrays (see above).  dens [n], pts [n,3], sem [n,64]
// fp32; cursor: int64[2] in device memory, zeroed by the caller before the FIRST chunk of a tile ({rows so far, rows that did not fit});
// capacity: rows the output arrays hold; min_bound: HOST array (the voxel grid's origin, metres); row0: lattice row of point 0.
extern "C" int ps_emit_kept(const float* dens, int64_t n, float threshold, const float* pts, const float* sem, int C, float pose_scale,
                            double voxel, const double* min_bound, int64_t row0, int64_t* cursor, int64_t capacity, void* workspace,
                            float* out_pts, float* out_dens, void* out_feat_f16, int64_t* out_vox, int64_t* out_row, void* stream) {
  PS_REQUIRE(C == 64, "ps_emit_kept: 64 feature channels");
  PS_REQUIRE(dens && pts && sem && min_bound && cursor && workspace && out_pts && out_dens && out_feat_f16 && out_vox, "ps_emit_kept: null argument");
  PS_REQUIRE(pose_scale > 0.f && voxel > 0.0 && capacity >= 0, "ps_emit_kept: pose_scale and voxel must be positive");
  if (n == 0) return 0;
  const int64_t nb = (n + kKeptBlock - 1) / kKeptBlock;
  PS_REQUIRE(nb < (int64_t(1) << 31), "ps_emit_kept: chunk too large");
  unsigned* counts = (unsigned*)workspace;
  int64_t* offsets = (int64_t*)((char*)workspace + ((nb * 4 + 15) & ~(int64_t)15));
  hipStream_t s = (hipStream_t)stream;
  kept_count_kernel<<<(unsigned)nb, kKeptBlock, 0, s>>>(dens, n, threshold, counts);
  kept_scan_kernel<<<1, 1024, 0, s>>>(counts, (int)nb, offsets, cursor, capacity);
  kept_emit_kernel<<<(unsigned)nb, kKeptBlock, 0, s>>>(dens, n, threshold, pts, sem, 1.0f / pose_scale, voxel, min_bound[0], min_bound[1], min_bound[2],
                                                       row0, offsets, capacity, KeptOut{out_pts, out_dens, (__half*)out_feat_f16, out_vox, out_row});
  PS_CHECK_LAUNCH();
}
