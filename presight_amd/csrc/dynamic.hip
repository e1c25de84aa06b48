// Dynamic branch of the dual (static + dynamic) field — BASELINE.json configs[3] "EmerNeRF-style dual field (two hash grids +
// flow MLP)".  The reference has no such field; the model is defined by oracle/dual_oracle.py, with the
// conventions of the static stack (ns/fields/PreSight/ingp_field.py:168-267, ns/field_components/encodings.py:324-384):
//
//     x4 = (u, t)                                  u: normalised + contracted position, t: normalised ray timestamp
//     e0 = H4(x4)                                  4-D multiresolution hash grid, 16 corners per (point, level)
//     flow = flow_scale * MLP_flow(e0)  [6]        forward | backward scene flow in units of u
//     feat = (e0 + H4(u + flow_f, t + dt) + H4(u + flow_b, t - dt)) / 3          temporal aggregation
//     feat -> the same fused MLP stack as the static field (field.hip) -> (sigma_d, rgb_d, sem_d)
//     per sample: sigma = sigma_s + sigma_d, w_d = sigma_d / max(sigma, 1e-6), c = c_s + w_d (c_d - c_s)
//
// Kernels here: the 4-D encode (lane-paired like the 3-D one, optionally aggregating the two warped position sets onto e0), the
// gradient of the encode w.r.t. the warped positions (what trains the flow), the flow MLP on the fp32 matrix cores (forward:
// writes the warped positions directly; backward: d(e0) = d(feat)/3 + the flow path), and the density-weighted blend.  The
// table gradient uses the binned fixed-point scatter of encode.hip (ps_grid4_scatter_binned: same record streams, 8 x-pairs).
#include "common.hpp"
#include "encode_core.hpp"
#include "field_io.hpp"
#include "hashgrid_core.hpp"
#include "mlp_core.hpp"

namespace {

using namespace ps;

constexpr uint32_t kPrimeY = 2654435761u, kPrimeZ = 805459861u, kPrimeT = 3674653429u;

// ------------------------------------------------------------------------------------------ points
// x4[n] = (u[n], times[n / S])
__global__ void dyn_points_kernel(const float* __restrict__ u, const float* __restrict__ times, int S, int64_t N, float* __restrict__ x4) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float t = times[(uint32_t)n / (uint32_t)S];
  *reinterpret_cast<f32x4*>(x4 + n * 4) = (f32x4){u[n * 3], u[n * 3 + 1], u[n * 3 + 2], t};
}

// ------------------------------------------------------------------------------------------ 4-D cell
struct Cell4 {
  int cx, cy, cz, ct, fx, fy, fz, ft;
  float ox, oy, oz, ot;
};
__device__ __forceinline__ Cell4 make_cell4(f32x4 x, float scale) {
#pragma clang fp contract(off)
  Cell4 c;
  const float sx = x[0] * scale, sy = x[1] * scale, sz = x[2] * scale, st = x[3] * scale;
  const float flx = floorf(sx), fly = floorf(sy), flz = floorf(sz), flt = floorf(st);
  c.cx = (int)ceilf(sx); c.cy = (int)ceilf(sy); c.cz = (int)ceilf(sz); c.ct = (int)ceilf(st);
  c.fx = (int)flx; c.fy = (int)fly; c.fz = (int)flz; c.ft = (int)flt;
  c.ox = sx - flx; c.oy = sy - fly; c.oz = sz - flz; c.ot = st - flt;
  return c;
}

// H4 of ONE point by a lane pair (lanes 2i / 2i+1 = floor-x / ceil-x corners, see grid_encode_pair_kernel in encode.hip): each
// lane gathers its 8 (y,z,t) rows, the x-blend takes the partner's product through a DPP quad swap; both lanes return the value.
// Blend order = the reference's (x, y, z), then t:  out = out_ceil_t * ot + out_floor_t * (1 - ot).
template <int F>
__device__ __forceinline__ void encode4_pair(const float* __restrict__ tl, const Cell4& c, uint32_t mask, int side, float (&v)[F]) {
#pragma clang fp contract(off)
  const uint32_t xs = (uint32_t)(side ? c.cx : c.fx);
  const uint32_t yc = (uint32_t)c.cy * kPrimeY, yf = (uint32_t)c.fy * kPrimeY;
  const uint32_t zc = (uint32_t)c.cz * kPrimeZ, zf = (uint32_t)c.fz * kPrimeZ;
  const uint32_t tc = (uint32_t)c.ct * kPrimeT, tf = (uint32_t)c.ft * kPrimeT;
  const uint32_t hyz[4] = {xs ^ yc ^ zc, xs ^ yf ^ zc, xs ^ yc ^ zf, xs ^ yf ^ zf};  // (c,c) (f,c) | (c,f) (f,f)
  Row<F> r[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[k].load(tl + (size_t)((hyz[k] ^ tc) & mask) * F);
    r[4 + k].load(tl + (size_t)((hyz[k] ^ tf) & mask) * F);
  }
  const float wx = side ? c.ox : 1.0f - c.ox;
  const float oy = c.oy, oz = c.oz, ot = c.ot, uy = 1.0f - oy, uz = 1.0f - oz, ut = 1.0f - ot;
#pragma unroll
  for (int f = 0; f < F; ++f) {
    float o2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float p[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float mine = r[4 * h + k].v[f] * wx;
        const float other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine), 0xB1, 0xF, 0xF, false));
        p[k] = side ? mine + other : other + mine;  // = v_ceil*ox + v_floor*(1-ox) in both lanes
      }
      const float f0312 = p[0] * oy + p[1] * uy;
      const float f4756 = p[2] * oy + p[3] * uy;
      o2[h] = f0312 * oz + f4756 * uz;
    }
    v[f] = o2[0] * ot + o2[1] * ut;
  }
}

template <int F>
__device__ __forceinline__ void store_row(float* o, const float (&v)[F]) {
  if constexpr (F == 1) o[0] = v[0];
  if constexpr (F == 2) *reinterpret_cast<f32x2*>(o) = (f32x2){v[0], v[1]};
  if constexpr (F == 4) *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
}
template <int F>
__device__ __forceinline__ void load_row(const float* o, float (&v)[F]) {
  Row<F> r;
  r.load(o);
#pragma unroll
  for (int f = 0; f < F; ++f) v[f] = r.v[f];
}

// records the binned table backward will emit for one position (encode.hip bin_kernel, D = 4): one per (y,z,t) combination in
// the slice of the floor-x corner (counted by the floor lane), one more in the slice of the ceil-x corner when the pair straddles a
// slice boundary (counted by the ceil lane); an upper bound -- the backward drops records whose gradient is exactly zero
__device__ __forceinline__ void count_records4(const Cell4& c, uint32_t mask, int log2_slice, int side, unsigned* __restrict__ cnt) {
  const bool together = ((((uint32_t)c.cx ^ (uint32_t)c.fx) & mask) >> log2_slice) == 0u;
  if (side != 0 && together) return;
  const uint32_t xs = (uint32_t)(side ? c.cx : c.fx);
  const uint32_t yc = (uint32_t)c.cy * kPrimeY, yf = (uint32_t)c.fy * kPrimeY;
  const uint32_t zc = (uint32_t)c.cz * kPrimeZ, zf = (uint32_t)c.fz * kPrimeZ;
  const uint32_t tc = (uint32_t)c.ct * kPrimeT, tf = (uint32_t)c.ft * kPrimeT;
  const uint32_t hyz[4] = {xs ^ yc ^ zc, xs ^ yf ^ zc, xs ^ yc ^ zf, xs ^ yf ^ zf};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    atomicAdd(&cnt[((hyz[k] ^ tc) & mask) >> log2_slice], 1u);
    atomicAdd(&cnt[((hyz[k] ^ tf) & mask) >> log2_slice], 1u);
  }
}

constexpr int kEnc4MaxSlices = 256;

// feat[l][n] = H4(x[n])                                              (e0 == null, x [N,4])
// feat[l][n] = (e0[l][n] + H4(x[n]) + H4(x[N + n])) / 3              (e0 != null, x [2N,4]: forward- and backward-warped sets)
// COUNT: slice_counts[level][slice] += the records of the table backward for the positions this launch encodes
template <int F, bool COUNT>
__global__ __launch_bounds__(256) void grid4_encode_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                           const float* __restrict__ scalings, int L, int log2T, int64_t N,
                                                           int64_t plane_stride, const float* __restrict__ e0, float* __restrict__ feat,
                                                           int group, int parts, unsigned* __restrict__ slice_counts, int log2_slice) {
#pragma clang fp contract(off)
  __shared__ unsigned cnt[COUNT ? kEnc4MaxSlices : 1];
  const int n_slices = COUNT ? (1 << (log2T - log2_slice)) : 0;
  if constexpr (COUNT) {
    for (int i = threadIdx.x; i < n_slices; i += 256) cnt[i] = 0u;
    __syncthreads();  // (before the early return below: every thread of a workgroup takes the same branch there)
  }
  const int64_t chunks = (N + 127) / 128;  // 128 points per pass of a 256-thread workgroup
  const int64_t groups = (chunks + group - 1) / group;
  int level;
  int64_t my_group;
  bool valid;
  enc_item(groups, L, parts, level, my_group, valid);
  if (!valid) return;
  const uint32_t mask = (1u << log2T) - 1u;
  const float scale = scalings[level];
  const float* tl = table + ((int64_t)level << log2T) * F;
  const int side = threadIdx.x & 1;
  for (int64_t chunk = my_group * group, end = min(chunks, chunk + group); chunk < end; ++chunk) {
    const int64_t n = chunk * 128 + (threadIdx.x >> 1);
    const bool ok = n < N;  // both lanes of a pair agree; inactive pairs still take part in the DPP swap
    const int64_t nn = ok ? n : N - 1;
    float v[F];
    const Cell4 ca = make_cell4(*reinterpret_cast<const f32x4*>(x + nn * 4), scale);
    encode4_pair<F>(tl, ca, mask, side, v);
    if constexpr (COUNT)
      if (ok) count_records4(ca, mask, log2_slice, side, cnt);
    if (e0 != nullptr) {
      float vb[F], v0[F];
      const Cell4 cb = make_cell4(*reinterpret_cast<const f32x4*>(x + (N + nn) * 4), scale);
      encode4_pair<F>(tl, cb, mask, side, vb);
      if constexpr (COUNT)
        if (ok) count_records4(cb, mask, log2_slice, side, cnt);
      load_row<F>(e0 + level * plane_stride + nn * F, v0);
#pragma unroll
      for (int f = 0; f < F; ++f) v[f] = ((v0[f] + v[f]) + vb[f]) / 3.0f;
    }
    if (ok && side == 0) store_row<F>(feat + level * plane_stride + n * F, v);
  }
  if constexpr (COUNT) {
    __syncthreads();
    for (int i = threadIdx.x; i < n_slices; i += 256)
      if (cnt[i]) atomicAdd(&slice_counts[level * n_slices + i], cnt[i]);
  }
}

// ------------------------------------------------------------------------------------------ d(encode) / d(position)
// H4 is multilinear inside a cell: d out[l][f] / d x_a = scalings[l] * (blend over the other three axes of (v_ceil_a - v_floor_a)).
// dx[m][a] = g_scale * sum_l scalings[l] * sum_f dfeat[l][m mod period][f] * dH4[l][f]/dx_a,  a in {x, y, z}  (the timestamp has
// no learnable input).  One thread per (point, axis-free) walks the levels; the 16 corner rows are gathered once per level.
template <int F>
__global__ __launch_bounds__(256) void grid4_input_grad_kernel(const float* __restrict__ x, const float* __restrict__ dfeat,
                                                               const float* __restrict__ table, const float* __restrict__ scalings,
                                                               int L, int log2T, int64_t M, int64_t period, int64_t plane_stride,
                                                               float g_scale, float* __restrict__ dx) {
  const int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int64_t n = (period > 0 && m >= period) ? m - period : m;
  const uint32_t mask = (1u << log2T) - 1u;
  const f32x4 xv = *reinterpret_cast<const f32x4*>(x + m * 4);
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int l = 0; l < L; ++l) {
    const float scale = scalings[l];
    const Cell4 c = make_cell4(xv, scale);
    const float* tl = table + ((int64_t)l << log2T) * F;
    float g[F];
    load_row<F>(dfeat + l * plane_stride + n * F, g);
    const uint32_t hx[2] = {(uint32_t)c.fx, (uint32_t)c.cx};  // index 1 = ceil
    const uint32_t hy[2] = {(uint32_t)c.fy * kPrimeY, (uint32_t)c.cy * kPrimeY};
    const uint32_t hz[2] = {(uint32_t)c.fz * kPrimeZ, (uint32_t)c.cz * kPrimeZ};
    const uint32_t ht[2] = {(uint32_t)c.ft * kPrimeT, (uint32_t)c.ct * kPrimeT};
    const float wx[2] = {1.0f - c.ox, c.ox}, wy[2] = {1.0f - c.oy, c.oy}, wz[2] = {1.0f - c.oz, c.oz}, wt[2] = {1.0f - c.ot, c.ot};
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ix = k & 1, iy = (k >> 1) & 1, iz = (k >> 2) & 1, it = (k >> 3) & 1;
      Row<F> r;
      r.load(tl + (size_t)((hx[ix] ^ hy[iy] ^ hz[iz] ^ ht[it]) & mask) * F);
      float dot = 0.f;
#pragma unroll
      for (int f = 0; f < F; ++f) dot = fmaf(r.v[f], g[f], dot);
      const float sgx = ix ? 1.0f : -1.0f, sgy = iy ? 1.0f : -1.0f, sgz = iz ? 1.0f : -1.0f;
      sx = fmaf(dot * sgx, wy[iy] * wz[iz] * wt[it], sx);
      sy = fmaf(dot * sgy, wx[ix] * wz[iz] * wt[it], sy);
      sz = fmaf(dot * sgz, wx[ix] * wy[iy] * wt[it], sz);
    }
    // an exactly integer coordinate has ceil == floor: both corners are the same row and the derivative is 0, as in autograd
    ax = fmaf(scale, sx, ax);
    ay = fmaf(scale, sy, ay);
    az = fmaf(scale, sz, az);
  }
  dx[m * 3] = g_scale * ax;
  dx[m * 3 + 1] = g_scale * ay;
  dx[m * 3 + 2] = g_scale * az;
}

// The same gradient, LEVEL-PARALLEL (round 4): a workgroup takes 2048 positions of ONE level, dealt like the encode (enc_item: every
// XCD's L2 holds one level's table at a time; the one-thread-walks-all-levels kernel above has all L tables in flight at once and
// thrashes the L2s: 4.8 ms at cfg 4), and writes the level's unscaled partial sums part[l][m] = (sx, sy, sz); a second pass adds the
// levels in order with the same fmaf chain as above, so the result is bit-identical.
template <int F>
__global__ __launch_bounds__(256) void grid4_input_grad_level_kernel(const float* __restrict__ x, const float* __restrict__ dfeat,
                                                                     const float* __restrict__ table, const float* __restrict__ scalings,
                                                                     int L, int log2T, int64_t M, int64_t period, int64_t plane_stride,
                                                                     int group, int parts, float* __restrict__ part) {
  const int64_t chunks = (M + 255) / 256;
  const int64_t groups = (chunks + group - 1) / group;
  int level;
  int64_t my_group;
  bool valid;
  enc_item(groups, L, parts, level, my_group, valid);
  if (!valid) return;
  const uint32_t mask = (1u << log2T) - 1u;
  const float scale = scalings[level];
  const float* tl = table + ((int64_t)level << log2T) * F;
  for (int64_t chunk = my_group * group, end = min(chunks, chunk + group); chunk < end; ++chunk) {
    const int64_t m = chunk * 256 + threadIdx.x;
    if (m >= M) continue;
    const int64_t n = (period > 0 && m >= period) ? m - period : m;
    const Cell4 c = make_cell4(*reinterpret_cast<const f32x4*>(x + m * 4), scale);
    float g[F];
    load_row<F>(dfeat + level * plane_stride + n * F, g);
    const uint32_t hx[2] = {(uint32_t)c.fx, (uint32_t)c.cx};  // index 1 = ceil
    const uint32_t hy[2] = {(uint32_t)c.fy * kPrimeY, (uint32_t)c.cy * kPrimeY};
    const uint32_t hz[2] = {(uint32_t)c.fz * kPrimeZ, (uint32_t)c.cz * kPrimeZ};
    const uint32_t ht[2] = {(uint32_t)c.ft * kPrimeT, (uint32_t)c.ct * kPrimeT};
    const float wx[2] = {1.0f - c.ox, c.ox}, wy[2] = {1.0f - c.oy, c.oy}, wz[2] = {1.0f - c.oz, c.oz}, wt[2] = {1.0f - c.ot, c.ot};
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ix = k & 1, iy = (k >> 1) & 1, iz = (k >> 2) & 1, it = (k >> 3) & 1;
      Row<F> r;
      r.load(tl + (size_t)((hx[ix] ^ hy[iy] ^ hz[iz] ^ ht[it]) & mask) * F);
      float dot = 0.f;
#pragma unroll
      for (int f = 0; f < F; ++f) dot = fmaf(r.v[f], g[f], dot);
      const float sgx = ix ? 1.0f : -1.0f, sgy = iy ? 1.0f : -1.0f, sgz = iz ? 1.0f : -1.0f;
      sx = fmaf(dot * sgx, wy[iy] * wz[iz] * wt[it], sx);
      sy = fmaf(dot * sgy, wx[ix] * wz[iz] * wt[it], sy);
      sz = fmaf(dot * sgz, wx[ix] * wy[iy] * wt[it], sz);
    }
    float* o = part + ((int64_t)level * M + m) * 3;
    o[0] = sx;
    o[1] = sy;
    o[2] = sz;
  }
}

__global__ __launch_bounds__(256) void grid4_input_grad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ scalings,
                                                                      int L, int64_t M, float g_scale, float* __restrict__ dx) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // (position, axis)
  if (i >= M * 3) return;
  float a = 0.f;
  for (int l = 0; l < L; ++l) a = fmaf(scalings[l], part[(int64_t)l * M * 3 + i], a);
  dx[i] = g_scale * a;
}

// ------------------------------------------------------------------------------------------ flow MLP
// flow = flow_scale * MLP(e0) with MLP = Linear(LF, H) ReLU Linear(H, H) ReLU Linear(H, 6); outputs 0..2 = forward flow,
// 3..5 = backward flow.  In the MFMA D layout lane (point j, group g) holds outputs 4g..4g+3 of its point: group 0 owns
// (f_x, f_y, f_z, b_x), group 1 owns (b_y, b_z, -, -), so the warped positions are written without any cross-lane traffic:
//     xw[n]     = (u + flow_f, t + dt)          xw[N + n] = (u + flow_b, t - dt)
template <class M, int PB>
__global__ __launch_bounds__(256) void flow_fwd_kernel(const float* __restrict__ e0, int64_t plane_stride, int LF, int F,
                                                       const float* __restrict__ packed, const float* __restrict__ x4, int64_t N,
                                                       float flow_scale, float dt, float* __restrict__ xw) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) float lds[M::FW];
  for (int i = threadIdx.x * 4; i < M::FW; i += 256 * 4) *reinterpret_cast<f32x4*>(lds + i) = *reinterpret_cast<const f32x4*>(packed + i);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = ps_lane(), j = lane & 15, g = lane >> 4;
  FeatCols<M::KS0> fc;
  fc.init(plane_stride, LF, F);
  const int64_t tiles = (N + 16 * PB - 1) / (16 * PB);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t first = tile * 16 * PB;
    float xin[PB][M::KS0], h1[PB][M::HB * 4], h2[PB][M::HB * 4], z[PB][M::NBO * 4];
    load_feat<M::KS0, PB>(e0, fc, F, first, N, xin);
    mlp_forward<M, PB>(LdsW{lds}, xin, h1, h2, z);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t p = first + pb * 16 + j;
      if (p < N && g < 2) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x4 + p * 4);
        const float f0 = flow_scale * z[pb][0], f1 = flow_scale * z[pb][1], f2 = flow_scale * z[pb][2], f3 = flow_scale * z[pb][3];
        if (g == 0) {
          *reinterpret_cast<f32x4*>(xw + p * 4) = (f32x4){xv[0] + f0, xv[1] + f1, xv[2] + f2, xv[3] + dt};
          xw[(N + p) * 4] = xv[0] + f3;
        } else {
          xw[(N + p) * 4 + 1] = xv[1] + f0;
          xw[(N + p) * 4 + 2] = xv[2] + f1;
          xw[(N + p) * 4 + 3] = xv[3] - dt;
        }
      }
    }
  }
}

// backward: d(flow) = flow_scale * d(warped position) (dxw [2N,3] from grid4_input_grad_kernel); writes
// de0x3 = dagg + 3 * d(e0 through the flow MLP) = THREE TIMES d(e0) (all three position sets then share the factor 1/3 of the
// aggregation, which the table scatter applies once: ps_grid4_scatter_binned out_scale) and one partial weight-gradient block
// per workgroup.
template <class M, int PB>
__global__ __launch_bounds__(256) void flow_bwd_kernel(const float* __restrict__ e0, int64_t plane_stride, int LF, int F,
                                                       const float* __restrict__ packed, const float* __restrict__ dxw,
                                                       const float* __restrict__ dagg, int64_t N, float flow_scale,
                                                       float* __restrict__ de0, float* __restrict__ gpart) {
#pragma clang fp contract(off)
  constexpr int SCR = M::SCRATCH_ROWS * kScratchLd;
  __shared__ __attribute__((aligned(16))) float lds[M::GPACKED + 4 * SCR + 16];
  float* gacc = lds;
  int* locks = reinterpret_cast<int*>(lds + M::GPACKED + 4 * SCR);
  for (int i = threadIdx.x; i < M::GPACKED; i += 256) gacc[i] = 0.0f;
  if (threadIdx.x < 16) locks[threadIdx.x] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = ps_lane(), j = lane & 15, g = lane >> 4;
  float* scratch = lds + M::GPACKED + wave * SCR;
  const GlobalW gw = make_global_w(packed, M::PACKED);
  FeatCols<M::KS0> fc;
  fc.init(plane_stride, LF, F);
  const int64_t tiles = (N + 16 * PB - 1) / (16 * PB);
  // workgroup-uniform trip count (the dW flush takes LDS locks); out-of-range tiles are fully masked
  for (int64_t base = (int64_t)blockIdx.x * 4; base < tiles; base += (int64_t)gridDim.x * 4) {
    const int64_t first = (base + wave) * 16 * PB;
    float xin[PB][M::KS0], h1[PB][M::HB * 4], h2[PB][M::HB * 4], z[PB][M::NBO * 4];
    load_feat<M::KS0, PB>(e0, fc, F, first, N, xin);
    mlp_forward<M, PB>(gw, xin, h1, h2, z);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t p = first + pb * 16 + j;
#pragma unroll
      for (int t = 0; t < M::NBO * 4; ++t) z[pb][t] = 0.0f;
      if (p < N) {
        if (g == 0) {
          z[pb][0] = flow_scale * dxw[p * 3];
          z[pb][1] = flow_scale * dxw[p * 3 + 1];
          z[pb][2] = flow_scale * dxw[p * 3 + 2];
          z[pb][3] = flow_scale * dxw[(N + p) * 3];
        } else if (g == 1) {
          z[pb][0] = flow_scale * dxw[(N + p) * 3 + 1];
          z[pb][1] = flow_scale * dxw[(N + p) * 3 + 2];
        }
      }
    }
    float dxin[PB][M::L0::IB * 4];
    mlp_backward<M, PB, true>(gw, scratch, gacc, locks, xin, h1, h2, z, dxin);
    float da[PB][M::KS0];
    load_feat<M::KS0, PB>(dagg, fc, F, first, N, da);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int t = 0; t < M::KS0; ++t) dxin[pb][t] = da[pb][t] + 3.0f * dxin[pb][t];
    store_dfeat<M::KS0, PB>(de0, fc, F, first, N, dxin);
  }
  __syncthreads();
  float* out = gpart + (size_t)blockIdx.x * M::GPACKED;
  for (int i = threadIdx.x; i < M::GPACKED; i += 256) out[i] = gacc[i];
}

// (L*F, hidden) of the flow MLP
#define PS_FLOW_CFGS(X) \
  X(32, 64)             \
  X(4, 32)

int flow_bwd_grid(int64_t N, int pb) {
  const int64_t tiles = (N + 16 * pb - 1) / (16 * pb);
  int64_t g = (tiles + 3) / 4;
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------------ blend
// 16 lanes per sample: lane q of a sample handles semantic channels 4q..4q+3, lane 0 also the density, lanes 0..2 colour k.
// forward:  sigma = ss + sd;  wd = sd / max(sigma, eps);  c = cs + wd (cd - cs)
constexpr float kBlendEps = 1e-6f;

__global__ __launch_bounds__(256) void blend_fwd_kernel(const float* __restrict__ ss, const float* __restrict__ rs,
                                                        const float* __restrict__ ms, const float* __restrict__ sd,
                                                        const float* __restrict__ rd, const float* __restrict__ md, int64_t N, int C,
                                                        float* __restrict__ sigma, float* __restrict__ rgb, float* __restrict__ sem) {
#pragma clang fp contract(off)
  const int64_t n = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int q = threadIdx.x & 15;
  if (n >= N) return;
  const float a = ss[n], b = sd[n];
  const float s = a + b;
  const float wd = b / fmaxf(s, kBlendEps);
  if (q == 0) sigma[n] = s;
  if (q < 3) {
    const float cs = rs[n * 3 + q], cd = rd[n * 3 + q];
    rgb[n * 3 + q] = cs + wd * (cd - cs);
  }
  for (int c = 4 * q; c < C; c += 64) {
    const f32x4 vs = *reinterpret_cast<const f32x4*>(ms + n * C + c), vd = *reinterpret_cast<const f32x4*>(md + n * C + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = vs[k] + wd * (vd[k] - vs[k]);
    *reinterpret_cast<f32x4*>(sem + n * C + c) = o;
  }
}

// backward:  d cs = dc (1 - wd),  d cd = dc wd,  d wd = sum_c dc (cd - cs);
//            sigma > eps: d wd / d ss = -sd / sigma^2, d wd / d sd = ss / sigma^2;  else (clamped): 0 and 1 / eps.
//            d ss = d sigma + d wd * (d wd / d ss),   d sd = d sigma + d wd * (d wd / d sd) + extra_dd
//            (extra_dd: nullable per-sample gradient that reaches the dynamic density directly, e.g. its sparsity regulariser)
__global__ __launch_bounds__(256) void blend_bwd_kernel(const float* __restrict__ ss, const float* __restrict__ rs,
                                                        const float* __restrict__ ms, const float* __restrict__ sd,
                                                        const float* __restrict__ rd, const float* __restrict__ md,
                                                        const float* __restrict__ dsigma, const float* __restrict__ drgb,
                                                        const float* __restrict__ dsem, int64_t N, int C, const float* __restrict__ extra_dd,
                                                        float* __restrict__ dss, float* __restrict__ drs, float* __restrict__ dms,
                                                        float* __restrict__ dsd, float* __restrict__ drd, float* __restrict__ dmd) {
#pragma clang fp contract(off)
  const int64_t n0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int q = threadIdx.x & 15;
  const bool ok = n0 < N;
  const int64_t n = ok ? n0 : N - 1;  // every lane takes part in the row reduction
  const float a = ss[n], b = sd[n];
  const float s = a + b;
  const bool clamped = !(s >= kBlendEps);
  const float den = clamped ? kBlendEps : s;
  const float wd = b / den;
  float dot = 0.f;
  if (q < 3 && drgb != nullptr) {
    const float cs = rs[n * 3 + q], cd = rd[n * 3 + q], d = drgb[n * 3 + q];
    dot += d * (cd - cs);
    if (ok) {
      drs[n * 3 + q] = d * (1.0f - wd);
      drd[n * 3 + q] = d * wd;
    }
  } else if (q < 3 && ok) {
    drs[n * 3 + q] = 0.f;
    drd[n * 3 + q] = 0.f;
  }
  for (int c = 4 * q; c < C; c += 64) {
    f32x4 o1 = (f32x4){0.f, 0.f, 0.f, 0.f}, o2 = o1;
    if (dsem != nullptr) {
      const f32x4 vs = *reinterpret_cast<const f32x4*>(ms + n * C + c), vd = *reinterpret_cast<const f32x4*>(md + n * C + c);
      const f32x4 d = *reinterpret_cast<const f32x4*>(dsem + n * C + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dot += d[k] * (vd[k] - vs[k]);
        o1[k] = d[k] * (1.0f - wd);
        o2[k] = d[k] * wd;
      }
    }
    if (ok) {
      *reinterpret_cast<f32x4*>(dms + n * C + c) = o1;
      *reinterpret_cast<f32x4*>(dmd + n * C + c) = o2;
    }
  }
  dot = ps_row16_sum(dot);
  if (ok && q == 0) {
    const float ds = dsigma != nullptr ? dsigma[n] : 0.f;
    const float inv2 = 1.0f / (den * den);
    const float dw_ds = clamped ? 0.0f : -b * inv2;
    const float dw_dd = clamped ? 1.0f / kBlendEps : a * inv2;
    dss[n] = ds + dot * dw_ds;
    dsd[n] = ds + dot * dw_dd + (extra_dd != nullptr ? extra_dd[n] : 0.0f);
  }
}

}  // namespace

extern "C" int ps_dyn_points(const float* u, const float* times, int S, int64_t N, float* x4, void* stream) {
  if (N == 0) return 0;
  PS_REQUIRE(u && times && x4 && S > 0, "ps_dyn_points: null argument");
  PS_REQUIRE(N < ((int64_t)1 << 32), "ps_dyn_points: at most 2^32 - 1 points");
  dyn_points_kernel<<<(unsigned)((N + 255) / 256), 256, 0, (hipStream_t)stream>>>(u, times, S, N, x4);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_grid_scatter_slices(int F, int log2T);  // encode.hip: table slices per level of the binned backward

extern "C" int ps_grid4_encode(const float* x, const float* table, const float* scalings, int L, int F, int log2T, int64_t N,
                               int64_t plane_stride, const float* e0, float* feat, uint32_t* slice_counts, void* stream) {
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_grid4_encode: features_per_level must be 1, 2 or 4");
  if (N == 0) return 0;
  PS_REQUIRE(((uintptr_t)x & 15) == 0, "ps_grid4_encode: positions must be 16-byte aligned [.,4] rows");
  const int group = 16;  // 2048 points of one level per workgroup
  const int64_t chunks = (N + 127) / 128, groups = (chunks + group - 1) / group;
  const int P = enc_parts(L);
  const dim3 grid((unsigned)(8 * (int64_t)(L * P / 8) * ((groups + P - 1) / P)));  // enc_item: L * P is a multiple of 16
  hipStream_t s = (hipStream_t)stream;
  const int n_slices = ps_grid_scatter_slices(F, log2T);
  PS_REQUIRE(slice_counts == nullptr || n_slices <= kEnc4MaxSlices, "ps_grid4_encode: too many table slices");
  int ls = 0;
  while ((n_slices << ls) < (1 << log2T)) ++ls;
#define X(FF)                                                                                                                              \
  if (F == FF) {                                                                                                                           \
    if (slice_counts != nullptr)                                                                                                           \
      grid4_encode_kernel<FF, true><<<grid, 256, 0, s>>>(x, table, scalings, L, log2T, N, plane_stride, e0, feat, group, P, slice_counts, ls);  \
    else                                                                                                                                   \
      grid4_encode_kernel<FF, false><<<grid, 256, 0, s>>>(x, table, scalings, L, log2T, N, plane_stride, e0, feat, group, P, nullptr, ls);      \
  }
  X(1) X(2) X(4)
#undef X
  PS_CHECK_LAUNCH();
}

extern "C" int ps_grid4_input_grad(const float* x, const float* dfeat, const float* table, const float* scalings, int L, int F,
                                   int log2T, int64_t M, int64_t period, int64_t plane_stride, float g_scale, float* dx, void* stream) {
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_grid4_input_grad: features_per_level must be 1, 2 or 4");
  PS_REQUIRE(period == 0 || M <= 2 * period, "ps_grid4_input_grad: at most two position sets per gradient plane");
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((M + 255) / 256);
#define X(FF) \
  if (F == FF) grid4_input_grad_kernel<FF><<<grid, 256, 0, s>>>(x, dfeat, table, scalings, L, log2T, M, period, plane_stride, g_scale, dx);
  X(1) X(2) X(4)
#undef X
  PS_CHECK_LAUNCH();
}

// level-parallel version: workspace [L, M, 3] floats (ps_grid4_input_grad_workspace bytes); bit-identical to ps_grid4_input_grad
extern "C" int64_t ps_grid4_input_grad_workspace(int L, int64_t M) { return (int64_t)L * M * 3 * 4; }

extern "C" int ps_grid4_input_grad_levels(const float* x, const float* dfeat, const float* table, const float* scalings, int L, int F,
                                          int log2T, int64_t M, int64_t period, int64_t plane_stride, float g_scale, float* dx,
                                          float* workspace, void* stream) {
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_grid4_input_grad_levels: features_per_level must be 1, 2 or 4");
  PS_REQUIRE(period == 0 || M <= 2 * period, "ps_grid4_input_grad_levels: at most two position sets per gradient plane");
  PS_REQUIRE(workspace != nullptr, "ps_grid4_input_grad_levels: workspace required");
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int group = 8;  // 2048 positions of one level per workgroup
  const int64_t chunks = (M + 255) / 256, groups = (chunks + group - 1) / group;
  const int P = enc_parts(L);
  const dim3 grid((unsigned)(8 * (int64_t)(L * P / 8) * ((groups + P - 1) / P)));
#define X(FF) \
  if (F == FF) grid4_input_grad_level_kernel<FF><<<grid, 256, 0, s>>>(x, dfeat, table, scalings, L, log2T, M, period, plane_stride, group, P, workspace);
  X(1) X(2) X(4)
#undef X
  grid4_input_grad_reduce_kernel<<<(unsigned)((M * 3 + 255) / 256), 256, 0, s>>>(workspace, scalings, L, M, g_scale, dx);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_flow_sizes(int LF, int hidden, int64_t N, int64_t* packed_floats, int64_t* grad_floats, int* n_parts) {
#define X(lf, h)                                    \
  if (LF == lf && hidden == h) {                    \
    using M = ps::MlpT<(lf + 3) / 4, h / 16, 1, 3>; \
    *packed_floats = M::PACKED;                     \
    *grad_floats = M::GPACKED;                      \
    *n_parts = flow_bwd_grid(N, 2);                 \
    return 0;                                       \
  }
  PS_FLOW_CFGS(X)
#undef X
  ps_set_error("ps_flow: unsupported (L*F, hidden)");
  return -2;
}

extern "C" int ps_flow_fwd(const float* e0, int64_t plane_stride, int LF, int F, int hidden, const float* packed, const float* x4,
                           int64_t N, float flow_scale, float dt, float* xw, void* stream) {
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
#define X(lf, h)                                                                                              \
  if (LF == lf && hidden == h) {                                                                              \
    using M = ps::MlpT<(lf + 3) / 4, h / 16, 1, 3>;                                                           \
    constexpr int PB = 4;                                                                                     \
    const int64_t tiles = (N + 16 * PB - 1) / (16 * PB);                                                      \
    int grid = (int)((tiles + 3) / 4);                                                                        \
    if (grid > 512) grid = 512;                                                                               \
    flow_fwd_kernel<M, PB><<<grid, 256, 0, s>>>(e0, plane_stride, LF, F, packed, x4, N, flow_scale, dt, xw);  \
    PS_CHECK_LAUNCH();                                                                                        \
  }
  PS_FLOW_CFGS(X)
#undef X
  ps_set_error("ps_flow_fwd: unsupported (L*F, hidden)");
  return -2;
}

extern "C" int ps_flow_bwd(const float* e0, int64_t plane_stride, int LF, int F, int hidden, const float* packed, const float* dxw,
                           const float* dagg, int64_t N, float flow_scale, float* de0, float* gpart, void* stream) {
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
#define X(lf, h)                                                                                                         \
  if (LF == lf && hidden == h) {                                                                                         \
    using M = ps::MlpT<(lf + 3) / 4, h / 16, 1, 3>;                                                                      \
    flow_bwd_kernel<M, 2><<<flow_bwd_grid(N, 2), 256, 0, s>>>(e0, plane_stride, LF, F, packed, dxw, dagg, N, flow_scale, de0, gpart); \
    PS_CHECK_LAUNCH();                                                                                                   \
  }
  PS_FLOW_CFGS(X)
#undef X
  ps_set_error("ps_flow_bwd: unsupported (L*F, hidden)");
  return -2;
}

extern "C" int ps_blend_fwd(const float* sigma_s, const float* rgb_s, const float* sem_s, const float* sigma_d, const float* rgb_d,
                            const float* sem_d, int64_t N, int C, float* sigma, float* rgb, float* sem, void* stream) {
  if (N == 0) return 0;
  PS_REQUIRE(C % 4 == 0, "ps_blend_fwd: semantic width must be a multiple of 4");
  blend_fwd_kernel<<<(unsigned)((N * 16 + 255) / 256), 256, 0, (hipStream_t)stream>>>(sigma_s, rgb_s, sem_s, sigma_d, rgb_d, sem_d, N, C,
                                                                                      sigma, rgb, sem);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_blend_bwd(const float* sigma_s, const float* rgb_s, const float* sem_s, const float* sigma_d, const float* rgb_d,
                            const float* sem_d, const float* dsigma, const float* drgb, const float* dsem, int64_t N, int C,
                            const float* extra_dsigma_d, float* dsigma_s, float* drgb_s, float* dsem_s, float* dsigma_d, float* drgb_d,
                            float* dsem_d, void* stream) {
  if (N == 0) return 0;
  PS_REQUIRE(C % 4 == 0, "ps_blend_bwd: semantic width must be a multiple of 4");
  blend_bwd_kernel<<<(unsigned)((N * 16 + 255) / 256), 256, 0, (hipStream_t)stream>>>(sigma_s, rgb_s, sem_s, sigma_d, rgb_d, sem_d, dsigma,
                                                                                      drgb, dsem, N, C, extra_dsigma_d, dsigma_s, drgb_s,
                                                                                      dsem_s, dsigma_d, drgb_d, dsem_d);
  PS_CHECK_LAUNCH();
}
