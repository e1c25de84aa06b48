// Per-(point, level) hash-grid cell maths shared by the operator-level and the fused field kernels.
// Follows ns/field_components/encodings.py:343-384 (torch fallback): corner ids
//   0=ccc 1=cfc 2=ffc 3=fcc 4=ccf 5=cff 6=fff 7=fcf   (x,y,z each ceil or floor)
// and the blend  f03=v0*ox+v3*(1-ox) ... out=f0312*oz+f4756*(1-oz).
#pragma once
#include "common.hpp"

namespace ps {

struct Cell {
  int cx, cy, cz, fx, fy, fz;
  float ox, oy, oz;
};

__device__ __forceinline__ Cell make_cell(float x, float y, float z, float scale) {
#pragma clang fp contract(off)
  Cell c;
  // rounded products (no fma contraction): the fractional offset must see the same fp32 `scaled` as the
  // reference, whose ulp at resolution ~2^11..2^14 is what dominates the interpolation error
  const float sx = x * scale, sy = y * scale, sz = z * scale;
  const float flx = floorf(sx), fly = floorf(sy), flz = floorf(sz);
  c.cx = (int)ceilf(sx);
  c.cy = (int)ceilf(sy);
  c.cz = (int)ceilf(sz);
  c.fx = (int)flx;
  c.fy = (int)fly;
  c.fz = (int)flz;
  c.ox = sx - flx;
  c.oy = sy - fly;
  c.oz = sz - flz;
  return c;
}

__device__ __forceinline__ void corner_hashes(const Cell& c, uint32_t mask, uint32_t (&h)[8]) {
  // partial products are shared between corners
  const uint32_t xc = (uint32_t)c.cx, xf = (uint32_t)c.fx;
  const uint32_t yc = (uint32_t)c.cy * 2654435761u, yf = (uint32_t)c.fy * 2654435761u;
  const uint32_t zc = (uint32_t)c.cz * 805459861u, zf = (uint32_t)c.fz * 805459861u;
  h[0] = (xc ^ yc ^ zc) & mask;
  h[1] = (xc ^ yf ^ zc) & mask;
  h[2] = (xf ^ yf ^ zc) & mask;
  h[3] = (xf ^ yc ^ zc) & mask;
  h[4] = (xc ^ yc ^ zf) & mask;
  h[5] = (xc ^ yf ^ zf) & mask;
  h[6] = (xf ^ yf ^ zf) & mask;
  h[7] = (xf ^ yc ^ zf) & mask;
}

template <int F>
struct Row;
template <>
struct Row<1> {
  float v[1];
  __device__ __forceinline__ void load(const float* p) { v[0] = *p; }
};
template <>
struct Row<2> {
  float v[2];
  __device__ __forceinline__ void load(const float* p) {
    f32x2 t = *reinterpret_cast<const f32x2*>(p);
    v[0] = t.x;
    v[1] = t.y;
  }
};
template <>
struct Row<4> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  }
};

// table_level points at row 0 of this level ([T, F] floats)
template <int F>
__device__ __forceinline__ void encode_level(const float* __restrict__ table_level, const Cell& c, uint32_t mask,
                                             float (&out)[F]) {
  uint32_t h[8];
  corner_hashes(c, mask, h);
  Row<F> r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k].load(table_level + (size_t)h[k] * F);
  const float ox = c.ox, oy = c.oy, oz = c.oz;
  const float ux = 1.0f - ox, uy = 1.0f - oy, uz = 1.0f - oz;
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const float f03 = r[0].v[f] * ox + r[3].v[f] * ux;
    const float f12 = r[1].v[f] * ox + r[2].v[f] * ux;
    const float f56 = r[5].v[f] * ox + r[6].v[f] * ux;
    const float f47 = r[4].v[f] * ox + r[7].v[f] * ux;
    const float f0312 = f03 * oy + f12 * uy;
    const float f4756 = f47 * oy + f56 * uy;
    out[f] = f0312 * oz + f4756 * uz;
  }
}

// dtable_level[h_k][f] += w_k * g[f]  for the 8 corners (fp32 hardware atomics)
template <int F>
__device__ __forceinline__ void scatter_level(float* __restrict__ dtable_level, const Cell& c, uint32_t mask,
                                              const float (&g)[F]) {
  uint32_t h[8];
  corner_hashes(c, mask, h);
  const float ox = c.ox, oy = c.oy, oz = c.oz;
  const float ux = 1.0f - ox, uy = 1.0f - oy, uz = 1.0f - oz;
  // weights in corner-id order (c -> o, f -> 1-o)
  const float w[8] = {ox * oy * oz, ox * uy * oz, ux * uy * oz, ux * oy * oz,
                      ox * oy * uz, ox * uy * uz, ux * uy * uz, ux * oy * uz};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (w[k] != 0.0f) {
      float* p = dtable_level + (size_t)h[k] * F;
#pragma unroll
      for (int f = 0; f < F; ++f) unsafeAtomicAdd(p + f, w[k] * g[f]);
    }
  }
}

}  // namespace ps
