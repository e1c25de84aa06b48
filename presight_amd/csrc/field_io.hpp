// Feature-plane I/O of the fused MLP kernels (field.hip, dynamic.hip): the hash-grid features and their gradients live in
// HBM as level planes feat[l][n][f]; a lane (point j, lane group g) of the MFMA layout supplies input column 4t+g at k-step t.
#pragma once
#include "common.hpp"

namespace ps {

// features are stored as level planes feat[l][n][f]; the first layer uses the LINEAR column map
// (k-step t, lane group g -> column 4t+g)
template <int KS0, int PB>
__device__ __forceinline__ void load_feat(const float* __restrict__ feat, int64_t plane_stride, int LF, int F, int64_t first,
                                          int64_t N, float (&x)[PB][KS0]) {
  const int lane = ps_lane(), j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int64_t p = first + pb * 16 + j;
#pragma unroll
    for (int t = 0; t < KS0; ++t) {
      const int col = 4 * t + g;
      const int level = col / F, f = col - level * F;
      x[pb][t] = (p < N && col < LF) ? feat[level * plane_stride + p * F + f] : 0.0f;
    }
  }
}

template <int KS0, int PB>
__device__ __forceinline__ void store_dfeat(float* __restrict__ dfeat, int64_t plane_stride, int LF, int F, int64_t first,
                                            int64_t N, const float (&dx)[PB][((KS0 + 3) / 4) * 4]) {
  const int lane = ps_lane(), j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int64_t p = first + pb * 16 + j;
#pragma unroll
    for (int t = 0; t < KS0; ++t) {
      const int col = 4 * t + g;
      const int level = col / F, f = col - level * F;
      if (p < N && col < LF) dfeat[level * plane_stride + p * F + f] = dx[pb][t];
    }
  }
}

// the same accesses with the per-lane column offsets hoisted out of the tile loop (col / F is a runtime division: 30 vector
// instructions per element when it sits next to the load)
template <int KS0>
struct FeatCols {
  int64_t off[KS0];  // level * plane_stride + f of column 4t + g, -1: no such column
  __device__ __forceinline__ void init(int64_t plane_stride, int LF, int F) {
    const int g = ps_lane() >> 4;
#pragma unroll
    for (int t = 0; t < KS0; ++t) {
      const int col = 4 * t + g;
      const int level = col / F, f = col - level * F;
      off[t] = col < LF ? level * plane_stride + f : (int64_t)-1;
    }
  }
};
template <int KS0, int PB>
__device__ __forceinline__ void load_feat(const float* __restrict__ feat, const FeatCols<KS0>& fc, int F, int64_t first, int64_t N,
                                          float (&x)[PB][KS0]) {
  const int j = ps_lane() & 15;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int64_t p = first + pb * 16 + j;
    const float* row = feat + p * F;
#pragma unroll
    for (int t = 0; t < KS0; ++t) x[pb][t] = (p < N && fc.off[t] >= 0) ? row[fc.off[t]] : 0.0f;
  }
}
template <int KS0, int PB>
__device__ __forceinline__ void store_dfeat(float* __restrict__ dfeat, const FeatCols<KS0>& fc, int F, int64_t first, int64_t N,
                                            const float (&dx)[PB][((KS0 + 3) / 4) * 4]) {
  const int j = ps_lane() & 15;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int64_t p = first + pb * 16 + j;
    float* row = dfeat + p * F;
#pragma unroll
    for (int t = 0; t < KS0; ++t)
      if (p < N && fc.off[t] >= 0) row[fc.off[t]] = dx[pb][t];
  }
}

}  // namespace ps
