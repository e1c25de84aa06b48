// Multiresolution hash-grid encoding, operator level (HashEncoding.forward / backward).
// Semantics: the reference's torch fallback, ns/field_components/encodings.py:324-384
// (ceil/floor corners, every level hashed, trilinear blend in x,y,z order).
//
// Launch shape: one thread per (point, level); a 256-thread workgroup handles 256
// consecutive points of ONE level.  Work items are ordered level-major and dealt to the
// 8 XCDs in contiguous ranges (workgroup b runs on XCD b%8 on MI355X), so each XCD's 4 MiB
// L2 only ever sees ~L/8 of the per-level tables instead of all of them.
#include "common.hpp"
#include "hashgrid_core.hpp"

namespace {

__device__ __forceinline__ void xcd_work_item(int64_t n_items, int64_t& item, bool& valid) {
  // block b -> XCD x = b % 8, slot i = b / 8;  XCD x owns items [x*per, (x+1)*per)
  const int64_t b = blockIdx.x;
  const int64_t per = (n_items + 7) / 8;
  const int64_t x = b & 7, i = b >> 3;
  item = x * per + i;
  valid = (i < per) && (item < n_items);
}

template <int F>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                           const float* __restrict__ scalings, int L, int log2T,
                                                           int64_t N, float* __restrict__ out) {
  const int64_t chunks = (N + 255) / 256;
  int64_t item;
  bool valid;
  xcd_work_item(chunks * L, item, valid);
  if (!valid) return;
  const int level = (int)(item / chunks);
  const int64_t n = (item % chunks) * 256 + threadIdx.x;
  if (n >= N) return;
  const float s = scalings[level];
  const uint32_t mask = (1u << log2T) - 1u;
  const float* tl = table + ((int64_t)level << log2T) * F;
  ps::Cell c = ps::make_cell(x[n * 3 + 0], x[n * 3 + 1], x[n * 3 + 2], s);
  float v[F];
  ps::encode_level<F>(tl, c, mask, v);
  float* o = out + n * (int64_t)(L * F) + level * F;
#pragma unroll
  for (int f = 0; f < F; ++f) o[f] = v[f];
}

template <int F>
__global__ __launch_bounds__(256) void hashgrid_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                           const float* __restrict__ scalings, int L, int log2T,
                                                           int64_t N, float* __restrict__ dtable) {
  const int64_t chunks = (N + 255) / 256;
  int64_t item;
  bool valid;
  xcd_work_item(chunks * L, item, valid);
  if (!valid) return;
  const int level = (int)(item / chunks);
  const int64_t n = (item % chunks) * 256 + threadIdx.x;
  if (n >= N) return;
  const float s = scalings[level];
  const uint32_t mask = (1u << log2T) - 1u;
  float* tl = dtable + ((int64_t)level << log2T) * F;
  ps::Cell c = ps::make_cell(x[n * 3 + 0], x[n * 3 + 1], x[n * 3 + 2], s);
  float g[F];
  const float* d = dout + n * (int64_t)(L * F) + level * F;
#pragma unroll
  for (int f = 0; f < F; ++f) g[f] = d[f];
  ps::scatter_level<F>(tl, c, mask, g);
}

__global__ void hashgrid_index_kernel(const float* __restrict__ x, const float* __restrict__ scalings, int L, int log2T,
                                      int64_t N, int64_t* __restrict__ idx) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N * L) return;
  const int64_t n = i / L;
  const int level = (int)(i % L);
  const uint32_t mask = (1u << log2T) - 1u;
  ps::Cell c = ps::make_cell(x[n * 3 + 0], x[n * 3 + 1], x[n * 3 + 2], scalings[level]);
  uint32_t h[8];
  ps::corner_hashes(c, mask, h);
#pragma unroll
  for (int k = 0; k < 8; ++k) idx[i * 8 + k] = (int64_t)h[k] + ((int64_t)level << log2T);
}

}  // namespace

extern "C" int ps_hashgrid_fwd(const float* x, const float* table, const float* scalings, int L, int F, int log2T,
                               int64_t N, float* out, void* stream) {
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_hashgrid_fwd: features_per_level must be 1, 2 or 4");
  PS_REQUIRE(log2T >= 1 && log2T <= 30 && L >= 1, "ps_hashgrid_fwd: bad L/log2T");
  if (N == 0) return 0;
  const int64_t chunks = (N + 255) / 256;
  const int64_t per = (chunks * L + 7) / 8;
  dim3 grid((unsigned)(per * 8)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (F == 1) hashgrid_fwd_kernel<1><<<grid, block, 0, s>>>(x, table, scalings, L, log2T, N, out);
  if (F == 2) hashgrid_fwd_kernel<2><<<grid, block, 0, s>>>(x, table, scalings, L, log2T, N, out);
  if (F == 4) hashgrid_fwd_kernel<4><<<grid, block, 0, s>>>(x, table, scalings, L, log2T, N, out);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_hashgrid_bwd(const float* x, const float* dout, const float* scalings, int L, int F, int log2T,
                               int64_t N, float* dtable, void* stream) {
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_hashgrid_bwd: features_per_level must be 1, 2 or 4");
  PS_REQUIRE(log2T >= 1 && log2T <= 30 && L >= 1, "ps_hashgrid_bwd: bad L/log2T");
  if (N == 0) return 0;
  const int64_t chunks = (N + 255) / 256;
  const int64_t per = (chunks * L + 7) / 8;
  dim3 grid((unsigned)(per * 8)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (F == 1) hashgrid_bwd_kernel<1><<<grid, block, 0, s>>>(x, dout, scalings, L, log2T, N, dtable);
  if (F == 2) hashgrid_bwd_kernel<2><<<grid, block, 0, s>>>(x, dout, scalings, L, log2T, N, dtable);
  if (F == 4) hashgrid_bwd_kernel<4><<<grid, block, 0, s>>>(x, dout, scalings, L, log2T, N, dtable);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_hashgrid_indices(const float* x, const float* scalings, int L, int log2T, int64_t N, int64_t* idx,
                                   void* stream) {
  if (N == 0) return 0;
  const int64_t total = N * L;
  hashgrid_index_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(x, scalings, L, log2T,
                                                                                                      N, idx);
  PS_CHECK_LAUNCH();
}
