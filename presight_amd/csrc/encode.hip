// Field-level hash-grid path: sample positions -> normalise/contract -> multires encode, and the
// table backward.  Data layout in HBM is "level planes": feat[l][n][f] (and dfeat likewise), so that
//   * the encode kernel (one workgroup = 2048 points of ONE level) writes coalesced,
//   * the MFMA field kernels read 16 consecutive points of a plane per lane group,
//   * the backward scatter streams exactly one plane.
//
// Forward (ps_grid_encode): work items are dealt to the 8 XCDs level-major, so that an XCD's L2 holds the level it is
// working on; lanes 2i / 2i+1 fetch the floor-x / ceil-x corners of point i in one gather (grid_encode_pair_kernel).
// The kernel is bound by L2 random-line requests (tools/microbench/gather.hip), not by bytes.
//
// Backward: MI355X sustains ~20 G global atomics/s and LDS *float* atomics retire ~1 lane / 10 cycles
// (profiles/r01_microbench_atomics.txt), both far too slow for the ~10^9 corner contributions of a step.  The product
// path is the BINNED fixed-point scatter in the second half of this file (ps_grid_scatter_binned: records shuffled
// through HBM into the stream of the table slice that owns their row, reduced with int64 LDS atomics); the slice-owner
// float scan right below (ps_grid_scatter: every owner workgroup streams all points of its level and accumulates the
// corners that fall into its LDS-resident slice with ds_add_f32) is kept as the independent cross-check of
// tests/test_hip_fields.py.  Multi-sub-field launches (K tables, csrc/ms_core.hpp) share both kernels.
//
// Reference semantics: ns/cameras/rays.py:49-58, ns/fields/PreSight/ingp_field.py:169-177,
// ns/field_components/encodings.py:343-384 (forward) and its autograd (index_put_ scatter-add).
#include "common.hpp"
#include "adam_core.hpp"
#include "encode_core.hpp"
#include "hashgrid_core.hpp"
#include "ms_core.hpp"
#include "pointwise_core.hpp"

namespace {

using ps::enc_item;
using ps::enc_parts;
using ps::xcd_item;

// u[n] = contract(normalise(position n)), sel[n] in {0,1}.  Positions are either given (pos != null)
// or generated from rays: point n = ray n/S, sample n%S.
__global__ void field_points_kernel(const float* __restrict__ pos, const float* __restrict__ origins,
                                    const float* __restrict__ dirs, const float* __restrict__ ebins, int S,
                                    const float* __restrict__ aabb, int contract, int64_t N, float* __restrict__ u,
                                    float* __restrict__ sel) {
#pragma clang fp contract(off)  // o + d*t rounded like torch (mul, then add): at 16384^3 resolution one ulp of u matters
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= N) return;
  float p[3];
  if (pos != nullptr) {
    p[0] = pos[n * 3];
    p[1] = pos[n * 3 + 1];
    p[2] = pos[n * 3 + 2];
  } else {
    const int64_t r = n / S;
    const int s = (int)(n % S);
    const float mid = (ebins[r * (S + 1) + s] + ebins[r * (S + 1) + s + 1]) / 2.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = origins[r * 3 + k] + dirs[r * 3 + k] * mid;
  }
  float q[3];
  const bool s_ = ps::normalize_contract(p[0], p[1], p[2], aabb, contract != 0, q);
  u[n * 3] = q[0];
  u[n * 3 + 1] = q[1];
  u[n * 3 + 2] = q[2];
  sel[n] = s_ ? 1.0f : 0.0f;
}

// COUNT: the training forward also counts, per (level, table slice of the binned backward), the records the backward
// will emit for these points (an upper bound: the backward drops records whose gradient is exactly zero), which saves
// the backward its own counting pass over all hashes.  A workgroup then walks `group` consecutive 256-point chunks of
// one level so that its LDS histogram is flushed with 1/group as many global atomics.
constexpr int kEncMaxSlices = 256;
template <int F, bool COUNT>
__global__ __launch_bounds__(256) void grid_encode_kernel(const float* __restrict__ u, const float* __restrict__ table,
                                                          const float* __restrict__ scalings, int L, int log2T, int64_t N,
                                                          int64_t plane_stride, float* __restrict__ feat,
                                                          unsigned* __restrict__ slice_counts, int log2_slice, int group) {
  __shared__ unsigned cnt[COUNT ? kEncMaxSlices : 1];
  const int64_t chunks = (N + 255) / 256;
  const int64_t groups = (chunks + group - 1) / group;
  int64_t item;
  bool valid;
  xcd_item(groups * L, item, valid);
  if (!valid) return;
  const int level = (int)(item / groups);
  const int n_slices = COUNT ? (1 << (log2T - log2_slice)) : 0;
  if constexpr (COUNT) {
    for (int i = threadIdx.x; i < n_slices; i += 256) cnt[i] = 0u;
    __syncthreads();
  }
  const uint32_t mask = (1u << log2T) - 1u;
  const float scale = scalings[level];
  const float* tl = table + ((int64_t)level << log2T) * F;
  for (int64_t chunk = (item % groups) * group, end = min(chunks, chunk + group); chunk < end; ++chunk) {
    const int64_t n = chunk * 256 + threadIdx.x;
    if (n >= N) break;  // only the last chunk is ragged; no barrier inside the loop
    ps::Cell c = ps::make_cell(u[n * 3], u[n * 3 + 1], u[n * 3 + 2], scale);
    float v[F];
    ps::encode_level<F>(tl, c, mask, v);
    float* o = feat + level * plane_stride + n * F;
    if constexpr (F == 1) o[0] = v[0];
    if constexpr (F == 2) *reinterpret_cast<f32x2*>(o) = (f32x2){v[0], v[1]};
    if constexpr (F == 4) *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
    if constexpr (COUNT) {
      // same record rule as bin_kernel: one record per x-pair in the slice of its floor-x corner, plus one in the
      // slice of the ceil-x corner when the pair straddles a slice boundary
      uint32_t h[8];
      ps::corner_hashes(c, mask, h);
      const bool together = ((((uint32_t)c.cx ^ (uint32_t)c.fx) & mask) >> log2_slice) == 0u;
      const uint32_t hf[4] = {h[3], h[2], h[7], h[6]}, hc[4] = {h[0], h[1], h[4], h[5]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(&cnt[hf[k] >> log2_slice], 1u);
        if (!together) atomicAdd(&cnt[hc[k] >> log2_slice], 1u);
      }
    }
  }
  if constexpr (COUNT) {
    __syncthreads();
    for (int i = threadIdx.x; i < n_slices; i += 256)
      if (cnt[i]) atomicAdd(&slice_counts[level * n_slices + i], cnt[i]);
  }
}

// Lane-paired variant: the two corners of an x-pair (floor-x / ceil-x, same y and z) hash to rows that differ only in
// low bits, i.e. they almost always share a cache line, but as two separate gather instructions they cost two L1 tag
// lookups per lane.  Here lanes 2i / 2i+1 fetch the floor-x / ceil-x corners of point i in the SAME instruction, so the
// L1 sees 32 distinct lines per gather instead of 64; the x-blend takes the partner's product through a DPP quad swap
// (a + b is commutative, so the reference's  v_ceil*ox + v_floor*(1-ox)  is reproduced bit for bit in both lanes).
template <int F, bool COUNT>
__global__ __launch_bounds__(256) void grid_encode_pair_kernel(const float* __restrict__ u, const float* __restrict__ table,
                                                               const float* __restrict__ scalings, int L, int log2T, int64_t N,
                                                               int64_t plane_stride, float* __restrict__ feat,
                                                               unsigned* __restrict__ slice_counts, int log2_slice, int group,
                                                               const float* const* __restrict__ tables,
                                                               const int* __restrict__ chunk_field, int parts) {
#pragma clang fp contract(off)
  __shared__ unsigned cnt[COUNT ? kEncMaxSlices : 1];
  const int64_t chunks = (N + 127) / 128;  // 128 points per pass of a 256-thread workgroup
  const int64_t groups = (chunks + group - 1) / group;
  int level;
  int64_t my_group;
  bool valid;
  enc_item(groups, L, parts, level, my_group, valid);
  if (!valid) return;
  // multi-sub-field launch (ms_core.hpp): a group of 2048 points is one chunk of the sorted layout and reads ITS sub-field's table
  if (chunk_field != nullptr) {
    const int kf = chunk_field[my_group];
    if (kf < 0) return;
    table = tables[kf];
    if constexpr (COUNT) slice_counts += (int64_t)kf * L * (1 << (log2T - log2_slice));
  }
  const int n_slices = COUNT ? (1 << (log2T - log2_slice)) : 0;
  if constexpr (COUNT) {
    for (int i = threadIdx.x; i < n_slices; i += 256) cnt[i] = 0u;
    __syncthreads();
  }
  const uint32_t mask = (1u << log2T) - 1u;
  const float scale = scalings[level];
  const float* tl = table + ((int64_t)level << log2T) * F;
  const int side = threadIdx.x & 1;  // 0: floor-x corners, 1: ceil-x corners
  for (int64_t chunk = my_group * group, end = min(chunks, chunk + group); chunk < end; ++chunk) {
    const int64_t n = chunk * 128 + (threadIdx.x >> 1);
    const bool ok = n < N;  // both lanes of a pair agree; inactive pairs still take part in the DPP swap
    const int64_t nn = ok ? n : N - 1;
    ps::Cell c = ps::make_cell(u[nn * 3], u[nn * 3 + 1], u[nn * 3 + 2], scale);
    const uint32_t xs = (uint32_t)(side ? c.cx : c.fx);
    const uint32_t yc = (uint32_t)c.cy * 2654435761u, yf = (uint32_t)c.fy * 2654435761u;
    const uint32_t zc = (uint32_t)c.cz * 805459861u, zf = (uint32_t)c.fz * 805459861u;
    // (y,z) combos in the order of the reference's blend: (c,c) (f,c) | (c,f) (f,f)
    const uint32_t h[4] = {(xs ^ yc ^ zc) & mask, (xs ^ yf ^ zc) & mask, (xs ^ yc ^ zf) & mask, (xs ^ yf ^ zf) & mask};
    ps::Row<F> r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k].load(tl + (size_t)h[k] * F);
    const float wx = side ? c.ox : 1.0f - c.ox;
    const float oy = c.oy, oz = c.oz, uy = 1.0f - oy, uz = 1.0f - oz;
    float v[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float p[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float mine = r[k].v[f] * wx;
        const float other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine), 0xB1, 0xF, 0xF, false));
        p[k] = side ? mine + other : other + mine;  // = v_ceil*ox + v_floor*(1-ox) in both lanes
      }
      const float f0312 = p[0] * oy + p[1] * uy;
      const float f4756 = p[2] * oy + p[3] * uy;
      v[f] = f0312 * oz + f4756 * uz;
    }
    if (ok && side == 0) {
      float* o = feat + level * plane_stride + n * F;
      if constexpr (F == 1) o[0] = v[0];
      if constexpr (F == 2) *reinterpret_cast<f32x2*>(o) = (f32x2){v[0], v[1]};
      if constexpr (F == 4) *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
    }
    if constexpr (COUNT) {
      // one record per x-pair in the slice of its floor-x corner (counted by the floor lane) + one in the slice of the
      // ceil-x corner when the pair straddles a slice boundary (counted by the ceil lane): same rule as bin_kernel
      const bool together = ((((uint32_t)c.cx ^ (uint32_t)c.fx) & mask) >> log2_slice) == 0u;
      if (ok && (side == 0 || !together)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&cnt[h[k] >> log2_slice], 1u);
      }
    }
  }
  if constexpr (COUNT) {
    __syncthreads();
    for (int i = threadIdx.x; i < n_slices; i += 256)
      if (cnt[i]) atomicAdd(&slice_counts[level * n_slices + i], cnt[i]);
  }
}

constexpr int kSliceBytes = 128 * 1024;
constexpr int kScatterThreads = 1024;

template <int F>
__global__ __launch_bounds__(kScatterThreads) void grid_scatter_kernel(const float* __restrict__ u,
                                                                       const float* __restrict__ dfeat,
                                                                       const float* __restrict__ scalings, int L, int log2T,
                                                                       int log2_slice, int64_t N, int64_t plane_stride,
                                                                       float* __restrict__ dtable, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float slice[];  // [entries][F]
  const int entries = 1 << log2_slice;
  const int n_slices = 1 << (log2T - log2_slice);
  int64_t item;
  bool valid;
  xcd_item((int64_t)L * n_slices, item, valid);
  if (!valid) return;
  const int level = (int)(item / n_slices);
  const uint32_t my_slice = (uint32_t)(item % n_slices);
  for (int i = threadIdx.x; i < entries * F; i += kScatterThreads) slice[i] = 0.0f;
  __syncthreads();
  const float s = scalings[level];
  const uint32_t mask = (1u << log2T) - 1u, low = (uint32_t)entries - 1u;
  const float* g_plane = dfeat + level * plane_stride;
  for (int64_t n = threadIdx.x; n < N; n += kScatterThreads) {
    ps::Cell c = ps::make_cell(u[n * 3], u[n * 3 + 1], u[n * 3 + 2], s);
    uint32_t h[8];
    ps::corner_hashes(c, mask, h);
    bool any = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) any |= ((h[k] >> log2_slice) == my_slice);
    if (!any) continue;
    float g[F];
    if constexpr (F == 1) g[0] = g_plane[n];
    if constexpr (F == 2) {
      const f32x2 t = *reinterpret_cast<const f32x2*>(g_plane + n * 2);
      g[0] = t.x;
      g[1] = t.y;
    }
    if constexpr (F == 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(g_plane + n * 4);
      g[0] = t.x;
      g[1] = t.y;
      g[2] = t.z;
      g[3] = t.w;
    }
    const float ox = c.ox, oy = c.oy, oz = c.oz, ux = 1.0f - ox, uy = 1.0f - oy, uz = 1.0f - oz;
    const float w[8] = {ox * oy * oz, ox * uy * oz, ux * uy * oz, ux * oy * oz,
                        ox * oy * uz, ox * uy * uz, ux * uy * uz, ux * oy * uz};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if ((h[k] >> log2_slice) == my_slice && w[k] != 0.0f) {
        float* p = slice + (h[k] & low) * F;
#pragma unroll
        for (int f = 0; f < F; ++f) atomicAdd(p + f, w[k] * g[f]);  // ds_add_f32
      }
    }
  }
  __syncthreads();
  float* out = dtable + (((int64_t)level << log2T) + ((int64_t)my_slice << log2_slice)) * F;
  if (accumulate) {
    for (int i = threadIdx.x; i < entries * F; i += kScatterThreads) out[i] += slice[i];
  } else {
    for (int i = threadIdx.x; i < entries * F; i += kScatterThreads) out[i] = slice[i];
  }
}

}  // namespace

extern "C" int ps_field_points(const float* pos, const float* origins, const float* dirs, const float* ebins, int S,
                               const float* aabb, int contract, int64_t N, float* u, float* sel, void* stream) {
  if (N == 0) return 0;
  PS_REQUIRE(pos != nullptr || (origins && dirs && ebins && S > 0), "ps_field_points: need positions or rays");
  field_points_kernel<<<(unsigned)((N + 255) / 256), 256, 0, (hipStream_t)stream>>>(pos, origins, dirs, ebins, S, aabb, contract,
                                                                                   N, u, sel);
  PS_CHECK_LAUNCH();
}

namespace {
int binned_log2_slice(int F, int log2T);  // defined with the binned backward below
}

// number of table slices per level of the binned backward (= length of one level's row in `slice_counts`)
extern "C" int ps_grid_scatter_slices(int F, int log2T) { return 1 << (log2T - binned_log2_slice(F, log2T)); }

namespace {
// shared by ps_grid_encode (one table) and ps_grid_encode_ms (K tables, chunk -> sub-field map)
int grid_encode_impl(const float* u, const float* table, const float* const* tables, const int* chunk_field, int K,
                     const float* scalings, int L, int F, int log2T, int64_t N, int64_t plane_stride, float* feat,
                     uint32_t* slice_counts, hipStream_t s) {
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_grid_encode: features_per_level must be 1, 2 or 4");
  const int ls = binned_log2_slice(F, log2T);
  if (slice_counts != nullptr) {
    PS_REQUIRE((1 << (log2T - ls)) <= kEncMaxSlices, "ps_grid_encode: too many table slices");
    hipError_t e = hipMemsetAsync(slice_counts, 0, (size_t)K * L * (1 << (log2T - ls)) * 4, s);
    if (e != hipSuccess) { ps_set_error(hipGetErrorString(e)); return (int)e; }
  }
  if (N == 0) return 0;
  static const bool paired = getenv("PS_ENCODE_UNPAIRED") == nullptr;
  PS_REQUIRE(paired || chunk_field == nullptr, "ps_grid_encode_ms: the unpaired debugging kernel has no multi-sub-field mode");
  const int group = paired ? 16 : 8;  // 2048 points of one level per workgroup (also measured faster than 256 without counting)
  static_assert(16 * 128 == ps::kMsChunk, "a paired encode group is one chunk of the multi-sub-field layout");
  const int64_t chunks = paired ? (N + 127) / 128 : (N + 255) / 256;
  const int64_t groups = (chunks + group - 1) / group;
  const int64_t per = (groups * L + 7) / 8;
  dim3 grid((unsigned)(per * 8)), block(256);
  if (paired) {
    const int P = enc_parts(L);
    grid = dim3((unsigned)(8 * (int64_t)(L * P / 8) * ((groups + P - 1) / P)));  // enc_item
#define PS_ENCP(FF)                                                                                                          \
  if (F == FF) {                                                                                                            \
    if (slice_counts != nullptr)                                                                                            \
      grid_encode_pair_kernel<FF, true><<<grid, block, 0, s>>>(u, table, scalings, L, log2T, N, plane_stride, feat, slice_counts, ls, group, tables, chunk_field, P); \
    else                                                                                                                    \
      grid_encode_pair_kernel<FF, false><<<grid, block, 0, s>>>(u, table, scalings, L, log2T, N, plane_stride, feat, nullptr, ls, group, tables, chunk_field, P);     \
  }
    PS_ENCP(1) PS_ENCP(2) PS_ENCP(4)
#undef PS_ENCP
    PS_CHECK_LAUNCH();
  }
#define PS_ENC(FF)                                                                                                          \
  if (F == FF) {                                                                                                            \
    if (slice_counts != nullptr)                                                                                            \
      grid_encode_kernel<FF, true><<<grid, block, 0, s>>>(u, table, scalings, L, log2T, N, plane_stride, feat, slice_counts, ls, group); \
    else                                                                                                                    \
      grid_encode_kernel<FF, false><<<grid, block, 0, s>>>(u, table, scalings, L, log2T, N, plane_stride, feat, nullptr, ls, group);     \
  }
  PS_ENC(1) PS_ENC(2) PS_ENC(4)
#undef PS_ENC
  PS_CHECK_LAUNCH();
}
}  // namespace

extern "C" int ps_grid_encode(const float* u, const float* table, const float* scalings, int L, int F, int log2T, int64_t N,
                              int64_t plane_stride, float* feat, uint32_t* slice_counts, void* stream) {
  return grid_encode_impl(u, table, nullptr, nullptr, 1, scalings, L, F, log2T, N, plane_stride, feat, slice_counts, (hipStream_t)stream);
}

extern "C" int ps_grid_encode_ms(const float* u, const float* const* tables, const float* scalings, int L, int F, int log2T,
                                 int64_t n_slots, int64_t plane_stride, float* feat, uint32_t* slice_counts, int K,
                                 const int32_t* chunk_field, void* stream) {
  PS_REQUIRE(tables != nullptr && chunk_field != nullptr && K >= 1, "ps_grid_encode_ms: need the table pointers and the chunk map");
  PS_REQUIRE(n_slots % ps::kMsChunk == 0, "ps_grid_encode_ms: the sorted layout is a whole number of chunks");
  return grid_encode_impl(u, nullptr, tables, chunk_field, K, scalings, L, F, log2T, n_slots, plane_stride, feat, slice_counts,
                          (hipStream_t)stream);
}

extern "C" int ps_grid_scatter(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t N,
                               int64_t plane_stride, float* dtable, int accumulate, void* stream) {
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_grid_scatter: features_per_level must be 1, 2 or 4");
  int log2_slice = 0;
  while ((1 << (log2_slice + 1)) * F * 4 <= kSliceBytes) ++log2_slice;
  if (log2_slice > log2T) log2_slice = log2T;
  const int n_slices = 1 << (log2T - log2_slice);
  const int64_t items = (int64_t)L * n_slices;
  const int64_t per = (items + 7) / 8;
  const size_t lds = (size_t)(1 << log2_slice) * F * 4;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(per * 8)), block(kScatterThreads);
#define PS_LAUNCH_SCATTER(FF)                                                                                         \
  {                                                                                                                   \
    static bool attr_set = false;                                                                                     \
    if (!attr_set) {                                                                                                  \
      hipFuncSetAttribute((const void*)grid_scatter_kernel<FF>, hipFuncAttributeMaxDynamicSharedMemorySize, kSliceBytes); \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    grid_scatter_kernel<FF><<<grid, block, lds, s>>>(u, dfeat, scalings, L, log2T, log2_slice, N, plane_stride, dtable, \
                                                     accumulate);                                                    \
  }
  if (F == 1) PS_LAUNCH_SCATTER(1)
  if (F == 2) PS_LAUNCH_SCATTER(2)
  if (F == 4) PS_LAUNCH_SCATTER(4)
#undef PS_LAUNCH_SCATTER
  PS_CHECK_LAUNCH();
}

// =================================================================================================
// Binned table backward (ps_grid_scatter_binned) — the fast path.
//
// The slice-owner scan above makes all ~32 owners of a level recompute every point's hashes, and its LDS *float*
// atomics retire ~1 lane / 10 cycles (profiles/r01_microbench_atomics.txt).  Here every corner contribution is
// computed ONCE, shuffled through HBM into the record stream of the slice that owns its table row, and each slice is
// then reduced by one workgroup with dense 64-bit INTEGER LDS atomics (3.3 T lane-ops/s chip-wide) on fixed-point
// values:  fixed = rint(w*g * 2^e),  e = 36 - ceil(log2(max|g|))  (max|g| per level, from a reduction over d(features)).
// |fixed| <= 2^36 and a row receives at most 8N <= 2^26 contributions, so the int64 sums cannot overflow; the
// resolution is 2^-36 of the largest feature gradient, i.e. finer than the fp32 sums of the reference for anything
// that matters, and — integer addition being associative — the gradient is bit-reproducible run to run.
//   phase 0  per-level absmax over d(features): published by the field backward kernel that produced them (proposal
//            fields) or folded into phase A's write pass, which reads every d(feature) anyway
//   phase A  bin_kernel x2: count pass (records per (level, slice)) -> exclusive prefix -> write pass: hashes +
//            weights, LDS counting sort of the workgroup's records by slice, one global cursor reservation per
//            (workgroup, slice), coalesced record runs to HBM at their exact final position
//   phase B  accumulate_kernel: one workgroup per (level, slice); records in, 128 KiB of int64 accumulators in LDS,
//            slice out (+= into the pre-zeroed table gradient)
// Stream sizes are exact (count pass -> prefix sum -> write pass), so nothing is ever dropped or capped.
// =================================================================================================
namespace {

// Workgroup shape of the bin kernel: threads x points per thread.  A workgroup's records leave as one run per slice: (F + 2) 4-byte
// planes of (points x 4 / slices) records, so the POINTS per workgroup set the run length (partial cache lines below ~32 records) and
// the LDS staging area, hence how many workgroups share a CU; the THREADS set how many waves work on it.  The kernel is a chain of
// latencies (point loads -> hashes -> LDS counting sort -> one global reservation per slice -> stores), so more waves per staged
// record help: 512 threads instead of 256 at the same points per workgroup, measured on one box in alternation --
//   cfg 2 14.16 -> 13.90 ms per step (main table backward 2.67 -> 2.58 with 1024 points, proposal 1.08 -> 1.04 with 512),
//   cfg 3 24.5 -> 24.2 ms, cfg 4 (4-D table, 512 instead of 256 points) 50.6 -> 49.0 ms.
// History of the points: F = 4 tables have 256 slices per level, 512 points gave runs of 8 records = 32 bytes per plane (PMC,
// production tile: 6.6 GB written for 4.0 GB of records), 1024 points double them (4.97 -> 4.34 ms).
// Round 5, F = 2 (cfg 2 / cfg 4 main tables): 1024 threads x ONE point instead of 512 x 2 -- the same 1024 points and 71 KiB of staging
// per workgroup, two workgroups per CU, but 8 waves per SIMD instead of 4 (PMC: the 512-thread kernel's waves WAIT 63 % of their
// cycles and issue 16 %) -- four alternating triples on one box: cfg 2 13.58 - 13.72 (512 x 2) -> 13.33 - 13.40 ms per step, cfg 4
// 49.3 -> 48.9 ms.  The same shape for F = 1 (proposal tables: 1024 instead of 512 points) measured neutral on cfg 2 and 0.3 ms
// WORSE on the routed production tile (proposal table backward 1.64 -> 1.78 ms), 1024 x 2 points worse everywhere: F = 1 stays.
#ifndef PS_BIN_THREADS
#define PS_BIN_THREADS 512
#endif
#ifndef PS_BIN_THREADS_F4
#define PS_BIN_THREADS_F4 PS_BIN_THREADS
#endif
#ifndef PS_BIN_THREADS_4D
#define PS_BIN_THREADS_4D 512
#endif
#ifndef PS_BIN_THREADS_F2
#define PS_BIN_THREADS_F2 1024
#endif
constexpr int bin_threads(int D, int F) { return D == 4 ? PS_BIN_THREADS_4D : (F == 4 ? PS_BIN_THREADS_F4 : (F == 2 ? PS_BIN_THREADS_F2 : PS_BIN_THREADS)); }
#ifndef PS_BIN_PPT
#define PS_BIN_PPT 1  // F = 2
#endif
#ifndef PS_BIN_PPT_F1
#define PS_BIN_PPT_F1 1
#endif
#ifndef PS_BIN_PPT_F4
#define PS_BIN_PPT_F4 2
#endif
#ifndef PS_BIN_PPT_4D
#define PS_BIN_PPT_4D 1
#endif
constexpr int bin_points_per_thread(int D, int F) { return D == 4 ? PS_BIN_PPT_4D : (F == 4 ? PS_BIN_PPT_F4 : (F == 1 ? PS_BIN_PPT_F1 : PS_BIN_PPT)); }
constexpr int bin_points(int D, int F) { return bin_threads(D, F) * bin_points_per_thread(D, F); }  // points per workgroup
constexpr int kMaxSlices = 256;
constexpr int kAccBytes = 128 * 1024;

// scale exponent from the max-|g| bit pattern: 2^e * gmax in [2^35, 2^36)
__device__ __forceinline__ float fixed_scale(unsigned gmax_bits, int headroom_log2) {
  if (gmax_bits == 0u) return 1.0f;
  const int ex = (int)((gmax_bits >> 23) & 0xffu) - 127;  // floor(log2 gmax)
  return ldexpf(1.0f, headroom_log2 - (ex + 1));
}

// __float2ll_rn(x) for |x| < 2^54 in six vector instructions (the library routine is a software sequence of ~20): rint in fp32 is exact,
// an integer-valued float splits exactly into a multiple of 2^24 and a remainder below it, and both halves fit 32-bit conversions
__device__ __forceinline__ long long fixed_ll(float x) {
  const float r = rintf(x);
  const float hi = floorf(r * (1.0f / 16777216.0f));
  const float lo = r - hi * 16777216.0f;  // in [0, 2^24): exact
  return (long long)(int)hi * 16777216ll + (long long)(int)lo;
}

// level 0 as a dense histogram (level0_hist_kernel below): side of the level's cube of cell corners, and whether it fits the budget
constexpr int kDense0Bytes = 80 * 1024;  // LDS budget of the histogram = its size in the workspace
__device__ __forceinline__ int dense0_side(const float* __restrict__ scalings) { return (int)ceilf(scalings[0]) + 1; }
__device__ __forceinline__ bool dense0_fits(int R, int F) { return (int64_t)R * R * R * F * 8 <= kDense0Bytes; }

// COUNT_ONLY = true : pass 1, per-(level, slice) record counts (cursors[] += bucket sizes)
// COUNT_ONLY = false: pass 2, cursors[] hold the exclusive prefix (stream start) of every (level, slice) and are
//                     advanced by the reservations; records are written at their exact final position.
// D = 3: u [N,3].  D = 4 (dynamic.hip, the (x,y,z,t) grid of the dynamic field): u [N,4], 8 x-pairs per (point, level) -- the
// (y,z) combinations at the ceil-t corner, then at the floor-t corner -- and half as many points per workgroup.
// period > 0 (up to three position sets of `period` points each, dynamic.hip): point n < period takes row n of dfeat, a point
// n >= period takes row (n - period) mod period of dfeat_b (dfeat itself when dfeat_b is null).
template <int F, bool COUNT_ONLY, int D = 3>
__global__ __launch_bounds__(bin_threads(D, F)) void bin_kernel(const float* __restrict__ u, const float* __restrict__ dfeat,
                                                          const float* __restrict__ scalings, int L, int log2T,
                                                          int log2_slice, int64_t N, int64_t plane_stride, int64_t n_rec_max,
                                                          unsigned* __restrict__ cursors, unsigned* __restrict__ rec_idx,
                                                          float* __restrict__ rec_val, const int* __restrict__ chunk_field,
                                                          unsigned* __restrict__ gmax_track /* nullable: per-level max |dfeat| bits */,
                                                          int64_t period, const float* __restrict__ dfeat_b,
                                                          int dense0 /* level 0 goes through level0_hist_kernel (when its cube fits) */) {
  constexpr int kBinPointsPerThread = bin_points_per_thread(D, F);
  constexpr int kBinThreads = bin_threads(D, F);
  constexpr int kBinPoints = kBinThreads * kBinPointsPerThread;  // points per workgroup
  constexpr int NP = D == 4 ? 8 : 4;  // x-pairs per (point, level)
  // LDS: per-slice counters / offsets / global bases + staged records (idx + F values) + slice id per staged record.
  // The staging area holds exactly the NP pair records of every point; the second record of a SPLIT pair (rare, see below)
  // never enters it.
  constexpr int kRec = COUNT_ONLY ? 1 : kBinPoints * NP;
  __shared__ unsigned cnt[kMaxSlices], off[kMaxSlices + 1], gbase[kMaxSlices];
  __shared__ unsigned s_idx[kRec];
  __shared__ float s_val[F + 1][kRec];  // plane F holds ox
  __shared__ unsigned char s_slice[kRec];
  const int n_slices = 1 << (log2T - log2_slice);
  // chunk-major: the workgroups in flight spread over all L levels, so their stream reservations (returning global atomics
  // on the 2 cache lines of a level's cursors) contend 1/L as much, and the L reads of a chunk's points hit in L2
  const int level = (int)(blockIdx.x % L);
  const int64_t first = (blockIdx.x / L) * kBinPoints;
  if constexpr (!COUNT_ONLY && D == 3) {
    if (dense0 && level == 0 && dense0_fits(dense0_side(scalings), F)) return;  // (workgroup-uniform; the count pass still counts level 0)
  }
  if (chunk_field != nullptr) {  // multi-sub-field launch: (sub-field, level) takes the place of the level in the stream index
    const int kf = chunk_field[first / ps::kMsChunk];
    if (kf < 0) return;
    cursors += (int64_t)kf * L * n_slices;
    if (gmax_track != nullptr) gmax_track += kf * L;
  }
  for (int i = threadIdx.x; i < n_slices; i += kBinThreads) cnt[i] = 0u;
  __syncthreads();
  const float s = scalings[level];
  const uint32_t mask = (1u << log2T) - 1u, low = (1u << log2_slice) - 1u;
  const float* g_plane_a = dfeat + level * plane_stride;
  const float* g_plane_b = (dfeat_b != nullptr ? dfeat_b : dfeat) + level * plane_stride;
  // hashes, weights, local position inside the slice bucket.  The two corners of an x-pair (ceil/floor in x) have
  // hashes that differ by cx^fx = 2^(t+1)-1 (low bits only), so they almost always live in the same slice: they travel
  // as ONE record {row of the floor corner, t, q[F] = w_yz * g[F], ox}; the accumulate kernel expands it.  A pair whose
  // xor reaches the slice bits -- x crosses a multiple of the slice's row count: impossible while the level's resolution is
  // below it (all of cfg 2), probability ~2^-log2_slice otherwise -- is SPLIT into two single-corner records (t = 31): the
  // floor corner takes the pair's slot, the ceil corner is emitted on a rare side path (one global reservation and one
  // uncoalesced write per record).  Keeping that second record out of the common path halves its per-thread record slots.
  // pairs per point-level, in (y,z) corner order: (c,c) (f,c) (c,f) (f,f) [D = 4: at ceil t, then the same four at floor t]
  float gmax_local = 0.f;  // write pass: max |d(feature)| of this thread's points (the accumulate kernel's fixed-point scale)
  uint32_t r_slice[kBinPointsPerThread][NP], r_pos[kBinPointsPerThread][NP], r_idx[kBinPointsPerThread][NP];
  float r_val[kBinPointsPerThread][NP][F], r_ox[kBinPointsPerThread][NP];
  bool any_split = false;
  // one point's cell, hashes and weights (also recomputed on the split side path)
  struct Pt {
    bool ok;
    float g[F], ox;
    uint32_t fx, cx, hyz[NP], xdiff;
    float wyz[NP];
  };
  auto point = [&](int q) {
    Pt p;
    const int64_t n = first + q * kBinThreads + threadIdx.x;
    p.ok = n < N;
#pragma unroll
    for (int f = 0; f < F; ++f) p.g[f] = 0.f;
    ps::Cell c = ps::make_cell(p.ok ? u[n * D] : 0.f, p.ok ? u[n * D + 1] : 0.f, p.ok ? u[n * D + 2] : 0.f, s);
    int64_t ng = n;
    const float* g_plane = g_plane_a;
    if (period > 0 && n >= period) {
      ng = n - period;
      if (ng >= period) ng -= period;
      g_plane = g_plane_b;
    }
    if (p.ok) {
      if constexpr (F == 1) p.g[0] = g_plane[ng];
      if constexpr (F == 2) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(g_plane + ng * 2);
        p.g[0] = t.x;
        p.g[1] = t.y;
      }
      if constexpr (F == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(g_plane + ng * 4);
        p.g[0] = t.x;
        p.g[1] = t.y;
        p.g[2] = t.z;
        p.g[3] = t.w;
      }
    }
    const uint32_t yc = (uint32_t)c.cy * 2654435761u, yf = (uint32_t)c.fy * 2654435761u;
    const uint32_t zc = (uint32_t)c.cz * 805459861u, zf = (uint32_t)c.fz * 805459861u;
    const float oy = c.oy, oz = c.oz, uy = 1.0f - oy, uz = 1.0f - oz;
    const uint32_t h4[4] = {yc ^ zc, yf ^ zc, yc ^ zf, yf ^ zf};
    const float w4[4] = {oy * oz, uy * oz, oy * uz, uy * uz};
    if constexpr (D == 4) {
#pragma clang fp contract(off)
      const float st = (p.ok ? u[n * D + 3] : 0.f) * s;
      const float flt = floorf(st), ot = st - flt;
      const uint32_t tc = (uint32_t)(int)ceilf(st) * 3674653429u, tf = (uint32_t)(int)flt * 3674653429u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        p.hyz[k] = h4[k] ^ tc;
        p.hyz[4 + k] = h4[k] ^ tf;
        p.wyz[k] = w4[k] * ot;
        p.wyz[4 + k] = w4[k] * (1.0f - ot);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        p.hyz[k] = h4[k];
        p.wyz[k] = w4[k];
      }
    }
    p.fx = (uint32_t)c.fx;
    p.cx = (uint32_t)c.cx;
    p.ox = c.ox;
    p.xdiff = (p.cx ^ p.fx) & mask;  // 0 (exact integer) or 2^(t+1)-1
    return p;
  };
#pragma unroll
  for (int q = 0; q < kBinPointsPerThread; ++q) {
    const Pt p = point(q);
    if constexpr (!COUNT_ONLY) {
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float a = fabsf(p.g[f]);
        gmax_local = (a == a) ? fmaxf(gmax_local, a) : __builtin_inff();  // a NaN stays visible as "non-finite"
      }
    }
    const bool together = (p.xdiff >> log2_slice) == 0u;
    const uint32_t tcode = p.xdiff == 0u ? 30u : (uint32_t)(31 - __clz((int)p.xdiff));  // t (xdiff = 2^(t+1)-1), 30 = same row
    const float wf = together ? 1.0f : 1.0f - p.ox;  // single floor corner of a split pair: its weight is folded into the values
    bool any_rec = false;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const uint32_t hf = (p.fx ^ p.hyz[k]) & mask;
      bool any = false;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float qv = p.wyz[k] * p.g[f];
        any |= (qv != 0.0f);
        r_val[q][k][f] = qv * wf;
      }
      r_slice[q][k] = hf >> log2_slice;
      r_idx[q][k] = (hf & low) | ((together ? tcode : 31u) << 16);
      r_ox[q][k] = together ? p.ox : 0.0f;
      const bool emit = p.ok && any;
      any_rec |= emit;
      r_pos[q][k] = emit ? atomicAdd(&cnt[r_slice[q][k]], 1u) : 0xffffffffu;  // ds_add_rtn_u32
    }
    any_split |= any_rec && !together && p.ox != 0.0f;
  }
  if (__builtin_expect(__ballot(any_split) != 0ull, 0)) {
    // split pairs: the ceil-x corner of every emitted pair as a single-corner record in ITS slice's stream
#pragma unroll
    for (int q = 0; q < kBinPointsPerThread; ++q) {
      const Pt p = point(q);
      if (!p.ok || (p.xdiff >> log2_slice) == 0u || p.ox == 0.0f) continue;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        float qv[F];
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) {
          qv[f] = p.wyz[k] * p.g[f];
          any |= (qv[f] != 0.0f);
        }
        if (!any) continue;
        const uint32_t hc = (p.cx ^ p.hyz[k]) & mask;
        const unsigned dst = atomicAdd(&cursors[level * n_slices + (hc >> log2_slice)], 1u);
        if constexpr (!COUNT_ONLY) {
          rec_idx[dst] = (hc & low) | (31u << 16);
#pragma unroll
          for (int f = 0; f < F; ++f) rec_val[(int64_t)f * n_rec_max + dst] = qv[f] * p.ox;
          rec_val[(int64_t)F * n_rec_max + dst] = 0.0f;
        }
      }
    }
  }
  __syncthreads();
  if constexpr (COUNT_ONLY) {
    if ((int)threadIdx.x < n_slices && cnt[threadIdx.x]) atomicAdd(&cursors[level * n_slices + threadIdx.x], cnt[threadIdx.x]);
    return;
  } else {
    // exclusive scan of the bucket sizes by the first wavefront (4 slices per lane + one wave scan, no block barriers)
    if (threadIdx.x < 64) {
      const int b = threadIdx.x * 4;
      unsigned c4[4], sum = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c4[i] = (b + i < n_slices) ? cnt[b + i] : 0u;
        sum += c4[i];
      }
      unsigned incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d, 64);
        if ((int)threadIdx.x >= d) incl += o;
      }
      unsigned run = incl - sum;
      if (threadIdx.x == 0) off[0] = 0u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        run += c4[i];
        off[b + i + 1] = run;
      }
    }
    __syncthreads();
    // one global reservation per (workgroup, slice): absolute position in the record arrays
    // (the returned position is kept in a register and goes to LDS only AFTER the staging below: the kernel is a chain of
    //  latencies -- 131 k short workgroups -- and this way the round trip of the reservation overlaps the staging)
    unsigned my_base = 0u;
    if ((int)threadIdx.x < n_slices) {
      const unsigned c = cnt[threadIdx.x];
      if (c) my_base = atomicAdd(&cursors[level * n_slices + threadIdx.x], c);
    }
    // the level's max |d(feature)| for the accumulate kernel's fixed-point scale: folded into this pass, which reads every
    // d(feature) anyway (saves the separate absmax pass over the plane).  The atomic is skipped when the published maximum is
    // already at least as large (a stale read only costs a redundant atomic: the maximum is monotonic; 0.5 M same-address
    // atomics per launch would serialise in L2 for milliseconds).  The published maximum is requested here, next to the
    // reservations, and looked at after the staging (it used to stand alone between two workgroup barriers).
    unsigned my_bits = 0u, published = 0xffffffffu;
    if (gmax_track != nullptr) {
      float m = gmax_local;
#pragma unroll
      for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
      if (ps_lane() == 0 && !(m <= 0.f)) {
        my_bits = isfinite(m) ? __float_as_uint(m) : 0x7fc00000u;
        // (agent-scope load: served by L2 — a plain load would sit in this CU's L1 and never see the other CUs' updates)
        published = __hip_atomic_load(gmax_track + level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // stage the records in bucket order.  The bucket offsets of all record slots are fetched first: written as "read the
    // offset, store the record" per slot, every slot exposed an LDS round trip (the compiler keeps the read behind the
    // previous slot's stores)
    unsigned r_dst[kBinPointsPerThread][NP];
#pragma unroll
    for (int q = 0; q < kBinPointsPerThread; ++q)
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const unsigned o = off[r_slice[q][k]];  // unconditional (r_slice is always a valid slice): no branch around the LDS read
        r_dst[q][k] = (r_pos[q][k] != 0xffffffffu) ? o + r_pos[q][k] : 0xffffffffu;
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < kBinPointsPerThread; ++q)
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const unsigned p = r_dst[q][k];
        if (p < (unsigned)kRec) {
          s_idx[p] = r_idx[q][k];
          s_slice[p] = (unsigned char)r_slice[q][k];
#pragma unroll
          for (int f = 0; f < F; ++f) s_val[f][p] = r_val[q][k][f];
          s_val[F][p] = r_ox[q][k];
        }
      }
    if ((int)threadIdx.x < n_slices) gbase[threadIdx.x] = my_base;
    if (my_bits > published) atomicMax(gmax_track + level, my_bits);  // (published stays 0xffffffff where nothing was requested)
    __syncthreads();
    // coalesced runs to the slice streams
    const unsigned total = off[n_slices];
    constexpr int kOut = 4;  // records per thread and round: their LDS look-ups (slice -> stream position) are issued together
    for (unsigned p0 = threadIdx.x; p0 < total; p0 += kBinThreads * kOut) {
      unsigned sl[kOut];
#pragma unroll
      for (int i = 0; i < kOut; ++i) sl[i] = (p0 + i * kBinThreads < total) ? s_slice[p0 + i * kBinThreads] : 0u;
      int64_t dst[kOut];
#pragma unroll
      for (int i = 0; i < kOut; ++i) dst[i] = (int64_t)gbase[sl[i]] + ((p0 + i * kBinThreads) - off[sl[i]]);
#pragma unroll
      for (int i = 0; i < kOut; ++i) {
        const unsigned p = p0 + i * kBinThreads;
        if (p < total) {
          rec_idx[dst[i]] = s_idx[p];
#pragma unroll
          for (int f = 0; f <= F; ++f) rec_val[(int64_t)f * n_rec_max + dst[i]] = s_val[f][p];
        }
      }
    }
  }
}

// ---- level 0 as a DENSE histogram (round 6) ------------------------------------------------------------------------------
// The coarsest level of a grid has (ceil(scale_0) + 1)^3 live cells (17^3 = 4913 at the reference's min_res = 16) however many
// points there are: as records its contributions are 1/L of the record traffic of both passes and -- consecutive samples of a ray
// share their cell -- the hot rows of the accumulate pass (2.7 x the time of a fine level, profiles/r05_scatter_levels.txt).
// Here they never become records: a workgroup sums the fixed-point contributions of its points per CELL CORNER in LDS (int64,
// the same rint(w g 2^e) terms the accumulate kernel would form, so the result is bit-identical), adds its non-zero cells to a
// dense int64 histogram in the workspace, and the accumulate workgroup of a level-0 slice adds the cells whose hash lands in
// its slice to its accumulators before the flush.  Requirements (host: one table, 3-D, no position sets, accumulate_kernel
// path; device: the cube fits the LDS budget): otherwise level 0 travels as records like every other level.
// A point with a corner outside the cube (|u| beyond [0, 1]: possible without scene contraction) is written as ordinary
// records into the level-0 streams (sized for all of them by the count pass), one reservation per record.

// max |d(feature)| of level 0 (the record writer folds this into its pass over the levels it writes; it no longer reads level 0)
template <int F>
__global__ __launch_bounds__(1024) void level0_absmax_kernel(const float* __restrict__ dfeat, const float* __restrict__ scalings, int64_t N,
                                                             unsigned* __restrict__ gmax_bits) {
  if (!dense0_fits(dense0_side(scalings), F)) return;
  float m = 0.f;
  const int64_t n = N * F, n4 = ((uintptr_t)dfeat & 15) == 0 ? n / 4 : 0;  // 16-byte loads over the aligned body, scalars for the rest
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 1024) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(dfeat + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = fabsf(t[k]);
      m = (a == a) ? fmaxf(m, a) : __builtin_inff();
    }
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) {
    const float a = fabsf(dfeat[i]);
    m = (a == a) ? fmaxf(m, a) : __builtin_inff();
  }
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
  // one atomic per WORKGROUP (same-address atomics serialise in L2: one per wave cost this kernel 0.1 ms for 10 us of reading)
  __shared__ float wmax[16];
  if (ps_lane() == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, wmax[w]);  // (a NaN never reaches here: non-finite inputs were turned into +inf above)
    if (!(m <= 0.f)) atomicMax(gmax_bits, isfinite(m) ? __float_as_uint(m) : 0x7fc00000u);
  }
}

// 256-thread workgroups: one wave per SIMD with < 96 registers and <= 80 KiB of LDS starts on a compute unit that holds a 400-register
// wave of the main field's matrix kernels on every SIMD (the proposal chain runs beside them); at 1024 threads the kernel waited for
// whole compute units (0.05 ms alone, 0.5 ms on average in the step)
constexpr int kHistThreads = 256;
template <int F>
__global__ __launch_bounds__(kHistThreads) void level0_hist_kernel(const float* __restrict__ u, const float* __restrict__ dfeat,
                                                           const float* __restrict__ scalings, int log2T, int log2_slice, int64_t N,
                                                           int64_t n_rec_max, unsigned* __restrict__ cursors, unsigned* __restrict__ rec_idx,
                                                           float* __restrict__ rec_val, const unsigned* __restrict__ gmax_bits,
                                                           int headroom_log2, unsigned long long* __restrict__ hist0) {
  constexpr int kRun = 8;  // consecutive points per thread: samples of one ray, merged in registers while they stay in one cell
  extern __shared__ __attribute__((aligned(16))) unsigned long long hist[];  // [R^3][F]
  const int R = dense0_side(scalings);
  if (!dense0_fits(R, F)) return;
  const int cells = R * R * R;
  const unsigned gbits = gmax_bits[0];
  if (gbits >= 0x7f800000u) return;  // a non-finite gradient on the level: the accumulate pass writes NaN whatever the sums are
  const float scale = fixed_scale(gbits, headroom_log2);
  const float s = scalings[0];
  const uint32_t mask = (1u << log2T) - 1u, low = (1u << log2_slice) - 1u;
  for (int i = threadIdx.x; i < cells * F; i += kHistThreads) hist[i] = 0ull;
  __syncthreads();
  int tgt[8];
  long long sum[8][F];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    tgt[k] = -1;
#pragma unroll
    for (int f = 0; f < F; ++f) sum[k][f] = 0;
  }
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (tgt[k] >= 0) {
#pragma unroll
        for (int f = 0; f < F; ++f)
          if (sum[k][f] != 0) atomicAdd(&hist[tgt[k] * F + f], (unsigned long long)sum[k][f]);  // ds_add_u64
      }
    }
  };
  for (int64_t base = (int64_t)blockIdx.x * kHistThreads * kRun; base < N; base += (int64_t)gridDim.x * kHistThreads * kRun) {
    const int64_t n0 = base + (int64_t)threadIdx.x * kRun;
    // all inputs of the run requested before the first is used (one memory round trip per run instead of one per point: with two
    // waves per SIMD nothing else hides them)
    float ux[kRun][3], gx[kRun][F];
#pragma unroll
    for (int q = 0; q < kRun; ++q) {
      const int64_t n = n0 + q < N ? n0 + q : N - 1;  // (an address select, not a conditional load)
#pragma unroll
      for (int d = 0; d < 3; ++d) ux[q][d] = u[n * 3 + d];
#pragma unroll
      for (int f = 0; f < F; ++f) gx[q][f] = dfeat[n * F + f];
    }
#pragma unroll
    for (int q = 0; q < kRun; ++q) {
      const int64_t n = n0 + q;
      if (n >= N) break;
      float g[F];
      bool live = false;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        g[f] = gx[q][f];
        live |= (g[f] != 0.0f);
      }
      if (!live) continue;  // (every product below would be zero: the record writer emits nothing for such a point either)
      const ps::Cell c = ps::make_cell(ux[q][0], ux[q][1], ux[q][2], s);
      const float oy = c.oy, oz = c.oz, uy = 1.0f - oy, uz = 1.0f - oz;
      const float w4[4] = {oy * oz, uy * oz, oy * uz, uy * uz};  // (y, z) corner order of the record writer: (c,c) (f,c) (c,f) (f,f)
      const int y4[4] = {c.cy, c.fy, c.cy, c.fy}, z4[4] = {c.cz, c.cz, c.fz, c.fz};
      const bool inside = c.fx >= 0 && c.fy >= 0 && c.fz >= 0 && c.cx < R && c.cy < R && c.cz < R;
      if (__builtin_expect(!inside, 0)) {
        // outside the cube: the records bin_kernel would have written for this (point, level 0)
        const uint32_t fx = (uint32_t)c.fx, cx = (uint32_t)c.cx, xdiff = (cx ^ fx) & mask;
        const bool together = (xdiff >> log2_slice) == 0u;
        const uint32_t tcode = xdiff == 0u ? 30u : (uint32_t)(31 - __clz((int)xdiff));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t hyz = ((uint32_t)y4[k] * 2654435761u) ^ ((uint32_t)z4[k] * 805459861u);
          float qv[F];
          bool any = false;
#pragma unroll
          for (int f = 0; f < F; ++f) {
            qv[f] = w4[k] * g[f];
            any |= (qv[f] != 0.0f);
          }
          if (!any) continue;
          const uint32_t hf = (fx ^ hyz) & mask;
          unsigned dst = atomicAdd(&cursors[hf >> log2_slice], 1u);
          rec_idx[dst] = (hf & low) | ((together ? tcode : 31u) << 16);
#pragma unroll
          for (int f = 0; f < F; ++f) rec_val[(int64_t)f * n_rec_max + dst] = together ? qv[f] : qv[f] * (1.0f - c.ox);
          rec_val[(int64_t)F * n_rec_max + dst] = together ? c.ox : 0.0f;
          if (!together && c.ox != 0.0f) {
            const uint32_t hc = (cx ^ hyz) & mask;
            dst = atomicAdd(&cursors[hc >> log2_slice], 1u);
            rec_idx[dst] = (hc & low) | (31u << 16);
#pragma unroll
            for (int f = 0; f < F; ++f) rec_val[(int64_t)f * n_rec_max + dst] = qv[f] * c.ox;
            rec_val[(int64_t)F * n_rec_max + dst] = 0.0f;
          }
        }
        continue;
      }
      // the eight corner cells in the order (floor-x, ceil-x) x (y, z) corner; a ceil-x corner that coincides with the floor one
      // (exact integer x: ox == 0) receives nothing, as in the accumulate kernel's t == 30 case
      const bool same_x = c.cx == c.fx;
      const float wf = same_x ? 1.0f : 1.0f - c.ox;
      int t8[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t8[k] = c.fx + R * (y4[k] + R * z4[k]);
        t8[4 + k] = same_x ? -1 : c.cx + R * (y4[k] + R * z4[k]);
      }
      bool same = true;
#pragma unroll
      for (int k = 0; k < 8; ++k) same &= (t8[k] == tgt[k]);
      if (!same) {
        flush();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          tgt[k] = t8[k];
#pragma unroll
          for (int f = 0; f < F; ++f) sum[k][f] = 0;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const float qv = w4[k] * g[f];
          sum[k][f] += fixed_ll(qv * wf * scale);
          if (!same_x) sum[4 + k][f] += fixed_ll(qv * c.ox * scale);
        }
      }
    }
  }
  flush();
  __syncthreads();
  for (int i = threadIdx.x; i < cells * F; i += kHistThreads) {
    const unsigned long long v = hist[i];
    if (v != 0ull) atomicAdd(&hist0[i], v);
  }
}

// exclusive prefix over the (level, slice) counts: starts[i], and cursors[i] := starts[i] for the second pass.
// One workgroup walks the n items in tiles of 4 TPB (four consecutive items per thread: coalesced loads / stores), wave scans + a
// 16-entry table per tile, the running total carried in a register.  (The first version gave every thread one contiguous run of n / 1024
// items: strided, uncoalesced accesses, twice -- 240 us for the 41 k items of a routed production tile, per table backward.)
// src: where the record counts come from -- the cursors themselves (counted by bin_kernel<COUNT_ONLY>) or the slice counts of the
// forward encode (read directly: the device-to-device copy into the cursors was a launch of its own); zero_bits / n_zero: the
// per-level absmax words to clear for the record-writing pass that follows (another memset launch otherwise); zero_hist / n_zero_hist:
// the dense level-0 histogram (level0_hist_kernel) to clear.
// TPB = 256 for tables of <= 4096 streams: this kernel sits at the head of a table backward that runs BESIDE the main field's matrix
// kernels (one 350 - 420-register wave per SIMD), and a 1024-thread workgroup (4 waves x 40 registers per SIMD) found no compute unit to
// start on until one of their workgroups retired -- 0.6 - 1.2 ms of waiting for 5 us of work on the proposal chain (round-6 timeline).
template <int TPB>
__global__ __launch_bounds__(TPB) void stream_offsets_kernel(unsigned* __restrict__ cursors, unsigned* __restrict__ counts,
                                                             unsigned* __restrict__ starts, int n, const unsigned* __restrict__ src,
                                                             unsigned* __restrict__ zero_bits, int n_zero,
                                                             unsigned long long* __restrict__ zero_hist, int n_zero_hist) {
  __shared__ unsigned wsum[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < n_zero; i += TPB) zero_bits[i] = 0u;
  for (int i = threadIdx.x; i < n_zero_hist; i += TPB) zero_hist[i] = 0ull;  // (the level-0 histogram of the pass that follows)
  unsigned carry = 0;
  for (int base = 0; base < n; base += TPB * 4) {
    const int i = base + (int)threadIdx.x * 4;
    unsigned c[4], r[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c[k] = (i + k < n) ? src[i + k] : 0u;
      r[k] = (c[k] + 3u) & ~3u;  // every stream starts on a multiple of 4 records (vector loads)
      s += r[k];
    }
    unsigned incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) {
      const unsigned v = wsum[w];
      before += w < wave ? v : 0u;
      total += v;
    }
    unsigned run = carry + before + (incl - s);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k < n) {
        counts[i + k] = c[k];
        starts[i + k] = run;
        cursors[i + k] = run;
        run += r[k];
      }
    carry += total;
    __syncthreads();  // wsum is rewritten by the next tile
  }
}

// Adam applied in the accumulate kernel's flush (ps_grid_scatter_binned_adam): single-process training exchanges no gradients, and a
// hash table receives exactly one gradient contribution per step, so the finished gradient of a slice never has to leave the chip --
// the flush reads the slice's parameters and moments, applies ps::adam_update (the optimizer kernels' own function: same bits) and
// writes them back.  Per parameter and step that removes the gradient's store, its load by the optimizer kernel and next step's
// zero fill: 12 of the 40 bytes a table entry moves per step (a production tile: 940 M entries).  g_base == null: not fused.
struct AdamFuse {
  const float* g_base;  // the flat gradient buffer the destination pointers point into: element offset = out - g_base
  float *p_base, *m_base, *v_base;  // flat parameters / first / second moments, same layout
  ps::AdamHyper h;
  float bc1, bc2_sqrt;            // bias corrections at the host-side step count (tables without a device-decided group)
  const int* group_of_field;      // [K] device-decided group of every sub-field (routed tiles; < 0 / null: host-decided), see adam.hip
  const int* flags;
  const int* steps;
};

template <int F>
__global__ __launch_bounds__(1024) void accumulate_kernel(const unsigned* __restrict__ cursors, const unsigned* __restrict__ starts,
                                                          const unsigned* __restrict__ rec_idx,
                                                          const float* __restrict__ rec_val, const unsigned* __restrict__ gmax_bits,
                                                          int L, int log2T, int log2_slice, int64_t n_rec_max, int headroom_log2,
                                                          int accumulate, float* __restrict__ dtable, float* const* __restrict__ dtables,
                                                          float out_scale /* the gradient is multiplied by this (1: exactly the plain result) */,
                                                          int item0 /* first item of this launch (the items may be dealt to several launches) */,
                                                          AdamFuse A, const float* __restrict__ scalings,
                                                          const unsigned long long* __restrict__ hist0 /* nullable: dense level-0 sums */) {
  extern __shared__ __attribute__((aligned(16))) long long acc[];  // [entries][F]
  const int entries = 1 << log2_slice;
  const int n_slices = 1 << (log2T - log2_slice);
  const int item = item0 + blockIdx.x;  // (sub-field * L + level) * n_slices + slice
  const int vlevel = item / n_slices, sl = item % n_slices;
  const int level = vlevel % L;
  const int64_t base = starts[item];  // multiple of 4 records (stream_offsets_kernel) -> 16-byte aligned vector loads
  const int64_t n = cursors[item] - base;  // the write pass advanced the cursor from the stream start to its end
  const bool fused = A.g_base != nullptr;
  // level 0 of a single table may have been summed per cell corner instead of written as records (level0_hist_kernel): its slices
  // then take those sums, whatever records they hold besides
  const int R0 = (hist0 != nullptr && vlevel == 0) ? dense0_side(scalings) : 0;
  const bool from_hist = R0 > 0 && dense0_fits(R0, F);
  if (n == 0 && accumulate && !fused && !from_hist) return;  // nothing to add (a sub-field without points: most slices of a multi-sub-field launch)
  float bc1 = A.bc1, bc2_sqrt = A.bc2_sqrt;
  if (fused && A.group_of_field != nullptr) {
    const int grp = A.group_of_field[vlevel / L];
    if (grp >= 0) {  // workgroup-uniform.  A sub-field that received no samples is not updated at all (adam.hip: torch with grad None)
      if (A.flags[grp] == 0) return;
      __shared__ float s_bc[2];
      if (threadIdx.x == 0) ps::adam_bias_corrections(A.h.b1, A.h.b2, A.steps[grp] + 1, s_bc[0], s_bc[1]);
      __syncthreads();
      bc1 = s_bc[0];
      bc2_sqrt = s_bc[1];
    }
  }
  if (dtables != nullptr) dtable = dtables[vlevel / L];
  const unsigned gbits = gmax_bits[vlevel];
  float* out = dtable + (((int64_t)level << log2T) + ((int64_t)sl << log2_slice)) * F;
  // Every lane takes kChunk CONSECUTIVE records of the stream and merges neighbours that hit the same pair of rows in
  // registers (int64 adds: associative, so merging does not change the result) before touching LDS.  The bin kernel
  // emits the records of one wavefront instruction in lane order = consecutive samples of a ray, which share their
  // cell on the coarse levels: there a handful of hot rows would otherwise serialise the LDS atomics of a whole wave.
  // record = {row | t<<16, ox, q[F]}: t < 30 -> pair (rows e and e ^ (2^(t+1)-1), weights 1-ox / ox), t == 30 -> both
  // corners on the same row (exact integer x), t == 31 -> single corner with its weight already applied.
  constexpr int kChunk = 8;
  unsigned e[kChunk];
  float v[F][kChunk], ox[kChunk];
  auto load_batch = [&](int64_t i0) {
#pragma unroll
    for (int h = 0; h < kChunk; h += 4) {
      const int64_t i = base + i0 + h;
      const u32x4 t = *reinterpret_cast<const u32x4*>(rec_idx + i);  // reads past n stay inside the workspace
      const f32x4 o = *reinterpret_cast<const f32x4*>(rec_val + (int64_t)F * n_rec_max + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        e[h + k] = t[k];
        ox[h + k] = o[k];
      }
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(rec_val + (int64_t)f * n_rec_max + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[f][h + k] = q[k];
      }
    }
  };
  // A workgroup owns a whole CU (128 KiB of LDS), so nothing overlaps its own memory latencies: the first batch of records and
  // -- for the dense read-modify-write flush -- the slice's current gradient are requested BEFORE the accumulators are zeroed
  // (production shape: ~4 k records per slice, the kernel is a chain of latencies: 3.2 ms for 41 k workgroups)
  const int64_t first_i0 = (int64_t)threadIdx.x * kChunk;
  if (first_i0 < n) load_batch(first_i0);
  const bool dense = fused || from_hist || !(accumulate && n * 4 < entries);  // (fused: every entry of the slice is updated, gradient or not)
  constexpr int kOutPerThread = kAccBytes / 8 / 1024;  // 16 values of the slice per thread
  float prev[kOutPerThread];
#pragma unroll
  for (int k = 0; k < kOutPerThread; ++k) {
    const int i = threadIdx.x + k * 1024;
    prev[k] = (dense && accumulate == 1 && i < entries * F) ? out[i] : 0.0f;  // (accumulate == 2: the destination is known to be zero)
  }
  for (int i = threadIdx.x; i < entries * F; i += 1024) acc[i] = 0;
  __syncthreads();
  const bool nan_level = gbits >= 0x7f800000u;  // a non-finite d(feature) on this level: the gradient is NaN, like torch's index_add of a NaN
  if (nan_level && !fused) {
    for (int i = threadIdx.x; i < entries * F; i += 1024) out[i] = __builtin_nanf("");
    return;
  }
  const float scale = fixed_scale(gbits, headroom_log2);
  const unsigned low = (unsigned)entries - 1u;
  if (from_hist && !nan_level) {
    // the cells whose hash lands in this slice (two cells may share a row: LDS atomics); 4913 cells for the workgroup's 1024 threads
    const uint32_t mask = (1u << log2T) - 1u;
    for (int c = threadIdx.x; c < R0 * R0 * R0; c += 1024) {
      const uint32_t x = (uint32_t)(c % R0), y = (uint32_t)((c / R0) % R0), z = (uint32_t)(c / (R0 * R0));
      const uint32_t h = (x ^ (y * 2654435761u) ^ (z * 805459861u)) & mask;
      if ((int)(h >> log2_slice) == sl) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const unsigned long long v = hist0[c * F + f];
          if (v != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[(h & low) * F + f]), v);
        }
      }
    }
  }
  for (int64_t i0 = first_i0; i0 < n && !nan_level; i0 += 1024 * kChunk) {
    if (i0 != first_i0) load_batch(i0);
    unsigned p_row = 0xffffffffu, p_rowc = 0xffffffffu;
    long long p_f[F], p_c[F];
#pragma unroll
    for (int f = 0; f < F; ++f) p_f[f] = p_c[f] = 0;
    auto flush = [&]() {
      if (p_row != 0xffffffffu) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
          atomicAdd(reinterpret_cast<unsigned long long*>(&acc[p_row * F + f]), (unsigned long long)p_f[f]);  // ds_add_u64
          if (p_rowc != p_row) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[p_rowc * F + f]), (unsigned long long)p_c[f]);
        }
      }
    };
#pragma unroll
    for (int k = 0; k < kChunk; ++k) {
      if (i0 + k < n) {
        const unsigned row = e[k] & 0xffffu, t = e[k] >> 16;
        const bool pair = t < 30u;
        const unsigned row_c = pair ? ((row ^ ((2u << t) - 1u)) & low) : row;
        const float wf = pair ? 1.0f - ox[k] : 1.0f;  // t == 30: c and f coincide (ox == 0); t == 31: weight already applied
        if (row != p_row || row_c != p_rowc) {
          flush();
          p_row = row;
          p_rowc = row_c;
#pragma unroll
          for (int f = 0; f < F; ++f) p_f[f] = p_c[f] = 0;
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
          p_f[f] += __float2ll_rn(v[f][k] * wf * scale);
          if (pair) p_c[f] += __float2ll_rn(v[f][k] * ox[k] * scale);
        }
      }
    }
    flush();
  }
  __syncthreads();
  const float inv = 1.0f / scale;
  if (!dense) {
    // Sparse flush (many sub-fields / few points per table: far fewer records than rows): walk the records once more and let
    // the first lane that reaches a row take its total out of LDS (64-bit exchange with 0) and add it to the gradient;
    // rows no record touched are neither read nor written.  Same values as the dense flush below.
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
      const unsigned e = rec_idx[base + i];
      const unsigned row = e & 0xffffu, t = e >> 16;
      const unsigned row_c = (t < 30u) ? ((row ^ ((2u << t) - 1u)) & low) : row;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned r = h ? row_c : row;
        if (h && row_c == row) break;
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const long long v = (long long)atomicExch(reinterpret_cast<unsigned long long*>(&acc[r * F + f]), 0ull);
          if (v != 0) {
            const float val = (float)((double)v * (double)inv * (double)out_scale);
            out[r * F + f] = accumulate == 1 ? out[r * F + f] + val : val;
          }
        }
      }
    }
    return;
  }
  if (fused) {
    // the slice's Adam step: parameters and both moments in (requested together: one round trip), updated, out; the gradient itself
    // is never written (its buffer keeps the zeros the step started with)
    const int64_t off = out - A.g_base;
    float *pp = A.p_base + off, *mm = A.m_base + off, *vv = A.v_base + off;
    // 16-byte accesses: a thread owns 4 consecutive entries in each of kOutPerThread / 4 rounds (the flat buffers and every slice
    // start on 16-byte boundaries: presight_amd.dist.FlatGrads pads its views, a slice holds a power of two >= 4 of values)
    constexpr int kVec = kOutPerThread / 4;
    f32x4 P[kVec], M[kVec], V[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
      const int i = (threadIdx.x + k * 1024) * 4;
      if (i < entries * F) {
        P[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pp + i));
        M[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mm + i));
        V[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vv + i));
      }
    }
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
      const int i = (threadIdx.x + k * 1024) * 4;
      if (i < entries * F) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float val = nan_level ? __builtin_nanf("") : (float)((double)acc[i + j] * (double)inv * (double)out_scale);
          float pk = P[k][j], mk = M[k][j], vk = V[k][j];
          ps::adam_update(pk, val, mk, vk, A.h, bc1, bc2_sqrt);
          P[k][j] = pk;
          M[k][j] = mk;
          V[k][j] = vk;
        }
        __builtin_nontemporal_store(P[k], reinterpret_cast<f32x4*>(pp + i));
        __builtin_nontemporal_store(M[k], reinterpret_cast<f32x4*>(mm + i));
        __builtin_nontemporal_store(V[k], reinterpret_cast<f32x4*>(vv + i));
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < kOutPerThread; ++k) {
    const int i = threadIdx.x + k * 1024;
    if (i < entries * F) {
      const float val = (float)((double)acc[i] * (double)inv * (double)out_scale);
      out[i] = accumulate == 1 ? prev[k] + val : val;
    }
  }
}

// ---- the fused update on tables of MANY SLICES WITH FEW RECORDS EACH (routed production tile: 16 sub-fields x 10 levels x 256
// slices = 41 k items, ~4 k records per item at 65 536 rays, ~500 at 8192 -- half of the items receive none).  There the pass is a
// stream of p / m / v (24 bytes per entry, 16 GB per launch) with a little LDS work in between, and accumulate_kernel above runs it as
// a chain of latencies: its workgroup owns the CU (128 KiB of LDS), so nothing overlaps record loads -> zero fill -> accumulation ->
// barrier -> p / m / v loads -> update -> stores (4.7 - 4.9 TB/s where a plain Adam stream sustains 6.4, tools/microbench/adam_stream.hip).
//   accumulate_adam_kernel   the same arithmetic with every load of the item requested UP FRONT: the first record batch, then the
//                            slice's parameters and moments (unconditional 16-byte loads: a conditional load makes hipcc merge the
//                            loaded registers behind an s_waitcnt vmcnt(0)) -- the records return first and are
//                            accumulated while p / m / v are still in flight; the first batch is peeled out of the record loop
//                            so that nothing in flight is live across a branch.  The second moments are requested behind the
//                            accumulation (PS_ACC_LATE_V: 16 registers less -- with all three arrays in flight the F = 4 kernel
//                            spills 8 registers at its 128-register budget).  A slice WITHOUT records skips accumulators and
//                            barriers.  Measured on one box, alternating (tools/ab_env.sh, table backward of the main tables =
//                            record writer + this pass): production tile at 65 536 rays 6.61 - 6.90 -> 6.26 - 6.58 ms, at 8192 rays
//                            3.93 - 3.97 -> 3.36 - 3.50 ms (16.1 GB of p / m / v: 2.5 ms at the 6.4 TB/s streaming ceiling).  Handing
//                            the empty slices to a separate streaming launch (256-thread workgroups, the microbenchmark's shape) was
//                            built and measured too: 3.69 - 3.76 ms with it, 3.65 - 3.66 without -- dropped.  So was a variant with
//                            TWO 512-thread workgroups per item, each owning half of the slice's rows in 64 KiB of LDS (two
//                            workgroups per CU cover each other's waits; both read all records, the second read from L2): 6.59 vs
//                            6.53 ms at 65 536 rays, 3.29 - 3.42 vs 3.25 - 3.30 at 8192 -- no gain, dropped.  The pass streams its
//                            20 GB (65 536 rays: 16.1 GB of p / m / v + 4 GB of records) at 4.8 TB/s, 16.2 GB at 5.3 TB/s at 8192 rays.
// Same element update (ps::adam_update) on the same fp32 gradient (integer sums are order-independent): bit-equal to the kernel above.
#ifndef PS_ACC_LATE_V
#define PS_ACC_LATE_V 1
#endif
template <int F, int CH>
__device__ __forceinline__ void acc_consume_batch(long long* acc, const unsigned (&e)[CH], const float (&v)[F][CH], const float (&ox)[CH],
                                                  int64_t i0, int64_t n, unsigned low, float scale) {
  unsigned p_row = 0xffffffffu, p_rowc = 0xffffffffu;
  long long p_f[F], p_c[F];
#pragma unroll
  for (int f = 0; f < F; ++f) p_f[f] = p_c[f] = 0;
  auto flush = [&]() {
    if (p_row != 0xffffffffu) {
#pragma unroll
      for (int f = 0; f < F; ++f) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&acc[p_row * F + f]), (unsigned long long)p_f[f]);  // ds_add_u64
        if (p_rowc != p_row) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[p_rowc * F + f]), (unsigned long long)p_c[f]);
      }
    }
  };
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    if (i0 + k < n) {
      const unsigned row = e[k] & 0xffffu, t = e[k] >> 16;
      const bool pair = t < 30u;
      const unsigned row_c = pair ? ((row ^ ((2u << t) - 1u)) & low) : row;
      const float wf = pair ? 1.0f - ox[k] : 1.0f;
      if (row != p_row || row_c != p_rowc) {
        flush();
        p_row = row;
        p_rowc = row_c;
#pragma unroll
        for (int f = 0; f < F; ++f) p_f[f] = p_c[f] = 0;
      }
#pragma unroll
      for (int f = 0; f < F; ++f) {
        p_f[f] += __float2ll_rn(v[f][k] * wf * scale);
        if (pair) p_c[f] += __float2ll_rn(v[f][k] * ox[k] * scale);
      }
    }
  }
  flush();
}

// the workgroup-uniform preamble of the fused kernels: -> false when the item's sub-field belongs to a device-decided group whose
// flag is down (not updated at all); bias corrections of such a group are evaluated here (one lane, broadcast through LDS)
__device__ __forceinline__ bool fused_item_live(const AdamFuse& A, int field, float& bc1, float& bc2_sqrt) {
  bc1 = A.bc1;
  bc2_sqrt = A.bc2_sqrt;
  if (A.group_of_field != nullptr) {
    const int grp = A.group_of_field[field];
    if (grp >= 0) {
      if (A.flags[grp] == 0) return false;
      __shared__ float s_bc[2];
      if (threadIdx.x == 0) ps::adam_bias_corrections(A.h.b1, A.h.b2, A.steps[grp] + 1, s_bc[0], s_bc[1]);
      __syncthreads();
      bc1 = s_bc[0];
      bc2_sqrt = s_bc[1];
    }
  }
  return true;
}

template <int F>
__global__ __launch_bounds__(1024) void accumulate_adam_kernel(const unsigned* __restrict__ cursors, const unsigned* __restrict__ starts,
                                                               const unsigned* __restrict__ rec_idx, const float* __restrict__ rec_val,
                                                               const unsigned* __restrict__ gmax_bits, int L, int log2T, int log2_slice,
                                                               int64_t n_rec_max, int headroom_log2, float* __restrict__ dtable,
                                                               float* const* __restrict__ dtables, int item0, AdamFuse A) {
  extern __shared__ __attribute__((aligned(16))) long long acc[];  // [entries][F], entries * F == kAccBytes / 8 (host-checked)
  constexpr int CH = F == 4 ? 4 : 8;                 // records per lane and batch
  constexpr int kVec = kAccBytes / 8 / 1024 / 4;     // 16-byte vectors of the slice per thread and array
  constexpr bool kLateV = PS_ACC_LATE_V != 0;        // the second moments requested behind the accumulation (16 registers less)
  const int entries = 1 << log2_slice;
  const int n_slices = 1 << (log2T - log2_slice);
  const int item = item0 + blockIdx.x;
  const int vlevel = item / n_slices, sl = item % n_slices;
  const int level = vlevel % L;
  const int64_t base = starts[item];
  const int64_t n = cursors[item] - base;
  float bc1, bc2_sqrt;
  if (!fused_item_live(A, vlevel / L, bc1, bc2_sqrt)) return;
  if (dtables != nullptr) dtable = dtables[vlevel / L];
  const unsigned gbits = gmax_bits[vlevel];
  float* out = dtable + (((int64_t)level << log2T) + ((int64_t)sl << log2_slice)) * F;
  const int64_t off = out - A.g_base;
  const float *pp = A.p_base + off, *mm = A.m_base + off, *vv = A.v_base + off;
  unsigned e[CH];
  float v[F][CH], ox[CH];
  auto load_batch = [&](int64_t i0) {
#pragma unroll
    for (int h = 0; h < CH; h += 4) {
      const int64_t i = base + i0 + h;
      const u32x4 t = *reinterpret_cast<const u32x4*>(rec_idx + i);  // reads past n stay inside the workspace
      const f32x4 o = *reinterpret_cast<const f32x4*>(rec_val + (int64_t)F * n_rec_max + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        e[h + k] = t[k];
        ox[h + k] = o[k];
      }
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(rec_val + (int64_t)f * n_rec_max + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[f][h + k] = q[k];
      }
    }
  };
  if (n == 0) {
    // A slice without records (half of the items of a production tile at 8192 rays): no accumulators, no barriers -- its update is
    // g = 0 (weight decay and moment decay still reach every entry, as in torch; a non-finite level: NaN for all of its entries)
    const float g0 = gbits >= 0x7f800000u ? __builtin_nanf("") : 0.0f;
    float *ppw = A.p_base + off, *mmw = A.m_base + off, *vvw = A.v_base + off;
    f32x4 P[kVec], M[kVec], V[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
      const int i = (threadIdx.x + k * 1024) * 4;
      P[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pp + i));
      M[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mm + i));
      V[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vv + i));
    }
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
      const int i = (threadIdx.x + k * 1024) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pk = P[k][j], mk = M[k][j], vk = V[k][j];
        ps::adam_update(pk, g0, mk, vk, A.h, bc1, bc2_sqrt);
        P[k][j] = pk;
        M[k][j] = mk;
        V[k][j] = vk;
      }
      __builtin_nontemporal_store(P[k], reinterpret_cast<f32x4*>(ppw + i));
      __builtin_nontemporal_store(M[k], reinterpret_cast<f32x4*>(mmw + i));
      __builtin_nontemporal_store(V[k], reinterpret_cast<f32x4*>(vvw + i));
    }
    return;
  }
  // 1. the first record batch (a lane without records re-reads the stream's first ones: an address select, not a conditional load)
  const int64_t first_i0 = (int64_t)threadIdx.x * CH;
  load_batch(first_i0 < n ? first_i0 : 0);
  // 2. parameters and moments of the slice, behind the records in the (in-order) return queue
  f32x4 P[kVec], M[kVec], V[kVec];
#pragma unroll
  for (int k = 0; k < kVec; ++k) {
    const int i = (threadIdx.x + k * 1024) * 4;
    P[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pp + i));
    M[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mm + i));
    if (!kLateV) V[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vv + i));
  }
  // 3. accumulators
  typedef long long ll2 __attribute__((ext_vector_type(2)));
  for (int i = threadIdx.x * 2; i < entries * F; i += 2048) *reinterpret_cast<ll2*>(&acc[i]) = (ll2){0, 0};
  __syncthreads();
  const bool nan_level = gbits >= 0x7f800000u;
  const float scale = fixed_scale(gbits, headroom_log2);
  const unsigned low = (unsigned)entries - 1u;
  if (!nan_level) {
    acc_consume_batch<F, CH>(acc, e, v, ox, first_i0, n, low, scale);  // (first_i0 >= n: no record passes the i0 + k < n test)
    for (int64_t i0 = first_i0 + 1024 * CH; i0 < n; i0 += 1024 * CH) {
      load_batch(i0);
      acc_consume_batch<F, CH>(acc, e, v, ox, i0, n, low, scale);
    }
  }
  __syncthreads();
  if (kLateV) {
#pragma unroll
    for (int k = 0; k < kVec; ++k) V[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vv + (threadIdx.x + k * 1024) * 4));
  }
  // 4. the slice's Adam step (the gradient itself is never written)
  const float inv = 1.0f / scale;
  float *ppw = A.p_base + off, *mmw = A.m_base + off, *vvw = A.v_base + off;
#pragma unroll
  for (int k = 0; k < kVec; ++k) {
    const int i = (threadIdx.x + k * 1024) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float val = nan_level ? __builtin_nanf("") : (float)((double)acc[i + j] * (double)inv * (double)1.0f);
      float pk = P[k][j], mk = M[k][j], vk = V[k][j];
      ps::adam_update(pk, val, mk, vk, A.h, bc1, bc2_sqrt);
      P[k][j] = pk;
      M[k][j] = mk;
      V[k][j] = vk;
    }
    __builtin_nontemporal_store(P[k], reinterpret_cast<f32x4*>(ppw + i));
    __builtin_nontemporal_store(M[k], reinterpret_cast<f32x4*>(mmw + i));
    __builtin_nontemporal_store(V[k], reinterpret_cast<f32x4*>(vvw + i));
  }
}

// records the streams can hold: 8 per (point, level) worst case + every stream start rounded up to 4 records + the
// vector-load overshoot of the last chunk; a multiple of 4 so that all planes stay 16-byte aligned
int64_t binned_rec_capacity(int64_t N, int L, int n_slices, int D = 3) {
  return ((N * L * (D == 4 ? 16 : 8) + 4 * (int64_t)L * n_slices + 64) + 3) & ~(int64_t)3;
}

int binned_log2_slice(int F, int log2T) {
  int ls = 0;
  while ((int64_t)(1 << (ls + 1)) * F * 8 <= kAccBytes) ++ls;
  if (ls > log2T) ls = log2T;
  while ((log2T - ls) > 8) ++ls;  // at most 256 slices (cannot happen for T <= 2^22)
  return ls;
}

}  // namespace

namespace {
int64_t binned_workspace(int L, int F, int log2T, int64_t N, int K, int D = 3) {
  const int ls = binned_log2_slice(F, log2T);
  const int n_slices = 1 << (log2T - ls);
  const int64_t n_rec_max = binned_rec_capacity(N, L, K * n_slices, D);
  return 4096 + (int64_t)K * L * n_slices * 4 * 3 + 16 + n_rec_max * 4 * (2 + F) + 256 + kDense0Bytes;  // (+ the level-0 histogram, behind the planes)
}

// K = 1, chunk_field = null: one table.  Otherwise the multi-sub-field launch of ms_core.hpp: "level" of every stream,
// cursor and maximum becomes (sub-field, level); N = slots of the sorted layout.
int scatter_binned_impl(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t N,
                        int64_t plane_stride, float* dtable, float* const* dtables, int K, const int* chunk_field, int accumulate,
                        const uint32_t* slice_counts, int absmax_ready, void* workspace, hipStream_t s, int D = 3, int64_t period = 0,
                        float out_scale = 1.0f, const float* dfeat_b = nullptr, int phase = 3, int item_begin = 0, int item_end = -1,
                        const AdamFuse* adam = nullptr) {
  // phase bit 0: prepare (counts, stream offsets, record write pass); bit 1: accumulate the items [item_begin, item_end) (item_end < 0:
  // all).  A caller that exchanges the gradient in pieces launches the accumulate pass once per piece (items are ordered
  // (sub-field, level, slice) = the order of the gradient in memory) and hands every piece over as soon as its launch is enqueued.
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_grid_scatter_binned: features_per_level must be 1, 2 or 4");
  PS_REQUIRE(D == 3 || (D == 4 && K == 1 && chunk_field == nullptr), "ps_grid_scatter_binned: 3-D grids, or one 4-D grid");
  PS_REQUIRE(period == 0 || (period > 0 && N <= 3 * period), "ps_grid_scatter_binned: at most three position sets");
  PS_REQUIRE(N * L * (D == 4 ? 16 : 8) + 4096 + 4 * (int64_t)K * L * 256 < ((int64_t)1 << 32), "ps_grid_scatter_binned: too many contributions for 32-bit stream offsets");
  PS_REQUIRE(K * L <= 1024, "ps_grid_scatter_binned: at most 1024 (sub-field, level) pairs");
  PS_REQUIRE(adam == nullptr || (accumulate == 2 && D == 3 && out_scale == 1.0f),
             "ps_grid_scatter_binned_adam: the fused update needs a zeroed destination it is the only contribution to");
  const AdamFuse fuse = adam != nullptr ? *adam : AdamFuse{nullptr, nullptr, nullptr, nullptr, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 1.f, 1.f, nullptr, nullptr, nullptr};
  const int ls = binned_log2_slice(F, log2T);
  const int n_slices = 1 << (log2T - ls);
  PS_REQUIRE(n_slices <= kMaxSlices, "ps_grid_scatter_binned: too many slices");
  const int64_t n_rec_max = binned_rec_capacity(N, L, K * n_slices, D);
  const int n_items = K * L * n_slices;
  PS_REQUIRE(((uintptr_t)workspace & 15) == 0, "ps_grid_scatter_binned: workspace must be 16-byte aligned");
  char* ws = (char*)workspace;
  unsigned* gmax_bits = (unsigned*)ws;            // [K*L] (+ padding to 4096)
  unsigned* cursors = (unsigned*)(ws + 4096);     // [n_items]
  unsigned* counts = cursors + n_items;           // [n_items]
  unsigned* starts = counts + n_items;            // [n_items]
  unsigned* rec_idx = starts + ((n_items + 3) & ~3);  // [n_rec_max], 16-byte aligned like every plane behind it
  float* rec_val = (float*)(rec_idx + n_rec_max); // [F+1][n_rec_max] (plane F = ox)
  unsigned long long* hist0 = (unsigned long long*)(((uintptr_t)(rec_val + (int64_t)(F + 1) * n_rec_max) + 255) & ~(uintptr_t)255);  // [kDense0Bytes / 8]
  // absmax_ready: the first K*L words of the workspace already hold the per-level max |d(feature)| bits (written by the
  // field backward kernel that produced dfeat) -> keep them and skip the absmax pass
  if (item_end < 0 || item_end > n_items) item_end = n_items;
  PS_REQUIRE(item_begin >= 0 && item_begin <= item_end, "ps_grid_scatter_binned: bad item range");
  hipError_t e = hipSuccess;
  // record counts (upper bounds) from the forward pass (ps_grid_encode): the prefix kernel reads them in place and clears the absmax
  // words itself -- no memset, no copy.  Without them the counting pass needs zeroed cursors (and absmax words) first.
  const bool counted = slice_counts != nullptr && N > 0;
  if ((phase & 1) && !counted) {
    e = absmax_ready ? hipMemsetAsync(ws + 4096, 0, (int64_t)n_items * 4, s) : hipMemsetAsync(ws, 0, 4096 + (int64_t)n_items * 4, s);
    if (e != hipSuccess) { ps_set_error(hipGetErrorString(e)); return (int)e; }
  }
  int headroom = 62 - 26;  // 8N <= 2^26 contributions per row
  {
    int bits = 0;
    while (((int64_t)1 << bits) < N * (D == 4 ? 16 : 8)) ++bits;
    if (bits > 26) headroom = 62 - bits;
  }
  const size_t lds = (size_t)(1 << ls) * F * 8;
  // Fused update on many slices with few records each -> the load-everything-up-front kernel (see accumulate_adam_kernel).  Average
  // records per item decide: a cfg-2 table (1024 items of 262 k records) stays on accumulate_kernel, whose record loop then is the
  // whole pass.  PS_ACC_STREAM=0 / 1 forces the choice (A/B runs).
  static const char* acc_stream_env = getenv("PS_ACC_STREAM");
  const double rec_per_item = (double)N * L * (D == 4 ? 8 : 4) / (double)(n_items > 0 ? n_items : 1);
  bool stream_fused = fuse.g_base != nullptr && lds == (size_t)kAccBytes && rec_per_item < 16.0 * 1024 * 8;
  if (acc_stream_env != nullptr && fuse.g_base != nullptr && lds == (size_t)kAccBytes) stream_fused = acc_stream_env[0] != '0';
  // level 0 as a dense histogram instead of records (level0_hist_kernel), on request (phase bit 3): one table, 3-D, no position sets,
  // the generic accumulate kernel, a histogram budget that holds at least the smallest cube.  Both phases of a split call carry the
  // bit and decide alike; the device side decides (from scalings[0]) whether the level's cube fits.  Measured neutral on cfg 2
  // (EXPERIMENTS.md A.7): presight_amd.field_ops requests it only under PRESIGHT_DENSE_LEVEL0=1.
  const bool dense0 = (phase & 8) && !(phase & 4) && D == 3 && K == 1 && chunk_field == nullptr && period == 0 && !stream_fused && N > 0 &&
                      L > 1 && (int64_t)8 * F * 8 <= kDense0Bytes;
  const int n_zero_hist = dense0 ? kDense0Bytes / 8 : 0;
#define PS_LAUNCH_BINNED(FF) PS_LAUNCH_BINNED_D(FF, 3)
#define PS_LAUNCH_BINNED_D(FF, DD)                                                                                            \
  {                                                                                                                       \
    static bool attr_set = false;                                                                                         \
    if (!attr_set) {                                                                                                      \
      hipFuncSetAttribute((const void*)accumulate_kernel<FF>, hipFuncAttributeMaxDynamicSharedMemorySize, kAccBytes);     \
      attr_set = true;                                                                                                    \
    }                                                                                                                     \
    const int64_t chunks = (N + bin_points(DD, FF) - 1) / bin_points(DD, FF);                                                   \
    if (phase & 1) {                                                                                                      \
      if (N > 0) {                                                                                                        \
        if (slice_counts == nullptr)                                                                                      \
          bin_kernel<FF, true, DD><<<(unsigned)(chunks * L), bin_threads(DD, FF), 0, s>>>(u, dfeat, scalings, L, log2T, ls, N,    \
                                                                              plane_stride, n_rec_max, cursors, rec_idx, rec_val, chunk_field, nullptr, period, dfeat_b, 0); \
      }                                                                                                                   \
      if (n_items <= 4096)                                                                                                \
        stream_offsets_kernel<256><<<1, 256, 0, s>>>(cursors, counts, starts, n_items, counted ? slice_counts : cursors,  \
                                                     (counted && !absmax_ready) ? gmax_bits : nullptr, (counted && !absmax_ready) ? K * L : 0, \
                                                     hist0, n_zero_hist);                                                 \
      else                                                                                                                \
        stream_offsets_kernel<1024><<<1, 1024, 0, s>>>(cursors, counts, starts, n_items, counted ? slice_counts : cursors, \
                                                       (counted && !absmax_ready) ? gmax_bits : nullptr, (counted && !absmax_ready) ? K * L : 0, \
                                                       hist0, n_zero_hist);                                               \
      if (N > 0)                                                                                                          \
        bin_kernel<FF, false, DD><<<(unsigned)(chunks * L), bin_threads(DD, FF), 0, s>>>(u, dfeat, scalings, L, log2T, ls, N,     \
                                                                             plane_stride, n_rec_max, cursors, rec_idx, rec_val, chunk_field, \
                                                                             absmax_ready ? nullptr : gmax_bits, period, dfeat_b, (int)dense0); \
      if constexpr (DD == 3) {                                                                                            \
        if (dense0) {                                                                                                     \
          static bool attr0_set = false;                                                                                  \
          if (!attr0_set) {                                                                                               \
            hipFuncSetAttribute((const void*)level0_hist_kernel<FF>, hipFuncAttributeMaxDynamicSharedMemorySize, kDense0Bytes); \
            attr0_set = true;                                                                                             \
          }                                                                                                               \
          const int64_t per_wg = kHistThreads * 8;                                                                        \
          const unsigned hist_grid = (unsigned)std::min<int64_t>(1024, (N + per_wg - 1) / per_wg);                        \
          if (!absmax_ready) level0_absmax_kernel<FF><<<256, 1024, 0, s>>>(dfeat, scalings, N, gmax_bits);               \
          level0_hist_kernel<FF><<<hist_grid, kHistThreads, kDense0Bytes, s>>>(u, dfeat, scalings, log2T, ls, N, n_rec_max, cursors, rec_idx, \
                                                                       rec_val, gmax_bits, headroom, hist0);              \
        }                                                                                                                 \
      }                                                                                                                   \
    }                                                                                                                     \
    if ((phase & 2) && item_end > item_begin) {                                                                           \
      if (stream_fused) {                                                                                                 \
        static bool attr2_set = false;                                                                                    \
        if (!attr2_set) {                                                                                                 \
          hipFuncSetAttribute((const void*)accumulate_adam_kernel<FF>, hipFuncAttributeMaxDynamicSharedMemorySize, kAccBytes); \
          attr2_set = true;                                                                                               \
        }                                                                                                                 \
        accumulate_adam_kernel<FF><<<(unsigned)(item_end - item_begin), 1024, lds, s>>>(cursors, starts, rec_idx, rec_val, gmax_bits, L, log2T, ls, \
                                                                                        n_rec_max, headroom, dtable, dtables, item_begin, fuse); \
      } else {                                                                                                            \
        accumulate_kernel<FF><<<(unsigned)(item_end - item_begin), 1024, lds, s>>>(cursors, starts, rec_idx, rec_val, gmax_bits, L, log2T, ls, \
                                                                                   n_rec_max, headroom, accumulate, dtable, dtables, out_scale, item_begin, fuse, \
                                                                                   scalings, dense0 ? hist0 : nullptr);           \
      }                                                                                                                   \
    }                                                                                                                     \
  }
  if (D == 3) {
    if (F == 1) PS_LAUNCH_BINNED(1)
    if (F == 2) PS_LAUNCH_BINNED(2)
    if (F == 4) PS_LAUNCH_BINNED(4)
  } else {
    if (F == 1) PS_LAUNCH_BINNED_D(1, 4)
    if (F == 2) PS_LAUNCH_BINNED_D(2, 4)
    if (F == 4) PS_LAUNCH_BINNED_D(4, 4)
  }
#undef PS_LAUNCH_BINNED
#undef PS_LAUNCH_BINNED_D
  PS_CHECK_LAUNCH();
}
}  // namespace

// ---- sparse gradient exchange (round 6; SURVEY.md 8e: "sparse exchange of touched rows") ------------------------------------------
// Under data parallelism the dense table gradient of a production tile is 3.76 GB per step and rank, of which a rank's 8192 rays touch
// a few percent.  The binned backward already holds the touched rows as RECORDS, sorted by the 128 KiB table slice that owns them:
// the ranks exchange the records of a slice with the slice's OWNER (all-to-all of the record streams, presight_amd/dist.py) and the
// owner accumulates all ranks' runs of a slice in ONE int64 LDS accumulator -- fixed point with a scale every rank derives from the
// MAX-reduced per-level |d(feature)| maximum, so the sum is exact and independent of the order of the runs -- converts once, scales by
// 1 / world and writes its shard of the (mean) gradient.  The kernel below is accumulate_kernel's record loop over n_runs runs per item.
namespace {
template <int F>
__global__ __launch_bounds__(1024) void accumulate_runs_kernel(const unsigned* __restrict__ run_starts, const unsigned* __restrict__ run_counts,
                                                               int n_runs, int n_local, const unsigned* __restrict__ rec_idx,
                                                               const float* __restrict__ rec_val, int64_t plane_stride,
                                                               const unsigned* __restrict__ gmax_bits, int L, int log2T, int log2_slice,
                                                               int headroom_log2, float* __restrict__ dtable, float* const* __restrict__ dtables,
                                                               float out_scale, int item0) {
  extern __shared__ __attribute__((aligned(16))) long long acc[];  // [entries][F]
  const int entries = 1 << log2_slice;
  const int n_slices = 1 << (log2T - log2_slice);
  const int li = blockIdx.x, item = item0 + li;  // item = (sub-field * L + level) * n_slices + slice
  const int vlevel = item / n_slices, sl = item % n_slices;
  const int level = vlevel % L;
  if (dtables != nullptr) dtable = dtables[vlevel / L];
  float* out = dtable + (((int64_t)level << log2T) + ((int64_t)sl << log2_slice)) * F;
  const unsigned gbits = gmax_bits[vlevel];
  for (int i = threadIdx.x; i < entries * F; i += 1024) acc[i] = 0;
  __syncthreads();
  if (gbits >= 0x7f800000u) {  // a non-finite d(feature) on this level on SOME rank: the gradient is NaN, as torch's index_add of a NaN
    for (int i = threadIdx.x; i < entries * F; i += 1024) out[i] = __builtin_nanf("");
    return;
  }
  const float scale = fixed_scale(gbits, headroom_log2);
  const unsigned low = (unsigned)entries - 1u;
  for (int r = 0; r < n_runs; ++r) {
    const int64_t base = run_starts[(int64_t)r * n_local + li];  // multiple of 4 records: 16-byte aligned vector loads
    const int64_t n = run_counts[(int64_t)r * n_local + li];
    for (int64_t i0 = (int64_t)threadIdx.x * 4; i0 < n; i0 += 1024 * 4) {
      const int64_t i = base + i0;
      const u32x4 t = *reinterpret_cast<const u32x4*>(rec_idx + i);  // (reads past n stay inside the run's 4-record padding)
      const f32x4 o = *reinterpret_cast<const f32x4*>(rec_val + (int64_t)F * plane_stride + i);
      f32x4 q[F];
#pragma unroll
      for (int f = 0; f < F; ++f) q[f] = *reinterpret_cast<const f32x4*>(rec_val + (int64_t)f * plane_stride + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (i0 + k >= n) break;
        const unsigned row = t[k] & 0xffffu, tt = t[k] >> 16;
        const bool pair = tt < 30u;  // record = {row | t << 16, ox, q[F]}: see accumulate_kernel
        const unsigned row_c = pair ? ((row ^ ((2u << tt) - 1u)) & low) : row;
        const float wf = pair ? 1.0f - o[k] : 1.0f;
#pragma unroll
        for (int f = 0; f < F; ++f) {
          atomicAdd(reinterpret_cast<unsigned long long*>(&acc[row * F + f]), (unsigned long long)__float2ll_rn(q[f][k] * wf * scale));
          if (pair) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[row_c * F + f]), (unsigned long long)__float2ll_rn(q[f][k] * o[k] * scale));
        }
      }
    }
  }
  __syncthreads();
  const float inv = 1.0f / scale;
  for (int i = threadIdx.x; i < entries * F; i += 1024) out[i] = (float)((double)acc[i] * (double)inv * (double)out_scale);
}
}  // namespace

// layout of the binned backward's workspace after a phase-1 call (ps_grid_scatter_binned_part / _ms_part with phase = 1), for callers
// that move the record streams themselves (the sparse gradient exchange): out[0..8] = byte offsets of {gmax_bits [K*L] u32, cursors
// [items] u32 (stream ENDS after phase 1), counts [items] u32 (upper bounds), starts [items] u32, rec_idx [n_rec_max] u32, rec_val
// [F+1][n_rec_max] f32 (plane F = ox)}, then n_rec_max, n_items, log2 of the rows per slice
extern "C" int ps_grid_scatter_layout(int L, int F, int log2T, int64_t N, int K, int64_t* out) {
  PS_REQUIRE(out != nullptr && (F == 1 || F == 2 || F == 4) && K >= 1, "ps_grid_scatter_layout: bad argument");
  const int ls = binned_log2_slice(F, log2T);
  const int n_slices = 1 << (log2T - ls);
  const int64_t n_rec_max = binned_rec_capacity(N, L, K * n_slices, 3);
  const int64_t n_items = (int64_t)K * L * n_slices;
  out[0] = 0;
  out[1] = 4096;
  out[2] = out[1] + n_items * 4;
  out[3] = out[2] + n_items * 4;
  out[4] = out[3] + ((n_items + 3) & ~(int64_t)3) * 4;
  out[5] = out[4] + n_rec_max * 4;
  out[6] = n_rec_max;
  out[7] = n_items;
  out[8] = ls;
  return 0;
}

// The owner's accumulate pass of the sparse exchange: items [item_begin, item_end) of the K-table launch, each with n_runs record runs
// (run r of local item i: records [run_starts[r * n_local + i], + run_counts[...]) of the planes rec_idx / rec_val [F+1][plane_stride]);
// gmax_bits [K*L]: the MAX over the ranks; n_points_total: points of ALL ranks (sets the fixed-point headroom, must be the same on every
// rank); the slices are WRITTEN (not added to): out = sum over all runs * out_scale.
extern "C" int ps_grid_accumulate_runs(const uint32_t* run_starts, const uint32_t* run_counts, int n_runs, const uint32_t* rec_idx,
                                       const float* rec_val, int64_t plane_stride, const uint32_t* gmax_bits, int L, int F, int log2T, int K,
                                       int64_t n_points_total, float* dtable, float* const* dtables, float out_scale, int item_begin,
                                       int item_end, void* stream) {
  PS_REQUIRE(F == 1 || F == 2 || F == 4, "ps_grid_accumulate_runs: features_per_level must be 1, 2 or 4");
  PS_REQUIRE(run_starts && run_counts && rec_idx && rec_val && gmax_bits && n_runs >= 1 && (dtable != nullptr || dtables != nullptr),
             "ps_grid_accumulate_runs: null argument");
  const int ls = binned_log2_slice(F, log2T);
  const int n_items = K * L * (1 << (log2T - ls));
  PS_REQUIRE(item_begin >= 0 && item_begin <= item_end && item_end <= n_items, "ps_grid_accumulate_runs: bad item range");
  if (item_end == item_begin) return 0;
  int headroom = 62 - 26;
  {
    int bits = 0;
    while (((int64_t)1 << bits) < n_points_total * 8) ++bits;
    if (bits > 26) headroom = 62 - bits;
  }
  const size_t lds = (size_t)(1 << ls) * F * 8;
  const int n_local = item_end - item_begin;
  hipStream_t s = (hipStream_t)stream;
#define PS_LAUNCH_RUNS(FF)                                                                                                          \
  {                                                                                                                                 \
    static bool attr_set = false;                                                                                                   \
    if (!attr_set) {                                                                                                                \
      hipFuncSetAttribute((const void*)accumulate_runs_kernel<FF>, hipFuncAttributeMaxDynamicSharedMemorySize, kAccBytes);           \
      attr_set = true;                                                                                                              \
    }                                                                                                                               \
    accumulate_runs_kernel<FF><<<(unsigned)n_local, 1024, lds, s>>>(run_starts, run_counts, n_runs, n_local, rec_idx, rec_val, plane_stride, \
                                                                   gmax_bits, L, log2T, ls, headroom, dtable, dtables, out_scale, item_begin); \
  }
  if (F == 1) PS_LAUNCH_RUNS(1)
  if (F == 2) PS_LAUNCH_RUNS(2)
  if (F == 4) PS_LAUNCH_RUNS(4)
#undef PS_LAUNCH_RUNS
  PS_CHECK_LAUNCH();
}

// bytes of scratch needed by ps_grid_scatter_binned
extern "C" int64_t ps_grid_scatter_workspace(int L, int F, int log2T, int64_t N) { return binned_workspace(L, F, log2T, N, 1); }
extern "C" int64_t ps_grid_scatter_workspace_ms(int L, int F, int log2T, int64_t n_slots, int K) { return binned_workspace(L, F, log2T, n_slots, K); }

extern "C" int ps_grid_scatter_binned(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T,
                                      int64_t N, int64_t plane_stride, float* dtable, int accumulate,
                                      const uint32_t* slice_counts, int absmax_ready, void* workspace, void* stream) {
  return scatter_binned_impl(u, dfeat, scalings, L, F, log2T, N, plane_stride, dtable, nullptr, 1, nullptr, accumulate, slice_counts,
                             absmax_ready, workspace, (hipStream_t)stream);
}

extern "C" int ps_grid_scatter_binned_ms(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T,
                                         int64_t n_slots, int64_t plane_stride, float* const* dtables, int K,
                                         const int32_t* chunk_field, const uint32_t* slice_counts, int absmax_ready, void* workspace,
                                         int dst_is_zero, void* stream) {
  PS_REQUIRE(dtables != nullptr && chunk_field != nullptr && K >= 1, "ps_grid_scatter_binned_ms: need the gradient pointers and the chunk map");
  PS_REQUIRE(n_slots % ps::kMsChunk == 0, "ps_grid_scatter_binned_ms: the sorted layout is a whole number of chunks");
  return scatter_binned_impl(u, dfeat, scalings, L, F, log2T, n_slots, plane_stride, nullptr, dtables, K, chunk_field, /*accumulate=*/dst_is_zero ? 2 : 1,
                             slice_counts, absmax_ready, workspace, (hipStream_t)stream);
}

// The same two scatters in PIECES, for a gradient that is exchanged bucket by bucket while the backward is still running
// (presight_amd/dist.py): phase 1 = prepare (count, stream offsets, record write pass), phase 2 = accumulate the items
// [item_begin, item_end) of ps_grid_scatter_items() = K * L * slices, ordered (sub-field, level, slice) like the gradient in memory;
// phase 3 with the full range = the one-call functions above.  Every accumulate call needs the same arguments as the prepare call.
extern "C" int ps_grid_scatter_items(int L, int F, int log2T, int K) { return K * L * (1 << (log2T - binned_log2_slice(F, log2T))); }

extern "C" int ps_grid_scatter_binned_part(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T,
                                           int64_t N, int64_t plane_stride, float* dtable, int accumulate,
                                           const uint32_t* slice_counts, int absmax_ready, void* workspace, int phase, int item_begin,
                                           int item_end, void* stream) {
  return scatter_binned_impl(u, dfeat, scalings, L, F, log2T, N, plane_stride, dtable, nullptr, 1, nullptr, accumulate, slice_counts,
                             absmax_ready, workspace, (hipStream_t)stream, 3, 0, 1.0f, nullptr, phase, item_begin, item_end);
}

extern "C" int ps_grid_scatter_binned_ms_part(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T,
                                              int64_t n_slots, int64_t plane_stride, float* const* dtables, int K,
                                              const int32_t* chunk_field, const uint32_t* slice_counts, int absmax_ready, void* workspace,
                                              int dst_is_zero, int phase, int item_begin, int item_end, void* stream) {
  PS_REQUIRE(dtables != nullptr && chunk_field != nullptr && K >= 1, "ps_grid_scatter_binned_ms_part: need the gradient pointers and the chunk map");
  PS_REQUIRE(n_slots % ps::kMsChunk == 0, "ps_grid_scatter_binned_ms_part: the sorted layout is a whole number of chunks");
  return scatter_binned_impl(u, dfeat, scalings, L, F, log2T, n_slots, plane_stride, nullptr, dtables, K, chunk_field, /*accumulate=*/dst_is_zero ? 2 : 1,
                             slice_counts, absmax_ready, workspace, (hipStream_t)stream, 3, 0, 1.0f, nullptr, phase, item_begin, item_end);
}

// ---- table backward + Adam in one pass (AdamFuse above).  dtable / dtables[*] point into the flat gradient buffer grad_base (they
// locate the parameters: element offset = dtable - grad_base into param_base / exp_avg_base / exp_avg_sq_base, all of one layout) and
// must hold zeros; nothing is written there.  Every slice of the items [item_begin, item_end) is updated, records or not.  One table
// (K = 1): at the host-side step count `step` >= 1.  Routed tile: sub-field k belongs to the device-decided group group_of_field[k]
// (>= 0; flags / steps as in ps_adam_step_ranges: a sub-field whose flag is down is left untouched, the step counts are advanced by
// the optimizer call that updates the group's other parameters) or, < 0, is updated at `step`.
namespace {
int make_fuse(AdamFuse& A, const float* grad_base, float* param_base, float* exp_avg_base, float* exp_avg_sq_base, float lr, float beta1,
              float beta2, float eps, float weight_decay, float grad_scale, int step, const int32_t* group_of_field,
              const int32_t* group_flags, const int32_t* group_steps) {
  PS_REQUIRE(grad_base && param_base && exp_avg_base && exp_avg_sq_base, "ps_grid_scatter_binned_adam: null flat buffer");
  PS_REQUIRE(step >= 1 || group_of_field != nullptr, "ps_grid_scatter_binned_adam: step counts from 1");
  PS_REQUIRE(group_of_field == nullptr || (group_flags != nullptr && group_steps != nullptr), "ps_grid_scatter_binned_adam: groups need flags and steps");
  const double st = step >= 1 ? (double)step : 1.0;
  A = AdamFuse{grad_base, param_base, exp_avg_base, exp_avg_sq_base, {lr, beta1, beta2, eps, weight_decay, grad_scale},
               (float)(1.0 - pow((double)beta1, st)), (float)sqrt(1.0 - pow((double)beta2, st)), group_of_field, group_flags, group_steps};
  return 0;
}
}  // namespace

extern "C" int ps_grid_scatter_binned_adam(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T, int64_t N,
                                           int64_t plane_stride, float* dtable, const uint32_t* slice_counts, int absmax_ready,
                                           void* workspace, int phase, int item_begin, int item_end, const float* grad_base,
                                           float* param_base, float* exp_avg_base, float* exp_avg_sq_base, float lr, float beta1, float beta2,
                                           float eps, float weight_decay, float grad_scale, int step, void* stream) {
  AdamFuse A;
  if (int rc = make_fuse(A, grad_base, param_base, exp_avg_base, exp_avg_sq_base, lr, beta1, beta2, eps, weight_decay, grad_scale, step,
                         nullptr, nullptr, nullptr))
    return rc;
  PS_REQUIRE(dtable != nullptr && dtable >= grad_base, "ps_grid_scatter_binned_adam: the destination must lie in the flat gradient buffer");
  PS_REQUIRE((((uintptr_t)dtable | (uintptr_t)grad_base | (uintptr_t)param_base | (uintptr_t)exp_avg_base | (uintptr_t)exp_avg_sq_base) & 15) == 0,
             "ps_grid_scatter_binned_adam: the flat buffers and the table's view must be 16-byte aligned");
  return scatter_binned_impl(u, dfeat, scalings, L, F, log2T, N, plane_stride, dtable, nullptr, 1, nullptr, /*accumulate=*/2, slice_counts,
                             absmax_ready, workspace, (hipStream_t)stream, 3, 0, 1.0f, nullptr, phase, item_begin, item_end, &A);
}

extern "C" int ps_grid_scatter_binned_ms_adam(const float* u, const float* dfeat, const float* scalings, int L, int F, int log2T,
                                              int64_t n_slots, int64_t plane_stride, float* const* dtables, int K,
                                              const int32_t* chunk_field, const uint32_t* slice_counts, int absmax_ready, void* workspace,
                                              int phase, int item_begin, int item_end, const float* grad_base, float* param_base,
                                              float* exp_avg_base, float* exp_avg_sq_base, float lr, float beta1, float beta2, float eps,
                                              float weight_decay, float grad_scale, int step, const int32_t* group_of_field,
                                              const int32_t* group_flags, const int32_t* group_steps, void* stream) {
  PS_REQUIRE(dtables != nullptr && chunk_field != nullptr && K >= 1, "ps_grid_scatter_binned_ms_adam: need the gradient pointers and the chunk map");
  PS_REQUIRE(n_slots % ps::kMsChunk == 0, "ps_grid_scatter_binned_ms_adam: the sorted layout is a whole number of chunks");
  AdamFuse A;
  if (int rc = make_fuse(A, grad_base, param_base, exp_avg_base, exp_avg_sq_base, lr, beta1, beta2, eps, weight_decay, grad_scale, step,
                         group_of_field, group_flags, group_steps))
    return rc;
  return scatter_binned_impl(u, dfeat, scalings, L, F, log2T, n_slots, plane_stride, nullptr, dtables, K, chunk_field, /*accumulate=*/2,
                             slice_counts, absmax_ready, workspace, (hipStream_t)stream, 3, 0, 1.0f, nullptr, phase, item_begin, item_end, &A);
}

// ---- 4-D grid of the dynamic field (csrc/dynamic.hip; BASELINE cfg 4): same record streams and accumulate kernel, 8 x-pairs
// per (point, level).  x [M,4]; with period > 0 up to three position sets of `period` points: set 0 takes its gradient rows from
// dfeat, the other sets from dfeat_b (null: dfeat) row m mod period; the table gradient is multiplied by out_scale.
// slice_counts (nullable): record counts per (level, slice) from ps_grid4_encode -> no counting pass.
extern "C" int64_t ps_grid4_scatter_workspace(int L, int F, int log2T, int64_t M) { return binned_workspace(L, F, log2T, M, 1, 4); }

extern "C" int ps_grid4_scatter_binned(const float* x, const float* dfeat, const float* dfeat_b, const float* scalings, int L, int F,
                                       int log2T, int64_t M, int64_t period, int64_t plane_stride, float out_scale, float* dtable,
                                       int accumulate, const uint32_t* slice_counts, void* workspace, void* stream) {
  return scatter_binned_impl(x, dfeat, scalings, L, F, log2T, M, plane_stride, dtable, nullptr, 1, nullptr, accumulate, slice_counts, 0,
                             workspace, (hipStream_t)stream, 4, period, out_scale, dfeat_b);
}
