// Multi-sub-field ("MS") execution: the K sub-fields of a PreSight tile (iNGPFieldMS / PropNetDensityFieldMS / SkyFieldMS,
// ns/fields/PreSight/ingp_field_ms.py:97-126) run in ONE launch per kernel, with no host synchronisation.
//
// The points of a call are routed to their nearest centroid and stably sorted by sub-field into a PADDED layout: every
// sub-field's group starts on a chunk boundary (kMsChunk points), so that a chunk — the unit the hash-grid kernels hand to a
// workgroup — belongs to exactly one sub-field.  Three small device arrays describe the layout (written by ps_ms_route):
//     field_start[K+1]   first chunk of every sub-field's group (field_start[K] = chunks in use)
//     chunk_field[C]     sub-field of every chunk, -1 for the unused chunks behind field_start[K]
//     perm[C*kMsChunk]   sorted slot -> index of the point in the caller's order, -1 for padding slots
// Kernels keep their INTERNAL per-point arrays (u, sel, feature planes, kept activations, d(features)) in sorted order and
// reach the caller's arrays (densities, colours, semantics and their gradients, per-ray directions / appearance codes)
// through perm, so no gather / un-sort pass exists.
#pragma once
#include "common.hpp"

namespace ps {

constexpr int kMsChunk = 2048;     // points per chunk (multiple of every kernel's tile: 2048-point encode groups, 512-point bin chunks)
constexpr int kMsMaxFields = 64;

// Persistent MLP kernels: the B workgroups of a launch are dealt to the sub-fields in proportion to their chunk counts
// (every non-empty sub-field gets at least one); each workgroup loads ITS sub-field's weights once.  The same function
// tells the gradient-unpack kernel which partial blocks belong to a sub-field.
struct MsBlock {
  int k;        // sub-field (-1: this workgroup has nothing to do)
  int j, n;     // index of the workgroup among the n workgroups of its sub-field
  int64_t first_pt, end_pt;  // the sub-field's slots in the sorted layout
};

__device__ __forceinline__ void ms_field_blocks(const int* __restrict__ field_start, int K, int B, int k, int& b0, int& n) {
  const int total = field_start[K];
  int nonempty = 0;
  for (int i = 0; i < K; ++i) nonempty += (field_start[i + 1] > field_start[i]) ? 1 : 0;
  const int extra = B > nonempty ? B - nonempty : 0;
  b0 = 0;
  n = 0;
  for (int i = 0; i <= k; ++i) {
    const int c = field_start[i + 1] - field_start[i];
    const int ni = c > 0 ? 1 + (int)((int64_t)extra * c / total) : 0;
    if (i == k)
      n = ni;
    else
      b0 += ni;
  }
}

// Hardware workgroup b runs on XCD b % 8 (observed placement, used for speed only): logical block ids are dealt XCD-major, so
// that the workgroups of ONE sub-field (a contiguous range of logical ids) share an XCD and its L2 keeps that sub-field's
// weight fragments hot (K = 16 main fields: 3.5 MB of packed weights, more than fits next to the streaming activations when
// every XCD serves every sub-field).  B must be a multiple of 8.
__device__ __forceinline__ int ms_logical_block(int b, int B) { return (b & 7) * (B >> 3) + (b >> 3); }

__device__ __forceinline__ MsBlock ms_block(const int* __restrict__ field_start, int K, int B, int b) {
  const int total = field_start[K];
  int nonempty = 0;
  for (int i = 0; i < K; ++i) nonempty += (field_start[i + 1] > field_start[i]) ? 1 : 0;
  const int extra = B > nonempty ? B - nonempty : 0;
  int b0 = 0;
  for (int k = 0; k < K; ++k) {
    const int c = field_start[k + 1] - field_start[k];
    if (c == 0) continue;
    const int n = 1 + (int)((int64_t)extra * c / total);
    if (b < b0 + n) return MsBlock{k, b - b0, n, (int64_t)field_start[k] * kMsChunk, (int64_t)field_start[k + 1] * kMsChunk};
    b0 += n;
  }
  return MsBlock{-1, 0, 0, 0, 0};
}

// index of sorted slot p in the CALLER's per-point arrays, -1 = no such point (single-field launches work in the caller's order)
template <bool MS>
__device__ __forceinline__ int64_t ms_orig_index(const int* __restrict__ perm, int64_t p, int64_t N) {
  if constexpr (MS)
    return p < N ? (int64_t)perm[p] : (int64_t)-1;
  else
    return p < N ? p : (int64_t)-1;
}

}  // namespace ps
