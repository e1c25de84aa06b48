// Work distribution of the hash-grid encode kernels (3-D: encode.hip, 4-D: dynamic.hip): which (level, group of points) a
// hardware workgroup takes, dealt so that every XCD's L2 holds one level at a time.
#pragma once
#include "common.hpp"

namespace ps {

__device__ __forceinline__ void xcd_item(int64_t n_items, int64_t& item, bool& valid) {
  const int64_t b = blockIdx.x;
  const int64_t per = (n_items + 7) / 8;
  const int64_t x = b & 7, i = b >> 3;
  item = x * per + i;
  valid = (i < per) && (item < n_items);
}

// Work distribution of the encode.  A workgroup takes one group of points of ONE level; hardware workgroup b runs on XCD
// b % 8, and an XCD should work on one level at a time so that its 4 MiB L2 keeps that level's table.  Dealing the levels
// to the XCDs in order (levels 2x, 2x+1 on XCD x) leaves the XCDs with the coarse levels idle for half of the kernel: a coarse
// level's gathers hit in L1 / L2, a hashed fine level's do not (measured with the gathers of the fine levels removed: the
// 4 coarsest of 16 levels finish in 0.54 of the kernel's 1.36 ms).  So every level is cut into P parts (L * P a multiple of
// 16) and the (level, part) blocks are dealt in MIRRORED level order 0, L-1, 1, L-2, ...: every XCD gets as many coarse as
// fine blocks, still one level at a time.  Grids of few levels (L <= 8: one or two levels per XCD) are cut four times finer,
// which evens out what the mirroring leaves (measured, proposal grid L8 F1 T2^20: 0.95 -> 0.74 -> 0.67 ms per launch; the
// 16-level main grid is best with the coarse cut: 1.36 -> 0.96 ms, 0.97 with the finer one).
static inline int enc_parts(int L) {
  int low = L & -L;  // gcd(L, 16)
  if (low > 16) low = 16;
  return (L <= 8 ? 4 : 1) * 16 / low;
}
__device__ __forceinline__ void enc_item(int64_t groups, int L, int P, int& level, int64_t& group, bool& valid) {
  const int64_t gp = (groups + P - 1) / P;  // groups per block
  const int B = L * P / 8;                  // blocks per XCD (even)
  const int64_t b = blockIdx.x;
  const int x = (int)(b & 7);
  const int64_t i = b >> 3;
  const int q = x * B + (int)(i / gp);      // block in the dealt sequence
  const int part = q / L, r = q % L;
  level = (r & 1) ? L - 1 - (r >> 1) : (r >> 1);
  group = (int64_t)part * gp + i % gp;
  valid = i < (int64_t)B * gp && group < groups;
}


}  // namespace ps
