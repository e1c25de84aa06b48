// Shared device helpers for the gfx950 (CDNA4) kernels of libpresight_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PS_WAVE 64

// last error text, readable through ps_last_error()
void ps_set_error(const char* msg);

#define PS_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) {                               \
      ps_set_error(hipGetErrorString(e__));                \
      return (int)e__;                                     \
    }                                                      \
    return 0;                                              \
  } while (0)

#define PS_REQUIRE(cond, msg) \
  do {                        \
    if (!(cond)) {            \
      ps_set_error(msg);      \
      return -1;              \
    }                         \
  } while (0)

static inline int ps_grid_for(int64_t work_items, int block, int max_blocks = 256 * 16) {
  int64_t b = (work_items + block - 1) / block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

__device__ __forceinline__ int ps_lane() { return threadIdx.x & 63; }

// exact-f32 matrix core op: D(16x16) += A(16x4) * B(4x16); A: lane l holds A[l&15][l>>4],
// B: lane l holds B[l>>4][l&15]; D: lane l, reg r holds D[4*(l>>4)+r][l&15].
__device__ __forceinline__ f32x4 ps_mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// sum over the 16 lanes that share (lane >> 4); result valid in every lane of the row
// sum over the 16 lanes of a row (lanes 16k..16k+15), result in every lane.  DPP modifiers on the vector ALU
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror): __shfl_xor lowers to ds_bpermute_b32, i.e. four trips
// through the LDS crossbar per value -- the fused MLP backward does ~500 of these reductions per 32 points.
__device__ __forceinline__ float ps_dpp_add(float v, const int ctrl_tag) {
  // ctrl must be a compile-time constant for the builtin: dispatch on a small tag
  int x = __builtin_bit_cast(int, v), y;
  switch (ctrl_tag) {
    case 0: y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false); break;   // quad_perm:[1,0,3,2]
    case 1: y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false); break;   // quad_perm:[2,3,0,1]
    case 2: y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false); break;  // row_half_mirror
    default: y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false); break; // row_mirror
  }
  return v + __builtin_bit_cast(float, y);
}
// v of the lane `N` places to the left in the same row of 16 lanes (0 where there is none): DPP row_shr
template <int N>
__device__ __forceinline__ float ps_row_shr(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xF, 0xF, true));
}
// inclusive prefix sum within each row of 16 lanes (four DPP adds, no LDS crossbar)
__device__ __forceinline__ float ps_row16_incl_scan(float v) {
  v += ps_row_shr<1>(v);
  v += ps_row_shr<2>(v);
  v += ps_row_shr<4>(v);
  v += ps_row_shr<8>(v);
  return v;
}
__device__ __forceinline__ float ps_row16_sum(float v) {
  v = ps_dpp_add(v, 0);
  v = ps_dpp_add(v, 1);
  v = ps_dpp_add(v, 2);
  v = ps_dpp_add(v, 3);
  return v;
}

__device__ __forceinline__ float ps_wave_sum(float v) {
  v = ps_row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// inclusive prefix sum across the 64 lanes of a wavefront
__device__ __forceinline__ float ps_wave_incl_scan(float v) {
  const int l = ps_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(v, d, 64);
    if (l >= d) v += o;
  }
  return v;
}

// instant-NGP spatial hash; low log2(T) bits equal the reference's int64 arithmetic
// (ns/field_components/encodings.py:324-341) because xor/multiply commute with truncation mod 2^32.
__device__ __forceinline__ uint32_t ps_hash3(int x, int y, int z, uint32_t mask) {
  return (((uint32_t)x) ^ ((uint32_t)y * 2654435761u) ^ ((uint32_t)z * 805459861u)) & mask;
}
