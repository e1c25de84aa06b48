// Operator-level fused MLP (MLP.forward / backward of ns/field_components/mlp.py:155-174) on the
// exact-fp32 matrix cores, plus the weight pack / gradient unpack kernels shared with the fused
// field kernels.  See mlp_core.hpp for the register/fragment layout.
#include "common.hpp"
#include "mlp_core.hpp"
#include "ms_core.hpp"

namespace {

using namespace ps;

// ------------------------------------------------------------------------------------------
// pack: torch layout W[out][in], b[out]  ->  forward fragments + transposed fragments
// colmap[t*4+g] = torch input column supplied by lane group g at k-step t (-1 = zero pad)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_layer(const float* __restrict__ W, const float* __restrict__ b, int out_dim, int in_dim,
                                           const int* __restrict__ colmap, int KS, int NB, float* __restrict__ fw_block,
                                           float* __restrict__ wt_block) {
  const int IB = (KS + 3) / 4;
  const int n_bias = NB * 16, n_wf = NB * IB * 256, n_wt = IB * NB * 256;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_bias + n_wf + n_wt; i += gridDim.x * blockDim.x) {
    if (i < n_bias) {
      fw_block[i] = (i < out_dim) ? b[i] : 0.0f;
    } else if (i < n_bias + n_wf) {
      // wf[nb][q][lane][r]: A-operand fragment of k-step t = 4q+r (zero beyond KS)
      const int e = i - n_bias;
      const int r = e & 3, lane = (e >> 2) & 63, q = (e >> 8) % IB, nb = (e >> 8) / IB;
      const int t = 4 * q + r;
      const int o = 16 * nb + (lane & 15);
      const int col = (t < KS) ? colmap[t * 4 + (lane >> 4)] : -1;
      fw_block[i] = (o < out_dim && col >= 0 && col < in_dim) ? W[(size_t)o * in_dim + col] : 0.0f;
    } else {
      // wtf[ib][q][lane][r]: transposed fragment of k-step t = 4q+r over the outputs
      const int e = i - n_bias - n_wf;
      const int r = e & 3, lane = (e >> 2) & 63, q = (e >> 8) % NB, ib = (e >> 8) / NB;
      const int o = 16 * q + 4 * (lane >> 4) + r;
      const int row = lane & 15;
      const int tin = 4 * ib + (row & 3);
      const int col = (tin < KS) ? colmap[tin * 4 + (row >> 2)] : -1;
      wt_block[e] = (o < out_dim && col >= 0 && col < in_dim) ? W[(size_t)o * in_dim + col] : 0.0f;
    }
  }
}

__global__ void mlp_pack_layer_kernel(const float* __restrict__ W, const float* __restrict__ b, int out_dim, int in_dim,
                                      const int* __restrict__ colmap, int KS, int NB, float* __restrict__ fw_block,
                                      float* __restrict__ wt_block) {
  pack_layer(W, b, out_dim, in_dim, colmap, KS, NB, fw_block, wt_block);
}

// batched variants: all layers of one fused stack in ONE launch (blockIdx.y / blockIdx.z = layer); the per-layer
// descriptors travel by value in the kernel arguments
constexpr int kMaxBatchedLayers = 8;
struct LayerDesc {
  const float* W;     // pack: torch weight [out,in]       unpack: partial blocks of this layer
  const float* b;     // pack: torch bias [out]            unpack: unused
  const int* colmap;
  float* dst0;        // pack: forward fragments           unpack: dW [out,in] (accumulated)
  float* dst1;        // pack: transposed fragments        unpack: db [out]    (accumulated)
  int out_dim, in_dim, KS, NB;
};
struct LayerBatch {
  LayerDesc l[kMaxBatchedLayers];
};

__global__ void mlp_pack_layers_kernel(LayerBatch a) {
  const LayerDesc& d = a.l[blockIdx.y];
  pack_layer(d.W, d.b, d.out_dim, d.in_dim, d.colmap, d.KS, d.NB, d.dst0, d.dst1);
}

// unpack: sum the per-workgroup partial gradient blocks and add into torch-layout grads.
// grid = (elements/256, part chunks): each thread sums a chunk of <= kUnpackChunk partials (coalesced across the
// 256 threads of a block) and adds it to the destination; with one chunk the result is a plain deterministic sum.
constexpr int kUnpackChunk = 32;
__device__ __forceinline__ void unpack_layer(const float* __restrict__ gpart, int n_parts, int64_t part_stride, int out_dim,
                                             int in_dim, const int* __restrict__ colmap, int KS, int NB, float* __restrict__ gW,
                                             float* __restrict__ gb, int chunk, int n_chunks) {
  const int IB = (KS + 3) / 4;
  const int n_w = NB * IB * 256, n_b = NB * 16;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_w + n_b) return;
  float* dst = nullptr;
  if (i < n_w) {
    const int r = i & 3, lane = (i >> 2) & 63, ib = (i >> 8) % IB, ob = (i >> 8) / IB;  // [tile][lane][r]
    const int o = 16 * ob + 4 * (lane >> 4) + r;
    const int row = lane & 15;
    const int tin = 4 * ib + (row & 3);
    const int col = (tin < KS) ? colmap[tin * 4 + (row >> 2)] : -1;
    if (o < out_dim && col >= 0 && col < in_dim) dst = gW + (size_t)o * in_dim + col;
  } else {
    const int o = i - n_w;
    if (o < out_dim) dst = gb + o;
  }
  if (dst == nullptr) return;
  const int p0 = chunk * kUnpackChunk, p1 = min(n_parts, p0 + kUnpackChunk);
  float s = 0.f;
  for (int p = p0; p < p1; ++p) s += gpart[(size_t)p * part_stride + i];
  if (n_chunks == 1)
    *dst += s;
  else
    unsafeAtomicAdd(dst, s);
}

__global__ void mlp_unpack_grad_layer_kernel(const float* __restrict__ gpart, int n_parts, int64_t part_stride, int out_dim,
                                             int in_dim, const int* __restrict__ colmap, int KS, int NB,
                                             float* __restrict__ gW, float* __restrict__ gb) {
  unpack_layer(gpart, n_parts, part_stride, out_dim, in_dim, colmap, KS, NB, gW, gb, blockIdx.y, gridDim.y);
}

__global__ void mlp_unpack_grad_layers_kernel(LayerBatch a, int n_parts, int64_t part_stride) {
  const LayerDesc& d = a.l[blockIdx.z];
  unpack_layer(d.W, n_parts, part_stride, d.out_dim, d.in_dim, d.colmap, d.KS, d.NB, d.dst0, d.dst1, blockIdx.y, gridDim.y);
}

// Descriptor TABLE variants for multi-sub-field launches (K x layers descriptors, too many for kernel arguments): the table
// lives in device memory; it is built once per model by the host (the parameters sit at fixed addresses in the optimizer's
// flat buffers) and re-used every step.
__global__ void mlp_pack_table_kernel(const LayerDesc* __restrict__ table) {
  const LayerDesc d = table[blockIdx.y];
  pack_layer(d.W, d.b, d.out_dim, d.in_dim, d.colmap, d.KS, d.NB, d.dst0, d.dst1);
}

// descriptor i belongs to sub-field i / layers_per_field; its partial blocks are those of the workgroups ms_field_blocks deals
// to that sub-field (parts_per_block partial blocks per workgroup).  blockIdx.y = chunk of kUnpackChunk partials.
__global__ void mlp_unpack_table_ms_kernel(const LayerDesc* __restrict__ table, int layers_per_field, const int* __restrict__ field_start,
                                           int K, int B, int parts_per_block, int64_t part_stride) {
  const LayerDesc d = table[blockIdx.z];
  const int k = blockIdx.z / layers_per_field;
  int b0, n;
  ps::ms_field_blocks(field_start, K, B, k, b0, n);
  const int lo = b0 * parts_per_block, hi = (b0 + n) * parts_per_block;
  const int p0 = max(lo, (int)blockIdx.y * kUnpackChunk), p1 = min(hi, ((int)blockIdx.y + 1) * kUnpackChunk);
  if (p0 >= p1) return;
  const bool single = (lo / kUnpackChunk) == ((hi - 1) / kUnpackChunk);
  const int IB = (d.KS + 3) / 4;
  const int n_w = d.NB * IB * 256, n_b = d.NB * 16;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_w + n_b) return;
  float* dst = nullptr;
  if (i < n_w) {
    const int r = i & 3, lane = (i >> 2) & 63, ib = (i >> 8) % IB, ob = (i >> 8) / IB;  // [tile][lane][r]
    const int o = 16 * ob + 4 * (lane >> 4) + r;
    const int row = lane & 15;
    const int tin = 4 * ib + (row & 3);
    const int col = (tin < d.KS) ? d.colmap[tin * 4 + (row >> 2)] : -1;
    if (o < d.out_dim && col >= 0 && col < d.in_dim) dst = d.dst0 + (size_t)o * d.in_dim + col;
  } else {
    const int o = i - n_w;
    if (o < d.out_dim) dst = d.dst1 + o;
  }
  if (dst == nullptr) return;
  float s = 0.f;
  for (int p = p0; p < p1; ++p) s += d.W[(size_t)p * part_stride + i];
  if (single)
    *dst += s;
  else
    unsafeAtomicAdd(dst, s);
}

// ------------------------------------------------------------------------------------------
// operator-level forward / backward
// ------------------------------------------------------------------------------------------
constexpr int ACT_NONE = 0, ACT_SIGMOID = 1;

template <class M, int PB>
__device__ __forceinline__ void load_rows_linear(const float* __restrict__ x, int64_t first, int64_t N, int dim,
                                                 float (&v)[PB][M::KS0]) {
  const int lane = ps_lane(), j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int64_t p = first + pb * 16 + j;
#pragma unroll
    for (int t = 0; t < M::KS0; ++t) {
      const int col = 4 * t + g;
      v[pb][t] = (p < N && col < dim) ? x[p * dim + col] : 0.0f;
    }
  }
}

template <class M, int PB, int ACT>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ packed,
                                                      float* __restrict__ y, int64_t N, int in_dim, int out_dim) {
  __shared__ __attribute__((aligned(16))) float lds[M::FW];
  for (int i = threadIdx.x * 4; i < M::FW; i += 256 * 4)
    *reinterpret_cast<f32x4*>(lds + i) = *reinterpret_cast<const f32x4*>(packed + i);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = ps_lane(), j = lane & 15, g = lane >> 4;
  const int64_t tiles = (N + 16 * PB - 1) / (16 * PB);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t first = tile * 16 * PB;
    float xin[PB][M::KS0], h1[PB][M::HB * 4], h2[PB][M::HB * 4], z[PB][M::NBO * 4];
    load_rows_linear<M, PB>(x, first, N, in_dim, xin);
    mlp_forward<M, PB>(LdsW{lds}, xin, h1, h2, z);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t p = first + pb * 16 + j;
#pragma unroll
      for (int t = 0; t < M::NBO * 4; ++t) {
        const int col = 16 * (t >> 2) + 4 * g + (t & 3);
        float v = z[pb][t];
        if (ACT == ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
        if (p < N && col < out_dim) y[p * out_dim + col] = v;
      }
    }
  }
}

template <class M, int PB, int ACT, bool WANT_DX>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ packed, float* __restrict__ dx,
                                                      float* __restrict__ gpart, int64_t N, int in_dim, int out_dim) {
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
  const int64_t tiles = (N + 16 * PB - 1) / (16 * PB);
  // workgroup-uniform trip count (the dW flush contains workgroup barriers); out-of-range tiles are fully masked
  for (int64_t base = (int64_t)blockIdx.x * 4; base < tiles; base += (int64_t)gridDim.x * 4) {
    const int64_t first = (base + wave) * 16 * PB;
    float xin[PB][M::KS0], h1[PB][M::HB * 4], h2[PB][M::HB * 4], z[PB][M::NBO * 4];
    load_rows_linear<M, PB>(x, first, N, in_dim, xin);
    mlp_forward<M, PB>(gw, xin, h1, h2, z);
    // dz = dy (* sigmoid') in D layout
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t p = first + pb * 16 + j;
#pragma unroll
      for (int t = 0; t < M::NBO * 4; ++t) {
        const int col = 16 * (t >> 2) + 4 * g + (t & 3);
        float d = (p < N && col < out_dim) ? dy[p * out_dim + col] : 0.0f;
        if (ACT == ACT_SIGMOID) {
          const float s = 1.0f / (1.0f + expf(-z[pb][t]));
          d *= s * (1.0f - s);
        }
        z[pb][t] = d;
      }
    }
    float dxin[PB][M::L0::IB * 4];
    mlp_backward<M, PB, WANT_DX>(gw, scratch, gacc, locks, xin, h1, h2, z, dxin);
    if constexpr (WANT_DX) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int64_t p = first + pb * 16 + j;
#pragma unroll
        for (int t = 0; t < M::KS0; ++t) {
          const int col = 4 * t + g;
          if (p < N && col < in_dim) dx[p * in_dim + col] = dxin[pb][t];
        }
      }
    }
  }
  __syncthreads();
  float* out = gpart + (size_t)blockIdx.x * M::GPACKED;
  for (int i = threadIdx.x; i < M::GPACKED; i += 256) out[i] = gacc[i];
}

struct MlpShape {
  int ks0, hb, nbo, nl;
};

template <class M>
int launch_fwd(const float* x, const float* packed, float* y, int64_t N, int in_dim, int out_dim, int act, hipStream_t s) {
  constexpr int PB = 4;
  const int64_t tiles = (N + 16 * PB - 1) / (16 * PB);
  int grid = (int)((tiles + 3) / 4);
  if (grid > 256 * 2) grid = 256 * 2;
  if (grid < 1) grid = 1;
  if (act == ACT_SIGMOID)
    mlp_fwd_kernel<M, PB, ACT_SIGMOID><<<grid, 256, 0, s>>>(x, packed, y, N, in_dim, out_dim);
  else
    mlp_fwd_kernel<M, PB, ACT_NONE><<<grid, 256, 0, s>>>(x, packed, y, N, in_dim, out_dim);
  PS_CHECK_LAUNCH();
}

template <class M>
int bwd_grid(int64_t N) {
  constexpr int PB = 2;
  const int64_t tiles = (N + 16 * PB - 1) / (16 * PB);
  int grid = (int)((tiles + 3) / 4);
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  return grid;
}

template <class M>
int launch_bwd(const float* x, const float* dy, const float* packed, float* dx, float* gpart, int64_t N, int in_dim,
               int out_dim, int act, hipStream_t s) {
  constexpr int PB = 2;
  const int grid = bwd_grid<M>(N);
  if (dx != nullptr) {
    if (act == ACT_SIGMOID)
      mlp_bwd_kernel<M, PB, ACT_SIGMOID, true><<<grid, 256, 0, s>>>(x, dy, packed, dx, gpart, N, in_dim, out_dim);
    else
      mlp_bwd_kernel<M, PB, ACT_NONE, true><<<grid, 256, 0, s>>>(x, dy, packed, dx, gpart, N, in_dim, out_dim);
  } else {
    if (act == ACT_SIGMOID)
      mlp_bwd_kernel<M, PB, ACT_SIGMOID, false><<<grid, 256, 0, s>>>(x, dy, packed, dx, gpart, N, in_dim, out_dim);
    else
      mlp_bwd_kernel<M, PB, ACT_NONE, false><<<grid, 256, 0, s>>>(x, dy, packed, dx, gpart, N, in_dim, out_dim);
  }
  PS_CHECK_LAUNCH();
}

// shapes used by the PreSight fields + the cfg-1 tiny variants
#define PS_MLP_SHAPES(X) \
  X(8, 4, 5, 2)          \
  X(10, 4, 5, 2)         \
  X(1, 2, 5, 2)          \
  X(16, 4, 4, 3)         \
  X(12, 4, 1, 3)         \
  X(12, 2, 1, 3)         \
  X(2, 4, 1, 2)          \
  X(1, 2, 1, 2)          \
  X(8, 2, 1, 3)          \
  X(4, 2, 4, 3)

}  // namespace

extern "C" int ps_mlp_shape_supported(int in_dim, int hidden, int out_dim, int num_layers) {
  const int ks0 = (in_dim + 3) / 4, hb = hidden / 16, nbo = (out_dim + 15) / 16;
  if (hidden % 16 != 0) return 0;
#define X(a, b, c, d) \
  if (ks0 == a && hb == b && nbo == c && num_layers == d) return 1;
  PS_MLP_SHAPES(X)
#undef X
  return 0;
}

// sizes (in floats) of the packed parameter block, of one partial gradient block, and the number of
// partial blocks ps_mlp_bwd writes for N rows
extern "C" int ps_mlp_sizes(int in_dim, int hidden, int out_dim, int num_layers, int64_t N, int64_t* packed_floats,
                            int64_t* grad_floats, int* n_parts) {
  const int ks0 = (in_dim + 3) / 4, hb = hidden / 16, nbo = (out_dim + 15) / 16;
#define X(a, b, c, d)                                               \
  if (ks0 == a && hb == b && nbo == c && num_layers == d) {         \
    using M = ps::MlpT<a, b, c, d>;                                 \
    *packed_floats = M::PACKED;                                     \
    *grad_floats = M::GPACKED;                                      \
    *n_parts = bwd_grid<M>(N);                                      \
    return 0;                                                       \
  }
  PS_MLP_SHAPES(X)
#undef X
  ps_set_error("ps_mlp_sizes: unsupported MLP shape");
  return -2;
}

extern "C" int ps_mlp_pack_layer(const float* W, const float* b, int out_dim, int in_dim, const int* colmap, int KS, int NB,
                                 float* fw_block, float* wt_block, void* stream) {
  const int IB = (KS + 3) / 4;
  const int total = NB * 16 + 2 * NB * IB * 256;
  mlp_pack_layer_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(W, b, out_dim, in_dim, colmap, KS, NB, fw_block,
                                                                              wt_block);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_mlp_unpack_grad_layer(const float* gpart, int n_parts, int64_t part_stride, int out_dim, int in_dim,
                                        const int* colmap, int KS, int NB, float* gW, float* gb, void* stream) {
  const int IB = (KS + 3) / 4;
  const int total = NB * IB * 256 + NB * 16;
  dim3 grid((total + 255) / 256, (n_parts + kUnpackChunk - 1) / kUnpackChunk);
  mlp_unpack_grad_layer_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(gpart, n_parts, part_stride, out_dim, in_dim, colmap, KS,
                                                                     NB, gW, gb);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_mlp_pack_layers(int n_layers, const float* const* W, const float* const* b, const int* out_dim,
                                  const int* in_dim, const int* const* colmap, const int* KS, const int* NB,
                                  float* const* fw_block, float* const* wt_block, void* stream) {
  PS_REQUIRE(n_layers >= 1 && n_layers <= kMaxBatchedLayers, "ps_mlp_pack_layers: 1..8 layers per call");
  LayerBatch a{};
  int total = 0;
  for (int i = 0; i < n_layers; ++i) {
    a.l[i] = LayerDesc{W[i], b[i], colmap[i], fw_block[i], wt_block[i], out_dim[i], in_dim[i], KS[i], NB[i]};
    const int IB = (KS[i] + 3) / 4;
    total = max(total, NB[i] * 16 + 2 * NB[i] * IB * 256);
  }
  mlp_pack_layers_kernel<<<dim3((total + 255) / 256, n_layers), 256, 0, (hipStream_t)stream>>>(a);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_mlp_unpack_grad_layers(int n_layers, const float* const* gpart, int n_parts, int64_t part_stride,
                                         const int* out_dim, const int* in_dim, const int* const* colmap, const int* KS,
                                         const int* NB, float* const* gW, float* const* gb, void* stream) {
  PS_REQUIRE(n_layers >= 1 && n_layers <= kMaxBatchedLayers, "ps_mlp_unpack_grad_layers: 1..8 layers per call");
  LayerBatch a{};
  int total = 0;
  for (int i = 0; i < n_layers; ++i) {
    a.l[i] = LayerDesc{gpart[i], nullptr, colmap[i], gW[i], gb[i], out_dim[i], in_dim[i], KS[i], NB[i]};
    total = max(total, NB[i] * ((KS[i] + 3) / 4) * 256 + NB[i] * 16);
  }
  dim3 grid((total + 255) / 256, (n_parts + kUnpackChunk - 1) / kUnpackChunk, n_layers);
  mlp_unpack_grad_layers_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a, n_parts, part_stride);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_mlp_fwd(const float* x, const float* packed, float* y, int64_t N, int in_dim, int hidden, int out_dim,
                          int num_layers, int out_act, void* stream) {
  if (N == 0) return 0;
  const int ks0 = (in_dim + 3) / 4, hb = hidden / 16, nbo = (out_dim + 15) / 16;
#define X(a, b, c, d)                                       \
  if (ks0 == a && hb == b && nbo == c && num_layers == d)   \
    return launch_fwd<ps::MlpT<a, b, c, d>>(x, packed, y, N, in_dim, out_dim, out_act, (hipStream_t)stream);
  PS_MLP_SHAPES(X)
#undef X
  ps_set_error("ps_mlp_fwd: unsupported MLP shape");
  return -2;
}

extern "C" int ps_mlp_bwd(const float* x, const float* dy, const float* packed, float* dx, float* gpart, int64_t N,
                          int in_dim, int hidden, int out_dim, int num_layers, int out_act, void* stream) {
  if (N == 0) return 0;
  const int ks0 = (in_dim + 3) / 4, hb = hidden / 16, nbo = (out_dim + 15) / 16;
#define X(a, b, c, d)                                       \
  if (ks0 == a && hb == b && nbo == c && num_layers == d)   \
    return launch_bwd<ps::MlpT<a, b, c, d>>(x, dy, packed, dx, gpart, N, in_dim, out_dim, out_act, (hipStream_t)stream);
  PS_MLP_SHAPES(X)
#undef X
  ps_set_error("ps_mlp_bwd: unsupported MLP shape");
  return -2;
}

extern "C" int ps_mlp_layer_desc_bytes(void) { return (int)sizeof(LayerDesc); }

extern "C" int ps_mlp_pack_table(const void* table, int n_layers, int max_elems, void* stream) {
  if (n_layers == 0) return 0;
  mlp_pack_table_kernel<<<dim3((max_elems + 255) / 256, n_layers), 256, 0, (hipStream_t)stream>>>((const LayerDesc*)table);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_mlp_unpack_table_ms(const void* table, int n_layers, int layers_per_field, const int32_t* field_start, int K, int B,
                                      int parts_per_block, int64_t part_stride, int max_elems, void* stream) {
  if (n_layers == 0) return 0;
  PS_REQUIRE(n_layers == K * layers_per_field, "ps_mlp_unpack_table_ms: one descriptor per (sub-field, layer)");
  dim3 grid((max_elems + 255) / 256, (B * parts_per_block + kUnpackChunk - 1) / kUnpackChunk, n_layers);
  mlp_unpack_table_ms_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const LayerDesc*)table, layers_per_field, field_start, K, B,
                                                                    parts_per_block, part_stride);
  PS_CHECK_LAUNCH();
}
