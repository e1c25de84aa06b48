// Fused field kernels: everything between the hash-grid features and the per-sample outputs of a
// PreSight field runs in ONE kernel per direction, activations never leave registers.
//
//   proposal field  (ns/fields/PreSight/prop_density_field.py:129-153)
//       feat[L*F] -> Linear(64) ReLU Linear(1) -> trunc_exp * selector
//   main field      (ns/fields/PreSight/ingp_field.py:168-237)
//       feat[L*F] -> Linear(64) ReLU Linear(80) = [sigma_raw | geo15 | sem_embed64]
//       sigma      = trunc_exp(sigma_raw) * selector
//       semantics  = Linear(64) ReLU Linear(64) ReLU Linear(64) (sem_embed)
//       rgb        = sigmoid(Linear(64) ReLU Linear(64) ReLU Linear(3) ([SH16(d) | geo15 | app]))
//
// The 80 outputs of the base MLP sit in MFMA D registers; blocks 1..4 are consumed as the B operand of
// the semantic head, block 0 (sigma_raw + geo15) is spliced between the SH and appearance k-steps of
// the colour head by giving that head's first layer a custom column map (host side, fields.py).
// Backward recomputes the forward from the features (no activation is stored), back-propagates through
// the three MLPs with the LDS-transposed dW scheme of mlp_core.hpp and emits d(features) as level planes
// for ps_grid_scatter.
#include "common.hpp"
#include "field_io.hpp"
// the only one-point-block forward of this file is the factored training node in its two-workgroups-per-CU shape (MainFwdShape):
// its layers keep ONE accumulator chain per output block, i.e. the summation order of the two-block kernels (same bits)
#define PS_FWD_PB1_SPLIT 0
#include "mlp_core.hpp"
#include "ms_core.hpp"
#include "pointwise_core.hpp"

#ifndef PS_MAIN_BWD_PB
#define PS_MAIN_BWD_PB 2     // point blocks (of 16) per wave and iteration of the main backward
#define PS_MAIN_BWD_WAVES 4  // waves per workgroup (they share the LDS weight-gradient accumulators)
#endif

#if defined(PS_TIMING)
// profiling build (tools/build_variant.sh timing -DPS_TIMING): shader-clock time of the phases of one main-backward tile, summed
// over all waves; read back with ps_debug_timing
__device__ unsigned long long g_ps_timing[16];
__device__ unsigned long long g_ps_timing_fwd[16];
__device__ unsigned long long g_ps_timing_rgb[16];
#define PS_TSTAMP(i)                                              \
  {                                                               \
    const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
    tacc[i] += now__ - tlast;                                     \
    tlast = now__;                                                \
  }
#else
#define PS_TSTAMP(i)
#endif

namespace {

using namespace ps;

// per-level max |d(feature)| for the fixed-point scale of the binned table backward (encode.hip), kept per lane over all
// tiles of the kernel and published once per wave: saves that backward a separate pass over d(features)
template <int KS0, int PB>
__device__ __forceinline__ void track_absmax(const float (&dx)[PB][((KS0 + 3) / 4) * 4], float (&mx)[KS0]) {
#pragma unroll
  for (int pb = 0; pb < PB; ++pb)
#pragma unroll
    for (int t = 0; t < KS0; ++t) {
      const float a = fabsf(dx[pb][t]);
      mx[t] = (a == a) ? fmaxf(mx[t], a) : __builtin_inff();  // fmaxf would drop a NaN: keep it visible as "non-finite"
    }
}
template <int KS0>
__device__ __forceinline__ void publish_absmax(const float (&mx)[KS0], int LF, int F, unsigned* __restrict__ level_absmax) {
  const int lane = ps_lane(), j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < KS0; ++t) {
    float m = mx[t];
    m = fmaxf(m, __shfl_xor(m, 8, 64));
    m = fmaxf(m, __shfl_xor(m, 4, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    const int col = 4 * t + g;
    // non-negative floats order like uints; NaN / inf publish the NaN pattern (see absmax_kernel in encode.hip)
    if (j == 0 && col < LF && !(m <= 0.0f)) atomicMax(level_absmax + col / F, isfinite(m) ? __float_as_uint(m) : 0x7fc00000u);
  }
}

// Ablation build only (tools/build_variant.sh l2acts -DPS_ABL_L2ACTS; the product library never defines it): every kept-activation
// (and dzb workspace) access of a wave goes to ONE 16-point block per wave of the first 256 workgroups -- 1024 blocks, 2.9 MB per XCD,
// resident in its L2 -- instead of the block's own rows: wrong numbers, right instruction stream.  The difference to the product
// is what the HBM round trip of the kept activations costs each of the four main-MLP kernels (EXPERIMENTS.md, round 6).
#if defined(PS_ABL_L2ACTS)
#define PS_ACT_BLK(blk) ((int64_t)((blockIdx.x & 255) * 4 + ((threadIdx.x >> 6) & 3)))
#else
#define PS_ACT_BLK(blk) (blk)
#endif

// Kept activations, in REGISTER order: for every 16-point block the buffer holds ACT_W/16 "neuron blocks" of [64 lanes][4]
// floats (1 KiB each) -- exactly the D registers of the block -- so that the forward's stores and the backward's loads are
// fully coalesced 16-byte-per-lane accesses (a torch-order [N, width] layout made the forward write 64-byte pieces at a
// 1.6 KB stride and cost it 1.4 ms).  col0 = first neuron column of the activation (multiple of 16), stride = ACT_W.
template <int NBLK, int PB>
__device__ __forceinline__ void store_act(float* __restrict__ acts, int stride, int col0, int64_t first, int64_t N,
                                          const float (&v)[PB][NBLK * 4]) {
  const int lane = ps_lane();
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int64_t blk = first / 16 + pb;  // 16-point block index
    if (blk * 16 < N) {
      float* base = acts + PS_ACT_BLK(blk) * (int64_t)stride * 16 + (int64_t)col0 * 16 + lane * 4;
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb)
        *reinterpret_cast<f32x4*>(base + nb * 256) = (f32x4){v[pb][4 * nb], v[pb][4 * nb + 1], v[pb][4 * nb + 2], v[pb][4 * nb + 3]};
    }
  }
}
// CLAMP: the loads are unconditional, blocks past the end read the last block instead (callers that only ever USE whole in-range
// tiles -- the factored kernels, and the routed (MS) kernels, whose per-point working arrays live in the chunk-padded sorted layout: a conditional load is a branch plus a merge with a constant, and such merges of values still in flight made the
// compiler place s_waitcnt vmcnt(0) right behind the load or at the loop's back edge)
template <int NBLK, int PB, bool CLAMP = false>
__device__ __forceinline__ void load_act(const float* __restrict__ acts, int stride, int col0, int64_t first, int64_t N,
                                         float (&v)[PB][NBLK * 4]) {
  const int lane = ps_lane();
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    int64_t blk = first / 16 + pb;
    if constexpr (CLAMP) blk = blk * 16 < N ? blk : (N - 1) / 16;
    const float* base = acts + PS_ACT_BLK(blk) * (int64_t)stride * 16 + (int64_t)col0 * 16 + lane * 4;
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
      f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (CLAMP || blk * 16 < N) t = *reinterpret_cast<const f32x4*>(base + nb * 256);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[pb][4 * nb + r] = t[r];
    }
  }
}

// ray of point n (n / S): point counts are < 2^31 (checked by the launchers), and a 32-bit division is a fifth of the 64-bit one
__device__ __forceinline__ int64_t ray_index(int64_t n, int S) { return (int64_t)((uint32_t)n / (uint32_t)S); }

__device__ __forceinline__ float trunc_exp_grad(float raw) { return expf(fminf(fmaxf(raw, -15.0f), 15.0f)); }

// index of sorted slot p in the CALLER's per-point arrays (densities, colours, semantics, their gradients), -1 = no such point.
// Single-field launches work in the caller's order; multi-sub-field launches (ms_core.hpp) go through perm.
template <bool MS>
__device__ __forceinline__ int64_t orig_index(const int* __restrict__ perm, int64_t p, int64_t N) {
  if constexpr (MS)
    return p < N ? (int64_t)perm[p] : (int64_t)-1;
  else
    return p < N ? p : (int64_t)-1;
}

// multi-sub-field launch: which sub-field, which tiles (see ms_core.hpp); single-field: all tiles, strided over the grid
struct TileRange {
  int64_t first_pt, end_pt;
  int j, n;
};

// ------------------------------------------------------------------------------------------ proposal field
// The proposal MLP ends in ONE output neuron.  As a 16x16 MFMA tile that layer would be 15/16 padding and, with its
// data-backward and weight-gradient tiles, 48 of the 88 matrix ops per 16 points; it runs on the vector ALU instead.
// In the D-register layout lane (j, g) already holds the 16 hidden neurons {16*nb + 4*g + r} of point j, so
//     z_j   = b + sum over the 4 lane groups of  sum_t wz_g[t] * h_g[t]        (16 FMAs + 2 cross-lane adds)
//     dh[t] = wz_g[t] * dz_j,      dWz_g[t] += h_g[t] * dz_j                   (no cross-lane traffic until the end)
// wz_g[t] is element (lane 16*g) of the packed forward fragment t of that layer (row 0 of the A operand; fragments are
// stored [t/4][lane][t%4]).
template <class M, class W>
__device__ __forceinline__ void load_scalar_head(const W& pz, float (&wz)[M::HB * 4], float& bz) {
  using LZ = typename M::LZ;
  const unsigned g = (unsigned)ps_lane() >> 4;
#pragma unroll
  for (int t = 0; t < M::HB * 4; ++t) wz[t] = pz.elem(LZ::WF_OFF + (t >> 2) * 256 + (t & 3), 64u * g);  // fragment (t, lane 16g)
  bz = pz.elem(LZ::BIAS_OFF, 0u);
}

template <int PB, int H>
__device__ __forceinline__ void scalar_head_fwd(const float (&wz)[H], float bz, const float (&h)[PB][H], float (&z)[PB]) {
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < H; ++t) s = fmaf(wz[t], h[pb][t], s);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    z[pb] = s + bz;
  }
}

template <class M, int PB, bool MS>
__global__ __launch_bounds__(256) void prop_fwd_kernel(const float* __restrict__ feat, int64_t plane_stride, int LF, int F,
                                                       const float* __restrict__ sel, const float* __restrict__ packed,
                                                       int64_t N, float* __restrict__ sigma, const int* __restrict__ perm,
                                                       const int* __restrict__ field_start, int K) {
  static_assert(M::NL == 2 && M::NBO == 1, "proposal MLP: two linear layers, scalar output");
  using L0 = typename M::L0;
  TileRange tr{0, N, (int)blockIdx.x, (int)gridDim.x};
  if constexpr (MS) {
    const MsBlock mb = ms_block(field_start, K, gridDim.x, blockIdx.x);
    if (mb.k < 0) return;
    packed += (int64_t)mb.k * M::PACKED;
    tr = TileRange{mb.first_pt, mb.end_pt, mb.j, mb.n};
    N = mb.end_pt;
  }
  __shared__ __attribute__((aligned(16))) float lds[M::FW];
  for (int i = threadIdx.x * 4; i < M::FW; i += 256 * 4)
    *reinterpret_cast<f32x4*>(lds + i) = *reinterpret_cast<const f32x4*>(packed + i);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = ps_lane(), j = lane & 15, g = lane >> 4;
  const LdsW pw{lds};
  float wz[M::HB * 4], bz;
  load_scalar_head<M>(pw.at(M::OFFZ), wz, bz);
  // inputs one tile ahead, taken over before the tile's stores (see prop_bwd_kernel / main_fwd_kernel)
  FeatCols<M::KS0> fc;
  fc.init(plane_stride, LF, F);
  struct In {
    float x[PB][M::KS0], sl[PB];
    int op[PB];
  };
  auto fetch = [&](int64_t first, In& v) {
    load_feat<M::KS0, PB>(feat, fc, F, first, N, v.x);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t p = first + pb * 16 + j;
      v.op[pb] = (int)orig_index<MS>(perm, p, N);
      v.sl[pb] = (p < N) ? sel[p] : 0.0f;
    }
  };
  In cur, nxt;
  auto consume = [&]() {
    cur = nxt;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {  // real copies, made HERE
#pragma unroll
      for (int t = 0; t < M::KS0; ++t) asm volatile("" : "+v"(cur.x[pb][t]));
      asm volatile("" : "+v"(cur.op[pb]), "+v"(cur.sl[pb]));
    }
  };
  const int64_t stride = (int64_t)tr.n * 4 * 16 * PB;
  int64_t first = tr.first_pt + ((int64_t)tr.j * 4 + wave) * 16 * PB;
  fetch(first, nxt);
  consume();
  fetch(first + stride, nxt);
  for (; first < N; first += stride) {
    float h1[PB][M::HB * 4], z[PB], out[PB];
    int op[PB];
    layer_fwd<L0, PB>(pw.at(M::OFF0), cur.x, h1);
    relu_inplace<PB, M::HB * 4>(h1);
    scalar_head_fwd<PB, M::HB * 4>(wz, bz, h1, z);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      out[pb] = expf(z[pb]) * cur.sl[pb];
      op[pb] = cur.op[pb];
    }
    __builtin_amdgcn_sched_barrier(0);
    consume();
    __builtin_amdgcn_sched_barrier(0);
    if (g == 0) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
        if (op[pb] >= 0) sigma[op[pb]] = out[pb];
    }
    fetch(first + 2 * stride, nxt);
  }
}

// All weight/bias gradients live in registers for the whole kernel (no LDS accumulators, no per-tile flush); every
// wave writes its own partial block at the end (gpart has 4 blocks per workgroup).
template <class M, int PB, bool MS>
__global__ __launch_bounds__(256) void prop_bwd_kernel(const float* __restrict__ feat, int64_t plane_stride, int LF, int F,
                                                       const float* __restrict__ sel, const float* __restrict__ packed,
                                                       const float* __restrict__ dsigma, int64_t N, float* __restrict__ dfeat,
                                                       float* __restrict__ gpart, unsigned* __restrict__ level_absmax,
                                                       const int* __restrict__ perm, const int* __restrict__ field_start, int K) {
  static_assert(M::NL == 2 && M::NBO == 1, "proposal MLP: two linear layers, scalar output");
  using L0 = typename M::L0;
  using LZ = typename M::LZ;
  constexpr int H = M::HB * 4;
  TileRange tr{0, N, (int)blockIdx.x, (int)gridDim.x};
  if constexpr (MS) {
    const MsBlock mb = ms_block(field_start, K, gridDim.x, blockIdx.x);
    if (mb.k < 0) return;
    packed += (int64_t)mb.k * M::PACKED;
    if (level_absmax != nullptr) level_absmax += mb.k * (LF / F);
    tr = TileRange{mb.first_pt, mb.end_pt, mb.j, mb.n};
    N = mb.end_pt;
  }
  float mx[M::KS0];
#pragma unroll
  for (int t = 0; t < M::KS0; ++t) mx[t] = 0.f;
  constexpr int SCR = L0::SCRATCH_ROWS * kScratchLd;
  __shared__ __attribute__((aligned(16))) float lds[M::PACKED + 4 * SCR];
  const int wave = threadIdx.x >> 6, lane = ps_lane(), j = lane & 15, g = lane >> 4;
  // the packed weights (17 KB) live in LDS: streamed from L2 they queue behind the wave's own HBM loads
  for (int i = threadIdx.x * 4; i < M::PACKED; i += 1024) *reinterpret_cast<f32x4*>(lds + i) = *reinterpret_cast<const f32x4*>(packed + i);
  __syncthreads();
  float* scratch = lds + M::PACKED + wave * SCR;
  const LdsW gw{lds};
  const LdsW p0 = gw.at(M::OFF0), t0 = gw.at(M::TOFF0);
  float wz[H], bz;
  load_scalar_head<M>(gw.at(M::OFFZ), wz, bz);
  f32x4 dw0[L0::NB][L0::IB], db0[L0::NB];
  float dwz[H], dbz = 0.f;
#pragma unroll
  for (int a = 0; a < L0::NB; ++a) {
    db0[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < L0::IB; ++b) dw0[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int t = 0; t < H; ++t) dwz[t] = 0.f;
  // The kernel does 30 matrix ops per 16 points: it is all memory latency.  Inputs are requested one tile ahead and taken over
  // right before the tile's stores go out (loads and stores share one counter: a load waited for behind a store waits for the
  // store to reach memory, see main_fwd_kernel).
  FeatCols<M::KS0> fc;
  fc.init(plane_stride, LF, F);
  struct In {
    float x[PB][M::KS0], ds[PB], sl[PB];
  };
  auto fetch = [&](int64_t first, In& v) {
    load_feat<M::KS0, PB>(feat, fc, F, first, N, v.x);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t p = first + pb * 16 + j;
      const int64_t op = orig_index<MS>(perm, p, N);
      v.ds[pb] = (op >= 0) ? dsigma[op] : 0.0f;
      v.sl[pb] = (p < N) ? sel[p] : 0.0f;
    }
  };
  In cur, nxt;
  auto consume = [&]() {
    cur = nxt;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {  // real copies, made HERE
#pragma unroll
      for (int t = 0; t < M::KS0; ++t) asm volatile("" : "+v"(cur.x[pb][t]));
      asm volatile("" : "+v"(cur.ds[pb]), "+v"(cur.sl[pb]));
    }
  };
  const int64_t stride = (int64_t)tr.n * 4 * 16 * PB;
  int64_t first = tr.first_pt + ((int64_t)tr.j * 4 + wave) * 16 * PB;
  fetch(first, nxt);
  consume();
  fetch(first + stride, nxt);
  for (; first < N; first += stride) {
    float h1[PB][H], z[PB], dh[PB][H], dx[PB][L0::IB * 4];
    const float(&x)[PB][M::KS0] = cur.x;
    layer_fwd<L0, PB>(p0, x, h1);
    relu_inplace<PB, H>(h1);
    scalar_head_fwd<PB, H>(wz, bz, h1, z);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const float d = cur.ds[pb] * cur.sl[pb] * trunc_exp_grad(z[pb]);  // the same value on the 4 lane groups; 0 past the end
      if (g == 0) dbz += d;
#pragma unroll
      for (int t = 0; t < H; ++t) {
        dwz[t] = fmaf(h1[pb][t], d, dwz[t]);
        dh[pb][t] = h1[pb][t] > 0.0f ? wz[t] * d : 0.0f;
      }
    }
    {
      // dX first (it hides the staging writes of the dW operands), the operands of point block pb+1 staged under the dW MFMAs of pb
      float a0[L0::KSO], dbp_unused[L0::NB];
      frags_bwd_block<L0>(t0, 0, a0);
      layer_bwd_pipe_core<L0, PB, true, false>(t0, a0, scratch, dw0, dbp_unused, dh, x, dx, NoPrefetch());
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int nb = 0; nb < L0::NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) db0[nb][r] += dh[pb][4 * nb + r];  // per-lane partial; reduced over lanes at the end
    }
    __builtin_amdgcn_sched_barrier(0);
    consume();
    __builtin_amdgcn_sched_barrier(0);
    store_dfeat<M::KS0, PB>(dfeat, fc, F, first, N, dx);
    fetch(first + 2 * stride, nxt);
    track_absmax<M::KS0, PB>(dx, mx);
  }
  if (level_absmax != nullptr) publish_absmax<M::KS0>(mx, LF, F, level_absmax);
  float* out = gpart + ((size_t)blockIdx.x * 4 + wave) * M::GPACKED;
  store_layer_acc<L0>(out + M::GOFF0, dw0, db0);
  // head gradient in the packed layout of a [tile][lane][4] dW block: output row 0 lives in lanes 0..15 (= input row),
  // register 0; input row 4*g' + r' of input block ib is the neuron of lane group g', k-step t = 4*ib + r'
  float* oz = out + M::GOFFZ;
#pragma unroll
  for (int t = 0; t < H; ++t) {
    const float s = ps_row16_sum(dwz[t]);
    if (j == 0) oz[LZ::GW_OFF + (((t >> 2) * 64) + 4 * g + (t & 3)) * 4] = s;
  }
  dbz = ps_row16_sum(dbz);
  if (lane == 0) oz[LZ::GB_OFF] = dbz;
}

// ------------------------------------------------------------------------------------------ main field
struct MainArgs {
  const float* feat;
  int64_t plane_stride;
  int LF, F;
  const float* sel;
  const float* dirs;  // [R,3]
  const float* app;   // [R,A] or null
  int S, A;
  const float* packed;  // [base | sem | rgb] packed blocks
  int64_t N;
  // forward outputs (any may be null)
  float* sigma;
  float* rgb;
  float* sem;
  // backward inputs / outputs
  const float* dsigma;
  const float* drgb;
  const float* dsem;
  const float* w;  // null: drgb [N,3] / dsem [N,64] per sample.  non-null [N]: they are per RAY ([R,3] / [R,64], gradients of
                   // the composited outputs) and the per-sample gradient is w[n] * d[ray of n] (composite backward fused in)
  float* dfeat;
  float* dapp;  // [R,A], accumulated with atomics
  float* gpart;
  // hidden activations of the three MLPs, [N, MainCfg::ACT_W] (training forward writes them, the backward reads them instead
  // of recomputing the forward: a third of its matrix ops and half of its L2 weight-fragment traffic); null = not kept
  float* acts;
  // multi-sub-field launch (ms_core.hpp): sorted slot -> caller's point index, sub-field groups, packed weights of sub-field k at
  // packed + k * packed_stride; null / 0 for a single field
  const int* perm;
  const int* field_start;
  int K;
  int64_t packed_stride;
  // three-kernel backward: d(base-MLP output) [ceil(N/16)*16, 80] in register order, written by main_bwd_sem_kernel /
  // main_bwd_rgb_kernel and read by main_bwd_base_kernel (null: the single fused kernel)
  float* dzb;
  // multi-sub-field three-kernel backward: d(appearance) per POINT [N, A] in the caller's order, written (the sorted layout
  // scatters the samples of a ray over the workgroups: 16 float atomics per point otherwise); the caller sums over the samples
  float* dapp_pt;
  // factored path (MainCfg FACT): per-ray part of the colour head's first pre-activation [R, hidden_color] (forward input) and the
  // per-16-point-block sums of the gradient w.r.t. it [ceil(N/16), hidden_color] (backward output); S % 16 == 0
  const float* rray;
  float* dr_part;
  // factored path, compositing of the semantic head fused in (S % (16 PB) == 0: the tiles of a ray run back to back on one wave):
  // forward: bin edges [R, S+1] in, rendering weights [N] and the composited last-hidden activations [R, 64] out (no per-sample
  // [N, 64] array leaves the kernel); backward (semantic kernel): <W_out^T d(semantics of the ray), last hidden activations> per
  // sample [N] out = the semantic head's part of d(weights)
  const float* ebins;
  float* w_out;
  float* hid_ray;
  float* dw_sem;
  // gated inference forward (ps_main_field_fwd_gated, prior extraction): the semantic head of a 32-point tile is evaluated only
  // if some point of the tile has (gate_a[n] + gate_b[n] + sigma[n]) / 3 >= gate_thr; the semantics of the other tiles stay unwritten
  const float* gate_a;
  const float* gate_b;
  float gate_thr;
  unsigned long long* gate_stats;  // nullable, device [2]: += (tiles whose semantic head ran, tiles visited) -- for the roofline row
};

// FACT (the factored semantic path of the training render node, one sub-field): the semantic head's input is a LINEAR function
// of the base MLP's hidden layer (the base output has no activation, ingp_field.py:130-151) and its output is composited
// LINEARLY over the ray (nerfacto_nusc_ms.py:530), so
//   * base layer 1 rows 16..79 and semantic layer 0 are merged into ONE 64 x 64 layer on the hidden activations
//     (W' = W_sem0 W_base1[16:80], b' = W_sem0 b_base1[16:80] + b_sem0: ps_merge_linear, once per step) -- the base MLP ends
//     in 16 outputs (sigma_raw | geo15), the semantic stack in the kernels is  h1 -> ReLU(W' h1 + b') -> ReLU(W_sem1 . + b_sem1);
//   * the semantic head's last (linear) layer is applied ONCE PER RAY to the composited hidden activations
//     (sum_n w_n (W h_n + b) = W (sum_n w_n h_n) + b sum_n w_n: ps_sem_out_fwd / _bwd), the kernels hand out h_n.
// 8192 of the 26752 MACs per sample disappear from the forward and 16384 from the backward; the function, its parameters and
// their gradients are the reference's (fp32 rounding of the re-associated sums aside).
// MERGE_ (inference forward only: ps_main_field_fwd_gated): rewrite 1 alone -- base output layer rows 16..79 merged into the semantic
// head's first layer, so the base MLP ends in 16 outputs and the three-layer head (merged, 64 -> 64, 64 -> 64) reads the base hidden
// layer; everything else (per-sample outputs, conditional loads, colour head) is the plain kernel.
template <int KS0_, int HB_, int HBC_, bool FACT_ = false, bool MERGE_ = false>
struct MainCfg {
  static constexpr bool FACT = FACT_;
  static constexpr bool MERGED = FACT_ || MERGE_;  // the semantic stack hangs off the base HIDDEN layer
  static constexpr int ZB_NB = MERGED ? 1 : 5;     // 16-neuron blocks of the base output
  using Base = MlpT<KS0_, HB_, ZB_NB, 2>;
  using Sem = std::conditional_t<FACT_, MlpT<HB_ * 4, 4, 4, 2>, std::conditional_t<MERGE_, MlpT<HB_ * 4, 4, 4, 3>, MlpT<16, 4, 4, 3>>>;
  // FACT: the colour head's first layer sees the 15 geometry features only (one 16-neuron block of the base output); its
  // direction (SH16) and appearance columns are the same for all samples of a ray and arrive as a per-ray pre-activation term
  // (ps_ray_colour_fwd -> MainArgs::rray), their gradients leave as per-block sums (MainArgs::dr_part -> ps_ray_colour_bwd)
  using Rgb = MlpT<FACT_ ? 4 : 12, HBC_, 1, 3>;
  static constexpr int P_BASE = 0, P_SEM = Base::PACKED, P_RGB = P_SEM + Sem::PACKED, PACKED = P_RGB + Rgb::PACKED;
  static constexpr int G_BASE = 0, G_SEM = Base::GPACKED, G_RGB = G_SEM + Sem::GPACKED, GPACKED = G_RGB + Rgb::GPACKED;
  static constexpr int FW_BASE = 0, FW_SEM = Base::FW, FW_RGB = FW_SEM + Sem::FW, FW = FW_RGB + Rgb::FW;
  // columns of the kept activations (D-register order: block nb, lane group g, register r = neuron 16nb + 4g + r)
  static constexpr int ACT_H1 = 0, ACT_ZB = ACT_H1 + HB_ * 16, ACT_S1 = ACT_ZB + ZB_NB * 16, ACT_S2 = ACT_S1 + 64, ACT_C1 = ACT_S2 + 64,
                       ACT_C2 = ACT_C1 + HBC_ * 16, ACT_CO = ACT_C2 + HBC_ * 16, ACT_W = ACT_CO + 16;
  // d(base output) workspace of the three-kernel backward, register order: block 0 = d(sigma_raw | geo15) from the colour kernel,
  // then d(semantic embedding) [64] / MERGED: d(base hidden layer) [HB * 16] from the semantic kernel
  static constexpr int DZB_W = 16 + (MERGED ? HB_ * 16 : 64);
  static constexpr int SCR_ROWS = Base::SCRATCH_ROWS > Sem::SCRATCH_ROWS
                                      ? (Base::SCRATCH_ROWS > Rgb::SCRATCH_ROWS ? Base::SCRATCH_ROWS : Rgb::SCRATCH_ROWS)
                                      : (Sem::SCRATCH_ROWS > Rgb::SCRATCH_ROWS ? Sem::SCRATCH_ROWS : Rgb::SCRATCH_ROWS);
};

// colour-head input: k-steps 0-3 SH16 of (d+1)/2, 4-7 base-output block 0 (sigma_raw slot has zero weight),
// 8-11 appearance embedding
template <int PB, bool MS>
__device__ __forceinline__ void build_colour_input(const MainArgs& a, int64_t first, const float (&zb)[PB][20],
                                                   float (&cin)[PB][12], int64_t (&ray_of)[PB]) {
  const int lane = ps_lane(), j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int64_t p = first + pb * 16 + j;
    int64_t r;
    if constexpr (MS) {
      const int64_t op = orig_index<true>(a.perm, p, a.N);
      r = ray_index(op >= 0 ? op : 0, a.S);
    } else {
      r = ray_index(p < a.N ? p : a.N - 1, a.S);
    }
    ray_of[pb] = r;
    float sh[16];
    sh4((a.dirs[r * 3] + 1.0f) / 2.0f, (a.dirs[r * 3 + 1] + 1.0f) / 2.0f, (a.dirs[r * 3 + 2] + 1.0f) / 2.0f, sh);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // select component 4t+g without dynamic register indexing
      const float v0 = sh[4 * t], v1 = sh[4 * t + 1], v2 = sh[4 * t + 2], v3 = sh[4 * t + 3];
      cin[pb][t] = g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3));
      cin[pb][4 + t] = zb[pb][t];
      const int c = 4 * t + g;
      cin[pb][8 + t] = (a.app != nullptr && c < a.A) ? a.app[r * a.A + c] : 0.0f;
    }
  }
}

// NW = 4 waves per workgroup (one per SIMD, 512 registers each) share ONE copy of the packed forward weights in LDS (112 KB).
// The training forward writes 1.9 KB per point (kept activations + outputs) and reads 0.2 KB; loads and stores share one
// in-order counter per type on this ISA and complete out of order with each other, so ANY wait for a load behind an
// outstanding store waits for that store to reach memory (phase timer, 8 waves x 2 per SIMD: 55 of 92 k cycles per tile and
// wave spent in such waits).  Hence: every input of a tile (features, selector, per-ray direction / appearance code, the
// caller's index of the point) is requested one tile AHEAD and copied out ("consumed") between the semantic MLP and its
// stores -- the only wait of a tile, behind the base stage's stores, which are a whole semantic MLP old by then.
template <class C, int PB, int NW, bool MS>
__global__ __launch_bounds__(NW * 64) void main_fwd_kernel(MainArgs a) {
  using Base = typename C::Base;
  using Sem = typename C::Sem;
  using Rgb = typename C::Rgb;
  TileRange tr{0, a.N, (int)blockIdx.x, (int)gridDim.x};
  if constexpr (MS) {
    const MsBlock mb = ms_block(a.field_start, a.K, gridDim.x, blockIdx.x);
    if (mb.k < 0) return;
    a.packed += (int64_t)mb.k * a.packed_stride;
    tr = TileRange{mb.first_pt, mb.end_pt, mb.j, mb.n};
    a.N = mb.end_pt;
  }
  __shared__ __attribute__((aligned(16))) float lds[C::FW];
  // forward blocks of the three MLPs, contiguous in LDS
  for (int i = threadIdx.x * 4; i < Base::FW; i += NW * 256)
    *reinterpret_cast<f32x4*>(lds + C::FW_BASE + i) = *reinterpret_cast<const f32x4*>(a.packed + C::P_BASE + i);
  for (int i = threadIdx.x * 4; i < Sem::FW; i += NW * 256)
    *reinterpret_cast<f32x4*>(lds + C::FW_SEM + i) = *reinterpret_cast<const f32x4*>(a.packed + C::P_SEM + i);
  for (int i = threadIdx.x * 4; i < Rgb::FW; i += NW * 256)
    *reinterpret_cast<f32x4*>(lds + C::FW_RGB + i) = *reinterpret_cast<const f32x4*>(a.packed + C::P_RGB + i);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = ps_lane(), j = lane & 15, g = lane >> 4;
  FeatCols<Base::KS0> fc;
  fc.init(a.plane_stride, a.LF, a.F);
  struct In {
    float x[PB][Base::KS0], sel[PB], dirv[PB][3], appv[PB][4];
    f32x4 rr[C::FACT ? PB : 1][C::FACT ? Rgb::HB : 1];  // FACT: the ray's colour-head term for this lane's neurons 16nb + 4g..4g+3
    float e0[C::FACT ? PB : 1], e1[C::FACT ? PB : 1];   // FACT: the sample's bin edges (subtracted after `consume`: a fetch only issues loads)
    float ga[PB], gb[PB];                               // gated inference forward: the two other densities of the point
    int op[PB];  // the point's index in the caller's arrays, -1: no such point
    int opn[MS ? PB : 1];  // MS: raw perm entries of the tile AFTER this one (requested with this tile's loads, see fetch)
  };
  // Multi-sub-field launches reach the caller's arrays (gates, directions, appearance codes, outputs) through perm[slot]: a load whose
  // address is a loaded value.  Issued inside one fetch, the dependent pair made the wave wait for perm -- and, loads returning in
  // order, for everything requested before it -- right there: one exposed memory latency per tile (PMC, prior extraction of a routed
  // tile: 8 M points per launch, 0.75 us of matrix work per tile, the gated forward at 0.34 matrix-pipe busy).  perm therefore runs
  // one tile further ahead than everything else: fetch(t) takes tile t's perm entries from registers (they arrived with tile t - 1's
  // inputs, `opn`) and requests those of tile t + 1.
  auto fetch = [&](int64_t first, int64_t next_first, const int (&op_in)[MS ? PB : 1], In& v) {
    if constexpr (MS) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int64_t q = next_first + pb * 16 + j;
        v.opn[pb] = a.perm[q < a.N ? q : a.N - 1];  // (unconditional; entries past the end are masked when they are taken over)
      }
    }
    if constexpr (C::FACT) {
      // whole tiles only (N % (16 PB) == 0) and every feature column exists (L*F % 4 == 0): the loads are UNCONDITIONAL, with the
      // point index clamped for the tiles requested past the end (never consumed).  A conditional load is a branch plus a merge of
      // the loaded value with a constant, and that merge made the compiler copy the prefetched registers at the loop's back edge
      // -- behind an s_waitcnt vmcnt(0) that exposed the whole memory latency once per tile (phase timer: 3.5 k of 34 k cycles).
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int64_t p = first + pb * 16 + j;
        const int64_t pc = p < a.N ? p : a.N - 1;
        const float* row = a.feat + pc * a.F;
#pragma unroll
        for (int t = 0; t < Base::KS0; ++t) v.x[pb][t] = row[fc.off[t]];
        v.op[pb] = (int)pc;
        v.sel[pb] = a.sel[pc];
        const int64_t r = ray_index(pc, a.S);
#pragma unroll
        for (int nb = 0; nb < Rgb::HB; ++nb) v.rr[pb][nb] = *reinterpret_cast<const f32x4*>(a.rray + r * (Rgb::HB * 16) + 16 * nb + 4 * g);
        const float* e = a.ebins + r * (a.S + 1) + (pc - r * a.S);
        v.e0[pb] = e[0];
        v.e1[pb] = e[1];
      }
      return;
    }
    if constexpr (MS) {  // the sorted layout pads every sub-field to whole chunks: unconditional loads, clamped past the end
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int64_t p = first + pb * 16 + j, pc = p < a.N ? p : a.N - 1;
        const float* row = a.feat + pc * a.F;
#pragma unroll
        for (int t = 0; t < Base::KS0; ++t) v.x[pb][t] = row[fc.off[t]];
      }
    } else {
      load_feat<Base::KS0, PB>(a.feat, fc, a.F, first, a.N, v.x);
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t p = first + pb * 16 + j;
      int64_t op;
      if constexpr (MS)
        op = p < a.N ? (int64_t)op_in[pb] : (int64_t)-1;
      else
        op = orig_index<MS>(a.perm, p, a.N);
      v.op[pb] = (int)op;
      if constexpr (MS)
        v.sel[pb] = a.sel[p < a.N ? p : a.N - 1];  // (padded slots carry selector 0)
      else
        v.sel[pb] = (a.sigma != nullptr && p < a.N) ? a.sel[p] : 0.0f;
      if constexpr (MS) {
        if (a.gate_a != nullptr) {  // (uniform; the loads themselves are unconditional, a slot without a point reads entry 0 and is masked later)
          v.ga[pb] = a.gate_a[op >= 0 ? op : 0];
          v.gb[pb] = a.gate_b[op >= 0 ? op : 0];
        } else {
          v.ga[pb] = v.gb[pb] = 0.0f;
        }
      } else {
        v.ga[pb] = (a.gate_a != nullptr && op >= 0) ? a.gate_a[op] : 0.0f;  // (the gates live in the caller's order)
        v.gb[pb] = (a.gate_a != nullptr && op >= 0) ? a.gate_b[op] : 0.0f;
      }
      int64_t r;
      if constexpr (MS)
        r = ray_index(op >= 0 ? op : 0, a.S);
      else
        r = ray_index(p < a.N ? p : a.N - 1, a.S);
#pragma unroll
      for (int k = 0; k < 3; ++k) v.dirv[pb][k] = (a.rgb != nullptr && first < a.N) ? a.dirs[r * 3 + k] : 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 4 * t + g;
        v.appv[pb][t] = (a.rgb != nullptr && a.app != nullptr && c < a.A && first < a.N) ? a.app[r * a.A + c] : 0.0f;
      }
    }
  };
  In nxt, cur;
  auto consume = [&]() {
    cur = nxt;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {  // real copies, made HERE (the wait for the loads must not sink below later stores)
#pragma unroll
      for (int t = 0; t < Base::KS0; ++t) asm volatile("" : "+v"(cur.x[pb][t]));
      if constexpr (C::FACT) {
#pragma unroll
        for (int nb = 0; nb < Rgb::HB; ++nb) asm volatile("" : "+v"(cur.rr[pb][nb]));
        asm volatile("" : "+v"(cur.sel[pb]), "+v"(cur.op[pb]), "+v"(cur.e0[pb]), "+v"(cur.e1[pb]));
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(cur.appv[pb][t]));
        asm volatile("" : "+v"(cur.sel[pb]), "+v"(cur.dirv[pb][0]), "+v"(cur.dirv[pb][1]), "+v"(cur.dirv[pb][2]), "+v"(cur.op[pb]));
        asm volatile("" : "+v"(cur.ga[pb]), "+v"(cur.gb[pb]));
        if constexpr (MS) asm volatile("" : "+v"(cur.opn[pb]));
      }
    }
  };
  PsTimer* tm = nullptr;
#if defined(PS_TIMING)
  PsTimer tm_;
  tm_.start();
  tm = &tm_;
#endif
  // Tile order.  A wave walks UNITS of `tpu` consecutive tiles; unit u of wave (j, wave) is number (j NW + wave) + i n NW.  Plain
  // calls: one tile per unit (the round-robin order).  FACT: unit = ray (S / (16 PB) tiles), so that the wave carries the ray's
  // optical depth and its weighted sums of the semantic activations in registers from tile to tile.
  struct TileIt {
    int64_t first;
    int k;
  };
  const int tpu = C::FACT ? a.S / (16 * PB) : 1;
  const int64_t unit_jump = ((int64_t)tr.n * NW - 1) * tpu * (16 * PB) + 16 * PB;  // last tile of a unit -> first tile of the wave's next unit
  auto advance = [&](TileIt t) {
    if (t.k + 1 < tpu) return TileIt{t.first + 16 * PB, t.k + 1};
    return TileIt{t.first + unit_jump, 0};
  };
  TileIt it0{tr.first_pt + ((int64_t)tr.j * NW + wave) * tpu * (16 * PB), 0};
  TileIt it1 = advance(it0), it2 = advance(it1);
  {
    int op0[MS ? PB : 1] = {0};
    if constexpr (MS) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int64_t q = it0.first + pb * 16 + j;
        op0[pb] = a.perm[q < a.N ? q : a.N - 1];
      }
    }
    fetch(it0.first, it1.first, op0, nxt);
  }
  consume();
  fetch(it1.first, it2.first, cur.opn, nxt);
  float carry = 0.0f;                    // FACT: optical depth of the ray up to this tile
  float racc[C::FACT ? 16 : 1] = {0.f};  // FACT: this lane's part of sum_n w_n s_n (neurons 16nb + 4g + r, samples j, j + 16, ...)
  unsigned n_tiles = 0, n_sem_tiles = 0;  // gated inference: tiles visited / tiles whose semantic head ran (uniform per wave)
  for (; it0.first < a.N; it0 = it1, it1 = it2, it2 = advance(it2)) {
    const int64_t first = it0.first;
    PS_STAMP(tm, 0)
    float zb[PB][C::ZB_NB * 4], dirv[PB][3], appv[PB][4];
    float h1f[C::MERGED ? PB : 1][C::MERGED ? Base::HB * 4 : 1];  // merged first layer: the base hidden layer feeds the semantic stack
    f32x4 rr[C::FACT ? PB : 1][C::FACT ? Rgb::HB : 1];
    int op_cur[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      op_cur[pb] = cur.op[pb];
      if constexpr (C::FACT) {
#pragma unroll
        for (int nb = 0; nb < Rgb::HB; ++nb) rr[pb][nb] = cur.rr[pb][nb];
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) dirv[pb][k] = cur.dirv[pb][k];
#pragma unroll
        for (int t = 0; t < 4; ++t) appv[pb][t] = cur.appv[pb][t];
      }
    }
    {
      float h1[PB][Base::HB * 4], h2[PB][Base::HB * 4];
      mlp_forward<Base, PB>(LdsW{lds + C::FW_BASE}, cur.x, h1, h2, zb);
      PS_STAMP(tm, 2)
      if (a.acts != nullptr) {
        store_act<Base::HB, PB>(a.acts, C::ACT_W, C::ACT_H1, first, a.N, h1);
        store_act<C::ZB_NB, PB>(a.acts, C::ACT_W, C::ACT_ZB, first, a.N, zb);
      }
      if constexpr (C::MERGED) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int t = 0; t < Base::HB * 4; ++t) h1f[pb][t] = h1[pb][t];
      }
    }
    float wq[C::FACT ? PB : 1];  // FACT: the rendering weight of this lane's sample (all four lane groups)
    if constexpr (C::FACT) {
      // rendering weights of the tile's samples (ns/model_components/ray_samplers.py get_weights: alpha_n T_n with
      // T_n = exp(-sum_{m<n} sigma_m delta_m)), row 0 of the wave = the 16 samples of a block; the other rows compute unused values
      // (sigma_raw lives in row 0 of the wave; it is broadcast to the four rows first, every row then forms the same weights with
      // row-local DPP scans: one trip through the LDS crossbar per block instead of seven)
      float sig[PB], dd[PB], ex[PB], tot[PB];
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        sig[pb] = expf(__shfl(zb[pb][0], j, 64)) * cur.sel[pb];
        dd[pb] = (cur.e1[pb] - cur.e0[pb]) * sig[pb];
        const float incl = ps_row16_incl_scan(dd[pb]);
        tot[pb] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 15));
        ex[pb] = ps_row_shr<1>(incl);
      }
      if (it0.k == 0) carry = 0.0f;
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        wq[pb] = nan_to_num((1.0f - expf(-dd[pb])) * expf(-(carry + ex[pb])));
        carry += tot[pb];
        if (g == 0) {
          a.sigma[first + pb * 16 + j] = sig[pb];
          a.w_out[first + pb * 16 + j] = wq[pb];
        }
      }
    } else if (a.sigma != nullptr && g == 0) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
        if (op_cur[pb] >= 0) a.sigma[op_cur[pb]] = expf(zb[pb][0]) * cur.sel[pb];
    }
    PS_STAMP(tm, 3)
    bool consumed = false;
    bool run_sem = C::FACT || a.sem != nullptr;
    if constexpr (!C::FACT) {
      if (a.gate_a != nullptr) {  // (workgroup-uniform pointer; the ballot makes the decision wave-uniform)
        bool hit = false;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          if (g == 0 && op_cur[pb] >= 0) {
            const float mean = ((cur.ga[pb] + cur.gb[pb]) + expf(zb[pb][0]) * cur.sel[pb]) / 3.0f;  // ps_mean_density's formula
            hit |= mean >= a.gate_thr;
          }
        }
        run_sem = run_sem && __ballot(hit) != 0ull;
        ++n_tiles;
        n_sem_tiles += run_sem ? 1u : 0u;
      }
    }
    if (run_sem) {
      float s1[PB][16], s2[PB][16], so[PB][16];
      if constexpr (C::FACT) {
        // merged first layer on the base hidden layer, second layer; `so` = the last HIDDEN activations (the output layer is
        // applied per ray after compositing)
#if defined(PS_SEM_BF16X3)  // exploratory build (tools/build_variant.sh bf16x3 -DPS_SEM_BF16X3): NOT the product arithmetic
        mlp_forward<Sem, PB, LdsW, NoPost, true>(LdsW{lds + C::FW_SEM}, h1f, s1, s2 /*unused*/, so);
#else
        mlp_forward<Sem, PB>(LdsW{lds + C::FW_SEM}, h1f, s1, s2 /*unused*/, so);
#endif
        relu_inplace<PB, 16>(so);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int t = 0; t < 16; ++t) s2[pb][t] = so[pb][t];
      } else if constexpr (C::MERGED) {
        mlp_forward<Sem, PB>(LdsW{lds + C::FW_SEM}, h1f, s1, s2, so);  // merged first layer on the base hidden layer, then the head as it is
      } else {
        float sin_[PB][16];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int t = 0; t < 16; ++t) sin_[pb][t] = zb[pb][4 + t];
        mlp_forward<Sem, PB>(LdsW{lds + C::FW_SEM}, sin_, s1, s2, so);
      }
      PS_STAMP(tm, 4)
      // the next tile's inputs are taken over HERE: the youngest stores in flight are the base stage's, a whole semantic MLP old
      __builtin_amdgcn_sched_barrier(0);
      consume();
      consumed = true;
      __builtin_amdgcn_sched_barrier(0);
      PS_STAMP(tm, 6)
      if (a.acts != nullptr) {
        store_act<4, PB>(a.acts, C::ACT_W, C::ACT_S1, first, a.N, s1);
        store_act<4, PB>(a.acts, C::ACT_W, C::ACT_S2, first, a.N, s2);
      }
      if constexpr (C::FACT) {
        // composited last-hidden activations of the ray: per-lane partial sums over the tiles, one cross-lane sum per ray
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int t = 0; t < 16; ++t) racc[t] = fmaf(so[pb][t], wq[pb], racc[t]);
        if (it0.k == tpu - 1) {
          const int64_t r = ray_index(first, a.S);
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
            f32x4 sum;
#pragma unroll
            for (int q = 0; q < 4; ++q) sum[q] = ps_row16_sum(racc[4 * nb + q]);
            if (j == 0) *reinterpret_cast<f32x4*>(a.hid_ray + r * 64 + 16 * nb + 4 * g) = sum;
          }
#pragma unroll
          for (int t = 0; t < 16; ++t) racc[t] = 0.0f;
        }
      } else {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          if (op_cur[pb] >= 0) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
              *reinterpret_cast<f32x4*>(a.sem + (int64_t)op_cur[pb] * 64 + 16 * nb + 4 * g) =
                  (f32x4){so[pb][4 * nb], so[pb][4 * nb + 1], so[pb][4 * nb + 2], so[pb][4 * nb + 3]};
          }
        }
      }
    }
    PS_STAMP(tm, 5)
    if (!consumed) {  // (no semantic head in this call; uniform)
      __builtin_amdgcn_sched_barrier(0);
      consume();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (a.rgb != nullptr) {
      float cin[PB][Rgb::KS0], c1[PB][Rgb::HB * 4], c2[PB][Rgb::HB * 4], co[PB][4];
      if constexpr (C::FACT) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int t = 0; t < 4; ++t) cin[pb][t] = zb[pb][t];
        // first layer on the geometry features; the ray's direction / appearance term joins the pre-activations before the ReLU
        mlp_forward<Rgb, PB>(LdsW{lds + C::FW_RGB}, cin, c1, c2, co, [&](float (&h)[PB][Rgb::HB * 4]) {
#pragma unroll
          for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int nb = 0; nb < Rgb::HB; ++nb)
#pragma unroll
              for (int r = 0; r < 4; ++r) h[pb][4 * nb + r] += rr[pb][nb][r];
        });
      } else {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          float sh[16];
          sh4((dirv[pb][0] + 1.0f) / 2.0f, (dirv[pb][1] + 1.0f) / 2.0f, (dirv[pb][2] + 1.0f) / 2.0f, sh);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float v0 = sh[4 * t], v1 = sh[4 * t + 1], v2 = sh[4 * t + 2], v3 = sh[4 * t + 3];
            cin[pb][t] = g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3));  // component 4t+g without dynamic register indexing
            cin[pb][4 + t] = zb[pb][t];
            cin[pb][8 + t] = appv[pb][t];
          }
        }
        mlp_forward<Rgb, PB>(LdsW{lds + C::FW_RGB}, cin, c1, c2, co);
      }
      PS_STAMP(tm, 7)
      if (a.acts != nullptr) {
        store_act<Rgb::HB, PB>(a.acts, C::ACT_W, C::ACT_C1, first, a.N, c1);
        store_act<Rgb::HB, PB>(a.acts, C::ACT_W, C::ACT_C2, first, a.N, c2);
        store_act<1, PB>(a.acts, C::ACT_W, C::ACT_CO, first, a.N, co);
      }
      if (g == 0) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          if (op_cur[pb] >= 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) a.rgb[(int64_t)op_cur[pb] * 3 + k] = 1.0f / (1.0f + expf(-co[pb][k]));
          }
        }
      }
    }
    fetch(it2.first, advance(it2).first, cur.opn, nxt);  // (`cur` holds the NEXT tile's inputs by now: its opn = perm of tile it2)
    PS_STAMP(tm, 8)
  }
#if defined(PS_TIMING)
  if (lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_ps_timing_fwd[i], tm_.acc[i]);
#endif
  if constexpr (!C::FACT) {
    if (a.gate_stats != nullptr && lane == 0 && n_tiles != 0) {
      atomicAdd(a.gate_stats, (unsigned long long)n_sem_tiles);
      atomicAdd(a.gate_stats + 1, (unsigned long long)n_tiles);
    }
  }
}

// STORED: the hidden activations come from the training forward (a.acts) instead of being recomputed
template <class C, int PB, int NW, bool STORED, bool MS>
__global__ __launch_bounds__(NW * 64) void main_bwd_kernel(MainArgs a) {
  TileRange tr{0, a.N, (int)blockIdx.x, (int)gridDim.x};
  if constexpr (MS) {
    const MsBlock mb = ms_block(a.field_start, a.K, gridDim.x, ms_logical_block(blockIdx.x, gridDim.x));
    if (mb.k < 0) return;  // its partial block is never read (ms_field_blocks)
    a.packed += (int64_t)mb.k * a.packed_stride;
    tr = TileRange{mb.first_pt, mb.end_pt, mb.j, mb.n};
    a.N = mb.end_pt;
  }
  constexpr int SCR = C::SCR_ROWS * kScratchLd;
  __shared__ __attribute__((aligned(16))) float lds[C::GPACKED + NW * SCR + 16];
  float* gacc = lds;
  int* locks = reinterpret_cast<int*>(lds + C::GPACKED + NW * SCR);
  for (int i = threadIdx.x; i < C::GPACKED; i += NW * 64) gacc[i] = 0.0f;
  if (threadIdx.x < 16) locks[threadIdx.x] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = ps_lane(), j = lane & 15, g = lane >> 4;
  float* scratch = lds + C::GPACKED + wave * SCR;
  const GlobalW pk_all = make_global_w(a.packed, C::PACKED);
  const GlobalW pk_base = pk_all.at(C::P_BASE), pk_sem = pk_all.at(C::P_SEM), pk_rgb = pk_all.at(C::P_RGB);
  // workgroup-uniform trip count (the dW flush contains workgroup barriers); out-of-range tiles are fully masked
#if defined(PS_TIMING)
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
  for (int64_t base = (int64_t)tr.j * NW; tr.first_pt + base * 16 * PB < a.N; base += (int64_t)tr.n * NW) {
    const int64_t first = tr.first_pt + (base + wave) * 16 * PB;
    PS_TSTAMP(0)
    // ---- recompute base
    float x[PB][C::Base::KS0], h1[PB][C::Base::HB * 4], hdummy[PB][C::Base::HB * 4], zb[PB][20];
    load_feat<C::Base::KS0, PB>(a.feat, a.plane_stride, a.LF, a.F, first, a.N, x);
    if constexpr (STORED) {
      load_act<C::Base::HB, PB>(a.acts, C::ACT_W, C::ACT_H1, first, a.N, h1);
      load_act<5, PB>(a.acts, C::ACT_W, C::ACT_ZB, first, a.N, zb);
    } else {
      mlp_forward<typename C::Base, PB>(pk_base, x, h1, hdummy, zb);
    }
    float dzb[PB][20];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t p = first + pb * 16 + j;
      const int64_t op = orig_index<MS>(a.perm, p, a.N);
#pragma unroll
      for (int t = 0; t < 20; ++t) dzb[pb][t] = 0.0f;
      if (a.dsigma != nullptr && g == 0 && op >= 0) dzb[pb][0] = a.dsigma[op] * a.sel[p] * trunc_exp_grad(zb[pb][0]);
    }
    PS_TSTAMP(1)
    // ---- semantic head
    if (a.dsem != nullptr) {
      float sin_[PB][16], s1[PB][16], s2[PB][16], so[PB][16];
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int t = 0; t < 16; ++t) sin_[pb][t] = zb[pb][4 + t];
      if constexpr (STORED) {
        load_act<4, PB>(a.acts, C::ACT_W, C::ACT_S1, first, a.N, s1);
        load_act<4, PB>(a.acts, C::ACT_W, C::ACT_S2, first, a.N, s2);
      } else {
        mlp_forward<typename C::Sem, PB>(pk_sem, sin_, s1, s2, so);
      }
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int64_t op = orig_index<MS>(a.perm, first + pb * 16 + j, a.N);
        const bool in = op >= 0;
        const float wp = (a.w != nullptr && in) ? a.w[op] : 1.0f;
        const float* src = a.dsem + (in ? ((a.w != nullptr) ? ray_index(op, a.S) : op) : 0) * 64 + 4 * g;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (in) d = *reinterpret_cast<const f32x4*>(src + 16 * nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) so[pb][4 * nb + r] = d[r] * wp;
        }
      }
      float dsin[PB][16];
      mlp_backward<typename C::Sem, PB, true>(pk_sem, scratch, gacc + C::G_SEM, locks + 3, sin_, s1, s2, so, dsin);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int t = 0; t < 16; ++t) dzb[pb][4 + t] += dsin[pb][t];
    }
    PS_TSTAMP(2)
    // ---- colour head
    if (a.drgb != nullptr) {
      float cin[PB][12], c1[PB][C::Rgb::HB * 4], c2[PB][C::Rgb::HB * 4], co[PB][4];
      int64_t ray_of[PB];
      build_colour_input<PB, MS>(a, first, zb, cin, ray_of);
      if constexpr (STORED) {
        load_act<C::Rgb::HB, PB>(a.acts, C::ACT_W, C::ACT_C1, first, a.N, c1);
        load_act<C::Rgb::HB, PB>(a.acts, C::ACT_W, C::ACT_C2, first, a.N, c2);
        load_act<1, PB>(a.acts, C::ACT_W, C::ACT_CO, first, a.N, co);
      } else {
        mlp_forward<typename C::Rgb, PB>(pk_rgb, cin, c1, c2, co);
      }
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int64_t op = orig_index<MS>(a.perm, first + pb * 16 + j, a.N);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float d = 0.0f;
          if (k < 3 && g == 0 && op >= 0) {
            const float s = 1.0f / (1.0f + expf(-co[pb][k]));
            const float up = (a.w != nullptr) ? a.w[op] * a.drgb[ray_of[pb] * 3 + k] : a.drgb[op * 3 + k];
            d = up * s * (1.0f - s);
          }
          co[pb][k] = d;
        }
      }
      float dcin[PB][12];
      mlp_backward<typename C::Rgb, PB, true>(pk_rgb, scratch, gacc + C::G_RGB, locks + 6, cin, c1, c2, co, dcin);
      // d(appearance) is per RAY: when a 16-point block lies inside one ray (S % 16 == 0) reduce it over the 16 lanes
      // first -> 16x fewer global atomics (67 M -> 4 M per step at cfg 2; MI355X does ~21 G atomics/s)
      // (multi-sub-field launches sort the points, so a block may straddle rays or contain padding: there the 16 lanes are
      //  reduced only when they all carry the same ray)
      bool block_in_ray = (a.S % 16) == 0;
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int64_t p = first + pb * 16 + j;
        bool pt_ok = p < a.N;
        if constexpr (MS) {
          const int64_t op = orig_index<true>(a.perm, p, a.N);
          pt_ok = op >= 0;
          const int rid = pt_ok ? (int)ray_of[pb] : -1;
          bool same = true;
          same &= rid == __builtin_amdgcn_update_dpp(0, rid, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
          same &= rid == __builtin_amdgcn_update_dpp(0, rid, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
          same &= rid == __builtin_amdgcn_update_dpp(0, rid, 0x141, 0xF, 0xF, false);  // row_half_mirror
          same &= rid == __builtin_amdgcn_update_dpp(0, rid, 0x140, 0xF, 0xF, false);  // row_mirror
          const unsigned long long ok = __ballot(same && pt_ok);
          block_in_ray = ((ok >> (16 * g)) & 0xffffull) == 0xffffull;  // all 16 lanes valid and on one ray
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          dzb[pb][t] += dcin[pb][4 + t];  // geo slots (the sigma_raw slot has zero weights -> exactly 0)
          if (a.dapp != nullptr) {
            const int c = 4 * t + g;
            float v = pt_ok ? dcin[pb][8 + t] : 0.0f;
            float vs = v;
            if constexpr (MS) vs = ps_row16_sum(v);  // block_in_ray differs per 16-lane row: the DPP reduction runs on all lanes
            if (block_in_ray) {
              if constexpr (!MS) vs = ps_row16_sum(v);
              if (j == 0 && first + pb * 16 < a.N && c < a.A) unsafeAtomicAdd(a.dapp + ray_of[pb] * a.A + c, vs);
            } else if (pt_ok && c < a.A) {
              unsafeAtomicAdd(a.dapp + ray_of[pb] * a.A + c, v);
            }
          }
        }
      }
    }
    PS_TSTAMP(3)
    // ---- base MLP backward -> d(features)
    float dx[PB][C::Base::L0::IB * 4];
    mlp_backward<typename C::Base, PB, true>(pk_base, scratch, gacc + C::G_BASE, locks + 0, x, h1, hdummy, dzb, dx);
    store_dfeat<C::Base::KS0, PB>(a.dfeat, a.plane_stride, a.LF, a.F, first, a.N, dx);
    PS_TSTAMP(4)
  }
#if defined(PS_TIMING)
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_ps_timing[i], tacc[i]);
#endif
  __syncthreads();
  float* out = a.gpart + (size_t)(MS ? ms_logical_block(blockIdx.x, gridDim.x) : (int)blockIdx.x) * C::GPACKED;
  for (int i = threadIdx.x; i < C::GPACKED; i += NW * 64) out[i] = gacc[i];
}

// ------------------------------------------------------------------------------------------ main backward as three kernels
// The fused backward above accumulates the weight gradients of all eight layers in 112 KB of LDS (a read-modify-write of
// every dW tile per 32 points) and therefore streams the transposed weight fragments (107 KB per 32 points) from L2, where
// they queue behind the wave's own HBM activation loads (vector memory returns in order): its matrix pipe runs at 50 %.
// Split by MLP, the weight gradients of ONE stack fit in the registers of a lone wave per SIMD (semantic head 195 + 12 of 512
// per lane) and its transposed fragments fit in LDS (<= 49 KB):
//   main_bwd_sem_kernel    kept activations zb[16:80], s1, s2 + d(semantics)   -> dW/db, d(zb[16:80]) -> dzb blocks 1..4
//   main_bwd_rgb_kernel    zb[0:16], c1, c2, co + d(rgb), d(sigma)             -> dW/db, d(appearance), d(zb[0:16]) -> dzb block 0
//   main_bwd_base_kernel   features, h1 + dzb                                 -> dW/db, d(features)
// dzb [ceil(N/16)*16, 80] is a workspace in register order (the price: 640 B per point of extra HBM traffic).  Each kernel
// reduces its four waves' accumulators through LDS once, at the end, and writes its share of the workgroup's partial block.
template <bool MS>
__device__ __forceinline__ bool main_bwd_setup(MainArgs& a, TileRange& tr, int& lb) {
  tr = TileRange{0, a.N, (int)blockIdx.x, (int)gridDim.x};
  lb = blockIdx.x;
  if constexpr (MS) {
    lb = ms_logical_block(blockIdx.x, gridDim.x);
    const MsBlock mb = ms_block(a.field_start, a.K, gridDim.x, lb);
    if (mb.k < 0) return false;  // its partial block is never read (ms_field_blocks)
    a.packed += (int64_t)mb.k * a.packed_stride;
    tr = TileRange{mb.first_pt, mb.end_pt, mb.j, mb.n};
    a.N = mb.end_pt;
  }
  return true;
}

// LDS of a one-stack backward kernel: [max(transposed fragments, gradient block)] [NW per-wave transposes]
template <class M, int NW>
struct OneStackLds {
  static constexpr int WT = M::PACKED - M::FW;
  static constexpr int HEAD = WT > M::GPACKED ? WT : M::GPACKED;
  static constexpr int SCR = M::SCRATCH_ROWS * kScratchLd;
  static constexpr int FLOATS = HEAD + NW * SCR;
};
template <class M, int NW>
__device__ __forceinline__ void load_transposed(float* __restrict__ lds, const float* __restrict__ packed_stack) {
  for (int i = threadIdx.x * 4; i < OneStackLds<M, NW>::WT; i += NW * 256)
    *reinterpret_cast<f32x4*>(lds + i) = *reinterpret_cast<const f32x4*>(packed_stack + M::FW + i);
  __syncthreads();
}
// the waves add their register accumulators into the (now free) head of the LDS, which then goes out as the partial block
template <class M, int NW>
__device__ __forceinline__ void reduce_store(float* __restrict__ lds, const MlpAcc<M>& acc, float* __restrict__ out) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // every wave is done with the fragments
#pragma unroll 1
  for (int w = 0; w < NW; ++w) {
    if (wave == w) acc.add_to(lds, w == 0);
    __syncthreads();
  }
  for (int i = threadIdx.x; i < M::GPACKED; i += NW * 64) out[i] = lds[i];
}

template <class C, int PB, int NW, bool MS>
__global__ __launch_bounds__(NW * 64) void main_bwd_sem_kernel(MainArgs a) {
  using Sem = typename C::Sem;
  using L = OneStackLds<Sem, NW>;
  TileRange tr;
  int lb;
  if (!main_bwd_setup<MS>(a, tr, lb)) return;
  __shared__ __attribute__((aligned(16))) float lds[L::FLOATS];
  load_transposed<Sem, NW>(lds, a.packed + C::P_SEM);
  const int wave = threadIdx.x >> 6, lane = ps_lane(), j = lane & 15, g = lane >> 4;
  float* scratch = lds + L::HEAD + wave * L::SCR;
  const LdsW pk{lds - Sem::FW};  // mlp_backward addresses the transposed blocks at TOFF* >= FW of the stack's packed block
  MlpAcc<Sem> acc;
  acc.zero();
  PsTimer* tm = nullptr;
#if defined(PS_TIMING)
  PsTimer tm_;
  tm_.start();
  tm = &tm_;
#endif
  // A lone wave per SIMD has nobody to hide HBM latency behind (measured: 4.4 k of 35 k cycles per tile waiting for the first
  // operands).  d(output) -- the only operand the first matrix phase needs -- is gathered one tile AHEAD (34 registers); the
  // kept activations of the current tile are requested at the top and arrive under that phase (mlp_backward_acc: H_LATE).
  struct Gather {
    float w[PB];
    f32x4 d[PB][4];
  };
  auto gather = [&](int64_t first, Gather& v) {
    if constexpr (C::FACT) {  // whole tiles, per-ray gradients: unconditional loads (clamped past the end, never used there)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int64_t p = first + pb * 16 + j, pc = p < a.N ? p : a.N - 1;
        v.w[pb] = a.w[pc];
        const float* src = a.dsem + ray_index(pc, a.S) * 64 + 4 * g;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) v.d[pb][nb] = *reinterpret_cast<const f32x4*>(src + 16 * nb);
      }
      return;
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t op = orig_index<MS>(a.perm, first + pb * 16 + j, a.N);
      const bool in = op >= 0;
      v.w[pb] = (a.w != nullptr && in) ? a.w[op] : 1.0f;
      const float* src = a.dsem + (in ? ((a.w != nullptr) ? ray_index(op, a.S) : op) : 0) * 64 + 4 * g;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        v.d[pb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (in) v.d[pb][nb] = *reinterpret_cast<const f32x4*>(src + 16 * nb);
      }
    }
  };
  const int64_t stride = (int64_t)tr.n * NW * 16 * PB;
  int64_t first = tr.first_pt + ((int64_t)tr.j * NW + wave) * 16 * PB;
  // Loads and stores share ONE in-order counter per type on this ISA (vmcnt) and complete out of order with each other: while
  // a store is outstanding every wait for a load becomes "wait for everything".  The gathered operands are therefore consumed
  // (and the next gather is issued) at the END of a tile, BEFORE its stores go out; the only waits behind the stores then come
  // 4 k cycles later, when everything has long arrived.
  Gather ga;
  float so[PB][16], w_cur[C::FACT ? PB : 1];
  auto consume = [&]() {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      if constexpr (C::FACT) w_cur[pb] = ga.w[pb];  // FACT: `so` holds the ray's gradient unscaled until its dot product is taken
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) so[pb][4 * nb + r] = C::FACT ? ga.d[pb][nb][r] : ga.d[pb][nb][r] * ga.w[pb];
    }
  };
  gather(first, ga);
  consume();
  gather(first + stride, ga);
  for (; first < a.N; first += stride) {
    PS_STAMP(tm, 0)
    float sin_[PB][Sem::KS0], s1[PB][16], s2[PB][16];
    load_act<4, PB, C::FACT || MS>(a.acts, C::ACT_W, C::ACT_S2, first, a.N, s2);
    load_act<4, PB, C::FACT || MS>(a.acts, C::ACT_W, C::ACT_S1, first, a.N, s1);
    if constexpr (C::MERGED)
      load_act<Sem::KS0 / 4, PB, C::FACT || MS>(a.acts, C::ACT_W, C::ACT_H1, first, a.N, sin_);  // the base hidden layer = the merged first layer's input
    else
      load_act<4, PB, MS>(a.acts, C::ACT_W, C::ACT_ZB + 16, first, a.N, sin_);  // base outputs 16..79 = the head's input
    PS_STAMP(tm, 1)
    float dsin[PB][Sem::L0::IB * 4];
    float dws[C::FACT ? PB : 1];
    if constexpr (C::FACT) {
      // the semantic head's part of d(weight of the sample) = <v_ray, s2_n>, v = W_out^T d(semantics of the ray)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        float dot = 0.0f;
#pragma unroll
        for (int t = 0; t < 16; ++t) dot = fmaf(so[pb][t], s2[pb][t], dot);
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        dws[pb] = dot;
#pragma unroll
        for (int t = 0; t < 16; ++t) so[pb][t] *= w_cur[pb];
      }
      // `so` = w[n] * v is the gradient w.r.t. the last hidden activations s2 (the output layer was applied per ray): ReLU mask,
      // then the two layers of the in-kernel stack; the result is d(base hidden layer)
      relu_mask<PB, 16>(so, s2);
    }
    mlp_backward_acc<Sem, PB, true>(
        pk, scratch, acc,
        [&](float (&x)[PB][Sem::KS0]) {
#pragma unroll
          for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int t = 0; t < Sem::KS0; ++t) x[pb][t] = sin_[pb][t];
        },
        s1, s2, so, dsin, tm);
    __builtin_amdgcn_sched_barrier(0);
    consume();                            // tile i+1's operands (requested a whole tile ago)
    gather(first + 2 * stride, ga);
    __builtin_amdgcn_sched_barrier(0);
    store_act<Sem::L0::IB, PB>(a.dzb, C::DZB_W, 16, first, a.N, dsin);  // FACT: d(base hidden layer) from the semantic stack
    if constexpr (C::FACT) {
      if (g == 0) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
          if (first + pb * 16 + j < a.N) a.dw_sem[first + pb * 16 + j] = dws[pb];
      }
    }
    PS_STAMP(tm, 8)
  }
#if defined(PS_TIMING)
  if (lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_ps_timing[i], tm_.acc[i]);
#endif
  reduce_store<Sem, NW>(lds, acc, a.gpart + (size_t)lb * C::GPACKED + C::G_SEM);
}

template <class C, int PB, int NW, bool MS>
__global__ __launch_bounds__(NW * 64) void main_bwd_rgb_kernel(MainArgs a) {
  using Rgb = typename C::Rgb;
  using L = OneStackLds<Rgb, NW>;
  TileRange tr;
  int lb;
  if (!main_bwd_setup<MS>(a, tr, lb)) return;
  __shared__ __attribute__((aligned(16))) float lds[L::FLOATS];
  load_transposed<Rgb, NW>(lds, a.packed + C::P_RGB);
  const int wave = threadIdx.x >> 6, lane = ps_lane(), j = lane & 15, g = lane >> 4;
  float* scratch = lds + L::HEAD + wave * L::SCR;
  const LdsW pk{lds - Rgb::FW};
  MlpAcc<Rgb> acc;
  acc.zero();
  // what the first matrix phase needs -- the pre-sigmoid colour, its upstream gradient and the last hidden layer -- is fetched
  // one tile AHEAD; everything else is requested at the top of the tile and arrives under the matrix work before its use
  // (the first layer's input is assembled from the per-ray directions / appearance codes right before that layer)
  struct Head {
    float co[PB][4], w[PB], dr[PB][3], c2[PB][Rgb::HB * 4];
    float ds[PB];  // d(sigma) * selector of the point (lane group 0), 0 elsewhere: the density gradient joins d(base output 0)
    float sel[C::FACT ? PB : 1];  // FACT: ds holds the raw d(sigma), the selector and the lane-group mask are applied in `consume`
    int64_t ray[PB];
  };
  auto fetch_head = [&](int64_t first, Head& h) {
    if constexpr (C::FACT) {
      // whole tiles: unconditional loads in every lane (clamped past the end), the products and the lane-group mask are applied
      // in `consume` -- a product formed here waits for its two loads on the spot, one full memory latency per tile
      load_act<1, PB, true>(a.acts, C::ACT_W, C::ACT_CO, first, a.N, h.co);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int64_t p = first + pb * 16 + j, pc = p < a.N ? p : a.N - 1;
        const int64_t r = ray_index(pc, a.S);
        h.ray[pb] = r;
        h.ds[pb] = a.dsigma[pc];
        h.sel[pb] = a.sel[pc];
        h.w[pb] = a.w[pc];
#pragma unroll
        for (int k = 0; k < 3; ++k) h.dr[pb][k] = a.drgb[r * 3 + k];
      }
      load_act<Rgb::HB, PB, true>(a.acts, C::ACT_W, C::ACT_C2, first, a.N, h.c2);
      return;
    }
    load_act<1, PB, MS>(a.acts, C::ACT_W, C::ACT_CO, first, a.N, h.co);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t p = first + pb * 16 + j;
      const int64_t op = orig_index<MS>(a.perm, p, a.N);
      int64_t r;
      if constexpr (MS)
        r = ray_index(op >= 0 ? op : 0, a.S);
      else
        r = ray_index(p < a.N ? p : a.N - 1, a.S);
      h.ray[pb] = r;
      h.ds[pb] = (a.dsigma != nullptr && g == 0 && op >= 0) ? a.dsigma[op] * a.sel[p] : 0.0f;
      h.w[pb] = 1.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) h.dr[pb][k] = 0.0f;
      if (g == 0 && op >= 0) {
        if (a.w != nullptr) h.w[pb] = a.w[op];
        const float* src = (a.w != nullptr) ? a.drgb + r * 3 : a.drgb + op * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) h.dr[pb][k] = src[k];
      }
    }
    load_act<Rgb::HB, PB, MS>(a.acts, C::ACT_W, C::ACT_C2, first, a.N, h.c2);
  };
  const int64_t stride = (int64_t)tr.n * NW * 16 * PB;
  int64_t first = tr.first_pt + ((int64_t)tr.j * NW + wave) * 16 * PB;
  // the head is consumed at the END of the previous tile, before that tile's stores and atomics (see main_bwd_sem_kernel)
  Head hd;
  float c2[PB][Rgb::HB * 4], co[PB][4], ds_head[PB];
  int64_t ray_head[PB];
  auto consume = [&]() {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      ray_head[pb] = hd.ray[pb];
      if constexpr (C::FACT)
        ds_head[pb] = g == 0 ? hd.ds[pb] * hd.sel[pb] : 0.0f;
      else
        ds_head[pb] = hd.ds[pb];
      asm volatile("" : "+v"(ds_head[pb]));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float d = 0.0f;
        if (k < 3) {
          const float sg = 1.0f / (1.0f + expf(-hd.co[pb][k]));
          d = hd.w[pb] * hd.dr[pb][k] * sg * (1.0f - sg);  // dr is zero outside lane group 0 / past the end
          if constexpr (C::FACT) d = g == 0 ? d : 0.0f;    // (FACT: every lane loaded the ray's gradient)
        }
        co[pb][k] = d;
      }
#pragma unroll
      for (int t = 0; t < Rgb::HB * 4; ++t) {
        c2[pb][t] = hd.c2[pb][t];
        asm volatile("" : "+v"(c2[pb][t]));  // a real copy, made HERE
      }
    }
  };
  fetch_head(first, hd);
  consume();
  fetch_head(first + stride, hd);
  PsTimer* tm = nullptr;
#if defined(PS_TIMING)
  PsTimer tm_;
  tm_.start();
  tm = &tm_;
#endif
  for (; first < a.N; first += stride) {
    PS_STAMP(tm, 0)
    float c1[PB][Rgb::HB * 4], zb0[PB][4], dirv[PB][3], appv[PB][4], ds_cur[PB];
    int64_t ray_of[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) ds_cur[pb] = ds_head[pb];
    load_act<Rgb::HB, PB, C::FACT || MS>(a.acts, C::ACT_W, C::ACT_C1, first, a.N, c1);
    load_act<1, PB, C::FACT || MS>(a.acts, C::ACT_W, C::ACT_ZB, first, a.N, zb0);  // sigma_raw | geo15
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t r = ray_head[pb];
      ray_of[pb] = r;
      if constexpr (!C::FACT) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dirv[pb][k] = a.dirs[r * 3 + k];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int c = 4 * t + g;
          appv[pb][t] = (a.app != nullptr && c < a.A) ? a.app[r * a.A + c] : 0.0f;
        }
      }
    }
    PS_STAMP(tm, 1)
    float dcin[PB][Rgb::L0::IB * 4];
    if constexpr (C::FACT) {
      // first layer on the geometry features only; the gradient w.r.t. its pre-activations, summed over the 16 points of a block
      // (one ray: S % 16 == 0), is what the per-ray direction / appearance term receives (ps_ray_colour_bwd)
      mlp_backward_acc<Rgb, PB, true, true>(
          pk, scratch, acc,
          [&](float (&cin)[PB][Rgb::KS0]) {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
              for (int t = 0; t < 4; ++t) cin[pb][t] = zb0[pb][t];
          },
          c1, c2, co, dcin, tm, NoPost(),
          [&](float (&dh)[PB][Rgb::HB * 4]) {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
              const int64_t blk = first / 16 + pb;
#pragma unroll
              for (int nb = 0; nb < Rgb::HB; ++nb) {
                f32x4 sum;
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] = ps_row16_sum(dh[pb][4 * nb + r]);
                if (j == 0 && blk * 16 < a.N) *reinterpret_cast<f32x4*>(a.dr_part + blk * (Rgb::HB * 16) + 16 * nb + 4 * g) = sum;
              }
            }
          });
    } else {
      mlp_backward_acc<Rgb, PB, true, true>(
          pk, scratch, acc,
          [&](float (&cin)[PB][12]) {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
              float sh[16];
              sh4((dirv[pb][0] + 1.0f) / 2.0f, (dirv[pb][1] + 1.0f) / 2.0f, (dirv[pb][2] + 1.0f) / 2.0f, sh);
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const float v0 = sh[4 * t], v1 = sh[4 * t + 1], v2 = sh[4 * t + 2], v3 = sh[4 * t + 3];
                cin[pb][t] = g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3));
                cin[pb][4 + t] = zb0[pb][t];
                cin[pb][8 + t] = appv[pb][t];
              }
            }
          },
          c1, c2, co, dcin, tm);
    }
    __builtin_amdgcn_sched_barrier(0);
    consume();
    fetch_head(first + 2 * stride, hd);
    __builtin_amdgcn_sched_barrier(0);
    PS_STAMP(tm, 8)
    // d(appearance) is per RAY: see main_bwd_kernel
    float dz0[PB][4];
    bool block_in_ray = (a.S % 16) == 0;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t p = first + pb * 16 + j;
      const int64_t op = orig_index<MS>(a.perm, p, a.N);
      bool pt_ok = p < a.N;
      if constexpr (C::FACT) {
#pragma unroll
        for (int t = 0; t < 4; ++t) dz0[pb][t] = dcin[pb][t];  // geo slots (the sigma_raw slot has zero weights -> exactly 0)
      } else {
        if constexpr (MS) {
          pt_ok = op >= 0;
          const int rid = pt_ok ? (int)ray_of[pb] : -1;
          bool same = true;
          same &= rid == __builtin_amdgcn_update_dpp(0, rid, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
          same &= rid == __builtin_amdgcn_update_dpp(0, rid, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
          same &= rid == __builtin_amdgcn_update_dpp(0, rid, 0x141, 0xF, 0xF, false);  // row_half_mirror
          same &= rid == __builtin_amdgcn_update_dpp(0, rid, 0x140, 0xF, 0xF, false);  // row_mirror
          const unsigned long long ok = __ballot(same && pt_ok);
          block_in_ray = ((ok >> (16 * g)) & 0xffffull) == 0xffffull;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          dz0[pb][t] = dcin[pb][4 + t];  // geo slots (the sigma_raw slot has zero weights -> exactly 0)
          if (MS && a.dapp_pt != nullptr) {
            const int c = 4 * t + g;
            if (pt_ok && c < a.A) a.dapp_pt[op * a.A + c] = dcin[pb][8 + t];
          } else if (a.dapp != nullptr) {
            const int c = 4 * t + g;
            float v = pt_ok ? dcin[pb][8 + t] : 0.0f;
            float vs = v;
            if constexpr (MS) vs = ps_row16_sum(v);
            if (block_in_ray) {
              if constexpr (!MS) vs = ps_row16_sum(v);
              if (j == 0 && first + pb * 16 < a.N && c < a.A) unsafeAtomicAdd(a.dapp + ray_of[pb] * a.A + c, vs);
            } else if (pt_ok && c < a.A) {
              unsafeAtomicAdd(a.dapp + ray_of[pb] * a.A + c, v);
            }
          }
        }
      }
      dz0[pb][0] += ds_cur[pb] * trunc_exp_grad(zb0[pb][0]);  // (ds is 0 outside lane group 0 / past the end: fetched with the head)
    }
    store_act<1, PB>(a.dzb, C::DZB_W, 0, first, a.N, dz0);
    PS_STAMP(tm, 9)
  }
#if defined(PS_TIMING)
  if (lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_ps_timing_rgb[i], tm_.acc[i]);
#endif
  reduce_store<Rgb, NW>(lds, acc, a.gpart + (size_t)lb * C::GPACKED + C::G_RGB);
}

template <class C, int PB, int NW, bool MS>
__global__ __launch_bounds__(NW * 64) void main_bwd_base_kernel(MainArgs a) {
  using Base = typename C::Base;
  using L = OneStackLds<Base, NW>;
  TileRange tr;
  int lb;
  if (!main_bwd_setup<MS>(a, tr, lb)) return;
  __shared__ __attribute__((aligned(16))) float lds[L::FLOATS];
  load_transposed<Base, NW>(lds, a.packed + C::P_BASE);
  const int wave = threadIdx.x >> 6;
  float* scratch = lds + L::HEAD + wave * L::SCR;
  const LdsW pk{lds - Base::FW};
  MlpAcc<Base> acc;
  acc.zero();
  FeatCols<Base::KS0> fc;
  fc.init(a.plane_stride, a.LF, a.F);
  // d(output) one tile ahead, the rest under the first matrix phase (see main_bwd_sem_kernel)
  const int64_t stride = (int64_t)tr.n * NW * 16 * PB;
  int64_t first = tr.first_pt + ((int64_t)tr.j * NW + wave) * 16 * PB;
  // (consumed at the end of the previous tile, before its stores: see main_bwd_sem_kernel)
  constexpr int DZ = C::DZB_W / 4;  // registers of one point block's d(base output) (+ FACT: d(hidden) from the semantic stack)
  float dzb_next[PB][DZ], dzb[PB][DZ];
  auto consume = [&]() {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int t = 0; t < DZ; ++t) {
        dzb[pb][t] = dzb_next[pb][t];
        asm volatile("" : "+v"(dzb[pb][t]));  // a real copy, made HERE (the wait for the load must not sink below the stores)
      }
  };
  load_act<DZ / 4, PB, C::FACT || MS>(a.dzb, C::DZB_W, 0, first, a.N, dzb_next);
  consume();
  load_act<DZ / 4, PB, C::FACT || MS>(a.dzb, C::DZB_W, 0, first + stride, a.N, dzb_next);
  for (; first < a.N; first += stride) {
    float h1[PB][Base::HB * 4], xin[PB][Base::KS0];
    load_act<Base::HB, PB, C::FACT || MS>(a.acts, C::ACT_W, C::ACT_H1, first, a.N, h1);
    if constexpr (C::FACT || MS) {  // whole tiles (the routed layout pads every sub-field to whole chunks), every feature column exists
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const float* row = a.feat + (first + pb * 16 + (ps_lane() & 15)) * a.F;
#pragma unroll
        for (int t = 0; t < Base::KS0; ++t) xin[pb][t] = row[fc.off[t]];
      }
    } else {
      load_feat<Base::KS0, PB>(a.feat, fc, a.F, first, a.N, xin);
    }
    float dx[PB][Base::L0::IB * 4];
    auto make_x = [&](float (&x)[PB][Base::KS0]) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int t = 0; t < Base::KS0; ++t) x[pb][t] = xin[pb][t];
    };
    if constexpr (C::MERGED) {
      float dz[PB][4];
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int t = 0; t < 4; ++t) dz[pb][t] = dzb[pb][t];
      // the semantic stack's gradient joins d(hidden) between the 16-wide output layer and the ReLU mask
      mlp_backward_acc<Base, PB, true>(pk, scratch, acc, make_x, h1, h1, dz, dx, nullptr, [&](float (&dh)[PB][Base::HB * 4]) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int t = 0; t < Base::HB * 4; ++t) dh[pb][t] += dzb[pb][4 + t];
      });
    } else {
      mlp_backward_acc<Base, PB, true>(pk, scratch, acc, make_x, h1, h1, dzb, dx);
    }
    __builtin_amdgcn_sched_barrier(0);
    consume();
    load_act<DZ / 4, PB, C::FACT || MS>(a.dzb, C::DZB_W, 0, first + 2 * stride, a.N, dzb_next);
    __builtin_amdgcn_sched_barrier(0);
    store_dfeat<Base::KS0, PB>(a.dfeat, fc, a.F, first, a.N, dx);
  }
  reduce_store<Base, NW>(lds, acc, a.gpart + (size_t)lb * C::GPACKED + C::G_BASE);
}

// Workgroups of the main field's backward kernels (and partial gradient blocks their callers reduce).  256 = one persistent workgroup
// per compute unit; PS_MAIN_BWD_BLOCKS (a multiple of 8) deals more, shorter ones: a kernel of exactly 256 workgroups that finds k
// compute units taken -- by the proposal chain on its side stream -- runs its last k workgroups in a second round, i.e. twice as long.
// Measured (tools/ab_env.sh, one box, alternating): cfg 2 13.30 / 13.47 ms at 256, 13.78 / 13.77 at 512, 13.75 / 13.73 at 1024; cfg 3
// 23.8 / 24.2 / 24.5 ms -- the shorter workgroups keep asking for compute units and starve the side chain instead; 256 stays.
#ifndef PS_MAIN_BWD_BLOCKS_DEFAULT
#define PS_MAIN_BWD_BLOCKS_DEFAULT 256
#endif
int main_bwd_blocks() {
  static const int n = [] {
    const char* e = getenv("PS_MAIN_BWD_BLOCKS");
    int v = e != nullptr ? atoi(e) : PS_MAIN_BWD_BLOCKS_DEFAULT;
    if (v < 8) v = 8;
    return v / 8 * 8;
  }();
  return n;
}

int grid_for_tiles_nw(int64_t N, int pts_per_tile, int waves, int max_blocks) {
  const int64_t tiles = (N + pts_per_tile - 1) / pts_per_tile;
  int64_t g = (tiles + waves - 1) / waves;
  if (g > max_blocks) g = max_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

int grid_for_tiles(int64_t N, int pts_per_tile, int max_blocks) {
  const int64_t tiles = (N + pts_per_tile - 1) / pts_per_tile;
  int64_t g = (tiles + 3) / 4;
  if (g > max_blocks) g = max_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

// multi-sub-field launch: every sub-field needs at least one workgroup (ms_block), the rest is dealt by size
int ms_grid(int64_t n_slots, int pts_per_tile, int waves, int max_blocks, int K) {
  const int64_t tiles = (n_slots + pts_per_tile - 1) / pts_per_tile;
  int64_t g = (tiles + waves - 1) / waves + K;
  if (g > max_blocks) g = max_blocks;
  if (g < K) g = K;
  return (int)((g + 7) / 8 * 8);  // ms_logical_block deals the workgroups XCD-major
}

#ifndef PS_PROP_FWD_PB
#define PS_PROP_FWD_PB 2  // (4: 134 us per launch, 2: 123 us)
#endif
#ifndef PS_PROP_BWD_BLOCKS
#define PS_PROP_BWD_BLOCKS 512
#endif
#ifndef PS_PROP_BWD_PB
#define PS_PROP_BWD_PB 2
#endif
constexpr int kPropFwdPB = PS_PROP_FWD_PB, kPropBwdPB = PS_PROP_BWD_PB, kMainFwdPB = 2, kMainFwdWaves = 4, kMainBwdPB = PS_MAIN_BWD_PB, kMainBwdWaves = PS_MAIN_BWD_WAVES;
// Shape of the main forward launch.  The kernel holds a wave's activations in registers: with two 16-point blocks per wave it needs
// > 256 registers (one wave per SIMD, one workgroup per CU) and every LDS operand read, activation store and non-matrix instruction
// of that wave stalls the matrix pipe of its SIMD.  Where TWO workgroups fit a CU's 160 KiB of LDS (the factored / merged one-sub-field
// stacks: 69-73 KiB each) the kernel runs ONE point block per wave (184-194 registers) in twice as many workgroups: two waves per SIMD
// cover each other's stalls -- cfg 2 main_fwd_kernel 1.76 -> 1.59 ms.  (The three backward kernels need >= 2 point blocks for their
// pipelined operand staging and their weight-gradient accumulators, and stay at one wave per SIMD.)
#ifndef PS_MAIN_FWD_TWO_WG
#define PS_MAIN_FWD_TWO_WG 1
#endif
template <class C>
struct MainFwdShape {
  // (two workgroups per CU: the training node of one sub-field only -- the other stacks' weight fragments fill > 80 KiB)
  static constexpr bool kTwo = PS_MAIN_FWD_TWO_WG && C::FACT && 2 * (C::FW * 4 + 2048) <= 160 * 1024;
  // everything else: the same two waves per SIMD as ONE workgroup of eight one-block waves (PS_MAIN_FWD_TWO_WG = 0: the old shape)
  static constexpr int kPB = PS_MAIN_FWD_TWO_WG ? 1 : kMainFwdPB;
  static constexpr int kWaves = (PS_MAIN_FWD_TWO_WG && !kTwo) ? 8 : kMainFwdWaves;
  static constexpr int kBlocks = kTwo ? 512 : 256;
};
constexpr int kPropBwdBlocks = PS_PROP_BWD_BLOCKS;  // 2 workgroups per CU: the kernel is latency bound and its registers allow 2 waves/SIMD

// (L*F, hidden) of the proposal nets
#define PS_PROP_CFGS(X) \
  X(8, 64)              \
  X(2, 32)
// (L*F, hidden, hidden_color) of the main field
#define PS_MAIN_CFGS(X) \
  X(32, 64, 64)         \
  X(40, 64, 64)         \
  X(4, 32, 32)

}  // namespace

extern "C" int ps_prop_field_sizes(int LF, int hidden, int64_t N, int64_t* packed_floats, int64_t* grad_floats, int* n_parts) {
#define X(lf, h)                                            \
  if (LF == lf && hidden == h) {                            \
    using M = ps::MlpT<(lf + 3) / 4, h / 16, 1, 2>;         \
    *packed_floats = M::PACKED;                             \
    *grad_floats = M::GPACKED;                              \
    *n_parts = 4 * grid_for_tiles(N, 16 * kPropBwdPB, kPropBwdBlocks); \
    return 0;                                               \
  }
  PS_PROP_CFGS(X)
#undef X
  ps_set_error("ps_prop_field: unsupported (L*F, hidden)");
  return -2;
}

namespace {
int prop_fwd_impl(const float* feat, int64_t plane_stride, int LF, int F, int hidden, const float* sel, const float* packed, int64_t N,
                  float* sigma, const int* perm, const int* field_start, int K, hipStream_t s) {
  if (N == 0) return 0;
#define X(lf, h)                                                                                                     \
  if (LF == lf && hidden == h) {                                                                                     \
    using M = ps::MlpT<(lf + 3) / 4, h / 16, 1, 2>;                                                                  \
    if (perm != nullptr)                                                                                             \
      prop_fwd_kernel<M, kPropFwdPB, true><<<ms_grid(N, 16 * kPropFwdPB, 4, 1024, K), 256, 0, s>>>(                    \
          feat, plane_stride, LF, F, sel, packed, N, sigma, perm, field_start, K);                                   \
    else                                                                                                             \
      prop_fwd_kernel<M, kPropFwdPB, false><<<grid_for_tiles(N, 16 * kPropFwdPB, 1024), 256, 0, s>>>(                  \
          feat, plane_stride, LF, F, sel, packed, N, sigma, nullptr, nullptr, 1);                                    \
    PS_CHECK_LAUNCH();                                                                                               \
  }
  PS_PROP_CFGS(X)
#undef X
  ps_set_error("ps_prop_field_fwd: unsupported (L*F, hidden)");
  return -2;
}

int prop_bwd_impl(const float* feat, int64_t plane_stride, int LF, int F, int hidden, const float* sel, const float* packed,
                  const float* dsigma, int64_t N, float* dfeat, float* gpart, uint32_t* level_absmax, const int* perm,
                  const int* field_start, int K, hipStream_t s) {
  if (N == 0) return 0;
  if (level_absmax != nullptr) {
    hipError_t e = hipMemsetAsync(level_absmax, 0, (size_t)K * (LF / F) * 4, s);
    if (e != hipSuccess) { ps_set_error(hipGetErrorString(e)); return (int)e; }
  }
#define X(lf, h)                                                                                                     \
  if (LF == lf && hidden == h) {                                                                                     \
    using M = ps::MlpT<(lf + 3) / 4, h / 16, 1, 2>;                                                                  \
    if (perm != nullptr)                                                                                             \
      prop_bwd_kernel<M, kPropBwdPB, true><<<ms_grid(N, 16 * kPropBwdPB, 4, kPropBwdBlocks, K), 256, 0, s>>>(          \
          feat, plane_stride, LF, F, sel, packed, dsigma, N, dfeat, gpart, level_absmax, perm, field_start, K);      \
    else                                                                                                             \
      prop_bwd_kernel<M, kPropBwdPB, false><<<grid_for_tiles(N, 16 * kPropBwdPB, kPropBwdBlocks), 256, 0, s>>>(        \
          feat, plane_stride, LF, F, sel, packed, dsigma, N, dfeat, gpart, level_absmax, nullptr, nullptr, 1);       \
    PS_CHECK_LAUNCH();                                                                                               \
  }
  PS_PROP_CFGS(X)
#undef X
  ps_set_error("ps_prop_field_bwd: unsupported (L*F, hidden)");
  return -2;
}
}  // namespace

extern "C" int ps_prop_field_fwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, const float* sel,
                                 const float* packed, int64_t N, float* sigma, void* stream) {
  return prop_fwd_impl(feat, plane_stride, LF, F, hidden, sel, packed, N, sigma, nullptr, nullptr, 1, (hipStream_t)stream);
}

extern "C" int ps_prop_field_bwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, const float* sel,
                                 const float* packed, const float* dsigma, int64_t N, float* dfeat, float* gpart,
                                 uint32_t* level_absmax, void* stream) {
  return prop_bwd_impl(feat, plane_stride, LF, F, hidden, sel, packed, dsigma, N, dfeat, gpart, level_absmax, nullptr, nullptr, 1,
                       (hipStream_t)stream);
}

// ---- multi-sub-field launches (ms_core.hpp): n_slots = slots of the sorted layout, packed = K packed blocks back to back,
// sigma / dsigma in the CALLER's point order (reached through perm), gpart = ps_*_field_parts_ms partial blocks
extern "C" int ps_prop_field_parts_ms(int64_t n_slots, int K) { return 4 * ms_grid(n_slots, 16 * kPropBwdPB, 4, kPropBwdBlocks, K); }
extern "C" int ps_main_field_parts_ms(int64_t n_slots, int K) { return ms_grid(n_slots, 16 * kMainBwdPB, kMainBwdWaves, main_bwd_blocks(), K); }

extern "C" int ps_prop_field_fwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, const float* sel,
                                    const float* packed, int64_t n_slots, float* sigma, const int32_t* perm,
                                    const int32_t* field_start, int K, void* stream) {
  PS_REQUIRE(perm != nullptr && field_start != nullptr && K >= 1, "ps_prop_field_fwd_ms: need the sorted layout");
  return prop_fwd_impl(feat, plane_stride, LF, F, hidden, sel, packed, n_slots, sigma, perm, field_start, K, (hipStream_t)stream);
}

extern "C" int ps_prop_field_bwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, const float* sel,
                                    const float* packed, const float* dsigma, int64_t n_slots, float* dfeat, float* gpart,
                                    uint32_t* level_absmax, const int32_t* perm, const int32_t* field_start, int K, void* stream) {
  PS_REQUIRE(perm != nullptr && field_start != nullptr && K >= 1, "ps_prop_field_bwd_ms: need the sorted layout");
  return prop_bwd_impl(feat, plane_stride, LF, F, hidden, sel, packed, dsigma, n_slots, dfeat, gpart, level_absmax, perm, field_start, K,
                       (hipStream_t)stream);
}

extern "C" int ps_main_field_sizes(int LF, int hidden, int hidden_color, int64_t N, int64_t* packed_floats,
                                   int64_t* grad_floats, int* n_parts, int64_t* offsets /*[6]: P_SEM,P_RGB,G_SEM,G_RGB,..*/) {
#define X(lf, h, hc)                                               \
  if (LF == lf && hidden == h && hidden_color == hc) {             \
    using C = MainCfg<(lf + 3) / 4, h / 16, hc / 16>;              \
    *packed_floats = C::PACKED;                                    \
    *grad_floats = C::GPACKED;                                     \
    *n_parts = grid_for_tiles_nw(N, 16 * kMainBwdPB, kMainBwdWaves, main_bwd_blocks()); \
    if (offsets) {                                                 \
      offsets[0] = C::P_BASE;                                      \
      offsets[1] = C::P_SEM;                                       \
      offsets[2] = C::P_RGB;                                       \
      offsets[3] = C::G_BASE;                                      \
      offsets[4] = C::G_SEM;                                       \
      offsets[5] = C::G_RGB;                                       \
    }                                                              \
    return 0;                                                      \
  }
  PS_MAIN_CFGS(X)
#undef X
  ps_set_error("ps_main_field: unsupported (L*F, hidden, hidden_color)");
  return -2;
}

// floats per point of the kept-activation buffer (ps_main_field_fwd `acts`), 0 if the shape is unsupported
extern "C" int ps_main_field_act_width(int LF, int hidden, int hidden_color) {
#define X(lf, h, hc) \
  if (LF == lf && hidden == h && hidden_color == hc) return MainCfg<(lf + 3) / 4, h / 16, hc / 16>::ACT_W;
  PS_MAIN_CFGS(X)
#undef X
  return 0;
}

// the factored semantic path (MainCfg<..., FACT = true>): sizes of its packed / gradient blocks, kept activations and workspace
extern "C" int ps_main_field_f_sizes(int LF, int hidden, int hidden_color, int64_t N, int64_t* packed_floats, int64_t* grad_floats,
                                     int* n_parts, int64_t* offsets /*[6]*/, int* act_width, int* dzb_width) {
#define X(lf, h, hc)                                               \
  if (LF == lf && hidden == h && hidden_color == hc) {             \
    using C = MainCfg<(lf + 3) / 4, h / 16, hc / 16, true>;        \
    *packed_floats = C::PACKED;                                    \
    *grad_floats = C::GPACKED;                                     \
    *n_parts = grid_for_tiles_nw(N, 16 * kMainBwdPB, kMainBwdWaves, main_bwd_blocks()); \
    if (offsets) {                                                 \
      offsets[0] = C::P_BASE;                                      \
      offsets[1] = C::P_SEM;                                       \
      offsets[2] = C::P_RGB;                                       \
      offsets[3] = C::G_BASE;                                      \
      offsets[4] = C::G_SEM;                                       \
      offsets[5] = C::G_RGB;                                       \
    }                                                              \
    if (act_width) *act_width = C::ACT_W;                          \
    if (dzb_width) *dzb_width = C::DZB_W;                          \
    return 0;                                                      \
  }
  PS_MAIN_CFGS(X)
#undef X
  ps_set_error("ps_main_field_f: unsupported (L*F, hidden, hidden_color)");
  return -2;
}

namespace {
int main_fwd_impl(MainArgs a, int hidden, int hidden_color, hipStream_t s, bool fact = false, bool merged = false) {
  if (a.N == 0) return 0;
  if (merged) {  // MainCfg MERGE_: base output rows 16..79 folded into the semantic head's first layer (one or K sub-fields)
    PS_REQUIRE(a.sem != nullptr && a.sigma != nullptr, "ps_main_field (merged network): density + semantics are always produced");
    PS_REQUIRE(a.gate_a == nullptr || (a.acts == nullptr && a.rgb == nullptr), "ps_main_field_fwd_gated: inference (no kept activations, no colour)");
    PS_REQUIRE(a.acts == nullptr || a.rgb != nullptr, "ps_main_field (merged network): activations are kept for full evaluations only");
    PS_REQUIRE(a.A <= 16 && a.S > 0, "ps_main_field (merged network): appearance dim must be <= 16");
    PS_REQUIRE(a.N < (int64_t(1) << 31), "ps_main_field (merged network): at most 2^31 - 1 points per call");
    // Gated routed launches are dealt 8 x more (shorter) workgroups than the chip holds.  The workgroups of a sub-field share its tiles
    // round-robin, but the SUB-FIELDS are dealt workgroups by point count while the semantic head (3.4 x the base MLP's matrix work)
    // runs only where the dense points are: on a lattice slab where 0.4 % of the tiles pass the gate, all in one small sub-field, the
    // launch took 2.05 ms against 0.73 ms with the gate shut (tools/dbg/gate_probe.py) -- its few workgroups ran alone at the end.
    // With short workgroups the hardware dispatcher fills the compute units the light sub-fields leave early.
    static const char* gate_env = getenv("PS_GATE_BLOCKS");
    const int gate_blocks = a.gate_a != nullptr ? (gate_env != nullptr ? std::max(1, atoi(gate_env)) : 8) : 1;  // (clamped: 0 / negative would void the block cap)
#define X(lf, h, hc)                                                                                                  \
  if (a.LF == lf && hidden == h && hidden_color == hc) {                                                              \
    using C = MainCfg<(lf + 3) / 4, h / 16, hc / 16, false, true>;                                                    \
    if (a.perm != nullptr) {                                                                                          \
      a.packed_stride = C::PACKED;                                                                                    \
      main_fwd_kernel<C, MainFwdShape<C>::kPB, MainFwdShape<C>::kWaves, true><<<ms_grid(a.N, 16 * MainFwdShape<C>::kPB, MainFwdShape<C>::kWaves, gate_blocks * MainFwdShape<C>::kBlocks, a.K), MainFwdShape<C>::kWaves * 64, 0, s>>>(a); \
    } else {                                                                                                          \
      main_fwd_kernel<C, MainFwdShape<C>::kPB, MainFwdShape<C>::kWaves, false><<<grid_for_tiles_nw(a.N, 16 * MainFwdShape<C>::kPB, MainFwdShape<C>::kWaves, MainFwdShape<C>::kBlocks), MainFwdShape<C>::kWaves * 64, 0, s>>>(a); \
    }                                                                                                                 \
    PS_CHECK_LAUNCH();                                                                                                \
  }
    PS_MAIN_CFGS(X)
#undef X
    ps_set_error("ps_main_field (merged network): unsupported (L*F, hidden, hidden_color)");
    return -2;
  }
  if (fact) {
    PS_REQUIRE(a.perm == nullptr && a.hid_ray != nullptr && a.rgb != nullptr && a.sigma != nullptr,
               "ps_main_field_f_fwd: one sub-field, all outputs");
#define X(lf, h, hc)                                                                                                  \
  if (a.LF == lf && hidden == h && hidden_color == hc) {                                                              \
    using C = MainCfg<(lf + 3) / 4, h / 16, hc / 16, true>;                                                           \
    main_fwd_kernel<C, MainFwdShape<C>::kPB, MainFwdShape<C>::kWaves, false><<<grid_for_tiles_nw(a.N, 16 * MainFwdShape<C>::kPB, MainFwdShape<C>::kWaves, MainFwdShape<C>::kBlocks), MainFwdShape<C>::kWaves * 64, 0, s>>>(a); \
    PS_CHECK_LAUNCH();                                                                                                \
  }
    PS_MAIN_CFGS(X)
#undef X
    ps_set_error("ps_main_field_f_fwd: unsupported (L*F, hidden, hidden_color)");
    return -2;
  }
  PS_REQUIRE(a.A <= 16 && a.S > 0, "ps_main_field_fwd: appearance dim must be <= 16");
  PS_REQUIRE(a.N < (int64_t(1) << 31), "ps_main_field_fwd: at most 2^31 - 1 points per call");
  PS_REQUIRE(a.acts == nullptr || (a.sem != nullptr && a.rgb != nullptr), "ps_main_field_fwd: activations are kept for full evaluations only");
#define X(lf, h, hc)                                                                                                  \
  if (a.LF == lf && hidden == h && hidden_color == hc) {                                                              \
    using C = MainCfg<(lf + 3) / 4, h / 16, hc / 16>;                                                                 \
    if (a.perm != nullptr) {                                                                                          \
      a.packed_stride = C::PACKED;                                                                                    \
      main_fwd_kernel<C, MainFwdShape<C>::kPB, MainFwdShape<C>::kWaves, true><<<ms_grid(a.N, 16 * MainFwdShape<C>::kPB, MainFwdShape<C>::kWaves, MainFwdShape<C>::kBlocks, a.K), MainFwdShape<C>::kWaves * 64, 0, s>>>(a); \
    } else {                                                                                                          \
      main_fwd_kernel<C, MainFwdShape<C>::kPB, MainFwdShape<C>::kWaves, false><<<grid_for_tiles_nw(a.N, 16 * MainFwdShape<C>::kPB, MainFwdShape<C>::kWaves, MainFwdShape<C>::kBlocks), MainFwdShape<C>::kWaves * 64, 0, s>>>(a); \
    }                                                                                                                 \
    PS_CHECK_LAUNCH();                                                                                                \
  }
  PS_MAIN_CFGS(X)
#undef X
  ps_set_error("ps_main_field_fwd: unsupported (L*F, hidden, hidden_color)");
  return -2;
}

// stages: which kernels of the three-kernel backward this call launches (bit 0 semantic head, 1 colour head, 2 base MLP)
int main_bwd_impl(MainArgs a, int hidden, int hidden_color, int stages, hipStream_t s, bool fact = false, bool merged = false) {
  const int st = stages & 7;
  if (a.N == 0) return 0;
  if (merged) {  // MainCfg MERGE_, three-kernel backward on kept activations (one or K sub-fields)
    PS_REQUIRE(a.acts != nullptr && a.dzb != nullptr && a.drgb != nullptr && a.dsem != nullptr,
               "ps_main_field_m_bwd: kept activations, the workspace and both head gradients are required");
    PS_REQUIRE(a.A <= 16 && a.S > 0 && a.N < (int64_t(1) << 31), "ps_main_field_m_bwd: appearance dim <= 16, at most 2^31 - 1 points");
#define X(lf, h, hc)                                                                                                  \
  if (a.LF == lf && hidden == h && hidden_color == hc) {                                                              \
    using C = MainCfg<(lf + 3) / 4, h / 16, hc / 16, false, true>;                                                    \
    if (a.perm != nullptr) {                                                                                          \
      a.packed_stride = C::PACKED;                                                                                    \
      const int grid = ms_grid(a.N, 16 * kMainBwdPB, kMainBwdWaves, main_bwd_blocks(), a.K);                                        \
      if (st & 1) main_bwd_sem_kernel<C, kMainBwdPB, kMainBwdWaves, true><<<grid, kMainBwdWaves * 64, 0, s>>>(a);    \
      if (st & 2) main_bwd_rgb_kernel<C, kMainBwdPB, kMainBwdWaves, true><<<grid, kMainBwdWaves * 64, 0, s>>>(a);    \
      if (st & 4) main_bwd_base_kernel<C, kMainBwdPB, kMainBwdWaves, true><<<grid, kMainBwdWaves * 64, 0, s>>>(a);   \
    } else {                                                                                                          \
      const int grid = grid_for_tiles_nw(a.N, 16 * kMainBwdPB, kMainBwdWaves, main_bwd_blocks());                                   \
      if (st & 1) main_bwd_sem_kernel<C, kMainBwdPB, kMainBwdWaves, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a);   \
      if (st & 2) main_bwd_rgb_kernel<C, kMainBwdPB, kMainBwdWaves, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a);   \
      if (st & 4) main_bwd_base_kernel<C, kMainBwdPB, kMainBwdWaves, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a);  \
    }                                                                                                                 \
    PS_CHECK_LAUNCH();                                                                                                \
  }
    PS_MAIN_CFGS(X)
#undef X
    ps_set_error("ps_main_field_m_bwd: unsupported (L*F, hidden, hidden_color)");
    return -2;
  }
  if (fact) {
    PS_REQUIRE(a.perm == nullptr && a.acts != nullptr && a.dzb != nullptr && a.w != nullptr && a.drgb != nullptr && a.dsem != nullptr,
               "ps_main_field_f_bwd: one sub-field, kept activations, the workspace and per-ray gradients are required");
#define X(lf, h, hc)                                                                                                  \
  if (a.LF == lf && hidden == h && hidden_color == hc) {                                                              \
    using C = MainCfg<(lf + 3) / 4, h / 16, hc / 16, true>;                                                           \
    const int grid = grid_for_tiles_nw(a.N, 16 * kMainBwdPB, kMainBwdWaves, main_bwd_blocks());                                     \
    if (st & 1) main_bwd_sem_kernel<C, kMainBwdPB, kMainBwdWaves, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a);      \
    if (st & 2) main_bwd_rgb_kernel<C, kMainBwdPB, kMainBwdWaves, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a);      \
    if (st & 4) main_bwd_base_kernel<C, kMainBwdPB, kMainBwdWaves, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a);     \
    PS_CHECK_LAUNCH();                                                                                                \
  }
    PS_MAIN_CFGS(X)
#undef X
    ps_set_error("ps_main_field_f_bwd: unsupported (L*F, hidden, hidden_color)");
    return -2;
  }
  PS_REQUIRE(a.acts == nullptr || (a.drgb != nullptr && a.dsem != nullptr), "ps_main_field_bwd: kept activations need both head gradients");
  PS_REQUIRE(a.A <= 16 && a.S > 0, "ps_main_field_bwd: appearance dim must be <= 16");
  PS_REQUIRE(a.N < (int64_t(1) << 31), "ps_main_field_bwd: at most 2^31 - 1 points per call");
#define X(lf, h, hc)                                                                                                  \
  if (a.LF == lf && hidden == h && hidden_color == hc) {                                                              \
    using C = MainCfg<(lf + 3) / 4, h / 16, hc / 16>;                                                                 \
    if (a.perm != nullptr) {                                                                                          \
      a.packed_stride = C::PACKED;                                                                                    \
      const int grid = ms_grid(a.N, 16 * kMainBwdPB, kMainBwdWaves, main_bwd_blocks(), a.K);                                        \
      if (a.acts != nullptr && a.dzb != nullptr) {                                                                    \
        if (st & 1) main_bwd_sem_kernel<C, kMainBwdPB, kMainBwdWaves, true><<<grid, kMainBwdWaves * 64, 0, s>>>(a);  \
        if (st & 2) main_bwd_rgb_kernel<C, kMainBwdPB, kMainBwdWaves, true><<<grid, kMainBwdWaves * 64, 0, s>>>(a);  \
        if (st & 4) main_bwd_base_kernel<C, kMainBwdPB, kMainBwdWaves, true><<<grid, kMainBwdWaves * 64, 0, s>>>(a); \
      } else if (a.acts != nullptr)                                                                                   \
        main_bwd_kernel<C, kMainBwdPB, kMainBwdWaves, true, true><<<grid, kMainBwdWaves * 64, 0, s>>>(a);             \
      else                                                                                                            \
        main_bwd_kernel<C, kMainBwdPB, kMainBwdWaves, false, true><<<grid, kMainBwdWaves * 64, 0, s>>>(a);            \
    } else {                                                                                                          \
      const int grid = grid_for_tiles_nw(a.N, 16 * kMainBwdPB, kMainBwdWaves, main_bwd_blocks());                                   \
      if (a.acts != nullptr && a.dzb != nullptr) {                                                                    \
        if (st & 1) main_bwd_sem_kernel<C, kMainBwdPB, kMainBwdWaves, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a);  \
        if (st & 2) main_bwd_rgb_kernel<C, kMainBwdPB, kMainBwdWaves, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a);  \
        if (st & 4) main_bwd_base_kernel<C, kMainBwdPB, kMainBwdWaves, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a); \
      } else if (a.acts != nullptr)                                                                                   \
        main_bwd_kernel<C, kMainBwdPB, kMainBwdWaves, true, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a);            \
      else                                                                                                            \
        main_bwd_kernel<C, kMainBwdPB, kMainBwdWaves, false, false><<<grid, kMainBwdWaves * 64, 0, s>>>(a);           \
    }                                                                                                                 \
    PS_CHECK_LAUNCH();                                                                                                \
  }
  PS_MAIN_CFGS(X)
#undef X
  ps_set_error("ps_main_field_bwd: unsupported (L*F, hidden, hidden_color)");
  return -2;
}
}  // namespace

extern "C" int ps_main_field_fwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                                 const float* sel, const float* dirs, const float* app, int S, int A, const float* packed,
                                 int64_t N, float* sigma, float* rgb, float* sem, float* acts, void* stream) {
  MainArgs a{};
  a.feat = feat; a.plane_stride = plane_stride; a.LF = LF; a.F = F; a.sel = sel; a.dirs = dirs; a.app = app; a.S = S; a.A = A;
  a.packed = packed; a.N = N; a.sigma = sigma; a.rgb = rgb; a.sem = sem; a.acts = acts; a.K = 1;
  return main_fwd_impl(a, hidden, hidden_color, (hipStream_t)stream);
}

// Inference forward with a GATED semantic head (prior extraction, ns/scripts/extract_priors.py:133-150: points whose mean density
// of the three fields stays below the threshold are dropped right after the query): density for every point; the semantic head
// -- 12 288 of the 15 360 MACs of a point in this kernel -- only for the 32-point tiles in which some point has
// (gate_a[n] + gate_b[n] + sigma[n]) / 3 >= gate_threshold.  sem rows of the other tiles are NOT written.  Pass the caller's
// threshold lowered by a few ulp: the caller decides with ps_mean_density's separately rounded arithmetic.
// The kernel runs the MERGED network (MainCfg MERGE_: base output rows 16..79 folded into the semantic head's first layer,
// W' = W_sem0 W_base1[16:], b' = W_sem0 b_base1[16:] + b_sem0 from ps_merge_linear_fwd): `packed` has the layout of
// ps_main_field_gated_sizes.  Densities are bit-identical to ps_main_field_fwd, semantics agree to fp32 rounding.
extern "C" int ps_main_field_fwd_gated(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                                       const float* sel, const float* packed, int64_t N, const float* gate_a, const float* gate_b,
                                       float gate_threshold, float* sigma, float* sem, unsigned long long* gate_stats, void* stream) {
  PS_REQUIRE(gate_a != nullptr && gate_b != nullptr && sigma != nullptr && sem != nullptr, "ps_main_field_fwd_gated: null argument");
  MainArgs a{};
  a.feat = feat; a.plane_stride = plane_stride; a.LF = LF; a.F = F; a.sel = sel; a.S = 1; a.A = 0;
  a.packed = packed; a.N = N; a.sigma = sigma; a.sem = sem; a.K = 1;
  a.gate_a = gate_a; a.gate_b = gate_b; a.gate_thr = gate_threshold; a.gate_stats = gate_stats;
  return main_fwd_impl(a, hidden, hidden_color, (hipStream_t)stream, false, true);
}

// packed / weight-fragment sizes of the gated inference forward: [base (L*F -> hidden -> 16) | semantic head with the merged first
// layer (hidden -> 64 -> 64 -> 64) | colour head (layout of ps_main_field_fwd; unused)]
extern "C" int ps_main_field_gated_sizes(int LF, int hidden, int hidden_color, int64_t* packed_floats, int64_t* offsets /*[3]*/) {
#define X(lf, h, hc)                                               \
  if (LF == lf && hidden == h && hidden_color == hc) {             \
    using C = MainCfg<(lf + 3) / 4, h / 16, hc / 16, false, true>; \
    *packed_floats = C::PACKED;                                    \
    if (offsets) {                                                 \
      offsets[0] = C::P_BASE;                                      \
      offsets[1] = C::P_SEM;                                       \
      offsets[2] = C::P_RGB;                                       \
    }                                                              \
    return 0;                                                      \
  }
  PS_MAIN_CFGS(X)
#undef X
  ps_set_error("ps_main_field_gated: unsupported (L*F, hidden, hidden_color)");
  return -2;
}

// multi-sub-field gated inference forward: sorted layout (n_slots, perm, field_start), `packed` = K blocks of the
// ps_main_field_gated_sizes layout back to back; gate_a / gate_b / sigma / sem in the CALLER's point order
extern "C" int ps_main_field_fwd_gated_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                                          const float* sel, const float* packed, int64_t n_slots, const float* gate_a, const float* gate_b,
                                          float gate_threshold, float* sigma, float* sem, unsigned long long* gate_stats, const int32_t* perm,
                                          const int32_t* field_start, int K, void* stream) {
  PS_REQUIRE(gate_a != nullptr && gate_b != nullptr && sigma != nullptr && sem != nullptr, "ps_main_field_fwd_gated_ms: null argument");
  PS_REQUIRE(perm != nullptr && field_start != nullptr && K >= 1, "ps_main_field_fwd_gated_ms: need the sorted layout");
  MainArgs a{};
  a.feat = feat; a.plane_stride = plane_stride; a.LF = LF; a.F = F; a.sel = sel; a.S = 1; a.A = 0;
  a.packed = packed; a.N = n_slots; a.sigma = sigma; a.sem = sem; a.perm = perm; a.field_start = field_start; a.K = K;
  a.gate_a = gate_a; a.gate_b = gate_b; a.gate_thr = gate_threshold; a.gate_stats = gate_stats;
  return main_fwd_impl(a, hidden, hidden_color, (hipStream_t)stream, false, true);
}

// The MERGED network as a TRAINING path for routed tiles (K >= 1 sub-fields in the sorted layout; DESIGN.md 4.5 rewrite 1 per
// sub-field): packed block per sub-field = [base (L*F -> hidden -> 16) | semantic head with the merged first layer (hidden -> 64 -> 64
// -> 64) | colour head]; same arguments as ps_main_field_fwd_ms / ps_main_field_bwd_ms, kept activations `acts`
// [n_slots, act_width] and the three-kernel backward's workspace `dzb_scratch` [n_slots, dzb_width] with the widths of
// ps_main_field_m_sizes.  The semantic kernel returns d(W'), d(b') of the merged layer in the semantic stack's first gradient slot
// (-> ps_merge_linear_bwd_batch) and the base kernel the 16-row output layer's.
extern "C" int ps_main_field_m_sizes(int LF, int hidden, int hidden_color, int64_t* packed_floats, int64_t* grad_floats,
                                     int64_t* offsets /*[6]: P_BASE,P_SEM,P_RGB,G_BASE,G_SEM,G_RGB*/, int* act_width, int* dzb_width) {
#define X(lf, h, hc)                                               \
  if (LF == lf && hidden == h && hidden_color == hc) {             \
    using C = MainCfg<(lf + 3) / 4, h / 16, hc / 16, false, true>; \
    *packed_floats = C::PACKED;                                    \
    *grad_floats = C::GPACKED;                                     \
    if (offsets) {                                                 \
      offsets[0] = C::P_BASE;                                      \
      offsets[1] = C::P_SEM;                                       \
      offsets[2] = C::P_RGB;                                       \
      offsets[3] = C::G_BASE;                                      \
      offsets[4] = C::G_SEM;                                       \
      offsets[5] = C::G_RGB;                                       \
    }                                                              \
    if (act_width) *act_width = C::ACT_W;                          \
    if (dzb_width) *dzb_width = C::DZB_W;                          \
    return 0;                                                      \
  }
  PS_MAIN_CFGS(X)
#undef X
  ps_set_error("ps_main_field_m: unsupported (L*F, hidden, hidden_color)");
  return -2;
}

extern "C" int ps_main_field_m_fwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                                      const float* sel, const float* dirs, const float* app, int S, int A, const float* packed,
                                      int64_t n_slots, float* sigma, float* rgb, float* sem, float* acts, const int32_t* perm,
                                      const int32_t* field_start, int K, void* stream) {
  PS_REQUIRE(perm != nullptr && field_start != nullptr && K >= 1, "ps_main_field_m_fwd_ms: need the sorted layout");
  MainArgs a{};
  a.feat = feat; a.plane_stride = plane_stride; a.LF = LF; a.F = F; a.sel = sel; a.dirs = dirs; a.app = app; a.S = S; a.A = A;
  a.packed = packed; a.N = n_slots; a.sigma = sigma; a.rgb = rgb; a.sem = sem; a.acts = acts; a.perm = perm; a.field_start = field_start; a.K = K;
  return main_fwd_impl(a, hidden, hidden_color, (hipStream_t)stream, false, true);
}

extern "C" int ps_main_field_m_bwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                                      const float* sel, const float* dirs, const float* app, int S, int A, const float* packed,
                                      const float* dsigma, const float* drgb, const float* dsem, const float* weights, int64_t n_slots,
                                      float* dfeat, float* dapp, float* gpart, const float* acts, float* dzb_scratch, float* dapp_points,
                                      const int32_t* perm, const int32_t* field_start, int K, int stages, void* stream) {
  PS_REQUIRE(perm != nullptr && field_start != nullptr && K >= 1, "ps_main_field_m_bwd_ms: need the sorted layout");
  MainArgs a{};
  a.dzb = dzb_scratch;
  a.dapp_pt = dapp_points;
  a.feat = feat; a.plane_stride = plane_stride; a.LF = LF; a.F = F; a.sel = sel; a.dirs = dirs; a.app = app; a.S = S; a.A = A;
  a.packed = packed; a.N = n_slots; a.dsigma = dsigma; a.drgb = drgb; a.dsem = dsem; a.w = weights; a.dfeat = dfeat; a.dapp = dapp; a.gpart = gpart; a.acts = const_cast<float*>(acts);
  a.perm = perm; a.field_start = field_start; a.K = K;
  return main_bwd_impl(a, hidden, hidden_color, stages, (hipStream_t)stream, false, true);
}

extern "C" int ps_main_field_bwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                                 const float* sel, const float* dirs, const float* app, int S, int A, const float* packed,
                                 const float* dsigma, const float* drgb, const float* dsem, const float* weights, int64_t N,
                                 float* dfeat, float* dapp, float* gpart, const float* acts, float* dzb_scratch, int stages, void* stream) {
  MainArgs a{};
  a.dzb = dzb_scratch;
  a.feat = feat; a.plane_stride = plane_stride; a.LF = LF; a.F = F; a.sel = sel; a.dirs = dirs; a.app = app; a.S = S; a.A = A;
  a.packed = packed; a.N = N; a.dsigma = dsigma; a.drgb = drgb; a.dsem = dsem; a.w = weights; a.dfeat = dfeat; a.dapp = dapp; a.gpart = gpart; a.acts = const_cast<float*>(acts); a.K = 1;
  return main_bwd_impl(a, hidden, hidden_color, stages, (hipStream_t)stream);
}

// multi-sub-field launches: feat / sel / acts / dfeat in the sorted layout (n_slots), everything else in the caller's order
extern "C" int ps_main_field_fwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                                    const float* sel, const float* dirs, const float* app, int S, int A, const float* packed,
                                    int64_t n_slots, float* sigma, float* rgb, float* sem, float* acts, const int32_t* perm,
                                    const int32_t* field_start, int K, void* stream) {
  PS_REQUIRE(perm != nullptr && field_start != nullptr && K >= 1, "ps_main_field_fwd_ms: need the sorted layout");
  MainArgs a{};
  a.feat = feat; a.plane_stride = plane_stride; a.LF = LF; a.F = F; a.sel = sel; a.dirs = dirs; a.app = app; a.S = S; a.A = A;
  a.packed = packed; a.N = n_slots; a.sigma = sigma; a.rgb = rgb; a.sem = sem; a.acts = acts; a.perm = perm; a.field_start = field_start; a.K = K;
  return main_fwd_impl(a, hidden, hidden_color, (hipStream_t)stream);
}

extern "C" int ps_main_field_bwd_ms(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                                    const float* sel, const float* dirs, const float* app, int S, int A, const float* packed,
                                    const float* dsigma, const float* drgb, const float* dsem, const float* weights, int64_t n_slots,
                                    float* dfeat, float* dapp, float* gpart, const float* acts, float* dzb_scratch, float* dapp_points,
                                    const int32_t* perm, const int32_t* field_start, int K, int stages, void* stream) {
  PS_REQUIRE(perm != nullptr && field_start != nullptr && K >= 1, "ps_main_field_bwd_ms: need the sorted layout");
  PS_REQUIRE(dapp_points == nullptr || (dzb_scratch != nullptr && acts != nullptr), "ps_main_field_bwd_ms: per-point d(appearance) belongs to the three-kernel backward");
  MainArgs a{};
  a.dzb = dzb_scratch;
  a.dapp_pt = dapp_points;
  a.feat = feat; a.plane_stride = plane_stride; a.LF = LF; a.F = F; a.sel = sel; a.dirs = dirs; a.app = app; a.S = S; a.A = A;
  a.packed = packed; a.N = n_slots; a.dsigma = dsigma; a.drgb = drgb; a.dsem = dsem; a.w = weights; a.dfeat = dfeat; a.dapp = dapp; a.gpart = gpart; a.acts = const_cast<float*>(acts);
  a.perm = perm; a.field_start = field_start; a.K = K;
  return main_bwd_impl(a, hidden, hidden_color, stages, (hipStream_t)stream);
}

#if defined(PS_TIMING)
extern "C" int ps_debug_timing(unsigned long long* out /*host[16]*/, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ps_timing), sizeof(unsigned long long) * 16);
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_ps_timing), z, sizeof(z));
  }
  return (int)e;
}
extern "C" int ps_debug_timing_rgb(unsigned long long* out /*host[16]*/, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ps_timing_rgb), sizeof(unsigned long long) * 16);
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_ps_timing_rgb), z, sizeof(z));
  }
  return (int)e;
}
extern "C" int ps_debug_timing_fwd(unsigned long long* out /*host[16]*/, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ps_timing_fwd), sizeof(unsigned long long) * 16);
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_ps_timing_fwd), z, sizeof(z));
  }
  return (int)e;
}
#endif

// ---- factored path (MainCfg FACT, csrc/factored.hip): `packed` holds [base (L*F -> hidden -> 16) | semantic stack (hidden -> 64 merged,
// 64 -> 64) | colour head with a geometry-only first layer]; ray_colour [N / S, hidden_color] is the per-ray part of the colour
// head's first layer (ps_ray_colour_fwd).  The forward also RENDERS the semantic branch: from the bin edges ebins [N / S, S + 1] it
// forms the rendering weights (written to `weights` [N]; ps_weights_fwd's formula) and hands out the composited LAST HIDDEN
// activations of the semantic head per ray, sem_hidden_ray [N / S, 64] (-> ps_sem_out_fwd); density [N] and colour [N, 3] stay per
// sample.  Backward: dsem_hidden is W_out^T d(semantics) per ray (ps_sem_out_bwd); stage 1 (semantic kernel) needs only the
// weights and writes dweights_sem [N] = the semantic head's part of d(weights); stages 2 | 4 need d(sigma) (after the composite /
// weights backward that consumes dweights_sem); dray_part [N / 16, hidden_color] receives the per-block sums of the gradient of
// ray_colour (-> ps_ray_colour_bwd).  S % 32 == 0: the tiles of a ray are whole and run back to back on one wavefront.
extern "C" int ps_main_field_f_fwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                                   const float* sel, const float* ray_colour, const float* ebins, int S, const float* packed, int64_t N,
                                   float* sigma, float* rgb, float* weights, float* sem_hidden_ray, float* acts, void* stream) {
  MainArgs a{};
  a.feat = feat; a.plane_stride = plane_stride; a.LF = LF; a.F = F; a.sel = sel; a.rray = ray_colour; a.ebins = ebins; a.S = S;
  a.packed = packed; a.N = N; a.sigma = sigma; a.rgb = rgb; a.w_out = weights; a.hid_ray = sem_hidden_ray; a.acts = acts; a.K = 1;
  PS_REQUIRE(acts != nullptr && ray_colour != nullptr && ebins != nullptr && sigma != nullptr && rgb != nullptr && weights != nullptr &&
                 sem_hidden_ray != nullptr,
             "ps_main_field_f_fwd: the factored path keeps its activations and produces all of its outputs");
  PS_REQUIRE(a.S > 0 && a.S % (16 * kMainFwdPB) == 0 && N % a.S == 0 && N < (int64_t(1) << 31) && LF % 4 == 0,
             "ps_main_field_f_fwd: samples per ray a multiple of 32, whole rays, at most 2^31 - 1 points, L*F a multiple of 4");
  return main_fwd_impl(a, hidden, hidden_color, (hipStream_t)stream, true);
}

extern "C" int ps_main_field_f_bwd(const float* feat, int64_t plane_stride, int LF, int F, int hidden, int hidden_color,
                                   const float* sel, int S, const float* packed, const float* dsigma, const float* drgb,
                                   const float* dsem_hidden, const float* weights, int64_t N, float* dfeat, float* dray_part,
                                   float* dweights_sem, float* gpart, const float* acts, float* dzb_scratch, int stages, void* stream) {
  MainArgs a{};
  a.dzb = dzb_scratch;
  a.feat = feat; a.plane_stride = plane_stride; a.LF = LF; a.F = F; a.sel = sel; a.S = S;
  a.packed = packed; a.N = N; a.dsigma = dsigma; a.drgb = drgb; a.dsem = dsem_hidden; a.w = weights; a.dfeat = dfeat; a.dr_part = dray_part;
  a.dw_sem = dweights_sem; a.gpart = gpart; a.acts = const_cast<float*>(acts); a.K = 1;
  PS_REQUIRE(((stages & 1) == 0 || dweights_sem != nullptr) && ((stages & 2) == 0 || dray_part != nullptr) &&
                 ((stages & 2) == 0 || dsigma != nullptr),
             "ps_main_field_f_bwd: an output / input of a requested stage is missing");
  PS_REQUIRE(a.S > 0 && a.S % (16 * kMainFwdPB) == 0 && N % a.S == 0 && N < (int64_t(1) << 31) && LF % 4 == 0,
             "ps_main_field_f_bwd: samples per ray a multiple of 32, whole rays, at most 2^31 - 1 points, L*F a multiple of 4");
  return main_bwd_impl(a, hidden, hidden_color, stages, (hipStream_t)stream, true);
}
