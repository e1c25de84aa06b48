// Exact-fp32 MFMA building blocks for the tiny MLPs of the NeRF fields (nn.Linear chains,
// ns/field_components/mlp.py:138-174), written for 64-wide wavefronts.
//
// Formulation.  A layer y = W x + b is evaluated "transposed": D = W * X^T with the weights as
// the MFMA A operand and 16 points as the B operand of v_mfma_f32_16x16x4_f32:
//     A[i][k]  lane l holds i = l&15 (output neuron in a block of 16), k = l>>4
//     B[k][j]  lane l holds k = l>>4, j = l&15 (point in a block of 16)
//     D[r][j]  lane l, reg r holds row 4*(l>>4)+r, column j = l&15
// With lane = (j, g) = (l&15, l>>4), a lane owns 4 consecutive neurons 4g..4g+3 of every 16-neuron
// block for "its" point j.  Because a contraction may enumerate k in any order, the D registers of
// one layer are *directly* the B operands of the next one: k-step t = 4*nb + r pairs the neurons
// {16*nb + 4*g + r : g = 0..3}.  Activations therefore never leave registers between layers; only
// the (tiny) weights are re-arranged, once per step, by the pack kernel into "fragment order"
//     wf[nb][t/4][lane][t%4] = W[16*nb + (lane&15)][colmap(t, lane>>4)]
// so that the A operands of FOUR consecutive k-steps are one 16-byte load per lane (one coalesced 1-KiB wave load:
// ds_read_b128 from LDS, buffer_load_dwordx4 from L2 -- a quarter of the load instructions of a per-k-step layout).  `colmap(t, g)` is the torch
// input column that group g supplies at k-step t; for chained layers it is 16*(t/4)+4*g+t%4.
//
// Backward.  dX^T = W^T dY^T uses the same trick with transposed fragments
//     wtf[ib][t/4][lane][t%4] = W[16*(t/4) + 4*(lane>>4) + t%4][colmap(4*ib + (lane&3), (lane&15)>>2)]
// and yields dX in exactly the register layout the forward input had.  dW = dY^T H contracts over
// points, which needs "lane = neuron" operands; both tiles take one trip through a small per-wave
// LDS scratch (written 4 bytes/lane, read back 16 bytes/lane), and the 16x16 dW tiles are
// accumulated in registers over the wave's points, then added into workgroup-level LDS
// accumulators (ds_add_f32) and finally written as one partial per workgroup.
#pragma once
#include <type_traits>
#include "common.hpp"

#ifndef PS_FWD_PB1_SPLIT
#define PS_FWD_PB1_SPLIT 1
#endif

namespace ps {

// ---- compile-time description of one layer -------------------------------------------------
template <int KS_, int NB_>
struct LayerT {
  static constexpr int KS = KS_;             // k-steps over the inputs (4 inputs each)
  static constexpr int NB = NB_;             // 16-neuron output blocks
  static constexpr int IB = (KS_ + 3) / 4;   // 16-row blocks of dX / columns of dW
  static constexpr int KSO = NB_ * 4;        // k-steps over the outputs (backward data)
  // forward block: bias[NB*16] | wf[NB][IB][64][4] (k-steps padded to a multiple of 4);  transposed block: wtf[IB][NB][64][4]
  static constexpr int BIAS_OFF = 0;
  static constexpr int WF_OFF = NB_ * 16;
  static constexpr int FW = WF_OFF + NB_ * IB * 256;
  static constexpr int WT = IB * NB_ * 256;
  // packed gradient block: dW tiles in MFMA D layout ([tile = ob*IB+ib][lane][r]), then the bias gradient
  static constexpr int GW_OFF = 0;
  static constexpr int GB_OFF = NB_ * IB * 256;
  static constexpr int GPACKED = GB_OFF + NB_ * 16;
  // per-wave LDS rows for the dW transposes: the two operands are staged one after the other (H first, then dY in groups
  // of at most 4 output blocks), so the scratch only has to hold the larger of the two
  static constexpr int OBG = NB_ < 4 ? NB_ : 4;
  // H lives in rows [0, IB*16), the current dY group in rows [IB*16, (IB+OBG)*16): the pipelined backward (layer_bwd_pipe) stages
  // the next point block's operands while the current one's fragments are in registers
  static constexpr int SH_ROW0 = 0, SY_ROW0 = IB * 16;
  static constexpr int SCRATCH_ROWS = (IB + OBG) * 16;
};

// profiling builds (-DPS_TIMING): shader-clock time per phase, per wave (s_memtime drains the wave's LDS / scalar queue, so the
// phases are slightly serialised against each other)
struct PsTimer {
  unsigned long long acc[16], last;
  __device__ __forceinline__ void start() {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    last = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void stamp(int i) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    acc[i] += now - last;
    last = now;
  }
};
#if defined(PS_TIMING)
#define PS_STAMP(tm, i) \
  if ((tm) != nullptr) (tm)->stamp(i);
#else
#define PS_STAMP(tm, i)
#endif

constexpr int kScratchLd = 20;  // floats per scratch row: 16 points + 4 pad (keeps 16-B alignment)

// ---- where weight fragments come from --------------------------------------------------------
// Forward-only kernels keep the packed forward blocks in LDS (ds_read with an immediate offset).
struct LdsW {
  const float* base;
  __device__ __forceinline__ f32x4 frag4(int float_off) const { return *reinterpret_cast<const f32x4*>(base + float_off + 4 * ps_lane()); }
  __device__ __forceinline__ f32x4 vec4(int float_off) const { return *reinterpret_cast<const f32x4*>(base + float_off); }
  __device__ __forceinline__ float elem(int float_off, unsigned lane_float_off) const { return base[float_off + lane_float_off]; }
  __device__ __forceinline__ LdsW at(int float_off) const { return LdsW{base + float_off}; }
};
// Backward kernels need LDS for the gradient accumulators, so they stream fragments from L2 through a
// buffer resource: address = SGPR descriptor + (lane*4 in ONE VGPR) + SGPR/immediate fragment offset.
// (Plain pointer arithmetic makes hipcc materialise a 64-bit per-lane pointer for every 4 KiB of
// fragments, hoist them out of the tile loop and spill them.)
struct GlobalW {
  __amdgpu_buffer_rsrc_t rsrc;
  int base;  // float offset inside the buffer
  __device__ __forceinline__ f32x4 frag4(int float_off) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)ps_lane() * 16u, (base + float_off) * 4, 0));
  }
  __device__ __forceinline__ f32x4 vec4(int float_off) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, 0u, (base + float_off) * 4, 0));
  }
  __device__ __forceinline__ f32x4 vec4_lane(int float_off, unsigned lane_float_off) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_float_off * 4u, (base + float_off) * 4, 0));
  }
  __device__ __forceinline__ float elem(int float_off, unsigned lane_float_off) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_float_off * 4u, (base + float_off) * 4, 0));
  }
  __device__ __forceinline__ GlobalW at(int float_off) const { return GlobalW{rsrc, base + float_off}; }
};
__device__ __forceinline__ GlobalW make_global_w(const float* p, unsigned n_floats) {
  return GlobalW{__builtin_amdgcn_make_buffer_rsrc((void*)p, 0, n_floats * 4u, 0x00020000), 0};
}

// ---- forward -------------------------------------------------------------------------------
// v[pb][t] is the B-operand array of a 16-point block: for D-chained data t = 4*nb + r.
// fragments of output block 0 (issued early by the caller so that their latency hides under earlier work)
// fragments + bias of one output block (what the MFMAs of that block consume)
template <class LT>
struct FwdFrags {
  float a[LT::KS];
  f32x4 b;  // bias of the lane's 4 neurons (D-register rows 4g..4g+3 of the block)
};
template <class LT, class W>
__device__ __forceinline__ void fetch_fwd(const W& params, int nb, FwdFrags<LT>& f) {
  if constexpr (std::is_same<W, LdsW>::value)
    f.b = params.vec4(LT::BIAS_OFF + 16 * nb + 4 * (ps_lane() >> 4));
  else
    f.b = params.vec4_lane(LT::BIAS_OFF + 16 * nb, 4u * ((unsigned)ps_lane() >> 4));
#pragma unroll
  for (int q = 0; q < LT::IB; ++q) {
    const f32x4 v = params.frag4(LT::WF_OFF + (nb * LT::IB + q) * 256);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * q + r < LT::KS) f.a[4 * q + r] = v[r];
  }
}
template <class LT, class W>
__device__ __forceinline__ void first_frags_fwd(const W& params, FwdFrags<LT>& f) {
  fetch_fwd<LT>(params, 0, f);
}
template <class LT, class W>
__device__ __forceinline__ void first_frags_bwd(const W& wt_block, float (&a)[LT::KSO]) {
#pragma unroll
  for (int q = 0; q < LT::NB; ++q) {
    const f32x4 v = wt_block.frag4(q * 256);
#pragma unroll
    for (int r = 0; r < 4; ++r) a[4 * q + r] = v[r];
  }
}
struct NoPrefetch {
  __device__ __forceinline__ void operator()() const {}
};

// `first`: fragments + bias of block 0, already requested.  `next()` is invoked before the MFMAs of the LAST block and may
// issue the first loads of whatever runs next.
template <class LT, int PB, class W, class Next>
__device__ __forceinline__ void layer_fwd_pf(const W& params, const FwdFrags<LT>& first, const float (&vin)[PB][LT::KS],
                                             float (&vout)[PB][LT::NB * 4], const Next& next) {
  // Weight fragments AND the bias are fetched one output block ahead: the loads of block nb+1 are issued (and fenced with
  // a scheduling barrier) BEFORE the MFMAs of block nb, so their L2/LDS latency hides under 16*PB matrix ops.
  // (Left to itself hipcc serialises  load -> s_waitcnt vmcnt(0) -> mfma  per k-step: 12x slower, measured; and a bias
  // fetched at the top of its own block -- it initialises the accumulators -- exposed one L2 round trip per block:
  // 1.7 ms of the main backward.)
  FwdFrags<LT> cur = first, nxt;
#pragma unroll
  for (int nb = 0; nb < LT::NB; ++nb) {
    f32x4 acc[PB];
    if (nb + 1 < LT::NB)
      fetch_fwd<LT>(params, nb + 1, nxt);
    else
      next();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PB == 1 && PS_FWD_PB1_SPLIT) {
      // a single accumulator would serialise on the 40-cycle MFMA dependency: split the k-steps over two
      // (PS_FWD_PB1_SPLIT = 0, set by a translation unit whose one-block kernels run two waves per SIMD: the other wave fills the
      //  dependency gap, and the sums keep the order of the two-block kernels -- bit-identical outputs)
      f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[0] = cur.b;
#pragma unroll
      for (int t = 0; t < LT::KS; ++t) {
        if (t & 1)
          acc2 = ps_mfma16(cur.a[t], vin[0][t], acc2);
        else
          acc[0] = ps_mfma16(cur.a[t], vin[0][t], acc[0]);
      }
      acc[0] += acc2;
    } else {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) acc[pb] = cur.b;
#pragma unroll
      for (int t = 0; t < LT::KS; ++t)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) acc[pb] = ps_mfma16(cur.a[t], vin[pb][t], acc[pb]);
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int r = 0; r < 4; ++r) vout[pb][4 * nb + r] = acc[pb][r];
    if (nb + 1 < LT::NB) cur = nxt;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- EXPLORATORY (never the product path: -DPS_SEM_BF16X3 builds only): the same layer with every fp32 operand split into two
// bf16 terms (x = hi + lo, 16 mantissa bits kept) and three v_mfma_f32_16x16x32_bf16 per 32 inputs (hi*hi + hi*lo + lo*hi, fp32
// accumulation): ~2^-16 relative error per product instead of 2^-24, 5.3x the matrix rate of the exact-fp32 MFMA.  The D -> B
// register chaining is unchanged: element i of the 8-wide bf16 operand of k-step s is the fp32 k-step 8 s + i.
typedef __bf16 ps_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void ps_split_bf16(const float (&v)[8], ps_bf16x8& hi, ps_bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hi[i] = (__bf16)v[i];
    lo[i] = (__bf16)(v[i] - (float)hi[i]);
  }
}
template <class LT, int PB, class W, class Next>
__device__ __forceinline__ void layer_fwd_pf_bf16x3(const W& params, const FwdFrags<LT>& first, const float (&vin)[PB][LT::KS],
                                                    float (&vout)[PB][LT::NB * 4], const Next& next) {
  static_assert(LT::KS % 8 == 0, "bf16x3 layer: inputs in groups of 32");
  constexpr int S = LT::KS / 8;
  ps_bf16x8 bh[PB][S], bl[PB][S];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb)
#pragma unroll
    for (int q = 0; q < S; ++q) {
      float t8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t8[i] = vin[pb][8 * q + i];
      ps_split_bf16(t8, bh[pb][q], bl[pb][q]);
    }
  FwdFrags<LT> cur = first, nxt;
#pragma unroll
  for (int nb = 0; nb < LT::NB; ++nb) {
    f32x4 acc[PB];
    if (nb + 1 < LT::NB)
      fetch_fwd<LT>(params, nb + 1, nxt);
    else
      next();
    __builtin_amdgcn_sched_barrier(0);
    ps_bf16x8 ah[S], al[S];
#pragma unroll
    for (int q = 0; q < S; ++q) {
      float t8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t8[i] = cur.a[8 * q + i];
      ps_split_bf16(t8, ah[q], al[q]);
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc[pb] = cur.b;
#pragma unroll
    for (int q = 0; q < S; ++q)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[q], bh[pb][q], acc[pb], 0, 0, 0);  // small terms first
        acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], bl[pb][q], acc[pb], 0, 0, 0);
        acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], bh[pb][q], acc[pb], 0, 0, 0);
      }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int r = 0; r < 4; ++r) vout[pb][4 * nb + r] = acc[pb][r];
    if (nb + 1 < LT::NB) cur = nxt;
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <class LT, int PB, class W>
__device__ __forceinline__ void layer_fwd(const W& params, const float (&vin)[PB][LT::KS], float (&vout)[PB][LT::NB * 4]) {
  FwdFrags<LT> a0;
  first_frags_fwd<LT>(params, a0);
  layer_fwd_pf<LT, PB>(params, a0, vin, vout, NoPrefetch());
}

template <int PB, int N>
__device__ __forceinline__ void relu_inplace(float (&v)[PB][N]) {
#pragma unroll
  for (int pb = 0; pb < PB; ++pb)
#pragma unroll
    for (int i = 0; i < N; ++i) v[pb][i] = fmaxf(v[pb][i], 0.0f);
}

// dv *= (h > 0)
template <int PB, int N>
__device__ __forceinline__ void relu_mask(float (&dv)[PB][N], const float (&h)[PB][N]) {
#pragma unroll
  for (int pb = 0; pb < PB; ++pb)
#pragma unroll
    for (int i = 0; i < N; ++i) dv[pb][i] = h[pb][i] > 0.0f ? dv[pb][i] : 0.0f;
}

// ---- backward (data) -----------------------------------------------------------------------
// dvin[pb][t], t = 4*ib + r, comes out in the layout the forward input of this layer had.
template <class LT, int PB, class W, class Next>
__device__ __forceinline__ void layer_bwd_data_pf(const W& wt_block, const float (&a_first)[LT::KSO],
                                                  const float (&dvout)[PB][LT::NB * 4], float (&dvin)[PB][LT::IB * 4],
                                                  const Next& next) {
  float a_cur[LT::KSO], a_nxt[LT::KSO];
#pragma unroll
  for (int t = 0; t < LT::KSO; ++t) a_cur[t] = a_first[t];
#pragma unroll
  for (int ib = 0; ib < LT::IB; ++ib) {
    if (ib + 1 < LT::IB) {
#pragma unroll
      for (int q = 0; q < LT::NB; ++q) {
        const f32x4 v = wt_block.frag4(((ib + 1) * LT::NB + q) * 256);
#pragma unroll
        for (int r = 0; r < 4; ++r) a_nxt[4 * q + r] = v[r];
      }
    } else {
      next();
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc[pb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (PB == 1) {
      f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < LT::KSO; ++t) {
        if (t & 1)
          acc2 = ps_mfma16(a_cur[t], dvout[0][t], acc2);
        else
          acc[0] = ps_mfma16(a_cur[t], dvout[0][t], acc[0]);
      }
      acc[0] += acc2;
    } else {
#pragma unroll
      for (int t = 0; t < LT::KSO; ++t)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) acc[pb] = ps_mfma16(a_cur[t], dvout[pb][t], acc[pb]);
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int r = 0; r < 4; ++r) dvin[pb][4 * ib + r] = acc[pb][r];
    if (ib + 1 < LT::IB) {
#pragma unroll
      for (int t = 0; t < LT::KSO; ++t) a_cur[t] = a_nxt[t];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// transposed fragments of input block ib (what the MFMAs of that block consume)
template <class LT, class W>
__device__ __forceinline__ void frags_bwd_block(const W& wt_block, int ib, float (&a)[LT::KSO]) {
#pragma unroll
  for (int q = 0; q < LT::NB; ++q) {
    const f32x4 v = wt_block.frag4((ib * LT::NB + q) * 256);
#pragma unroll
    for (int r = 0; r < 4; ++r) a[4 * q + r] = v[r];
  }
}

template <class LT, int PB, class W>
__device__ __forceinline__ void layer_bwd_data(const W& wt_block, const float (&dvout)[PB][LT::NB * 4],
                                               float (&dvin)[PB][LT::IB * 4]) {
  float a0[LT::KSO];
  first_frags_bwd<LT>(wt_block, a0);
  layer_bwd_data_pf<LT, PB>(wt_block, a0, dvout, dvin, NoPrefetch());
}

// ---- backward (weights) --------------------------------------------------------------------
// Accumulates dW += dY^T H and db += sum_p dY over the PB*16 points held by this wave.
//   scratch : per-wave LDS, LT::SCRATCH_ROWS * kScratchLd floats
//   gacc    : workgroup LDS accumulator block of this layer, LT::GPACKED floats
// select v[r] for a runtime r without dynamic register indexing
__device__ __forceinline__ float ps_sel4(const f32x4& v, int r) {
  return r == 0 ? v[0] : (r == 1 ? v[1] : (r == 2 ? v[2] : v[3]));
}

// `lock` is this layer's LDS spin-lock word (zero-initialised by the kernel).
template <class LT, int PB>
__device__ __forceinline__ void layer_bwd_weights(float* __restrict__ scratch, float* __restrict__ gacc, int* __restrict__ lock,
                                                  const float (&dvout)[PB][LT::NB * 4], const float (&vin)[PB][LT::KS]) {
  const int lane = ps_lane();
  const int j = lane & 15, g = lane >> 4;
  if constexpr (PB == 1) {
    // One point block per wave (kernels that run two waves per SIMD and must fit 256 registers): the dW tiles are computed and
    // flushed TWO output blocks at a time, so that only 2*IB tiles are live instead of NB*IB.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < LT::IB * 4; ++t)
      scratch[(16 * (t >> 2) + 4 * g + (t & 3)) * kScratchLd + j] = (t < LT::KS) ? vin[0][t < LT::KS ? t : 0] : 0.0f;
    __builtin_amdgcn_wave_barrier();
    f32x4 bfrag[LT::IB];
#pragma unroll
    for (int ib = 0; ib < LT::IB; ++ib) bfrag[ib] = *reinterpret_cast<const f32x4*>(scratch + (16 * ib + j) * kScratchLd + 4 * g);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ob0 = 0; ob0 < LT::NB; ob0 += 2) {
      constexpr int dummy = 0;
      (void)dummy;
#pragma unroll
      for (int t = 4 * ob0; t < 4 * (ob0 + 2) && t < LT::NB * 4; ++t)
        scratch[(16 * ((t >> 2) - ob0) + 4 * g + (t & 3)) * kScratchLd + j] = dvout[0][t];
      __builtin_amdgcn_wave_barrier();
      f32x4 dw2[2][LT::IB];
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int ib = 0; ib < LT::IB; ++ib) dw2[o][ib] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        if (ob0 + o < LT::NB) {
          const f32x4 afrag = *reinterpret_cast<const f32x4*>(scratch + (16 * o + j) * kScratchLd + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ib = 0; ib < LT::IB; ++ib) dw2[o][ib] = ps_mfma16(afrag[r], bfrag[ib][r], dw2[o][ib]);
        }
      }
      __builtin_amdgcn_wave_barrier();
      f32x4 db2[2];
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) db2[o][r] = (ob0 + o < LT::NB) ? ps_row16_sum(dvout[0][4 * (ob0 + o < LT::NB ? ob0 + o : 0) + r]) : 0.0f;
      if (lane == 0) {
        while (atomicCAS(lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        if (ob0 + o < LT::NB) {
#pragma unroll
          for (int ib = 0; ib < LT::IB; ++ib) {
            f32x4* dst = reinterpret_cast<f32x4*>(gacc + LT::GW_OFF + (((ob0 + o) * LT::IB + ib) * 64 + lane) * 4);
            *dst = *dst + dw2[o][ib];
          }
          if (j == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gacc[LT::GB_OFF + 16 * (ob0 + o) + 4 * g + r] += db2[o][r];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) atomicExch(lock, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  f32x4 dw[LT::NB][LT::IB];
#pragma unroll
  for (int ob = 0; ob < LT::NB; ++ob)
#pragma unroll
    for (int ib = 0; ib < LT::IB; ++ib) dw[ob][ib] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    __builtin_amdgcn_sched_barrier(0);
    // stage H: row = neuron (D-row numbering), column = point j; then read it back as B fragments (k-step r of this
    // 16-point block contracts over points {4*g + r})
#pragma unroll
    for (int t = 0; t < LT::IB * 4; ++t)
      scratch[(16 * (t >> 2) + 4 * g + (t & 3)) * kScratchLd + j] = (t < LT::KS) ? vin[pb][t < LT::KS ? t : 0] : 0.0f;
    __builtin_amdgcn_wave_barrier();  // LDS ops of one wave execute in order; only stop compiler reordering
    f32x4 bfrag[LT::IB];
#pragma unroll
    for (int ib = 0; ib < LT::IB; ++ib) {
      bfrag[ib] = *reinterpret_cast<const f32x4*>(scratch + (16 * ib + j) * kScratchLd + 4 * g);
    }
    __builtin_amdgcn_wave_barrier();
    // stage dY over the same rows, at most OBG output blocks at a time
#pragma unroll
    for (int ob0 = 0; ob0 < LT::NB; ob0 += LT::OBG) {
#pragma unroll
      for (int t = 4 * ob0; t < 4 * (ob0 + LT::OBG) && t < LT::NB * 4; ++t)
        scratch[(16 * ((t >> 2) - ob0) + 4 * g + (t & 3)) * kScratchLd + j] = dvout[pb][t];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ob = ob0; ob < ob0 + LT::OBG && ob < LT::NB; ++ob) {
        const f32x4 afrag = *reinterpret_cast<const f32x4*>(scratch + (16 * (ob - ob0) + j) * kScratchLd + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ib = 0; ib < LT::IB; ++ib) dw[ob][ib] = ps_mfma16(afrag[r], bfrag[ib][r], dw[ob][ib]);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // bias gradient partials: sum over the wave's points, reduced over the 16 lanes of a row
  f32x4 db[LT::NB];
#pragma unroll
  for (int nb = 0; nb < LT::NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = 0.f;
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) s += dvout[pb][4 * nb + r];
      db[nb][r] = ps_row16_sum(s);
    }
  // Flush into the workgroup accumulators WITHOUT LDS float atomics (ds_add_f32 retires ~1 lane per 10 cycles on
  // gfx950, measured 20x slower than integer LDS atomics or plain LDS traffic): the wave takes this layer's LDS
  // spin lock (integer compare-and-swap, fast), does plain read-modify-writes and releases.  The four waves of a
  // workgroup drift apart after the first collision, so the lock is almost always free; there is no workgroup
  // barrier on this path.
  if (lane == 0) {
    while (atomicCAS(lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // accumulator layout [tile][lane][4]: one 16-byte LDS read + write per 16x16 tile and lane
#pragma unroll
  for (int ob = 0; ob < LT::NB; ++ob)
#pragma unroll
    for (int ib = 0; ib < LT::IB; ++ib) {
      f32x4* dst = reinterpret_cast<f32x4*>(gacc + LT::GW_OFF + ((ob * LT::IB + ib) * 64 + lane) * 4);
      *dst = *dst + dw[ob][ib];
    }
  if (j == 0) {
#pragma unroll
    for (int nb = 0; nb < LT::NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) gacc[LT::GB_OFF + 16 * nb + 4 * g + r] += db[nb][r];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) atomicExch(lock, 0);
  __builtin_amdgcn_sched_barrier(0);
}

// bias-gradient partials + flush of one layer's dW tiles into the workgroup accumulators (see layer_bwd_weights)
template <class LT, int PB>
__device__ __forceinline__ void layer_bwd_flush(float* __restrict__ gacc, int* __restrict__ lock, const f32x4 (&dw)[LT::NB][LT::IB],
                                                const float (&dvout)[PB][LT::NB * 4]) {
  const int lane = ps_lane();
  const int j = lane & 15, g = lane >> 4;
  f32x4 db[LT::NB];
#pragma unroll
  for (int nb = 0; nb < LT::NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = 0.f;
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) s += dvout[pb][4 * nb + r];
      db[nb][r] = ps_row16_sum(s);
    }
  if (lane == 0) {
    while (atomicCAS(lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // read-modify-write of the tiles in batches of kBatch: written as "*dst += dw" per tile the compiler keeps every read behind
  // the previous tile's write (it cannot prove the lane-indexed addresses distinct) and each tile exposes an LDS round trip
  // to the lone wave of the SIMD (16 per 64x64 layer, 8 layers per tile of points)
#ifndef PS_FLUSH_BATCH
#define PS_FLUSH_BATCH 4  // 4: 19.9 ms/step, 8: 19.9-20.0 (2 spills), 16: 20.6
#endif
  constexpr int kTiles = LT::NB * LT::IB, kBatch = PS_FLUSH_BATCH;
#pragma unroll
  for (int t0 = 0; t0 < kTiles; t0 += kBatch) {
    f32x4 cur[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k)
      if (t0 + k < kTiles) cur[k] = *reinterpret_cast<const f32x4*>(gacc + LT::GW_OFF + ((t0 + k) * 64 + lane) * 4);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < kBatch; ++k)
      if (t0 + k < kTiles)
        *reinterpret_cast<f32x4*>(gacc + LT::GW_OFF + ((t0 + k) * 64 + lane) * 4) = cur[k] + dw[(t0 + k) / LT::IB][(t0 + k) % LT::IB];
    __builtin_amdgcn_wave_barrier();
  }
  if (j == 0) {
    float bcur[LT::NB * 4];
#pragma unroll
    for (int i = 0; i < LT::NB * 4; ++i) bcur[i] = gacc[LT::GB_OFF + 16 * (i >> 2) + 4 * g + (i & 3)];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < LT::NB * 4; ++i) gacc[LT::GB_OFF + 16 * (i >> 2) + 4 * g + (i & 3)] = bcur[i] + db[i >> 2][i & 3];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) atomicExch(lock, 0);
  __builtin_amdgcn_sched_barrier(0);
}

// Pipelined backward of ONE layer (PB >= 2): data backward FIRST, weight gradient second.
//   dW = dY^T H needs its operands transposed through the per-wave LDS scratch (write 4 B / lane, read 16 B / lane).  Issued
//   right where they are consumed, every transpose exposes an LDS round trip to the single wave of its SIMD (measured:
//   1.3 of the main backward's 6.0 ms).  Here the scratch has separate H / dY regions and
//     1. the operands of point block 0 are WRITTEN before the dX MFMAs of the layer (dX = W^T dY only needs dY), which hide
//        the write latency completely,
//     2. the operands of point block pb+1 are written right after the fragments of block pb have been read back, i.e.
//        before block pb's dW MFMAs, which hide that latency too;
//   what stays exposed is one LDS read latency per point block.  `next()` is invoked before the last block's dW MFMAs (the
//   caller requests the first transposed fragments of the next layer there).
//   The dW tiles are ADDED to `dw` (the caller zeroes them per tile of points and flushes, or keeps them over the whole kernel);
//   WANT_DB: `dbp[ob]` += this lane's share of the bias gradient of output block ob, taken from the transposed dY fragments
//   (lane (j, g): neuron 16*ob + j summed over the points 4g..4g+3 of every block; reduce over g at the end).
//   H_LATE: the layer's input `vin` is still in flight from HBM when the layer starts (the first layer of a backward kernel):
//   it is staged after the dX MFMAs instead of before them, which exposes one LDS round trip and hides the HBM latency.
template <class LT, int PB, bool WANT_DX, bool WANT_DB, bool H_LATE = false, class W, class Next>
__device__ __forceinline__ void layer_bwd_pipe_core(const W& wt_block, const float (&a_first)[LT::KSO], float* __restrict__ scratch,
                                                    f32x4 (&dw)[LT::NB][LT::IB], float (&dbp)[LT::NB],
                                                    const float (&dvout)[PB][LT::NB * 4], const float (&vin)[PB][LT::KS],
                                                    float (&dvin)[PB][LT::IB * 4], const Next& next, PsTimer* tm = nullptr, int slot = 0) {
  static_assert(PB >= 2, "the pipelined backward works on >= 2 point blocks per wave");
  const int lane = ps_lane();
  const int j = lane & 15, g = lane >> 4;
  float* sh = scratch + LT::SH_ROW0 * kScratchLd;
  float* sy = scratch + LT::SY_ROW0 * kScratchLd;
  constexpr int NG = (LT::NB + LT::OBG - 1) / LT::OBG;  // dY groups (2 only for the 80-wide base output)
  auto write_h = [&](const float (&h)[LT::KS]) {
#pragma unroll
    for (int t = 0; t < LT::IB * 4; ++t) sh[(16 * (t >> 2) + 4 * g + (t & 3)) * kScratchLd + j] = (t < LT::KS) ? h[t < LT::KS ? t : 0] : 0.0f;
  };
  auto write_dy = [&](const float (&dy)[LT::NB * 4], int ob0) {
#pragma unroll
    for (int t = 0; t < LT::OBG * 4; ++t)
      if (4 * ob0 + t < LT::NB * 4) sy[(16 * (t >> 2) + 4 * g + (t & 3)) * kScratchLd + j] = dy[(4 * ob0 + t) < LT::NB * 4 ? 4 * ob0 + t : 0];
  };
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!H_LATE) write_h(vin[0]);
  write_dy(dvout[0], 0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (WANT_DX) layer_bwd_data_pf<LT, PB>(wt_block, a_first, dvout, dvin, NoPrefetch());
  if constexpr (H_LATE) {
    __builtin_amdgcn_sched_barrier(0);
    write_h(vin[0]);
  }
  PS_STAMP(tm, slot)
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    f32x4 bfrag[LT::IB];
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
      const int ob0 = grp * LT::OBG;
      __builtin_amdgcn_sched_barrier(0);
      if (grp > 0) write_dy(dvout[pb], ob0);  // (exposed round trip; only the 80-wide layer has a second group)
      __builtin_amdgcn_wave_barrier();
      if (grp == 0) {
#pragma unroll
        for (int ib = 0; ib < LT::IB; ++ib) bfrag[ib] = *reinterpret_cast<const f32x4*>(sh + (16 * ib + j) * kScratchLd + 4 * g);
      }
      f32x4 afrag[LT::OBG];
#pragma unroll
      for (int o = 0; o < LT::OBG; ++o)
        if (ob0 + o < LT::NB) afrag[o] = *reinterpret_cast<const f32x4*>(sy + (16 * o + j) * kScratchLd + 4 * g);
      __builtin_amdgcn_wave_barrier();
      if (grp == NG - 1) {
        // LDS executes a wave's operations in order: these writes land after the reads above
        if (pb + 1 < PB) {
          write_h(vin[pb + 1 < PB ? pb + 1 : 0]);
          write_dy(dvout[pb + 1 < PB ? pb + 1 : 0], 0);
        } else {
          next();
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int o = 0; o < LT::OBG; ++o)
        if (ob0 + o < LT::NB) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ib = 0; ib < LT::IB; ++ib) dw[ob0 + o][ib] = ps_mfma16(afrag[o][r], bfrag[ib][r], dw[ob0 + o][ib]);
          if constexpr (WANT_DB) dbp[ob0 + o] += (afrag[o][0] + afrag[o][1]) + (afrag[o][2] + afrag[o][3]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  PS_STAMP(tm, slot + 1)
}

template <class LT, int PB, bool WANT_DX, class W, class Next>
__device__ __forceinline__ void layer_bwd_pipe(const W& wt_block, const float (&a_first)[LT::KSO], float* __restrict__ scratch,
                                               float* __restrict__ gacc, int* __restrict__ lock, const float (&dvout)[PB][LT::NB * 4],
                                               const float (&vin)[PB][LT::KS], float (&dvin)[PB][LT::IB * 4], const Next& next) {
  f32x4 dw[LT::NB][LT::IB];
#pragma unroll
  for (int ob = 0; ob < LT::NB; ++ob)
#pragma unroll
    for (int ib = 0; ib < LT::IB; ++ib) dw[ob][ib] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dbp[LT::NB];
  layer_bwd_pipe_core<LT, PB, WANT_DX, false>(wt_block, a_first, scratch, dw, dbp, dvout, vin, dvin, next);
  layer_bwd_flush<LT, PB>(gacc, lock, dw, dvout);
}

// Wave-resident weight gradients of one layer: a 64 x 64 layer is 64 registers per lane, so a kernel that differentiates ONE
// MLP (<= 195 accumulator registers of the 512 a lone wave per SIMD owns) keeps every dW tile in registers over all its points
// -- no flush, no LDS accumulators, no lock -- and the LDS holds the transposed weight fragments instead.
template <class LT>
struct LayerAcc {
  f32x4 dw[LT::NB][LT::IB];
  float dbp[LT::NB];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int ob = 0; ob < LT::NB; ++ob) {
      dbp[ob] = 0.0f;
#pragma unroll
      for (int ib = 0; ib < LT::IB; ++ib) dw[ob][ib] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  // add (or, first: store) this wave's accumulators into an LDS block laid out like the packed gradient block of the layer
  __device__ __forceinline__ void add_to(float* __restrict__ blk, bool first) const {
    const int lane = ps_lane(), j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ob = 0; ob < LT::NB; ++ob) {
#pragma unroll
      for (int ib = 0; ib < LT::IB; ++ib) {
        f32x4* dst = reinterpret_cast<f32x4*>(blk + LT::GW_OFF + ((ob * LT::IB + ib) * 64 + lane) * 4);
        *dst = first ? dw[ob][ib] : *dst + dw[ob][ib];
      }
      float v = dbp[ob];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) blk[LT::GB_OFF + 16 * ob + j] = first ? v : blk[LT::GB_OFF + 16 * ob + j] + v;
    }
  }
};
template <class M>
struct MlpAcc {
  LayerAcc<typename M::L0> l0;
  LayerAcc<typename M::L1> l1;  // unused (and optimised away) when NL == 2
  LayerAcc<typename M::LZ> lz;
  __device__ __forceinline__ void zero() {
    l0.zero();
    if constexpr (M::NL == 3) l1.zero();
    lz.zero();
  }
  __device__ __forceinline__ void add_to(float* __restrict__ blk, bool first) const {
    l0.add_to(blk + M::GOFF0, first);
    if constexpr (M::NL == 3) l1.add_to(blk + M::GOFF1, first);
    lz.add_to(blk + M::GOFFZ, first);
  }
};

// Register-resident variant for small MLPs (proposal nets: 2128 gradient words = 52 registers per lane): the dW / db
// tiles stay in the caller's registers over ALL tiles of the kernel, so there is no per-tile flush at all.
template <class LT, int PB>
__device__ __forceinline__ void layer_bwd_weights_acc(float* __restrict__ scratch, f32x4 (&dw)[LT::NB][LT::IB], f32x4 (&db)[LT::NB],
                                                      const float (&dvout)[PB][LT::NB * 4], const float (&vin)[PB][LT::KS]) {
  const int lane = ps_lane();
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < LT::IB * 4; ++t)
      scratch[(16 * (t >> 2) + 4 * g + (t & 3)) * kScratchLd + j] = (t < LT::KS) ? vin[pb][t < LT::KS ? t : 0] : 0.0f;
    __builtin_amdgcn_wave_barrier();
    f32x4 bfrag[LT::IB];
#pragma unroll
    for (int ib = 0; ib < LT::IB; ++ib) bfrag[ib] = *reinterpret_cast<const f32x4*>(scratch + (16 * ib + j) * kScratchLd + 4 * g);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ob0 = 0; ob0 < LT::NB; ob0 += LT::OBG) {
#pragma unroll
      for (int t = 4 * ob0; t < 4 * (ob0 + LT::OBG) && t < LT::NB * 4; ++t)
        scratch[(16 * ((t >> 2) - ob0) + 4 * g + (t & 3)) * kScratchLd + j] = dvout[pb][t];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ob = ob0; ob < ob0 + LT::OBG && ob < LT::NB; ++ob) {
        const f32x4 afrag = *reinterpret_cast<const f32x4*>(scratch + (16 * (ob - ob0) + j) * kScratchLd + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ib = 0; ib < LT::IB; ++ib) dw[ob][ib] = ps_mfma16(afrag[r], bfrag[ib][r], dw[ob][ib]);
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int nb = 0; nb < LT::NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) db[nb][r] += dvout[pb][4 * nb + r];  // per-lane partial; reduced over lanes at the end
  }
  __builtin_amdgcn_sched_barrier(0);
}

// write one wave's register accumulators of a layer as a partial gradient block (same layout as the LDS accumulators)
template <class LT>
__device__ __forceinline__ void store_layer_acc(float* __restrict__ gblock, const f32x4 (&dw)[LT::NB][LT::IB], const f32x4 (&db)[LT::NB]) {
  const int lane = ps_lane();
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ob = 0; ob < LT::NB; ++ob)
#pragma unroll
    for (int ib = 0; ib < LT::IB; ++ib)
      *reinterpret_cast<f32x4*>(gblock + LT::GW_OFF + ((ob * LT::IB + ib) * 64 + lane) * 4) = dw[ob][ib];
#pragma unroll
  for (int nb = 0; nb < LT::NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = ps_row16_sum(db[nb][r]);
      if (j == 0) gblock[LT::GB_OFF + 16 * nb + 4 * g + r] = s;
    }
}

// ---- whole-MLP description -----------------------------------------------------------------
// NL linear layers: KS0*4 inputs -> HB*16 hidden (NL-1 times) -> NBO*16 outputs.
template <int KS0_, int HB_, int NBO_, int NL_>
struct MlpT {
  static constexpr int KS0 = KS0_, HB = HB_, NBO = NBO_, NL = NL_;
  static_assert(NL_ == 2 || NL_ == 3, "2 or 3 linear layers");
  using L0 = LayerT<KS0_, HB_>;
  using L1 = LayerT<HB_ * 4, HB_>;                 // only when NL == 3
  using LZ = LayerT<HB_ * 4, NBO_>;                // last layer
  // packed parameters: [forward blocks of all layers][transposed blocks of all layers]
  static constexpr int OFF0 = 0;
  static constexpr int OFF1 = L0::FW;
  static constexpr int OFFZ = OFF1 + (NL_ == 3 ? L1::FW : 0);
  static constexpr int FW = OFFZ + LZ::FW;
  static constexpr int TOFF0 = FW;
  static constexpr int TOFF1 = TOFF0 + L0::WT;
  static constexpr int TOFFZ = TOFF1 + (NL_ == 3 ? L1::WT : 0);
  static constexpr int PACKED = TOFFZ + LZ::WT;
  static constexpr int GOFF0 = 0;
  static constexpr int GOFF1 = L0::GPACKED;
  static constexpr int GOFFZ = GOFF1 + (NL_ == 3 ? L1::GPACKED : 0);
  static constexpr int GPACKED = GOFFZ + LZ::GPACKED;
  static constexpr int SCRATCH_ROWS =
      (L0::SCRATCH_ROWS > LZ::SCRATCH_ROWS ? L0::SCRATCH_ROWS : LZ::SCRATCH_ROWS) > L1::SCRATCH_ROWS
          ? (L0::SCRATCH_ROWS > LZ::SCRATCH_ROWS ? L0::SCRATCH_ROWS : LZ::SCRATCH_ROWS)
          : L1::SCRATCH_ROWS;
};

struct NoPost {
  template <class T>
  __device__ __forceinline__ void operator()(T&) const {}
};

// forward through all layers, keeping the post-ReLU hidden activations (needed by backward).  `post_first(h1)` runs on the first
// layer's pre-activations before their ReLU (a per-point term that is added there: the per-ray part of the colour head's input).
template <class M, int PB, class W, class PostFirst = NoPost, bool BF16X3 = false>
__device__ __forceinline__ void mlp_forward(const W& params, const float (&x)[PB][M::KS0],
                                            float (&h1)[PB][M::HB * 4], float (&h2)[PB][M::HB * 4],
                                            float (&z)[PB][M::NBO * 4], const PostFirst& post_first = PostFirst()) {
  using L0 = typename M::L0;
  using L1 = typename M::L1;
  using LZ = typename M::LZ;
  const W p0 = params.at(M::OFF0), p1 = params.at(M::OFF1), pz = params.at(M::OFFZ);
  FwdFrags<L0> a0;
  FwdFrags<LZ> az;
  first_frags_fwd<L0>(p0, a0);
  if constexpr (M::NL == 3) {
    FwdFrags<L1> a1;
    layer_fwd_pf<L0, PB>(p0, a0, x, h1, [&]() { first_frags_fwd<L1>(p1, a1); });
    post_first(h1);
    relu_inplace<PB, M::HB * 4>(h1);
    layer_fwd_pf<L1, PB>(p1, a1, h1, h2, [&]() { first_frags_fwd<LZ>(pz, az); });
    relu_inplace<PB, M::HB * 4>(h2);
    layer_fwd_pf<LZ, PB>(pz, az, h2, z, NoPrefetch());
  } else if constexpr (BF16X3) {  // (exploratory builds: both layers of a two-layer stack as split-bf16 products)
    layer_fwd_pf_bf16x3<L0, PB>(p0, a0, x, h1, [&]() { first_frags_fwd<LZ>(pz, az); });
    post_first(h1);
    relu_inplace<PB, M::HB * 4>(h1);
    layer_fwd_pf_bf16x3<LZ, PB>(pz, az, h1, z, NoPrefetch());
  } else {
    layer_fwd_pf<L0, PB>(p0, a0, x, h1, [&]() { first_frags_fwd<LZ>(pz, az); });
    post_first(h1);
    relu_inplace<PB, M::HB * 4>(h1);
    layer_fwd_pf<LZ, PB>(pz, az, h1, z, NoPrefetch());
  }
}

// backward through all layers given dz (gradient w.r.t. the last layer's pre-activation output).
// WANT_DX selects whether dX (layout of x) is produced.
// `params` is the global packed block (transposed fragments are read from it through L2).
template <class M, int PB, bool WANT_DX, class W>
__device__ __forceinline__ void mlp_backward(const W& params, float* __restrict__ scratch,
                                             float* __restrict__ gacc, int* __restrict__ locks /*[3]*/,
                                             const float (&x)[PB][M::KS0],
                                             const float (&h1)[PB][M::HB * 4], const float (&h2)[PB][M::HB * 4],
                                             const float (&dz)[PB][M::NBO * 4], float (&dx)[PB][M::L0::IB * 4]) {
  using L0 = typename M::L0;
  using L1 = typename M::L1;
  using LZ = typename M::LZ;
  const W tz = params.at(M::TOFFZ), t1 = params.at(M::TOFF1), t0 = params.at(M::TOFF0);
  float dh[PB][M::HB * 4];
  if constexpr (PB >= 2) {
    // pipelined order (layer_bwd_pipe): per layer dX first, then dW; the first transposed fragments of the next layer are
    // requested before the last dW MFMAs of the current one
    // (requesting the fragments two input blocks ahead instead of one was measured: no gain, 12 more registers)
    float az[LZ::KSO];
    frags_bwd_block<LZ>(tz, 0, az);
    float a0[L0::KSO];
    auto req0 = [&]() {
      if constexpr (WANT_DX) frags_bwd_block<L0>(t0, 0, a0);
    };
    if constexpr (M::NL == 3) {
      float a1[L1::KSO];
      layer_bwd_pipe<LZ, PB, true>(tz, az, scratch, gacc + M::GOFFZ, locks + 2, dz, h2, dh, [&]() { frags_bwd_block<L1>(t1, 0, a1); });
      relu_mask<PB, M::HB * 4>(dh, h2);
      float dh1[PB][M::HB * 4];
      layer_bwd_pipe<L1, PB, true>(t1, a1, scratch, gacc + M::GOFF1, locks + 1, dh, h1, dh1, req0);
      relu_mask<PB, M::HB * 4>(dh1, h1);
      layer_bwd_pipe<L0, PB, WANT_DX>(t0, a0, scratch, gacc + M::GOFF0, locks + 0, dh1, x, dx, NoPrefetch());
    } else {
      layer_bwd_pipe<LZ, PB, true>(tz, az, scratch, gacc + M::GOFFZ, locks + 2, dz, h1, dh, req0);
      relu_mask<PB, M::HB * 4>(dh, h1);
      layer_bwd_pipe<L0, PB, WANT_DX>(t0, a0, scratch, gacc + M::GOFF0, locks + 0, dh, x, dx, NoPrefetch());
    }
    return;
  }
  // the first transposed fragments of every data-backward layer are requested before the (long, fragment-free)
  // weight-gradient phase that precedes it
  float az[LZ::KSO];
  first_frags_bwd<LZ>(tz, az);
  if constexpr (M::NL == 3) {
    layer_bwd_weights<LZ, PB>(scratch, gacc + M::GOFFZ, locks + 2, dz, h2);
    layer_bwd_data_pf<LZ, PB>(tz, az, dz, dh, NoPrefetch());
    relu_mask<PB, M::HB * 4>(dh, h2);
    float dh1[PB][M::HB * 4];
    float a1[L1::KSO];
    first_frags_bwd<L1>(t1, a1);
    layer_bwd_weights<L1, PB>(scratch, gacc + M::GOFF1, locks + 1, dh, h1);
    layer_bwd_data_pf<L1, PB>(t1, a1, dh, dh1, NoPrefetch());
    relu_mask<PB, M::HB * 4>(dh1, h1);
    float a0[L0::KSO];
    if constexpr (WANT_DX) first_frags_bwd<L0>(t0, a0);
    layer_bwd_weights<L0, PB>(scratch, gacc + M::GOFF0, locks + 0, dh1, x);
    if constexpr (WANT_DX) layer_bwd_data_pf<L0, PB>(t0, a0, dh1, dx, NoPrefetch());
  } else {
    layer_bwd_weights<LZ, PB>(scratch, gacc + M::GOFFZ, locks + 2, dz, h1);
    layer_bwd_data_pf<LZ, PB>(tz, az, dz, dh, NoPrefetch());
    relu_mask<PB, M::HB * 4>(dh, h1);
    float a0[L0::KSO];
    if constexpr (WANT_DX) first_frags_bwd<L0>(t0, a0);
    layer_bwd_weights<L0, PB>(scratch, gacc + M::GOFF0, locks + 0, dh, x);
    if constexpr (WANT_DX) layer_bwd_data_pf<L0, PB>(t0, a0, dh, dx, NoPrefetch());
  }
}

// mlp_backward with wave-resident weight gradients (MlpAcc): no LDS accumulators, no flush
// mlp_backward with wave-resident weight gradients (MlpAcc): no LDS accumulators, no flush.  The kernel's loads are still in
// flight when this starts: the last layer's input (h2 / h1) is staged late (H_LATE), and `make_x(x)` produces the first
// layer's input right before it is needed (the colour head builds it from gathered per-ray data).
// H1_LATE: the middle layer's input h1 is staged late as well (a narrow last layer gives the loads too little cover).
// `post_last(dh)` runs on the gradient w.r.t. the last hidden layer right after the last layer's data backward (before the ReLU
// mask): a second consumer of that hidden layer adds its gradient there (the factored main field: the semantic head hangs off
// the base MLP's hidden layer).
// `pre_first(d)` sees the gradient w.r.t. the FIRST layer's pre-activations (after the ReLU mask), before that layer's backward.
template <class M, int PB, bool WANT_DX, bool H1_LATE = false, class W, class MakeX, class PostLast = NoPost, class PreFirst = NoPost>
__device__ __forceinline__ void mlp_backward_acc(const W& params, float* __restrict__ scratch, MlpAcc<M>& acc, const MakeX& make_x,
                                                 const float (&h1)[PB][M::HB * 4], const float (&h2)[PB][M::HB * 4],
                                                 const float (&dz)[PB][M::NBO * 4], float (&dx)[PB][M::L0::IB * 4],
                                                 PsTimer* tm = nullptr, const PostLast& post_last = PostLast(),
                                                 const PreFirst& pre_first = PreFirst()) {
  using L0 = typename M::L0;
  using L1 = typename M::L1;
  using LZ = typename M::LZ;
  static_assert(PB >= 2, "pipelined backward only");
  const W tz = params.at(M::TOFFZ), t1 = params.at(M::TOFF1), t0 = params.at(M::TOFF0);
  float dh[PB][M::HB * 4];
  float az[LZ::KSO];
  frags_bwd_block<LZ>(tz, 0, az);
  float a0[L0::KSO];
  auto req0 = [&]() {
    if constexpr (WANT_DX) frags_bwd_block<L0>(t0, 0, a0);
  };
  float x[PB][M::KS0];
  if constexpr (M::NL == 3) {
    float a1[L1::KSO];
    layer_bwd_pipe_core<LZ, PB, true, true, true>(tz, az, scratch, acc.lz.dw, acc.lz.dbp, dz, h2, dh, [&]() { frags_bwd_block<L1>(t1, 0, a1); }, tm, 2);
    post_last(dh);
    relu_mask<PB, M::HB * 4>(dh, h2);
    float dh1[PB][M::HB * 4];
    layer_bwd_pipe_core<L1, PB, true, true, H1_LATE>(t1, a1, scratch, acc.l1.dw, acc.l1.dbp, dh, h1, dh1, req0, tm, 4);
    relu_mask<PB, M::HB * 4>(dh1, h1);
    pre_first(dh1);
    make_x(x);
    layer_bwd_pipe_core<L0, PB, WANT_DX, true>(t0, a0, scratch, acc.l0.dw, acc.l0.dbp, dh1, x, dx, NoPrefetch(), tm, 6);
  } else {
    layer_bwd_pipe_core<LZ, PB, true, true, true>(tz, az, scratch, acc.lz.dw, acc.lz.dbp, dz, h1, dh, req0, tm, 2);
    post_last(dh);
    relu_mask<PB, M::HB * 4>(dh, h1);
    pre_first(dh);
    make_x(x);
    layer_bwd_pipe_core<L0, PB, WANT_DX, true>(t0, a0, scratch, acc.l0.dw, acc.l0.dbp, dh, x, dx, NoPrefetch(), tm, 6);
  }
}

}  // namespace ps
