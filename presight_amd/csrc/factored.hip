// Weight-space and per-ray pieces of the FACTORED semantic path of the training render node (field.hip MainCfg FACT):
//
//   semantics(ray) = sum_n w_n * Head(E_n),   Head = Linear_out o ReLU o Linear_1 o ReLU o Linear_0,   E_n = W_e h_n + b_e
//   (E = rows 16..79 of the base MLP's output layer, no activation in between: ns/fields/PreSight/ingp_field.py:130-151, 193-237;
//    compositing: ns/models/PreSight/nerfacto_nusc_ms.py:530)
//
//   1. Linear_0 o (W_e, b_e) is ONE linear map of the hidden activations h_n:  W' = W_0 W_e,  b' = W_0 b_e + b_0
//      (ps_merge_linear_fwd, 64 x 64 x 64 MACs once per step; ps_merge_linear_bwd carries d(W'), d(b') back to the four tensors);
//   2. Linear_out commutes with the compositing sum:  semantics(ray) = W_out (sum_n w_n s_n) + b_out sum_n w_n  with s_n the last
//      hidden activations (ps_sem_out_fwd, one 64 x 64 layer per RAY instead of per sample); backward: v = W_out^T d(sem) is the
//      per-ray gradient the field kernels scale by w_n, d(accumulation) += <d(sem), b_out>, dW_out = d(sem)^T (sum_n w_n s_n).
// Same function, same parameters, same gradients as the reference; only fp32 sums are re-associated.
#include "common.hpp"

namespace {

// W'[o][i] = sum_k W0[o][k] * We[k][i],  b'[o] = sum_k W0[o][k] * be[k] + b0[o];   W0 [O,K], We [K,I] (row stride we_ld), b' [O]
__global__ __launch_bounds__(256) void merge_linear_fwd_kernel(const float* __restrict__ W0, const float* __restrict__ b0,
                                                               const float* __restrict__ We, const float* __restrict__ be, int O, int K,
                                                               int I, float* __restrict__ Wm, float* __restrict__ bm) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < O * I) {
    const int o = idx / I, i = idx % I;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(W0[o * K + k], We[k * I + i], s);
    Wm[idx] = s;
  } else if (idx < O * I + O) {
    const int o = idx - O * I;
    float s = b0[o];
    for (int k = 0; k < K; ++k) s = fmaf(W0[o * K + k], be[k], s);
    bm[o] = s;
  }
}

// dW0[o][k] += sum_i dWm[o][i] We[k][i] + dbm[o] be[k];  db0[o] += dbm[o];  dWe[k][i] += sum_o W0[o][k] dWm[o][i];
// dbe[k] += sum_o W0[o][k] dbm[o]
__global__ __launch_bounds__(256) void merge_linear_bwd_kernel(const float* __restrict__ dWm, const float* __restrict__ dbm,
                                                               const float* __restrict__ W0, const float* __restrict__ We,
                                                               const float* __restrict__ be, int O, int K, int I,
                                                               float* __restrict__ dW0, float* __restrict__ db0,
                                                               float* __restrict__ dWe, float* __restrict__ dbe) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < O * K) {
    const int o = idx / K, k = idx % K;
    float s = dbm[o] * be[k];
    for (int i = 0; i < I; ++i) s = fmaf(dWm[o * I + i], We[k * I + i], s);
    dW0[idx] += s;
  } else if (idx < O * K + K * I) {
    const int e = idx - O * K, k = e / I, i = e % I;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(W0[o * K + k], dWm[o * I + i], s);
    dWe[e] += s;
  } else if (idx < O * K + K * I + O) {
    const int o = idx - O * K - K * I;
    db0[o] += dbm[o];
  } else if (idx < O * K + K * I + O + K) {
    const int k = idx - O * K - K * I - O;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(W0[o * K + k], dbm[o], s);
    dbe[k] += s;
  }
}

// sem[r][c] = sum_k H[r][k] W[c][k] + b[c] * acc[r]     (C = 64 outputs = one lane each, one wavefront per ray)
constexpr int kC = 64;
__global__ __launch_bounds__(256) void sem_out_fwd_kernel(const float* __restrict__ H, const float* __restrict__ acc,
                                                          const float* __restrict__ W, const float* __restrict__ b, int64_t R,
                                                          float* __restrict__ sem) {
  __shared__ float Wt[kC][kC + 1];  // Wt[k][c] = W[c][k]
  for (int i = threadIdx.x; i < kC * kC; i += 256) Wt[i % kC][i / kC] = W[i];
  __syncthreads();
  const int lane = ps_lane(), wave = threadIdx.x >> 6;
  const float bias = b[lane];
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < R; r += (int64_t)gridDim.x * 4) {
    const float h = H[r * kC + lane];
    float s = 0.f;
#pragma unroll 16
    for (int k = 0; k < kC; ++k) s = fmaf(__shfl(h, k, 64), Wt[k][lane], s);
    sem[r * kC + lane] = s + bias * acc[r];
  }
}

// v[r][k] = sum_c d[r][c] W[c][k];  cray[r] = sum_c d[r][c] b[c];  dW[c][k] += sum_r d[r][c] H[r][k];  db[c] += sum_r d[r][c] acc[r]
// A workgroup takes a contiguous range of rays; thread t owns dW entries (c = t / 4, k in [16 (t % 4), +16)) over that range.
__global__ __launch_bounds__(256) void sem_out_bwd_kernel(const float* __restrict__ d, const float* __restrict__ H,
                                                          const float* __restrict__ acc, const float* __restrict__ W,
                                                          const float* __restrict__ b, int64_t R, int64_t rays_per_block,
                                                          float* __restrict__ v, float* __restrict__ cray, float* __restrict__ dW,
                                                          float* __restrict__ db) {
  __shared__ float Ws[kC][kC + 1];  // Ws[c][k] = W[c][k]
  __shared__ float sd[4][kC], sh[4][kC], sa[4];
  for (int i = threadIdx.x; i < kC * kC; i += 256) Ws[i / kC][i % kC] = W[i];
  __syncthreads();
  const int lane = ps_lane(), wave = threadIdx.x >> 6;
  const int c_own = threadIdx.x >> 2, k0 = (threadIdx.x & 3) * 16;
  float dw[16], dbl = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) dw[i] = 0.f;
  const float bias = b[lane];
  const int64_t r_begin = (int64_t)blockIdx.x * rays_per_block, r_end = min(R, r_begin + rays_per_block);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 4) {
    const int64_t r = r0 + wave;
    const bool ok = r < r_end;
    const float dv = ok ? d[r * kC + lane] : 0.f;
    // v[r][lane] = sum_c d[r][c] W[c][lane]
    float s = 0.f;
#pragma unroll 16
    for (int c = 0; c < kC; ++c) s = fmaf(__shfl(dv, c, 64), Ws[c][lane], s);
    float cb = dv * bias;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) cb += __shfl_xor(cb, sft, 64);
    if (ok) {
      v[r * kC + lane] = s;
      if (lane == 0) cray[r] = cb;
    }
    sd[wave][lane] = dv;
    sh[wave][lane] = ok ? H[r * kC + lane] : 0.f;
    if (lane == 0) sa[wave] = ok ? acc[r] : 0.f;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float dc = sd[w][c_own];
#pragma unroll
      for (int i = 0; i < 16; ++i) dw[i] = fmaf(dc, sh[w][k0 + i], dw[i]);
      if ((threadIdx.x & 3) == 0) dbl = fmaf(dc, sa[w], dbl);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) unsafeAtomicAdd(dW + c_own * kC + k0 + i, dw[i]);
  if ((threadIdx.x & 3) == 0) unsafeAtomicAdd(db + c_own, dbl);
}

}  // namespace

extern "C" int ps_merge_linear_fwd(const float* W0, const float* b0, const float* We, const float* be, int O, int K, int I, float* Wm,
                                   float* bm, void* stream) {
  PS_REQUIRE(W0 && b0 && We && be && Wm && bm && O > 0 && K > 0 && I > 0, "ps_merge_linear_fwd: null argument");
  merge_linear_fwd_kernel<<<(unsigned)((O * I + O + 255) / 256), 256, 0, (hipStream_t)stream>>>(W0, b0, We, be, O, K, I, Wm, bm);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_merge_linear_bwd(const float* dWm, const float* dbm, const float* W0, const float* We, const float* be, int O, int K,
                                   int I, float* dW0, float* db0, float* dWe, float* dbe, void* stream) {
  PS_REQUIRE(dWm && dbm && W0 && We && be && dW0 && db0 && dWe && dbe, "ps_merge_linear_bwd: null argument");
  merge_linear_bwd_kernel<<<(unsigned)((O * K + K * I + O + K + 255) / 256), 256, 0, (hipStream_t)stream>>>(dWm, dbm, W0, We, be, O, K, I, dW0,
                                                                                                          db0, dWe, dbe);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_sem_out_fwd(const float* H, const float* acc, const float* W, const float* b, int64_t R, int C, float* sem, void* stream) {
  PS_REQUIRE(C == kC, "ps_sem_out_fwd: 64 semantic channels");
  if (R == 0) return 0;
  int grid = (int)((R + 3) / 4);
  if (grid > 2048) grid = 2048;
  sem_out_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(H, acc, W, b, R, sem);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_sem_out_bwd(const float* dsem, const float* H, const float* acc, const float* W, const float* b, int64_t R, int C,
                              float* v, float* cray, float* dW, float* db, void* stream) {
  PS_REQUIRE(C == kC, "ps_sem_out_bwd: 64 semantic channels");
  if (R == 0) return 0;
  const int64_t per = 256;  // rays per workgroup
  sem_out_bwd_kernel<<<(unsigned)((R + per - 1) / per), 256, 0, (hipStream_t)stream>>>(dsem, H, acc, W, b, R, per, v, cray, dW, db);
  PS_CHECK_LAUNCH();
}
