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
//   3. the first layer of the colour head splits over its inputs [SH16(dir) | geo15 | appearance]: the SH and appearance columns see
//      values that are constant along a ray (ns/fields/PreSight/ingp_field.py:239-262: directions and the camera's embedding are
//      broadcast to the samples), so  W_0 x_n + b_0 = W_0[:, geo] geo_n + b_0 + c_ray,  c_ray = W_0[:, sh] SH(dir) + W_0[:, app] app
//      (ps_ray_colour_fwd, 32 MACs per neuron per RAY instead of per sample; the field kernel adds c_ray before the ReLU and only
//      multiplies the 15 geometry features).  Backward: the field kernel sums d(pre-activation) over each 16-sample block,
//      ps_ray_colour_bwd reduces the blocks of a ray and forms d(W_0[:, sh | app]) and d(appearance).
// Same function, same parameters, same gradients as the reference; only fp32 sums are re-associated.
#include <algorithm>

#include "common.hpp"
#include "pointwise_core.hpp"

namespace {

// W'[o][i] = sum_k W0[o][k] * We[k][i],  b'[o] = sum_k W0[o][k] * be[k] + b0[o];   W0 [O,K], We [K,I] (row stride we_ld), b' [O]
// (both maps: every workgroup first copies the small matrices into LDS -- a thread per output walking 64 dependent global loads took
//  24 / 44 us per launch for 64^3 MACs; the sums keep their order)
__device__ __forceinline__ void merge_linear_fwd_body(const float* __restrict__ W0, const float* __restrict__ b0,
                                                      const float* __restrict__ We, const float* __restrict__ be, int O, int K, int I,
                                                      float* __restrict__ Wm, float* __restrict__ bm) {
  extern __shared__ float ml_lds[];
  float* sW0 = ml_lds;          // [O][K]
  float* sWe = sW0 + O * K;     // [K][I]
  for (int t = threadIdx.x; t < O * K; t += 256) sW0[t] = W0[t];
  for (int t = threadIdx.x; t < K * I; t += 256) sWe[t] = We[t];
  __syncthreads();
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < O * I) {
    const int o = idx / I, i = idx % I;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(sW0[o * K + k], sWe[k * I + i], s);
    Wm[idx] = s;
  } else if (idx < O * I + O) {
    const int o = idx - O * I;
    float s = b0[o];
    for (int k = 0; k < K; ++k) s = fmaf(sW0[o * K + k], be[k], s);
    bm[o] = s;
  }
}

// dW0[o][k] += sum_i dWm[o][i] We[k][i] + dbm[o] be[k];  db0[o] += dbm[o];  dWe[k][i] += sum_o W0[o][k] dWm[o][i];
// dbe[k] += sum_o W0[o][k] dbm[o]
__global__ __launch_bounds__(256) void merge_linear_fwd_kernel(const float* __restrict__ W0, const float* __restrict__ b0,
                                                               const float* __restrict__ We, const float* __restrict__ be, int O, int K,
                                                               int I, float* __restrict__ Wm, float* __restrict__ bm) {
  merge_linear_fwd_body(W0, b0, We, be, O, K, I, Wm, bm);
}

__device__ __forceinline__ void merge_linear_bwd_body(const float* __restrict__ dWm, const float* __restrict__ dbm,
                                                      const float* __restrict__ W0, const float* __restrict__ We,
                                                      const float* __restrict__ be, int O, int K, int I, float* __restrict__ dW0,
                                                      float* __restrict__ db0, float* __restrict__ dWe, float* __restrict__ dbe) {
  extern __shared__ float ml_lds[];
  float* sdWm = ml_lds;              // [O][I]
  float* sW0 = sdWm + O * I;         // [O][K]
  float* sWe = sW0 + O * K;          // [K][I + 1] (lanes walk k at a fixed i: the pad keeps them on different banks)
  for (int t = threadIdx.x; t < O * I; t += 256) sdWm[t] = dWm[t];
  for (int t = threadIdx.x; t < O * K; t += 256) sW0[t] = W0[t];
  for (int t = threadIdx.x; t < K * I; t += 256) sWe[(t / I) * (I + 1) + t % I] = We[t];
  __syncthreads();
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < O * K) {
    const int o = idx / K, k = idx % K;
    float s = dbm[o] * be[k];
    for (int i = 0; i < I; ++i) s = fmaf(sdWm[o * I + i], sWe[k * (I + 1) + i], s);
    dW0[idx] += s;
  } else if (idx < O * K + K * I) {
    const int e = idx - O * K, k = e / I, i = e % I;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(sW0[o * K + k], sdWm[o * I + i], s);
    dWe[e] += s;
  } else if (idx < O * K + K * I + O) {
    const int o = idx - O * K - K * I;
    db0[o] += dbm[o];
  } else if (idx < O * K + K * I + O + K) {
    const int k = idx - O * K - K * I - O;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(sW0[o * K + k], dbm[o], s);
    dbe[k] += s;
  }
}

__global__ __launch_bounds__(256) void merge_linear_bwd_kernel(const float* __restrict__ dWm, const float* __restrict__ dbm,
                                                               const float* __restrict__ W0, const float* __restrict__ We,
                                                               const float* __restrict__ be, int O, int K, int I,
                                                               float* __restrict__ dW0, float* __restrict__ db0,
                                                               float* __restrict__ dWe, float* __restrict__ dbe) {
  merge_linear_bwd_body(dWm, dbm, W0, We, be, O, K, I, dW0, db0, dWe, dbe);
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
    // (fully unrolled: lane k of h is a compile-time lane -> v_readlane into a scalar register instead of a trip through the LDS crossbar)
#pragma unroll
    for (int k = 0; k < kC; ++k) s = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, h), k)), Wt[k][lane], s);
    sem[r * kC + lane] = s + bias * acc[r];
  }
}

// v[r][k] = sum_c d[r][c] W[c][k];  cray[r] = sum_c d[r][c] b[c];  dW[c][k] += sum_r d[r][c] H[r][k];  db[c] += sum_r d[r][c] acc[r]
// A workgroup walks tiles of 64 rays (one barrier pair per tile: the per-ray work is a few hundred cycles, a tile per barrier
// keeps the kernel off the load -> barrier latency chain that cost the 4-rays-per-barrier version 0.3 ms).  Per tile:
//   thread t owns dW entries (c = t / 4, k in [16 (t % 4), +16)) and v entries (ray t % 64, k in [16 (t / 64), +16)).
constexpr int kTile = 64;
__global__ __launch_bounds__(256) void sem_out_bwd_kernel(const float* __restrict__ d, const float* __restrict__ H,
                                                          const float* __restrict__ acc, const float* __restrict__ W,
                                                          const float* __restrict__ b, int64_t R, int64_t rays_per_block,
                                                          float* __restrict__ v, float* __restrict__ cray, float* __restrict__ dW,
                                                          float* __restrict__ db) {
  __shared__ float Ws[kC][kC + 4];     // Ws[c][k] = W[c][k]; rows 16-byte aligned (wave-uniform ds_read_b128 broadcasts)
  __shared__ float sd[kTile][kC + 1];  // sd[ray][c]
  __shared__ float sh[kTile][kC + 4];  // sh[ray][k]
  __shared__ float sa[kTile], sb[kC];
  for (int i = threadIdx.x; i < kC * kC; i += 256) Ws[i / kC][i % kC] = W[i];
  if (threadIdx.x < kC) sb[threadIdx.x] = b[threadIdx.x];
  const int c_own = threadIdx.x >> 2, k0 = (threadIdx.x & 3) * 16;
  const int v_ray = threadIdx.x & 63, vk0 = (threadIdx.x >> 6) * 16;
  float dw[16], dbl = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) dw[i] = 0.f;
  const int64_t r_begin = (int64_t)blockIdx.x * rays_per_block, r_end = min(R, r_begin + rays_per_block);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += kTile) {
    const int n_rays = (int)min((int64_t)kTile, r_end - r0);
    __syncthreads();  // the previous tile's readers are done (first pass: W / b are in place)
#pragma unroll 4
    for (int i = threadIdx.x; i < kTile * kC; i += 256) {
      const int ray = i >> 6, c = i & 63;
      const bool ok = ray < n_rays;
      sd[ray][c] = ok ? d[(r0 + ray) * kC + c] : 0.f;
      sh[ray][c] = ok ? H[(r0 + ray) * kC + c] : 0.f;
    }
    if (threadIdx.x < kTile) sa[threadIdx.x] = threadIdx.x < n_rays ? acc[r0 + threadIdx.x] : 0.f;
    __syncthreads();
    // v and cray of ray v_ray
    {
      float vv[16], cb = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) vv[i] = 0.f;
#pragma unroll 8
      for (int c = 0; c < kC; ++c) {
        const float dc = sd[v_ray][c];
#pragma unroll
        for (int i = 0; i < 16; ++i) vv[i] = fmaf(dc, Ws[c][vk0 + i], vv[i]);
        cb = fmaf(dc, sb[c], cb);
      }
      if (v_ray < n_rays) {
#pragma unroll
        for (int i = 0; i < 16; i += 4)
          *reinterpret_cast<f32x4*>(v + (r0 + v_ray) * kC + vk0 + i) = (f32x4){vv[i], vv[i + 1], vv[i + 2], vv[i + 3]};
        if (vk0 == 0) cray[r0 + v_ray] = cb;
      }
    }
    // dW / db
#pragma unroll 4
    for (int ray = 0; ray < kTile; ++ray) {
      const float dc = sd[ray][c_own];
#pragma unroll
      for (int i = 0; i < 16; ++i) dw[i] = fmaf(dc, sh[ray][k0 + i], dw[i]);
      if ((threadIdx.x & 3) == 0) dbl = fmaf(dc, sa[ray], dbl);
    }
  }
  // one coalesced atomic per 64 consecutive entries (staged through LDS): per-thread atomics with a 16-float stride between lanes
  // made 512 workgroups x 4096 scattered atomics the whole cost of this kernel (0.28 ms)
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) sh[c_own][k0 + i] = dw[i];
  __syncthreads();
#pragma unroll
  for (int i = threadIdx.x; i < kC * kC; i += 256) unsafeAtomicAdd(dW + i, sh[i >> 6][i & 63]);
  if ((threadIdx.x & 3) == 0) unsafeAtomicAdd(db + c_own, dbl);
}

// c[r][n] = sum_{c<16} W0[n][c] SH_c(dirs[r]) + sum_{a<A} W0[n][31 + a] app[r][a];   W0 [HC, 31 + A] (torch layout)
__global__ __launch_bounds__(256) void ray_colour_fwd_kernel(const float* __restrict__ dirs, const float* __restrict__ app,
                                                             const float* __restrict__ W0, int64_t R, int A, int HC,
                                                             float* __restrict__ cray) {
  __shared__ float Wt[32][129];  // Wt[c][n]: c < 16 SH, 16 + a appearance
  const int ld = 31 + A;
  for (int i = threadIdx.x; i < 32 * HC; i += 256) {
    const int n = i / 32, c = i % 32;
    Wt[c][n] = c < 16 ? W0[n * ld + c] : (c - 16 < A ? W0[n * ld + 31 + (c - 16)] : 0.f);
  }
  __syncthreads();
  const int rays_per_pass = 256 / HC, n = threadIdx.x % HC, sub = threadIdx.x / HC;
  for (int64_t r = (int64_t)blockIdx.x * rays_per_pass + sub; r < R; r += (int64_t)gridDim.x * rays_per_pass) {
    float sh[16];
    ps::sh4((dirs[r * 3] + 1.0f) / 2.0f, (dirs[r * 3 + 1] + 1.0f) / 2.0f, (dirs[r * 3 + 2] + 1.0f) / 2.0f, sh);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s = fmaf(sh[c], Wt[c][n], s);
    for (int a = 0; a < A; ++a) s = fmaf(app[r * A + a], Wt[16 + a][n], s);
    cray[r * HC + n] = s;
  }
}

// d[r][n] = sum over the S/16 blocks of ray r of dpart[block][n];  dW0[n][sh | app columns] += sum_r d[r][n] x[r][c];
// dapp[r][a] = sum_n d[r][n] W0[n][31 + a].  A workgroup walks tiles of 64 rays of a contiguous range; thread t owns the dW0
// entries (n = t / 4, 8 of the 32 input columns) and dapp entries (ray t % 64, a in [4 (t / 64), +4)).
__global__ __launch_bounds__(256) void ray_colour_bwd_kernel(const float* __restrict__ dpart, const float* __restrict__ dirs,
                                                             const float* __restrict__ app, const float* __restrict__ W0, int64_t R,
                                                             int blocks_per_ray, int A, int HC, int64_t rays_per_block,
                                                             float* __restrict__ dW0, float* __restrict__ dapp) {
  __shared__ float Wa[kC][16 + 4];     // Wa[n][a] = W0[n][31 + a]
  __shared__ float sx[kTile][32 + 4];  // x[ray][c]: SH16 | appearance (zero padded)
  __shared__ float sd[kTile][kC + 1];  // d[ray][n]
  const int ld = 31 + A;
  for (int i = threadIdx.x; i < kC * 16; i += 256) {
    const int n = i / 16, a = i % 16;
    Wa[n][a] = (n < HC && a < A) ? W0[n * ld + 31 + a] : 0.f;
  }
  const int n_own = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 8;
  const int a_ray = threadIdx.x & 63, a0 = (threadIdx.x >> 6) * 4;
  float dw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) dw[i] = 0.f;
  const int64_t r_begin = (int64_t)blockIdx.x * rays_per_block, r_end = min(R, r_begin + rays_per_block);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += kTile) {
    const int n_rays = (int)min((int64_t)kTile, r_end - r0);
    __syncthreads();
    {  // thread t reduces the blocks of (ray 4 e + t / 64, neuron t % 64), e < 16: 16 independent loads per block index in flight
      const int n = threadIdx.x & 63, ray0 = threadIdx.x >> 6;
      float dv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) dv[e] = 0.f;
      for (int blk = 0; blk < blocks_per_ray; ++blk) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ray = 4 * e + ray0;
          if (ray < n_rays && n < HC) dv[e] += dpart[((r0 + ray) * blocks_per_ray + blk) * HC + n];
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) sd[4 * e + ray0][n] = dv[e];
    }
    if (threadIdx.x < kTile) {
      const int64_t r = r0 + threadIdx.x;
      float sh[16];
      if (threadIdx.x < n_rays) {
        ps::sh4((dirs[r * 3] + 1.0f) / 2.0f, (dirs[r * 3 + 1] + 1.0f) / 2.0f, (dirs[r * 3 + 2] + 1.0f) / 2.0f, sh);
      } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) sh[c] = 0.f;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) sx[threadIdx.x][c] = sh[c];
    } else if (threadIdx.x < 2 * kTile) {
      const int ray = threadIdx.x - kTile;
      for (int a = 0; a < 16; ++a) sx[ray][16 + a] = (ray < n_rays && a < A) ? app[(r0 + ray) * A + a] : 0.f;
    }
    __syncthreads();
    if (dapp != nullptr && a0 < A) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int n = 0; n < kC; ++n) {
        const float dn = sd[a_ray][n];
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = fmaf(dn, Wa[n][a0 + i], s[i]);
      }
      if (a_ray < n_rays) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (a0 + i < A) dapp[(r0 + a_ray) * A + a0 + i] = s[i];
      }
    }
#pragma unroll 4
    for (int ray = 0; ray < kTile; ++ray) {
      const float dn = sd[ray][n_own];
#pragma unroll
      for (int i = 0; i < 8; ++i) dw[i] = fmaf(dn, sx[ray][c0 + i], dw[i]);
    }
  }
  // lanes -> consecutive columns of one row of dW0 (staged through LDS; see sem_out_bwd_kernel)
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) sx[n_own][c0 + i] = dw[i];
  __syncthreads();
  for (int i = threadIdx.x; i < HC * 32; i += 256) {
    const int n = i >> 5, c = i & 31;
    if (c < 16)
      unsafeAtomicAdd(dW0 + n * ld + c, sx[n][c]);
    else if (c - 16 < A)
      unsafeAtomicAdd(dW0 + n * ld + 31 + (c - 16), sx[n][c]);
  }
}

}  // namespace

// per-ray term of the colour head's first layer (see 3. above): dirs [R,3], app [R,A] or null (A = 0), W0 [HC, 31 + A] in the torch
// layout, cray [R, HC]
extern "C" int ps_ray_colour_fwd(const float* dirs, const float* app, const float* W0, int64_t R, int A, int HC, float* cray, void* stream) {
  PS_REQUIRE(dirs && W0 && cray, "ps_ray_colour_fwd: null argument");
  PS_REQUIRE(A >= 0 && A <= 16 && (A == 0 || app != nullptr) && (HC == 32 || HC == 64), "ps_ray_colour_fwd: appearance dim <= 16, 32 or 64 neurons");
  if (R == 0) return 0;
  const int64_t per = 256 / HC;
  int grid = (int)std::min<int64_t>((R + per - 1) / per, 2048);
  ray_colour_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dirs, app, W0, R, A, HC, cray);
  PS_CHECK_LAUNCH();
}

// dpart [R * S / 16, HC]: per-16-sample-block sums of d(first-layer pre-activation) from ps_main_field_f_bwd (S % 16 == 0);
// dW0 [HC, 31 + A] receives (+=) its SH and appearance columns, dapp [R, A] (or null) is written
extern "C" int ps_ray_colour_bwd(const float* dpart, const float* dirs, const float* app, const float* W0, int64_t R, int S, int A, int HC,
                                 float* dW0, float* dapp, void* stream) {
  PS_REQUIRE(dpart && dirs && W0 && dW0, "ps_ray_colour_bwd: null argument");
  PS_REQUIRE(A >= 0 && A <= 16 && (A == 0 || app != nullptr) && (HC == 32 || HC == 64) && S > 0 && S % 16 == 0,
             "ps_ray_colour_bwd: appearance dim <= 16, 32 or 64 neurons, samples per ray a multiple of 16");
  if (R == 0) return 0;
  const int64_t per = 128;  // rays per workgroup (two tiles; 64: 99 us, 128: 82 us, 256: 124 us at 65536 rays x 8 blocks)
  ray_colour_bwd_kernel<<<(unsigned)((R + per - 1) / per), 256, 0, (hipStream_t)stream>>>(dpart, dirs, app, W0, R, S / 16, A, HC, per, dW0,
                                                                                         A > 0 ? dapp : nullptr);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_merge_linear_fwd(const float* W0, const float* b0, const float* We, const float* be, int O, int K, int I, float* Wm,
                                   float* bm, void* stream) {
  PS_REQUIRE(W0 && b0 && We && be && Wm && bm && O > 0 && K > 0 && I > 0, "ps_merge_linear_fwd: null argument");
  PS_REQUIRE((size_t)(O * K + K * I) * 4 <= 60 * 1024, "ps_merge_linear_fwd: layers of at most ~85 x 85");
  merge_linear_fwd_kernel<<<(unsigned)((O * I + O + 255) / 256), 256, (size_t)(O * K + K * I) * 4, (hipStream_t)stream>>>(W0, b0, We, be, O, K,
                                                                                                                    I, Wm, bm);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_merge_linear_bwd(const float* dWm, const float* dbm, const float* W0, const float* We, const float* be, int O, int K,
                                   int I, float* dW0, float* db0, float* dWe, float* dbe, void* stream) {
  PS_REQUIRE(dWm && dbm && W0 && We && be && dW0 && db0 && dWe && dbe, "ps_merge_linear_bwd: null argument");
  const size_t lds = (size_t)(O * I + O * K + K * (I + 1)) * 4;
  PS_REQUIRE(lds <= 60 * 1024, "ps_merge_linear_bwd: layers of at most ~70 x 70");
  merge_linear_bwd_kernel<<<(unsigned)((O * K + K * I + O + K + 255) / 256), 256, lds, (hipStream_t)stream>>>(dWm, dbm, W0, We, be, O, K, I,
                                                                                                           dW0, db0, dWe, dbe);
  PS_CHECK_LAUNCH();
}

// The same two maps for K sub-fields in ONE launch each (routed tiles: K merged layers per step).  ptrs: device table of K rows of
// addresses (int64): forward [W0, b0, We, be], backward [W0, We, be, dW0, db0, dWe, dbe]; Wm / bm / dWm / dbm: [K, O, I] / [K, O]
// contiguous.  blockIdx.y = sub-field.
namespace {
__global__ __launch_bounds__(256) void merge_linear_fwd_batch_kernel(const int64_t* __restrict__ ptrs, int O, int K, int I,
                                                                     float* __restrict__ Wm, float* __restrict__ bm) {
  const int64_t* row = ptrs + 4 * blockIdx.y;
  merge_linear_fwd_body((const float*)row[0], (const float*)row[1], (const float*)row[2], (const float*)row[3], O, K, I,
                        Wm + (size_t)blockIdx.y * O * I, bm + (size_t)blockIdx.y * O);
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void merge_linear_bwd_batch_kernel(const int64_t* __restrict__ ptrs, const float* __restrict__ dWm,
                                                                     const float* __restrict__ dbm, int O, int K, int I) {
  const int64_t* row = ptrs + 7 * blockIdx.y;
  merge_linear_bwd_body(dWm + (size_t)blockIdx.y * O * I, dbm + (size_t)blockIdx.y * O, (const float*)row[0], (const float*)row[1],
                        (const float*)row[2], O, K, I, (float*)row[3], (float*)row[4], (float*)row[5], (float*)row[6]);
}
}  // namespace

extern "C" int ps_merge_linear_bwd_batch(const int64_t* ptrs, int n_fields, const float* dWm, const float* dbm, int O, int K, int I,
                                         void* stream) {
  PS_REQUIRE(ptrs && dWm && dbm && n_fields > 0 && O > 0 && K > 0 && I > 0, "ps_merge_linear_bwd_batch: null argument");
  const size_t lds = (size_t)(O * I + O * K + K * (I + 1)) * 4;
  PS_REQUIRE(lds <= 60 * 1024, "ps_merge_linear_bwd_batch: layers of at most ~70 x 70");
  merge_linear_bwd_batch_kernel<<<dim3((unsigned)((O * K + K * I + O + K + 255) / 256), (unsigned)n_fields), 256, lds, (hipStream_t)stream>>>(
      ptrs, dWm, dbm, O, K, I);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_merge_linear_fwd_batch(const int64_t* ptrs, int n_fields, int O, int K, int I, float* Wm, float* bm, void* stream) {
  PS_REQUIRE(ptrs && Wm && bm && n_fields > 0 && O > 0 && K > 0 && I > 0, "ps_merge_linear_fwd_batch: null argument");
  PS_REQUIRE((size_t)(O * K + K * I) * 4 <= 60 * 1024, "ps_merge_linear_fwd_batch: layers of at most ~85 x 85");
  merge_linear_fwd_batch_kernel<<<dim3((unsigned)((O * I + O + 255) / 256), (unsigned)n_fields), 256, (size_t)(O * K + K * I) * 4,
                                  (hipStream_t)stream>>>(ptrs, O, K, I, Wm, bm);
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
  const int64_t per = 128;  // rays per workgroup (two tiles; 64: 74 us, 128: 62 us, 256: 92 us at 65536 rays)
  sem_out_bwd_kernel<<<(unsigned)((R + per - 1) / per), 256, 0, (hipStream_t)stream>>>(dsem, H, acc, W, b, R, per, v, cray, dW, db);
  PS_CHECK_LAUNCH();
}
