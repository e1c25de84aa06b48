// A chain of 64 -> 64 ReLU layers on register-resident activations (D registers of one layer = B operand of the next, weights
// fetched from LDS one output block ahead -- the structure of mlp_core.hpp's layer_fwd_pf), on two MFMA tile shapes:
//   v_mfma_f32_16x16x4_f32   16-point blocks, 1 (PB1) or 2 (PB2) blocks per wave          -- what the product kernels use
//   v_mfma_f32_32x32x2_f32   one 32-point block per wave; D register j of output block ob holds neuron 32 ob + 8 (j / 4) + 4 (lane / 32)
//                            + j % 4 of point lane % 32, and is the B operand of k-step 16 ob + j as it stands (the k order is free:
//                            the A fragments are packed to match) -- half the matrix instructions, 64 issue cycles each
// Same FLOP rate on paper (256 FLOP / cycle / CU).  The question: does the longer instruction leave the chain
// closer to the matrix peak?      hipcc --offload-arch=gfx950 -O3 tools/microbench/mlp_chain_tiles.hip -o /tmp/mlp_chain && /tmp/mlp_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NL = 4;  // distinct layers in LDS (64 KiB of weights + biases), walked REPS times per tile
constexpr int W_PER_LAYER = 64 * 64, LAYER_STRIDE = W_PER_LAYER + 64;

__host__ __device__ inline float input_value(long point, int feature) {
  unsigned h = (unsigned)(point * 2654435761u) ^ (unsigned)(feature * 40503u + 977u);
  h ^= h >> 13;
  h *= 0x5bd1e995u;
  h ^= h >> 15;
  return (float)(h & 0xffffu) * (1.0f / 65536.0f);
}

// feature held by D register j (= k-step index inside its block) for lane half h
__host__ __device__ inline int feat32(int s, int h) { return 32 * (s / 16) + 8 * ((s % 16) / 4) + 4 * h + (s % 16) % 4; }
__host__ __device__ inline int feat16(int t, int g) { return 16 * (t / 4) + 4 * g + t % 4; }

// ---------------------------------------------------------------- 16x16x4
template <int PB>
__global__ __launch_bounds__(512) void chain16(const float* __restrict__ packed, float* __restrict__ out, int tiles_per_wave, int reps,
                                               int store) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < NL * LAYER_STRIDE; i += blockDim.x) lds[i] = packed[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave_in_grid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  float total = 0.f;
  for (int tile = 0; tile < tiles_per_wave; ++tile) {
    const long p0 = ((long)wave_in_grid * tiles_per_wave + tile) * (16 * PB);
    float v[PB][16];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int t = 0; t < 16; ++t) v[pb][t] = input_value(p0 + pb * 16 + (lane & 15), feat16(t, lane >> 4));
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const float* w = lds + l * LAYER_STRIDE;
        float vo[PB][16];
        f32x4 a_cur[4], a_nxt[4], b_cur, b_nxt;
#pragma unroll
        for (int q = 0; q < 4; ++q) a_cur[q] = *reinterpret_cast<const f32x4*>(w + q * 256 + 4 * lane);
        b_cur = *reinterpret_cast<const f32x4*>(w + W_PER_LAYER + 4 * (lane >> 4));
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          if (nb + 1 < 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a_nxt[q] = *reinterpret_cast<const f32x4*>(w + ((nb + 1) * 4 + q) * 256 + 4 * lane);
            b_nxt = *reinterpret_cast<const f32x4*>(w + W_PER_LAYER + 16 * (nb + 1) + 4 * (lane >> 4));
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x4 acc[PB];
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) acc[pb] = b_cur;
#pragma unroll
          for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) acc[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t >> 2][t & 3], v[pb][t], acc[pb], 0, 0, 0);
#pragma unroll
          for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int r = 0; r < 4; ++r) vo[pb][4 * nb + r] = fmaxf(acc[pb][r], 0.f);
          if (nb + 1 < 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a_cur[q] = a_nxt[q];
            b_cur = b_nxt;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int t = 0; t < 16; ++t) v[pb][t] = vo[pb][t];
      }
    }
    if (store) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int t = 0; t < 16; ++t) out[(p0 + pb * 16 + (lane & 15)) * 64 + feat16(t, lane >> 4)] = v[pb][t];
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int t = 0; t < 16; ++t) total += v[pb][t];
  }
  if (total == 12345.678f) out[0] = total;
}

// ---------------------------------------------------------------- 32x32x2
__global__ __launch_bounds__(512) void chain32(const float* __restrict__ packed, float* __restrict__ out, int tiles_per_wave, int reps,
                                               int store) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < NL * LAYER_STRIDE; i += blockDim.x) lds[i] = packed[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave_in_grid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  float total = 0.f;
  for (int tile = 0; tile < tiles_per_wave; ++tile) {
    const long p0 = ((long)wave_in_grid * tiles_per_wave + tile) * 32;
    float v[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) v[s] = input_value(p0 + (lane & 31), feat32(s, lane >> 5));
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const float* w = lds + l * LAYER_STRIDE;
        float vo[32];
        f32x4 a_cur[8], a_nxt[8], b_cur[4], b_nxt[4];
#pragma unroll
        for (int q = 0; q < 8; ++q) a_cur[q] = *reinterpret_cast<const f32x4*>(w + q * 256 + 4 * lane);
#pragma unroll
        for (int g = 0; g < 4; ++g) b_cur[g] = *reinterpret_cast<const f32x4*>(w + W_PER_LAYER + 8 * g + 4 * (lane >> 5));
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
          if (ob == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a_nxt[q] = *reinterpret_cast<const f32x4*>(w + (8 + q) * 256 + 4 * lane);
#pragma unroll
            for (int g = 0; g < 4; ++g) b_nxt[g] = *reinterpret_cast<const f32x4*>(w + W_PER_LAYER + 32 + 8 * g + 4 * (lane >> 5));
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x16 acc;
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[j] = b_cur[j >> 2][j & 3];
#pragma unroll
          for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s >> 2][s & 3], v[s], acc, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 16; ++j) vo[16 * ob + j] = fmaxf(acc[j], 0.f);
          if (ob == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a_cur[q] = a_nxt[q];
#pragma unroll
            for (int g = 0; g < 4; ++g) b_cur[g] = b_nxt[g];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) v[s] = vo[s];
      }
    }
    if (store) {
#pragma unroll
      for (int s = 0; s < 32; ++s) out[(p0 + (lane & 31)) * 64 + feat32(s, lane >> 5)] = v[s];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) total += v[s];
  }
  if (total == 12345.678f) out[0] = total;
}

// ---------------------------------------------------------------- host
static void pack16(const std::vector<float>& W, const std::vector<float>& b, std::vector<float>& p) {
  p.assign(NL * LAYER_STRIDE, 0.f);
  for (int l = 0; l < NL; ++l) {
    for (int nb = 0; nb < 4; ++nb)
      for (int q = 0; q < 4; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r)
            p[l * LAYER_STRIDE + (nb * 4 + q) * 256 + 4 * lane + r] = W[(l * 64 + 16 * nb + lane % 16) * 64 + feat16(4 * q + r, lane / 16)];
    for (int n = 0; n < 64; ++n) p[l * LAYER_STRIDE + W_PER_LAYER + n] = b[l * 64 + n];
  }
}
static void pack32(const std::vector<float>& W, const std::vector<float>& b, std::vector<float>& p) {
  p.assign(NL * LAYER_STRIDE, 0.f);
  for (int l = 0; l < NL; ++l) {
    for (int ob = 0; ob < 2; ++ob)
      for (int q = 0; q < 8; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r)
            p[l * LAYER_STRIDE + (ob * 8 + q) * 256 + 4 * lane + r] = W[(l * 64 + 32 * ob + lane % 32) * 64 + feat32(4 * q + r, lane / 32)];
    for (int n = 0; n < 64; ++n) p[l * LAYER_STRIDE + W_PER_LAYER + n] = b[l * 64 + n];
  }
}

template <class K>
static double time_kernel(K launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  std::vector<float> W(NL * 64 * 64), b(NL * 64);
  unsigned s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
  };
  for (auto& x : W) x = rnd() * 0.35f;
  for (auto& x : b) x = rnd() * 0.1f;
  std::vector<float> p16, p32;
  pack16(W, b, p16);
  pack32(W, b, p32);
  float *d16, *d32, *dout;
  const int lds_bytes = 96 * 1024;  // (66 KB used: the allocation pins ONE workgroup per CU, so threads / 256 = waves per SIMD)
  const int packed_bytes = NL * LAYER_STRIDE * 4;
  hipMalloc(&d16, packed_bytes);
  hipMalloc(&d32, packed_bytes);
  hipMemcpy(d16, p16.data(), packed_bytes, hipMemcpyHostToDevice);
  hipMemcpy(d32, p32.data(), packed_bytes, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)chain16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipFuncSetAttribute((const void*)chain16<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipFuncSetAttribute((const void*)chain32, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  // ---- correctness: 4 workgroups x 4 waves x 2 tiles, one pass over the NL layers, against a double-precision host chain
  const int vb = 4, vt = 256, vtiles = 2;
  const long vpoints = (long)vb * (vt / 64) * vtiles * 32;
  hipMalloc(&dout, vpoints * 64 * 4);
  std::vector<float> ref(vpoints * 64), got(vpoints * 64);
  for (long p = 0; p < vpoints; ++p) {
    double x[64], y[64];
    for (int f = 0; f < 64; ++f) x[f] = input_value(p, f);
    for (int l = 0; l < NL; ++l) {
      for (int n = 0; n < 64; ++n) {
        double a = b[l * 64 + n];
        for (int k = 0; k < 64; ++k) a += (double)W[(l * 64 + n) * 64 + k] * x[k];
        y[n] = a > 0 ? a : 0;
      }
      for (int n = 0; n < 64; ++n) x[n] = y[n];
    }
    for (int f = 0; f < 64; ++f) ref[p * 64 + f] = (float)x[f];
  }
  auto check = [&](const char* name) {
    hipMemcpy(got.data(), dout, vpoints * 64 * 4, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
      worst = fmax(worst, fabs((double)got[i] - ref[i]));
      scale = fmax(scale, fabs((double)ref[i]));
    }
    printf("check %-22s max |diff| %.3e of max |ref| %.3e  %s\n", name, worst, scale, worst <= 1e-5 * scale ? "ok" : "MISMATCH");
  };
  hipMemset(dout, 0, vpoints * 64 * 4);
  chain16<2><<<vb, vt, lds_bytes>>>(d16, dout, vtiles, 1, 1);
  check("16x16x4, 2 blocks");
  hipMemset(dout, 0, vpoints * 64 * 4);
  chain16<1><<<vb, vt, lds_bytes>>>(d16, dout, 2 * vtiles, 1, 1);
  check("16x16x4, 1 block");
  hipMemset(dout, 0, vpoints * 64 * 4);
  chain32<<<vb, vt, lds_bytes>>>(d32, dout, vtiles, 1, 1);
  check("32x32x2");
  // ---- timing
  const int blocks = 256 * 4, reps = 8;
  for (int wps = 1; wps <= 2; ++wps) {
    const int threads = 256 * wps;
    const long points = (long)blocks * (threads / 64) * 64 * 32;  // 64 tiles of 32 points per wave (128 of 16 for the one-block kernel)
    const double flop = (double)points * reps * NL * 64 * 64 * 2;
    double ms = time_kernel([&]() { chain16<1><<<blocks, threads, lds_bytes>>>(d16, dout, 128, reps, 0); });
    printf("waves/SIMD %d  16x16x4 one block   %.3f ms  %.1f TFLOP/s\n", wps, ms, flop / ms / 1e9);
    ms = time_kernel([&]() { chain16<2><<<blocks, threads, lds_bytes>>>(d16, dout, 64, reps, 0); });
    printf("waves/SIMD %d  16x16x4 two blocks  %.3f ms  %.1f TFLOP/s\n", wps, ms, flop / ms / 1e9);
    ms = time_kernel([&]() { chain32<<<blocks, threads, lds_bytes>>>(d32, dout, 64, reps, 0); });
    printf("waves/SIMD %d  32x32x2             %.3f ms  %.1f TFLOP/s\n", wps, ms, flop / ms / 1e9);
  }
  return 0;
}
