// Fused sky field (ns/fields/PreSight/sky_field.py:95-110): per RAY
//     d_enc     = SH4((dir + 1) / 2)                                   (base_field.py:136-142 + encodings.py:711-714)
//     rgb       = sigmoid(Linear(W) ReLU Linear(W) ReLU Linear(3) ([d_enc | appearance]))
//     semantics = Linear(W) ReLU Linear(W) ReLU Linear(64) (d_enc)
// in one kernel per direction instead of SH + concat + two operator-level MLPs (and their autograd glue), and — for the
// multi-sub-field sky model (sky_field_ms.py:97-114, routed by ray ORIGIN) — all K sub-fields in that one launch
// (ms_core.hpp).  Exact-fp32 MFMA stacks of mlp_core.hpp; W = 32 (PreSight's sky_mlp_dims), appearance dim <= 16.
#include "common.hpp"
#include "mlp_core.hpp"
#include "ms_core.hpp"
#include "pointwise_core.hpp"

namespace {

using namespace ps;

struct SkyArgs {
  const float* dirs;  // [R,3]
  const float* app;   // [R,A] or null
  int A;
  const float* packed;  // per sub-field [colour stack | semantic stack]
  int64_t N;            // rays (single field) or slots of the sorted layout
  float* rgb;           // [R,3]
  float* sem;           // [R,64] (null: no semantic head)
  const float* drgb;    // backward inputs
  const float* dsem;
  float* dapp;   // [R,A] written (every ray exactly once)
  float* gpart;  // one partial gradient block per workgroup
  const int* perm;
  const int* field_start;
  int K;
};

template <int KS0R>
struct SkyCfg {
  using Rgb = MlpT<KS0R, 2, 1, 3>;
  using Sem = MlpT<4, 2, 4, 3>;
  static constexpr int P_RGB = 0, P_SEM = Rgb::PACKED, PACKED = P_SEM + Sem::PACKED;
  static constexpr int G_RGB = 0, G_SEM = Rgb::GPACKED, GPACKED = G_SEM + Sem::GPACKED;
  static constexpr int FW_RGB = 0, FW_SEM = Rgb::FW, FW = FW_SEM + Sem::FW;
  static constexpr int SCR_ROWS = Rgb::SCRATCH_ROWS > Sem::SCRATCH_ROWS ? Rgb::SCRATCH_ROWS : Sem::SCRATCH_ROWS;
};

template <int PB, int KS0R, bool MS>
__device__ __forceinline__ void sky_inputs(const SkyArgs& a, int64_t first, float (&xs)[PB][4], float (&xr)[PB][KS0R], int64_t (&ray)[PB]) {
  const int lane = ps_lane(), j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int64_t op = ms_orig_index<MS>(a.perm, first + pb * 16 + j, a.N);
    ray[pb] = op;
    const int64_t r = op >= 0 ? op : 0;
    float sh[16];
    sh4((a.dirs[r * 3] + 1.0f) / 2.0f, (a.dirs[r * 3 + 1] + 1.0f) / 2.0f, (a.dirs[r * 3 + 2] + 1.0f) / 2.0f, sh);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float v0 = sh[4 * t], v1 = sh[4 * t + 1], v2 = sh[4 * t + 2], v3 = sh[4 * t + 3];
      xs[pb][t] = g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3));  // column 4t+g without dynamic register indexing
      xr[pb][t] = xs[pb][t];
    }
#pragma unroll
    for (int t = 4; t < KS0R; ++t) {
      const int c = 4 * (t - 4) + g;
      xr[pb][t] = (a.app != nullptr && c < a.A) ? a.app[r * a.A + c] : 0.0f;
    }
  }
}

struct SkyTiles {
  int64_t first_pt, end_pt;
  int j, n;
};

template <class C, int KS0R, int PB, bool MS>
__global__ __launch_bounds__(256) void sky_fwd_kernel(SkyArgs a) {
  SkyTiles tr{0, a.N, (int)blockIdx.x, (int)gridDim.x};
  if constexpr (MS) {
    const MsBlock mb = ms_block(a.field_start, a.K, gridDim.x, blockIdx.x);
    if (mb.k < 0) return;
    a.packed += (int64_t)mb.k * C::PACKED;
    tr = SkyTiles{mb.first_pt, mb.end_pt, mb.j, mb.n};
    a.N = mb.end_pt;
  }
  __shared__ __attribute__((aligned(16))) float lds[C::FW];
  for (int i = threadIdx.x * 4; i < C::Rgb::FW; i += 256 * 4)
    *reinterpret_cast<f32x4*>(lds + C::FW_RGB + i) = *reinterpret_cast<const f32x4*>(a.packed + C::P_RGB + i);
  for (int i = threadIdx.x * 4; i < C::Sem::FW; i += 256 * 4)
    *reinterpret_cast<f32x4*>(lds + C::FW_SEM + i) = *reinterpret_cast<const f32x4*>(a.packed + C::P_SEM + i);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = ps_lane(), g = lane >> 4;
  for (int64_t tile = (int64_t)tr.j * 4 + wave;; tile += (int64_t)tr.n * 4) {
    const int64_t first = tr.first_pt + tile * 16 * PB;
    if (first >= a.N) break;
    float xs[PB][4], xr[PB][KS0R];
    int64_t ray[PB];
    sky_inputs<PB, KS0R, MS>(a, first, xs, xr, ray);
    {
      float c1[PB][8], c2[PB][8], co[PB][4];
      mlp_forward<typename C::Rgb, PB>(LdsW{lds + C::FW_RGB}, xr, c1, c2, co);
      if (g == 0) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
          if (ray[pb] >= 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) a.rgb[ray[pb] * 3 + k] = 1.0f / (1.0f + expf(-co[pb][k]));
          }
      }
    }
    if (a.sem != nullptr) {
      float s1[PB][8], s2[PB][8], so[PB][16];
      mlp_forward<typename C::Sem, PB>(LdsW{lds + C::FW_SEM}, xs, s1, s2, so);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
        if (ray[pb] >= 0) {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
            *reinterpret_cast<f32x4*>(a.sem + ray[pb] * 64 + 16 * nb + 4 * g) =
                (f32x4){so[pb][4 * nb], so[pb][4 * nb + 1], so[pb][4 * nb + 2], so[pb][4 * nb + 3]};
        }
    }
  }
}

template <class C, int KS0R, int PB, bool MS>
__global__ __launch_bounds__(256) void sky_bwd_kernel(SkyArgs a) {
  SkyTiles tr{0, a.N, (int)blockIdx.x, (int)gridDim.x};
  if constexpr (MS) {
    const MsBlock mb = ms_block(a.field_start, a.K, gridDim.x, blockIdx.x);
    if (mb.k < 0) return;
    a.packed += (int64_t)mb.k * C::PACKED;
    tr = SkyTiles{mb.first_pt, mb.end_pt, mb.j, mb.n};
    a.N = mb.end_pt;
  }
  constexpr int SCR = C::SCR_ROWS * kScratchLd;
  __shared__ __attribute__((aligned(16))) float lds[C::GPACKED + 4 * SCR + 16];
  float* gacc = lds;
  int* locks = reinterpret_cast<int*>(lds + C::GPACKED + 4 * SCR);
  for (int i = threadIdx.x; i < C::GPACKED; i += 256) gacc[i] = 0.0f;
  if (threadIdx.x < 16) locks[threadIdx.x] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = ps_lane(), g = lane >> 4;
  float* scratch = lds + C::GPACKED + wave * SCR;
  const GlobalW pk = make_global_w(a.packed, C::PACKED);
  const GlobalW pk_rgb = pk.at(C::P_RGB), pk_sem = pk.at(C::P_SEM);
  // workgroup-uniform trip count (the dW flush takes LDS locks shared by the workgroup's waves); out-of-range tiles are masked
  for (int64_t base = (int64_t)tr.j * 4; tr.first_pt + base * 16 * PB < a.N; base += (int64_t)tr.n * 4) {
    const int64_t first = tr.first_pt + (base + wave) * 16 * PB;
    float xs[PB][4], xr[PB][KS0R];
    int64_t ray[PB];
    sky_inputs<PB, KS0R, MS>(a, first, xs, xr, ray);
    {
      float c1[PB][8], c2[PB][8], co[PB][4];
      mlp_forward<typename C::Rgb, PB>(pk_rgb, xr, c1, c2, co);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float d = 0.0f;
          if (k < 3 && g == 0 && ray[pb] >= 0 && a.drgb != nullptr) {
            const float s = 1.0f / (1.0f + expf(-co[pb][k]));
            d = a.drgb[ray[pb] * 3 + k] * s * (1.0f - s);
          }
          co[pb][k] = d;
        }
      float dxr[PB][C::Rgb::L0::IB * 4];
      mlp_backward<typename C::Rgb, PB, true>(pk_rgb, scratch, gacc + C::G_RGB, locks + 0, xr, c1, c2, co, dxr);
      if (a.dapp != nullptr) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
          if (ray[pb] >= 0) {
#pragma unroll
            for (int t = 4; t < KS0R; ++t) {
              const int c = 4 * (t - 4) + g;
              if (c < a.A) a.dapp[ray[pb] * a.A + c] = dxr[pb][t];
            }
          }
      }
    }
    if (a.dsem != nullptr) {
      float s1[PB][8], s2[PB][8], so[PB][16];
      mlp_forward<typename C::Sem, PB>(pk_sem, xs, s1, s2, so);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const float* src = a.dsem + (ray[pb] >= 0 ? ray[pb] : 0) * 64 + 4 * g;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (ray[pb] >= 0) d = *reinterpret_cast<const f32x4*>(src + 16 * nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) so[pb][4 * nb + r] = d[r];
        }
      }
      float dxs[PB][C::Sem::L0::IB * 4];
      mlp_backward<typename C::Sem, PB, false>(pk_sem, scratch, gacc + C::G_SEM, locks + 3, xs, s1, s2, so, dxs);
    }
  }
  __syncthreads();
  float* out = a.gpart + (size_t)blockIdx.x * C::GPACKED;
  for (int i = threadIdx.x; i < C::GPACKED; i += 256) out[i] = gacc[i];
}

constexpr int kSkyFwdPB = 4, kSkyBwdPB = 2, kSkyMaxBlocks = 256;

int sky_grid(int64_t N, int pb, int K, bool ms) {
  const int64_t tiles = (N + 16 * pb - 1) / (16 * pb);
  int64_t g = (tiles + 3) / 4 + (ms ? K : 0);
  if (g > kSkyMaxBlocks) g = kSkyMaxBlocks;
  if (g < (ms ? K : 1)) g = ms ? K : 1;
  return (int)g;
}

// appearance dims the colour stack is instantiated for: 16 + A inputs -> k-steps
#define PS_SKY_CFGS(X) \
  X(8)                 \
  X(5)                 \
  X(4)

}  // namespace

/* hidden width 32, semantic dim 64; A = appearance columns (k-steps = ceil((16 + A) / 4) must be instantiated) */
extern "C" int ps_sky_field_supported(int A, int width, int num_layers, int semantic_dim) {
  if (width != 32 || num_layers != 3 || (semantic_dim != 64 && semantic_dim != 0)) return 0;
  const int ks = (16 + A + 3) / 4;
#define X(k) \
  if (ks == k) return 1;
  PS_SKY_CFGS(X)
#undef X
  return 0;
}

extern "C" int ps_sky_field_sizes(int A, int64_t N, int K, int ms, int64_t* packed_floats, int64_t* grad_floats, int* n_parts,
                                  int64_t* offsets /*[4]: P_RGB, P_SEM, G_RGB, G_SEM*/) {
  const int ks = (16 + A + 3) / 4;
#define X(k)                            \
  if (ks == k) {                        \
    using C = SkyCfg<k>;                \
    *packed_floats = C::PACKED;         \
    *grad_floats = C::GPACKED;          \
    *n_parts = sky_grid(N, kSkyBwdPB, K, ms != 0); \
    if (offsets) {                      \
      offsets[0] = C::P_RGB;            \
      offsets[1] = C::P_SEM;            \
      offsets[2] = C::G_RGB;            \
      offsets[3] = C::G_SEM;            \
    }                                   \
    return 0;                           \
  }
  PS_SKY_CFGS(X)
#undef X
  ps_set_error("ps_sky_field: unsupported appearance dimension");
  return -2;
}

extern "C" int ps_sky_field_fwd(const float* dirs, const float* app, int A, const float* packed, int64_t N, float* rgb, float* sem,
                                const int32_t* perm, const int32_t* field_start, int K, void* stream) {
  if (N == 0) return 0;
  SkyArgs a{};
  a.dirs = dirs; a.app = app; a.A = A; a.packed = packed; a.N = N; a.rgb = rgb; a.sem = sem; a.perm = perm; a.field_start = field_start; a.K = K;
  const int ks = (16 + A + 3) / 4;
  hipStream_t s = (hipStream_t)stream;
#define X(k)                                                                                               \
  if (ks == k) {                                                                                           \
    using C = SkyCfg<k>;                                                                                   \
    if (perm != nullptr)                                                                                   \
      sky_fwd_kernel<C, k, kSkyFwdPB, true><<<sky_grid(N, kSkyFwdPB, K, true), 256, 0, s>>>(a);             \
    else                                                                                                   \
      sky_fwd_kernel<C, k, kSkyFwdPB, false><<<sky_grid(N, kSkyFwdPB, 1, false), 256, 0, s>>>(a);           \
    PS_CHECK_LAUNCH();                                                                                     \
  }
  PS_SKY_CFGS(X)
#undef X
  ps_set_error("ps_sky_field_fwd: unsupported appearance dimension");
  return -2;
}

extern "C" int ps_sky_field_bwd(const float* dirs, const float* app, int A, const float* packed, const float* drgb, const float* dsem,
                                int64_t N, float* dapp, float* gpart, const int32_t* perm, const int32_t* field_start, int K,
                                void* stream) {
  if (N == 0) return 0;
  SkyArgs a{};
  a.dirs = dirs; a.app = app; a.A = A; a.packed = packed; a.N = N; a.drgb = drgb; a.dsem = dsem; a.dapp = dapp; a.gpart = gpart;
  a.perm = perm; a.field_start = field_start; a.K = K;
  const int ks = (16 + A + 3) / 4;
  hipStream_t s = (hipStream_t)stream;
#define X(k)                                                                                               \
  if (ks == k) {                                                                                           \
    using C = SkyCfg<k>;                                                                                   \
    if (perm != nullptr)                                                                                   \
      sky_bwd_kernel<C, k, kSkyBwdPB, true><<<sky_grid(N, kSkyBwdPB, K, true), 256, 0, s>>>(a);             \
    else                                                                                                   \
      sky_bwd_kernel<C, k, kSkyBwdPB, false><<<sky_grid(N, kSkyBwdPB, 1, false), 256, 0, s>>>(a);           \
    PS_CHECK_LAUNCH();                                                                                     \
  }
  PS_SKY_CFGS(X)
#undef X
  ps_set_error("ps_sky_field_bwd: unsupported appearance dimension");
  return -2;
}
