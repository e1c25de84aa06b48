// Fused Adam step over one parameter tensor (torch.optim.Adam semantics as configured by the
// reference: ns/engine/optimizers.py:133-140, ns/configs/method_configs.py:158-168 — lr 1e-2, betas (0.9, 0.999),
// eps 1e-15, L2 weight decay 1e-5 added to the gradient, no amsgrad).  One pass: 4 streams in (p, g, m, v), 3 out.
#include "common.hpp"
#include "adam_core.hpp"

namespace {

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gs) {
  const int64_t i0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4;
  if (i0 >= n) return;
  const ps::AdamHyper h{lr, b1, b2, eps, wd, gs};
  if (i0 + 3 < n) {
    f32x4 P = *reinterpret_cast<f32x4*>(p + i0), G = *reinterpret_cast<const f32x4*>(g + i0);
    f32x4 M = *reinterpret_cast<f32x4*>(m + i0), V = *reinterpret_cast<f32x4*>(v + i0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = P[k], mk = M[k], vk = V[k];
      ps::adam_update(pk, G[k], mk, vk, h, bc1, bc2_sqrt);
      P[k] = pk;
      M[k] = mk;
      V[k] = vk;
    }
    *reinterpret_cast<f32x4*>(p + i0) = P;
    *reinterpret_cast<f32x4*>(m + i0) = M;
    *reinterpret_cast<f32x4*>(v + i0) = V;
  } else {
    for (int64_t i = i0; i < n; ++i) ps::adam_update(p[i], g[i], m[i], v[i], h, bc1, bc2_sqrt);
  }
}

// The same update over up to kMaxRanges disjoint ranges of one flat buffer in ONE launch, each range with its own bias
// corrections (torch.optim.Adam keeps state["step"] per parameter and advances it only when that parameter has a gradient:
// ns/engine/optimizers.py:133-140 with zero_grad(set_to_none=True), ns/engine/trainer.py:470).  The table travels as a
// kernel argument, so there is no host->device copy; a block finds its range by a linear search over the block prefix.
//
// Device-decided ranges.  Whether a routed sub-field received samples this step is only known on the device (the router
// never synchronises with the host, csrc/route.hip).  Such a range carries a GROUP id >= 0: flags[group] != 0 means "received
// a gradient this step" and steps[group] is its torch-style step count, both in device memory.  A range whose flag is 0 is
// skipped entirely -- no moment decay, no weight-decay-only update, no step-count advance -- exactly like torch.optim.Adam on
// a parameter whose .grad is None (the reference's sub-field loop never calls an empty sub-field, ingp_field_ms.py:97-126,
// and DDP(find_unused_parameters=True) leaves its gradients None).  The step counts of the flagged groups are advanced by
// adam_commit_kernel AFTER all update launches of the step (stream order: no block reads a count that is being written).
constexpr int kMaxRanges = 32;
constexpr int kRangeBlockElems = 256 * 4 * 4;  // elements per workgroup: 256 threads x 4 vectors of 4
struct AdamRanges {
  int n;
  int64_t start[kMaxRanges], count[kMaxRanges];
  unsigned blk0[kMaxRanges + 1];
  float bc1[kMaxRanges], bc2_sqrt[kMaxRanges];  // host-decided ranges (group < 0)
  int group[kMaxRanges];
};

__global__ __launch_bounds__(256) void adam_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, AdamRanges R, float lr, float b1, float b2, float eps,
                                                          float wd, float gs, const int* __restrict__ flags, const int* __restrict__ steps) {
  int r = 0;
  while (r + 1 < R.n && blockIdx.x >= R.blk0[r + 1]) ++r;
  const int64_t base = R.start[r], n = R.count[r];
  float bc1 = R.bc1[r], bc2_sqrt = R.bc2_sqrt[r];
  const int grp = R.group[r];
  if (grp >= 0) {  // workgroup-uniform
    if (flags[grp] == 0) return;
    // two double-precision pow() per THREAD cost more than the block's 114 KB of traffic: one lane evaluates them
    __shared__ float s_bc[2];
    if (threadIdx.x == 0)
      ps::adam_bias_corrections(b1, b2, steps[grp] + 1, s_bc[0], s_bc[1]);
    __syncthreads();
    bc1 = s_bc[0];
    bc2_sqrt = s_bc[1];
  }
  const ps::AdamHyper h{lr, b1, b2, eps, wd, gs};
  const int64_t first = (int64_t)(blockIdx.x - R.blk0[r]) * kRangeBlockElems;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t i0 = first + (q * 256 + threadIdx.x) * 4;
    if (i0 >= n) continue;
    float* pp = p + base + i0;
    const float* gg = g + base + i0;
    float* mm = m + base + i0;
    float* vv = v + base + i0;
    if (i0 + 3 < n) {
      // every byte is touched once per step: nontemporal loads / stores keep the 26 GB stream of a production tile out of the way of
      // the caches (tools/microbench/adam_stream.hip: 6.11 -> 6.44 TB/s on 940 M parameters)
      f32x4 P = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pp)), G = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gg));
      f32x4 M = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mm)), V = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vv));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
      float pk = P[k], mk = M[k], vk = V[k];
      ps::adam_update(pk, G[k], mk, vk, h, bc1, bc2_sqrt);
      P[k] = pk;
      M[k] = mk;
      V[k] = vk;
    }
      __builtin_nontemporal_store(P, reinterpret_cast<f32x4*>(pp));
      __builtin_nontemporal_store(M, reinterpret_cast<f32x4*>(mm));
      __builtin_nontemporal_store(V, reinterpret_cast<f32x4*>(vv));
    } else {
      for (int64_t i = 0; i0 + i < n; ++i) ps::adam_update(pp[i], gg[i], mm[i], vv[i], h, bc1, bc2_sqrt);
    }
  }
}

// groups referenced by one ps_adam_step_ranges call (an optimizer step may be issued in several calls -- e.g. the proposal networks'
// ranges first and the fields' on another stream -- and every group's count must advance exactly once)
struct GroupMask {
  unsigned long long w[4];
};
__global__ void adam_commit_kernel(const int* __restrict__ flags, int* __restrict__ steps, int n_groups, GroupMask mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_groups && ((mask.w[i >> 6] >> (i & 63)) & 1ull) && flags[i] != 0) steps[i] += 1;
}

}  // namespace

// step >= 1; bias corrections bc1 = 1 - b1^step, bc2 = 1 - b2^step are evaluated on the host in double precision.
// grad_scale multiplies the gradient before the weight decay is added.  1.0 = the reference's default: PreSight runs
// update_grad_scaler=False, so optimizer.step() sees the 2**10-scaled gradients and weight_decay * p is added to THOSE
// (ns/engine/trainer.py:481-486, ns/engine/optimizers.py:133-140).  1 / loss scale = its update_grad_scaler=True branch, where
// GradScaler.step unscales before the optimizer runs (ns/engine/optimizers.py:118-131).
extern "C" int ps_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, float grad_scale, void* stream) {
  if (n == 0) return 0;
  PS_REQUIRE(step >= 1, "ps_adam_step: step counts from 1");
  PS_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
             "ps_adam_step: buffers must be 16-byte aligned (presight_amd.dist.FlatGrads pads its views accordingly)");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const int64_t threads = (n + 3) / 4;
  adam_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps,
                                                                                 weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
  PS_CHECK_LAUNCH();
}

// n_ranges disjoint, non-empty ranges [start[i], start[i]+count[i]) (floats, start a multiple of 4) of the flat buffers;
// start / count / step / group are HOST arrays.  group == NULL or group[i] < 0: range i is updated at the host-side step count
// step[i] >= 1.  group[i] >= 0: the device decides (see adam_ranges_kernel) from group_flags[group[i]] / group_steps[group[i]]
// (device int32 arrays of n_groups entries); after the last launch the step counts of the flagged groups THIS CALL REFERENCES are
// advanced by one (an optimizer step may be issued as several calls over disjoint ranges).
// ceil(n_ranges / 32) launches (+ 1 when groups are given).
extern "C" int ps_adam_step_ranges(float* p, const float* g, float* m, float* v, int n_ranges, const int64_t* start,
                                   const int64_t* count, const int* step, const int* group, const int32_t* group_flags,
                                   int32_t* group_steps, int n_groups, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, float grad_scale, void* stream) {
  PS_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "ps_adam_step_ranges: buffers must be 16-byte aligned");
  PS_REQUIRE(n_groups <= 256, "ps_adam_step_ranges: at most 256 device-decided groups");
  GroupMask mask{{0ull, 0ull, 0ull, 0ull}};
  for (int i = 0; i < n_ranges; ++i) {
    if (group != nullptr && group[i] >= 0 && group[i] < 256) mask.w[group[i] >> 6] |= 1ull << (group[i] & 63);
    PS_REQUIRE(count[i] > 0, "ps_adam_step_ranges: empty range");
    PS_REQUIRE((start[i] & 3) == 0, "ps_adam_step_ranges: range starts must be multiples of 4 floats");
    const int grp = group != nullptr ? group[i] : -1;
    PS_REQUIRE(grp >= 0 || step[i] >= 1, "ps_adam_step_ranges: step counts from 1");
    PS_REQUIRE(grp < n_groups && (grp < 0 || (group_flags != nullptr && group_steps != nullptr)), "ps_adam_step_ranges: bad group id");
  }
  for (int r0 = 0; r0 < n_ranges; r0 += kMaxRanges) {
    AdamRanges R;
    R.n = 0;
    unsigned blocks = 0;
    for (int i = r0; i < n_ranges && i < r0 + kMaxRanges; ++i) {
      const int grp = group != nullptr ? group[i] : -1;
      const double st = grp < 0 ? (double)step[i] : 1.0;
      R.start[R.n] = start[i];
      R.count[R.n] = count[i];
      R.blk0[R.n] = blocks;
      R.bc1[R.n] = (float)(1.0 - pow((double)beta1, st));
      R.bc2_sqrt[R.n] = (float)sqrt(1.0 - pow((double)beta2, st));
      R.group[R.n] = grp;
      blocks += (unsigned)((count[i] + kRangeBlockElems - 1) / kRangeBlockElems);
      ++R.n;
    }
    R.blk0[R.n] = blocks;
    adam_ranges_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, R, lr, beta1, beta2, eps, weight_decay, grad_scale,
                                                                group_flags, group_steps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ps_set_error(hipGetErrorString(e)); return (int)e; }
  }
  if (group != nullptr && n_groups > 0 && group_flags != nullptr)
    adam_commit_kernel<<<(unsigned)((n_groups + 255) / 256), 256, 0, (hipStream_t)stream>>>(group_flags, group_steps, n_groups, mask);
  PS_CHECK_LAUNCH();
}

// flags[group_of_field[k]] = 1 for every sub-field k of the routed layout (ps_ms_route's field_start) that received points;
// group_of_field[k] < 0: sub-field k has no device-decided group.  One tiny launch, no host synchronisation.
namespace {
__global__ void ms_mark_groups_kernel(const int* __restrict__ field_start, int K, const int* __restrict__ group_of_field,
                                      int* __restrict__ flags) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K && field_start[k + 1] > field_start[k] && group_of_field[k] >= 0) flags[group_of_field[k]] = 1;
}
}  // namespace
// zero up to 16 ranges of one buffer in ONE launch: the start-of-step clear of the gradient ranges the previous step wrote
// (optimizer.zero_grad, ns/engine/trainer.py:470) -- a handful of small, non-adjacent MLP ranges between the hash tables
namespace {
struct ZeroRanges {
  int64_t start[16], count[16];
};
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ p, ZeroRanges z) {
  float* q = p + z.start[blockIdx.y];
  const int64_t n = z.count[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) q[i] = 0.0f;
}
}  // namespace

extern "C" int ps_zero_ranges(float* p, int n_ranges, const int64_t* start, const int64_t* count, void* stream) {
  PS_REQUIRE(p != nullptr && start != nullptr && count != nullptr && n_ranges >= 1 && n_ranges <= 16, "ps_zero_ranges: 1..16 ranges");
  ZeroRanges z;
  int64_t longest = 0;
  for (int i = 0; i < 16; ++i) {
    z.start[i] = i < n_ranges ? start[i] : 0;
    z.count[i] = i < n_ranges ? count[i] : 0;
    PS_REQUIRE(z.start[i] >= 0 && z.count[i] >= 0, "ps_zero_ranges: negative range");
    longest = z.count[i] > longest ? z.count[i] : longest;
  }
  if (longest == 0) return 0;
  const int64_t bx = (longest + 256 * 8 - 1) / (256 * 8);
  zero_ranges_kernel<<<dim3((unsigned)(bx < 1 ? 1 : (bx > 1024 ? 1024 : bx)), (unsigned)n_ranges), 256, 0, (hipStream_t)stream>>>(p, z);
  PS_CHECK_LAUNCH();
}

extern "C" int ps_ms_mark_groups(const int32_t* field_start, int K, const int32_t* group_of_field, int32_t* flags, void* stream) {
  PS_REQUIRE(field_start != nullptr && group_of_field != nullptr && flags != nullptr && K >= 1, "ps_ms_mark_groups: null argument");
  ms_mark_groups_kernel<<<(unsigned)((K + 63) / 64), 64, 0, (hipStream_t)stream>>>(field_start, K, group_of_field, flags);
  PS_CHECK_LAUNCH();
}
