/* CPU ORACLE (C part) — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the integer / index arithmetic of the hot path, used by tests/ to cross-check the
 * torch-CPU oracle (oracle/nerf_oracle.py) and the HIP kernels bit for bit:
 *   - ref_hash_indices : the 8 corner rows per level of the torch-fallback hash grid
 *                        (ns/field_components/encodings.py:324-361; int64 arithmetic like the reference)
 *   - ref_hash_encode  : gather + trilinear blend in the reference's order (encodings.py:363-384), fp32
 *   - ref_hash_scatter : its autograd (scatter-add of w_k * dout into the table gradient), fp32, sequential
 *   - ref_voxel_index  : Open3D voxel_down_sample_and_trace index rule used by
 *                        ns/scripts/extract_priors.py:216-245: floor((p - (min_bound - voxel/2)) / voxel), in double
 * Parity status: pinned through tests/test_oracle_c.py against the reference-generated fixture tests/golden/hashgrid.npz.
 * Build: make -C oracle   ->  oracle/_build/liboracle_ref.so
 */
#include <math.h>
#include <stdint.h>

#define PRIME_Y 2654435761LL
#define PRIME_Z 805459861LL

static int64_t hash3(int32_t x, int32_t y, int32_t z, int64_t T) {
  int64_t h = ((int64_t)x * 1LL) ^ ((int64_t)y * PRIME_Y) ^ ((int64_t)z * PRIME_Z);
  int64_t m = h % T; /* python-style modulo: result in [0, T) */
  if (m < 0) m += T;
  return m;
}

/* corner order: ccc cfc ffc fcc ccf cff fff fcf */
static void corners(const float* x, float s, int32_t c[3], int32_t f[3], float o[3]) {
  for (int k = 0; k < 3; ++k) {
    volatile float sc = x[k] * s; /* rounded to fp32 like torch */
    c[k] = (int32_t)ceilf(sc);
    f[k] = (int32_t)floorf(sc);
    o[k] = sc - floorf(sc);
  }
}

static void corner_ids(const int32_t c[3], const int32_t f[3], int64_t T, int64_t off, int64_t id[8]) {
  id[0] = hash3(c[0], c[1], c[2], T) + off;
  id[1] = hash3(c[0], f[1], c[2], T) + off;
  id[2] = hash3(f[0], f[1], c[2], T) + off;
  id[3] = hash3(f[0], c[1], c[2], T) + off;
  id[4] = hash3(c[0], c[1], f[2], T) + off;
  id[5] = hash3(c[0], f[1], f[2], T) + off;
  id[6] = hash3(f[0], f[1], f[2], T) + off;
  id[7] = hash3(f[0], c[1], f[2], T) + off;
}

void ref_hash_indices(const float* x, const float* scalings, int L, int log2T, int64_t N, int64_t* idx /*[N,L,8]*/) {
  const int64_t T = (int64_t)1 << log2T;
  for (int64_t n = 0; n < N; ++n)
    for (int l = 0; l < L; ++l) {
      int32_t c[3], f[3];
      float o[3];
      corners(x + 3 * n, scalings[l], c, f, o);
      corner_ids(c, f, T, (int64_t)l * T, idx + (n * L + l) * 8);
    }
}

void ref_hash_encode(const float* x, const float* table, const float* scalings, int L, int F, int log2T, int64_t N,
                     float* out /*[N,L*F]*/) {
  const int64_t T = (int64_t)1 << log2T;
  for (int64_t n = 0; n < N; ++n)
    for (int l = 0; l < L; ++l) {
      int32_t c[3], f[3];
      float o[3];
      int64_t id[8];
      corners(x + 3 * n, scalings[l], c, f, o);
      corner_ids(c, f, T, (int64_t)l * T, id);
      for (int k = 0; k < F; ++k) {
        volatile float v[8];
        for (int q = 0; q < 8; ++q) v[q] = table[id[q] * F + k];
        volatile float ux = 1.0f - o[0], uy = 1.0f - o[1], uz = 1.0f - o[2];
        volatile float a, b, f03, f12, f56, f47, f0312, f4756;
        a = v[0] * o[0]; b = v[3] * ux; f03 = a + b;
        a = v[1] * o[0]; b = v[2] * ux; f12 = a + b;
        a = v[5] * o[0]; b = v[6] * ux; f56 = a + b;
        a = v[4] * o[0]; b = v[7] * ux; f47 = a + b;
        a = f03 * o[1]; b = f12 * uy; f0312 = a + b;
        a = f47 * o[1]; b = f56 * uy; f4756 = a + b;
        a = f0312 * o[2]; b = f4756 * uz;
        out[n * (int64_t)(L * F) + l * F + k] = a + b;
      }
    }
}

void ref_hash_scatter(const float* x, const float* dout, const float* scalings, int L, int F, int log2T, int64_t N,
                      float* dtable /*[L*T,F], accumulated*/) {
  const int64_t T = (int64_t)1 << log2T;
  for (int64_t n = 0; n < N; ++n)
    for (int l = 0; l < L; ++l) {
      int32_t c[3], f[3];
      float o[3];
      int64_t id[8];
      corners(x + 3 * n, scalings[l], c, f, o);
      corner_ids(c, f, T, (int64_t)l * T, id);
      const float ux = 1.0f - o[0], uy = 1.0f - o[1], uz = 1.0f - o[2];
      const float w[8] = {o[0] * o[1] * o[2], o[0] * uy * o[2], ux * uy * o[2], ux * o[1] * o[2],
                          o[0] * o[1] * uz,   o[0] * uy * uz,   ux * uy * uz,   ux * o[1] * uz};
      for (int q = 0; q < 8; ++q)
        for (int k = 0; k < F; ++k) dtable[id[q] * F + k] += w[q] * dout[n * (int64_t)(L * F) + l * F + k];
    }
}

void ref_voxel_index(const float* pts, int64_t n, double voxel, const double* min_bound, int64_t* idx /*[n,3]*/) {
  for (int64_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) idx[i * 3 + k] = (int64_t)floor(((double)pts[i * 3 + k] - (min_bound[k] - voxel * 0.5)) / voxel);
}
