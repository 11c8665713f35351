// spz_ply_kernels.hip — .ply vertex rows <-> GaussianCloud arrays on the GPU.
//
// Replaces, in the reference (/root/reference/src/cc/load-spz.cc):
//   rows -> cloud : the AoS->SoA loop of loadSplatFromPly (:814-839, sh [channel][coeff] ->
//                   [coeff][channel]) and the convertCoordinates(RDF, to) pass that follows (:842)
//   cloud -> rows : the row assembly of saveSplatToPly with its from->RDF flips (:858-893)
//
// Both are pure shuffles (HBM-bound, no arithmetic beyond +-1 products).  A block owns a tile of
// 64 Gaussians: the tile's source bytes are read with lane-contiguous 16-byte loads into LDS, the
// permutation happens in LDS, and the destination is written lane-contiguously again, so neither
// side of the AoS<->SoA transpose touches HBM with a strided pattern.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "spz_amd.h"
#include "spz_common.hpp"

namespace {

using namespace spz_amd_detail;

// Non-temporal loads / stores (every byte is touched once; see spz_kernel_params.hpp on why pairs of
// kernels are tuned together).
#ifndef SPZ_PLY_NTL
#define SPZ_PLY_NTL 1
#endif
#ifndef SPZ_PLY_NTS
#define SPZ_PLY_NTS 1
#endif
#ifndef SPZ_PLY_LANE_IS_FIELD
#define SPZ_PLY_LANE_IS_FIELD 1
#endif
constexpr int kPlyBlock = 256;
constexpr int kPlyPoints = 64;  // Gaussians per tile
constexpr int kMaxStride = 255; // (64 * 255 + 45) * 4 B < 64 KiB of LDS

struct PlyParams {
  const float *rows_in;
  float *rows_out;
  const float *in[6];  // positions, scales, rotations, alphas, colors, sh
  float *out[6];
  unsigned long long n;
  spz_amd_ply_columns cols;
  uint32_t flip_p, flip_q, flip_sh15;
};

typedef float ply_v4f __attribute__((ext_vector_type(4)));
typedef ply_v4f ply_v4f_a4 __attribute__((aligned(4)));
typedef float ply_f32_a4 __attribute__((aligned(4)));

__device__ __forceinline__ F32x4 ply_load4(const float *p) {
#if SPZ_PLY_NTL
  ply_v4f t = __builtin_nontemporal_load(reinterpret_cast<const ply_v4f_a4 *>(p));
  F32x4 v;
  v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w;
  return v;
#else
  return *reinterpret_cast<const F32x4 *>(p);
#endif
}
__device__ __forceinline__ void ply_store4(float *p, F32x4 v) {
#if SPZ_PLY_NTS
  ply_v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<ply_v4f_a4 *>(p));
#else
  *reinterpret_cast<F32x4 *>(p) = v;
#endif
}
__device__ __forceinline__ void ply_store1(float *p, float v) {
#if SPZ_PLY_NTS
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

__device__ __forceinline__ void copy_to_lds(float *__restrict__ dst, const float *__restrict__ src, uint32_t count) {
  const uint32_t vec = count & ~3u;
  for (uint32_t i = threadIdx.x * 4u; i < vec; i += kPlyBlock * 4u) {
    F32x4 v = ply_load4(src + i);
    dst[i] = v.x; dst[i + 1] = v.y; dst[i + 2] = v.z; dst[i + 3] = v.w;
  }
  for (uint32_t i = vec + threadIdx.x; i < count; i += kPlyBlock) dst[i] = src[i];
}

__global__ __launch_bounds__(kPlyBlock) void spz_ply_rows_to_cloud_kernel(const PlyParams p, uint32_t inv_d_magic) {
  extern __shared__ float tile[];  // [np][stride] rows, then d words: sh element e -> column | flip << 16
  const uint32_t stride = (uint32_t)p.cols.stride;
  const uint32_t shd = (uint32_t)p.cols.sh_dim;
  const uint32_t d = shd * 3u;
  const unsigned long long first = (unsigned long long)blockIdx.x * kPlyPoints;
  const unsigned long long left = p.n - first;
  const uint32_t np = left < (unsigned long long)kPlyPoints ? (uint32_t)left : (uint32_t)kPlyPoints;
  uint32_t *shtab = reinterpret_cast<uint32_t *>(tile + (size_t)kPlyPoints * stride);
  const uint32_t tid = threadIdx.x;
  if (tid < d) {
    // cloud element e = coeff j, channel ch  <-  file column sh[j + ch * shDim]  (load-spz.cc:834-838)
    const uint32_t j = tid / 3u, ch = tid - j * 3u;
    shtab[tid] = (uint32_t)p.cols.sh[j + ch * shd] | (((p.flip_sh15 >> j) & 1u) << 16);
  }
  copy_to_lds(tile, p.rows_in + first * stride, np * stride);
  __syncthreads();
  // positions: value * flipP[axis] (convertCoordinates multiplies every element, load-spz.cc:842)
  for (uint32_t i = tid; i < np * 3u; i += kPlyBlock) {
    const uint32_t pt = i / 3u, a = i - pt * 3u;
    ply_store1(&p.out[0][first * 3u + i], mul_pm1(tile[pt * stride + (uint32_t)p.cols.position[a]], (p.flip_p >> a) & 1u));
  }
  for (uint32_t i = tid; i < np * 3u; i += kPlyBlock) {
    const uint32_t pt = i / 3u, a = i - pt * 3u;
    ply_store1(&p.out[1][first * 3u + i], tile[pt * stride + (uint32_t)p.cols.scale[a]]);
    ply_store1(&p.out[4][first * 3u + i], tile[pt * stride + (uint32_t)p.cols.color[a]]);
  }
  for (uint32_t i = tid; i < np * 4u; i += kPlyBlock) {
    const uint32_t pt = i >> 2, c = i & 3u;
    const float v = tile[pt * stride + (uint32_t)p.cols.rotation[c]];
    ply_store1(&p.out[2][first * 4u + i], (c < 3u) ? mul_pm1(v, (p.flip_q >> c) & 1u) : v);  // w is never multiplied
  }
  for (uint32_t i = tid; i < np; i += kPlyBlock) ply_store1(&p.out[3][first + i], tile[i * stride + (uint32_t)p.cols.alpha]);
  // sh: file [channel][coeff] -> cloud [coeff][channel], times flipSh[coeff]; 76 % of the output bytes
  if (d) {
    auto element = [&](uint32_t i) -> float {
      const uint32_t pt = __umulhi(i, inv_d_magic);  // i / d, exact for i < 2^16
      const uint32_t w = shtab[i - pt * d];
      return mul_pm1(tile[pt * stride + (w & 0xffffu)], w >> 16);
    };
    float *dst = p.out[5] + first * d;
    const uint32_t total = np * d, vec = total & ~3u;
    for (uint32_t i = tid * 4u; i < vec; i += kPlyBlock * 4u) {
      F32x4 v;
      v.x = element(i); v.y = element(i + 1); v.z = element(i + 2); v.w = element(i + 3);
      ply_store4(dst + i, v);
    }
    for (uint32_t i = vec + tid; i < total; i += kPlyBlock) dst[i] = element(i);
  }
}

// Row field f -> where its value sits in the staged SoA tile: one packed word per field, built once
// per block, so that the per-element work is a multiply-shift, two LDS reads and a sign flip instead
// of divisions by the runtime row length.
//   bits  0..15  LDS float offset of point 0's value
//   bits 16..23  floats between consecutive points
//   bits 24..25  0 = copy bits, 1 = times +1, 2 = times -1 (both multiply like the reference: NaNs are
//                quieted), 3 = constant 0.0f (normals)
__device__ __forceinline__ uint32_t ply_field_word(uint32_t f, uint32_t shd, uint32_t flip_p, uint32_t flip_q,
                                                   uint32_t flip_sh15) {
  const uint32_t P = kPlyPoints, d = shd * 3u;
  auto word = [](uint32_t off, uint32_t step, uint32_t mode) { return off | (step << 16) | (mode << 24); };
  if (f < 3u) return word(f, 3u, 1u + ((flip_p >> f) & 1u));                    // load-spz.cc:861-863
  if (f < 6u) return word(0u, 0u, 3u);                                           // normals, :865
  if (f < 9u) return word(11u * P + (f - 6u), 3u, 0u);                           // colours, :867-869
  if (f < 9u + d) {                                                              // sh, :872-880
    const uint32_t g = f - 9u, ch = g / shd, j = g - ch * shd;
    return word(14u * P + j * 3u + ch, d, 1u + ((flip_sh15 >> j) & 1u));
  }
  const uint32_t t = f - 9u - d;
  if (t == 0u) return word(10u * P, 1u, 0u);                                     // alpha, :882
  if (t < 4u) return word(3u * P + (t - 1u), 3u, 0u);                            // scales, :884-886
  if (t == 4u) return word(6u * P + 3u, 4u, 0u);                                 // rot w, :888
  return word(6u * P + (t - 5u), 4u, 1u + ((flip_q >> (t - 5u)) & 1u));          // rot xyz, :889-891
}

__global__ __launch_bounds__(kPlyBlock) void spz_cloud_to_ply_rows_kernel(const PlyParams p, uint32_t inv_d_magic) {
  extern __shared__ float tile[];  // [pos 3P | scale 3P | rot 4P | alpha P | color 3P | sh dP | field words D]
  const uint32_t shd = (uint32_t)p.cols.sh_dim;
  const uint32_t d = shd * 3u;
  const uint32_t D = 17u + d;  // floats per row
  const unsigned long long first = (unsigned long long)blockIdx.x * kPlyPoints;
  const unsigned long long left = p.n - first;
  const uint32_t np = left < (unsigned long long)kPlyPoints ? (uint32_t)left : (uint32_t)kPlyPoints;
  float *pos = tile, *scl = tile + 3 * kPlyPoints, *rot = tile + 6 * kPlyPoints, *alp = tile + 10 * kPlyPoints,
        *col = tile + 11 * kPlyPoints, *sh = tile + 14 * kPlyPoints;
  uint32_t *words = reinterpret_cast<uint32_t *>(tile + (14u + d) * kPlyPoints);
  if (threadIdx.x < D) words[threadIdx.x] = ply_field_word(threadIdx.x, shd, p.flip_p, p.flip_q, p.flip_sh15);
  copy_to_lds(pos, p.in[0] + first * 3u, np * 3u);
  copy_to_lds(scl, p.in[1] + first * 3u, np * 3u);
  copy_to_lds(rot, p.in[2] + first * 4u, np * 4u);
  copy_to_lds(alp, p.in[3] + first, np);
  copy_to_lds(col, p.in[4] + first * 3u, np * 3u);
  if (d) copy_to_lds(sh, p.in[5] + first * d, np * d);
  __syncthreads();
  float *dst = p.rows_out + first * D;
  const uint32_t total = np * D;
#if SPZ_PLY_LANE_IS_FIELD
  {
    // lane = (point within the wave's group, field): the field word is loop-invariant per lane
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t ppw = 64u / D;  // points per wave per step (D <= 62)
    const uint32_t sub = __umulhi(lane, inv_d_magic), f = lane - sub * D;
    if (sub < ppw) {
      const uint32_t w = words[f];
      const uint32_t mode = w >> 24, off = w & 0xffffu, step = (w >> 16) & 0xffu;
      for (uint32_t pt = wave * ppw + sub; pt < np; pt += (kPlyBlock / 64) * ppw) {
        const uint32_t b = __float_as_uint(tile[off + pt * step]);
        const uint32_t mul = is_nan_bits(b) ? (b | 0x00400000u) : (b ^ ((mode & 2u) << 30));
        ply_store1(&dst[pt * D + f], __uint_as_float(mode == 0u ? b : (mode == 3u ? 0u : mul)));
      }
    }
    return;
  }
#endif
  auto element = [&](uint32_t i) -> float {
    const uint32_t pt = __umulhi(i, inv_d_magic);  // i / D, exact for i < 2^16 (i < 64 * 62)
    const uint32_t w = words[i - pt * D];
    const uint32_t mode = w >> 24;
    const uint32_t b = __float_as_uint(tile[(w & 0xffffu) + pt * ((w >> 16) & 0xffu)]);
    const uint32_t mul = is_nan_bits(b) ? (b | 0x00400000u) : (b ^ ((mode & 2u) << 30));
    return __uint_as_float(mode == 0u ? b : (mode == 3u ? 0u : mul));
  };
  const uint32_t vec = total & ~3u;
  for (uint32_t i = threadIdx.x * 4u; i < vec; i += kPlyBlock * 4u) {
    F32x4 v;
    v.x = element(i); v.y = element(i + 1); v.z = element(i + 2); v.w = element(i + 3);
    ply_store4(dst + i, v);
  }
  for (uint32_t i = vec + threadIdx.x; i < total; i += kPlyBlock) dst[i] = element(i);
}

int check_columns(const spz_amd_ply_columns *c) {
  if (c == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  if (c->sh_dim < 0 || c->sh_dim > 15 || c->stride < 14 + 3 * c->sh_dim || c->stride > kMaxStride) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  auto ok = [&](int32_t v) { return v >= 0 && v < c->stride; };
  for (int i = 0; i < 3; ++i) {
    if (!ok(c->position[i]) || !ok(c->scale[i]) || !ok(c->color[i])) return SPZ_AMD_ERR_INVALID_ARG;
  }
  for (int i = 0; i < 4; ++i) {
    if (!ok(c->rotation[i])) return SPZ_AMD_ERR_INVALID_ARG;
  }
  if (!ok(c->alpha)) return SPZ_AMD_ERR_INVALID_ARG;
  for (int i = 0; i < 3 * c->sh_dim; ++i) {
    if (!ok(c->sh[i])) return SPZ_AMD_ERR_INVALID_ARG;
  }
  return SPZ_AMD_OK;
}

void set_flips(PlyParams *p, int from, int to) {
  const FlipMasks fm = flip_masks(from, to);
  p->flip_p = fm.p;
  p->flip_q = fm.q;
  p->flip_sh15 = fm.sh15;
}

uint32_t tiles(uint64_t n) { return (uint32_t)((n + kPlyPoints - 1) / kPlyPoints); }

}  // namespace

extern "C" {

int spz_amd_ply_default_columns(int sh_dim, spz_amd_ply_columns *out) {
  if (out == nullptr || sh_dim < 0 || sh_dim > 15) return SPZ_AMD_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  out->sh_dim = sh_dim;
  out->stride = 17 + 3 * sh_dim;
  for (int i = 0; i < 3; ++i) {
    out->position[i] = i;
    out->color[i] = 6 + i;
  }
  for (int i = 0; i < 3 * sh_dim; ++i) out->sh[i] = 9 + i;
  const int t = 9 + 3 * sh_dim;
  out->alpha = t;
  for (int i = 0; i < 3; ++i) out->scale[i] = t + 1 + i;
  out->rotation[3] = t + 4;                                 // rot_0 = w
  for (int i = 0; i < 3; ++i) out->rotation[i] = t + 5 + i;  // rot_1..3 = x y z
  return SPZ_AMD_OK;
}

int spz_amd_ply_rows_to_cloud_device(const float *d_rows, uint64_t n, const spz_amd_ply_columns *cols,
                                     int to_coord, const spz_amd_cloud_out *cl, void *hip_stream) {
  int rc = check_columns(cols);
  if (rc != SPZ_AMD_OK) return rc;
  if (cl == nullptr || !valid_coord(to_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  if (n == 0) return SPZ_AMD_OK;
  if (!d_rows || !cl->positions || !cl->scales || !cl->rotations || !cl->alphas || !cl->colors ||
      (cols->sh_dim > 0 && !cl->sh) || n > 0xffffffffull * kPlyPoints) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  int device = 0;
  rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  PlyParams p = {};
  p.rows_in = d_rows;
  float *outs[6] = {cl->positions, cl->scales, cl->rotations, cl->alphas, cl->colors, cl->sh};
  for (int i = 0; i < 6; ++i) p.out[i] = outs[i];
  p.n = n;
  p.cols = *cols;
  set_flips(&p, SPZ_AMD_RDF, to_coord);  // load-spz.cc:842
  const uint32_t d = 3u * (uint32_t)cols->sh_dim;
  const size_t lds = ((size_t)kPlyPoints * (size_t)cols->stride + d) * sizeof(float);
  const uint32_t magic = d ? (uint32_t)(0x100000000ull / d) + 1u : 0u;  // umulhi(i, magic) == i / d for i < 65536
  hipLaunchKernelGGL(spz_ply_rows_to_cloud_kernel, dim3(tiles(n)), dim3(kPlyBlock), lds,
                     static_cast<hipStream_t>(hip_stream), p, magic);
  SPZ_HIP_TRY(hipGetLastError());
  return SPZ_AMD_OK;
}

int spz_amd_cloud_to_ply_rows_device(const spz_amd_cloud_in *cl, uint64_t n, int sh_dim, int from_coord,
                                     float *d_rows, void *hip_stream) {
  if (cl == nullptr || sh_dim < 0 || sh_dim > 15 || !valid_coord(from_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  if (n == 0) return SPZ_AMD_OK;
  if (!d_rows || !cl->positions || !cl->scales || !cl->rotations || !cl->alphas || !cl->colors ||
      (sh_dim > 0 && !cl->sh) || n > 0xffffffffull * kPlyPoints) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  int device = 0;
  int rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  PlyParams p = {};
  p.rows_out = d_rows;
  const float *ins[6] = {cl->positions, cl->scales, cl->rotations, cl->alphas, cl->colors, cl->sh};
  for (int i = 0; i < 6; ++i) p.in[i] = ins[i];
  p.n = n;
  p.cols.sh_dim = sh_dim;
  p.cols.stride = 17 + 3 * sh_dim;
  set_flips(&p, from_coord, SPZ_AMD_RDF);  // load-spz.cc:856
  const uint32_t D = 17u + 3u * (uint32_t)sh_dim;
  const size_t lds = ((size_t)kPlyPoints * (size_t)(14 + 3 * sh_dim) + D) * sizeof(float);
  const uint32_t magic = (uint32_t)(0x100000000ull / D) + 1u;  // umulhi(i, magic) == i / D for i < 65536
  hipLaunchKernelGGL(spz_cloud_to_ply_rows_kernel, dim3(tiles(n)), dim3(kPlyBlock), lds,
                     static_cast<hipStream_t>(hip_stream), p, magic);
  SPZ_HIP_TRY(hipGetLastError());
  return SPZ_AMD_OK;
}

int spz_amd_ply_rows_to_cloud_host(const float *h_rows, uint64_t n, const spz_amd_ply_columns *cols, int to_coord,
                                   const spz_amd_cloud_out *h, int device) {
  int rc = check_columns(cols);
  if (rc != SPZ_AMD_OK) return rc;
  if (h == nullptr || !valid_coord(to_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  if (n == 0) return SPZ_AMD_OK;
  const size_t d = (size_t)cols->sh_dim * 3;
  if (!h_rows || !h->positions || !h->scales || !h->rotations || !h->alphas || !h->colors || (d && !h->sh)) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  const size_t cnt[6] = {n * 3, n * 3, n * 4, n, n * 3, n * d};
  float *dst[6] = {h->positions, h->scales, h->rotations, h->alphas, h->colors, h->sh};
  const size_t row_bytes = n * (size_t)cols->stride * sizeof(float);
  size_t total = Workspace::aligned(row_bytes);
  for (int i = 0; i < 6; ++i) total += Workspace::aligned(cnt[i] * sizeof(float));
  Workspace ws;
  rc = ws.open(device, total);
  if (rc != SPZ_AMD_OK) return rc;
  void *rows = ws.take(row_bytes);
  SPZ_HIP_TRY(hipMemcpyAsync(rows, h_rows, row_bytes, hipMemcpyHostToDevice, nullptr));
  void *fb[6];
  for (int i = 0; i < 6; ++i) fb[i] = ws.take(cnt[i] * sizeof(float));
  spz_amd_cloud_out dc = {(float *)fb[0], (float *)fb[1], (float *)fb[2], (float *)fb[3], (float *)fb[4], (float *)fb[5]};
  rc = spz_amd_ply_rows_to_cloud_device((const float *)rows, n, cols, to_coord, &dc, nullptr);
  if (rc != SPZ_AMD_OK) return rc;
  for (int i = 0; i < 6; ++i) {
    if (cnt[i]) SPZ_HIP_TRY(hipMemcpyAsync(dst[i], fb[i], cnt[i] * sizeof(float), hipMemcpyDeviceToHost, nullptr));
  }
  SPZ_HIP_TRY(hipStreamSynchronize(nullptr));
  return SPZ_AMD_OK;
}

int spz_amd_cloud_to_ply_rows_host(const spz_amd_cloud_in *h, uint64_t n, int sh_dim, int from_coord, float *h_rows,
                                   int device) {
  if (h == nullptr || sh_dim < 0 || sh_dim > 15 || !valid_coord(from_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  if (n == 0) return SPZ_AMD_OK;
  const size_t d = (size_t)sh_dim * 3;
  if (!h_rows || !h->positions || !h->scales || !h->rotations || !h->alphas || !h->colors || (d && !h->sh)) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  const size_t cnt[6] = {n * 3, n * 3, n * 4, n, n * 3, n * d};
  const float *src[6] = {h->positions, h->scales, h->rotations, h->alphas, h->colors, h->sh};
  const size_t row_bytes = n * (size_t)(17 + d) * sizeof(float);
  size_t total = Workspace::aligned(row_bytes);
  for (int i = 0; i < 6; ++i) total += Workspace::aligned(cnt[i] * sizeof(float));
  Workspace ws;
  rc = ws.open(device, total);
  if (rc != SPZ_AMD_OK) return rc;
  void *fb[6];
  for (int i = 0; i < 6; ++i) {
    fb[i] = ws.take(cnt[i] * sizeof(float));
    if (cnt[i]) SPZ_HIP_TRY(hipMemcpyAsync(fb[i], src[i], cnt[i] * sizeof(float), hipMemcpyHostToDevice, nullptr));
  }
  void *rows = ws.take(row_bytes);
  spz_amd_cloud_in dc = {(const float *)fb[0], (const float *)fb[1], (const float *)fb[2],
                         (const float *)fb[3], (const float *)fb[4], (const float *)fb[5]};
  rc = spz_amd_cloud_to_ply_rows_device(&dc, n, sh_dim, from_coord, (float *)rows, nullptr);
  if (rc != SPZ_AMD_OK) return rc;
  SPZ_HIP_TRY(hipMemcpy(h_rows, rows, row_bytes, hipMemcpyDeviceToHost));
  return SPZ_AMD_OK;
}

}  // extern "C"
