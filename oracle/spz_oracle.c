/*
 * spz_oracle.c — plain-C CPU restatement of the reference SPZ pack/unpack path.
 *
 * TEST INFRASTRUCTURE ONLY (see spz_oracle.h).  Parity: PINNED against the
 * reference's own C++ (oracle/_ref/libspz_ref.so, built by oracle/Makefile from
 * /root/reference/src/cc) and the golden vectors in tests/golden/.
 *
 * Every function cites the reference lines it restates.  Arithmetic is kept in
 * the reference's exact order: one rounding per operation, no FMA (build with
 * -ffp-contract=off on an x86-64 baseline target, like the reference's own
 * CMake build which emits no FMA), libm expf/logf/roundf/powf as the reference
 * calls them.  Conversions that are undefined behaviour in the reference
 * (float -> int of NaN / out-of-range values) are written with the same cast
 * so the same compiler emits the same cvttss2si; tests do not pin that domain.
 */
#include "spz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* load-spz.cc:45-46 */
static const float kColorScale = 0.15f;
static const float kSqrt1_2 = (float)0.707106781186547524401;

/* load-spz.cc:129-139 */
#define SPZO_MAGIC 0x5053474eu
#define SPZO_FLAG_ANTIALIASED 0x1
#define SPZO_MAX_POINTS 10000000u /* load-spz.cc:549 */

/* ---- splat-types.h:43-53 axesMatch ---------------------------------------- */
static void axes_match(int a, int b, int m[3]) {
  int aNum = a - 1, bNum = b - 1;
  if (aNum < 0 || bNum < 0) {
    m[0] = m[1] = m[2] = 1;
    return;
  }
  m[0] = ((aNum >> 0) & 1) == ((bNum >> 0) & 1);
  m[1] = ((aNum >> 1) & 1) == ((bNum >> 1) & 1);
  m[2] = ((aNum >> 2) & 1) == ((bNum >> 2) & 1);
}

/* ---- splat-types.h:55-81 coordinateConverter ------------------------------ */
void spzo_coordinate_converter(int from, int to, spzo_converter_t *c) {
  int m[3];
  axes_match(from, to, m);
  float x = m[0] ? 1.0f : -1.0f;
  float y = m[1] ? 1.0f : -1.0f;
  float z = m[2] ? 1.0f : -1.0f;
  c->flipP[0] = x;
  c->flipP[1] = y;
  c->flipP[2] = z;
  c->flipQ[0] = y * z;
  c->flipQ[1] = x * z;
  c->flipQ[2] = x * y;
  const float sh[15] = {y, z, x, x * y, y * z, 1.0f, x * z, 1.0f, y, x * y * z, y, z, x, z, x};
  memcpy(c->flipSh, sh, sizeof(sh));
}

/* ---- load-spz.cc:58-72 ---------------------------------------------------- */
int spzo_dim_for_degree(int degree) {
  switch (degree) {
    case 0: return 0;
    case 1: return 3;
    case 2: return 8;
    case 3: return 15;
    default: return 0;
  }
}

/* ---- load-spz.cc:74 toUint8 ------------------------------------------------ */
uint8_t spzo_to_uint8(float x) {
  float r = roundf(x);
  /* std::clamp(v, lo, hi): (v < lo) ? lo : (hi < v) ? hi : v */
  r = (r < 0.0f) ? 0.0f : ((255.0f < r) ? 255.0f : r);
  return (uint8_t)r;
}

/* ---- load-spz.cc:77-81 quantizeSH ------------------------------------------ */
uint8_t spzo_quantize_sh(float x, int bucketSize) {
  int32_t q = (int)(roundf(x * 128.0f) + 128.0f);
  q = (q + bucketSize / 2) / bucketSize * bucketSize;
  q = (q < 0) ? 0 : ((255 < q) ? 255 : q);
  return (uint8_t)q;
}

/* ---- load-spz.cc:83 -------------------------------------------------------- */
float spzo_unquantize_sh(uint8_t x) { return ((float)x - 128.0f) / 128.0f; }

/* ---- load-spz.cc:85,87 ------------------------------------------------------ */
float spzo_sigmoid(float x) { return 1 / (1 + expf(-x)); }
float spzo_inv_sigmoid(float x) { return logf(x / (1.0f - x)); }

/* load-spz.cc:301 / :518 / :522 composites */
uint8_t spzo_alpha_byte(float a) { return spzo_to_uint8(spzo_sigmoid(a) * 255.0f); }
float spzo_alpha_value(uint8_t b) { return spzo_inv_sigmoid(b / 255.0f); }
float spzo_color_value(uint8_t b) { return ((b / 255.0f) - 0.5f) / kColorScale; }

/* ---- splat-types.cc:8-27 halfToFloat ---------------------------------------- */
float spzo_half_to_float(uint16_t h) {
  int sgn = ((h >> 15) & 0x1);
  int exponent = ((h >> 10) & 0x1f);
  int mantissa = h & 0x3ff;
  float signMul = sgn == 1 ? -1.0 : 1.0;
  if (exponent == 0) {
    return signMul * powf(2.0f, -14.0f) * (float)mantissa / 1024.0f;
  }
  if (exponent == 31) {
    return mantissa != 0 ? (float)NAN : signMul * (float)INFINITY;
  }
  return signMul * powf(2.0f, (float)exponent - 15.0f) * (1.0f + (float)mantissa / 1024.0f);
}

/* ---- load-spz.cc:216-255 packQuaternionSmallestThree ------------------------ */
void spzo_pack_quat_smallest_three(uint8_t r[4], const float rotation[4],
                                   const spzo_converter_t *c) {
  /* normalized(Quat4f) splat-types.cc:71-74 */
  float q[4];
  float norm = sqrtf(rotation[0] * rotation[0] + rotation[1] * rotation[1] +
                     rotation[2] * rotation[2] + rotation[3] * rotation[3]);
  q[0] = rotation[0] / norm;
  q[1] = rotation[1] / norm;
  q[2] = rotation[2] / norm;
  q[3] = rotation[3] / norm;
  q[0] *= c->flipQ[0];
  q[1] *= c->flipQ[1];
  q[2] *= c->flipQ[2];

  unsigned iLargest = 0;
  for (unsigned i = 1; i < 4; ++i) {
    if (fabsf(q[i]) > fabsf(q[iLargest])) {
      iLargest = i;
    }
  }
  unsigned negate = q[iLargest] < 0;
  uint32_t comp = iLargest;
  for (unsigned i = 0; i < 4; ++i) {
    if (i != iLargest) {
      uint32_t negbit = (unsigned)(q[i] < 0) ^ negate;
      uint32_t mag = (uint32_t)((float)((1u << 9u) - 1u) * (fabsf(q[i]) / kSqrt1_2) + 0.5f);
      comp = (comp << 10u) | (negbit << 9u) | mag;
    }
  }
  r[0] = comp & 0xff;
  r[1] = (comp >> 8) & 0xff;
  r[2] = (comp >> 16) & 0xff;
  r[3] = (comp >> 24) & 0xff;
}

/* ---- load-spz.cc:333-345 unpackQuaternionFirstThree ------------------------- */
void spzo_unpack_quat_first_three(float rotation[4], const uint8_t r[3],
                                  const spzo_converter_t *c) {
  float xyz[3];
  for (int i = 0; i < 3; i++) {
    /* times(Vec3f, 1/127.5f) then plus(.., {-1,-1,-1}) then times(.., flipQ) */
    float t = (float)r[i] * (1.0f / 127.5f);
    t = t + (-1.0f);
    xyz[i] = t * c->flipQ[i];
  }
  rotation[0] = xyz[0];
  rotation[1] = xyz[1];
  rotation[2] = xyz[2];
  /* squaredNorm = dot(v, v) = a0*b0 + a1*b1 + a2*b2 (splat-types.h:211-215) */
  float sq = xyz[0] * xyz[0] + xyz[1] * xyz[1] + xyz[2] * xyz[2];
  float d = 1.0f - sq;
  float m = (0.0f < d) ? d : 0.0f; /* std::max(0.0f, d) */
  rotation[3] = sqrtf(m);
}

/* ---- load-spz.cc:347-381 unpackQuaternionSmallestThree ---------------------- */
void spzo_unpack_quat_smallest_three(float rotation[4], const uint8_t r[4],
                                     const spzo_converter_t *c) {
  uint32_t comp = (uint32_t)r[0] + ((uint32_t)r[1] << 8) + ((uint32_t)r[2] << 16) +
                  ((uint32_t)r[3] << 24);
  const uint32_t c_mask = (1u << 9u) - 1u;
  const int i_largest = comp >> 30;
  float sum_squares = 0;
  for (int i = 3; i >= 0; --i) {
    if (i != i_largest) {
      uint32_t mag = comp & c_mask;
      uint32_t negbit = (comp >> 9u) & 0x1u;
      comp = comp >> 10u;
      rotation[i] = kSqrt1_2 * ((float)mag) / (float)c_mask;
      if (negbit == 1) {
        rotation[i] = -rotation[i];
      }
      sum_squares += rotation[i] * rotation[i];
    }
  }
  /* unqualified sqrt(float) inside namespace spz resolves to ::sqrt(double) */
  rotation[i_largest] = (float)sqrt((double)(1.0f - sum_squares));
  for (int i = 0; i < 3; i++) {
    rotation[i] *= c->flipQ[i];
  }
}

/* ---- stream geometry (load-spz.cc:533-546, :571-590) ------------------------ */
size_t spzo_stream_size(int64_t n, int sh_degree, int version) {
  size_t posB = (version == 1) ? 6 : 9;
  size_t rotB = (version >= 3) ? 4 : 3;
  size_t shB = (size_t)spzo_dim_for_degree(sh_degree) * 3;
  return 16 + (size_t)n * (posB + 1 + 3 + 3 + rotB + shB);
}

static void put_u32(uint8_t *p, uint32_t v) {
  p[0] = v & 0xff;
  p[1] = (v >> 8) & 0xff;
  p[2] = (v >> 16) & 0xff;
  p[3] = (v >> 24) & 0xff;
}
static uint32_t get_u32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* ---- packGaussians load-spz.cc:257-331 + serialize :533-546 ------------------ */
size_t spzo_pack(const float *positions, const float *scales, const float *rotations,
                 const float *alphas, const float *colors, const float *sh,
                 int32_t numPoints, int shDegree, int antialiased, int from_coord,
                 int version, uint8_t *stream) {
  if (numPoints < 0 || shDegree < 0 || shDegree > 3) return 0; /* checkSizes :106-109 */
  if (version != 3 && version != 2) return 0;
  const size_t N = (size_t)numPoints;
  const int shDim = spzo_dim_for_degree(shDegree);
  spzo_converter_t c;
  spzo_coordinate_converter(from_coord, 4 /* RUB */, &c);

  /* header :131-139, :534-539 */
  put_u32(stream + 0, SPZO_MAGIC);
  put_u32(stream + 4, (uint32_t)version);
  put_u32(stream + 8, (uint32_t)numPoints);
  stream[12] = (uint8_t)shDegree;
  stream[13] = 12; /* fractionalBits :270 */
  stream[14] = antialiased ? SPZO_FLAG_ANTIALIASED : 0;
  stream[15] = 0;

  const size_t rotB = (version >= 3) ? 4 : 3;
  uint8_t *p_pos = stream + 16;
  uint8_t *p_alpha = p_pos + N * 9;
  uint8_t *p_color = p_alpha + N;
  uint8_t *p_scale = p_color + N * 3;
  uint8_t *p_rot = p_scale + N * 3;
  uint8_t *p_sh = p_rot + N * rotB;

  /* positions :280-288 */
  const float scale = (1 << 12);
  for (size_t i = 0; i < N * 3; i++) {
    const int32_t fixed32 = (int32_t)roundf(c.flipP[i % 3] * positions[i] * scale);
    p_pos[i * 3 + 0] = fixed32 & 0xff;
    p_pos[i * 3 + 1] = (fixed32 >> 8) & 0xff;
    p_pos[i * 3 + 2] = (fixed32 >> 16) & 0xff;
  }
  /* scales :290-292 */
  for (size_t i = 0; i < N * 3; i++) {
    p_scale[i] = spzo_to_uint8((scales[i] + 10.0f) * 16.0f);
  }
  /* rotations :294-297 */
  if (version >= 3) {
    for (size_t i = 0; i < N; i++) {
      spzo_pack_quat_smallest_three(&p_rot[4 * i], &rotations[4 * i], &c);
    }
  } else {
    /* PARITY UNPINNED: no v2 encoder exists in /root/reference.  Restates the
     * published upstream nianticlabs/spz v1.x first-three encoder: normalise,
     * flip xyz, scale by +-127.5 so that w >= 0, offset by 127.5, toUint8. */
    for (size_t i = 0; i < N; i++) {
      const float *r = &rotations[4 * i];
      float norm = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
      float q[4] = {r[0] / norm, r[1] / norm, r[2] / norm, r[3] / norm};
      q[0] *= c.flipQ[0];
      q[1] *= c.flipQ[1];
      q[2] *= c.flipQ[2];
      float s = (q[3] < 0) ? -127.5f : 127.5f;
      for (int k = 0; k < 3; k++) {
        p_rot[3 * i + k] = spzo_to_uint8(q[k] * s + 127.5f);
      }
    }
  }
  /* alphas :299-302 */
  for (size_t i = 0; i < N; i++) {
    p_alpha[i] = spzo_to_uint8(spzo_sigmoid(alphas[i]) * 255.0f);
  }
  /* colours :304-307 */
  for (size_t i = 0; i < N * 3; i++) {
    p_color[i] = spzo_to_uint8(colors[i] * (kColorScale * 255.0f) + (0.5f * 255.0f));
  }
  /* spherical harmonics :309-328 */
  if (shDegree > 0) {
    const int sh1Bits = 5, shRestBits = 4;
    const size_t shPerPoint = (size_t)shDim * 3;
    for (size_t i = 0; i < N * shPerPoint; i += shPerPoint) {
      size_t j = 0, k = 0;
      for (; j < 9; j += 3, k++) {
        p_sh[i + j + 0] = spzo_quantize_sh(c.flipSh[k] * sh[i + j + 0], 1 << (8 - sh1Bits));
        p_sh[i + j + 1] = spzo_quantize_sh(c.flipSh[k] * sh[i + j + 1], 1 << (8 - sh1Bits));
        p_sh[i + j + 2] = spzo_quantize_sh(c.flipSh[k] * sh[i + j + 2], 1 << (8 - sh1Bits));
      }
      for (; j < shPerPoint; j += 3, k++) {
        p_sh[i + j + 0] = spzo_quantize_sh(c.flipSh[k] * sh[i + j + 0], 1 << (8 - shRestBits));
        p_sh[i + j + 1] = spzo_quantize_sh(c.flipSh[k] * sh[i + j + 1], 1 << (8 - shRestBits));
        p_sh[i + j + 2] = spzo_quantize_sh(c.flipSh[k] * sh[i + j + 2], 1 << (8 - shRestBits));
      }
    }
  }
  return (size_t)(p_sh + N * (size_t)shDim * 3 - stream);
}

/* ---- deserializePackedGaussians header checks load-spz.cc:548-568 ------------ */
int spzo_peek(const uint8_t *stream, size_t size, uint32_t *version, uint32_t *num_points,
              int *sh_degree, int *fractional_bits, int *antialiased) {
  if (size < 16 || get_u32(stream) != SPZO_MAGIC) return -1;
  uint32_t v = get_u32(stream + 4);
  if (v < 1 || v > 3) return -2;
  uint32_t n = get_u32(stream + 8);
  if (n > SPZO_MAX_POINTS) return -3;
  if (stream[12] > 3) return -4;
  if (size < spzo_stream_size(n, stream[12], (int)v)) return -5; /* :591-594 */
  if (version) *version = v;
  if (num_points) *num_points = n;
  if (sh_degree) *sh_degree = stream[12];
  if (fractional_bits) *fractional_bits = stream[13];
  if (antialiased) *antialiased = (stream[14] & SPZO_FLAG_ANTIALIASED) != 0;
  return 0;
}

/* ---- GaussianCloud::convertCoordinates splat-types.h:134-164 ------------------ */
void spzo_convert_coordinates(float *positions, float *rotations, float *sh,
                              int32_t numPoints, int shDim, int from_coord, int to_coord) {
  if (numPoints == 0) return;
  const size_t N = (size_t)numPoints;
  spzo_converter_t c;
  spzo_coordinate_converter(from_coord, to_coord, &c);
  for (size_t i = 0; i < N * 3; i += 3) {
    positions[i + 0] *= c.flipP[0];
    positions[i + 1] *= c.flipP[1];
    positions[i + 2] *= c.flipP[2];
  }
  for (size_t i = 0; i < N * 4; i += 4) {
    rotations[i + 0] *= c.flipQ[0];
    rotations[i + 1] *= c.flipQ[1];
    rotations[i + 2] *= c.flipQ[2];
  }
  size_t idx = 0;
  for (size_t i = 0; i < N; i++) {
    for (int j = 0; j < shDim; ++j, idx += 3) {
      float flip = c.flipSh[j];
      sh[idx + 0] *= flip;
      sh[idx + 1] *= flip;
      sh[idx + 2] *= flip;
    }
  }
}

/* ---- unpackGaussians load-spz.cc:467-531 -------------------------------------- */
int spzo_unpack(const uint8_t *stream, size_t size, int to_coord, float *positions,
                float *scales, float *rotations, float *alphas, float *colors, float *sh) {
  uint32_t version, n;
  int shDegree, fractionalBits, aa;
  int rc = spzo_peek(stream, size, &version, &n, &shDegree, &fractionalBits, &aa);
  if (rc != 0) return rc;
  const size_t N = n;
  const int shDim = spzo_dim_for_degree(shDegree);
  const int usesFloat16 = version == 1;
  const int smallestThree = version >= 3;
  const size_t posB = usesFloat16 ? 6 : 9;
  const size_t rotB = smallestThree ? 4 : 3;
  const uint8_t *p_pos = stream + 16;
  const uint8_t *p_alpha = p_pos + N * posB;
  const uint8_t *p_color = p_alpha + N;
  const uint8_t *p_scale = p_color + N * 3;
  const uint8_t *p_rot = p_scale + N * 3;
  const uint8_t *p_sh = p_rot + N * rotB;
  spzo_converter_t ident;
  spzo_coordinate_converter(0, 0, &ident); /* default CoordinateConverter: all +1 */

  if (usesFloat16) { /* :487-492 */
    for (size_t i = 0; i < N * 3; i++) {
      uint16_t h = (uint16_t)(p_pos[2 * i] | (p_pos[2 * i + 1] << 8));
      positions[i] = spzo_half_to_float(h);
    }
  } else { /* :493-503 */
    /* x86 `shl` masks the count to 5 bits; (1 << fractionalBits) is an int */
    float scale = 1.0 / (1 << (fractionalBits & 31));
    for (size_t i = 0; i < N * 3; i++) {
      int32_t fixed32 = p_pos[i * 3 + 0];
      fixed32 |= p_pos[i * 3 + 1] << 8;
      fixed32 |= p_pos[i * 3 + 2] << 16;
      fixed32 |= (fixed32 & 0x800000) ? 0xff000000 : 0;
      positions[i] = (float)fixed32 * scale;
    }
  }
  for (size_t i = 0; i < N * 3; i++) { /* :505-507 */
    scales[i] = p_scale[i] / 16.0f - 10.0f;
  }
  for (size_t i = 0; i < N; i++) { /* :509-515 */
    if (smallestThree) {
      spzo_unpack_quat_smallest_three(&rotations[4 * i], &p_rot[4 * i], &ident);
    } else {
      spzo_unpack_quat_first_three(&rotations[4 * i], &p_rot[3 * i], &ident);
    }
  }
  for (size_t i = 0; i < N; i++) { /* :517-519 */
    alphas[i] = spzo_inv_sigmoid(p_alpha[i] / 255.0f);
  }
  for (size_t i = 0; i < N * 3; i++) { /* :521-523 */
    colors[i] = ((p_color[i] / 255.0f) - 0.5f) / kColorScale;
  }
  for (size_t i = 0; i < N * (size_t)shDim * 3; i++) { /* :525-527 */
    sh[i] = spzo_unquantize_sh(p_sh[i]);
  }
  /* :529 result.convertCoordinates(RUB, o.to) */
  spzo_convert_coordinates(positions, rotations, sh, (int32_t)n, shDim, 4 /* RUB */, to_coord);
  return 0;
}

/* ---- loadSplatFromPly value loop, load-spz.cc:814-839, then :842 ------------------------------ */
void spzo_ply_rows_to_cloud(const float *values, int32_t numPoints, const spzo_ply_columns_t *cols, int to_coord,
                            float *positions, float *scales, float *rotations, float *alphas, float *colors,
                            float *sh) {
  const size_t fields = (size_t)cols->stride;
  const int shDim = cols->sh_dim;
  size_t o3 = 0, o4 = 0, o1 = 0, osh = 0;
  for (size_t p = 0; p < (size_t)numPoints; p++) {
    const size_t i = p * fields;
    for (int j = 0; j < 3; j++) positions[o3 + j] = values[i + cols->position[j]];
    for (int j = 0; j < 3; j++) scales[o3 + j] = values[i + cols->scale[j]];
    for (int j = 0; j < 4; j++) rotations[o4 + j] = values[i + cols->rotation[j]];
    alphas[o1] = values[i + cols->alpha];
    for (int j = 0; j < 3; j++) colors[o3 + j] = values[i + cols->color[j]];
    /* [N,C,S] -> [N,S,C] */
    for (int j = 0; j < shDim; j++) {
      sh[osh++] = values[i + cols->sh[j]];
      sh[osh++] = values[i + cols->sh[j + shDim]];
      sh[osh++] = values[i + cols->sh[j + 2 * shDim]];
    }
    o3 += 3;
    o4 += 4;
    o1 += 1;
  }
  spzo_convert_coordinates(positions, rotations, sh, numPoints, shDim, 6 /* RDF */, to_coord);
}

/* ---- saveSplatToPly row assembly, load-spz.cc:846-893 ------------------------------------------ */
void spzo_cloud_to_ply_rows(const float *positions, const float *scales, const float *rotations,
                            const float *alphas, const float *colors, const float *sh, int32_t N, int shDim,
                            int from_coord, float *values) {
  const int D = 17 + shDim * 3;
  spzo_converter_t c;
  spzo_coordinate_converter(from_coord, 6 /* RDF */, &c);
  size_t outIdx = 0, i3 = 0, i4 = 0;
  for (int32_t i = 0; i < N; i++) {
    values[outIdx++] = c.flipP[0] * positions[i3 + 0];
    values[outIdx++] = c.flipP[1] * positions[i3 + 1];
    values[outIdx++] = c.flipP[2] * positions[i3 + 2];
    values[outIdx++] = 0.0f; /* normals */
    values[outIdx++] = 0.0f;
    values[outIdx++] = 0.0f;
    values[outIdx++] = colors[i3 + 0];
    values[outIdx++] = colors[i3 + 1];
    values[outIdx++] = colors[i3 + 2];
    for (int j = 0; j < shDim; j++) values[outIdx++] = c.flipSh[j] * sh[((size_t)i * shDim + j) * 3];
    for (int j = 0; j < shDim; j++) values[outIdx++] = c.flipSh[j] * sh[((size_t)i * shDim + j) * 3 + 1];
    for (int j = 0; j < shDim; j++) values[outIdx++] = c.flipSh[j] * sh[((size_t)i * shDim + j) * 3 + 2];
    values[outIdx++] = alphas[i];
    values[outIdx++] = scales[i3 + 0];
    values[outIdx++] = scales[i3 + 1];
    values[outIdx++] = scales[i3 + 2];
    values[outIdx++] = rotations[i4 + 3];
    values[outIdx++] = c.flipQ[0] * rotations[i4 + 0];
    values[outIdx++] = c.flipQ[1] * rotations[i4 + 1];
    values[outIdx++] = c.flipQ[2] * rotations[i4 + 2];
    i3 += 3;
    i4 += 4;
  }
  (void)D;
}

/* ---- GaussianCloud::medianVolume, splat-types.h:170-185 ------------------------------------------
 * sums[i] = (s[3i] + s[3i+1]) + s[3i+2] in f32; sorted ascending; the element of rank n/2 (the upper
 * median for even n); volume = (double)(pi*4/3) * (double)expf(median), narrowed to f32 on return. */
static int cmp_float(const void *a, const void *b) {
  const float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}

float spzo_median_volume(const float *scales, int32_t n) {
  if (n <= 0) return 0.01f;
  float *sums = (float *)malloc((size_t)n * sizeof(float));
  if (!sums) return 0.01f;
  for (int32_t i = 0; i < n; ++i) {
    const float a = scales[3 * (size_t)i], b = scales[3 * (size_t)i + 1], c = scales[3 * (size_t)i + 2];
    sums[i] = (a + b) + c;
  }
  qsort(sums, (size_t)n, sizeof(float), cmp_float);
  const float median = sums[n / 2];
  free(sums);
  return (float)((M_PI * 4 / 3) * (double)expf(median));
}
