// spz_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4) for the SPZ
// per-Gaussian quantise / dequantise hot path.  The C ABI that launches them is
// spz_abi.hip; what the two share is spz_kernel_params.hpp.
//
// What this replaces in the reference (/root/reference/src/cc):
//   encode  = packGaussians (load-spz.cc:257-331) + packQuaternionSmallestThree
//             (:216-255) + serializePackedGaussians (:533-546)
//   decode  = section slicing of deserializePackedGaussians (:569-590) +
//             unpackGaussians (:467-531) + unpackQuaternion{FirstThree,SmallestThree}
//             (:333-381) + GaussianCloud::convertCoordinates (splat-types.h:134-164)
//             fused into one pass
//   flip    = GaussianCloud::convertCoordinates as a standalone in-place pass.
//
// Design: this is an HBM-bound byte transform, no MFMA.  ONE
// launch per direction covers all six sections: the grid is a flat list of
// tiles (256 threads x UNROLL "units", one tile per block), a block finds its section with a few
// wave-uniform scalar compares and runs that section's body.  A unit is four
// consecutive float elements (one 16-byte access on the float side) and the
// 4 / 12 / 8 / 3 packed bytes they map to, so both sides of every section are
// accessed lane-contiguously: 1 KiB per wave-instruction on the float side,
// 256 B (768 B for 24-bit positions) on the byte side.  Section bases in the
// stream are 16+9N, 16+10N, ... and therefore land on any byte alignment; the
// byte side uses unaligned dword / dwordx3 accesses (global memory on gfx950
// runs in unaligned-access mode), so there is no head/tail peel and no
// per-section alignment case split.  Coordinate flips are sign-bit XORs picked
// from small bit masks; every table-driven decode (alpha, colour) reads a 256
// entry LDS table; the alpha encode (the only transcendental on the path) is an
// 8-step branch-free binary search over 255 float thresholds in LDS.
//
// Bit-exactness rules: no FMA contraction anywhere on
// the quantisation path (file-wide contract(off) + register barriers after the
// multiplies that feed adds), round-half-away implemented exactly, IEEE divide
// and sqrt (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), f32
// denormals preserved (hipcc default), float->int conversions reproduce the
// x86-64 cvttss2si results of the reference build also outside the defined
// domain (NaN / overflow -> "integer indefinite").
#include <hip/hip_runtime.h>

#include <cstdint>

#include "spz_amd.h"
#include "spz_common.hpp"
#include "spz_kernel_params.hpp"

#pragma clang fp contract(off)

namespace spz_amd_detail {
namespace {

// ------------------------------------------------------------------------------------------
// Unaligned access types.  The float side needs dword alignment only; the byte side none.
// ------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(1))) U32x1 { uint32_t a; };
struct __attribute__((packed, aligned(1))) U32x2 { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) U32x3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(1))) U8x3 { uint8_t a, b, c; };

struct Raw3 { uint32_t a, b, c; };

typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef v4f_t v4f_a4 __attribute__((aligned(4)));
typedef uint32_t u32_a1 __attribute__((aligned(1)));

template <bool NT>
__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32_a1 *>(p));
  else return *reinterpret_cast<const u32_a1 *>(p);
}
template <bool NT>
__device__ __forceinline__ void st_u32(uint8_t *p, uint32_t v) {
  if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32_a1 *>(p));
  else *reinterpret_cast<u32_a1 *>(p) = v;
}

// ------------------------------------------------------------------------------------------
// Exact scalar helpers
// ------------------------------------------------------------------------------------------
// Keeps a product in a register of its own so that no later add can be contracted with it,
// whatever the compiler flags are.
__device__ __forceinline__ float fmul_sep(float a, float b) {
  float r = a * b;
  asm volatile("" : "+v"(r));
  return r;
}
__device__ __forceinline__ float fadd_sep(float a, float b) {
  float r = a + b;
  asm volatile("" : "+v"(r));
  return r;
}

// Same product for operands that cannot be NaN (or whose NaN sign cannot reach the output): a
// plain sign-bit flip.
__device__ __forceinline__ float xor_sign(float v, uint32_t neg) {
  return __uint_as_float(__float_as_uint(v) ^ (neg << 31));
}

// std::round: half away from zero, exact (load-spz.cc:74,78,284).
__device__ __forceinline__ float round_half_away(float x) {
  float t = __builtin_truncf(x);
  float d = __builtin_fabsf(x - t);  // exact
  float one = __builtin_copysignf(1.0f, x);
  return (d >= 0.5f) ? (t + one) : t;
}

// static_cast<int32_t>(float) as the reference's x86-64 build executes it (cvttss2si):
// NaN and out-of-range give 0x80000000.
__device__ __forceinline__ int32_t cvt_i32_x86(float r) {
  return (r >= -2147483648.0f && r < 2147483648.0f) ? (int32_t)r : (int32_t)0x80000000;
}

// static_cast<uint32_t>(float) as x86-64 gcc executes it: 64-bit cvttss2si, low 32 bits.
__device__ __forceinline__ uint32_t cvt_u32_x86(float r) {
  if (r > -9223372036854775808.0f && r < 9223372036854775808.0f) {
    return (uint32_t)(long long)r;
  }
  return 0u;
}

// toUint8 (load-spz.cc:74): static_cast<uint8_t>(clamp(round(x), 0, 255)), as a float in
// [0, 255].  std::clamp lets a NaN through and the x86 cast turns it into 0; fmaxf(NaN, 0) = 0
// gives the same.
__device__ __forceinline__ float to_uint8_f(float x) {
  float r = round_half_away(x);
  return __builtin_fminf(__builtin_fmaxf(r, 0.0f), 255.0f);
}
__device__ __forceinline__ uint32_t to_uint8(float x) { return (uint32_t)to_uint8_f(x); }

// quantizeSH (load-spz.cc:77-81) for bucket b in {8, 16}, as a float in [0, 255]:
//   q = (int)(round(128 x) + 128);  q = (q + b/2) / b * b;  clamp(q, 0, 255).
// With r = round(128 x) (an integer-valued float) and 128 a multiple of b,
//   (q + b/2) / b * b  ==  b * floor((r + b/2) / b) + 128   wherever q + b/2 >= 0,
// and every negative q + b/2 (C division truncates toward zero) ends <= 0 and clamps to 0, as
// does the floor form.  All products are by powers of two; the sums are exact below 2^24 and
// far outside the clamp range above it.  The reference's (int) cast is "integer indefinite"
// (INT_MIN -> clamps to 0) for NaN and once r + 128.0f reaches 2^31, i.e. from r = 2^31 - 128 up.
__device__ __forceinline__ float quantize_sh_f(float x, bool degree1) {
  const float inv_b = degree1 ? 0.125f : 0.0625f;
  const float b = degree1 ? 8.0f : 16.0f;
  float r = round_half_away(x * 128.0f);
  // exact products: a fused multiply-add rounds exactly like the separate operations here
  float k = __builtin_floorf(__builtin_fmaf(r, inv_b, 0.5f));
  float v = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(k, b, 128.0f), 0.0f), 255.0f);
  return (r < 2147483520.0f) ? v : 0.0f;
}

// Four values already in [0, 255] and integral -> one little-endian dword.
__device__ __forceinline__ uint32_t pack_u8x4(float a, float b, float c, float d) {
  uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(a, 0u, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(b, 1u, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(c, 2u, w);
  return __builtin_amdgcn_cvt_pk_u8_f32(d, 3u, w);
}

constexpr float kSqrt1_2 = (float)0.707106781186547524401;  // load-spz.cc:46

// ------------------------------------------------------------------------------------------
// Correctly rounded divisions without the IEEE expansion's operand scaling.
//
// hipcc expands an f32 `a / b` into v_div_scale x2, v_rcp, five fma/mul, v_div_fmas, v_div_fixup
// (11 VALU operations); a quaternion costs seven of them.  Inside the exponent window where
// v_div_scale does not scale and v_div_fixup passes its operand through, the same arithmetic is
// (a) for the four quotients x_i / norm: ONE refined reciprocal shared by all four, then the
//     expansion's own two residual corrections (quat_quotient);
// (b) for the divisions by the constants 0.70710677f and 511.0f: reciprocal multiply + one fma
//     residual correction (div_by_const), which equals the IEEE quotient for every dividend that
//     is zero or in [2^-100, 2^126] — checked over ALL such floats by spz_selftest_kernel
//     (tests/test_gpu_parity.py::test_fast_divisions_exhaustive), as are (a) on 2^31 operand pairs
//     and sqrt_cr on every float of its window.
// Operands outside the window (quat_fast_ok) take the general forms below, unchanged from the
// reference-shaped arithmetic; both forms give identical bits wherever the fast one is used.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kFastLoBits = 0x2b800000u;  // 2^-40
constexpr uint32_t kFastHiBits = 0x53800000u;  // 2^40

// Every component is zero or has 2^-40 <= |x| <= 2^40, and at least one is not zero: then no square
// under/overflows, norm is in [2^-40, 2^41], every quotient is zero or >= 2^-81 and every residual
// of the corrections below is exactly representable.
__device__ __forceinline__ bool quat_fast_ok(F32x4 r) {
  const uint32_t a0 = __float_as_uint(r.x) & 0x7fffffffu, a1 = __float_as_uint(r.y) & 0x7fffffffu;
  const uint32_t a2 = __float_as_uint(r.z) & 0x7fffffffu, a3 = __float_as_uint(r.w) & 0x7fffffffu;
  // a - 1 wraps a zero to 0xffffffff, so zeros pass the lower bound
  const uint32_t lo = min(min(a0 - 1u, a1 - 1u), min(a2 - 1u, a3 - 1u));
  const uint32_t hi = max(max(a0, a1), max(a2, a3));
  return lo >= kFastLoBits - 1u && hi <= kFastHiBits && hi != 0u;
}

// Correctly rounded sqrt for 2^-80 <= x <= 2^82 (no denormal scaling, no zero / inf fix-up): v_sqrt_f32
// is within one ulp, the two neighbours are tried with exact residuals (the IEEE expansion's own step).
__device__ __forceinline__ float sqrt_cr(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float dn = __uint_as_float(__float_as_uint(s) - 1u);
  const float up = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_dn = __builtin_fmaf(-dn, s, x);
  const float r_up = __builtin_fmaf(-up, s, x);
  float o = (r_dn <= 0.0f) ? dn : s;
  o = (r_up > 0.0f) ? up : o;
  return o;
}

// 1 / b refined once (v_rcp_f32 + one Newton step): the reciprocal the IEEE expansion uses.
__device__ __forceinline__ float refined_rcp(float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y0, 1.0f);
  return __builtin_fmaf(e, y0, y0);
}

// a / b for a >= 0 (zero or >= 2^-100), b > 0, quotient zero or normal; y = refined_rcp(b).
__device__ __forceinline__ float quat_quotient(float a, float b, float y) {
  const float m = a * y;
  const float r0 = __builtin_fmaf(-b, m, a);
  const float q1 = __builtin_fmaf(r0, y, m);
  const float r1 = __builtin_fmaf(-b, q1, a);
  return __builtin_fmaf(r1, y, q1);
}

// x / c for a constant c with rc = RN(1 / c); x zero or in [2^-100, 2^126].
__device__ __forceinline__ float div_by_const(float x, float c, float rc) {
  const float m = x * rc;
  const float rem = __builtin_fmaf(-m, c, x);
  return __builtin_fmaf(rem, rc, m);
}
constexpr float kRcpSqrt1_2 = 1.0f / kSqrt1_2;
constexpr float kRcp511 = 1.0f / 511.0f;

struct Quat4 { float q0, q1, q2, q3; };

// normalized() (splat-types.cc:71-74) followed by the xyz flip (load-spz.cc:224-227), general operands.
__device__ __forceinline__ Quat4 normalized_flipped(F32x4 r, uint32_t flip_q) {
  float n2 = fadd_sep(fadd_sep(fadd_sep(fmul_sep(r.x, r.x), fmul_sep(r.y, r.y)), fmul_sep(r.z, r.z)),
                      fmul_sep(r.w, r.w));
  float norm = __builtin_sqrtf(n2);
  Quat4 q;
  q.q0 = xor_sign(r.x / norm, flip_q & 1u);
  q.q1 = xor_sign(r.y / norm, (flip_q >> 1) & 1u);
  q.q2 = xor_sign(r.z / norm, (flip_q >> 2) & 1u);
  q.q3 = r.w / norm;
  return q;
}

// The same for operands inside the quat_fast_ok window: same bits, 4 + 9 + 20 operations instead of 4 + 15 + 44.
__device__ __forceinline__ Quat4 normalized_flipped_fast(F32x4 r, uint32_t flip_q) {
  float n2 = fadd_sep(fadd_sep(fadd_sep(fmul_sep(r.x, r.x), fmul_sep(r.y, r.y)), fmul_sep(r.z, r.z)),
                      fmul_sep(r.w, r.w));
  const float norm = sqrt_cr(n2);
  const float y = refined_rcp(norm);
  const uint32_t sx = (__float_as_uint(r.x) >> 31) ^ (flip_q & 1u), sy = (__float_as_uint(r.y) >> 31) ^ ((flip_q >> 1) & 1u);
  const uint32_t sz = (__float_as_uint(r.z) >> 31) ^ ((flip_q >> 2) & 1u), sw = __float_as_uint(r.w) >> 31;
  Quat4 q;
  q.q0 = xor_sign(quat_quotient(__builtin_fabsf(r.x), norm, y), sx);
  q.q1 = xor_sign(quat_quotient(__builtin_fabsf(r.y), norm, y), sy);
  q.q2 = xor_sign(quat_quotient(__builtin_fabsf(r.z), norm, y), sz);
  q.q3 = xor_sign(quat_quotient(__builtin_fabsf(r.w), norm, y), sw);
  return q;
}

// The bit-field assembly of packQuaternionSmallestThree (load-spz.cc:229-254), general operands.
__device__ __forceinline__ uint32_t smallest_three_fields(Quat4 q) {
  const float q0 = q.q0, q1 = q.q1, q2 = q.q2, q3 = q.q3;
  // argmax |q|, strict >, first wins
  uint32_t iL = 0;
  float best = __builtin_fabsf(q0);
  if (__builtin_fabsf(q1) > best) { iL = 1; best = __builtin_fabsf(q1); }
  if (__builtin_fabsf(q2) > best) { iL = 2; best = __builtin_fabsf(q2); }
  if (__builtin_fabsf(q3) > best) { iL = 3; best = __builtin_fabsf(q3); }
  float qL = (iL == 0) ? q0 : (iL == 1) ? q1 : (iL == 2) ? q2 : q3;
  uint32_t negate = (qL < 0.0f) ? 1u : 0u;
  uint32_t comp = iL;
  const float qs[4] = {q0, q1, q2, q3};
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    uint32_t negbit = ((qs[i] < 0.0f) ? 1u : 0u) ^ negate;
    float m = fmul_sep(511.0f, __builtin_fabsf(qs[i]) / kSqrt1_2) + 0.5f;
    uint32_t mag = cvt_u32_x86(m);
    uint32_t next = (comp << 10) | (negbit << 9) | mag;
    comp = (i != iL) ? next : comp;
  }
  return comp;
}

// The same for |q_i| zero or in [2^-100, 2]: only the three kept components are quantised, the
// divisions by sqrt(1/2) are reciprocal multiplies with a residual correction, and 511 t + 0.5 < 2^31
// converts with one instruction.
__device__ __forceinline__ uint32_t smallest_three_fields_fast(Quat4 q) {
  const float a0 = __builtin_fabsf(q.q0), a1 = __builtin_fabsf(q.q1), a2 = __builtin_fabsf(q.q2), a3 = __builtin_fabsf(q.q3);
  uint32_t iL = 0;
  float best = a0;
  if (a1 > best) { iL = 1; best = a1; }
  if (a2 > best) { iL = 2; best = a2; }
  if (a3 > best) { iL = 3; best = a3; }
  const float qL = (iL == 0) ? q.q0 : (iL == 1) ? q.q1 : (iL == 2) ? q.q2 : q.q3;
  const uint32_t negate = (qL < 0.0f) ? 1u : 0u;
  // the components other than iL, in index order
  const float c0 = (iL == 0) ? q.q1 : q.q0;
  const float c1 = (iL <= 1) ? q.q2 : q.q1;
  const float c2 = (iL <= 2) ? q.q3 : q.q2;
  uint32_t comp = iL;
  const float cs[3] = {c0, c1, c2};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const uint32_t negbit = ((cs[i] < 0.0f) ? 1u : 0u) ^ negate;
    const float t = div_by_const(__builtin_fabsf(cs[i]), kSqrt1_2, kRcpSqrt1_2);
    const float m = fmul_sep(511.0f, t) + 0.5f;
    comp = (comp << 10) | (negbit << 9) | (uint32_t)m;
  }
  return comp;
}

// packQuaternionSmallestThree (load-spz.cc:216-255) incl. normalized() (splat-types.cc:71-74).
__device__ __forceinline__ uint32_t pack_quat_smallest_three(F32x4 r, uint32_t flip_q) {
#if SPZ_QUAT_FAST
  if (quat_fast_ok(r)) return smallest_three_fields_fast(normalized_flipped_fast(r, flip_q));
#endif
  return smallest_three_fields(normalized_flipped(r, flip_q));
}

// PARITY UNPINNED (no v2 encoder in the reference): upstream nianticlabs/spz v1.x
// first-three encoder — normalise, flip, scale by +-127.5 so that w >= 0, offset, toUint8.
__device__ __forceinline__ uint32_t pack_quat_first_three(F32x4 r, uint32_t flip_q) {
  Quat4 q;
#if SPZ_QUAT_FAST
  if (quat_fast_ok(r)) q = normalized_flipped_fast(r, flip_q);
  else
#endif
    q = normalized_flipped(r, flip_q);
  float s = (q.q3 < 0.0f) ? -127.5f : 127.5f;
  uint32_t b0 = to_uint8(fmul_sep(q.q0, s) + 127.5f);
  uint32_t b1 = to_uint8(fmul_sep(q.q1, s) + 127.5f);
  uint32_t b2 = to_uint8(fmul_sep(q.q2, s) + 127.5f);
  return b0 | (b1 << 8) | (b2 << 16);
}

// unpackQuaternionSmallestThree (load-spz.cc:347-381) followed by the flip pass.
template <bool FAST>
__device__ __forceinline__ F32x4 unpack_quat_smallest_three_t(uint32_t comp, uint32_t flip_q) {
  const uint32_t iL = comp >> 30;
  float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float sum = 0.0f;
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    const bool take = ((uint32_t)i != iL);
    uint32_t mag = comp & 511u;
    uint32_t neg = (comp >> 9) & 1u;
    // (sqrt1_2 * mag) / 511.f: the dividend is zero or in [0.707, 361.4] for the 512 magnitudes
    float c = fmul_sep(kSqrt1_2, (float)mag);
    if constexpr (FAST) c = div_by_const(c, 511.0f, kRcp511);
    else c = c / 511.0f;
    c = __uint_as_float(__float_as_uint(c) ^ (neg << 31));
    float s2 = sum + fmul_sep(c, c);
    v[i] = take ? c : 0.0f;
    sum = take ? s2 : sum;
    comp = take ? (comp >> 10) : comp;
  }
  // sqrt(1.0f - sum): ::sqrt(double) rounded to float == correctly rounded sqrtf; a negative
  // argument yields the x86 default NaN (sign bit set).
  float d = 1.0f - sum;
  float big = (d < 0.0f) ? __uint_as_float(0xffc00000u) : __builtin_sqrtf(d);
  float x = (iL == 0) ? big : v[0];
  float y = (iL == 1) ? big : v[1];
  float z = (iL == 2) ? big : v[2];
  float w = (iL == 3) ? big : v[3];
  F32x4 o;
  o.x = mul_pm1(x, flip_q & 1u);
  o.y = mul_pm1(y, (flip_q >> 1) & 1u);
  o.z = mul_pm1(z, (flip_q >> 2) & 1u);
  o.w = w;
  return o;
}
__device__ __forceinline__ F32x4 unpack_quat_smallest_three(uint32_t comp, uint32_t flip_q) {
  return unpack_quat_smallest_three_t<SPZ_QUAT_FAST != 0>(comp, flip_q);
}

// unpackQuaternionFirstThree (load-spz.cc:333-345) followed by the flip pass.
__device__ __forceinline__ F32x4 unpack_quat_first_three(uint32_t r3, uint32_t flip_q) {
  constexpr float k = 1.0f / 127.5f;
  float x = fmul_sep((float)(r3 & 0xffu), k) + (-1.0f);
  float y = fmul_sep((float)((r3 >> 8) & 0xffu), k) + (-1.0f);
  float z = fmul_sep((float)((r3 >> 16) & 0xffu), k) + (-1.0f);
  float sq = fadd_sep(fadd_sep(fmul_sep(x, x), fmul_sep(y, y)), fmul_sep(z, z));
  float d = 1.0f - sq;
  float m = (0.0f < d) ? d : 0.0f;  // std::max(0.0f, d)
  F32x4 o;
  o.x = xor_sign(x, flip_q & 1u);
  o.y = xor_sign(y, (flip_q >> 1) & 1u);
  o.z = xor_sign(z, (flip_q >> 2) & 1u);
  o.w = __builtin_sqrtf(m);
  return o;
}

// halfToFloat (splat-types.cc:8-27): exact widening; every NaN becomes the positive quiet NaN.
__device__ __forceinline__ float half_to_float(uint32_t h) {
  h &= 0xffffu;
  if ((h & 0x7c00u) == 0x7c00u && (h & 0x3ffu) != 0) return __uint_as_float(0x7fc00000u);
  _Float16 f;
  uint16_t hh = (uint16_t)h;
  __builtin_memcpy(&f, &hh, 2);
  return (float)f;
}

// Alpha encode: number of thresholds <= a (NaN -> 0), thresholds ascending, thr[255] = NaN.
__device__ __forceinline__ uint32_t alpha_byte(float a, const float *thr) {
  uint32_t lo = 0;
#pragma unroll
  for (uint32_t step = 128; step >= 1; step >>= 1) {
    lo += (a >= thr[lo + step - 1]) ? step : 0u;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------
// Per-unit sign-bit selectors
// ------------------------------------------------------------------------------------------
// positions: element e = 4u + c sits on axis e % 3 = (u + c) % 3.
__device__ __forceinline__ uint32_t pos_sign_bits(uint32_t flip_p, uint32_t unit_mod3) {
  uint32_t rep = flip_p | (flip_p << 3);  // bits 0..5
  return (rep >> unit_mod3) & 0xfu;
}

// ------------------------------------------------------------------------------------------
// Decode
// ------------------------------------------------------------------------------------------
template <int KIND, class G>
__device__ __forceinline__ Raw3 load_raw(const uint8_t *__restrict__ src, unsigned long long u) {
  Raw3 r = {0u, 0u, 0u};
  if constexpr (KIND == KIND_POS24) {
    if constexpr (G::kNtLoad) {
      const uint8_t *q = src + u * 12ull;
      r.a = ld_u32<true>(q); r.b = ld_u32<true>(q + 4); r.c = ld_u32<true>(q + 8);
    } else {
      U32x3 t = *reinterpret_cast<const U32x3 *>(src + u * 12ull);
      r.a = t.a; r.b = t.b; r.c = t.c;
    }
  } else if constexpr (KIND == KIND_POS16) {
    U32x2 t = *reinterpret_cast<const U32x2 *>(src + u * 8ull);
    r.a = t.a; r.b = t.b;
  } else if constexpr (KIND == KIND_ROT_F3) {
    U8x3 t = *reinterpret_cast<const U8x3 *>(src + u * 3ull);
    r.a = (uint32_t)t.a | ((uint32_t)t.b << 8) | ((uint32_t)t.c << 16);
  } else {
    r.a = ld_u32<G::kNtLoad>(src + u * 4ull);
  }
  return r;
}

// Partial last unit: only `m` (1..3) elements exist; read exactly their bytes.
template <int KIND>
__device__ __forceinline__ Raw3 load_raw_partial(const uint8_t *__restrict__ src, unsigned long long u, uint32_t m) {
  constexpr uint32_t bpe = (KIND == KIND_POS24) ? 3u : (KIND == KIND_POS16) ? 2u : 1u;  // bytes per element
  const uint8_t *p = src + u * (4ull * bpe);
  uint32_t w[3] = {0u, 0u, 0u};
  for (uint32_t i = 0; i < m * bpe; ++i) w[i >> 2] |= (uint32_t)p[i] << ((i & 3u) * 8u);
  return Raw3{w[0], w[1], w[2]};
}

struct DecodeCtx {
  uint32_t flip_p, flip_q;
  unsigned long long sh_mask_ext;
  float pos_scale;
  const float *lut;  // LDS
};

template <int KIND, int D>
__device__ __forceinline__ F32x4 decode_unit(Raw3 raw, uint32_t phase, const DecodeCtx &c) {
  F32x4 o;
  if constexpr (KIND == KIND_POS24) {
    // load-spz.cc:496-502: sign-extended 24-bit * scale, then flipP
    uint32_t v0 = raw.a & 0xffffffu;
    uint32_t v1 = (raw.a >> 24) | ((raw.b & 0xffffu) << 8);
    uint32_t v2 = (raw.b >> 16) | ((raw.c & 0xffu) << 16);
    uint32_t v3 = raw.c >> 8;
    uint32_t sb = pos_sign_bits(c.flip_p, phase);
    o.x = xor_sign((float)((int32_t)(v0 << 8) >> 8) * c.pos_scale, sb & 1u);
    o.y = xor_sign((float)((int32_t)(v1 << 8) >> 8) * c.pos_scale, (sb >> 1) & 1u);
    o.z = xor_sign((float)((int32_t)(v2 << 8) >> 8) * c.pos_scale, (sb >> 2) & 1u);
    o.w = xor_sign((float)((int32_t)(v3 << 8) >> 8) * c.pos_scale, (sb >> 3) & 1u);
  } else if constexpr (KIND == KIND_POS16) {
    uint32_t sb = pos_sign_bits(c.flip_p, phase);
    o.x = mul_pm1(half_to_float(raw.a), sb & 1u);
    o.y = mul_pm1(half_to_float(raw.a >> 16), (sb >> 1) & 1u);
    o.z = mul_pm1(half_to_float(raw.b), (sb >> 2) & 1u);
    o.w = mul_pm1(half_to_float(raw.b >> 16), (sb >> 3) & 1u);
  } else if constexpr (KIND == KIND_ALPHA) {
    const float *t = c.lut + kTableAlphaDec;  // invSigmoid(b / 255.0f), load-spz.cc:518
    o.x = t[raw.a & 0xffu];
    o.y = t[(raw.a >> 8) & 0xffu];
    o.z = t[(raw.a >> 16) & 0xffu];
    o.w = t[raw.a >> 24];
  } else if constexpr (KIND == KIND_COLOR) {
    const float *t = c.lut + kTableColorDec;  // ((b / 255.0f) - 0.5f) / 0.15f, load-spz.cc:522
    o.x = t[raw.a & 0xffu];
    o.y = t[(raw.a >> 8) & 0xffu];
    o.z = t[(raw.a >> 16) & 0xffu];
    o.w = t[raw.a >> 24];
  } else if constexpr (KIND == KIND_SCALE) {
    // b / 16.0f - 10.0f, load-spz.cc:506 (the quotient is exact)
    o.x = (float)(raw.a & 0xffu) / 16.0f - 10.0f;
    o.y = (float)((raw.a >> 8) & 0xffu) / 16.0f - 10.0f;
    o.z = (float)((raw.a >> 16) & 0xffu) / 16.0f - 10.0f;
    o.w = (float)(raw.a >> 24) / 16.0f - 10.0f;
  } else if constexpr (KIND == KIND_ROT_S3) {
    o = unpack_quat_smallest_three(raw.a, c.flip_q);
  } else if constexpr (KIND == KIND_ROT_F3) {
    o = unpack_quat_first_three(raw.a, c.flip_q);
  } else {  // KIND_SH: (b - 128) / 128 (load-spz.cc:83), then flipSh of coefficient (e % D) / 3
    uint32_t sb = (uint32_t)(c.sh_mask_ext >> phase) & 0xfu;
    o.x = xor_sign(((float)(raw.a & 0xffu) - 128.0f) / 128.0f, sb & 1u);
    o.y = xor_sign(((float)((raw.a >> 8) & 0xffu) - 128.0f) / 128.0f, (sb >> 1) & 1u);
    o.z = xor_sign(((float)((raw.a >> 16) & 0xffu) - 128.0f) / 128.0f, (sb >> 2) & 1u);
    o.w = xor_sign(((float)(raw.a >> 24) - 128.0f) / 128.0f, (sb >> 3) & 1u);
  }
  return o;
}

// Phase of a unit inside its tile: positions -> unit % 3, sh -> (4 * unit) % D, else unused.
template <int KIND, int D, class G>
__device__ __forceinline__ uint32_t tile_phase_base(uint32_t tile_local) {
  if constexpr (KIND == KIND_POS24 || KIND == KIND_POS16 || KIND == KIND_FLIP_POS) {
    return ((tile_local % 3u) * ((uint32_t)G::kTileUnits % 3u)) % 3u;
  } else if constexpr ((KIND == KIND_SH || KIND == KIND_FLIP_SH) && D > 0) {
    return ((tile_local % (uint32_t)D) * ((4u * (uint32_t)G::kTileUnits) % (uint32_t)D)) % (uint32_t)D;
  } else {
    return 0u;
  }
}
template <int KIND, int D>
__device__ __forceinline__ uint32_t unit_phase(uint32_t base, uint32_t local) {
  if constexpr (KIND == KIND_POS24 || KIND == KIND_POS16 || KIND == KIND_FLIP_POS) {
    return (base + local) % 3u;
  } else if constexpr ((KIND == KIND_SH || KIND == KIND_FLIP_SH) && D > 0) {
    return (base + 4u * local) % (uint32_t)D;
  } else {
    return 0u;
  }
}

template <bool NT>
__device__ __forceinline__ void store_f4(float *__restrict__ dst, unsigned long long u, F32x4 v) {
  if constexpr (NT) {
    v4f_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f_a4 *>(dst + u * 4ull));
  } else {
    *reinterpret_cast<F32x4 *>(dst + u * 4ull) = v;
  }
}
template <bool NT>
__device__ __forceinline__ F32x4 load_f4(const float *__restrict__ src, unsigned long long u) {
  if constexpr (NT) {
    v4f_t t = __builtin_nontemporal_load(reinterpret_cast<const v4f_a4 *>(src + u * 4ull));
    F32x4 v;
    v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w;
    return v;
  } else {
    return *reinterpret_cast<const F32x4 *>(src + u * 4ull);
  }
}

template <int KIND, int D, class G>
__device__ __forceinline__ void decode_tile(const SecDesc &s, uint32_t tile_local, const DecodeCtx &c) {
  const unsigned long long base = (unsigned long long)tile_local * G::kTileUnits;
  const uint32_t tid = threadIdx.x;
  const uint8_t *__restrict__ src = s.bytes;
  float *__restrict__ dst = s.floats;
  const unsigned long long full_units = s.n_elems >> 2;
  const uint32_t pb = tile_phase_base<KIND, D, G>(tile_local);
  if (base + G::kTileUnits <= full_units) {
    Raw3 raw[G::kUnroll];
#pragma unroll
    for (int r = 0; r < G::kUnroll; ++r) raw[r] = load_raw<KIND, G>(src, base + G::local_unit(r, tid));
#pragma unroll
    for (int r = 0; r < G::kUnroll; ++r) {
      const uint32_t local = G::local_unit(r, tid);
      store_f4<G::kNtStore>(dst, base + local, decode_unit<KIND, D>(raw[r], unit_phase<KIND, D>(pb, local), c));
    }
  } else {
#pragma unroll
    for (int r = 0; r < G::kUnroll; ++r) {
      const uint32_t local = G::local_unit(r, tid);
      const unsigned long long u = base + local;
      if (u < full_units) {
        store_f4<G::kNtStore>(dst, u, decode_unit<KIND, D>(load_raw<KIND, G>(src, u), unit_phase<KIND, D>(pb, local), c));
      } else if (u < s.n_units) {
        if constexpr (KIND != KIND_ROT_S3 && KIND != KIND_ROT_F3) {
          const uint32_t m = (uint32_t)(s.n_elems - u * 4ull);  // 1..3 valid elements
          F32x4 v = decode_unit<KIND, D>(load_raw_partial<KIND>(src, u, m), unit_phase<KIND, D>(pb, local), c);
          float *p = dst + u * 4ull;
          p[0] = v.x;
          if (m > 1) p[1] = v.y;
          if (m > 2) p[2] = v.z;
        }
      }
    }
  }
}

template <int KIND, class G>
__device__ __forceinline__ void decode_tile_sh_dispatch(const SecDesc &s, uint32_t tile_local, const DecodeCtx &c,
                                                        uint32_t sh_d) {
  if (sh_d == 45u) decode_tile<KIND, 45, G>(s, tile_local, c);
  else if (sh_d == 24u) decode_tile<KIND, 24, G>(s, tile_local, c);
  else decode_tile<KIND, 9, G>(s, tile_local, c);
}

// First tile of a block.  Default: block b takes tile b (the dispatcher deals consecutive blocks
// round-robin over the 8 XCDs, so each XCD streams every 8th 16 KiB chunk).  SPZ_XCD_REMAP: each XCD
// gets one contiguous eighth of the tiles instead (measured: no gain, there is no inter-block reuse
// for an XCD-local L2 to capture; kept as a tuning switch).
__device__ __forceinline__ uint32_t first_tile(const KParams &p) {
#if SPZ_XCD_REMAP
  const uint32_t chunk = (p.total_tiles + 7u) / 8u;
  const uint32_t t = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
  return ((blockIdx.x >> 3) < chunk && t < p.total_tiles) ? t : 0xffffffffu;
#else
  (void)p;
  return blockIdx.x;
#endif
}

__device__ __forceinline__ uint32_t find_section(const KParams &p, uint32_t tile) {
  uint32_t si = 0;
#pragma unroll
  for (uint32_t k = 1; k < SPZ_AMD_NUM_SECTIONS; ++k) {
    if (k < p.n_sec && tile >= p.sec[k].tile_begin) si = k;
  }
  return si;
}

// Tile of the fused grid -> (section, tile within the section).
struct TileRef { uint32_t si, tl; };
__device__ __forceinline__ TileRef locate_tile(const KParams &p, uint32_t tile) {
  TileRef t;
  if (tile < p.il_tiles) {
    const uint32_t group = tile / (uint32_t)SPZ_IL_GROUP, within = tile % (uint32_t)SPZ_IL_GROUP;
    const uint32_t rep = group / kIlPeriod, k = group - rep * kIlPeriod;
    t.si = p.il_sec[k];
    t.tl = (rep * p.il_count[t.si] + p.il_rank[k]) * (uint32_t)SPZ_IL_GROUP + within;
  } else {
    t.si = find_section(p, tile);
    t.tl = tile - p.sec[t.si].tile_begin + p.sec[t.si].tile_skip;
  }
  return t;
}

template <class G>
__device__ __forceinline__ void stage_tables(float *lut, const float *__restrict__ tables) {
  for (uint32_t i = threadIdx.x; i < kTableFloats; i += G::kBlock) lut[i] = tables[i];
  __syncthreads();
}

}  // namespace

template <class G>
__device__ __forceinline__ void decode_one_tile(const KParams &p, const SecDesc &s, uint32_t tl, const DecodeCtx &c,
                                                float *lut, bool &lut_ready) {
  switch (s.kind) {
    case KIND_POS24: decode_tile<KIND_POS24, 0, G>(s, tl, c); break;
    case KIND_POS16: decode_tile<KIND_POS16, 0, G>(s, tl, c); break;
    case KIND_ALPHA:
    case KIND_COLOR:
      if (!lut_ready) {
        stage_tables<G>(lut, p.tables);
        lut_ready = true;
      }
      if (s.kind == KIND_ALPHA) decode_tile<KIND_ALPHA, 0, G>(s, tl, c);
      else decode_tile<KIND_COLOR, 0, G>(s, tl, c);
      break;
    case KIND_SCALE: decode_tile<KIND_SCALE, 0, G>(s, tl, c); break;
    case KIND_ROT_S3: decode_tile<KIND_ROT_S3, 0, G>(s, tl, c); break;
    case KIND_ROT_F3: decode_tile<KIND_ROT_F3, 0, G>(s, tl, c); break;
    case KIND_SH: decode_tile_sh_dispatch<KIND_SH, G>(s, tl, c, p.sh_d); break;
    default: break;
  }
}

__global__ __launch_bounds__(DecGeom::kBlock) void spz_decode_kernel(const KParams p) {
  __shared__ float lut[kTableFloats];
  bool lut_ready = false;
  DecodeCtx c;
  c.flip_p = p.flip_p;
  c.flip_q = p.flip_q;
  c.sh_mask_ext = p.sh_mask_ext;
  c.pos_scale = p.pos_scale;
  c.lut = lut;
  for (uint32_t t0 = first_tile(p); t0 < p.total_tiles; t0 = SPZ_XCD_REMAP ? 0xffffffffu : t0 + gridDim.x) {
    const uint32_t tile = SPZ_DEC_REVERSE ? p.total_tiles - 1u - t0 : t0;
    const TileRef ref = locate_tile(p, tile);
    const SecDesc &s = p.sec[ref.si];
    const uint32_t tl = ref.tl;
    if (SPZ_DEC_REVERSE && SPZ_DEC_NTL && t0 < p.plain_tiles) decode_one_tile<DecGeomHot>(p, s, tl, c, lut, lut_ready);
    else decode_one_tile<DecGeom>(p, s, tl, c, lut, lut_ready);
  }
}

namespace {

// ------------------------------------------------------------------------------------------
// Encode
// ------------------------------------------------------------------------------------------
struct EncodeCtx {
  uint32_t flip_p, flip_q;
  unsigned long long sh_mask_ext;
  const float *lut;  // LDS
};

// Returns the packed bytes of one unit in (a, b, c): 12 bytes for positions, 4 (or 3) otherwise.
template <int KIND, int D>
__device__ __forceinline__ Raw3 encode_unit(F32x4 v, uint32_t phase, const EncodeCtx &c) {
  Raw3 o = {0u, 0u, 0u};
  if constexpr (KIND == KIND_POS24) {
    // load-spz.cc:282-288: (int32)round(flipP * p * 4096), low 24 bits LE
    uint32_t sb = pos_sign_bits(c.flip_p, phase);
    uint32_t f0 = (uint32_t)cvt_i32_x86(round_half_away(xor_sign(v.x, sb & 1u) * 4096.0f)) & 0xffffffu;
    uint32_t f1 = (uint32_t)cvt_i32_x86(round_half_away(xor_sign(v.y, (sb >> 1) & 1u) * 4096.0f)) & 0xffffffu;
    uint32_t f2 = (uint32_t)cvt_i32_x86(round_half_away(xor_sign(v.z, (sb >> 2) & 1u) * 4096.0f)) & 0xffffffu;
    uint32_t f3 = (uint32_t)cvt_i32_x86(round_half_away(xor_sign(v.w, (sb >> 3) & 1u) * 4096.0f)) & 0xffffffu;
    o.a = f0 | (f1 << 24);
    o.b = (f1 >> 8) | (f2 << 16);
    o.c = (f2 >> 16) | (f3 << 8);
  } else if constexpr (KIND == KIND_ALPHA) {
    // toUint8(sigmoid(a) * 255) (load-spz.cc:85,301) == number of thresholds <= a
    const float *t = c.lut + kTableAlphaThr;
    o.a = alpha_byte(v.x, t) | (alpha_byte(v.y, t) << 8) | (alpha_byte(v.z, t) << 16) | (alpha_byte(v.w, t) << 24);
  } else if constexpr (KIND == KIND_COLOR) {
    // toUint8(c * (0.15f * 255.0f) + (0.5f * 255.0f)), load-spz.cc:306: mul and add round separately
    constexpr float k = 0.15f * 255.0f;
    constexpr float h = 0.5f * 255.0f;
    o.a = pack_u8x4(to_uint8_f(fmul_sep(v.x, k) + h), to_uint8_f(fmul_sep(v.y, k) + h),
                    to_uint8_f(fmul_sep(v.z, k) + h), to_uint8_f(fmul_sep(v.w, k) + h));
  } else if constexpr (KIND == KIND_SCALE) {
    // toUint8((s + 10.0f) * 16.0f), load-spz.cc:291
    o.a = pack_u8x4(to_uint8_f(fadd_sep(v.x, 10.0f) * 16.0f), to_uint8_f(fadd_sep(v.y, 10.0f) * 16.0f),
                    to_uint8_f(fadd_sep(v.z, 10.0f) * 16.0f), to_uint8_f(fadd_sep(v.w, 10.0f) * 16.0f));
  } else if constexpr (KIND == KIND_ROT_S3) {
    o.a = pack_quat_smallest_three(v, c.flip_q);
  } else if constexpr (KIND == KIND_ROT_F3) {
    o.a = pack_quat_first_three(v, c.flip_q);
  } else {  // KIND_SH, load-spz.cc:315-327: 5 bits for the 9 degree-1 values, 4 bits for the rest
    uint32_t sb = (uint32_t)(c.sh_mask_ext >> phase) & 0xfu;
    uint32_t j0 = phase, j1 = phase + 1, j2 = phase + 2, j3 = phase + 3;
    j1 -= (j1 >= (uint32_t)D) ? (uint32_t)D : 0u;
    j2 -= (j2 >= (uint32_t)D) ? (uint32_t)D : 0u;
    j3 -= (j3 >= (uint32_t)D) ? (uint32_t)D : 0u;
    o.a = pack_u8x4(quantize_sh_f(xor_sign(v.x, sb & 1u), j0 < 9u), quantize_sh_f(xor_sign(v.y, (sb >> 1) & 1u), j1 < 9u),
                    quantize_sh_f(xor_sign(v.z, (sb >> 2) & 1u), j2 < 9u), quantize_sh_f(xor_sign(v.w, (sb >> 3) & 1u), j3 < 9u));
  }
  return o;
}

template <int KIND, class G>
__device__ __forceinline__ void store_raw(uint8_t *__restrict__ dst, unsigned long long u, Raw3 r) {
  if constexpr (KIND == KIND_POS24) {
    if constexpr (G::kNtStore) {
      uint8_t *q = dst + u * 12ull;
      st_u32<true>(q, r.a); st_u32<true>(q + 4, r.b); st_u32<true>(q + 8, r.c);
    } else {
      U32x3 t;
      t.a = r.a; t.b = r.b; t.c = r.c;
      *reinterpret_cast<U32x3 *>(dst + u * 12ull) = t;
    }
  } else if constexpr (KIND == KIND_ROT_F3) {
    U8x3 t;
    t.a = (uint8_t)r.a; t.b = (uint8_t)(r.a >> 8); t.c = (uint8_t)(r.a >> 16);
    *reinterpret_cast<U8x3 *>(dst + u * 3ull) = t;
  } else {
    st_u32<G::kNtStore>(dst + u * 4ull, r.a);
  }
}

template <int KIND, int D, class G>
__device__ __forceinline__ void encode_tile(const SecDesc &s, uint32_t tile_local, const EncodeCtx &c) {
  const unsigned long long base = (unsigned long long)tile_local * G::kTileUnits;
  const uint32_t tid = threadIdx.x;
  const float *__restrict__ src = s.floats;
  uint8_t *__restrict__ dst = s.bytes;
  const unsigned long long full_units = s.n_elems >> 2;
  const uint32_t pb = tile_phase_base<KIND, D, G>(tile_local);
  if (base + G::kTileUnits <= full_units) {
    F32x4 v[G::kUnroll];
#pragma unroll
    for (int r = 0; r < G::kUnroll; ++r) v[r] = load_f4<G::kNtLoad>(src, base + G::local_unit(r, tid));
#pragma unroll
    for (int r = 0; r < G::kUnroll; ++r) {
      const uint32_t local = G::local_unit(r, tid);
      store_raw<KIND, G>(dst, base + local, encode_unit<KIND, D>(v[r], unit_phase<KIND, D>(pb, local), c));
#if SPZ_ENC_SCHED_BARRIER
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  } else {
#pragma unroll
    for (int r = 0; r < G::kUnroll; ++r) {
      const uint32_t local = G::local_unit(r, tid);
      const unsigned long long u = base + local;
      if (u < full_units) {
        store_raw<KIND, G>(dst, u, encode_unit<KIND, D>(load_f4<G::kNtLoad>(src, u), unit_phase<KIND, D>(pb, local), c));
      } else if (u < s.n_units) {
        if constexpr (KIND != KIND_ROT_S3 && KIND != KIND_ROT_F3) {
          const uint32_t m = (uint32_t)(s.n_elems - u * 4ull);  // 1..3 valid elements
          const float *p = src + u * 4ull;
          F32x4 v;
          v.x = p[0];
          v.y = (m > 1) ? p[1] : 0.0f;
          v.z = (m > 2) ? p[2] : 0.0f;
          v.w = 0.0f;
          Raw3 r3 = encode_unit<KIND, D>(v, unit_phase<KIND, D>(pb, local), c);
          constexpr uint32_t bpe = (KIND == KIND_POS24) ? 3u : 1u;
          uint8_t *q = dst + u * (4ull * bpe);
          const uint32_t w[3] = {r3.a, r3.b, r3.c};
          for (uint32_t i = 0; i < m * bpe; ++i) q[i] = (uint8_t)(w[i >> 2] >> ((i & 3u) * 8u));
        }
      }
    }
  }
}

template <int KIND>
__device__ __forceinline__ void encode_tile_sh_dispatch(const SecDesc &s, uint32_t tile_local, const EncodeCtx &c,
                                                        uint32_t sh_d) {
  if (sh_d == 45u) encode_tile<KIND, 45, EncGeom>(s, tile_local, c);
  else if (sh_d == 24u) encode_tile<KIND, 24, EncGeom>(s, tile_local, c);
  else encode_tile<KIND, 9, EncGeom>(s, tile_local, c);
}

}  // namespace

__global__ __launch_bounds__(EncGeom::kBlock, SPZ_ENC_MIN_WAVES) void spz_encode_kernel(const KParams p) {
  __shared__ float lut[kTableFloats];
  bool lut_ready = false;
  EncodeCtx c;
  c.flip_p = p.flip_p;
  c.flip_q = p.flip_q;
  c.sh_mask_ext = p.sh_mask_ext;
  c.lut = lut;
  if (blockIdx.x == 0 && threadIdx.x < 16 && p.header_dst != nullptr) {
    // PackedGaussiansHeader, load-spz.cc:131-139,534-539
    p.header_dst[threadIdx.x] = (uint8_t)(p.header_words[threadIdx.x >> 2] >> ((threadIdx.x & 3u) * 8u));
  }
  for (uint32_t tile = first_tile(p); tile < p.total_tiles; tile = SPZ_XCD_REMAP ? 0xffffffffu : tile + gridDim.x) {
    const TileRef ref = locate_tile(p, tile);
    const SecDesc &s = p.sec[ref.si];
    const uint32_t tl = ref.tl;
    switch (s.kind) {
      case KIND_POS24: encode_tile<KIND_POS24, 0, EncGeom>(s, tl, c); break;
      case KIND_ALPHA:
        if (!lut_ready) {
          stage_tables<EncGeom>(lut, p.tables);
          lut_ready = true;
        }
        encode_tile<KIND_ALPHA, 0, EncGeom>(s, tl, c);
        break;
      case KIND_COLOR: encode_tile<KIND_COLOR, 0, EncGeom>(s, tl, c); break;
      case KIND_SCALE: encode_tile<KIND_SCALE, 0, EncGeom>(s, tl, c); break;
      case KIND_ROT_S3: encode_tile<KIND_ROT_S3, 0, EncGeom>(s, tl, c); break;
      case KIND_ROT_F3: encode_tile<KIND_ROT_F3, 0, EncGeom>(s, tl, c); break;
      case KIND_SH: encode_tile_sh_dispatch<KIND_SH>(s, tl, c, p.sh_d); break;
      default: break;
    }
  }
}

namespace {

// ------------------------------------------------------------------------------------------
// Standalone in-place convertCoordinates (splat-types.h:134-164)
// ------------------------------------------------------------------------------------------
template <int KIND, int D>
__device__ __forceinline__ F32x4 flip_unit(F32x4 v, uint32_t phase, const KParams &p) {
  uint32_t sb;
  if constexpr (KIND == KIND_FLIP_POS) sb = pos_sign_bits(p.flip_p, phase);
  else if constexpr (KIND == KIND_FLIP_ROT) sb = p.flip_q & 7u;
  else sb = (uint32_t)(p.sh_mask_ext >> phase) & 0xfu;
  F32x4 o;
  o.x = mul_pm1(v.x, sb & 1u);
  o.y = mul_pm1(v.y, (sb >> 1) & 1u);
  o.z = mul_pm1(v.z, (sb >> 2) & 1u);
  o.w = (KIND == KIND_FLIP_ROT) ? v.w : mul_pm1(v.w, (sb >> 3) & 1u);  // w is never touched (:146)
  return o;
}

template <int KIND, int D>
__device__ __forceinline__ void flip_tile(const SecDesc &s, uint32_t tile_local, const KParams &p) {
  const unsigned long long base = (unsigned long long)tile_local * FlipGeom::kTileUnits;
  const uint32_t tid = threadIdx.x;
  float *buf = s.floats;
  const unsigned long long full_units = s.n_elems >> 2;
  const uint32_t pb = tile_phase_base<KIND, D, FlipGeom>(tile_local);
#pragma unroll
  for (int r = 0; r < FlipGeom::kUnroll; ++r) {
    const uint32_t local = FlipGeom::local_unit(r, tid);
    const unsigned long long u = base + local;
    const uint32_t ph = unit_phase<KIND, D>(pb, local);
    if (u < full_units) {
      store_f4<false>(buf, u, flip_unit<KIND, D>(load_f4<false>(buf, u), ph, p));
    } else if (u < s.n_units) {
      const uint32_t m = (uint32_t)(s.n_elems - u * 4ull);
      float *q = buf + u * 4ull;
      F32x4 v;
      v.x = q[0];
      v.y = (m > 1) ? q[1] : 0.0f;
      v.z = (m > 2) ? q[2] : 0.0f;
      v.w = 0.0f;
      F32x4 o = flip_unit<KIND, D>(v, ph, p);
      q[0] = o.x;
      if (m > 1) q[1] = o.y;
      if (m > 2) q[2] = o.z;
    }
  }
}

}  // namespace

__global__ __launch_bounds__(FlipGeom::kBlock) void spz_flip_kernel(const KParams p) {
  for (uint32_t tile = first_tile(p); tile < p.total_tiles; tile = SPZ_XCD_REMAP ? 0xffffffffu : tile + gridDim.x) {
    const uint32_t si = find_section(p, tile);
    const SecDesc &s = p.sec[si];
    const uint32_t tl = tile - s.tile_begin;
    switch (s.kind) {
      case KIND_FLIP_POS: flip_tile<KIND_FLIP_POS, 0>(s, tl, p); break;
      case KIND_FLIP_ROT: flip_tile<KIND_FLIP_ROT, 0>(s, tl, p); break;
      case KIND_FLIP_SH:
        if (p.sh_d == 45u) flip_tile<KIND_FLIP_SH, 45>(s, tl, p);
        else if (p.sh_d == 24u) flip_tile<KIND_FLIP_SH, 24>(s, tl, p);
        else flip_tile<KIND_FLIP_SH, 9>(s, tl, p);
        break;
      default: break;
    }
  }
}

namespace {

// One gathered value of an element-wise section: element e of the output is float (e % FPP) of the
// point indices[e / FPP].  Same arithmetic as the bulk decode (decode_unit), one element at a time.
template <int KIND, int FPP>
__device__ __forceinline__ float gather_element(const GatherParams &p, unsigned long long e) {
  const unsigned long long g = e / (unsigned)FPP;
  const uint32_t o = (uint32_t)(e - g * (unsigned)FPP);
  uint32_t i = p.indices[g];
  i = i < p.num_points ? i : p.num_points - 1u;
  if constexpr (KIND == KIND_POS24) {
    const uint8_t *b = p.positions + ((unsigned long long)i * 3u + o) * 3u;
    const uint32_t v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
    return xor_sign((float)((int32_t)(v << 8) >> 8) * p.pos_scale, (p.flip_p >> o) & 1u);
  } else if constexpr (KIND == KIND_POS16) {
    const uint8_t *b = p.positions + ((unsigned long long)i * 3u + o) * 2u;
    return mul_pm1(half_to_float((uint32_t)b[0] | ((uint32_t)b[1] << 8)), (p.flip_p >> o) & 1u);
  } else if constexpr (KIND == KIND_ALPHA) {
    return p.tables[kTableAlphaDec + p.alphas[i]];
  } else if constexpr (KIND == KIND_COLOR) {
    return p.tables[kTableColorDec + p.colors[(unsigned long long)i * 3u + o]];
  } else {  // KIND_SCALE
    return (float)p.scales[(unsigned long long)i * 3u + o] / 16.0f - 10.0f;
  }
}

}  // namespace

// IDX: 32-bit while the element count fits, so that the division by the constant D is one multiply-high.
template <uint32_t D, class IDX>
__device__ __forceinline__ void gather_sh(const GatherParams &p, unsigned long long t0, unsigned long long stride) {
  const unsigned long long total = p.count * D;
  const unsigned long long mask = p.sh_elem_mask;
  for (unsigned long long e = t0; e < total; e += stride) {
    const IDX g = (IDX)e / (IDX)D;
    const uint32_t o = (uint32_t)((IDX)e - g * (IDX)D);
    uint32_t i = p.indices[g];
    i = i < p.num_points ? i : p.num_points - 1u;
    const float v = ((float)p.sh[(unsigned long long)i * D + o] - 128.0f) / 128.0f;
    p.out_sh[e] = xor_sign(v, (uint32_t)(mask >> o) & 1u);
  }
}

__global__ __launch_bounds__(256) void spz_decode_gather_kernel(const GatherParams p) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long c = p.count;
  const uint32_t d = p.sh_dim * 3u;
  // sh: element e -> point e / d, element e % d of its record.  One byte and one float per thread keeps every
  // lane busy and a point's 45 reads in one or two sectors (measured against 4-float units per thread and
  // against a lane group per point: both lose on random indices, profiles/r01_gather_bench.jsonl).
  if (c * 45ull <= 0xffffffffull) {
    if (d == 45u) gather_sh<45u, uint32_t>(p, t0, stride);
    else if (d == 24u) gather_sh<24u, uint32_t>(p, t0, stride);
    else if (d == 9u) gather_sh<9u, uint32_t>(p, t0, stride);
  } else {
    if (d == 45u) gather_sh<45u, unsigned long long>(p, t0, stride);
    else if (d == 24u) gather_sh<24u, unsigned long long>(p, t0, stride);
    else if (d == 9u) gather_sh<9u, unsigned long long>(p, t0, stride);
  }
  for (unsigned long long e = t0; e < c * 3u; e += stride) {
    p.out_positions[e] = (p.version == 1u) ? gather_element<KIND_POS16, 3>(p, e) : gather_element<KIND_POS24, 3>(p, e);
    p.out_scales[e] = gather_element<KIND_SCALE, 3>(p, e);
    p.out_colors[e] = gather_element<KIND_COLOR, 3>(p, e);
  }
  for (unsigned long long g = t0; g < c; g += stride) {
    p.out_alphas[g] = gather_element<KIND_ALPHA, 1>(p, g);
    uint32_t i = p.indices[g];
    i = i < p.num_points ? i : p.num_points - 1u;
    F32x4 q;
    if (p.version >= 3u) {
      const uint8_t *b = p.rotations + (unsigned long long)i * 4u;
      q = unpack_quat_smallest_three((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24),
                                     p.flip_q);
    } else {
      const uint8_t *b = p.rotations + (unsigned long long)i * 3u;
      q = unpack_quat_first_three((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16), p.flip_q);
    }
    store_f4<false>(p.out_rotations, g, q);
  }
}

namespace {

// ------------------------------------------------------------------------------------------
// Self test of the fast quaternion arithmetic (SelfTestMode in spz_kernel_params.hpp)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {  // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// A float inside the quat_fast_ok window from 32 hashed bits: sign, exponent 2^-40 .. 2^39, mantissa;
// about one in sixteen is a zero.
__device__ __forceinline__ float window_float(uint32_t h) {
  const uint32_t sign = h & 0x80000000u;
  if (((h >> 23) & 0xfu) == 0u && (h & 0x40000000u)) return __uint_as_float(sign);
  const uint32_t e = 87u + ((h >> 23) & 0xffu) % 80u;
  return __uint_as_float(sign | (e << 23) | (h & 0x7fffffu));
}

// Quaternions whose components are close in magnitude (near-ties of the arg-max) or spread over the window.
__device__ __forceinline__ F32x4 window_quat(unsigned long long i) {
  const unsigned long long h0 = mix64(i * 2ull), h1 = mix64(i * 2ull + 1ull);
  F32x4 r;
  r.x = window_float((uint32_t)h0);
  r.y = window_float((uint32_t)(h0 >> 32));
  r.z = window_float((uint32_t)h1);
  r.w = window_float((uint32_t)(h1 >> 32));
  if ((i & 3ull) != 0ull) {  // three of four: every component within a few ulp .. a factor 2 of the first
    const uint32_t base = __float_as_uint(r.x) & 0x7fffffffu;
    const uint32_t spread = (i & 4ull) ? 0x7u : 0x7fffffu;
    auto near = [&](float v, uint32_t h) {
      const uint32_t b = base == 0u ? 0x3f800000u : base;
      return __uint_as_float((__float_as_uint(v) & 0x80000000u) | ((b & ~spread) | (h & spread)));
    };
    r.y = near(r.y, (uint32_t)(h0 >> 40));
    r.z = near(r.z, (uint32_t)(h1 >> 8));
    r.w = near(r.w, (uint32_t)(h1 >> 40));
  }
  return r;
}

}  // namespace

__global__ __launch_bounds__(256) void spz_selftest_kernel(const SelfTestParams p) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long bad = 0, first = ~0ull, seen = 0;
  for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < p.count; k += stride) {
    const unsigned long long i = p.begin + k;
    bool differs = false, used = true;
    if (p.mode == SELFTEST_DIV_SQRT1_2 || p.mode == SELFTEST_DIV_511) {
      const float x = __uint_as_float((uint32_t)i);
      used = i <= 0x7f800000ull && (i == 0ull || (x >= 0x1p-100f && x <= 0x1p126f));
      if (used) {
        float fast, plain;
        if (p.mode == SELFTEST_DIV_SQRT1_2) {
          fast = div_by_const(x, kSqrt1_2, kRcpSqrt1_2);
          plain = x / kSqrt1_2;
        } else {
          fast = div_by_const(x, 511.0f, kRcp511);
          plain = x / 511.0f;
        }
        differs = __float_as_uint(fast) != __float_as_uint(plain);
      }
    } else if (p.mode == SELFTEST_SQRT) {
      const float x = __uint_as_float((uint32_t)i);
      used = i <= 0x7f800000ull && x >= 0x1p-80f && x <= 0x1p82f;
      if (used) differs = __float_as_uint(sqrt_cr(x)) != __float_as_uint(__builtin_sqrtf(x));
    } else if (p.mode == SELFTEST_QUOTIENT) {
      // b: exponent 2^-40 .. 2^41; a: zero or 2^-40 <= a <= 2 b
      const unsigned long long h = mix64(i);
      const uint32_t eb = 87u + (uint32_t)(h & 0xffu) % 82u;
      const float b = __uint_as_float((eb << 23) | (uint32_t)((h >> 8) & 0x7fffffu));
      uint32_t ea = eb + 1u - (uint32_t)((h >> 31) & 0xffu) % (eb + 2u - 87u);
      float a = __uint_as_float((ea << 23) | (uint32_t)((h >> 40) & 0x7fffffu));
      if ((i & 1ull) != 0ull) a = __uint_as_float((__float_as_uint(b) & ~0xfu) | (uint32_t)((h >> 60) & 0xfu));  // a ~ b
      if (a > 2.0f * b) a = b;
      const float fast = quat_quotient(a, b, refined_rcp(b));
      const float plain = a / b;
      differs = __float_as_uint(fast) != __float_as_uint(plain);
    } else if (p.mode == SELFTEST_PACK_S3 || p.mode == SELFTEST_PACK_F3) {
      const F32x4 r = window_quat(i);
      const uint32_t flip_q = (uint32_t)(mix64(~i) & 7u);
      used = quat_fast_ok(r);
      if (used) {
        const Quat4 qf = normalized_flipped_fast(r, flip_q), qg = normalized_flipped(r, flip_q);
        differs = __float_as_uint(qf.q0) != __float_as_uint(qg.q0) || __float_as_uint(qf.q1) != __float_as_uint(qg.q1) ||
                  __float_as_uint(qf.q2) != __float_as_uint(qg.q2) || __float_as_uint(qf.q3) != __float_as_uint(qg.q3);
        if (p.mode == SELFTEST_PACK_S3) differs = differs || smallest_three_fields_fast(qf) != smallest_three_fields(qg);
      }
    } else if (p.mode == SELFTEST_UNPACK_S3) {
      const uint32_t flip_q = (uint32_t)(i >> 32) & 7u;
      const F32x4 f = unpack_quat_smallest_three_t<true>((uint32_t)i, flip_q), g = unpack_quat_smallest_three_t<false>((uint32_t)i, flip_q);
      differs = __float_as_uint(f.x) != __float_as_uint(g.x) || __float_as_uint(f.y) != __float_as_uint(g.y) ||
                __float_as_uint(f.z) != __float_as_uint(g.z) || __float_as_uint(f.w) != __float_as_uint(g.w);
    } else {
      used = false;
    }
    seen += used ? 1ull : 0ull;
    if (differs) {
      ++bad;
      first = i < first ? i : first;
    }
  }
  // one atomic per wave
  for (int off = 32; off >= 1; off >>= 1) {
    bad += __shfl_down(bad, off);
    seen += __shfl_down(seen, off);
    const unsigned long long o = __shfl_down(first, off);
    first = o < first ? o : first;
  }
  if ((threadIdx.x & 63u) == 0u) {
    if (bad) atomicAdd(&p.mismatches[0], bad);
    if (first != ~0ull) atomicMin(&p.mismatches[1], first);
    atomicAdd(&p.mismatches[2], seen);
  }
}

}  // namespace spz_amd_detail