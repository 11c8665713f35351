/* fastdiv_emulation — CPU emulation of the scaling-free divisions of spz_kernels.hip (div_by_const, quat_quotient,
 * sqrt_cr), used while designing them; the proof that counts runs on the GPU itself (spz_amd_selftest_device).
 *   gcc -O2 -mfma -fopenmp -ffp-contract=off -o build/fastdiv_emulation tools/fastdiv_emulation.c -lm && build/fastdiv_emulation
 * 1. x / 0.70710677f and x / 511.f by reciprocal multiply + one fma residual correction against the IEEE quotient for
 *    EVERY non-negative float: prints the exponent ranges in which they differ (only x < 2^-105, where the residual
 *    underflows, and x > 2^126, where the quotient overflows: outside the kernels' window).
 * 2. the shared-reciprocal quotient (the IEEE expansion's own two corrections, reciprocal perturbed by -1/0/+1 ulp to
 *    stand for v_rcp_f32's 1-ulp result) and the neighbour-testing square root (v_sqrt_f32 likewise) on 2e9 hashed
 *    operands inside the window: zero differences expected. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static inline float div_by_const(float x, float c, float r) {
  float m = x * r, rem = fmaf(-m, c, x);
  return fmaf(rem, r, m);
}
static inline float quat_quotient(float a, float b, float y0) {
  float e = fmaf(-b, y0, 1.0f), y = fmaf(e, y0, y0);
  float m = a * y, r0 = fmaf(-b, m, a), q1 = fmaf(r0, y, m), r1 = fmaf(-b, q1, a);
  return fmaf(r1, y, q1);
}
static inline float sqrt_cr(float x, int pert) {
  float s = u2f(f2u(sqrtf(x)) + pert), dn = u2f(f2u(s) - 1), up = u2f(f2u(s) + 1);
  float rdn = fmaf(-dn, s, x), rup = fmaf(-up, s, x);
  float o = (rdn <= 0) ? dn : s;
  return (rup > 0) ? up : o;
}

int main(void) {
  volatile float cv = (float)0.707106781186547524401;
  const float consts[2] = {cv, 511.0f};
  for (int k = 0; k < 2; ++k) {
    const float c = consts[k], r = 1.0f / c;
    long long hist[256] = {0};
#pragma omp parallel for
    for (long long i = 0; i <= 0x7f800000LL; ++i) {
      const float x = u2f((uint32_t)i);
      if (f2u(div_by_const(x, c, r)) != f2u(x / c)) {
#pragma omp atomic
        hist[i >> 23]++;
      }
    }
    printf("x / %a: biased exponents with differences:", c);
    for (int e = 0; e < 256; ++e) if (hist[e]) printf(" %d(%lld)", e, hist[e]);
    printf("\n");
  }
  long long badq = 0, bads = 0;
#pragma omp parallel for reduction(+ : badq, bads)
  for (long long i = 0; i < 2000000000LL; ++i) {
    const uint64_t h = mix64((uint64_t)i);
    const uint32_t eb = 87u + (uint32_t)(h & 0xff) % 82u;
    const float b = u2f((eb << 23) | (uint32_t)((h >> 8) & 0x7fffff));
    const uint32_t ea = eb + 1u - (uint32_t)((h >> 31) & 0xff) % (eb + 2u - 87u);
    float a = u2f((ea << 23) | (uint32_t)((h >> 40) & 0x7fffff));
    if (i & 1) a = u2f((f2u(b) & ~0xfu) | (uint32_t)((h >> 60) & 0xf));
    if (a > 2.0f * b) a = b;
    const int pert = (int)((h >> 20) % 3) - 1;
    if (f2u(quat_quotient(a, b, u2f(f2u(1.0f / b) + pert))) != f2u(a / b)) badq++;
    const uint32_t es = 47u + (uint32_t)((h >> 12) & 0xff) % 162u;
    const float x = u2f((es << 23) | (uint32_t)((h >> 33) & 0x7fffff));
    if (f2u(sqrt_cr(x, pert)) != f2u(sqrtf(x))) bads++;
  }
  printf("shared-reciprocal quotient: %lld differences; neighbour-tested sqrt: %lld differences (2e9 operands each)\n", badq, bads);
  return (badq || bads) ? 1 : 0;
}
