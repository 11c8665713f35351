// membench.hip — what this GPU's HBM delivers for the ACCESS SHAPES of the SPZ kernels, with the
// arithmetic removed.  Gives the measured ceiling the decode / encode kernels are compared against
// (cdna_hip_programming.md §5.4 rule 10: a ceiling claim needs a reference measured on the same
// hardware).  Build: hipcc --offload-arch=gfx950 -O3 -o build/membench tools/membench.hip
//
//   copy16      float4 -> float4 copy (the guide's 6.29 TB/s reference shape)
//   fill16      write-only float4 stream
//   read16      read-only float4 stream (sum folded into a never-taken store)
//   dec_shape   per unit: read 4 B, write 16 B   (decode's 65 B in / 236 B out, same 22/78 mix for sh)
//   enc_shape   per unit: read 16 B, write 4 B   (encode's mix)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
      std::exit(1);                                                               \
    }                                                                             \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy16(const v4f *__restrict__ in, v4f *__restrict__ out, size_t n) {
  size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  v4f v[U];
#pragma unroll
  for (int r = 0; r < U; ++r) {
    size_t i = base + (size_t)r * 256;
    if (i < n) v[r] = NT ? __builtin_nontemporal_load(in + i) : in[i];
  }
#pragma unroll
  for (int r = 0; r < U; ++r) {
    size_t i = base + (size_t)r * 256;
    if (i < n) out[i] = v[r];
  }
}

template <int U>
__global__ __launch_bounds__(256) void fill16(v4f *__restrict__ out, size_t n, float x) {
  size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  v4f v = {x, x + 1, x + 2, x + 3};
#pragma unroll
  for (int r = 0; r < U; ++r) {
    size_t i = base + (size_t)r * 256;
    if (i < n) out[i] = v;
  }
}

template <int U>
__global__ __launch_bounds__(256) void read16(const v4f *__restrict__ in, float *__restrict__ sink, size_t n) {
  size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < U; ++r) {
    size_t i = base + (size_t)r * 256;
    if (i < n) {
      v4f v = in[i];
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 1.2345e-30f) sink[0] = acc;  // never true; keeps the loads alive
}

template <int U>
__global__ __launch_bounds__(256) void dec_shape(const uint32_t *__restrict__ in, v4f *__restrict__ out, size_t n) {
  size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  uint32_t w[U];
#pragma unroll
  for (int r = 0; r < U; ++r) {
    size_t i = base + (size_t)r * 256;
    if (i < n) w[r] = in[i];
  }
#pragma unroll
  for (int r = 0; r < U; ++r) {
    size_t i = base + (size_t)r * 256;
    if (i < n) {
      v4f v = {(float)(w[r] & 0xff), (float)((w[r] >> 8) & 0xff), (float)((w[r] >> 16) & 0xff), (float)(w[r] >> 24)};
      out[i] = v;
    }
  }
}

// decode shape with 16-byte loads: a wave reads 1 KiB of packed bytes with one dwordx4 per lane,
// transposes through LDS (lane i of round r needs dword r*64+i), and writes four float4 rounds.
template <bool NT>
__global__ __launch_bounds__(256) void dec_shape16(const uint32_t *__restrict__ in, v4f *__restrict__ out, size_t n) {
  __shared__ uint32_t lds[256 * 4];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  size_t base = (size_t)blockIdx.x * 1024 + (size_t)wave * 256;   // dwords
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  if (base + 256 <= n) {
    u4 w = NT ? __builtin_nontemporal_load(reinterpret_cast<const u4 *>(in + base) + lane)
              : reinterpret_cast<const u4 *>(in + base)[lane];
    uint32_t *my = lds + wave * 256;
    reinterpret_cast<u4 *>(my)[lane] = w;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t d = my[r * 64 + lane];
      v4f v = {(float)(d & 0xff), (float)((d >> 8) & 0xff), (float)((d >> 16) & 0xff), (float)(d >> 24)};
      if (NT) __builtin_nontemporal_store(v, out + base + r * 64 + lane);
      else out[base + r * 64 + lane] = v;
    }
  }
}

template <int U>
__global__ __launch_bounds__(256) void dec_shape_nt(const uint32_t *__restrict__ in, v4f *__restrict__ out, size_t n) {
  size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  uint32_t w[U];
#pragma unroll
  for (int r = 0; r < U; ++r) {
    size_t i = base + (size_t)r * 256;
    if (i < n) w[r] = __builtin_nontemporal_load(in + i);
  }
#pragma unroll
  for (int r = 0; r < U; ++r) {
    size_t i = base + (size_t)r * 256;
    if (i < n) {
      v4f v = {(float)(w[r] & 0xff), (float)((w[r] >> 8) & 0xff), (float)((w[r] >> 16) & 0xff), (float)(w[r] >> 24)};
      __builtin_nontemporal_store(v, out + i);
    }
  }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void enc_shape(const v4f *__restrict__ in, uint32_t *__restrict__ out, size_t n) {
  size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  v4f v[U];
#pragma unroll
  for (int r = 0; r < U; ++r) {
    size_t i = base + (size_t)r * 256;
    if (i < n) v[r] = NT ? __builtin_nontemporal_load(in + i) : in[i];
  }
#pragma unroll
  for (int r = 0; r < U; ++r) {
    size_t i = base + (size_t)r * 256;
    if (i < n) {
      out[i] = (uint32_t)v[r].x | ((uint32_t)v[r].y << 8) | ((uint32_t)v[r].z << 16) | ((uint32_t)v[r].w << 24);
    }
  }
}

template <class F>
void time_it(const char *name, double bytes, int iters, F launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipDeviceSynchronize());
  std::vector<float> ms(iters);
  for (int i = 0; i < iters; ++i) {
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms[i], a, b));
  }
  std::sort(ms.begin(), ms.end());
  double med = ms[iters / 2];
  std::printf("{\"shape\": \"%s\", \"bytes\": %.0f, \"ms_med\": %.4f, \"ms_min\": %.4f, \"GBps_med\": %.1f}\n", name, bytes, med,
              ms[0], bytes / (med * 1e-3) / 1e9);
  std::fflush(stdout);
}

int main() {
  const size_t n16 = 147500000;  // float4 units of the bench workload (10 M x 59 floats / 4)
  v4f *a, *b;
  uint32_t *w;
  float *sink;
  CHECK(hipMalloc(&a, n16 * 16));
  CHECK(hipMalloc(&b, n16 * 16));
  CHECK(hipMalloc(&w, n16 * 4));
  CHECK(hipMalloc(&sink, 16));
  CHECK(hipMemset(a, 1, n16 * 16));
  CHECK(hipMemset(b, 2, n16 * 16));
  CHECK(hipMemset(w, 3, n16 * 4));
  const int it = 20;
  auto grid = [&](int U) { return dim3((unsigned)((n16 + 256 * U - 1) / (256 * U))); };
  time_it("copy16_u4", 2.0 * n16 * 16, it, [&] { hipLaunchKernelGGL((copy16<4, false>), grid(4), dim3(256), 0, 0, a, b, n16); });
  time_it("copy16_u8", 2.0 * n16 * 16, it, [&] { hipLaunchKernelGGL((copy16<8, false>), grid(8), dim3(256), 0, 0, a, b, n16); });
  time_it("copy16_u4_ntl", 2.0 * n16 * 16, it, [&] { hipLaunchKernelGGL((copy16<4, true>), grid(4), dim3(256), 0, 0, a, b, n16); });
  time_it("fill16_u4", 1.0 * n16 * 16, it, [&] { hipLaunchKernelGGL((fill16<4>), grid(4), dim3(256), 0, 0, b, n16, 1.0f); });
  time_it("fill16_u8", 1.0 * n16 * 16, it, [&] { hipLaunchKernelGGL((fill16<8>), grid(8), dim3(256), 0, 0, b, n16, 1.0f); });
  time_it("read16_u4", 1.0 * n16 * 16, it, [&] { hipLaunchKernelGGL((read16<4>), grid(4), dim3(256), 0, 0, a, sink, n16); });
  time_it("read16_u8", 1.0 * n16 * 16, it, [&] { hipLaunchKernelGGL((read16<8>), grid(8), dim3(256), 0, 0, a, sink, n16); });
  time_it("dec_shape_u4", 20.0 * n16, it, [&] { hipLaunchKernelGGL((dec_shape<4>), grid(4), dim3(256), 0, 0, w, b, n16); });
  time_it("dec_shape_u8", 20.0 * n16, it, [&] { hipLaunchKernelGGL((dec_shape<8>), grid(8), dim3(256), 0, 0, w, b, n16); });
  time_it("dec_shape_u4_nt", 20.0 * n16, it, [&] { hipLaunchKernelGGL((dec_shape_nt<4>), grid(4), dim3(256), 0, 0, w, b, n16); });
  time_it("dec_shape16", 20.0 * n16, it, [&] { hipLaunchKernelGGL((dec_shape16<false>), grid(4), dim3(256), 0, 0, w, b, n16); });
  time_it("dec_shape16_nt", 20.0 * n16, it, [&] { hipLaunchKernelGGL((dec_shape16<true>), grid(4), dim3(256), 0, 0, w, b, n16); });
  time_it("dec_shape_u4_nt_again", 20.0 * n16, it, [&] { hipLaunchKernelGGL((dec_shape_nt<4>), grid(4), dim3(256), 0, 0, w, b, n16); });
  time_it("enc_shape_u4", 20.0 * n16, it, [&] { hipLaunchKernelGGL((enc_shape<4, false>), grid(4), dim3(256), 0, 0, a, w, n16); });
  time_it("enc_shape_u4_ntl", 20.0 * n16, it, [&] { hipLaunchKernelGGL((enc_shape<4, true>), grid(4), dim3(256), 0, 0, a, w, n16); });
  time_it("enc_shape_u8_ntl", 20.0 * n16, it, [&] { hipLaunchKernelGGL((enc_shape<8, true>), grid(8), dim3(256), 0, 0, a, w, n16); });
  hipMemcpyAsync(b, a, n16 * 16, hipMemcpyDeviceToDevice, 0);
  time_it("hipMemcpyDtoD", 2.0 * n16 * 16, it, [&] { (void)hipMemcpyAsync(b, a, n16 * 16, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
