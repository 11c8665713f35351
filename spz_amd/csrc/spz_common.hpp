// spz_common.hpp — pieces shared by the HIP translation units of libspz_amd.so
// (spz_kernels.hip + spz_abi.hip: pack/unpack/flip kernels and their device entry points; spz_hostpath.hip: the
// host-pointer entry points; spz_ply_kernels.hip: .ply row shuffles; spz_median.hip; spz_exchange.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <mutex>

#include "spz_amd.h"

namespace spz_amd_detail {

// 16-byte float access that needs only dword alignment.
struct __attribute__((packed, aligned(4))) F32x4 { float x, y, z, w; };

__device__ __forceinline__ bool is_nan_bits(uint32_t b) { return (b & 0x7fffffffu) > 0x7f800000u; }


// v * (neg ? -1.0f : 1.0f) with the x86 mulss result also for NaN (a NaN operand is returned
// quieted, sign untouched) and for zeros (0 * -1 = -0).
__device__ __forceinline__ float mul_pm1(float v, uint32_t neg) {
  uint32_t b = __float_as_uint(v);
  uint32_t flipped = b ^ (neg << 31);
  return __uint_as_float(is_nan_bits(b) ? (b | 0x00400000u) : flipped);
}


// ---- host side ---------------------------------------------------------------------------------
extern thread_local int g_last_hip_error;

#define SPZ_HIP_TRY(expr)                                   \
  do {                                                      \
    hipError_t e_ = (expr);                                 \
    if (e_ != hipSuccess) {                                 \
      ::spz_amd_detail::g_last_hip_error = (int)e_;         \
      return SPZ_AMD_ERR_HIP;                               \
    }                                                       \
  } while (0)

constexpr int kMaxDevices = 64;

// axesMatch + coordinateConverter (splat-types.h:43-81) reduced to sign masks.
struct FlipMasks {
  uint32_t p;           // bit a: axis a negated
  uint32_t q;           // bit i: quaternion x/y/z negated
  uint32_t sh15;        // bit k: sh coefficient k negated
};

inline FlipMasks flip_masks(int from, int to) {
  FlipMasks m = {0, 0, 0};
  int a = from - 1, b = to - 1;
  if (a < 0 || b < 0) return m;
  const uint32_t x = (((a >> 0) & 1) != ((b >> 0) & 1)) ? 1u : 0u;  // 1 = negative
  const uint32_t y = (((a >> 1) & 1) != ((b >> 1) & 1)) ? 1u : 0u;
  const uint32_t z = (((a >> 2) & 1) != ((b >> 2) & 1)) ? 1u : 0u;
  m.p = x | (y << 1) | (z << 2);
  m.q = (y ^ z) | ((x ^ z) << 1) | ((x ^ y) << 2);
  // flipSh = {y, z, x, xy, yz, 1, xz, 1, y, xyz, y, z, x, z, x}
  const uint32_t sh[15] = {y, z, x, x ^ y, y ^ z, 0, x ^ z, 0, y, x ^ y ^ z, y, z, x, z, x};
  for (int k = 0; k < 15; ++k) m.sh15 |= sh[k] << k;
  return m;
}


inline bool valid_coord(int c) { return c >= 0 && c <= 8; }

inline int sh_dim_for_degree(int d) {  // load-spz.cc:58-72
  switch (d) {
    case 0: return 0;
    case 1: return 3;
    case 2: return 8;
    case 3: return 15;
    default: return -1;
  }
}

// Current HIP device, or SPZ_AMD_ERR_NO_DEVICE when the runtime has none.
inline int current_device(int *device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_last_hip_error = (int)e;
    return SPZ_AMD_ERR_NO_DEVICE;
  }
  SPZ_HIP_TRY(hipGetDevice(device));
  return SPZ_AMD_OK;
}


// Device scratch for the *_host entry points: ONE grow-only allocation per device, kept between calls
// (a 10 M-point SH3 save needs 3 GB of it; allocating and freeing that per call cost more than the
// copies), guarded by a per-device mutex that is held for the duration of the host call and released by
// spz_amd_release_device_memory().  With it live the two copy streams and the events of the chunked
// H2D / kernel / D2H pipeline (spz_hostpath.hip).
constexpr int kPipeChunksMax = 64;
struct HostPipe {
  hipStream_t up = nullptr, down = nullptr;   // uploads + kernels / downloads
  hipEvent_t *events = nullptr;
  int n_events = 0;
};
int workspace_acquire(int device, size_t bytes, void **base, HostPipe **pipe);  // locks `device`
void workspace_release(int device);                                              // unlocks
void workspace_free_all();                                                       // drops the cached allocations

// Large transient device blocks of the container stage (spz_lz77.hip: ~23 bytes per input byte; spz_inflate_dev.hip:
// ~17 per compressed byte): kept between calls, up to three per device, because a hipMalloc of several GB that
// follows the hipFree of another has been seen to take 0.4 s on some boxes of the pool (the free is still in
// progress).  Returned by spz_amd_release_device_memory() like the host path's workspace.  The current device is
// `device`.  A block of at least `bytes`; when all cached ones are in use, a plain allocation that scratch_release frees.
int scratch_acquire(int device, size_t bytes, void **block);
void scratch_release(int device, void *block);   // idle blocks beyond SPZ_AMD_SCRATCH_KEEP_MIB (default 32 GiB) are freed here
// Device memory a new block can count on: what the runtime reports free plus the idle cached blocks (scratch_acquire
// gives those up first), capped by SPZ_AMD_DEVICE_MEM_LIMIT_MIB when set.  The current device is `device`.
int device_free_bytes(int device, size_t *free_bytes);

class Workspace {
 public:
  ~Workspace() {
    if (open_) workspace_release(device_);
  }
  // Reserves `bytes` (already a sum of aligned() sizes) on the CURRENT device `device`.
  int open(int device, size_t bytes) {
    int rc = workspace_acquire(device, bytes, &base_, &pipe_);
    if (rc != SPZ_AMD_OK) return rc;
    device_ = device;
    open_ = true;
    return SPZ_AMD_OK;
  }
  static size_t aligned(size_t bytes) { return (bytes + 255) & ~size_t(255); }
  void *take(size_t bytes) {
    void *p = bytes ? static_cast<char *>(base_) + used_ : nullptr;
    used_ += aligned(bytes);
    return p;
  }
  HostPipe *pipe() const { return pipe_; }

 private:
  void *base_ = nullptr;
  HostPipe *pipe_ = nullptr;
  size_t used_ = 0;
  int device_ = 0;
  bool open_ = false;
};

// A stream of its own for a stage: level +1 = in front of the others when both have work for the chip (the pipeline
// whose copies the link waits for), -1 = behind them (kernels that fill idle CUs beside it), 0 = the default.
// SPZ_AMD_STREAM_PRIORITIES=0 makes them all equal (measurements).
inline hipError_t create_stream(hipStream_t *st, int level) {
  static const bool use = []() {
    const char *e = std::getenv("SPZ_AMD_STREAM_PRIORITIES");
    return e == nullptr || std::atoi(e) != 0;
  }();
  int least = 0, greatest = 0;  // numerically: greatest priority = the smaller number
  if (!use || level == 0 || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) {
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
  }
  return hipStreamCreateWithPriority(st, hipStreamNonBlocking, level > 0 ? greatest : least);
}

// Host -> device copy of a large pageable range that the runtime has not seen before.  hipMemcpy pins such a range
// page by page first; where the range is backed by huge pages that is cheap and the copy runs at the link's rate
// (409 MB in 8 ms), where it is not (a fresh std::vector whose huge-page request the kernel could not honour, in whole or
// from some point on: seen in about one process in four on the pool's boxes) at 0.7-1.4 GB/s.  The range goes up in
// 32 MiB steps, each timed; the first slow one switches the rest to staging through two pinned 32 MiB buffers of
// this library's own (a memcpy per step, overlapped with the previous step's DMA: ~10 GB/s whatever backs the range).
// Blocking: the data is on the device when this returns.
inline hipError_t upload_adaptive(void *d_dst, const void *h_src, size_t bytes, hipStream_t st) {
  constexpr size_t kStep = size_t(32) << 20;
  hipError_t e;
  char *dst = static_cast<char *>(d_dst);
  const char *src = static_cast<const char *>(h_src);
  size_t left = bytes;
  // (the first steps are short — 4, 8, 16 MiB — so that a file of a few tens of MB whose pages pin slowly is not most
  // of the way up at 3 - 4 GB/s before that is noticed: a 41 MB member took 9 ms instead of 4)
  size_t step = size_t(4) << 20;
  while (left > 0) {  // direct, while it is fast
    const size_t n = left < step ? left : step;
    const bool full = n == step;
    if (step < kStep) step *= 2;
    const auto t0 = std::chrono::steady_clock::now();
    e = hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    dst += n;
    src += n;
    left -= n;
    if (full && static_cast<double>(n) / dt < 5e9) {
      if (std::getenv("SPZ_AMD_LZ_TIMING")) {
        std::fprintf(stderr, "[upload] %.1f GB/s after %zu MiB: the rest (%zu MiB) through pinned staging\n", n / dt / 1e9,
                     (bytes - left) >> 20, left >> 20);
      }
      break;
    }
  }
  if (left == 0) return hipSuccess;
  // staging buffers and events belong to the device they were made on (an event of another device cannot be recorded
  // on this one's stream): one set per device, made on first use
  static std::mutex mu;
  static void *pinned_all[kMaxDevices][2] = {};
  static hipEvent_t done_all[kMaxDevices][2] = {};
  int dev = 0;
  e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    e = hipMemcpyAsync(dst, src, left, hipMemcpyHostToDevice, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
  }
  std::lock_guard<std::mutex> lock(mu);
  void **pinned = pinned_all[dev];
  hipEvent_t *done = done_all[dev];
  for (int i = 0; i < 2; ++i) {
    if (pinned[i] == nullptr) {
      e = hipHostMalloc(&pinned[i], kStep, hipHostMallocDefault);
      if (e == hipSuccess) {
        e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
        if (e != hipSuccess) (void)hipHostFree(pinned[i]);
      }
      if (e != hipSuccess) {  // no pinned memory (or event) to be had: the runtime's own way
        pinned[i] = nullptr;
        done[i] = nullptr;
        e = hipMemcpyAsync(dst, src, left, hipMemcpyHostToDevice, st);
        return e != hipSuccess ? e : hipStreamSynchronize(st);
      }
    }
  }
  bool used[2] = {false, false};
  for (int k = 0; left > 0; ++k) {
    const int b = k & 1;
    const size_t n = left < kStep ? left : kStep;
    if (used[b]) {
      e = hipEventSynchronize(done[b]);
      if (e != hipSuccess) return e;
    }
    std::memcpy(pinned[b], src, n);
    e = hipMemcpyAsync(dst, pinned[b], n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(done[b], st);
    if (e != hipSuccess) return e;
    used[b] = true;
    dst += n;
    src += n;
    left -= n;
  }
  return hipStreamSynchronize(st);
}

struct DeviceGuard {
  int prev = -1;
  bool active = false;
  int enter(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
      g_last_hip_error = (int)e;
      return SPZ_AMD_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return SPZ_AMD_ERR_INVALID_ARG;
    SPZ_HIP_TRY(hipGetDevice(&prev));
    SPZ_HIP_TRY(hipSetDevice(device));
    active = true;
    return SPZ_AMD_OK;
  }
  ~DeviceGuard() {
    if (active && prev >= 0) (void)hipSetDevice(prev);
  }
};


}  // namespace spz_amd_detail
