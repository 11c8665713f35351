// spz_hostpath.hip — the host-pointer (PCIe) entry points of include/spz_amd.h: what GaussianCloud-level
// callers reach (saveSpz/loadSpz hand over host vectors, load-spz.cc:258-331,483-531).  The cached device
// workspace, the two copy streams and the chunked upload / kernel / download pipeline live here; the
// kernels are reached only through the device entry points of spz_abi.hip, so nothing in this file
// decides what a kernel launch looks like (it is not one of the sources profiles/pmc_traffic.json hashes).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include "spz_amd.h"
#include "spz_common.hpp"

namespace spz_amd_detail {

namespace {
std::mutex g_ws_mutex[kMaxDevices];
void *g_ws_ptr[kMaxDevices] = {};
size_t g_ws_cap[kMaxDevices] = {};
HostPipe g_pipe[kMaxDevices];
constexpr int kPipeEvents = kPipeChunksMax;  // one event per chunk of a host call

hipError_t pipe_create(HostPipe *p) {
  if (p->up) return hipSuccess;
  hipError_t e = create_stream(&p->up, +1);
  if (e == hipSuccess) e = create_stream(&p->down, +1);
  if (e == hipSuccess) {
    p->events = new hipEvent_t[kPipeEvents];
    for (p->n_events = 0; p->n_events < kPipeEvents && e == hipSuccess; ++p->n_events) {
      e = hipEventCreateWithFlags(&p->events[p->n_events], hipEventDisableTiming);
    }
  }
  return e;
}

void pipe_destroy(HostPipe *p) {
  for (int i = 0; i < p->n_events; ++i) (void)hipEventDestroy(p->events[i]);
  delete[] p->events;
  if (p->up) (void)hipStreamDestroy(p->up);
  if (p->down) (void)hipStreamDestroy(p->down);
  *p = HostPipe();
}
}  // namespace

int workspace_acquire(int device, size_t bytes, void **base, HostPipe **pipe) {
  if (device < 0 || device >= kMaxDevices) return SPZ_AMD_ERR_INVALID_ARG;
  g_ws_mutex[device].lock();
  *base = nullptr;
  hipError_t e = pipe_create(&g_pipe[device]);
  if (e == hipSuccess && g_ws_cap[device] < bytes) {
    if (g_ws_ptr[device]) (void)hipFree(g_ws_ptr[device]);
    g_ws_ptr[device] = nullptr;
    g_ws_cap[device] = 0;
    size_t want = bytes < (size_t(1) << 20) ? (size_t(1) << 20) : bytes;
    e = hipMalloc(&g_ws_ptr[device], want);
    if (e == hipSuccess) g_ws_cap[device] = want;
  }
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    g_ws_mutex[device].unlock();
    return SPZ_AMD_ERR_HIP;
  }
  *base = g_ws_ptr[device];
  *pipe = &g_pipe[device];
  return SPZ_AMD_OK;
}

void workspace_release(int device) { g_ws_mutex[device].unlock(); }

namespace {
struct ScratchSlot {
  void *ptr = nullptr;
  size_t bytes = 0;
  bool in_use = false;
};
constexpr int kScratchSlots = 3;
ScratchSlot g_scratch[kMaxDevices][kScratchSlots];
std::mutex g_scratch_mutex;
}  // namespace

// SPZ_AMD_SCRATCH_KEEP_MIB: how much idle scratch a device may keep between calls (default 32768 = 32 GiB: the blocks of
// one 10 M-point SH3 saveSpz + loadSpz are 15 + 8 GB; 0 = keep nothing).  Read at every release.
static size_t scratch_keep_bytes() {
  const char *e = std::getenv("SPZ_AMD_SCRATCH_KEEP_MIB");
  const unsigned long long mib = e ? std::strtoull(e, nullptr, 10) : 32768ull;
  return (size_t)mib << 20;
}

// SPZ_AMD_DEVICE_MEM_LIMIT_MIB: an artificial ceiling on what the container stage believes is free on the device (its
// decline-to-the-host paths can then be exercised on a 288 GB card); unset: what hipMemGetInfo says.
int device_free_bytes(int device, size_t *free_bytes) {
  size_t free_b = 0, total_b = 0;
  SPZ_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  {  // idle cached blocks are free for the asking: scratch_acquire gives them up before it allocates
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (device >= 0 && device < kMaxDevices) {
      for (int i = 0; i < kScratchSlots; ++i) {
        if (!g_scratch[device][i].in_use) free_b += g_scratch[device][i].bytes;
      }
    }
  }
  if (const char *e = std::getenv("SPZ_AMD_DEVICE_MEM_LIMIT_MIB")) {
    const size_t cap = (size_t)std::strtoull(e, nullptr, 10) << 20;
    if (cap < free_b) free_b = cap;
  }
  *free_bytes = free_b;
  return SPZ_AMD_OK;
}

int scratch_acquire(int device, size_t bytes, void **block) {
  *block = nullptr;
  if (device < 0 || device >= kMaxDevices) return SPZ_AMD_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  ScratchSlot *slots = g_scratch[device];
  int best = -1;
  for (int i = 0; i < kScratchSlots; ++i) {
    if (slots[i].in_use) continue;
    if (slots[i].ptr && slots[i].bytes >= bytes && (best < 0 || slots[i].bytes < slots[best].bytes)) best = i;
  }
  if (best >= 0) {
    slots[best].in_use = true;
    *block = slots[best].ptr;
    return SPZ_AMD_OK;
  }
  // nothing cached is large enough: idle blocks go (smallest first) until the new one fits beside what is left
  int slot = -1;
  for (;;) {
    size_t free_b = 0, total_b = 0;
    const bool known = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    int idle = -1, empty = -1;
    for (int i = 0; i < kScratchSlots; ++i) {
      if (slots[i].in_use) continue;
      if (slots[i].ptr == nullptr) empty = i;
      else if (idle < 0 || slots[i].bytes < slots[idle].bytes) idle = i;
    }
    if (empty >= 0 && (idle < 0 || (known && free_b >= bytes + (size_t(256) << 20)))) {
      slot = empty;
      break;
    }
    if (idle < 0) break;  // every slot busy: not cached
    (void)hipFree(slots[idle].ptr);
    slots[idle] = ScratchSlot();
  }
  void *p = nullptr;
  SPZ_HIP_TRY(hipMalloc(&p, bytes));
  if (slot >= 0) {
    slots[slot].ptr = p;
    slots[slot].bytes = bytes;
    slots[slot].in_use = true;
  }
  *block = p;
  return SPZ_AMD_OK;
}

void scratch_release(int device, void *block) {
  if (block == nullptr || device < 0 || device >= kMaxDevices) return;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  ScratchSlot *slots = g_scratch[device];
  bool cached = false;
  for (int i = 0; i < kScratchSlots; ++i) {
    if (slots[i].ptr == block) {
      slots[i].in_use = false;
      cached = true;
    }
  }
  if (!cached) {
    (void)hipFree(block);
    return;
  }
  // what stays idle between calls is bounded: the largest idle blocks go first until the rest is within the budget
  const size_t keep = scratch_keep_bytes();
  for (;;) {
    size_t idle_total = 0;
    int largest = -1;
    for (int i = 0; i < kScratchSlots; ++i) {
      if (slots[i].in_use || slots[i].ptr == nullptr) continue;
      idle_total += slots[i].bytes;
      if (largest < 0 || slots[i].bytes > slots[largest].bytes) largest = i;
    }
    if (largest < 0 || idle_total <= keep) break;
    (void)hipDeviceSynchronize();  // nothing of the call that used the block may still be queued when it goes
    const hipError_t e = hipFree(slots[largest].ptr);
    if (std::getenv("SPZ_AMD_LZ_TIMING")) {
      std::fprintf(stderr, "[scratch] idle %zu MiB > budget %zu MiB: freed a block of %zu MiB (%s)\n", idle_total >> 20, keep >> 20,
                   slots[largest].bytes >> 20, hipGetErrorString(e));
    }
    if (e != hipSuccess) g_last_hip_error = (int)e;
    slots[largest] = ScratchSlot();
  }
}

void kept_stream_free_idle();
void lz_streams_free_idle();  // spz_lz77.hip: the container stage's per-device streams
void workspace_free_all() {
  kept_stream_free_idle();
  lz_streams_free_idle();
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess) return;
  for (int d = 0; d < kMaxDevices; ++d) {
    std::lock_guard<std::mutex> lock(g_ws_mutex[d]);
    if (g_ws_ptr[d] || g_pipe[d].up) {
      if (hipSetDevice(d) == hipSuccess) {
        if (g_ws_ptr[d]) (void)hipFree(g_ws_ptr[d]);
        pipe_destroy(&g_pipe[d]);
      }
      g_ws_ptr[d] = nullptr;
      g_ws_cap[d] = 0;
    }
  }
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    for (int d = 0; d < kMaxDevices; ++d) {
      for (int i = 0; i < kScratchSlots; ++i) {
        ScratchSlot &sl = g_scratch[d][i];
        if (sl.ptr && !sl.in_use && hipSetDevice(d) == hipSuccess) {
          (void)hipFree(sl.ptr);
          sl = ScratchSlot();
        }
      }
    }
  }
  (void)hipSetDevice(prev);
}
}  // namespace spz_amd_detail

namespace {

using namespace spz_amd_detail;

// ---- chunked host pipeline -------------------------------------------------------------------
// PCIe is full duplex and the kernels are ~50x faster than the link, so a host-pointer call is cut into
// point-range chunks (the stream is attribute-major: a chunk is six fragments, exactly the shard entry
// points' geometry): the calling thread uploads chunk k and launches its kernel on pipe->up while a
// second thread downloads the results of chunk k-1 on pipe->down.  Two host threads because a copy
// between device and PAGEABLE host memory may occupy the thread that issues it.
int plan_chunks(uint64_t n, size_t float_bytes_per_point, uint64_t *chunk_points) {
  static const size_t target = []() {
    const char *e = std::getenv("SPZ_AMD_HOST_CHUNK_MIB");
    const long v = e ? std::atol(e) : 0;
    return (size_t)(v > 0 ? v : 160) << 20;  // profiles/r02_host_chunk_scan.txt
  }();
  const unsigned long long total = n * float_bytes_per_point;
  unsigned long long chunks = (total + target - 1) / target;
  chunks = chunks < 1 ? 1 : (chunks > (unsigned long long)kPipeChunksMax ? (unsigned long long)kPipeChunksMax : chunks);
  unsigned long long cp = (n + chunks - 1) / chunks;
  cp = (cp + 1023ull) & ~1023ull;
  *chunk_points = cp;
  return (int)((n + cp - 1) / cp);
}

template <class Up, class Down>
int run_pipeline(HostPipe *pipe, int device, int chunks, Up up, Down down) {
  if (chunks > pipe->n_events) return SPZ_AMD_ERR_INVALID_ARG;
  if (chunks == 1) {  // small calls: no second thread
    int rc = up(0);
    if (rc != SPZ_AMD_OK) return rc;
    SPZ_HIP_TRY(hipEventRecord(pipe->events[0], pipe->up));
    SPZ_HIP_TRY(hipStreamWaitEvent(pipe->down, pipe->events[0], 0));
    rc = down(0);
    hipError_t e = hipStreamSynchronize(pipe->down);
    if (rc != SPZ_AMD_OK) return rc;
    SPZ_HIP_TRY(e);
    return SPZ_AMD_OK;
  }
  std::mutex m;
  std::condition_variable cv;
  int recorded = 0;      // events[0 .. recorded) have been recorded on pipe->up
  bool stop = false;     // the uploader failed: the downloader must not wait for more
  int down_rc = SPZ_AMD_OK, down_hip = 0;
  auto download_all = [&]() {
    int rc = SPZ_AMD_OK;
    if (hipSetDevice(device) != hipSuccess) rc = SPZ_AMD_ERR_HIP;
    for (int k = 0; k < chunks && rc == SPZ_AMD_OK; ++k) {
      {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&]() { return recorded > k || stop; });
        if (recorded <= k) break;
      }
      if (hipStreamWaitEvent(pipe->down, pipe->events[k], 0) != hipSuccess) rc = SPZ_AMD_ERR_HIP;
      else rc = down(k);
    }
    const hipError_t e = hipStreamSynchronize(pipe->down);
    if (rc == SPZ_AMD_OK && e != hipSuccess) {
      g_last_hip_error = (int)e;
      rc = SPZ_AMD_ERR_HIP;
    }
    down_rc = rc;
    down_hip = g_last_hip_error;
  };
  std::thread downloader;
  try {
    downloader = std::thread(download_all);
  } catch (const std::system_error &) {
    // no second thread to be had: the chunks one after the other on this one (correct, not overlapped)
    for (int k = 0; k < chunks; ++k) {
      int rc = up(k);
      if (rc != SPZ_AMD_OK) return rc;
      SPZ_HIP_TRY(hipEventRecord(pipe->events[k], pipe->up));
      SPZ_HIP_TRY(hipStreamWaitEvent(pipe->down, pipe->events[k], 0));
      rc = down(k);
      if (rc != SPZ_AMD_OK) return rc;
    }
    SPZ_HIP_TRY(hipStreamSynchronize(pipe->down));
    return SPZ_AMD_OK;
  }
  int rc = SPZ_AMD_OK;
  for (int k = 0; k < chunks && rc == SPZ_AMD_OK; ++k) {
    rc = up(k);
    if (rc == SPZ_AMD_OK && hipEventRecord(pipe->events[k], pipe->up) != hipSuccess) rc = SPZ_AMD_ERR_HIP;
    {
      std::lock_guard<std::mutex> lock(m);
      if (rc == SPZ_AMD_OK) recorded = k + 1;
      else stop = true;
    }
    cv.notify_one();
  }
  downloader.join();
  if (rc != SPZ_AMD_OK) {
    (void)hipStreamSynchronize(pipe->up);
    return rc;
  }
  if (down_rc != SPZ_AMD_OK) {
    g_last_hip_error = down_hip;
    (void)hipStreamSynchronize(pipe->up);
  }
  return down_rc;
}

}  // namespace

// A device copy of the last stream an encode produced, for a caller that goes on to the container stage
// (spz_amd_encode_host_keep -> spz_amd_zlib_parse_open_dev): one grow-only buffer per device, handed to one caller at a time.
namespace {
struct KeptStream {
  void *ptr = nullptr;
  size_t cap = 0;
  bool in_use = false;
};
KeptStream g_kept[kMaxDevices];
std::mutex g_kept_mutex;

uint8_t *kept_acquire(int device, size_t bytes) {  // the current device is `device`; nullptr: not available
  std::lock_guard<std::mutex> lock(g_kept_mutex);
  KeptStream &k = g_kept[device];
  if (k.in_use) return nullptr;
  if (k.cap < bytes) {
    if (k.ptr) (void)hipFree(k.ptr);
    k = KeptStream();
    if (hipMalloc(&k.ptr, bytes) != hipSuccess) {
      k.ptr = nullptr;
      return nullptr;
    }
    k.cap = bytes;
  }
  k.in_use = true;
  return static_cast<uint8_t *>(k.ptr);
}

int encode_host_impl(const spz_amd_cloud_in *h, uint64_t n, int sh_degree, int antialiased, int from_coord, int version,
                     uint8_t *h_stream, size_t capacity, int device, const uint8_t **d_keep, void *zlib_session = nullptr,
                     size_t tail_bytes = 0, void (*tail_ready)(void *) = nullptr, void *tail_arg = nullptr);
}  // namespace

namespace spz_amd_detail {
void kept_stream_free_idle() {
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_kept_mutex);
  for (int d = 0; d < kMaxDevices; ++d) {
    KeptStream &k = g_kept[d];
    if (k.ptr && !k.in_use && hipSetDevice(d) == hipSuccess) {
      (void)hipFree(k.ptr);
      k = KeptStream();
    }
  }
  (void)hipSetDevice(prev);
}
}  // namespace spz_amd_detail

extern "C" {

void spz_amd_kept_stream_release(int device, const uint8_t *d_stream) {
  if (d_stream == nullptr || device < 0 || device >= kMaxDevices) return;
  std::lock_guard<std::mutex> lock(g_kept_mutex);
  if (g_kept[device].ptr == d_stream) g_kept[device].in_use = false;
}

int spz_amd_encode_host(const spz_amd_cloud_in *h, uint64_t n, int sh_degree, int antialiased, int from_coord,
                        int version, uint8_t *h_stream, size_t capacity, int device) {
  return encode_host_impl(h, n, sh_degree, antialiased, from_coord, version, h_stream, capacity, device, nullptr);
}

int spz_amd_encode_host_keep(const spz_amd_cloud_in *h, uint64_t n, int sh_degree, int antialiased, int from_coord,
                             int version, uint8_t *h_stream, size_t capacity, int device, const uint8_t **d_stream) {
  if (d_stream == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  *d_stream = nullptr;
  return encode_host_impl(h, n, sh_degree, antialiased, from_coord, version, h_stream, capacity, device, d_stream);
}

int spz_amd_encode_host_keep_session(const spz_amd_cloud_in *h, uint64_t n, int sh_degree, int antialiased, int from_coord,
                                     int version, uint8_t *h_stream, size_t capacity, int device, const uint8_t **d_stream,
                                     void *zlib_session) {
  if (d_stream == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  *d_stream = nullptr;
  return encode_host_impl(h, n, sh_degree, antialiased, from_coord, version, h_stream, capacity, device, d_stream, zlib_session);
}

int spz_amd_encode_host_keep_session_tail(const spz_amd_cloud_in *h, uint64_t n, int sh_degree, int antialiased, int from_coord,
                                          int version, uint8_t *h_stream, size_t capacity, int device, const uint8_t **d_stream,
                                          void *zlib_session, size_t tail_bytes, void (*tail_ready)(void *), void *tail_arg) {
  if (d_stream == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  *d_stream = nullptr;
  return encode_host_impl(h, n, sh_degree, antialiased, from_coord, version, h_stream, capacity, device, d_stream, zlib_session, tail_bytes,
                          tail_ready, tail_arg);
}

}  // extern "C"

namespace {
int encode_host_impl(const spz_amd_cloud_in *h, uint64_t n, int sh_degree, int antialiased, int from_coord, int version,
                     uint8_t *h_stream, size_t capacity, int device, const uint8_t **d_keep, void *zlib_session, size_t tail_bytes,
                     void (*tail_ready)(void *), void *tail_arg) {
  if (h == nullptr || h_stream == nullptr || !valid_coord(from_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  spz_amd_layout lay;
  int rc = spz_amd_stream_layout(n, sh_degree, version, &lay);
  if (rc != SPZ_AMD_OK) return rc;
  if (version == 1) return SPZ_AMD_ERR_UNSUPPORTED;
  if (capacity < lay.total_bytes) return SPZ_AMD_ERR_CAPACITY;
  const int sd = sh_dim_for_degree(sh_degree);
  if (n > 0 && (!h->positions || !h->scales || !h->rotations || !h->alphas || !h->colors || (sd > 0 && !h->sh))) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  if (n == 0) {  // a zero-point stream is its header (:534-539): nothing for a device to do
    const spz_amd_header hdr = {(uint32_t)version, 0u, (uint8_t)sh_degree, 12, (uint8_t)(antialiased ? 1 : 0), 0};
    return spz_amd_write_header(&hdr, h_stream);
  }
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  const size_t fpp[6] = {3, 3, 4, 1, 3, (size_t)sd * 3};  // floats per point, spz_amd_cloud_in order
  const float *src[6] = {h->positions, h->scales, h->rotations, h->alphas, h->colors, h->sh};
  // the device stream: in the workspace, or — for a caller that wants it kept — in the buffer it then owns
  uint8_t *kept = d_keep ? kept_acquire(device, Workspace::aligned(lay.total_bytes)) : nullptr;
  struct KeptGuard {
    int device;
    const uint8_t *p;
    ~KeptGuard() { spz_amd_kept_stream_release(device, p); }
  } kept_guard{device, kept};
  size_t total = kept ? 0 : Workspace::aligned(lay.total_bytes);
  for (int i = 0; i < 6; ++i) total += Workspace::aligned(n * fpp[i] * sizeof(float));
  Workspace ws;
  rc = ws.open(device, total);
  if (rc != SPZ_AMD_OK) return rc;
  float *fb[6];
  for (int i = 0; i < 6; ++i) fb[i] = static_cast<float *>(ws.take(n * fpp[i] * sizeof(float)));
  uint8_t *sb = kept ? kept : static_cast<uint8_t *>(ws.take(lay.total_bytes));
  HostPipe *pipe = ws.pipe();
  uint64_t cp = 0;
  const int chunks = plan_chunks(n, (14 + (size_t)sd * 3) * sizeof(float), &cp);
  // chunk k: upload the floats of points [k cp, ...), encode them into their six fragments of the device
  // stream; the downloader copies those fragments (and, with the first chunk, the header) to the host
  auto up = [&](int k) -> int {
    const uint64_t first = (uint64_t)k * cp, count = std::min<uint64_t>(cp, n - first);
    for (int i = 0; i < 6; ++i) {
      if (fpp[i]) {
        SPZ_HIP_TRY(hipMemcpyAsync(fb[i] + first * fpp[i], src[i] + first * fpp[i], count * fpp[i] * sizeof(float),
                                   hipMemcpyHostToDevice, pipe->up));
      }
    }
    const spz_amd_cloud_in d = {fb[0] + first * 3, fb[1] + first * 3, fb[2] + first * 4,
                                fb[3] + first,     fb[4] + first * 3, fb[5] + first * fpp[5]};
    return spz_amd_encode_shard_device(&d, first, count, n, sh_degree, antialiased, from_coord, version, k == 0 ? 1 : 0,
                                       sb, lay.total_bytes, pipe->up);
  };
  auto down = [&](int k) -> int {
    const uint64_t first = (uint64_t)k * cp, count = std::min<uint64_t>(cp, n - first);
    if (k == 0) SPZ_HIP_TRY(hipMemcpyAsync(h_stream, sb, 16, hipMemcpyDeviceToHost, pipe->down));
    for (int sec = 0; sec < SPZ_AMD_NUM_SECTIONS; ++sec) {
      const uint64_t off = lay.offset[sec] + first * lay.bytes_per_point[sec], len = count * lay.bytes_per_point[sec];
      if (len) SPZ_HIP_TRY(hipMemcpyAsync(h_stream + off, sb + off, len, hipMemcpyDeviceToHost, pipe->down));
    }
    return SPZ_AMD_OK;
  };
  if (zlib_session != nullptr && kept != nullptr) {
    // Section-major, for a caller whose container stage starts on the finished prefix of the stream (the stream is
    // attribute-major, so point-range chunks finish no section before the last of them): chunk 0 = the five small
    // arrays of ALL points (20 of the 20 + D bytes per point: the stream's first sections, final after one launch),
    // chunks 1.. = sh ranges, each extending the finished prefix.  After every launch the session is told how far the
    // stream is final; its table and match kernels then run beside the remaining uploads.
    const size_t sh_fpp = fpp[5];
    std::vector<uint64_t> sh_first;  // piece k covers points [sh_first[k], sh_first[k + 1])
    int sh_chunks = 0;
    uint64_t n_tail = 0, n_main = n;  // the last n_tail points' sh go first (see below); the pieces cover [0, n_main)
    if (sh_fpp) {
      // Pieces of ~240 MiB of floats (60 MB of stream, ~110 table segments), the last ones shrinking (each half of what
      // is left, down to 64 MiB).  A piece's table kernels take ~2 ms however few segments it has (one workgroup walks
      // its 512 KiB serially), so small pieces cost kernel time — the 160 MiB chunks of the plain pipeline, eleven
      // pieces of 78 segments for 10 M points, spent 50 ms on tables that take 19 ms over the whole input at once — and
      // large ones leave the container stage a large last piece to do when the upload ends.  profiles/README.md (round 3)
      // has the scan: every choice from 96 to 480 MiB is within 4 % now that the match kernel is short.
      static const size_t target = []() {
        const char *e = std::getenv("SPZ_AMD_HOST_SESSION_CHUNK_MIB");
        const long v = e ? std::atol(e) : 0;
        return (size_t)(v > 0 ? v : 240) << 20;
      }();
      static const size_t smallest = []() {
        const char *e = std::getenv("SPZ_AMD_HOST_SESSION_LAST_MIB");
        const long v = e ? std::atol(e) : 0;
        return (size_t)(v > 0 ? v : 64) << 20;
      }();
      const size_t point_bytes = sh_fpp * sizeof(float);
      // (a small cloud — under 1 GiB of floats, 20 ms of upload — is not cut finer than that: every feed is a round of
      // table kernels of ~2 ms however little it covers, and they follow one another)
      const bool small_cloud = n * (14 + sh_fpp) * sizeof(float) < (size_t(1) << 30);
      // a caller that wants the stream's END early (its container stage's serial tail job: the last tail_bytes of the
      // stream) gets the last points' sh as the first sh piece
      if (tail_ready != nullptr && tail_bytes > 0) {
        const uint64_t pts = ((tail_bytes + lay.bytes_per_point[SPZ_AMD_SEC_SH] - 1) / lay.bytes_per_point[SPZ_AMD_SEC_SH] + 1023ull) & ~1023ull;
        if (pts * 4 <= n) n_tail = pts;  // (a cloud this small has no use for it)
      }
      n_main = n - n_tail;
      uint64_t at = 0;
      while (at < n_main && (int)sh_first.size() < kPipeChunksMax - 5) {
        sh_first.push_back(at);
        const unsigned long long left = (n_main - at) * (unsigned long long)point_bytes;
        unsigned long long want = small_cloud ? target : std::min<unsigned long long>(target, std::max<unsigned long long>(smallest, left / 2));
        uint64_t count = (want / point_bytes + 1023ull) & ~1023ull;
        if (count >= n_main - at || (n_main - at - count) * (unsigned long long)point_bytes < smallest / 2) count = n_main - at;  // no sliver at the end
        at += count;
      }
      if (at < n_main) at = n_main;  // (more pieces than events: the last one takes the rest)
      sh_first.push_back(n_main);
      sh_chunks = (int)sh_first.size() - 1;
    }
    // The five small sections go up in three steps in the stream's order — positions; alphas, colours and scales;
    // rotations — so that the container stage starts after 120 MB of floats (2 ms) and not after all 560 MB (10 ms:
    // a fifth of the whole upload, during which the GPU had nothing to do and which it then lacked at the end).
    struct SmallStep {
      int n_arrays, arrays[5];  // spz_amd_cloud_in order: positions 0, scales 1, rotations 2, alphas 3, colors 4
      unsigned mask;
      int next_section;         // the stream is final up to this section's offset afterwards
    };
    static const SmallStep three_steps[3] = {
        {1, {0, 0, 0, 0, 0}, 1u << SPZ_AMD_SEC_POSITIONS, SPZ_AMD_SEC_ALPHAS},
        {3, {3, 4, 1, 0, 0}, (1u << SPZ_AMD_SEC_ALPHAS) | (1u << SPZ_AMD_SEC_COLORS) | (1u << SPZ_AMD_SEC_SCALES), SPZ_AMD_SEC_ROTATIONS},
        {1, {2, 0, 0, 0, 0}, 1u << SPZ_AMD_SEC_ROTATIONS, SPZ_AMD_SEC_SH}};
    static const SmallStep one_step[1] = {{5, {0, 1, 2, 3, 4}, 0x1fu, SPZ_AMD_SEC_SH}};
    // (for a cloud whose whole upload is a few milliseconds, one step: see the sh pieces above)
    const bool few_steps = n * (14 + sh_fpp) * sizeof(float) < (size_t(1) << 30);
    const SmallStep *const small_steps = few_steps ? one_step : three_steps;
    const int kSmallSteps = few_steps ? 1 : 3;
    const int tail_steps = n_tail ? 1 : 0, first_sh_step = kSmallSteps + tail_steps;
    if (first_sh_step + sh_chunks > pipe->n_events) return SPZ_AMD_ERR_INVALID_ARG;
    const uint64_t small_end = lay.offset[SPZ_AMD_SEC_SH];
    const uint64_t sh_bpp = lay.bytes_per_point[SPZ_AMD_SEC_SH];
    auto up2 = [&](int k) -> int {
      if (k < kSmallSteps) {
        const SmallStep &st = small_steps[k];
        for (int j = 0; j < st.n_arrays; ++j) {
          const int i = st.arrays[j];
          SPZ_HIP_TRY(hipMemcpyAsync(fb[i], src[i], n * fpp[i] * sizeof(float), hipMemcpyHostToDevice, pipe->up));
        }
        const spz_amd_cloud_in d = {fb[0], fb[1], fb[2], fb[3], fb[4], fb[5]};
        const int erc = spz_amd_encode_shard_sections_device(&d, 0, n, n, sh_degree, antialiased, from_coord, version, k == 0 ? 1 : 0, st.mask,
                                                             sb, lay.total_bytes, pipe->up);
        if (erc != SPZ_AMD_OK) return erc;
        const bool last = k + 1 == kSmallSteps && !sh_fpp;
        return spz_amd_zlib_session_feed(zlib_session, sb, last ? lay.total_bytes : lay.offset[st.next_section], pipe->up);
      }
      const bool tail_piece = k < first_sh_step;
      const uint64_t first = tail_piece ? n_main : sh_first[k - first_sh_step];
      const uint64_t count = tail_piece ? n_tail : sh_first[k - first_sh_step + 1] - first;
      SPZ_HIP_TRY(hipMemcpyAsync(fb[5] + first * sh_fpp, src[5] + first * sh_fpp, count * sh_fpp * sizeof(float), hipMemcpyHostToDevice,
                                 pipe->up));
      const spz_amd_cloud_in d = {fb[0] + first * 3, fb[1] + first * 3, fb[2] + first * 4, fb[3] + first, fb[4] + first * 3,
                                  fb[5] + first * sh_fpp};
      const int erc = spz_amd_encode_shard_sections_device(&d, first, count, n, sh_degree, antialiased, from_coord, version, 0, 0x20u, sb,
                                                           lay.total_bytes, pipe->up);
      if (erc != SPZ_AMD_OK) return erc;
      if (tail_piece) return SPZ_AMD_OK;  // not a prefix of the stream: the session gets it with its last feed
      // (the last points' sh, if they went first, are final by now as well: the last piece completes the stream)
      const uint64_t upto = first + count == n_main ? lay.total_bytes : small_end + (first + count) * sh_bpp;
      return spz_amd_zlib_session_feed(zlib_session, sb, upto, pipe->up);
    };
    auto down2 = [&](int k) -> int {
      if (k < kSmallSteps) {
        const uint64_t from = k == 0 ? 0 : lay.offset[small_steps[k - 1].next_section];  // the header with the first
        const uint64_t to = lay.offset[small_steps[k].next_section];
        if (to > from) SPZ_HIP_TRY(hipMemcpyAsync(h_stream + from, sb + from, to - from, hipMemcpyDeviceToHost, pipe->down));
        return SPZ_AMD_OK;
      }
      const bool tail_piece = k < first_sh_step;
      const uint64_t first = tail_piece ? n_main : sh_first[k - first_sh_step];
      const uint64_t count = tail_piece ? n_tail : sh_first[k - first_sh_step + 1] - first;
      const uint64_t off = small_end + first * sh_bpp, len = count * sh_bpp;
      if (len) SPZ_HIP_TRY(hipMemcpyAsync(h_stream + off, sb + off, len, hipMemcpyDeviceToHost, pipe->down));
      if (tail_piece) {  // the stream's last bytes are on the host: the caller's work on them starts now
        SPZ_HIP_TRY(hipStreamSynchronize(pipe->down));
        tail_ready(tail_arg);
      }
      return SPZ_AMD_OK;
    };
    rc = run_pipeline(pipe, device, first_sh_step + sh_chunks, up2, down2);
  } else {
    rc = run_pipeline(pipe, device, chunks, up, down);
  }
  if (rc == SPZ_AMD_OK && kept) {
    *d_keep = kept;
    kept_guard.p = nullptr;  // the caller's now: spz_amd_kept_stream_release
  }
  return rc;
}
}  // namespace

extern "C" {

int spz_amd_decode_host_ex(const uint8_t *h_stream, size_t size, uint64_t max_points, int to_coord,
                           const spz_amd_cloud_out *h, int device) {
  if (h_stream == nullptr || h == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  spz_amd_header hdr;
  int rc = spz_amd_peek_header_ex(h_stream, size, max_points, &hdr);
  if (rc != SPZ_AMD_OK) return rc;
  const uint64_t n = hdr.num_points;
  if (n == 0) return SPZ_AMD_OK;
  if (!valid_coord(to_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  spz_amd_layout lay;
  rc = spz_amd_stream_layout(n, hdr.sh_degree, (int)hdr.version, &lay);
  if (rc != SPZ_AMD_OK) return rc;
  const int sd = sh_dim_for_degree(hdr.sh_degree);
  if (!h->positions || !h->scales || !h->rotations || !h->alphas || !h->colors || (sd > 0 && !h->sh)) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  const size_t fpp[6] = {3, 3, 4, 1, 3, (size_t)sd * 3};
  float *dst[6] = {h->positions, h->scales, h->rotations, h->alphas, h->colors, h->sh};
  size_t total = Workspace::aligned(lay.total_bytes);
  for (int i = 0; i < 6; ++i) total += Workspace::aligned(n * fpp[i] * sizeof(float));
  Workspace ws;
  rc = ws.open(device, total);
  if (rc != SPZ_AMD_OK) return rc;
  uint8_t *sb = static_cast<uint8_t *>(ws.take(lay.total_bytes));
  float *fb[6];
  for (int i = 0; i < 6; ++i) fb[i] = static_cast<float *>(ws.take(n * fpp[i] * sizeof(float)));
  HostPipe *pipe = ws.pipe();
  uint64_t cp = 0;
  const int chunks = plan_chunks(n, (14 + (size_t)sd * 3) * sizeof(float), &cp);
  // chunk k: upload the six fragments of points [k cp, ...), decode them; the downloader copies the floats out
  auto up = [&](int k) -> int {
    const uint64_t first = (uint64_t)k * cp, count = std::min<uint64_t>(cp, n - first);
    for (int sec = 0; sec < SPZ_AMD_NUM_SECTIONS; ++sec) {
      const uint64_t off = lay.offset[sec] + first * lay.bytes_per_point[sec], len = count * lay.bytes_per_point[sec];
      if (len) SPZ_HIP_TRY(hipMemcpyAsync(sb + off, h_stream + off, len, hipMemcpyHostToDevice, pipe->up));
    }
    const spz_amd_cloud_out d = {fb[0] + first * 3, fb[1] + first * 3, fb[2] + first * 4,
                                 fb[3] + first,     fb[4] + first * 3, fb[5] + first * fpp[5]};
    return spz_amd_decode_shard_device(sb, lay.total_bytes, &hdr, first, count, to_coord, &d, pipe->up);
  };
  auto down = [&](int k) -> int {
    const uint64_t first = (uint64_t)k * cp, count = std::min<uint64_t>(cp, n - first);
    for (int i = 0; i < 6; ++i) {
      if (fpp[i]) {
        SPZ_HIP_TRY(hipMemcpyAsync(dst[i] + first * fpp[i], fb[i] + first * fpp[i], count * fpp[i] * sizeof(float),
                                   hipMemcpyDeviceToHost, pipe->down));
      }
    }
    return SPZ_AMD_OK;
  };
  return run_pipeline(pipe, device, chunks, up, down);
}

int spz_amd_decode_host_from_device(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr, int to_coord,
                                    const spz_amd_cloud_out *h, int device) {
  if (d_stream == nullptr || hdr == nullptr || h == nullptr || !valid_coord(to_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  const uint64_t n = hdr->num_points;
  spz_amd_layout lay;
  int rc = spz_amd_stream_layout(n, hdr->sh_degree, (int)hdr->version, &lay);
  if (rc != SPZ_AMD_OK) return rc;
  if (size < lay.total_bytes) return SPZ_AMD_ERR_SHORT_STREAM;
  if (n == 0) return SPZ_AMD_OK;
  const int sd = sh_dim_for_degree(hdr->sh_degree);
  if (!h->positions || !h->scales || !h->rotations || !h->alphas || !h->colors || (sd > 0 && !h->sh)) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  const size_t fpp[6] = {3, 3, 4, 1, 3, (size_t)sd * 3};
  float *dst[6] = {h->positions, h->scales, h->rotations, h->alphas, h->colors, h->sh};
  size_t total = 0;
  for (int i = 0; i < 6; ++i) total += Workspace::aligned(n * fpp[i] * sizeof(float));
  Workspace ws;
  rc = ws.open(device, total);
  if (rc != SPZ_AMD_OK) return rc;
  float *fb[6];
  for (int i = 0; i < 6; ++i) fb[i] = static_cast<float *>(ws.take(n * fpp[i] * sizeof(float)));
  HostPipe *pipe = ws.pipe();
  uint64_t cp = 0;
  const int chunks = plan_chunks(n, (14 + (size_t)sd * 3) * sizeof(float), &cp);
  // the stream is where it is: chunk k is a kernel launch, the downloader copies its floats out
  auto up = [&](int k) -> int {
    const uint64_t first = (uint64_t)k * cp, count = std::min<uint64_t>(cp, n - first);
    const spz_amd_cloud_out d = {fb[0] + first * 3, fb[1] + first * 3, fb[2] + first * 4,
                                 fb[3] + first,     fb[4] + first * 3, fb[5] + first * fpp[5]};
    return spz_amd_decode_shard_device(d_stream, size, hdr, first, count, to_coord, &d, pipe->up);
  };
  auto down = [&](int k) -> int {
    const uint64_t first = (uint64_t)k * cp, count = std::min<uint64_t>(cp, n - first);
    for (int i = 0; i < 6; ++i) {
      if (fpp[i]) {
        SPZ_HIP_TRY(hipMemcpyAsync(dst[i] + first * fpp[i], fb[i] + first * fpp[i], count * fpp[i] * sizeof(float),
                                   hipMemcpyDeviceToHost, pipe->down));
      }
    }
    return SPZ_AMD_OK;
  };
  return run_pipeline(pipe, device, chunks, up, down);
}

int spz_amd_decode_host(const uint8_t *h_stream, size_t size, int to_coord, const spz_amd_cloud_out *h,
                        int device) {
  return spz_amd_decode_host_ex(h_stream, size, SPZ_AMD_REFERENCE_MAX_POINTS, to_coord, h, device);
}

// The gather decode for host index lists and host outputs; the stream comes from the host (h_stream, uploaded) or lies on
// the device already (d_stream).
static int decode_gather_host_impl(const uint8_t *h_stream, const uint8_t *d_stream, size_t size, const spz_amd_header &hdr,
                                   const uint32_t *h_indices, uint64_t count, int to_coord, const spz_amd_cloud_out *h, int device) {
  spz_amd_layout lay;
  int rc = spz_amd_stream_layout(hdr.num_points, hdr.sh_degree, (int)hdr.version, &lay);
  if (rc != SPZ_AMD_OK) return rc;
  if (size < lay.total_bytes) return SPZ_AMD_ERR_SHORT_STREAM;
  if (count == 0) return SPZ_AMD_OK;
  const int sd = sh_dim_for_degree(hdr.sh_degree);
  if (hdr.num_points == 0 || h_indices == nullptr || !h->positions || !h->scales || !h->rotations || !h->alphas ||
      !h->colors || (sd > 0 && !h->sh)) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  for (uint64_t i = 0; i < count; ++i) {
    if (h_indices[i] >= hdr.num_points) return SPZ_AMD_ERR_INVALID_ARG;
  }
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  const size_t cnt[6] = {count * 3, count * 3, count * 4, count, count * 3, count * (size_t)sd * 3};
  float *dst[6] = {h->positions, h->scales, h->rotations, h->alphas, h->colors, h->sh};
  size_t total = (d_stream ? 0 : Workspace::aligned(lay.total_bytes)) + Workspace::aligned(count * sizeof(uint32_t));
  for (int i = 0; i < 6; ++i) total += Workspace::aligned(cnt[i] * sizeof(float));
  Workspace ws;
  rc = ws.open(device, total);
  if (rc != SPZ_AMD_OK) return rc;
  const void *sb = d_stream;
  if (d_stream == nullptr) {
    void *up = ws.take(lay.total_bytes);
    SPZ_HIP_TRY(hipMemcpyAsync(up, h_stream, lay.total_bytes, hipMemcpyHostToDevice, nullptr));
    sb = up;
  }
  void *ib = ws.take(count * sizeof(uint32_t));
  SPZ_HIP_TRY(hipMemcpyAsync(ib, h_indices, count * sizeof(uint32_t), hipMemcpyHostToDevice, nullptr));
  void *fb[6];
  for (int i = 0; i < 6; ++i) fb[i] = ws.take(cnt[i] * sizeof(float));
  spz_amd_cloud_out d = {(float *)fb[0], (float *)fb[1], (float *)fb[2], (float *)fb[3], (float *)fb[4], (float *)fb[5]};
  rc = spz_amd_decode_gather_device((const uint8_t *)sb, lay.total_bytes, &hdr, (const uint32_t *)ib, count, to_coord, &d,
                                    nullptr);
  if (rc != SPZ_AMD_OK) return rc;
  for (int i = 0; i < 6; ++i) {
    if (cnt[i]) SPZ_HIP_TRY(hipMemcpyAsync(dst[i], fb[i], cnt[i] * sizeof(float), hipMemcpyDeviceToHost, nullptr));
  }
  SPZ_HIP_TRY(hipStreamSynchronize(nullptr));
  return SPZ_AMD_OK;
}

int spz_amd_decode_gather_host(const uint8_t *h_stream, size_t size, uint64_t max_points, const uint32_t *h_indices,
                               uint64_t count, int to_coord, const spz_amd_cloud_out *h, int device) {
  if (h_stream == nullptr || h == nullptr || !valid_coord(to_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  spz_amd_header hdr;
  const int rc = spz_amd_peek_header_ex(h_stream, size, max_points, &hdr);
  if (rc != SPZ_AMD_OK) return rc;
  return decode_gather_host_impl(h_stream, nullptr, size, hdr, h_indices, count, to_coord, h, device);
}

int spz_amd_decode_gather_host_from_device(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr, const uint32_t *h_indices,
                                           uint64_t count, int to_coord, const spz_amd_cloud_out *h, int device) {
  if (d_stream == nullptr || hdr == nullptr || h == nullptr || !valid_coord(to_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  if (hdr->version < 1 || hdr->version > 3) return SPZ_AMD_ERR_VERSION;
  if (hdr->sh_degree > 3) return SPZ_AMD_ERR_SH_DEGREE;
  return decode_gather_host_impl(nullptr, d_stream, size, *hdr, h_indices, count, to_coord, h, device);
}

int spz_amd_convert_coordinates_host(float *h_positions, float *h_rotations, float *h_sh, uint64_t n,
                                     int sh_degree, int from_coord, int to_coord, int device) {
  const int sd = sh_dim_for_degree(sh_degree);
  if (sd < 0 || !valid_coord(from_coord) || !valid_coord(to_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  if (n == 0) return SPZ_AMD_OK;
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  const size_t fpp[3] = {3, 4, (size_t)sd * 3};  // floats per point: positions, rotations, sh
  float *hp[3] = {h_positions, h_rotations, h_sh};
  size_t total = 0, bytes_per_point = 0;
  for (int i = 0; i < 3; ++i) {
    if (!hp[i] || !fpp[i]) continue;
    total += Workspace::aligned(n * fpp[i] * sizeof(float));
    bytes_per_point += fpp[i] * sizeof(float);
  }
  if (total == 0) return SPZ_AMD_OK;
  Workspace ws;
  rc = ws.open(device, total);
  if (rc != SPZ_AMD_OK) return rc;
  float *b[3] = {nullptr, nullptr, nullptr};
  for (int i = 0; i < 3; ++i) {
    if (hp[i] && fpp[i]) b[i] = static_cast<float *>(ws.take(n * fpp[i] * sizeof(float)));
  }
  HostPipe *pipe = ws.pipe();
  uint64_t cp = 0;
  const int chunks = plan_chunks(n, bytes_per_point, &cp);
  // the same pipeline as encode / decode: chunk k+1 goes up and is flipped while chunk k-1 comes back (the pass is
  // element-wise and its sh pattern repeats per point, so any point range is a pass of its own)
  auto up = [&](int k) -> int {
    const uint64_t first = (uint64_t)k * cp, count = std::min<uint64_t>(cp, n - first);
    for (int i = 0; i < 3; ++i) {
      if (b[i]) {
        SPZ_HIP_TRY(hipMemcpyAsync(b[i] + first * fpp[i], hp[i] + first * fpp[i], count * fpp[i] * sizeof(float),
                                   hipMemcpyHostToDevice, pipe->up));
      }
    }
    return spz_amd_convert_coordinates_device(b[0] ? b[0] + first * 3 : nullptr, b[1] ? b[1] + first * 4 : nullptr,
                                              b[2] ? b[2] + first * fpp[2] : nullptr, count, sh_degree, from_coord, to_coord,
                                              pipe->up);
  };
  auto down = [&](int k) -> int {
    const uint64_t first = (uint64_t)k * cp, count = std::min<uint64_t>(cp, n - first);
    for (int i = 0; i < 3; ++i) {
      if (b[i]) {
        SPZ_HIP_TRY(hipMemcpyAsync(hp[i] + first * fpp[i], b[i] + first * fpp[i], count * fpp[i] * sizeof(float),
                                   hipMemcpyDeviceToHost, pipe->down));
      }
    }
    return SPZ_AMD_OK;
  };
  return run_pipeline(pipe, device, chunks, up, down);
}

}  // extern "C"
