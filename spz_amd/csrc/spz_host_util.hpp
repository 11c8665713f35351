// spz_host_util.hpp — output-buffer helpers of the C++ layer (spz_host.cpp, spz_ply.cpp, spz_inflate.cpp).
//
// The reference's API hands results back in std::vector (load-spz.h:69-100).  vector::resize
// value-initialises, which for a 10 M-point cloud is a full pass of zero stores and first-touch page
// faults over 650 MB (stream) or 2.36 GB (floats) that the device is about to overwrite — more time than
// the PCIe transfer itself.  resizeUninitialized() sets the size without writing the elements;
// Prefault maps the fresh pages from several threads while the first transfer is under way.
#pragma once

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <system_error>
#include <thread>
#include <type_traits>
#include <vector>

namespace spz {
namespace detail {

// v->resize(n) without value-initialising new elements, for trivial T.  libstdc++ keeps a vector as three
// pointers {begin, end, end of storage}; that is checked on the instance itself before `end` is moved,
// and anything unexpected (another standard library, a sanitizer build that tracks container bounds)
// takes the ordinary resize.
template <class T>
void resizeUninitialized(std::vector<T> *v, size_t n) {
  static_assert(std::is_trivial<T>::value, "only for trivial element types");
#if defined(__GLIBCXX__) && !defined(__SANITIZE_ADDRESS__)
  if (n > v->size() && (n - v->size()) * sizeof(T) >= (size_t(1) << 20)) {
    T *rep[3];
    if constexpr (sizeof(std::vector<T>) == sizeof(rep)) {  // not so with _GLIBCXX_DEBUG: ordinary resize then
      v->reserve(n);
      std::memcpy(rep, static_cast<const void *>(v), sizeof(rep));
      if (rep[0] == v->data() && rep[1] == v->data() + v->size() && rep[2] == v->data() + v->capacity() &&
          v->capacity() >= n) {
        rep[1] = rep[0] + n;
        std::memcpy(static_cast<void *>(v), rep, sizeof(rep));
        return;
      }
    }
  }
#endif
  v->resize(n);
}

// CPUs this process may actually use: the smaller of the online count, the scheduler affinity mask and the
// cgroup CPU quota (a container on a 256-thread host is often allowed 16: worker pools sized by
// hardware_concurrency() there only add contention, which is what "no scaling from 32 to 128 threads" was).
inline unsigned effectiveCpuCount() {
  static const unsigned n = []() {
    unsigned c = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) c = std::min<unsigned>(c, std::max(1, CPU_COUNT(&set)));
    auto quota = [](const char *path, bool v2) -> double {
      FILE *f = std::fopen(path, "r");
      if (!f) return 0.0;
      char a[64] = "", b[64] = "";
      const int got = std::fscanf(f, "%63s %63s", a, b);
      std::fclose(f);
      if (v2) return (got == 2 && std::strcmp(a, "max") != 0 && std::atof(b) > 0) ? std::atof(a) / std::atof(b) : 0.0;
      return got >= 1 ? std::atof(a) : 0.0;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", true);                       // cgroup v2: "<quota> <period>" or "max <period>"
    if (q <= 0.0) {
      const double us = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);  // cgroup v1
      const double period = quota("/sys/fs/cgroup/cpu/cpu.cfs_period_us", false);
      if (us > 0.0 && period > 0.0) q = us / period;
    }
    if (q > 0.0) c = std::min<unsigned>(c, std::max(1u, static_cast<unsigned>(q + 0.999)));
    return c;
  }();
  return n;
}

// Asks for transparent huge pages behind a freshly allocated buffer that is about to be written once (a no-op where
// THP is off or "always"): its first-touch faults then come 2 MiB at a time.
inline void adviseHugePages(void *p, size_t bytes) {
#ifdef MADV_HUGEPAGE
  const size_t huge = size_t(2) << 20;
  const uintptr_t lo = (reinterpret_cast<uintptr_t>(p) + huge - 1) & ~(huge - 1);
  const uintptr_t hi = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(huge - 1);
  if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
#else
  (void)p;
  (void)bytes;
#endif
}

inline int prefaultThreads() {
  static const int n = []() {
    const char *e = std::getenv("SPZ_AMD_PREFAULT_THREADS");
    if (e) return std::max(0, std::atoi(e));
    // measured on the GPU box's host (tools/pcie_probe.hip, 2.36 GB): 1 thread 28 GB/s, 4 threads 110 GB/s,
    // 16 and 32 threads 35 GB/s (they contend in the page allocator)
    return static_cast<int>(std::min<unsigned>(4u, effectiveCpuCount()));
  }();
  return n;
}

// The outputs are mapped BEFORE the device call starts (default).  Mapping them beside it (SPZ_AMD_PREFAULT_CONCURRENT=1)
// looked better on paper and measured worse: the copies' own pinning of source and destination pages and the
// populate threads contend in the kernel's mm (10 M SH3 unpack: 72-79 ms mapped first, 119 ms side by side; even
// beside the UPLOAD only, the downloads waiting for the mapping to finish, 116 ms).
inline bool prefaultJoinFirst() {
  static const bool v = []() {
    const char *e = std::getenv("SPZ_AMD_PREFAULT_CONCURRENT");
    return !(e && std::atoi(e) != 0);
  }();
  return v;
}

// Maps the pages of freshly allocated buffers from a few threads; joins in the destructor.
//  * MADV_HUGEPAGE first: with transparent huge pages in "madvise" mode (this image, the GPU boxes) the range
//    is then backed by 2 MiB pages — 1 200 faults instead of 576 000 for the 2.36 GB of a 10 M-point cloud.
//    Measured on the GPU box's host: 4 KiB pages map at 16-23 GB/s whatever the thread count and a device
//    copy into untouched pages runs at 13-16 GB/s against 56 GB/s into mapped ones, which made the first
//    touch of the outputs cost more than the PCIe transfer; huge pages map at 110 GB/s.
//  * MADV_POPULATE_WRITE then maps them "as if written", contents untouched (so it could also run while a
//    device-to-host copy is landing in them).  Where the kernel lacks the advice nothing is done and the
//    pages fault on first use.
//  * The buffers are cut into 32 MiB segments, taken by the threads in the order of their relative position
//    (all buffers advance front to back together: the order in which the chunks of the device pipeline arrive).
class Prefault {
 public:
  Prefault() = default;
  Prefault(const Prefault &) = delete;
  Prefault &operator=(const Prefault &) = delete;
  ~Prefault() { join(); }
  void add(void *p, size_t bytes) {
#if defined(MADV_POPULATE_WRITE) && defined(MADV_HUGEPAGE)
    if (prefaultThreads() <= 0 || bytes < (size_t(32) << 20)) return;
    const size_t huge = size_t(2) << 20;
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(p) + huge - 1) & ~(huge - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(huge - 1);
    if (hi <= lo) return;
    (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    const size_t seg = size_t(32) << 20;
    for (uintptr_t a = lo; a < hi; a += seg) {
      segs_.push_back({a, std::min<size_t>(seg, hi - a), static_cast<double>(a - lo) / static_cast<double>(hi - lo)});
    }
#else
    (void)p;
    (void)bytes;
#endif
  }
  void start(bool join_at_start = true) {
    if (segs_.empty()) return;
    std::sort(segs_.begin(), segs_.end(), [](const Seg &x, const Seg &y) { return x.where < y.where; });
    const int threads = std::min<int>(prefaultThreads(), static_cast<int>(segs_.size()));
    try {
      for (int t = 0; t < threads; ++t) {
        pool_.emplace_back([this]() {
          for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= segs_.size()) return;
#ifdef MADV_POPULATE_WRITE
            (void)madvise(reinterpret_cast<void *>(segs_[i].addr), segs_[i].len, MADV_POPULATE_WRITE);
#endif
          }
        });
      }
    } catch (const std::system_error &) {
      // no thread to be had: whatever is not mapped by the threads that did start faults in on first use
    }
    if (join_at_start && prefaultJoinFirst()) join();
  }
  // start(), but never waits: for mapping beside device work that copies nothing (kernels), where the two do not contend
  void startBeside() { start(false); }
  void join() {
    for (auto &t : pool_) t.join();
    pool_.clear();
  }

 private:
  struct Seg {
    uintptr_t addr;
    size_t len;
    double where;  // relative position inside its buffer
  };
  std::vector<Seg> segs_;
  std::vector<std::thread> pool_;
  std::atomic<size_t> next_{0};
};

// SPZ_AMD_FILE_IO_THREADS: threads that read a large file (0 = the reference's single stream read always; default 8, at
// most the usable CPUs).
inline unsigned fileIoThreads() {
  static const unsigned n = []() {
    const char *e = std::getenv("SPZ_AMD_FILE_IO_THREADS");
    const long v = e ? std::atol(e) : 8;
    return static_cast<unsigned>(std::max(0l, std::min(v, static_cast<long>(std::max(1u, effectiveCpuCount())))));
  }();
  return n;
}
constexpr size_t kParallelIoMin = size_t(16) << 20, kIoPiece = size_t(8) << 20;

// `size` bytes of the file from `file_offset` on into buf, in 8 MiB pieces by several threads (pread).  false: a read
// failed or the file is shorter.
inline bool parallelRead(int fd, uint8_t *buf, size_t size, size_t file_offset = 0) {
  const size_t pieces = (size + kIoPiece - 1) / kIoPiece;
  const unsigned threads = static_cast<unsigned>(std::min<size_t>(std::max(1u, fileIoThreads()), pieces));
  std::atomic<size_t> next{0};
  std::atomic<bool> ok{true};
  auto work = [&]() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= pieces || !ok.load()) return;
      size_t off = i * kIoPiece;
      const size_t end = std::min(size, off + kIoPiece);
      while (off < end) {
        const ssize_t n = ::pread(fd, buf + off, end - off, static_cast<off_t>(file_offset + off));
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) {  // an error, or a file that is (or has become) shorter
          ok.store(false);
          return;
        }
        off += static_cast<size_t>(n);
      }
    }
  };
  std::vector<std::thread> pool;
  try {
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work);
  } catch (const std::system_error &) {
    // fewer threads than asked for: this one does what they would have
  }
  work();
  for (auto &t : pool) t.join();
  return ok.load();
}

}  // namespace detail
}  // namespace spz
