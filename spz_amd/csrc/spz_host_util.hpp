// spz_host_util.hpp — output-buffer helpers of the C++ layer (spz_host.cpp, spz_ply.cpp, spz_inflate.cpp).
//
// The reference's API hands results back in std::vector (load-spz.h:69-100).  vector::resize
// value-initialises, which for a 10 M-point cloud is a full pass of zero stores and first-touch page
// faults over 650 MB (stream) or 2.36 GB (floats) that the device is about to overwrite — more time than
// the PCIe transfer itself.  resizeUninitialized() sets the size without writing the elements;
// Prefault maps the fresh pages from several threads while the first transfer is under way.
#pragma once

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
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
    v->reserve(n);
    T *rep[3];
    static_assert(sizeof(std::vector<T>) == sizeof(rep), "unexpected std::vector representation");
    std::memcpy(rep, static_cast<const void *>(v), sizeof(rep));
    if (rep[0] == v->data() && rep[1] == v->data() + v->size() && rep[2] == v->data() + v->capacity() &&
        v->capacity() >= n) {
      rep[1] = rep[0] + n;
      std::memcpy(static_cast<void *>(v), rep, sizeof(rep));
      return;
    }
  }
#endif
  v->resize(n);
}

inline int prefaultThreads() {
  static const int n = []() {
    const char *e = std::getenv("SPZ_AMD_PREFAULT_THREADS");
    if (e) return std::max(0, std::atoi(e));
    return static_cast<int>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency() / 2)));
  }();
  return n;
}

// Maps the pages of freshly allocated buffers from a few threads; joins in the destructor.
//  * MADV_HUGEPAGE first: with transparent huge pages in "madvise" mode (this image, the GPU boxes) the range
//    is then backed by 2 MiB pages — 1 200 faults instead of 576 000 for the 2.36 GB of a 10 M-point cloud.
//    Measured on the GPU box's host: 4 KiB pages map at 16-23 GB/s whatever the thread count (and a device
//    copy into untouched pages runs at 13 GB/s against 56 GB/s into mapped ones), which made the first
//    touch of the outputs cost more than the PCIe transfer.
//  * MADV_POPULATE_WRITE then maps them "as if written", contents untouched, so it may run while a
//    device-to-host copy is already landing in them.  Where the kernel lacks the advice nothing is done and
//    the pages fault on first use.
class Prefault {
 public:
  Prefault() = default;
  Prefault(const Prefault &) = delete;
  Prefault &operator=(const Prefault &) = delete;
  ~Prefault() { join(); }
  void add(void *p, size_t bytes) {
#if defined(MADV_POPULATE_WRITE) && defined(MADV_HUGEPAGE)
    const int threads = prefaultThreads();
    if (threads <= 0 || bytes < (size_t(32) << 20)) return;
    const size_t huge = size_t(2) << 20;
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(p) + huge - 1) & ~(huge - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(huge - 1);
    if (hi <= lo) return;
    (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    const size_t pieces = std::min<size_t>(static_cast<size_t>(threads), (hi - lo) / (size_t(16) << 20) + 1);
    const size_t per = (((hi - lo) / pieces) + huge - 1) & ~(huge - 1);
    for (uintptr_t a = lo; a < hi; a += per) {
      const size_t len = std::min<size_t>(per, hi - a);
      pool_.emplace_back([a, len]() { (void)madvise(reinterpret_cast<void *>(a), len, MADV_POPULATE_WRITE); });
    }
#else
    (void)p;
    (void)bytes;
#endif
  }
  void join() {
    for (auto &t : pool_) t.join();
    pool_.clear();
  }

 private:
  std::vector<std::thread> pool_;
};

}  // namespace detail
}  // namespace spz
