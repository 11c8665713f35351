// pcie_probe — what the host-pointer path can count on from this box: how long hipMemcpyAsync takes to
// RETURN for pageable memory, copy rates for pageable / registered / hipHostMalloc memory in both
// directions, whether two host threads get H2D and D2H to overlap, what first-touching the output pages
// costs (memset, MADV_POPULATE_WRITE with 1..32 threads, DMA into untouched pages), hipHostRegister cost.
//   hipcc -O2 --offload-arch=gfx950 -o build/pcie_probe tools/pcie_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(2); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static void *fresh(size_t bytes) {  // untouched anonymous pages, like a large malloc
  void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) { perror("mmap"); exit(2); }
  return p;
}
static double populate(void *p, size_t bytes, int threads) {
  const double t0 = now();
  std::vector<std::thread> pool;
  const size_t per = ((bytes / threads) + 4095) & ~size_t(4095);
  for (int t = 0; t < threads; ++t) {
    const size_t off = per * t;
    if (off >= bytes) break;
    const size_t len = std::min(per, bytes - off);
    pool.emplace_back([=]() {
      if (madvise((char *)p + off, len, MADV_POPULATE_WRITE) != 0) {
        for (size_t i = 0; i < len; i += 4096) ((volatile char *)p)[off + i] = 0;
      }
    });
  }
  for (auto &t : pool) t.join();
  return now() - t0;
}

int main(int argc, char **argv) {
  const size_t big = (argc > 1 ? atoll(argv[1]) : 2360) * 1000000ull;   // float side of 10 M SH3
  const size_t small = (argc > 2 ? atoll(argv[2]) : 650) * 1000000ull;  // stream side
  CK(hipSetDevice(0));
  void *d_big, *d_small;
  CK(hipMalloc(&d_big, big));
  CK(hipMalloc(&d_small, small));
  CK(hipMemset(d_big, 1, big));
  CK(hipMemset(d_small, 2, small));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  {
    FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char buf[128] = "?";
    if (f) { if (!fgets(buf, sizeof buf, f)) buf[0] = 0; fclose(f); }
    printf("{\"what\": \"thp\", \"enabled\": \"%.*s\", \"cores\": %u}\n", (int)strcspn(buf, "\n"), buf, std::thread::hardware_concurrency());
  }
  // 1. pageable (already touched) H2D / D2H: call-return time vs completion time
  char *h_big = (char *)fresh(big), *h_small = (char *)fresh(small);
  double t0 = now();
  memset(h_big, 3, big);
  printf("{\"what\": \"memset_first_touch\", \"GB\": %.2f, \"s\": %.4f}\n", big / 1e9, now() - t0);
  memset(h_small, 4, small);
  for (int rep = 0; rep < 2; ++rep) {
    t0 = now();
    CK(hipMemcpyAsync(d_big, h_big, big, hipMemcpyHostToDevice, s1));
    const double t_ret = now() - t0;
    CK(hipStreamSynchronize(s1));
    const double t_all = now() - t0;
    printf("{\"what\": \"pageable_h2d\", \"rep\": %d, \"GB\": %.2f, \"call_returns_s\": %.4f, \"done_s\": %.4f, \"GBps\": %.1f}\n", rep, big / 1e9, t_ret, t_all, big / 1e9 / t_all);
    t0 = now();
    CK(hipMemcpyAsync(h_small, d_small, small, hipMemcpyDeviceToHost, s2));
    const double r_ret = now() - t0;
    CK(hipStreamSynchronize(s2));
    const double r_all = now() - t0;
    printf("{\"what\": \"pageable_d2h\", \"rep\": %d, \"GB\": %.2f, \"call_returns_s\": %.4f, \"done_s\": %.4f, \"GBps\": %.1f}\n", rep, small / 1e9, r_ret, r_all, small / 1e9 / r_all);
  }
  // 2. both directions at once from two host threads, pageable, in 8 chunks each
  for (int chunks : {1, 8, 32}) {
    t0 = now();
    std::thread up([&]() {
      CK(hipSetDevice(0));
      for (int k = 0; k < chunks; ++k) CK(hipMemcpyAsync((char *)d_big + big / chunks * k, h_big + big / chunks * k, big / chunks, hipMemcpyHostToDevice, s1));
      CK(hipStreamSynchronize(s1));
    });
    std::thread down([&]() {
      CK(hipSetDevice(0));
      for (int k = 0; k < chunks; ++k) CK(hipMemcpyAsync(h_small + small / chunks * k, (char *)d_small + small / chunks * k, small / chunks, hipMemcpyDeviceToHost, s2));
      CK(hipStreamSynchronize(s2));
    });
    up.join();
    down.join();
    const double t = now() - t0;
    printf("{\"what\": \"pageable_duplex_two_threads\", \"chunks\": %d, \"s\": %.4f, \"GBps_sum\": %.1f}\n", chunks, t, (big + small) / 1e9 / t);
  }
  // 3. same from ONE thread (is the async call really asynchronous for pageable memory?)
  t0 = now();
  for (int k = 0; k < 8; ++k) {
    CK(hipMemcpyAsync((char *)d_big + big / 8 * k, h_big + big / 8 * k, big / 8, hipMemcpyHostToDevice, s1));
    CK(hipMemcpyAsync(h_small + small / 8 * k, (char *)d_small + small / 8 * k, small / 8, hipMemcpyDeviceToHost, s2));
  }
  const double t_issue = now() - t0;
  CK(hipStreamSynchronize(s1));
  CK(hipStreamSynchronize(s2));
  printf("{\"what\": \"pageable_duplex_one_thread\", \"issue_s\": %.4f, \"s\": %.4f}\n", t_issue, now() - t0);
  // 4. registered in place
  t0 = now();
  CK(hipHostRegister(h_big, big, hipHostRegisterDefault));
  const double t_reg = now() - t0;
  t0 = now();
  CK(hipHostRegister(h_small, small, hipHostRegisterDefault));
  const double t_reg2 = now() - t0;
  printf("{\"what\": \"hipHostRegister\", \"GB\": %.2f, \"s\": %.4f, \"GBps\": %.1f, \"small_s\": %.4f}\n", big / 1e9, t_reg, big / 1e9 / t_reg, t_reg2);
  t0 = now();
  CK(hipMemcpyAsync(d_big, h_big, big, hipMemcpyHostToDevice, s1));
  const double q_ret = now() - t0;
  CK(hipMemcpyAsync(h_small, d_small, small, hipMemcpyDeviceToHost, s2));
  CK(hipStreamSynchronize(s1));
  const double q_up = now() - t0;
  CK(hipStreamSynchronize(s2));
  printf("{\"what\": \"registered_duplex_one_thread\", \"call_returns_s\": %.5f, \"h2d_done_s\": %.4f, \"GBps_h2d\": %.1f, \"all_done_s\": %.4f}\n", q_ret, q_up, big / 1e9 / q_up, now() - t0);
  t0 = now();
  CK(hipHostUnregister(h_big));
  CK(hipHostUnregister(h_small));
  printf("{\"what\": \"hipHostUnregister\", \"s\": %.4f}\n", now() - t0);
  munmap(h_big, big);
  // 5. first touch of the OUTPUT pages (what a fresh std::vector costs): stream-sized and float-sized
  for (int threads : {1, 4, 16, 32}) {
    void *p = fresh(big);
    const double t = populate(p, big, threads);
    printf("{\"what\": \"populate_write\", \"GB\": %.2f, \"threads\": %d, \"s\": %.4f, \"GBps\": %.1f}\n", big / 1e9, threads, t, big / 1e9 / t);
    munmap(p, big);
  }
#ifndef MADV_HUGEPAGE
#define MADV_HUGEPAGE 14
#endif
  for (int threads : {1, 4, 16, 32}) {   // the same with transparent huge pages asked for (THP mode "madvise")
    void *p = fresh(big + (size_t(2) << 20));
    char *q = (char *)(((uintptr_t)p + (size_t(2) << 20) - 1) & ~((uintptr_t)(size_t(2) << 20) - 1));
    const size_t len = big & ~((size_t(2) << 20) - 1);
    const int adv = madvise(q, len, MADV_HUGEPAGE);
    const double t = populate(q, len, threads);
    printf("{\"what\": \"populate_write_hugepage\", \"madvise_rc\": %d, \"GB\": %.2f, \"threads\": %d, \"s\": %.4f, \"GBps\": %.1f}\n", adv, len / 1e9, threads, t, len / 1e9 / t);
    if (threads == 16) {
      t0 = now();
      CK(hipMemcpy(q, d_big, len, hipMemcpyDeviceToHost));
      const double t2 = now() - t0;
      printf("{\"what\": \"d2h_into_populated_hugepages\", \"GB\": %.2f, \"s\": %.4f, \"GBps\": %.1f}\n", len / 1e9, t2, len / 1e9 / t2);
    }
    munmap(p, big + (size_t(2) << 20));
  }
  {
    void *p = fresh(big + (size_t(2) << 20));   // DMA into untouched pages of a huge-page-advised range
    char *q = (char *)(((uintptr_t)p + (size_t(2) << 20) - 1) & ~((uintptr_t)(size_t(2) << 20) - 1));
    const size_t len = big & ~((size_t(2) << 20) - 1);
    (void)madvise(q, len, MADV_HUGEPAGE);
    t0 = now();
    CK(hipMemcpy(q, d_big, len, hipMemcpyDeviceToHost));
    const double t = now() - t0;
    printf("{\"what\": \"d2h_into_untouched_hugepage_range\", \"GB\": %.2f, \"s\": %.4f, \"GBps\": %.1f}\n", len / 1e9, t, len / 1e9 / t);
    munmap(p, big + (size_t(2) << 20));
  }
  {
    void *p = fresh(big);   // DMA straight into untouched pageable memory
    t0 = now();
    CK(hipMemcpy(p, d_big, big, hipMemcpyDeviceToHost));
    const double t = now() - t0;
    printf("{\"what\": \"d2h_into_untouched_pages\", \"GB\": %.2f, \"s\": %.4f, \"GBps\": %.1f}\n", big / 1e9, t, big / 1e9 / t);
    t0 = now();
    CK(hipMemcpy(p, d_big, big, hipMemcpyDeviceToHost));
    const double t2 = now() - t0;
    printf("{\"what\": \"d2h_into_touched_pages\", \"GB\": %.2f, \"s\": %.4f, \"GBps\": %.1f}\n", big / 1e9, t2, big / 1e9 / t2);
    munmap(p, big);
  }
  {
    void *p = fresh(big);   // populate with 16 threads WHILE the stream is uploaded, then D2H in 8 chunks
    t0 = now();
    std::thread pop([&]() { populate(p, big, 16); });
    CK(hipMemcpy(d_small, h_small, small, hipMemcpyHostToDevice));
    pop.join();
    const double t_pop = now() - t0;
    for (int k = 0; k < 8; ++k) CK(hipMemcpyAsync((char *)p + big / 8 * k, (char *)d_big + big / 8 * k, big / 8, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s2));
    printf("{\"what\": \"load_shape: populate||h2d(stream) then d2h(floats)\", \"populate_and_h2d_s\": %.4f, \"total_s\": %.4f}\n", t_pop, now() - t0);
    munmap(p, big);
  }
  // 6. hipHostMalloc staging: alloc cost + rate
  t0 = now();
  void *pin;
  CK(hipHostMalloc(&pin, size_t(256) << 20, hipHostMallocDefault));
  printf("{\"what\": \"hipHostMalloc_256MiB\", \"s\": %.4f}\n", now() - t0);
  t0 = now();
  for (int k = 0; k < 8; ++k) CK(hipMemcpyAsync(pin, d_big, size_t(256) << 20, hipMemcpyDeviceToHost, s2));
  CK(hipStreamSynchronize(s2));
  printf("{\"what\": \"pinned_d2h\", \"GBps\": %.1f}\n", 8 * 0.268435456 / (now() - t0));
  t0 = now();
  CK(hipMalloc(&pin, size_t(3) << 30));
  const double t_m = now() - t0;
  t0 = now();
  CK(hipFree(pin));
  printf("{\"what\": \"hipMalloc_3GiB\", \"malloc_s\": %.4f, \"free_s\": %.4f}\n", t_m, now() - t0);
  return 0;
}
