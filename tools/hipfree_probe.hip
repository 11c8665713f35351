// Does hipFree give device memory back (as hipMemGetInfo sees it)?  Plain malloc/free, then with a kernel that used the
// block, then with a copy from pageable host memory into it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(char *p, size_t n) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = 1; }
static size_t free_mib() { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); return f >> 20; }
int main() {
  const size_t n = 900ull << 20;
  std::printf("start %zu\n", free_mib());
  for (int mode = 0; mode < 4; ++mode) {
    for (int k = 0; k < 3; ++k) {
      char *p = nullptr;
      CHECK(hipMalloc(reinterpret_cast<void **>(&p), n));
      if (mode >= 1) { hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, nullptr, p, n); CHECK(hipDeviceSynchronize()); }
      if (mode >= 2) { std::vector<char> h(64 << 20, 3); CHECK(hipMemcpy(p, h.data(), h.size(), hipMemcpyHostToDevice)); }
      if (mode >= 3) { hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); std::vector<char> h(64 << 20);
                       CHECK(hipMemcpyAsync(h.data(), p, h.size(), hipMemcpyDeviceToHost, s)); CHECK(hipStreamSynchronize(s)); CHECK(hipStreamDestroy(s)); }
      const size_t held = free_mib();
      CHECK(hipFree(p));
      std::printf("mode %d call %d: free while held %zu, after hipFree %zu\n", mode, k, held, free_mib());
    }
  }
  return 0;
}
