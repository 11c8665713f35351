// placement_scan.hip — what decides whether an sh3 decode runs at 0.47 ms or at 0.57 ms?  
//
// One slab per "kind" of device memory, the decode's seven buffers (stream + six float arrays) laid out inside it at
// offsets the experiment chooses, the library's own kernels timed with HIP events.  Experiments (one JSON line each
// config, both tile orders):
//   repeat   the packed layout N times                      -> noise floor of one placement
//   stream   the stream moved in steps of 2 MiB ... 1 GiB   -> relative offset read side / write side
//   sh       the 1.8 GB sh array moved likewise             -> relative offset of the largest write stream
//   shuffle  (vmm kinds) the slab's 2 MiB granules re-mapped in another random order, the layout untouched
// Kinds: malloc (hipMalloc), vmm1 (ONE hipMemCreate handle mapped whole), vmmN (one handle per granule, mapped in order),
// vmmS (one handle per granule, mapped in a random order).
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude -o build/placement_scan tools/placement_scan.hip -Lspz_amd/lib -lspz_amd -Wl,-rpath,$PWD/spz_amd/lib
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "spz_amd.h"

#define CHECK(x)                                                                             \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                          \
    }                                                                                        \
  } while (0)

namespace {

constexpr uint64_t kN = 10'000'000;
constexpr int kDeg = 3;
constexpr uint64_t MiB = 1ull << 20, GiB = 1ull << 30;

__global__ void fill_kernel(float *p, uint64_t n, float lo, float hi, uint32_t seed) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = lo + (hi - lo) * ((x >> 8) * (1.0f / 16777216.0f));
  }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void write_kernel(u32x4 *p, uint64_t n16) {
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) __builtin_nontemporal_store(v, p + i);
}
__global__ __launch_bounds__(256) void read_kernel(const u32x4 *p, uint64_t n16, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
    const u32x4 v = __builtin_nontemporal_load(p + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

struct Slab {
  std::string kind;
  uint8_t *base = nullptr;
  uint64_t bytes = 0, gran = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
  bool vmm = false;
};

hipMemAllocationProp vmm_prop() {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  return prop;
}

void map_handles(Slab &s, const std::vector<uint32_t> &order) {
  const uint64_t per = s.bytes / s.handles.size();
  for (size_t i = 0; i < order.size(); ++i) CHECK(hipMemMap(s.base + i * per, per, 0, s.handles[order[i]], 0));
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = 0;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CHECK(hipMemSetAccess(s.base, s.bytes, &acc, 1));
}

Slab make_slab(const std::string &kind, uint64_t bytes, std::mt19937_64 &rng) {
  Slab s;
  s.kind = kind;
  s.bytes = bytes;
  if (kind == "malloc") {
    CHECK(hipMalloc(reinterpret_cast<void **>(&s.base), bytes));
    return s;
  }
  s.vmm = true;
  hipMemAllocationProp prop = vmm_prop();
  size_t gran = 0;
  CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  s.gran = std::max<uint64_t>(gran, 2 * MiB);
  CHECK(hipMemAddressReserve(reinterpret_cast<void **>(&s.base), bytes, s.gran, nullptr, 0));
  const uint64_t per = kind == "vmm1" ? bytes : s.gran;
  s.handles.resize(bytes / per);
  for (auto &h : s.handles) CHECK(hipMemCreate(&h, per, &prop, 0));
  std::vector<uint32_t> order(s.handles.size());
  for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
  if (kind == "vmmS") std::shuffle(order.begin(), order.end(), rng);
  map_handles(s, order);
  return s;
}

void reshuffle(Slab &s, std::mt19937_64 &rng) {
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemUnmap(s.base, s.bytes));
  std::vector<uint32_t> order(s.handles.size());
  for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
  std::shuffle(order.begin(), order.end(), rng);
  map_handles(s, order);
}

void free_slab(Slab &s) {
  CHECK(hipDeviceSynchronize());
  if (!s.vmm) {
    CHECK(hipFree(s.base));
    return;
  }
  CHECK(hipMemUnmap(s.base, s.bytes));
  for (auto h : s.handles) CHECK(hipMemRelease(h));
  CHECK(hipMemAddressFree(s.base, s.bytes));
}

struct Layout {
  uint64_t off[7];  // positions, scales, rotations, alphas, colors, sh, stream
};
const uint64_t kFloats[6] = {3 * kN, 3 * kN, 4 * kN, kN, 3 * kN, 45 * kN};

uint64_t up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

Layout packed_layout(uint64_t stream_bytes, uint64_t align) {
  Layout l;
  uint64_t at = 0;
  for (int k = 0; k < 6; ++k) {
    l.off[k] = at;
    at = up(at + kFloats[k] * 4, align);
  }
  l.off[6] = at;
  (void)stream_bytes;
  return l;
}

uint64_t base_rest_bytes();

struct Timing {
  float dec_seq, dec_il, enc;
};

spz_amd_cloud_in g_in;
uint64_t g_stream_bytes;

uint64_t base_rest_bytes() {  // positions, scales, rotations, alphas, colours and the stream, each begun at a 2 MiB multiple
  uint64_t at = 0;
  for (int k = 0; k < 7; ++k) {
    if (k == 5) continue;
    at = up(at + (k < 6 ? kFloats[k] * 4 : g_stream_bytes), 2 * MiB);
  }
  return at;
}

Timing measure_ptrs(uint8_t *const ptr[7], int reps);

Timing measure(const Slab &s, const Layout &l, int reps) {
  uint8_t *ptr[7];
  for (int k = 0; k < 7; ++k) {
    const uint64_t len = k < 6 ? kFloats[k] * 4 : g_stream_bytes;
    if (l.off[k] + len > s.bytes) {
      std::fprintf(stderr, "layout leaves the slab\n");
      std::exit(2);
    }
    ptr[k] = s.base + l.off[k];
  }
  return measure_ptrs(ptr, reps);
}

Timing measure_ptrs(uint8_t *const ptr[7], int reps) {
  spz_amd_cloud_out out;
  out.positions = reinterpret_cast<float *>(ptr[0]);
  out.scales = reinterpret_cast<float *>(ptr[1]);
  out.rotations = reinterpret_cast<float *>(ptr[2]);
  out.alphas = reinterpret_cast<float *>(ptr[3]);
  out.colors = reinterpret_cast<float *>(ptr[4]);
  out.sh = reinterpret_cast<float *>(ptr[5]);
  uint8_t *stream = ptr[6];
  spz_amd_header hdr = {3, (uint32_t)kN, (uint8_t)kDeg, 12, 0, 0};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  Timing t = {};
  auto med = [&](std::vector<float> &v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
  };
  std::vector<float> ve;
  for (int r = 0; r < 3; ++r) {
    CHECK(hipEventRecord(e0, nullptr));
    if (spz_amd_encode_device(&g_in, kN, kDeg, 0, SPZ_AMD_RDF, 3, stream, g_stream_bytes, nullptr) != SPZ_AMD_OK) std::exit(3);
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ve.push_back(ms);
  }
  t.enc = med(ve);
  for (int order = 0; order < 2; ++order) {
    setenv("SPZ_AMD_GRID_ORDER", order ? "interleaved" : "sequential", 1);
    std::vector<float> v;
    for (int r = 0; r < reps + 2; ++r) {
      CHECK(hipEventRecord(e0, nullptr));
      if (spz_amd_decode_device(stream, g_stream_bytes, &hdr, SPZ_AMD_RDF, &out, nullptr) != SPZ_AMD_OK) std::exit(3);
      CHECK(hipEventRecord(e1, nullptr));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) v.push_back(ms);
    }
    (order ? t.dec_il : t.dec_seq) = med(v);
  }
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return t;
}

void report(const Slab &s, const char *exp, uint64_t param, const Layout &l, const Timing &t) {
  std::printf("{\"kind\": \"%s\", \"exp\": \"%s\", \"param_mib\": %.3f, \"base\": \"%p\", \"stream_off_mib\": %.3f, \"sh_off_mib\": %.3f, "
              "\"dec_seq_ms\": %.4f, \"dec_il_ms\": %.4f, \"enc_ms\": %.4f}\n",
              s.kind.c_str(), exp, (double)param / MiB, (void *)s.base, (double)l.off[6] / MiB, (double)l.off[5] / MiB, t.dec_seq, t.dec_il,
              t.enc);
  std::fflush(stdout);
}

}  // namespace

int main(int argc, char **argv) {
  const std::string kinds_arg = argc > 1 ? argv[1] : "malloc,vmm1,vmmN,vmmS";
  const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
  const uint64_t seed = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1;
  std::mt19937_64 rng(seed);
  CHECK(hipSetDevice(0));
  spz_amd_layout lay;
  if (spz_amd_stream_layout(kN, kDeg, 3, &lay) != SPZ_AMD_OK) return 3;
  g_stream_bytes = lay.total_bytes;
  // the input cloud: its own allocations (only the encode reads them)
  float *in[6];
  const float lo[6] = {-10.f, -8.f, -1.f, -6.f, -2.f, -0.5f}, hi[6] = {10.f, 0.f, 1.f, 6.f, 2.f, 0.5f};
  for (int k = 0; k < 6; ++k) {
    CHECK(hipMalloc(reinterpret_cast<void **>(&in[k]), kFloats[k] * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, nullptr, in[k], kFloats[k], lo[k], hi[k], 0x9e3779b9u * (k + 1));
  }
  CHECK(hipDeviceSynchronize());
  g_in = {in[0], in[1], in[2], in[3], in[4], in[5]};

  const uint64_t slab_bytes = 8 * GiB;
  std::vector<std::string> kinds;
  for (size_t p = 0; p < kinds_arg.size();) {
    const size_t q = kinds_arg.find(',', p);
    kinds.push_back(kinds_arg.substr(p, q == std::string::npos ? q : q - p));
    if (q == std::string::npos) break;
    p = q + 1;
  }
  for (const std::string &kind : kinds) {
    if (kind == "separate" || kind == "separate_spaced") {  // seven hipMallocs, fresh every trial (what a torch process does)
      std::vector<void *> spacers;
      for (int trial = 0; trial < 24; ++trial) {
        uint8_t *ptr[7];
        for (int k = 0; k < 7; ++k) {
          if (kind == "separate_spaced") {
            void *sp = nullptr;
            CHECK(hipMalloc(&sp, ((rng() % 199) + 1) * MiB));
            spacers.push_back(sp);
          }
          CHECK(hipMalloc(reinterpret_cast<void **>(&ptr[k]), k < 6 ? kFloats[k] * 4 : g_stream_bytes));
        }
        const Timing t = measure_ptrs(ptr, reps);
        std::printf("{\"kind\": \"%s\", \"exp\": \"fresh\", \"param_mib\": %d, \"ptrs\": [\"%p\", \"%p\", \"%p\", \"%p\", \"%p\", \"%p\", \"%p\"], "
                    "\"dec_seq_ms\": %.4f, \"dec_il_ms\": %.4f, \"enc_ms\": %.4f}\n",
                    kind.c_str(), trial, (void *)ptr[0], (void *)ptr[1], (void *)ptr[2], (void *)ptr[3], (void *)ptr[4], (void *)ptr[5], (void *)ptr[6],
                    t.dec_seq, t.dec_il, t.enc);
        std::fflush(stdout);
        CHECK(hipDeviceSynchronize());
        for (int k = 0; k < 7; ++k) CHECK(hipFree(ptr[k]));
        if (spacers.size() > 40) {
          for (int q = 0; q < 20; ++q) CHECK(hipFree(spacers[q]));
          spacers.erase(spacers.begin(), spacers.begin() + 20);
        }
      }
      for (void *sp : spacers) CHECK(hipFree(sp));
      continue;
    }
    if (kind == "separate_offset") {  // separate allocations, every array begun a random number of 2 MiB granules into its own
      for (int trial = 0; trial < 16; ++trial) {
        uint8_t *ptr[7], *base[7];
        uint64_t skip[7];
        for (int k = 0; k < 7; ++k) {
          skip[k] = (rng() % 127 + 1) * 2 * MiB;
          CHECK(hipMalloc(reinterpret_cast<void **>(&base[k]), (k < 6 ? kFloats[k] * 4 : g_stream_bytes) + skip[k]));
          ptr[k] = base[k] + skip[k];
        }
        const Timing t = measure_ptrs(ptr, reps);
        std::printf("{\"kind\": \"%s\", \"exp\": \"fresh\", \"param_mib\": %d, \"dec_seq_ms\": %.4f, \"dec_il_ms\": %.4f, \"enc_ms\": %.4f}\n", kind.c_str(),
                    trial, t.dec_seq, t.dec_il, t.enc);
        std::fflush(stdout);
        CHECK(hipDeviceSynchronize());
        for (int k = 0; k < 7; ++k) CHECK(hipFree(base[k]));
      }
      continue;
    }
    if (kind == "vmm_per_array") {  // one physical handle per array, mapped back to back into ONE address range
      hipMemAllocationProp prop = vmm_prop();
      const uint64_t gran = 2 * MiB;
      for (int trial = 0; trial < 8; ++trial) {
        uint64_t len[7], total = 0;
        for (int k = 0; k < 7; ++k) {
          len[k] = up(k < 6 ? kFloats[k] * 4 : g_stream_bytes, gran);
          total += len[k];
        }
        uint8_t *va = nullptr;
        CHECK(hipMemAddressReserve(reinterpret_cast<void **>(&va), total, gran, nullptr, 0));
        hipMemGenericAllocationHandle_t h[7];
        uint8_t *ptr[7];
        uint64_t at = 0;
        for (int k = 0; k < 7; ++k) {
          CHECK(hipMemCreate(&h[k], len[k], &prop, 0));
          CHECK(hipMemMap(va + at, len[k], 0, h[k], 0));
          ptr[k] = va + at;
          at += len[k];
        }
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = 0;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CHECK(hipMemSetAccess(va, total, &acc, 1));
        const Timing t = measure_ptrs(ptr, reps);
        std::printf("{\"kind\": \"%s\", \"exp\": \"fresh\", \"param_mib\": %d, \"dec_seq_ms\": %.4f, \"dec_il_ms\": %.4f, \"enc_ms\": %.4f}\n", kind.c_str(),
                    trial, t.dec_seq, t.dec_il, t.enc);
        std::fflush(stdout);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemUnmap(va, total));
        for (int k = 0; k < 7; ++k) CHECK(hipMemRelease(h[k]));
        CHECK(hipMemAddressFree(va, total));
      }
      continue;
    }
    if (kind.rfind("aligned", 0) == 0) {  // "aligned<MiB>": one slab, every array begun at a multiple of <MiB>
      const uint64_t al = std::strtoull(kind.c_str() + 7, nullptr, 10) * MiB;
      Layout l;
      uint64_t at = 0;
      const int order[7] = {5, 6, 2, 0, 1, 4, 3};  // largest first
      for (int q = 0; q < 7; ++q) {
        const int k = order[q];
        l.off[k] = at;
        at = up(at + (k < 6 ? kFloats[k] * 4 : g_stream_bytes), al);
      }
      for (int trial = 0; trial < 4; ++trial) {
        Slab s = make_slab("malloc", at, rng);
        s.kind = kind;
        report(s, "fresh", (uint64_t)trial * MiB, l, measure(s, l, reps));
        report(s, "fresh", (uint64_t)trial * MiB, l, measure(s, l, reps));
        free_slab(s);
      }
      continue;
    }
    if (kind.rfind("matrix", 0) == 0) {  // "matrix<K>": K chunks of 3 GiB; outputs in chunk i, stream in chunk j, every pair
      const int K = std::atoi(kind.c_str() + 6) > 0 ? std::atoi(kind.c_str() + 6) : 32;
      std::vector<uint8_t *> chunk(K);
      for (int i = 0; i < K; ++i) CHECK(hipMalloc(reinterpret_cast<void **>(&chunk[i]), 3 * GiB));
      const Layout base = packed_layout(g_stream_bytes, 2 * MiB);
      setenv("SPZ_AMD_GRID_ORDER", "interleaved", 1);
      spz_amd_header hdr = {3, (uint32_t)kN, (uint8_t)kDeg, 12, 0, 0};
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      for (int j = 0; j < K; ++j) {
        uint8_t *stream = chunk[j] + (3 * GiB - 640 * MiB);
        if (spz_amd_encode_device(&g_in, kN, kDeg, 0, SPZ_AMD_RDF, 3, stream, g_stream_bytes, nullptr) != SPZ_AMD_OK) return 3;
        std::printf("{\"kind\": \"matrix\", \"exp\": \"row\", \"stream_chunk\": %d, \"ptr\": \"%p\", \"dec_il_ms\": [", j, (void *)chunk[j]);
        for (int i = 0; i < K; ++i) {
          spz_amd_cloud_out out;
          out.positions = reinterpret_cast<float *>(chunk[i] + base.off[0]);
          out.scales = reinterpret_cast<float *>(chunk[i] + base.off[1]);
          out.rotations = reinterpret_cast<float *>(chunk[i] + base.off[2]);
          out.alphas = reinterpret_cast<float *>(chunk[i] + base.off[3]);
          out.colors = reinterpret_cast<float *>(chunk[i] + base.off[4]);
          out.sh = reinterpret_cast<float *>(chunk[i] + base.off[5]);
          float best = 1e9f;
          for (int r = 0; r < 4; ++r) {
            CHECK(hipEventRecord(e0, nullptr));
            if (spz_amd_decode_device(stream, g_stream_bytes, &hdr, SPZ_AMD_RDF, &out, nullptr) != SPZ_AMD_OK) return 3;
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
          }
          std::printf("%s%.3f", i ? ", " : "", best);
        }
        std::printf("]}\n");
        std::fflush(stdout);
      }
      for (int i = 0; i < K; ++i) CHECK(hipFree(chunk[i]));
      continue;
    }
    if (kind.rfind("chunkbw", 0) == 0) {  // "chunkbw<K>": write-only / read-only bandwidth of K chunks of 3 GiB, one at a time
      const int K = std::atoi(kind.c_str() + 7) > 0 ? std::atoi(kind.c_str() + 7) : 32;
      std::vector<uint8_t *> chunk(K);
      for (int i = 0; i < K; ++i) CHECK(hipMalloc(reinterpret_cast<void **>(&chunk[i]), 3 * GiB));
      uint32_t *sink = nullptr;
      CHECK(hipMalloc(reinterpret_cast<void **>(&sink), 256));
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      const uint64_t n16 = 2 * GiB / 16;
      for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < K; ++i) {
          float w = 1e9f, r = 1e9f;
          for (int rep = 0; rep < 4; ++rep) {
            float ms;
            CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(write_kernel, dim3(8192), dim3(256), 0, nullptr, reinterpret_cast<u32x4 *>(chunk[i]), n16);
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < w) w = ms;
            CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(read_kernel, dim3(8192), dim3(256), 0, nullptr, reinterpret_cast<const u32x4 *>(chunk[i]), n16, sink);
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < r) r = ms;
          }
          std::printf("{\"kind\": \"chunkbw\", \"exp\": \"bw\", \"pass\": %d, \"chunk\": %d, \"ptr\": \"%p\", \"write_GBps\": %.0f, \"read_GBps\": %.0f}\n", pass, i,
                      (void *)chunk[i], 2.0 * GiB / 1e6 / w, 2.0 * GiB / 1e6 / r);
        }
      }
      std::fflush(stdout);
      for (int i = 0; i < K; ++i) CHECK(hipFree(chunk[i]));
      continue;
    }
    if (kind.rfind("spread", 0) == 0) {  // "spread<K>": K chunks of 3 GiB; the seven arrays in one chunk, in two, or each in its own
      const int K = std::atoi(kind.c_str() + 6) > 0 ? std::atoi(kind.c_str() + 6) : 48;
      std::vector<uint8_t *> chunk(K);
      for (int i = 0; i < K; ++i) CHECK(hipMalloc(reinterpret_cast<void **>(&chunk[i]), 3 * GiB));
      const Layout base = packed_layout(g_stream_bytes, 2 * MiB);
      for (int trial = 0; trial < 60; ++trial) {
        int where[7];
        const int mode = trial % 3;  // 0: all in one chunk; 1: sh alone, the rest together; 2: every array its own chunk
        const int a = (int)(rng() % K);
        int b = (int)(rng() % K);
        if (b == a) b = (a + 1) % K;
        std::vector<int> perm(K);
        for (int i = 0; i < K; ++i) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        uint8_t *ptr[7];
        for (int k = 0; k < 7; ++k) {
          where[k] = mode == 0 ? a : mode == 1 ? (k == 5 ? a : b) : perm[k];
          ptr[k] = chunk[where[k]] + base.off[k];
        }
        const Timing t = measure_ptrs(ptr, reps);
        std::printf("{\"kind\": \"spread\", \"exp\": \"fresh\", \"param_mib\": %d, \"where\": [%d, %d, %d, %d, %d, %d, %d], "
                    "\"dec_seq_ms\": %.4f, \"dec_il_ms\": %.4f, \"enc_ms\": %.4f}\n",
                    mode, where[0], where[1], where[2], where[3], where[4], where[5], where[6], t.dec_seq, t.dec_il, t.enc);
        std::fflush(stdout);
      }
      for (int i = 0; i < K; ++i) CHECK(hipFree(chunk[i]));
      continue;
    }
    if (kind == "variants") {  // build/variants/libspz_amd_<name>.so (tools/tune.py build), $VARIANTS: a slow and a fast placement each
      const char *ve = std::getenv("VARIANTS");
      std::vector<std::string> names;
      for (std::string v = ve ? ve : "seq,il_g8"; !v.empty();) {
        const size_t q = v.find(',');
        names.push_back(v.substr(0, q));
        v = q == std::string::npos ? "" : v.substr(q + 1);
      }
      const int K = 12;
      std::vector<uint8_t *> chunk(K);
      for (int i = 0; i < K; ++i) CHECK(hipMalloc(reinterpret_cast<void **>(&chunk[i]), 3 * GiB));
      const Layout base = packed_layout(g_stream_bytes, 2 * MiB);
      // placements: "same" = all seven arrays in chunk 0; "apart k" = sh in chunk 0, the rest in chunk k
      struct Place { std::string name; uint8_t *ptr[7]; };
      std::vector<Place> places;
      for (int k = 0; k < K; ++k) {
        Place pl;
        pl.name = k == 0 ? "same" : "apart" + std::to_string(k);
        for (int a = 0; a < 7; ++a) pl.ptr[a] = (a == 5 ? chunk[0] : chunk[k]) + base.off[a];
        places.push_back(pl);
      }
      typedef int (*enc_fn)(const spz_amd_cloud_in *, uint64_t, int, int, int, int, uint8_t *, size_t, void *);
      typedef int (*dec_fn)(const uint8_t *, size_t, const spz_amd_header *, int, const spz_amd_cloud_out *, void *);
      unsetenv("SPZ_AMD_GRID_ORDER");
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      spz_amd_header hdr = {3, (uint32_t)kN, (uint8_t)kDeg, 12, 0, 0};
      for (const std::string &name : names) {
        const std::string path = std::string("build/variants/libspz_amd_") + name + ".so";
        void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
          std::fprintf(stderr, "skip %s: %s\n", name.c_str(), dlerror());
          continue;
        }
        enc_fn enc = reinterpret_cast<enc_fn>(dlsym(h, "spz_amd_encode_device"));
        dec_fn dec = reinterpret_cast<dec_fn>(dlsym(h, "spz_amd_decode_device"));
        for (const Place &pl : places) {
          spz_amd_cloud_out out = {reinterpret_cast<float *>(pl.ptr[0]), reinterpret_cast<float *>(pl.ptr[1]), reinterpret_cast<float *>(pl.ptr[2]),
                                   reinterpret_cast<float *>(pl.ptr[3]), reinterpret_cast<float *>(pl.ptr[4]), reinterpret_cast<float *>(pl.ptr[5])};
          std::vector<float> te, td;
          for (int r = 0; r < reps + 2; ++r) {
            float ms;
            CHECK(hipEventRecord(e0, nullptr));
            if (enc(&g_in, kN, kDeg, 0, SPZ_AMD_RDF, 3, pl.ptr[6], g_stream_bytes, nullptr) != SPZ_AMD_OK) return 3;
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) te.push_back(ms);
            CHECK(hipEventRecord(e0, nullptr));
            if (dec(pl.ptr[6], g_stream_bytes, &hdr, SPZ_AMD_RDF, &out, nullptr) != SPZ_AMD_OK) return 3;
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) td.push_back(ms);
          }
          std::sort(te.begin(), te.end());
          std::sort(td.begin(), td.end());
          std::printf("{\"kind\": \"variants\", \"exp\": \"variant\", \"variant\": \"%s\", \"place\": \"%s\", \"enc_ms\": %.4f, \"dec_ms\": %.4f}\n", name.c_str(),
                      pl.name.c_str(), te[te.size() / 2], td[td.size() / 2]);
          std::fflush(stdout);
        }
      }
      for (int i = 0; i < K; ++i) CHECK(hipFree(chunk[i]));
      continue;
    }
    if (kind == "pick") {  // the selection procedure: reference = sh inside the block of the rest (always the slow kind);
                           // candidates = separate sh allocations, spacers in between; how often is a fast one found, how soon?
      const uint64_t rest_bytes = up(base_rest_bytes(), 2 * MiB), sh_bytes = up(kFloats[5] * 4, 2 * MiB);
      for (int trial = 0; trial < 12; ++trial) {
        uint8_t *T = nullptr;
        CHECK(hipMalloc(reinterpret_cast<void **>(&T), rest_bytes + sh_bytes));
        auto rest_ptrs = [&](uint8_t *blk, uint8_t *shp, uint8_t *ptr[7]) {
          uint64_t at = 0;
          for (int k = 0; k < 7; ++k) {
            if (k == 5) continue;
            ptr[k] = blk + at;
            at = up(at + (k < 6 ? kFloats[k] * 4 : g_stream_bytes), 2 * MiB);
          }
          ptr[5] = shp;
        };
        uint8_t *ptr[7];
        rest_ptrs(T, T + rest_bytes, ptr);
        const Timing ref = measure_ptrs(ptr, 3);
        CHECK(hipDeviceSynchronize());
        CHECK(hipFree(T));
        uint8_t *R = nullptr;
        CHECK(hipMalloc(reinterpret_cast<void **>(&R), rest_bytes));
        std::vector<void *> held;
        std::printf("{\"kind\": \"pick\", \"exp\": \"pick\", \"trial\": %d, \"ref_il_ms\": %.4f, \"ref_enc_ms\": %.4f, \"cand_il_ms\": [", trial, ref.dec_il, ref.enc);
        for (int c = 0; c < 6; ++c) {
          uint8_t *shp = nullptr;
          CHECK(hipMalloc(reinterpret_cast<void **>(&shp), sh_bytes));
          rest_ptrs(R, shp, ptr);
          const Timing t = measure_ptrs(ptr, 3);
          std::printf("%s%.4f", c ? ", " : "", t.dec_il);
          held.push_back(shp);
          void *sp = nullptr;
          CHECK(hipMalloc(&sp, (uint64_t)(c + 1) * 6 * GiB));
          held.push_back(sp);
        }
        std::printf("]}\n");
        std::fflush(stdout);
        CHECK(hipDeviceSynchronize());
        for (void *h : held) CHECK(hipFree(h));
        CHECK(hipFree(R));
      }
      continue;
    }
    if (kind == "slabs") {  // a fresh slab per trial, packed layout
      for (int trial = 0; trial < 12; ++trial) {
        Slab s = make_slab("malloc", 4 * GiB, rng);
        s.kind = "slabs";
        const Layout base = packed_layout(g_stream_bytes, 2 * MiB);
        report(s, "fresh", (uint64_t)trial * MiB, base, measure(s, base, reps));
        void *sp = nullptr;
        CHECK(hipMalloc(&sp, ((rng() % 199) + 1) * MiB));  // kept: the next slab lies elsewhere
        free_slab(s);
      }
      continue;
    }
    Slab s = make_slab(kind, slab_bytes, rng);
    const Layout base = packed_layout(g_stream_bytes, 2 * MiB);
    const uint64_t packed_end = up(base.off[6] + g_stream_bytes, 2 * MiB);
    for (int r = 0; r < 6; ++r) report(s, "repeat", r * MiB, base, measure(s, base, reps));
    // the stream moved: 2 MiB steps, then coarser ones
    std::vector<uint64_t> steps;
    for (uint64_t k = 1; k <= 48; ++k) steps.push_back(k * 2 * MiB);
    for (uint64_t k = 1; k <= 16; ++k) steps.push_back(96 * MiB + k * 32 * MiB);
    for (uint64_t k = 1; k <= 16; ++k) steps.push_back(608 * MiB + k * 256 * MiB);
    for (uint64_t d : steps) {
      Layout l = base;
      l.off[6] = base.off[6] + d;
      if (l.off[6] + g_stream_bytes > s.bytes) break;
      report(s, "stream", d, l, measure(s, l, reps));
    }
    // the sh array moved (behind everything else)
    for (uint64_t d : steps) {
      Layout l = base;
      l.off[5] = packed_end + d;
      if (l.off[5] + kFloats[5] * 4 > s.bytes) break;
      report(s, "sh", d, l, measure(s, l, reps));
    }
    // small offsets of the stream (sub-granule)
    for (uint64_t d = 4096; d < 2 * MiB; d *= 2) {
      Layout l = base;
      l.off[6] = base.off[6] + d;
      report(s, "stream_fine", d, l, measure(s, l, reps));
    }
    if (kind == "vmmS" || kind == "vmmN") {
      for (int r = 0; r < 12; ++r) {
        reshuffle(s, rng);
        report(s, "shuffle", r * MiB, base, measure(s, base, reps));
      }
    }
    free_slab(s);
  }
  return 0;
}
