// spz_lz77.hip — zlib's level-6 LZ77 parse on the MI355X, exact (include/spz_amd.h: spz_amd_zlib_parse_*).
//
// The reference's container step (compressGzipped, load-spz.cc:186-214: one zlib stream, default level) is what
// is left of a saveSpz once the quantise step runs on the GPU, and its bytes have to stay the reference's.
// zlib's parse looks serial (hash chains, lazy matching) but factors into three stages that are not
// (spz_lz77_core.hpp): the chains are a pure function of the input, longest_match() at a position depends on
// the parse state only through a threshold, and the lazy state machine over the resulting tables is O(1) per
// position.  One kernel per stage:
//
//   lz_links_kernel   one workgroup per 512 KiB of input (+32 KiB warm-up): zlib's head[] as 65536 16-bit
//                     entries in LDS (128 KiB), 512 positions per round; each wave finds the equal hashes among its
//                     64 positions with a readlane/ballot loop, then the eight waves take the table in position order.
//                     Entries older than 32 KiB are retired by a sweep every 8192 positions, which is what keeps
//                     16-bit positions unambiguous.  Output: link[p], 2 B per input byte.
//   lz_match_kernel   one workgroup per 16 KiB of positions: the 48 KiB of input those positions can reach and
//                     their 96 KiB of links are staged in LDS (144 KiB of the CU's 160), every thread walks the
//                     chains of 16 positions exactly like longest_match does — all reads are LDS reads — and
//                     writes the two results per position (chain budget 128 and 32).
//   lz_parse_kernel   one lane per 64 KiB job: deflate_slow's loop with the table lookup in place of the search
//                     (each lane keeps the 32 entries around its position in LDS: one HBM latency per 32 positions).
//                     Pass 0 records each job's lazy state over its first 8192 loop tops, pass 1 emits symbols
//                     and ends a job where its state equals its successor's record (spz_deflate.cpp's splice).
//   lz_compact_kernel the jobs' contributed symbol ranges, concatenated.
//
// The last 64-96 KiB of the input (where zlib's lookahead runs out and its window's stale bytes matter) are
// parsed by spz_deflate.cpp's serial job on the host; its records arrive here as the last job's successor.
// HBM per input byte: 1 (input) + 2 (links) + 8 (tables) + 1 (records) + 3.4 (job symbols); the dense symbol
// arrays reuse the tables' memory.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "spz_amd.h"
#include "spz_common.hpp"
#include "spz_lz77_core.hpp"

namespace spz_amd_detail {
namespace {

using namespace spz_lz;

// ---- stage 1: links ------------------------------------------------------------------------------------
constexpr uint32_t kLinkThreads = 512, kLinkSegment = 1u << 19, kSweepEvery = 8192, kRetiredAge = 40000;

__global__ __launch_bounds__(512) void lz_links_kernel(const uint8_t *__restrict__ d, uint64_t n_pos,
                                                       uint16_t *__restrict__ link) {
  __shared__ uint16_t head[HASH_MASK + 1];  // low 16 bits of the newest position per hash
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t s0 = (uint64_t)blockIdx.x * kLinkSegment;
  const uint64_t s1 = (s0 + kLinkSegment < n_pos) ? s0 + kLinkSegment : n_pos;
  const uint64_t start = s0 >= W ? s0 - W : 0;  // the chains of the first positions reach 32 KiB back
  for (uint32_t i = tid; i <= HASH_MASK; i += kLinkThreads) head[i] = (uint16_t)((uint32_t)start - kRetiredAge);
  __syncthreads();
  for (uint64_t P = start; P < s1; P += kLinkThreads) {
    if (P != start && ((P - start) & (kSweepEvery - 1)) == 0) {
      // retire what no later position can reach: an entry never gets 65536 positions old
      for (uint32_t i = tid; i <= HASH_MASK; i += kLinkThreads) {
        const uint16_t age = (uint16_t)((uint32_t)P - head[i]);
        if (age >= W) head[i] = (uint16_t)((uint32_t)P - kRetiredAge);
      }
      __syncthreads();
    }
    const uint64_t p = P + tid;
    const bool valid = p < s1;
    uint32_t h = 0x10000u + lane;  // a hash of its own: equal to nobody's
    if (valid) h = hash3(d[p], d[p + 1], d[p + 2]);
    // lanes of this wave with the same hash
    unsigned long long cls = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const uint32_t hj = (uint32_t)__builtin_amdgcn_readlane((int)h, j);
      const bool eq = hj == h;
      const unsigned long long m = __ballot(eq);
      if (eq) cls = m;
    }
    const unsigned long long below = cls & ((1ull << lane) - 1ull);
    const bool newest = (cls >> lane) == 1ull;
    for (uint32_t turn = 0; turn < kLinkThreads / 64; ++turn) {
      if (wave == turn && valid) {
        uint32_t g;
        if (below != 0ull) {
          g = lane - (63u - (uint32_t)__clzll((long long)below));
        } else {
          const uint16_t age = (uint16_t)((uint32_t)p - head[h]);
          g = age < W ? age : 0u;
        }
        if (newest) head[h] = (uint16_t)p;
        if (p >= s0) link[p] = (uint16_t)g;
      }
      __syncthreads();
    }
  }
}

// ---- stage 2: match tables -----------------------------------------------------------------------------
constexpr uint32_t kMatchTile = 16384, kMatchThreads = 1024;
constexpr uint32_t kMatchDataDwords = (W + kMatchTile + kReadAhead + 3) / 4 + 2;
constexpr uint32_t kMatchLinkDwords = (W + kMatchTile) / 2;

struct LdsData {
  const uint32_t *s;
  __device__ __forceinline__ uint32_t load4(int32_t pos) const {  // pos: bytes from the window's origin
    const uint32_t off = (uint32_t)pos;
    const uint32_t a = s[off >> 2], b = s[(off >> 2) + 1];
    return __builtin_amdgcn_alignbyte(b, a, off & 3u);
  }
};
struct LdsLink {
  const uint16_t *s;
  __device__ __forceinline__ uint32_t operator()(int32_t pos) const { return s[(uint32_t)pos]; }
};

__global__ __launch_bounds__(1024) void lz_match_kernel(const uint8_t *__restrict__ d, const uint16_t *__restrict__ link,
                                                        uint64_t n_pos, uint64_t size, uint32_t *__restrict__ r128,
                                                        uint32_t *__restrict__ r32) {
  __shared__ uint32_t s_data[kMatchDataDwords];
  __shared__ uint32_t s_link[kMatchLinkDwords];
  const uint32_t tid = threadIdx.x;
  const uint64_t t0 = (uint64_t)blockIdx.x * kMatchTile;
  const long long origin = (long long)t0 - (long long)W;  // a multiple of 16 KiB; window position 0
  const uint32_t *d32 = reinterpret_cast<const uint32_t *>(d);
  const uint32_t *l32 = reinterpret_cast<const uint32_t *>(link);
  for (uint32_t i = tid; i < kMatchDataDwords; i += kMatchThreads) {
    const long long pos = origin + 4ll * i;
    s_data[i] = pos >= 0 ? d32[pos >> 2] : 0u;
  }
  for (uint32_t i = tid; i < kMatchLinkDwords; i += kMatchThreads) {
    const long long pos = origin + 2ll * i;
    s_link[i] = pos >= 0 ? l32[pos >> 1] : 0u;
  }
  __syncthreads();
  const LdsData data = {s_data};
  const LdsLink lk = {reinterpret_cast<const uint16_t *>(s_link)};
  for (uint32_t it = 0; it < kMatchTile / kMatchThreads; ++it) {
    const uint32_t local = it * kMatchThreads + tid;
    const uint64_t p = t0 + local;
    if (p < n_pos) {
      // the window base in window coordinates; a base below the window is out of every candidate's reach
      const long long b = (long long)base_at(p, size) - origin;
      uint32_t e128, e32;
      find_matches<int32_t>(data, lk, (int32_t)(W + local), (int32_t)(b > 0 ? b : 0), &e128, &e32);
      r128[p] = e128;
      r32[p] = e32;
    }
  }
}

// ---- stage 3: lazy state machine -------------------------------------------------------------------------
struct JobInfo {
  uint32_t lo, hi, spliced;
};

constexpr uint32_t kParseWindow = 32;  // table entries a lane fetches at a time

__global__ __launch_bounds__(64) void lz_parse_kernel(int pass, const uint32_t *__restrict__ r128,
                                                      const uint32_t *__restrict__ r32, uint32_t head_end, uint32_t n_jobs,
                                                      uint2 *__restrict__ rec, uint16_t *__restrict__ sym_dist,
                                                      uint8_t *__restrict__ sym_lc, JobInfo *__restrict__ info) {
  // The loop's next position depends on the entry it has just read, so a read from HBM per loop top would be all
  // latency: each lane keeps the 32 entries around its position in LDS ([entry][lane]: conflict-free) and
  // refills them with eight 16-byte loads when it leaves them.
  __shared__ uint32_t s_win[kParseWindow * 64];
  const uint32_t lane = threadIdx.x;
  const uint32_t j = blockIdx.x * 64u + lane;
  if (j >= n_jobs) return;
  const uint32_t begin = j * kJobBytes;
  uint32_t win_base = 0xffffffffu;
  auto e128 = [&](uint32_t pos) {
    const uint32_t b = pos & ~(kParseWindow - 1u);
    if (b != win_base) {
      const uint4 *src = reinterpret_cast<const uint4 *>(r128 + b);
#pragma unroll
      for (uint32_t q = 0; q < kParseWindow / 4; ++q) {
        const uint4 v = src[q];
        s_win[(4 * q + 0) * 64 + lane] = v.x;
        s_win[(4 * q + 1) * 64 + lane] = v.y;
        s_win[(4 * q + 2) * 64 + lane] = v.z;
        s_win[(4 * q + 3) * 64 + lane] = v.w;
      }
      win_base = b;
    }
    return s_win[(pos - b) * 64 + lane];
  };
  auto e32 = [&](uint32_t pos) { return r32[pos]; };
  uint32_t s = begin;
  LazyState<uint32_t> st;
  uint32_t nsym = 0;
  if (pass == 0) {
    if (j == 0) return;  // nobody splices into the first job
    uint2 *r = rec + (size_t)j * kRecordWindow;
    auto count = [&](uint32_t, uint32_t) { ++nsym; };
    while (s - begin < kRecordWindow) {
      r[s - begin] = make_uint2(pack_state(st, s), nsym);
      lazy_step(s, st, e128, e32, count);
    }
    return;
  }
  const uint32_t next = (begin + kJobBytes < head_end) ? begin + kJobBytes : head_end;
  const uint2 *succ = rec + (size_t)(j + 1) * kRecordWindow;
  uint16_t *od = sym_dist + (size_t)j * kJobSymbolStride;
  uint8_t *ol = sym_lc + (size_t)j * kJobSymbolStride;
  auto emit = [&](uint32_t dist, uint32_t lc) {
    od[nsym] = (uint16_t)dist;
    ol[nsym] = (uint8_t)lc;
    ++nsym;
  };
  uint32_t spliced = 0;
  for (;;) {
    if (s >= next) {
      if (s - next >= kRecordWindow) break;
      const uint2 r = succ[s - next];
      if (r.x == pack_state(st, s)) {
        info[j].hi = nsym;
        info[j + 1].lo = r.y;
        spliced = 1;
        break;
      }
    }
    if (nsym + 1 >= kJobSymbolStride) break;
    lazy_step(s, st, e128, e32, emit);
  }
  info[j].spliced = spliced;
}

// ---- stage 4: the contributed ranges, concatenated ---------------------------------------------------------
__global__ __launch_bounds__(256) void lz_compact_kernel(const uint16_t *__restrict__ sym_dist,
                                                         const uint8_t *__restrict__ sym_lc,
                                                         const JobInfo *__restrict__ info,
                                                         const unsigned long long *__restrict__ goff,
                                                         uint16_t *__restrict__ dense_dist, uint8_t *__restrict__ dense_lc) {
  const uint32_t j = blockIdx.x;
  const uint32_t lo = info[j].lo, n = info[j].hi - lo;
  const size_t src = (size_t)j * kJobSymbolStride + lo;
  const unsigned long long dst = goff[j];
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    dense_dist[dst + i] = sym_dist[src + i];
    dense_lc[dst + i] = sym_lc[src + i];
  }
}

struct LzContext {
  int device = 0;
  void *block = nullptr;
  const uint16_t *dense_dist = nullptr;
  const uint8_t *dense_lc = nullptr;
  uint64_t num_symbols = 0;
};

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace
}  // namespace spz_amd_detail

using namespace spz_amd_detail;

extern "C" {

int spz_amd_zlib_parse_open(const uint8_t *h_data, uint64_t size, uint64_t tail_begin, const uint32_t *h_tail_rec,
                            uint32_t n_rec, int device, void **ctx, uint64_t *num_symbols,
                            uint32_t *tail_first_symbol) {
  if (h_data == nullptr || h_tail_rec == nullptr || ctx == nullptr || num_symbols == nullptr ||
      tail_first_symbol == nullptr) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  *ctx = nullptr;
  if (tail_begin == 0 || tail_begin % W != 0 || n_rec < kRecordWindow || size >= (1ull << 32)) return SPZ_AMD_ERR_INVALID_ARG;
  const uint64_t n_pos = tail_begin + kTableSlack;  // positions the tables cover
  if (n_pos + kReadAhead + MIN_LOOKAHEAD > size) return SPZ_AMD_ERR_INVALID_ARG;
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;

  static const bool timing = std::getenv("SPZ_AMD_LZ_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[lz77] %-10s %.4f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };

  const uint32_t n_jobs = (uint32_t)((tail_begin + kJobBytes - 1) / kJobBytes);
  const uint32_t n_tiles = (uint32_t)((n_pos + kMatchTile - 1) / kMatchTile);
  const size_t pos_padded = (size_t)n_tiles * kMatchTile;
  // carve one allocation
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t at = off;
    off += round_up(bytes, 256);
    return at;
  };
  const size_t data_bytes = pos_padded + kReadAhead + 64;  // what the last tile stages
  const size_t o_data = carve(data_bytes);
  const size_t o_link = carve(pos_padded * sizeof(uint16_t));
  const size_t o_r128 = carve(pos_padded * sizeof(uint32_t));
  const size_t o_r32 = carve(pos_padded * sizeof(uint32_t));
  const size_t o_rec = carve((size_t)(n_jobs + 1) * kRecordWindow * sizeof(uint2));
  const size_t o_sd = carve((size_t)n_jobs * kJobSymbolStride * sizeof(uint16_t));
  const size_t o_sl = carve((size_t)n_jobs * kJobSymbolStride);
  const size_t o_info = carve((size_t)(n_jobs + 1) * sizeof(JobInfo));
  const size_t o_goff = carve((size_t)n_jobs * sizeof(unsigned long long));
  const size_t total = off;
  size_t free_b = 0, total_b = 0;
  SPZ_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  if (total + (size_t(256) << 20) > free_b) return SPZ_AMD_ERR_UNSUPPORTED;  // the caller parses on the host
  char *block = nullptr;
  SPZ_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&block), total));
  struct Free {
    char *p;
    ~Free() {
      if (p) (void)hipFree(p);
    }
  } holder{block};
  uint8_t *d_data = reinterpret_cast<uint8_t *>(block + o_data);
  uint16_t *d_link = reinterpret_cast<uint16_t *>(block + o_link);
  uint32_t *d_r128 = reinterpret_cast<uint32_t *>(block + o_r128);
  uint32_t *d_r32 = reinterpret_cast<uint32_t *>(block + o_r32);
  uint2 *d_rec = reinterpret_cast<uint2 *>(block + o_rec);
  uint16_t *d_sd = reinterpret_cast<uint16_t *>(block + o_sd);
  uint8_t *d_sl = reinterpret_cast<uint8_t *>(block + o_sl);
  JobInfo *d_info = reinterpret_cast<JobInfo *>(block + o_info);
  unsigned long long *d_goff = reinterpret_cast<unsigned long long *>(block + o_goff);

  hipStream_t st = nullptr;
  const size_t upload = std::min<size_t>(size, data_bytes);
  SPZ_HIP_TRY(hipMemcpyAsync(d_data, h_data, upload, hipMemcpyHostToDevice, st));
  if (upload < data_bytes) SPZ_HIP_TRY(hipMemsetAsync(d_data + upload, 0, data_bytes - upload, st));
  SPZ_HIP_TRY(hipMemsetAsync(d_link, 0, pos_padded * sizeof(uint16_t), st));
  SPZ_HIP_TRY(hipMemsetAsync(d_rec, 0, (size_t)(n_jobs + 1) * kRecordWindow * sizeof(uint2), st));
  SPZ_HIP_TRY(hipMemsetAsync(d_info, 0, (size_t)(n_jobs + 1) * sizeof(JobInfo), st));
  SPZ_HIP_TRY(hipMemcpyAsync(d_rec + (size_t)n_jobs * kRecordWindow, h_tail_rec, (size_t)kRecordWindow * sizeof(uint2),
                             hipMemcpyHostToDevice, st));
  lap("upload");
  const uint32_t n_seg = (uint32_t)((n_pos + kLinkSegment - 1) / kLinkSegment);
  hipLaunchKernelGGL(lz_links_kernel, dim3(n_seg), dim3(kLinkThreads), 0, st, d_data, n_pos, d_link);
  SPZ_HIP_TRY(hipGetLastError());
  lap("links");
  hipLaunchKernelGGL(lz_match_kernel, dim3(n_tiles), dim3(kMatchThreads), 0, st, d_data, d_link, n_pos, size, d_r128, d_r32);
  SPZ_HIP_TRY(hipGetLastError());
  lap("matches");
  const uint32_t parse_blocks = (n_jobs + 63u) / 64u;
  hipLaunchKernelGGL(lz_parse_kernel, dim3(parse_blocks), dim3(64), 0, st, 0, d_r128, d_r32, (uint32_t)tail_begin, n_jobs, d_rec,
                     d_sd, d_sl, d_info);
  SPZ_HIP_TRY(hipGetLastError());
  lap("records");
  hipLaunchKernelGGL(lz_parse_kernel, dim3(parse_blocks), dim3(64), 0, st, 1, d_r128, d_r32, (uint32_t)tail_begin, n_jobs, d_rec,
                     d_sd, d_sl, d_info);
  SPZ_HIP_TRY(hipGetLastError());
  std::vector<JobInfo> info(n_jobs + 1);
  SPZ_HIP_TRY(hipMemcpyAsync(info.data(), d_info, info.size() * sizeof(JobInfo), hipMemcpyDeviceToHost, st));
  SPZ_HIP_TRY(hipStreamSynchronize(st));
  lap("parse");
  std::vector<unsigned long long> goff(n_jobs);
  unsigned long long total_syms = 0;
  for (uint32_t j = 0; j < n_jobs; ++j) {
    if (!info[j].spliced || info[j].hi < info[j].lo) return SPZ_AMD_ERR_UNSUPPORTED;  // no meeting point: host parse
    goff[j] = total_syms;
    total_syms += info[j].hi - info[j].lo;
  }
  if (total_syms > n_pos) return SPZ_AMD_ERR_UNSUPPORTED;
  SPZ_HIP_TRY(hipMemcpyAsync(d_goff, goff.data(), goff.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
  // the tables are done with: their memory takes the dense arrays (2 B and 1 B per symbol, at most one symbol per position)
  uint16_t *dense_dist = reinterpret_cast<uint16_t *>(d_r128);
  uint8_t *dense_lc = reinterpret_cast<uint8_t *>(d_r32);
  hipLaunchKernelGGL(lz_compact_kernel, dim3(n_jobs), dim3(256), 0, st, d_sd, d_sl, d_info, d_goff, dense_dist, dense_lc);
  SPZ_HIP_TRY(hipGetLastError());
  SPZ_HIP_TRY(hipStreamSynchronize(st));
  lap("compact");
  LzContext *c = new (std::nothrow) LzContext();
  if (c == nullptr) return SPZ_AMD_ERR_HIP;
  c->device = device;
  c->block = block;
  holder.p = nullptr;
  c->dense_dist = dense_dist;
  c->dense_lc = dense_lc;
  c->num_symbols = total_syms;
  *ctx = c;
  *num_symbols = total_syms;
  *tail_first_symbol = info[n_jobs].lo;
  return SPZ_AMD_OK;
}

int spz_amd_zlib_parse_fetch(void *ctx, uint16_t *h_dist, uint8_t *h_lc) {
  LzContext *c = static_cast<LzContext *>(ctx);
  if (c == nullptr || (c->num_symbols > 0 && (h_dist == nullptr || h_lc == nullptr))) return SPZ_AMD_ERR_INVALID_ARG;
  if (c->num_symbols == 0) return SPZ_AMD_OK;
  DeviceGuard guard;
  int rc = guard.enter(c->device);
  if (rc != SPZ_AMD_OK) return rc;
  SPZ_HIP_TRY(hipMemcpyAsync(h_dist, c->dense_dist, c->num_symbols * sizeof(uint16_t), hipMemcpyDeviceToHost, nullptr));
  SPZ_HIP_TRY(hipMemcpyAsync(h_lc, c->dense_lc, c->num_symbols, hipMemcpyDeviceToHost, nullptr));
  SPZ_HIP_TRY(hipStreamSynchronize(nullptr));
  return SPZ_AMD_OK;
}

void spz_amd_zlib_parse_close(void *ctx) {
  LzContext *c = static_cast<LzContext *>(ctx);
  if (c == nullptr) return;
  DeviceGuard guard;
  if (guard.enter(c->device) == SPZ_AMD_OK && c->block) (void)hipFree(c->block);
  delete c;
}

}  // extern "C"
