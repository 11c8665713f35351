// spz_lz77.hip — zlib's level-6 LZ77 parse on the MI355X, exact (include/spz_amd.h: spz_amd_zlib_parse_*).
//
// The reference's container step (compressGzipped, load-spz.cc:186-214: one zlib stream, default level) is what
// is left of a saveSpz once the quantise step runs on the GPU, and its bytes have to stay the reference's.
// zlib's parse looks serial (hash chains, lazy matching) but factors into three stages that are not
// (spz_lz77_core.hpp): the chains are a pure function of the input, longest_match() at a position depends on
// the parse state only through a threshold, and the lazy state machine over the resulting tables is O(1) per
// position.  One kernel per stage:
//
//   lz_table_kernel   (twice) one workgroup per 512 KiB of input: a 65536-entry table in LDS gives every position the
//                     distance to the previous position with the same well-mixed hash of its three bytes (link[])
//                     and its running count among the positions with the same zlib hash (rank[]).
//   lz_chain_kernel   per position, how many steps of zlib's chain its link is worth (delta[], one byte).
//   lz_match_kernel   one workgroup per 4 KiB of positions, their reach of links, deltas and input bytes in LDS
//                     (144 KiB): every lane walks link[] — positions with the same three bytes, where zlib's own
//                     chain of 128 is mostly strangers for .spz data — adding up delta[] to know how far along
//                     zlib's chain a candidate is, and takes the tile's next position when its walk ends; two
//                     results per position (chain budget 128 and 32).
//   lz_parse_kernel   one lane per 8 KiB job: deflate_slow's loop with the table lookup in place of the search
//                     (each lane keeps the 32 entries around its position in LDS: one HBM latency per 32 positions),
//                     recording the lazy state at every loop top (one word per input position).
//   lz_stitch_kernel  every job continues into its successor's range until its state equals the recorded one
//                     (spz_deflate.cpp's splice): a few hundred positions.
//   lz_compact_kernel the jobs' contributed symbol ranges and stitches, concatenated.
//
// The last 64-96 KiB of the input (where zlib's lookahead runs out and its window's stale bytes matter) are
// parsed by spz_deflate.cpp's serial job on the host; its records arrive here as the last job's successor.
// HBM per input byte: 1 (input) + 5.1 (links, ranks, deltas) + 8 (tables) + 4 (records) + 6 (job and stitch symbols); the dense symbol
// arrays reuse the tables' memory.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "spz_amd.h"
#include "spz_common.hpp"
#include "spz_huff_core.hpp"
#include "spz_lz77_core.hpp"

namespace spz_amd_detail {
namespace {

using namespace spz_lz;

// ---- stage 1: chains -------------------------------------------------------------------------------------
// One table of 16-bit entries in LDS (LINK: 32768 of them, 64 KiB; RANK: 65536, 128 KiB), one workgroup per 512 KiB of input (+32 KiB warm-up), 512
// positions per round: each wave finds the equal keys among its 64 positions with a readlane/ballot loop, then
// the eight waves take the table in position order.
//   LINK: key = hash2, entry = low 16 bits of the newest position; out = distance to it (0: none within 32 KiB).
//         Entries older than 32 KiB are retired by a sweep every 8192 positions, which keeps 16-bit positions
//         unambiguous.
//   RANK: key = zlib's hash, entry = positions seen so far (mod 2^16); out = that count, into the segment's own
//         slab [s0 - 32 KiB, s1): ranks are compared only along one walk, which never leaves such a range.
constexpr uint32_t kLinkThreads = 512, kLinkSegment = 1u << 19, kSweepEvery = 8192, kRetiredAge = 40000;
constexpr uint32_t kRankSlab = W + kLinkSegment;
enum TableMode { TABLE_LINK = 0, TABLE_RANK = 1 };

template <int MODE>
__global__ __launch_bounds__(512) void lz_table_kernel(const uint8_t *__restrict__ d, uint64_t n_pos,
                                                       uint16_t *__restrict__ out, uint32_t first_segment) {
  constexpr uint32_t kEntries = MODE == TABLE_LINK ? (1u << HASH2_BITS) : HASH_MASK + 1;  // 64 KiB (two workgroups per CU) / 128 KiB
  __shared__ uint16_t table[kEntries];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t segment = blockIdx.x + first_segment;
  const uint64_t s0 = (uint64_t)segment * kLinkSegment;
  const uint64_t s1 = (s0 + kLinkSegment < n_pos) ? s0 + kLinkSegment : n_pos;
  const uint64_t start = s0 >= W ? s0 - W : 0;  // the walks of the first positions reach 32 KiB back
  const uint16_t fresh = MODE == TABLE_LINK ? (uint16_t)((uint32_t)start - kRetiredAge) : (uint16_t)0;
  for (uint32_t i = tid; i < kEntries; i += kLinkThreads) table[i] = fresh;
  __syncthreads();
  uint16_t *slab = out + (size_t)segment * kRankSlab;  // RANK: entry i is position s0 - W + i
  // The workgroup is alone on its CU (the table takes 128 KiB of LDS), so nothing hides a load's latency for it: the
  // next round's three bytes (one unaligned dword; the input is padded) are requested before this round's work.
  auto load3 = [&](uint64_t q) -> uint32_t {
    uint32_t v = 0;
    if (q < s1) __builtin_memcpy(&v, d + q, 4);
    return v;
  };
  uint32_t bytes_next = load3(start + tid);
  for (uint64_t P = start; P < s1; P += kLinkThreads) {
    const uint32_t bytes = bytes_next;
    bytes_next = load3(P + kLinkThreads + tid);
    if (MODE == TABLE_LINK && P != start && ((P - start) & (kSweepEvery - 1)) == 0) {
      // retire what no later position can reach: an entry never gets 65536 positions old
      for (uint32_t i = tid; i < kEntries; i += kLinkThreads) {
        const uint16_t age = (uint16_t)((uint32_t)P - table[i]);
        if (age >= W) table[i] = (uint16_t)((uint32_t)P - kRetiredAge);
      }
      __syncthreads();
    }
    const uint64_t p = P + tid;
    const bool valid = p < s1;
    uint32_t h = 0x10000u;  // no position's key (the lanes past the end do nothing with their class)
    if (valid) {
      const uint8_t b0 = (uint8_t)bytes, b1 = (uint8_t)(bytes >> 8), b2 = (uint8_t)(bytes >> 16);
      h = MODE == TABLE_LINK ? hash2(b0, b1, b2) : hash3(b0, b1, b2);
    }
    // (The table divided among the waves by key instead of the positions — every wave looks through all 512 keys of a
    // round for its own, queues them and handles 64 at a time, so that no wave ever waits for another's table accesses
    // and a round has one barrier instead of eight — was built and measured in round 3: 12.0 + 9.9 ms against this
    // kernel's 10.5 + 8.3; the queueing costs more instructions than the barriers cost time.)
    // lanes of this wave with the same key
    // bit by bit: the lanes that agree with this one on bit b are the ballot of that bit or its complement
    // (17 ballots instead of 64 readlane / compare rounds)
    unsigned long long cls = ~0ull;
#pragma unroll
    for (int b = 0; b <= 16; ++b) {
      const bool bit = ((h >> b) & 1u) != 0u;
      const unsigned long long m = __ballot(bit);
      cls &= bit ? m : ~m;
    }
    const unsigned long long below = cls & ((1ull << lane) - 1ull);
    const bool newest = (cls >> lane) == 1ull;
    for (uint32_t turn = 0; turn < kLinkThreads / 64; ++turn) {
      if (wave == turn && valid) {
        if (MODE == TABLE_LINK) {
          uint32_t g;
          if (below != 0ull) {
            g = lane - (63u - (uint32_t)__clzll((long long)below));
          } else {
            const uint16_t age = (uint16_t)((uint32_t)p - table[h]);
            g = age < W ? age : 0u;
          }
          if (newest) table[h] = (uint16_t)p;
          if (p >= s0) out[p] = (uint16_t)g;
        } else {
          const uint32_t r = (uint32_t)table[h] + (uint32_t)__popcll(below);
          if (newest) table[h] = (uint16_t)(r + 1u);
          slab[p + W - s0] = (uint16_t)r;
        }
      }
      __syncthreads();
    }
  }
}

// ---- stage 1b: chain steps ------------------------------------------------------------------------------
// delta[p] = how far along zlib's chain p's link goes (spz_lz77_core.hpp: chain_delta): two gathers per position
// (the link target's rank and bytes), all of them independent — the same reads as dependent steps of stage 2's walks
// cost it a trip to the vector cache per candidate.
__global__ __launch_bounds__(256) void lz_chain_kernel(const uint8_t *__restrict__ d, const uint16_t *__restrict__ link,
                                                       const uint16_t *__restrict__ rank_slabs, uint8_t *__restrict__ delta,
                                                       uint64_t first, uint64_t end, uint64_t n_pos) {
  // four consecutive positions per thread: one dword of deltas
  const uint64_t p0 = first + ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u;
  if (p0 >= end) return;
  uint32_t out = 0;
  uint64_t own;
  __builtin_memcpy(&own, d + p0, 8);  // the bytes of the four positions' triples (the input is padded)
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    const uint64_t p = p0 + i;
    uint32_t v = 0;
    const uint32_t gap = p < n_pos ? link[p] : 0u;
    if (gap != 0u) {
      const uint64_t seg = p / kLinkSegment;
      const uint16_t *slab = rank_slabs + seg * kRankSlab;  // entry i is position seg * kLinkSegment - W + i
      const uint64_t at = p + W - seg * kLinkSegment;
      uint32_t theirs;
      __builtin_memcpy(&theirs, d + (p - gap), 4);
      const uint32_t mine = (uint32_t)(own >> (8 * i));
      v = chain_delta(slab[at], slab[at - gap], ((mine ^ theirs) & 0xffffffu) == 0u);
    }
    out |= v << (8 * i);
  }
  *reinterpret_cast<uint32_t *>(delta + p0) = out;
}

// ---- stage 2: match tables -----------------------------------------------------------------------------
// One workgroup per 4 KiB of positions: the links, chain steps and input bytes of the 36 KiB of positions those can
// reach are staged in LDS (144 KiB of the CU's 160).  A walk's step reads link and delta at the current position and
// the candidate's eight bytes — all from LDS, and only the first two on the chain of dependent reads; global memory
// is touched for the 16-bit ranks after a stranger and for matches longer than eight bytes.
constexpr uint32_t kMatchTile = 4096, kMatchThreads = 1024;
constexpr uint32_t kMatchWindow = W + kMatchTile;
constexpr uint32_t kMatchDataDwords = kMatchWindow / 4 + 4;  // a position's eight bytes are read as three aligned dwords (a multiple of 4)

struct WindowData {
  const uint8_t *d;       // the input
  long long origin;       // absolute position of window position 0
  const uint32_t *s_data; // the window's bytes in LDS
  __device__ __forceinline__ uint32_t load4(int32_t pos) const {  // anywhere (the long compares run past the window)
    uint32_t v;
    __builtin_memcpy(&v, d + (origin + pos), 4);  // one unaligned dword load
    return v;
  }
  __device__ __forceinline__ uint64_t load8(int32_t pos) const {  // a window position
    const uint32_t i = (uint32_t)pos >> 2, sh = (uint32_t)pos & 3u;
    const uint32_t d0 = s_data[i], d1 = s_data[i + 1], d2 = s_data[i + 2];
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
    return ((uint64_t)hi << 32) | lo;
  }
};
struct WindowU16 {
  const uint16_t *s;
  __device__ __forceinline__ uint32_t operator()(int32_t pos) const { return s[(uint32_t)pos]; }
};
struct WindowU8 {
  const uint8_t *s;
  __device__ __forceinline__ uint32_t operator()(int32_t pos) const { return s[(uint32_t)pos]; }
};
struct WindowRank {  // the tile's segment's slab from the window's first position on, in global memory
  const uint16_t *g;
  __device__ __forceinline__ uint32_t operator()(int32_t pos) const { return g[pos]; }
};

__global__ __launch_bounds__(1024) void lz_match_kernel(const uint8_t *__restrict__ d, const uint16_t *__restrict__ link,
                                                        const uint8_t *__restrict__ delta, const uint16_t *__restrict__ rank_slabs,
                                                        uint64_t n_pos, uint64_t size, uint2 *__restrict__ r,
                                                        uint32_t first_tile, uint32_t refill,
                                                        unsigned long long *__restrict__ stats) {
  // one array, so that the order is this one: the bytes and the deltas within reach of a 16-bit instruction offset
  __shared__ uint4 s_all[(kMatchDataDwords * 4 + kMatchWindow + kMatchWindow * 2) / 16 + 1];
  uint32_t *const s_data = reinterpret_cast<uint32_t *>(s_all);
  uint4 *const s_delta = s_all + kMatchDataDwords / 4;
  uint4 *const s_link = s_delta + kMatchWindow / 16;
  uint32_t &s_next = *reinterpret_cast<uint32_t *>(s_link + kMatchWindow / 8);
  const uint32_t tid = threadIdx.x;
  if (tid == 0) s_next = 0u;
  const uint64_t t0 = (uint64_t)(blockIdx.x + first_tile) * kMatchTile;
  const long long origin = (long long)t0 - (long long)W;  // window position 0; a multiple of 4 KiB
  const uint64_t seg = t0 / kLinkSegment;
  {
    // 144 KiB in nine 16-byte loads per thread, all requested before the first is written to LDS (one load, one wait,
    // one write at a time this took a fifth of the tile's time)
    constexpr uint32_t kLinkVecs = kMatchWindow / 8, kDeltaVecs = kMatchWindow / 16, kDataVecs = kMatchWindow / 16;
    constexpr uint32_t kVecs = kLinkVecs + kDeltaVecs + kDataVecs, kPer = kVecs / kMatchThreads;
    static_assert(kVecs % kMatchThreads == 0, "the staging loop has no remainder");
    const uint4 *gl = reinterpret_cast<const uint4 *>(link + origin);    // (origin < 0: not dereferenced below 0)
    const uint4 *gk = reinterpret_cast<const uint4 *>(delta + origin);
    const uint4 *gd = reinterpret_cast<const uint4 *>(d + origin);
    uint4 v[kPer];
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) {
      const uint32_t i = tid + j * kMatchThreads;
      v[j] = make_uint4(0u, 0u, 0u, 0u);
      if (i < kLinkVecs) {
        if (origin + 8ll * i >= 0) v[j] = gl[i];
      } else if (i < kLinkVecs + kDeltaVecs) {
        if (origin + 16ll * (i - kLinkVecs) >= 0) v[j] = gk[i - kLinkVecs];
      } else {
        if (origin + 16ll * (i - kLinkVecs - kDeltaVecs) >= 0) v[j] = gd[i - kLinkVecs - kDeltaVecs];
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) {
      const uint32_t i = tid + j * kMatchThreads;
      if (i < kLinkVecs) s_link[i] = v[j];
      else if (i < kLinkVecs + kDeltaVecs) s_delta[i - kLinkVecs] = v[j];
      else reinterpret_cast<uint4 *>(s_data)[i - kLinkVecs - kDeltaVecs] = v[j];
    }
    if (tid < 4) s_data[kMatchWindow / 4 + tid] = reinterpret_cast<const uint32_t *>(d + origin)[kMatchWindow / 4 + tid];  // the last positions' read-ahead
  }
  __syncthreads();
  const WindowData data = {d, origin, s_data};
  const WindowU16 lk = {reinterpret_cast<const uint16_t *>(s_link)};
  const WindowU8 dk = {reinterpret_cast<const uint8_t *>(s_delta)};
  // the segment's slab starts at position seg * kLinkSegment - W: the window starts (t0 - seg * kLinkSegment) entries in
  const WindowRank rk = {rank_slabs + seg * kRankSlab + (t0 - seg * kLinkSegment)};
  // The window base (which positions zlib's window has dropped) is a step function of the position with steps only at
  // positions = -262 and -261 modulo W (base_at: through (s + 261) / W and s + 262 >= ...), so a tile — 4 KiB, aligned —
  // has at most three stretches of one value: evaluated once here (scalars) instead of in 64-bit arithmetic per position.
  const uint64_t step_b = ((t0 + 262u) / W + 1u) * W - 262u, step_a = step_b + 1u;  // the first such pair at or after t0
  const uint32_t local_b = step_b - t0 < kMatchTile ? (uint32_t)(step_b - t0) : kMatchTile;
  const uint32_t local_a = step_a - t0 < kMatchTile ? (uint32_t)(step_a - t0) : kMatchTile;
  auto window_base = [&](uint64_t p) {  // in window coordinates; a base below the window is out of every candidate's reach
    const long long b = (long long)base_at(p, size) - origin;
    return (int32_t)(b > 0 ? b : 0);
  };
  const int32_t base_0 = window_base(t0), base_b = window_base(step_b), base_a = window_base(step_a);
  const uint32_t n_local = n_pos - t0 < kMatchTile ? (uint32_t)(n_pos - t0) : kMatchTile;  // (the caller launches no tile past n_pos)
  // Walks differ in length by two orders of magnitude (ten candidates on average for an .spz stream, two hundred for
  // the longest), so a wave that gave each lane one position and waited for the slowest had one lane in twelve at
  // work.  Every lane is a walker of its own instead: one candidate per round; once `refill` lanes of the wave (16; 8 … 32 measure the same, 4 is 8 % slower) have
  // finished theirs, they write their results and take the tile's next positions (in any order: a position's entries
  // depend on nothing but the input).  Results are written then and not when a walk ends: some walk ends in nearly
  // every round, and the wave would run the store's address arithmetic each time for the sake of three lanes.
  const uint32_t lane = tid & 63u;
  const unsigned long long below_me = (1ull << lane) - 1ull;
  MatchWalk<int32_t> walk;
  bool active = false, loaded = false, dry = false;
  uint32_t local = 0;
  uint32_t n_rounds = 0, n_steps = 0;  // for SPZ_AMD_LZ_WALK_STATS' report (stats == nullptr otherwise)
  uint2 *const o = r + t0;  // {budget 128, budget 32} per position: one store (a lane's store is a cache line of its own)
  for (;;) {
    // ---- the idle lanes write what they found and take new positions
    const unsigned long long idle = __ballot(!active);
    const uint32_t n_idle = (uint32_t)__popcll(idle);
    if (!active && loaded) {
      uint32_t e128, e32;
      walk.finish(&e128, &e32);
      o[local] = make_uint2(e128, e32);
      loaded = false;
    }
    if (dry) break;  // (all lanes are idle: the loop below ran until they were)
    uint32_t first = 0;
    if (lane == 0) first = atomicAdd(&s_next, n_idle);
    first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
    if (!active) {
      local = first + (uint32_t)__popcll(idle & below_me);
      if (local < n_local) {
        walk.start(data, (int32_t)(W + local), local >= local_a ? base_a : (local >= local_b ? base_b : base_0));
        active = true;
        loaded = true;
      }
    }
    dry = first + n_idle >= n_local;  // the tile's positions are all given out: the walks are run to their ends
    // ---- one candidate per lane and round, until enough lanes are idle again (a loop of its own: the walks' state
    // stays where it is, which it did not when taking positions was a branch of the same loop)
    const uint32_t enough = dry ? 64u : refill;
    for (;;) {
      const unsigned long long walking = __ballot(active);
      if (64u - (uint32_t)__popcll(walking) >= enough) break;
      ++n_rounds;
      n_steps += (uint32_t)__popcll(walking);
      if (active) active = walk.step(data, lk, dk, rk);
    }
  }
  if (stats != nullptr && lane == 0) {
    atomicAdd(&stats[0], (unsigned long long)n_rounds);
    atomicAdd(&stats[1], (unsigned long long)n_steps);
    atomicMax(&stats[2], (unsigned long long)n_rounds);
  }
}

// ---- stage 3: lazy state machine -------------------------------------------------------------------------
struct JobInfo {
  uint32_t lo;        // first symbol of the job's own parse that it contributes (set by its predecessor's stitch)
  uint32_t n;         // symbols of its own parse
  uint32_t extra;     // symbols of its stitch
  uint32_t spliced;
  // where its own parse stopped: the first loop top at or past its end, with the lazy state there
  uint32_t end_s, end_available, end_length, end_byte, end_start;
};

constexpr uint32_t kParseWindow = 32;  // table entries a lane fetches at a time
// A job records its lazy state at the loop tops of its first kRecordWindow positions only: its predecessor's continuation
// meets it within a few hundred positions, a few thousand after long runs — and one that has not by then sends the
// stage to the next job size, as one that never does.  (Each record is a scattered 4-byte store, 64 cache lines per
// instruction: recorded at every loop top they were a third of what the kernel's stores cost.)
constexpr uint32_t kRecordWindow = 2048;  // for the smallest jobs; larger ones (the second attempt on) record throughout

// The loop's next position depends on the entry it has just read, so a read from HBM per loop top would be all
// latency: each lane keeps the 32 entries from its position on in LDS ([entry][lane]: conflict-free), for both tables.
// The refill is the wave's, not the lane's: when any lane has left its window every lane fetches anew from where it
// is, so the 64 lanes' refills — which would otherwise fall on different steps and stall all of them each time —
// cost one wait (measured on the 650 MB stream: 15.5 ms with per-lane refills on aligned windows).
struct EntryWindow {
  const uint32_t *s_win;
  uint32_t lane, base;
  __device__ __forceinline__ uint32_t operator()(uint32_t pos) const { return s_win[(pos - base) * 64 + lane]; }
};

__device__ __forceinline__ void fill_windows(uint32_t *s_win128, uint32_t *s_win32, const uint2 *table, uint32_t from, uint32_t lane) {
#pragma unroll
  for (uint32_t q = 0; q < kParseWindow / 2; ++q) {
    uint4 v;
    __builtin_memcpy(&v, table + from + 2 * q, sizeof(v));  // 8-byte aligned
    s_win128[(2 * q + 0) * 64 + lane] = v.x;
    s_win32[(2 * q + 0) * 64 + lane] = v.y;
    s_win128[(2 * q + 1) * 64 + lane] = v.z;
    s_win32[(2 * q + 1) * 64 + lane] = v.w;
  }
}

__global__ __launch_bounds__(64) void lz_parse_kernel(const uint2 *__restrict__ r,
                                                      uint32_t job_bytes, uint32_t head_end, uint32_t n_jobs,
                                                      uint32_t *__restrict__ rec, uint32_t *__restrict__ sym,
                                                      JobInfo *__restrict__ info, uint32_t first_job) {
  __shared__ uint32_t s_win128[kParseWindow * 64];
  __shared__ uint32_t s_win32[kParseWindow * 64];
  const uint32_t lane = threadIdx.x;
  const uint32_t j = first_job + blockIdx.x * 64u + lane;  // jobs [first_job, n_jobs) in this launch
  const bool mine = j < n_jobs;
  const uint32_t begin = mine ? j * job_bytes : 0u;
  const uint32_t next = mine ? (begin + job_bytes < head_end ? begin + job_bytes : head_end) : 0u;
  const uint32_t record = job_bytes <= kSmallestJob ? kRecordWindow : job_bytes;
  EntryWindow e128 = {s_win128, lane, 0u}, e32 = {s_win32, lane, 0u};
  // A job's symbols, one word each (distance | length or literal << 16), are stored four at a time: the lanes' symbol
  // arrays are 32 KiB apart, so a store is 64 cache lines whatever its width, and as 2 + 1 bytes per symbol the stores
  // were what the kernel's time went to.
  uint32_t *const o = sym + (size_t)j * job_symbol_stride(job_bytes);  // 16-byte aligned (the stride is a multiple of 4)
  uint32_t s = begin, nsym = 0, base = 0xffffffffu - kParseWindow;  // no window yet
  LazyState<uint32_t> st;
  uint4 last4 = make_uint4(0u, 0u, 0u, 0u);  // the last symbols, newest in .w
  auto emit = [&](uint32_t dist, uint32_t lc) {
    last4.x = last4.y;
    last4.y = last4.z;
    last4.z = last4.w;
    last4.w = dist | (lc << 16);
    if ((nsym & 3u) == 3u) *reinterpret_cast<uint4 *>(o + (nsym - 3u)) = last4;
    ++nsym;
  };
  for (;;) {
    const bool active = s < next;
    if (!__any(active)) break;
    if (__any(active && s - base >= kParseWindow)) {
      if (active) {
        base = s;  // the tables are padded past the last position a job reads
        fill_windows(s_win128, s_win32, r, s, lane);
        e128.base = e32.base = s;
      }
    }
    if (active) {
      if (s - begin < record) rec[s] = pack_state(st, s);
      lazy_step(s, st, e128, e32, emit);
    }
  }
  if (!mine) return;
  {  // the symbols since the last full four
    const uint32_t left = nsym & 3u;
    if (left >= 3u) o[nsym - 3u] = last4.y;
    if (left >= 2u) o[nsym - 2u] = last4.z;
    if (left >= 1u) o[nsym - 1u] = last4.w;
  }
  JobInfo &ji = info[j];
  ji.n = nsym;
  ji.end_s = s;
  ji.end_available = st.match_available;
  ji.end_length = st.match_length;
  ji.end_byte = st.byte_before;
  ji.end_start = st.match_start;
}

__global__ __launch_bounds__(64) void lz_stitch_kernel(const uint2 *__restrict__ r,
                                                       uint32_t job_bytes, uint32_t head_end, uint32_t n_jobs,
                                                       const uint32_t *__restrict__ rec, uint16_t *__restrict__ x_dist,
                                                       uint8_t *__restrict__ x_lc, JobInfo *__restrict__ info) {
  const uint32_t j = blockIdx.x * 64u + threadIdx.x;
  if (j >= n_jobs) return;
  const uint32_t next = (j + 1) * job_bytes < head_end ? (j + 1) * job_bytes : head_end;
  const uint32_t stop = (uint32_t)stitch_end(j, n_jobs, job_bytes, head_end);
  auto e128 = [&](uint32_t pos) { return r[pos].x; };
  auto e32 = [&](uint32_t pos) { return r[pos].y; };
  uint16_t *od = x_dist + (size_t)j * job_symbol_stride(job_bytes);
  uint8_t *ol = x_lc + (size_t)j * job_symbol_stride(job_bytes);
  uint32_t s = info[j].end_s, nsym = 0, spliced = 0;
  LazyState<uint32_t> st;
  st.match_available = info[j].end_available;
  st.match_length = info[j].end_length;
  st.byte_before = info[j].end_byte;
  st.match_start = info[j].end_start;
  auto emit = [&](uint32_t dist, uint32_t lc) {
    od[nsym] = (uint16_t)dist;
    ol[nsym] = (uint8_t)lc;
    ++nsym;
  };
  while (s < stop) {
    if (rec[s] == pack_state(st, s)) {
      // symbols the successor's own parse had emitted before this loop top
      uint32_t s2 = next, cnt = 0;
      LazyState<uint32_t> st2;
      auto count = [&](uint32_t, uint32_t) { ++cnt; };
      while (s2 < s) lazy_step(s2, st2, e128, e32, count);
      if (s2 == s) {
        info[j + 1].lo = cnt;
        info[j].end_s = s;  // the meeting point
        spliced = 1;
      }
      break;
    }
    lazy_step(s, st, e128, e32, emit);
  }
  info[j].extra = nsym;
  info[j].spliced = spliced;
}

// ---- stage 4: the contributed ranges, concatenated ---------------------------------------------------------
__global__ __launch_bounds__(256) void lz_compact_kernel(const uint32_t *__restrict__ sym,
                                                         const uint16_t *__restrict__ x_dist, const uint8_t *__restrict__ x_lc,
                                                         const JobInfo *__restrict__ info, uint32_t job_bytes,
                                                         const unsigned long long *__restrict__ goff,
                                                         uint16_t *__restrict__ dense_dist, uint8_t *__restrict__ dense_lc) {
  const uint32_t j = blockIdx.x;
  const uint32_t lo = info[j].lo, n = info[j].n - lo, x = info[j].extra;
  const size_t src = (size_t)j * job_symbol_stride(job_bytes) + lo, xsrc = (size_t)j * job_symbol_stride(job_bytes);
  const unsigned long long dst = goff[j];
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    const uint32_t v = sym[src + i];
    dense_dist[dst + i] = (uint16_t)v;
    dense_lc[dst + i] = (uint8_t)(v >> 16);
  }
  for (uint32_t i = threadIdx.x; i < x; i += 256) {
    dense_dist[dst + n + i] = x_dist[xsrc + i];
    dense_lc[dst + n + i] = x_lc[xsrc + i];
  }
}

// ---- Huffman stage (trees.c's tally and compress_block; the trees themselves are built on the host) ----------
// Block b is symbols [b * block_syms, ...) of the dense stream.  Stats: the literal/length and distance
// frequencies (what _tr_tally counts), the input bytes the block covers and the length of its last symbol.
__device__ __forceinline__ uint32_t d_code_of(const spz_amd_deflate_static *t, uint32_t dist_minus_1) {
  return dist_minus_1 < 256u ? t->dist_code[dist_minus_1] : t->dist_code[256u + (dist_minus_1 >> 7)];
}

__global__ __launch_bounds__(256) void lz_block_stats_kernel(const uint16_t *__restrict__ dist, const uint8_t *__restrict__ lc,
                                                             unsigned long long total, uint32_t block_syms,
                                                             const spz_amd_deflate_static *__restrict__ tables,
                                                             uint16_t *__restrict__ lfreq, uint16_t *__restrict__ dfreq,
                                                             uint32_t *__restrict__ bytes, uint32_t *__restrict__ last_len) {
  __shared__ uint32_t hl[288], hd[32], s_bytes;
  const uint32_t tid = threadIdx.x, b = blockIdx.x;
  for (uint32_t i = tid; i < 288; i += 256) hl[i] = 0;
  if (tid < 32) hd[tid] = 0;
  if (tid == 0) s_bytes = 0;
  __syncthreads();
  const unsigned long long g0 = (unsigned long long)b * block_syms;
  const unsigned long long g1 = (g0 + block_syms < total) ? g0 + block_syms : total;
  uint32_t mine = 0;
  for (unsigned long long i = g0 + tid; i < g1; i += 256) {
    const uint32_t d = dist[i], l = lc[i];
    uint32_t len = 1;
    if (d == 0) {
      atomicAdd(&hl[l], 1u);
    } else {
      atomicAdd(&hl[tables->length_code[l] + 257u], 1u);
      atomicAdd(&hd[d_code_of(tables, d - 1u)], 1u);
      len = l + MIN_MATCH;
    }
    mine += len;
    if (i + 1 == g1) last_len[b] = len;
  }
  atomicAdd(&s_bytes, mine);
  __syncthreads();
  for (uint32_t i = tid; i < 286; i += 256) lfreq[(size_t)b * 286 + i] = (uint16_t)hl[i];
  if (tid < 30) dfreq[(size_t)b * 30 + tid] = (uint16_t)hd[tid];
  if (tid == 0) {
    bytes[b] = s_bytes;
    if (g1 == g0) last_len[b] = 0;
  }
}

// ---- validity of the symbol stream (always on) -------------------------------------------------------------------
// Whatever the table, match, parse, stitch and compaction stages did, the symbols that are about to be coded must
// reproduce the input: symbol after symbol has its input position (a scan of the lengths inside its Huffman block, the
// block's first position from the caller's layout), a literal must be the byte there, a match must copy bytes equal to
// the ones at its position from at most 32 KiB back and never from before the input, and the block's symbols must
// cover exactly the input range the layout gives it.  One streaming read of the symbols and the input.  With this the
// LZ77 stage can only be lossless; the Huffman stage is covered by the per-block bit counts the host compares with
// its plan (and, when asked for, by inflating the finished member: SPZ_AMD_GZIP_VERIFY).
__global__ __launch_bounds__(256) void lz_validate_kernel(const uint16_t *__restrict__ dist, const uint8_t *__restrict__ lc,
                                                          unsigned long long total, uint32_t block_syms,
                                                          const uint8_t *__restrict__ input, unsigned long long size,
                                                          const spz_amd_deflate_block *__restrict__ blocks, uint32_t first_block,
                                                          uint32_t *__restrict__ bad) {
  __shared__ uint32_t s_wave[2][4];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, b = blockIdx.x + first_block;
  const unsigned long long g0 = (unsigned long long)b * block_syms;
  const unsigned long long g1 = (g0 + block_syms < total) ? g0 + block_syms : total;
  unsigned long long at = blocks[b].input_begin;  // input position of the round's first symbol (the same in every thread)
  uint32_t wrong = 0, round = 0;
  for (unsigned long long g = g0; g < g1; g += 256, ++round) {
    const unsigned long long i = g + tid;
    const bool have = i < g1;
    const uint32_t d = have ? dist[i] : 0u, l = have ? lc[i] : 0u;
    const uint32_t len = have ? (d ? l + MIN_MATCH : 1u) : 0u;
    uint32_t incl = len;
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(incl, off, 64);
      if (lane >= off) incl += y;
    }
    uint32_t *sw = s_wave[round & 1u];  // two sets: a wave may be a round ahead of the others
    if (lane == 63u) sw[wave] = incl;
    __syncthreads();
    const uint32_t w0 = sw[0], w1 = sw[1], w2 = sw[2], w3 = sw[3];
    const uint32_t before = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
    const unsigned long long pos = at + before + incl - len;
    at += (unsigned long long)w0 + w1 + w2 + w3;
    if (!have) continue;
    if (pos + len > size) {
      wrong = 1;
    } else if (d == 0u) {
      if (input[pos] != (uint8_t)l) wrong = 1;
    } else if (d > W || d > pos) {
      wrong = 1;
    } else {  // the input is known whole, so an overlapping copy is just two ranges that must be equal (buffer padded by 64)
      const uint8_t *p = input + pos, *q = p - d;
      uint32_t k = 0;
      for (; k + 8 <= len; k += 8) {
        unsigned long long x, y;
        __builtin_memcpy(&x, p + k, 8);
        __builtin_memcpy(&y, q + k, 8);
        if (x != y) wrong = 1;
      }
      if (k < len) {
        unsigned long long x, y;
        __builtin_memcpy(&x, p + k, 8);
        __builtin_memcpy(&y, q + k, 8);
        if (((x ^ y) << (8u * (8u - (len - k)))) != 0ull) wrong = 1;
      }
    }
  }
  if (tid == 0 && at != (unsigned long long)blocks[b].input_begin + blocks[b].input_bytes) wrong = 1;
  if (wrong) atomicAdd(bad, 1u);
}

// Encode: a thread owns a contiguous run of its block's symbols; the runs' bit offsets come from a scan of their
// lengths.  Words inside a run are the thread's own (plain stores); the first and the last word of a run may be
// shared with the neighbours and are OR-ed into the zero-filled body.
struct BitSink {
  uint32_t *body;            // the deflate body as little-endian words, zero-filled
  unsigned long long word;   // index of the word `acc` starts in
  unsigned long long acc;    // pending bits, bit 0 = bit `lead` ... of that word
  uint32_t fill;             // bits in acc (including the lead-in)
  bool first;                // the word about to be written is the run's first (shared with the previous run)
  __device__ __forceinline__ void put(uint32_t value, uint32_t nbits) {
    acc |= (unsigned long long)value << fill;
    fill += nbits;
    if (fill >= 32u) {
      const uint32_t w = (uint32_t)acc;
      if (first) atomicOr(&body[word], w);
      else body[word] = w;
      first = false;
      ++word;
      acc >>= 32;
      fill -= 32u;
    }
  }
  __device__ __forceinline__ void finish() {
    if (fill > 0u) atomicOr(&body[word], (uint32_t)acc);
  }
};

// ---- the trees on the device (spz_huff_core.hpp: the host writer's own source).  Building one block's trees is a serial
// walk over a heap, so a lane takes a block; its working memory (4.5 KB) is in LDS, which is what makes the walk's
// dependent accesses cheap, and only kTreeLanes lanes of a wave work so that four workgroups share a CU.
namespace hf = spz::huff;
constexpr int kTreeLanes = 8;
constexpr uint32_t kHeaderStride = 144;  // words per block header: 31 lead bits + 3 + 14 + 57 + 316 * 14 bits at most

struct TreeLds {
  uint16_t lt_freq[hf::HEAP_SIZE], lt_dl[hf::HEAP_SIZE];
  uint16_t dt_freq[2 * hf::D_CODES + 1], dt_dl[2 * hf::D_CODES + 1];
  uint16_t bt_freq[2 * hf::BL_CODES + 1], bt_dl[2 * hf::BL_CODES + 1], bt_code[hf::BL_CODES + 1];
  uint16_t heap[hf::HEAP_SIZE], bl_count[hf::MAX_BITS + 1], next_code[hf::MAX_BITS + 1];
  uint8_t depth[hf::HEAP_SIZE + 1];
};

struct BlockTrees {  // what the header needs beyond the literal and distance code lengths
  long long opt_len, static_len;
  uint16_t bl_code[hf::BL_CODES];
  uint8_t bl_len[hf::BL_CODES];
  uint8_t max_blindex;
  uint16_t lcodes, dcodes;
};

__global__ __launch_bounds__(64) void lz_tree_kernel(const uint16_t *__restrict__ lfreq, const uint16_t *__restrict__ dfreq,
                                                     uint32_t num_blocks, spz_amd_deflate_codes *__restrict__ codes,
                                                     BlockTrees *__restrict__ trees, spz_amd_deflate_plan *__restrict__ plan) {
  __shared__ TreeLds lds[kTreeLanes];
  const uint32_t lane = threadIdx.x, b = blockIdx.x * kTreeLanes + lane;
  if (lane >= (uint32_t)kTreeLanes || b >= num_blocks) return;
  TreeLds &m = lds[lane];
  for (int i = 0; i < hf::L_CODES; ++i) m.lt_freq[i] = lfreq[(size_t)b * hf::L_CODES + i];
  for (int i = 0; i < hf::D_CODES; ++i) m.dt_freq[i] = dfreq[(size_t)b * hf::D_CODES + i];
  spz_amd_deflate_codes &cd = codes[b];
  hf::TreeRef<uint16_t> lt = {m.lt_freq, m.lt_dl, m.lt_dl, cd.lcode, 0}, dt = {m.dt_freq, m.dt_dl, m.dt_dl, cd.dcode, 0},
                        bt = {m.bt_freq, m.bt_dl, m.bt_dl, m.bt_code, 0};
  hf::Work w = {m.heap, m.depth, m.bl_count, m.next_code, 0, 0, 0, 0};
  const int max_blindex = hf::plan_trees(lt, dt, bt, w, static_cast<const uint16_t *>(m.lt_dl), static_cast<const uint16_t *>(m.dt_dl));
  for (int i = 0; i < hf::L_CODES; ++i) {
    const uint16_t l = m.lt_dl[i];
    cd.llen[i] = (uint8_t)l;
    if (l == 0) cd.lcode[i] = 0;
  }
  for (int i = 0; i < hf::D_CODES; ++i) {
    const uint16_t l = m.dt_dl[i];
    cd.dlen[i] = (uint8_t)l;
    if (l == 0) cd.dcode[i] = 0;
  }
  BlockTrees &t = trees[b];
  t.opt_len = w.opt_len;
  t.static_len = w.static_len;
  for (int i = 0; i < hf::BL_CODES; ++i) {
    t.bl_len[i] = (uint8_t)m.bt_dl[i];
    t.bl_code[i] = m.bt_dl[i] ? m.bt_code[i] : (uint16_t)0;
  }
  t.max_blindex = (uint8_t)max_blindex;
  t.lcodes = (uint16_t)(lt.max_code + 1);
  t.dcodes = (uint16_t)(dt.max_code + 1);
  plan[b].opt_len = w.opt_len;
  plan[b].static_len = w.static_len;
}

// What a block writes before its first symbol (trees.c: _tr_stored_block's header, the static block's type bits, or
// send_all_trees), on the 32-bit grid of the body; a static block's codes replace the block's own.
struct HeaderSink {
  uint32_t *out;
  uint32_t words;
  unsigned long long acc;
  uint32_t fill, total;
  __device__ __forceinline__ void put(uint32_t value, uint32_t nbits) {
    acc |= (unsigned long long)value << fill;
    fill += nbits;
    total += nbits;
    if (fill >= 32u) {
      if (words < kHeaderStride) out[words++] = (uint32_t)acc;  // a header cannot be longer (the stride is its bound)
      acc >>= 32;
      fill -= 32u;
    }
  }
};

__global__ __launch_bounds__(64) void lz_header_kernel(spz_amd_deflate_block *__restrict__ blocks, spz_amd_deflate_codes *__restrict__ codes,
                                                       const BlockTrees *__restrict__ trees, uint32_t *__restrict__ header,
                                                       uint32_t *__restrict__ header_bits, uint32_t num_blocks) {
  const uint32_t b = blockIdx.x * 64u + threadIdx.x;
  if (b >= num_blocks) return;
  const spz_amd_deflate_block blk = blocks[b];
  const uint32_t last = (b + 1u == num_blocks) ? 1u : 0u;
  HeaderSink s = {header + (size_t)b * kHeaderStride, 0u, 0ull, (uint32_t)(blk.bit_start & 31ull), 0u};
  if (blk.choice == 0u) {
    s.put(last, 3);
    const uint32_t pad = (8u - (s.fill & 7u)) & 7u;  // bi_windup
    if (pad) s.put(0u, pad);
    s.put(blk.input_bytes & 0xffffu, 16);
    s.put(~blk.input_bytes & 0xffffu, 16);
  } else if (blk.choice == 1u) {
    s.put(2u + last, 3);
    spz_amd_deflate_codes &cd = codes[b];
    for (uint32_t n = 0; n < (uint32_t)hf::L_CODES; ++n) {
      const uint32_t len = (uint32_t)hf::static_llen((int)n);
      const uint32_t code = n <= 143u ? 0x30u + n : n <= 255u ? 0x190u + (n - 144u) : n <= 279u ? n - 256u : 0xC0u + (n - 280u);
      cd.lcode[n] = (uint16_t)hf::bit_reverse(code, (int)len);
      cd.llen[n] = (uint8_t)len;
    }
    for (uint32_t n = 0; n < (uint32_t)hf::D_CODES; ++n) {
      cd.dcode[n] = (uint16_t)hf::bit_reverse(n, 5);
      cd.dlen[n] = 5;
    }
  } else {
    const BlockTrees &t = trees[b];
    const spz_amd_deflate_codes &cd = codes[b];
    s.put(4u + last, 3);
    const uint32_t blcodes = (uint32_t)t.max_blindex + 1u;
    s.put((uint32_t)t.lcodes - 257u, 5);
    s.put((uint32_t)t.dcodes - 1u, 5);
    s.put(blcodes - 4u, 4);
    for (uint32_t rank = 0; rank < blcodes; ++rank) s.put(t.bl_len[hf::bl_order((int)rank)], 3);
    auto send = [&](int code, int extra_value, int extra_bits) {
      s.put(t.bl_code[code], t.bl_len[code]);
      if (extra_bits) s.put((uint32_t)extra_value, (uint32_t)extra_bits);
    };
    hf::walk_lengths(static_cast<const uint8_t *>(cd.llen), (int)t.lcodes - 1, send);
    hf::walk_lengths(static_cast<const uint8_t *>(cd.dlen), (int)t.dcodes - 1, send);
  }
  if (s.fill > 0u && s.words < kHeaderStride) s.out[s.words++] = (uint32_t)s.acc;
  blocks[b].header_word_begin = b * kHeaderStride;
  blocks[b].header_words = s.words;
  blocks[b].header_bits = s.total;
  header_bits[b] = s.total;
}

constexpr uint32_t kEncodeThreads = 256;

__global__ __launch_bounds__(256) void lz_encode_kernel(const uint16_t *__restrict__ dist, const uint8_t *__restrict__ lc,
                                                        unsigned long long total, uint32_t block_syms,
                                                        const uint8_t *__restrict__ input,
                                                        const spz_amd_deflate_static *__restrict__ tables,
                                                        const spz_amd_deflate_block *__restrict__ blocks,
                                                        const spz_amd_deflate_codes *__restrict__ codes,
                                                        const uint32_t *__restrict__ header_words, uint32_t *__restrict__ body,
                                                        unsigned long long *__restrict__ symbol_bits, uint32_t first_block) {
  __shared__ spz_amd_deflate_codes cd;
  __shared__ spz_amd_deflate_static tb;
  __shared__ uint32_t s_scan[kEncodeThreads];
  const uint32_t tid = threadIdx.x, b = blockIdx.x + first_block;
  const spz_amd_deflate_block blk = blocks[b];
  // the block's header (type bits, trees, or the stored block's length words): placed by the host on the word grid
  for (uint32_t i = tid; i < blk.header_words; i += kEncodeThreads) {
    const uint32_t w = header_words[blk.header_word_begin + i];
    if (w != 0u) atomicOr(&body[(blk.bit_start >> 5) + i], w);
  }
  if (blk.choice == 0u) {  // stored: the input bytes themselves, byte-aligned after the header
    uint8_t *out = reinterpret_cast<uint8_t *>(body) + ((blk.bit_start + blk.header_bits) >> 3);
    for (uint32_t i = tid; i < blk.input_bytes; i += kEncodeThreads) out[i] = input[(size_t)blk.input_begin + i];
    if (tid == 0) symbol_bits[b] = 8ull * blk.input_bytes;
    return;
  }
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(codes + b);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&cd);
    for (uint32_t i = tid; i < sizeof(spz_amd_deflate_codes) / 4; i += kEncodeThreads) dst[i] = src[i];
    const uint32_t *ts = reinterpret_cast<const uint32_t *>(tables);
    uint32_t *td = reinterpret_cast<uint32_t *>(&tb);
    for (uint32_t i = tid; i < sizeof(spz_amd_deflate_static) / 4; i += kEncodeThreads) td[i] = ts[i];
  }
  __syncthreads();
  const unsigned long long g0 = (unsigned long long)b * block_syms;
  const unsigned long long g1 = (g0 + block_syms < total) ? g0 + block_syms : total;
  const uint32_t n = (uint32_t)(g1 - g0) + 1u;  // the block's symbols and its END_BLOCK
  const uint32_t per = (n + kEncodeThreads - 1u) / kEncodeThreads;
  const uint32_t i0 = tid * per < n ? tid * per : n, i1 = i0 + per < n ? i0 + per : n;
  // what one symbol puts into the stream: the length part and the distance part (each at most 28 bits)
  auto symbol = [&](bool end_block, uint32_t d, uint32_t l, uint32_t *v0, uint32_t *n0, uint32_t *v1, uint32_t *n1) {
    *v1 = 0;
    *n1 = 0;
    if (end_block) {
      *v0 = cd.lcode[256];
      *n0 = cd.llen[256];
      return;
    }
    if (d == 0u) {
      *v0 = cd.lcode[l];
      *n0 = cd.llen[l];
      return;
    }
    const uint32_t code = tb.length_code[l];
    const uint32_t el = tb.extra_lbits[code];
    *v0 = cd.lcode[code + 257u] | (((l - tb.base_length[code]) & ((1u << el) - 1u)) << cd.llen[code + 257u]);  // 258: no extra bits
    *n0 = cd.llen[code + 257u] + el;
    const uint32_t dm = d - 1u;
    const uint32_t dc = dm < 256u ? tb.dist_code[dm] : tb.dist_code[256u + (dm >> 7)];
    const uint32_t ed = tb.extra_dbits[dc];
    *v1 = cd.dcode[dc] | (((dm - tb.base_dist[dc]) & ((1u << ed) - 1u)) << cd.dlen[dc]);
    *n1 = cd.dlen[dc] + ed;
  };
  // A thread's run is contiguous, the lanes' runs 128 symbols apart: read symbol by symbol, every load instruction
  // touches 64 different cache lines and a line is fetched again for each of its 64 entries unless it survives in the
  // vector cache between iterations (it does not: 16 KiB of lines per wave).  Sixteen symbols per load instead — 32 + 16
  // bytes per lane, unaligned (a block starts at an odd symbol index) — and a line is touched by two instructions.
  constexpr uint32_t kChunk = 16;
  auto for_each_symbol = [&](auto &&body) {
    for (uint32_t c = i0; c < i1; c += kChunk) {
      uint16_t dv[kChunk];
      uint8_t lv[kChunk];
      __builtin_memcpy(dv, dist + g0 + c, sizeof(dv));  // a few entries past the block's (or the stream's) last are in bounds:
      __builtin_memcpy(lv, lc + g0 + c, sizeof(lv));    // the arrays have a position's worth of entries each
#pragma unroll
      for (uint32_t k = 0; k < kChunk; ++k) {
        const uint32_t i = c + k;
        if (i < i1) body(i + 1u == n, (uint32_t)dv[k], (uint32_t)lv[k]);
      }
    }
  };
  uint32_t bits = 0;
  for_each_symbol([&](bool end_block, uint32_t d, uint32_t l) {
    uint32_t v0, n0, v1, n1;
    symbol(end_block, d, l, &v0, &n0, &v1, &n1);
    bits += n0 + n1;
  });
  // exclusive scan of the runs' bit counts (a block's symbols take at most 32768 * 48 bits)
  s_scan[tid] = bits;
  __syncthreads();
  for (uint32_t off = 1; off < kEncodeThreads; off <<= 1) {
    const uint32_t v = tid >= off ? s_scan[tid - off] : 0u;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  const uint32_t before = s_scan[tid] - bits;
  if (tid == kEncodeThreads - 1) symbol_bits[b] = s_scan[tid];
  if (i0 >= i1) return;
  const unsigned long long at = blk.bit_start + blk.header_bits + before;
  BitSink sink = {body, at >> 5, 0ull, (uint32_t)(at & 31ull), true};
  for_each_symbol([&](bool end_block, uint32_t d, uint32_t l) {
    uint32_t v0, n0, v1, n1;
    symbol(end_block, d, l, &v0, &n0, &v1, &n1);
    sink.put(v0, n0);
    if (n1) sink.put(v1, n1);
  });
  sink.finish();
}

struct LzContext {
  int device = 0;
  void *block = nullptr;
  uint16_t *dense_dist = nullptr;
  uint8_t *dense_lc = nullptr;
  uint64_t num_symbols = 0, capacity = 0;   // symbols in / that fit into the dense arrays
  const uint8_t *data = nullptr;            // the whole input
  uint64_t size = 0;
  // memory of the parse that the Huffman stage reuses
  char *scratch_a = nullptr, *scratch_b = nullptr, *scratch_c = nullptr;
  size_t scratch_a_bytes = 0, scratch_b_bytes = 0, scratch_c_bytes = 0;
  uint32_t stats_blocks = 0, trees_blocks = 0;  // block counts the statistics / the trees on the device were made for
  // encode_planned packs the bits in kBodyParts launches; encode_finish_ex copies each part out when its launch is done
  static constexpr int kBodyParts = 4;
  struct CopyLane *lane = nullptr;     // the device's copy stream and events, while this context holds them
  uint64_t part_end[kBodyParts] = {};  // byte offsets in the body
  int n_parts = 0;
  uint32_t validated_blocks = 0;       // blocks lz_validate_kernel has been enqueued for since the body was zeroed
  bool header_bits_dense = false;      // lz_header_kernel has filled EncodeArrays::header_bits (encode_planned)
};

// One copy stream and its events per device, made once (creating and destroying them per call cost what the overlap
// won) and lent to one context at a time; a second writer at the same time copies its body out in one piece.
struct CopyLane {
  hipStream_t stream = nullptr;
  hipEvent_t part_done[LzContext::kBodyParts] = {};
  bool in_use = false;
};
CopyLane g_copy_lane[kMaxDevices];
std::mutex g_copy_lane_mutex;

CopyLane *copy_lane_acquire(int device) {  // the current device is `device`
  if (device < 0 || device >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lock(g_copy_lane_mutex);
  CopyLane &l = g_copy_lane[device];
  if (l.in_use) return nullptr;
  if (l.stream == nullptr) {
    if (hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess) {
      l.stream = nullptr;
      return nullptr;
    }
    for (hipEvent_t &e : l.part_done) {
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;  // asked for again next time
    }
  }
  for (hipEvent_t e : l.part_done) {
    if (e == nullptr) return nullptr;
  }
  l.in_use = true;
  return &l;
}
void copy_lane_release(CopyLane *l) {
  if (l == nullptr) return;
  std::lock_guard<std::mutex> lock(g_copy_lane_mutex);
  l->in_use = false;
}

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace
}  // namespace spz_amd_detail

using namespace spz_amd_detail;

extern "C" {

}  // extern "C"

// Device-to-device copy by a kernel: a hipMemcpyAsync between device buffers goes to the copy engine, where — measured —
// it queues behind the host-to-device uploads that a feed is supposed to run beside (the session's kernels then start
// when the uploads end: no overlap at all).  Source and destination have the same offset inside their buffers, so they
// are aligned alike.
__global__ __launch_bounds__(256) void lz_copy_kernel(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, unsigned long long n) {
  const unsigned long long head = (16ull - (reinterpret_cast<unsigned long long>(dst) & 15ull)) & 15ull;
  const unsigned long long tid = (unsigned long long)blockIdx.x * 256ull + threadIdx.x, nthreads = (unsigned long long)gridDim.x * 256ull;
  if ((reinterpret_cast<unsigned long long>(src) & 15ull) != (reinterpret_cast<unsigned long long>(dst) & 15ull) || n < 64ull) {
    for (unsigned long long i = tid; i < n; i += nthreads) dst[i] = src[i];
    return;
  }
  if (tid < head) dst[tid] = src[tid];
  const unsigned long long groups = (n - head) / 16ull;
  const uint4 *s4 = reinterpret_cast<const uint4 *>(src + head);
  uint4 *d4 = reinterpret_cast<uint4 *>(dst + head);
  for (unsigned long long g = tid; g < groups; g += nthreads) d4[g] = s4[g];
  const unsigned long long done = head + groups * 16ull;
  if (tid < n - done) dst[done + tid] = src[done + tid];
}

extern "C" {

// ---- a parse whose table and match stages are fed as the input becomes final -----------------------------------------
// saveSpz: the stream's bytes become final section by section while the floats are still uploading (the GPU would idle
// through that upload); tables and match tables are a pure function of the input bytes, so they can start on the finished
// prefix.  A session holds the parse's block and how far the stages have come; spz_amd_zlib_parse_open_dev is a session
// fed once, whole.
struct LzSession {
  int device = 0;
  uint64_t size = 0, tail_begin = 0, n_pos = 0;
  uint32_t max_jobs = 0, n_tiles = 0, n_seg = 0;
  size_t pos_padded = 0, data_bytes = 0, rec_words = 0;
  size_t o_data = 0, o_link = 0, o_delta = 0, o_rank = 0, o_r128 = 0, o_r32 = 0, o_rec = 0, o_sd = 0, o_sl = 0, o_xd = 0, o_xl = 0, o_info = 0, o_goff = 0, o_stats = 0;
  char *block = nullptr;
  // A session fed in pieces runs its stages on three streams of its own: the table kernels of one piece are a few dozen
  // workgroups (one per 512 KiB, alone on their CUs for ~2 ms whatever the piece's size) and would leave the chip idle
  // if the match kernel of the piece before — which fills it — had to wait for them: LINK and RANK tables side by side,
  // the match tiles of piece k beside the tables of piece k + 1 (measured in one stream: 103 ms of kernels for twelve
  // pieces where the whole input at once takes 75).  A session fed once uses the default stream throughout.
  hipStream_t stream = nullptr;    // copies and LINK tables (null: the default stream)
  hipStream_t rank_stream = nullptr, match_stream = nullptr;
  hipEvent_t producer_done = nullptr, copy_done = nullptr, link_done = nullptr, rank_done = nullptr;
  bool own_stream = false, streams_from_cache = false;
  uint64_t fed = 0;                // input bytes copied into the block
  uint32_t seg_done = 0, tile_done = 0;
  bool zeroed = false;
  uint8_t *d_data() const { return reinterpret_cast<uint8_t *>(block + o_data); }
  uint16_t *d_link() const { return reinterpret_cast<uint16_t *>(block + o_link); }
  uint16_t *d_rank() const { return reinterpret_cast<uint16_t *>(block + o_rank); }
  uint8_t *d_delta() const { return reinterpret_cast<uint8_t *>(block + o_delta); }
  uint32_t *d_r128() const { return reinterpret_cast<uint32_t *>(block + o_r128); }
  uint32_t *d_r32() const { return reinterpret_cast<uint32_t *>(block + o_r32); }
};

// Idle lanes of a wave of the match kernel at which they take new positions (SPZ_AMD_LZ_REFILL, 1 ... 64; measured in
// round 3: see profiles/README.md)
static uint32_t lz_refill_lanes() {
  static const uint32_t v = []() {
    const char *e = std::getenv("SPZ_AMD_LZ_REFILL");
    const long x = e ? std::atol(e) : 16;
    return (uint32_t)(x < 1 ? 1 : (x > 64 ? 64 : x));
  }();
  return v;
}

static uint64_t lz_tail_begin(uint64_t size) { return (size - 2ull * W) / W * W; }  // spz_deflate.cpp: compressWithHeadParser

// A fed session's three streams and four events, kept per device between sessions: creating and destroying them per
// saveSpz (a stream is a hardware queue) was 7 - 10 ms of every call — more than a cloud under 4 M points gains from
// starting its container stage beside the upload.  One set per device; a second session at the same time on the same
// device makes (and destroys) its own.
namespace {
struct LzStreamSet {
  hipStream_t stream = nullptr, rank_stream = nullptr, match_stream = nullptr;
  hipEvent_t producer_done = nullptr, copy_done = nullptr, link_done = nullptr, rank_done = nullptr;
  bool made() const { return stream && rank_stream && match_stream && producer_done && copy_done && link_done && rank_done; }
  void destroy() {
    for (hipStream_t st : {stream, rank_stream, match_stream}) {
      if (st) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
      }
    }
    for (hipEvent_t e : {producer_done, copy_done, link_done, rank_done}) {
      if (e) (void)hipEventDestroy(e);
    }
    *this = LzStreamSet();
  }
  bool make() {  // the current device's
    if (create_stream(&stream, -1) != hipSuccess || create_stream(&rank_stream, -1) != hipSuccess ||
        create_stream(&match_stream, -1) != hipSuccess || hipEventCreateWithFlags(&producer_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&copy_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&link_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&rank_done, hipEventDisableTiming) != hipSuccess) {
      destroy();
      return false;
    }
    return true;
  }
};
std::mutex g_lz_streams_mutex;
LzStreamSet g_lz_streams[kMaxDevices];
bool g_lz_streams_in_use[kMaxDevices] = {};
}  // namespace

extern "C++" {
namespace spz_amd_detail {
void lz_streams_free_idle() {  // spz_amd_release_device_memory(): the sets no session holds go
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_lz_streams_mutex);
  for (int d = 0; d < kMaxDevices; ++d) {
    if (!g_lz_streams_in_use[d] && g_lz_streams[d].stream != nullptr && hipSetDevice(d) == hipSuccess) g_lz_streams[d].destroy();
  }
  (void)hipSetDevice(prev);
}
}  // namespace spz_amd_detail
}  // extern "C++"

static void lz_session_destroy(LzSession *q) {
  if (q == nullptr) return;
  DeviceGuard guard;
  if (guard.enter(q->device) == SPZ_AMD_OK) {
    if (q->own_stream) {
      LzStreamSet set;
      set.stream = q->stream;
      set.rank_stream = q->rank_stream;
      set.match_stream = q->match_stream;
      set.producer_done = q->producer_done;
      set.copy_done = q->copy_done;
      set.link_done = q->link_done;
      set.rank_done = q->rank_done;
      for (hipStream_t st : {q->stream, q->rank_stream, q->match_stream}) {
        if (st) (void)hipStreamSynchronize(st);  // nothing of this session may still be queued when its block goes
      }
      bool kept = false;
      if (q->streams_from_cache) {
        std::lock_guard<std::mutex> lock(g_lz_streams_mutex);
        g_lz_streams_in_use[q->device] = false;
        kept = true;
      }
      if (!kept) set.destroy();
    }
    if (q->block) scratch_release(q->device, q->block);
  }
  delete q;
}

// Geometry and memory of a parse of `size` bytes; the current device is `device`.
static int lz_session_create(uint64_t size, uint64_t tail_begin, int device, bool own_stream, LzSession **out) {
  *out = nullptr;
  if (tail_begin == 0 || tail_begin % W != 0 || size >= (1ull << 32)) return SPZ_AMD_ERR_INVALID_ARG;
  const uint64_t n_pos = tail_begin + kTableSlack;  // positions the tables cover
  if (n_pos + kReadAhead + MIN_LOOKAHEAD > size) return SPZ_AMD_ERR_INVALID_ARG;
  LzSession *q = new (std::nothrow) LzSession();
  if (q == nullptr) return SPZ_AMD_ERR_HIP;
  q->device = device;
  q->size = size;
  q->tail_begin = tail_begin;
  q->n_pos = n_pos;
  q->max_jobs = (uint32_t)(tail_begin / kSmallestJob);  // tail_begin is a multiple of W = 4 of them
  q->n_tiles = (uint32_t)((n_pos + kMatchTile - 1) / kMatchTile);
  q->pos_padded = (size_t)q->n_tiles * kMatchTile;
  // carve one allocation
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t at = off;
    off += round_up(bytes, 256);
    return at;
  };
  q->data_bytes = std::max<size_t>(q->pos_padded + kReadAhead + 64, round_up(size, 4) + 64);  // what the last tile stages; the whole input
  q->o_data = carve(q->data_bytes);
  q->n_seg = (uint32_t)((n_pos + kLinkSegment - 1) / kLinkSegment);
  q->o_link = carve(q->pos_padded * sizeof(uint16_t));
  q->o_rank = carve((size_t)q->n_seg * kRankSlab * sizeof(uint16_t));
  q->o_delta = carve(q->pos_padded);
  q->o_r128 = carve(q->pos_padded * sizeof(uint2));  // {budget 128, budget 32} per position
  q->o_r32 = q->o_r128 + q->pos_padded * sizeof(uint32_t);  // (the second half, for the dense literal / length array later)
  q->rec_words = (size_t)tail_begin + kTailWindow;
  q->o_rec = carve(q->rec_words * sizeof(uint32_t));
  const size_t sym_entries = (size_t)q->max_jobs * job_symbol_stride(kSmallestJob);  // larger jobs need fewer
  q->o_sd = carve(sym_entries * sizeof(uint32_t));  // a word per symbol: distance | length or literal << 16
  q->o_sl = q->o_sd;
  q->o_xd = carve(sym_entries * sizeof(uint16_t));
  q->o_xl = carve(sym_entries);
  q->o_info = carve((size_t)(q->max_jobs + 1) * sizeof(JobInfo));
  q->o_goff = carve((size_t)q->max_jobs * sizeof(unsigned long long));
  q->o_stats = carve(64);
  const size_t total = off;
  size_t free_b = 0;
  int rc = device_free_bytes(device, &free_b);
  if (rc == SPZ_AMD_OK && total + (size_t(256) << 20) > free_b) rc = SPZ_AMD_ERR_UNSUPPORTED;  // the caller parses on the host
  if (rc == SPZ_AMD_OK) rc = scratch_acquire(device, total, reinterpret_cast<void **>(&q->block));
  if (rc == SPZ_AMD_OK && own_stream) {
    LzStreamSet set;
    bool have = false;
    if (device >= 0 && device < kMaxDevices) {
      std::lock_guard<std::mutex> lock(g_lz_streams_mutex);
      if (!g_lz_streams_in_use[device]) {
        if (!g_lz_streams[device].made()) (void)g_lz_streams[device].make();
        if (g_lz_streams[device].made()) {
          set = g_lz_streams[device];
          g_lz_streams_in_use[device] = true;
          q->streams_from_cache = true;
          have = true;
        }
      }
    }
    if (!have && !set.make()) rc = SPZ_AMD_ERR_HIP;
    q->stream = set.stream;
    q->rank_stream = set.rank_stream;
    q->match_stream = set.match_stream;
    q->producer_done = set.producer_done;
    q->copy_done = set.copy_done;
    q->link_done = set.link_done;
    q->rank_done = set.rank_done;
    q->own_stream = true;
  }
  if (rc != SPZ_AMD_OK) {
    lz_session_destroy(q);
    return rc;
  }
  *out = q;
  return SPZ_AMD_OK;
}

// Bytes [0, upto) of the input are final (device memory `d_src` once `producer`'s queued work is done, or host memory
// `h_src`): brings them into the block and launches the table and match kernels for every segment and tile they cover.
// Nothing is waited for.  upto == size: everything that is left.
static int lz_session_feed(LzSession *q, const uint8_t *d_src, const uint8_t *h_src, uint64_t upto, hipStream_t producer) {
  hipStream_t st = q->stream;
  if (upto > q->size) upto = q->size;
  if (!q->zeroed) {
    SPZ_HIP_TRY(hipMemsetAsync(q->d_link(), 0, q->pos_padded * sizeof(uint16_t), st));
    SPZ_HIP_TRY(hipMemsetAsync(q->block + q->o_stats, 0, 64, st));
    q->zeroed = true;
  }
  if (upto > q->fed) {
    if (d_src != nullptr) {
      if (q->own_stream && producer != st) {  // the producer's kernels first
        SPZ_HIP_TRY(hipEventRecord(q->producer_done, producer));
        SPZ_HIP_TRY(hipStreamWaitEvent(st, q->producer_done, 0));
      }
      const unsigned long long nb = upto - q->fed;
      const uint32_t blocks = (uint32_t)std::min<unsigned long long>(4096ull, (nb / 16ull + 255ull) / 256ull + 1ull);
      hipLaunchKernelGGL(lz_copy_kernel, dim3(blocks), dim3(256), 0, st, q->d_data() + q->fed, d_src + q->fed, nb);
      SPZ_HIP_TRY(hipGetLastError());
    } else {
      SPZ_HIP_TRY(upload_adaptive(q->d_data() + q->fed, h_src + q->fed, upto - q->fed, st));
    }
    q->fed = upto;
    if (upto == q->size && q->size < q->data_bytes) SPZ_HIP_TRY(hipMemsetAsync(q->d_data() + q->size, 0, q->data_bytes - q->size, st));
    if (q->own_stream) {  // the RANK tables read the same bytes
      SPZ_HIP_TRY(hipEventRecord(q->copy_done, st));
      SPZ_HIP_TRY(hipStreamWaitEvent(q->rank_stream, q->copy_done, 0));
    }
  }
  hipStream_t rst = q->own_stream ? q->rank_stream : st, mst = q->own_stream ? q->match_stream : st;
  const bool all = q->fed == q->size;
  // a segment reads its positions' three bytes (one dword each): final up to its end + 4; the last segments at the end
  uint32_t seg_to = all ? q->n_seg : (uint32_t)std::min<uint64_t>(q->n_seg, q->fed >= 4 ? (q->fed - 4) / kLinkSegment : 0);
  if (seg_to > q->seg_done) {
    hipLaunchKernelGGL(lz_table_kernel<TABLE_LINK>, dim3(seg_to - q->seg_done), dim3(kLinkThreads), 0, st, q->d_data(), q->n_pos, q->d_link(),
                       q->seg_done);
    SPZ_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(lz_table_kernel<TABLE_RANK>, dim3(seg_to - q->seg_done), dim3(kLinkThreads), 0, rst, q->d_data(), q->n_pos, q->d_rank(),
                       q->seg_done);
    SPZ_HIP_TRY(hipGetLastError());
    if (q->own_stream) {  // the chain steps and the match tiles wait for both tables (and, through them, for the bytes)
      SPZ_HIP_TRY(hipEventRecord(q->link_done, st));
      SPZ_HIP_TRY(hipEventRecord(q->rank_done, rst));
      SPZ_HIP_TRY(hipStreamWaitEvent(mst, q->link_done, 0));
      SPZ_HIP_TRY(hipStreamWaitEvent(mst, q->rank_done, 0));
    }
    {
      const uint64_t c_first = (uint64_t)q->seg_done * kLinkSegment;
      const uint64_t c_end = seg_to == q->n_seg ? (uint64_t)q->pos_padded : (uint64_t)seg_to * kLinkSegment;  // multiples of 4 KiB
      const uint32_t blocks = (uint32_t)((c_end - c_first + 1023) / 1024);
      hipLaunchKernelGGL(lz_chain_kernel, dim3(blocks), dim3(256), 0, mst, q->d_data(), q->d_link(), q->d_rank(), q->d_delta(), c_first, c_end,
                         q->n_pos);
      SPZ_HIP_TRY(hipGetLastError());
    }
    q->seg_done = seg_to;
  }
  // a tile's walks read links and ranks up to its last position and input bytes kReadAhead + a dword beyond it
  const uint64_t tables_to = all ? q->pos_padded : (uint64_t)q->seg_done * kLinkSegment;
  const uint64_t bytes_to = all ? q->pos_padded : (q->fed > kReadAhead + 8 ? q->fed - kReadAhead - 8 : 0);
  uint32_t tile_to = (uint32_t)std::min<uint64_t>(q->n_tiles, std::min(tables_to, bytes_to) / kMatchTile);
  if (all) tile_to = q->n_tiles;
  if (tile_to > q->tile_done) {
    if (q->own_stream) {  // tiles whose tables were done in an earlier feed still need this feed's bytes (their read-ahead)
      SPZ_HIP_TRY(hipStreamWaitEvent(mst, q->copy_done, 0));
    }
    hipLaunchKernelGGL(lz_match_kernel, dim3(tile_to - q->tile_done), dim3(kMatchThreads), 0, mst, q->d_data(), q->d_link(), q->d_delta(), q->d_rank(), q->n_pos,
                       q->size, reinterpret_cast<uint2 *>(q->d_r128()), q->tile_done, lz_refill_lanes(),
                       std::getenv("SPZ_AMD_LZ_WALK_STATS") ? reinterpret_cast<unsigned long long *>(q->block + q->o_stats) : nullptr);
    SPZ_HIP_TRY(hipGetLastError());
    q->tile_done = tile_to;
  }
  // (The lazy state machine of the jobs whose match tables are enqueued was fed here as well, for one session of
  // measurements: its single-wave workgroups run for milliseconds and each holds 16 KiB of a CU's LDS, which keeps the
  // 144 KiB match workgroups and the 128 KiB table workgroups off that CU — the pack went from 85 ms to 118 ms with the
  // jobs in the match stream and to 161 ms with a stream of their own.  The jobs run after the last tile again, 10 ms.)
  return SPZ_AMD_OK;
}

}  // extern "C" (reopened below: the helpers above have C++ linkage)

extern "C" {

int spz_amd_zlib_session_open(uint64_t size, int device, void **session) {
  if (session == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  *session = nullptr;
  if (size < 16ull * W || size >= (1ull << 32) - 2ull * W) return SPZ_AMD_ERR_UNSUPPORTED;  // compressWithHeadParser's range
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  LzSession *q = nullptr;
  rc = lz_session_create(size, lz_tail_begin(size), device, /*own_stream=*/true, &q);
  if (rc != SPZ_AMD_OK) return rc;
  *session = q;
  return SPZ_AMD_OK;
}

int spz_amd_zlib_session_feed(void *session, const uint8_t *d_stream, uint64_t final_upto, void *producer_stream) {
  LzSession *q = static_cast<LzSession *>(session);
  if (q == nullptr || d_stream == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  DeviceGuard guard;
  int rc = guard.enter(q->device);
  if (rc != SPZ_AMD_OK) return rc;
  return lz_session_feed(q, d_stream, nullptr, final_upto, static_cast<hipStream_t>(producer_stream));
}

void spz_amd_zlib_session_close(void *session) { lz_session_destroy(static_cast<LzSession *>(session)); }

static int parse_open_impl(LzSession *q, const uint8_t *h_data, const uint8_t *d_copy, const uint32_t *h_tail_rec, uint32_t n_rec, void **ctx,
                           uint64_t *num_symbols, uint32_t *tail_first_symbol, void (*produce_tail_rec)(void *), void *produce_arg);

int spz_amd_zlib_parse_open_session(void *session, const uint8_t *h_data, const uint8_t *d_stream, uint64_t size, uint64_t tail_begin,
                                    const uint32_t *h_tail_rec, uint32_t n_rec, void **ctx, uint64_t *num_symbols,
                                    uint32_t *tail_first_symbol, void (*produce_tail_rec)(void *), void *produce_arg) {
  LzSession *q = static_cast<LzSession *>(session);
  if (q == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  if (h_data == nullptr || h_tail_rec == nullptr || ctx == nullptr || num_symbols == nullptr || tail_first_symbol == nullptr ||
      n_rec < kTailWindow || size != q->size || tail_begin != q->tail_begin) {
    lz_session_destroy(q);
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  *ctx = nullptr;
  DeviceGuard guard;
  int rc = guard.enter(q->device);
  if (rc != SPZ_AMD_OK) {
    lz_session_destroy(q);
    return rc;
  }
  return parse_open_impl(q, h_data, d_stream, h_tail_rec, n_rec, ctx, num_symbols, tail_first_symbol, produce_tail_rec, produce_arg);
}

int spz_amd_zlib_parse_open_dev(const uint8_t *h_data, const uint8_t *d_copy, uint64_t size, uint64_t tail_begin,
                                const uint32_t *h_tail_rec, uint32_t n_rec, int device, void **ctx, uint64_t *num_symbols,
                                uint32_t *tail_first_symbol, void (*produce_tail_rec)(void *), void *produce_arg) {
  if (h_data == nullptr || h_tail_rec == nullptr || ctx == nullptr || num_symbols == nullptr ||
      tail_first_symbol == nullptr) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  *ctx = nullptr;
  if (n_rec < kTailWindow) return SPZ_AMD_ERR_INVALID_ARG;
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  LzSession *q = nullptr;
  rc = lz_session_create(size, tail_begin, device, /*own_stream=*/false, &q);
  if (rc != SPZ_AMD_OK) return rc;
  return parse_open_impl(q, h_data, d_copy, h_tail_rec, n_rec, ctx, num_symbols, tail_first_symbol, produce_tail_rec, produce_arg);
}

// The rest of the parse on a session (consumed: its block goes to the context or back to the cache).
static int parse_open_impl(LzSession *q, const uint8_t *h_data, const uint8_t *d_copy, const uint32_t *h_tail_rec, uint32_t n_rec, void **ctx,
                           uint64_t *num_symbols, uint32_t *tail_first_symbol, void (*produce_tail_rec)(void *), void *produce_arg) {
  (void)n_rec;
  struct Drop {
    LzSession *q;
    ~Drop() { lz_session_destroy(q); }
  } drop{q};
  const int device = q->device;
  const uint64_t size = q->size, tail_begin = q->tail_begin, n_pos = q->n_pos;
  const size_t pos_padded = q->pos_padded, rec_words = q->rec_words;
  const size_t o_rec = q->o_rec, o_sd = q->o_sd, o_xd = q->o_xd, o_info = q->o_info;
  char *block = q->block;
  int rc = SPZ_AMD_OK;
  static const bool timing = std::getenv("SPZ_AMD_LZ_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[lz77] %-10s %.4f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  if (timing && q->own_stream) {
    std::fprintf(stderr, "[lz77] fed beside the upload: %u of %u table segments, %u of %u match tiles enqueued so far\n", q->seg_done, q->n_seg,
                 q->tile_done, q->n_tiles);
  }
  uint8_t *d_data = q->d_data();
  uint32_t *d_r128 = q->d_r128(), *d_r32 = q->d_r32();
  uint32_t *d_rec = reinterpret_cast<uint32_t *>(block + o_rec);
  uint32_t *d_sym = reinterpret_cast<uint32_t *>(block + q->o_sd);
  uint16_t *d_xd = reinterpret_cast<uint16_t *>(block + q->o_xd);
  uint8_t *d_xl = reinterpret_cast<uint8_t *>(block + q->o_xl);
  JobInfo *d_info = reinterpret_cast<JobInfo *>(block + q->o_info);
  unsigned long long *d_goff = reinterpret_cast<unsigned long long *>(block + q->o_goff);
  // what the feeds have not covered yet (everything, for a session fed once)
  rc = lz_session_feed(q, d_copy, d_copy ? nullptr : h_data, size, q->stream);
  if (rc != SPZ_AMD_OK) return rc;
  // the caller's tail parse can run now, beside the kernels above: its records are not read before this point
  if (produce_tail_rec) {
    const auto t0 = std::chrono::steady_clock::now();
    produce_tail_rec(produce_arg);
    if (timing) std::fprintf(stderr, "[lz77] the caller's tail parse %.4f s (beside the kernels)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  std::vector<uint32_t> tail_states(kTailWindow);
  for (uint32_t k = 0; k < kTailWindow; ++k) tail_states[k] = h_tail_rec[2 * k];
  if (q->own_stream) {  // the later stages run on the default stream
    SPZ_HIP_TRY(hipStreamSynchronize(q->stream));
    SPZ_HIP_TRY(hipStreamSynchronize(q->rank_stream));
    SPZ_HIP_TRY(hipStreamSynchronize(q->match_stream));
  }
  lap("tables+matches");
  if (std::getenv("SPZ_AMD_LZ_WALK_STATS")) {  // (the counting itself costs the match kernel a third of its time)
    unsigned long long h[3] = {0, 0, 0};
    SPZ_HIP_TRY(hipMemcpy(h, q->block + q->o_stats, sizeof(h), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "[lz77] walks: %llu positions, %llu candidates (%.1f each), %llu rounds of a wave (%.1f per tile and wave, longest %llu), %.0f %% of the lanes at work\n",
                 (unsigned long long)q->n_pos, h[1], (double)h[1] / (double)q->n_pos, h[0], (double)h[0] / ((double)q->n_tiles * (kMatchThreads / 64)), h[2],
                 h[0] ? 100.0 * (double)h[1] / (64.0 * (double)h[0]) : 0.0);
  }
  hipStream_t st = nullptr;
  // stage 3, with larger jobs if two neighbours do not meet (the tables do not depend on the jobs)
  uint32_t n_jobs = 0, job_bytes = 0;
  std::vector<JobInfo> info;
  std::vector<unsigned long long> goff;
  unsigned long long total_syms = 0;
  bool met = false;
  for (uint32_t attempt = 0; attempt < sizeof(kJobSizes) / sizeof(kJobSizes[0]) && !met; ++attempt) {
    job_bytes = kJobSizes[attempt];
    n_jobs = (uint32_t)((tail_begin + job_bytes - 1) / job_bytes);
    const uint32_t parse_blocks = (n_jobs + 63u) / 64u;
    SPZ_HIP_TRY(hipMemsetAsync(d_rec, 0, rec_words * sizeof(uint32_t), st));
    SPZ_HIP_TRY(hipMemcpyAsync(d_rec + tail_begin, tail_states.data(), (size_t)kTailWindow * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    SPZ_HIP_TRY(hipMemsetAsync(d_info, 0, (size_t)(n_jobs + 1) * sizeof(JobInfo), st));
    hipLaunchKernelGGL(lz_parse_kernel, dim3(parse_blocks), dim3(64), 0, st, reinterpret_cast<const uint2 *>(d_r128), job_bytes, (uint32_t)tail_begin, n_jobs, d_rec,
                       d_sym, d_info, 0u);
    SPZ_HIP_TRY(hipGetLastError());
    lap("parse");
    hipLaunchKernelGGL(lz_stitch_kernel, dim3(parse_blocks), dim3(64), 0, st, reinterpret_cast<const uint2 *>(d_r128), job_bytes, (uint32_t)tail_begin, n_jobs,
                       d_rec, d_xd, d_xl, d_info);
    SPZ_HIP_TRY(hipGetLastError());
    info.resize(n_jobs + 1);
    SPZ_HIP_TRY(hipMemcpyAsync(info.data(), d_info, info.size() * sizeof(JobInfo), hipMemcpyDeviceToHost, st));
    SPZ_HIP_TRY(hipStreamSynchronize(st));
    lap("stitch");
    met = true;
    goff.assign(n_jobs, 0);
    total_syms = 0;
    for (uint32_t j = 0; j < n_jobs; ++j) {
      if (!info[j].spliced || info[j].n < info[j].lo) {
        met = false;
        break;
      }
      goff[j] = total_syms;
      total_syms += info[j].n - info[j].lo + info[j].extra;
    }
    if (timing && !met) std::fprintf(stderr, "[lz77] jobs of %u bytes: two neighbours did not meet\n", job_bytes);
  }
  if (!met) return SPZ_AMD_ERR_UNSUPPORTED;  // no meeting point even between the largest jobs: host parse
  if (total_syms > n_pos) return SPZ_AMD_ERR_UNSUPPORTED;
  {  // the last job met the host's tail parse: the host's own symbol count at that loop top must be the one counted here
    const uint32_t k = info[n_jobs - 1].end_s - (uint32_t)tail_begin;
    if (k >= kTailWindow || h_tail_rec[2 * k + 1] != info[n_jobs].lo) return SPZ_AMD_ERR_UNSUPPORTED;
  }
  SPZ_HIP_TRY(hipMemcpyAsync(d_goff, goff.data(), goff.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
  // the tables are done with: their memory takes the dense arrays (2 B and 1 B per symbol, at most one symbol per position)
  uint16_t *dense_dist = reinterpret_cast<uint16_t *>(d_r128);
  uint8_t *dense_lc = reinterpret_cast<uint8_t *>(d_r32);
  hipLaunchKernelGGL(lz_compact_kernel, dim3(n_jobs), dim3(256), 0, st, d_sym, d_xd, d_xl, d_info, job_bytes, d_goff, dense_dist,
                     dense_lc);
  SPZ_HIP_TRY(hipGetLastError());
  SPZ_HIP_TRY(hipStreamSynchronize(st));
  lap("compact");
  LzContext *c = new (std::nothrow) LzContext();
  if (c == nullptr) return SPZ_AMD_ERR_HIP;
  c->device = device;
  c->block = block;
  q->block = nullptr;  // the context's now; the session's stream and event go with the session
  c->dense_dist = dense_dist;
  c->dense_lc = dense_lc;
  c->num_symbols = total_syms;
  c->capacity = std::min<uint64_t>(pos_padded * sizeof(uint32_t) / sizeof(uint16_t), pos_padded * sizeof(uint32_t));
  c->data = d_data;
  c->size = size;
  c->scratch_a = block + o_rec;   // the deflate body
  c->scratch_a_bytes = rec_words * sizeof(uint32_t);
  c->scratch_b = block + o_sd;    // codes, block descriptors, header words
  c->scratch_b_bytes = o_xd - o_sd;
  c->scratch_c = block + o_xd;    // frequencies, per-block counters, static tables
  c->scratch_c_bytes = o_info - o_xd;
  *ctx = c;
  *num_symbols = total_syms;
  *tail_first_symbol = info[n_jobs].lo;
  return SPZ_AMD_OK;
}

int spz_amd_zlib_parse_open_ex(const uint8_t *h_data, uint64_t size, uint64_t tail_begin, const uint32_t *h_tail_rec,
                               uint32_t n_rec, int device, void **ctx, uint64_t *num_symbols, uint32_t *tail_first_symbol,
                               void (*produce_tail_rec)(void *), void *produce_arg) {
  return spz_amd_zlib_parse_open_dev(h_data, nullptr, size, tail_begin, h_tail_rec, n_rec, device, ctx, num_symbols, tail_first_symbol,
                                     produce_tail_rec, produce_arg);
}

int spz_amd_zlib_parse_open(const uint8_t *h_data, uint64_t size, uint64_t tail_begin, const uint32_t *h_tail_rec,
                            uint32_t n_rec, int device, void **ctx, uint64_t *num_symbols,
                            uint32_t *tail_first_symbol) {
  return spz_amd_zlib_parse_open_ex(h_data, size, tail_begin, h_tail_rec, n_rec, device, ctx, num_symbols, tail_first_symbol,
                                    nullptr, nullptr);
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
  if (guard.enter(c->device) == SPZ_AMD_OK) {
    if (c->n_parts > 0) (void)hipDeviceSynchronize();  // an encode_planned nobody finished: nothing may still use the block
    copy_lane_release(c->lane);
    if (c->block) scratch_release(c->device, c->block);
  }
  delete c;
}

int spz_amd_zlib_parse_append(void *ctx, const uint16_t *h_dist, const uint8_t *h_lc, uint64_t n) {
  LzContext *c = static_cast<LzContext *>(ctx);
  if (c == nullptr || (n > 0 && (h_dist == nullptr || h_lc == nullptr))) return SPZ_AMD_ERR_INVALID_ARG;
  if (c->num_symbols + n > c->capacity) return SPZ_AMD_ERR_CAPACITY;
  if (n == 0) return SPZ_AMD_OK;
  DeviceGuard guard;
  int rc = guard.enter(c->device);
  if (rc != SPZ_AMD_OK) return rc;
  SPZ_HIP_TRY(hipMemcpyAsync(c->dense_dist + c->num_symbols, h_dist, n * sizeof(uint16_t), hipMemcpyHostToDevice, nullptr));
  SPZ_HIP_TRY(hipMemcpyAsync(c->dense_lc + c->num_symbols, h_lc, n, hipMemcpyHostToDevice, nullptr));
  SPZ_HIP_TRY(hipStreamSynchronize(nullptr));
  c->num_symbols += n;
  return SPZ_AMD_OK;
}

// Layout of the block statistics in scratch_c.
namespace {
struct StatArrays {
  spz_amd_deflate_static *tables;
  uint16_t *lfreq, *dfreq;
  uint32_t *bytes, *last;
};
static bool stat_arrays(LzContext *c, uint32_t num_blocks, StatArrays *a) {
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t at = off;
    off += round_up(bytes, 256);
    return c->scratch_c + at;
  };
  a->tables = reinterpret_cast<spz_amd_deflate_static *>(carve(sizeof(spz_amd_deflate_static)));
  a->lfreq = reinterpret_cast<uint16_t *>(carve((size_t)num_blocks * 286 * sizeof(uint16_t)));
  a->dfreq = reinterpret_cast<uint16_t *>(carve((size_t)num_blocks * 30 * sizeof(uint16_t)));
  a->bytes = reinterpret_cast<uint32_t *>(carve((size_t)num_blocks * sizeof(uint32_t)));
  a->last = reinterpret_cast<uint32_t *>(carve((size_t)num_blocks * sizeof(uint32_t)));
  return off <= c->scratch_c_bytes;
}
}  // namespace

int spz_amd_zlib_block_stats(void *ctx, const spz_amd_deflate_static *tables, uint32_t block_symbols, uint32_t num_blocks,
                             uint16_t *h_lfreq, uint16_t *h_dfreq, uint32_t *h_bytes, uint32_t *h_last_len) {
  LzContext *c = static_cast<LzContext *>(ctx);
  if (c == nullptr || tables == nullptr || block_symbols == 0 || num_blocks == 0 || h_bytes == nullptr || h_last_len == nullptr ||
      (h_lfreq == nullptr) != (h_dfreq == nullptr)) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  if ((uint64_t)num_blocks * block_symbols < c->num_symbols) return SPZ_AMD_ERR_INVALID_ARG;
  DeviceGuard guard;
  int rc = guard.enter(c->device);
  if (rc != SPZ_AMD_OK) return rc;
  StatArrays a;
  if (!stat_arrays(c, num_blocks, &a)) return SPZ_AMD_ERR_CAPACITY;
  hipStream_t st = nullptr;
  // Fault injection for the tests of the symbol check (tests/test_gpu_gzip_device.py): SPZ_AMD_TEST_CORRUPT_SYMBOL=<index>
  // changes that symbol BEFORE the statistics, so trees, plan and bit counts are all consistent with the wrong symbol
  // and only a check against the input can notice.  A literal becomes another literal; a match moves one byte nearer
  // (or farther, from distance 1).
  if (const char *e = std::getenv("SPZ_AMD_TEST_CORRUPT_SYMBOL")) {
    const uint64_t at = std::strtoull(e, nullptr, 10) % (c->num_symbols ? c->num_symbols : 1);
    uint16_t d = 0;
    uint8_t l = 0;
    SPZ_HIP_TRY(hipMemcpy(&d, c->dense_dist + at, sizeof(d), hipMemcpyDeviceToHost));
    SPZ_HIP_TRY(hipMemcpy(&l, c->dense_lc + at, sizeof(l), hipMemcpyDeviceToHost));
    if (d == 0) l ^= 1u;
    else d = d > 1 ? d - 1 : d + 1;
    SPZ_HIP_TRY(hipMemcpy(c->dense_dist + at, &d, sizeof(d), hipMemcpyHostToDevice));
    SPZ_HIP_TRY(hipMemcpy(c->dense_lc + at, &l, sizeof(l), hipMemcpyHostToDevice));
  }
  SPZ_HIP_TRY(hipMemcpyAsync(a.tables, tables, sizeof(*tables), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(lz_block_stats_kernel, dim3(num_blocks), dim3(256), 0, st, c->dense_dist, c->dense_lc,
                     (unsigned long long)c->num_symbols, block_symbols, a.tables, a.lfreq, a.dfreq, a.bytes, a.last);
  SPZ_HIP_TRY(hipGetLastError());
  if (h_lfreq != nullptr) {  // a caller that builds the trees itself
    SPZ_HIP_TRY(hipMemcpyAsync(h_lfreq, a.lfreq, (size_t)num_blocks * 286 * sizeof(uint16_t), hipMemcpyDeviceToHost, st));
    SPZ_HIP_TRY(hipMemcpyAsync(h_dfreq, a.dfreq, (size_t)num_blocks * 30 * sizeof(uint16_t), hipMemcpyDeviceToHost, st));
  }
  SPZ_HIP_TRY(hipMemcpyAsync(h_bytes, a.bytes, (size_t)num_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  SPZ_HIP_TRY(hipMemcpyAsync(h_last_len, a.last, (size_t)num_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  SPZ_HIP_TRY(hipStreamSynchronize(st));
  c->stats_blocks = num_blocks;
  c->trees_blocks = 0;
  return SPZ_AMD_OK;
}

// Layout of the Huffman stage's device arrays in the memory the parse has left behind.
namespace {
struct EncodeArrays {
  spz_amd_deflate_static *tables;
  spz_amd_deflate_block *blocks;
  spz_amd_deflate_codes *codes;
  unsigned long long *bits;
  BlockTrees *trees;
  spz_amd_deflate_plan *plan;
  uint32_t *invalid;       // lz_validate_kernel's count of blocks' threads that found a symbol wrong
  uint32_t *header_bits;   // every block's header length, dense (what encode_finish_ex hands back)
  uint32_t *header;
  size_t header_capacity;  // words
};
static bool encode_arrays(LzContext *c, uint32_t total_blocks, EncodeArrays *a) {
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t at = off;
    off += round_up(bytes, 256);
    return c->scratch_b + at;
  };
  a->tables = reinterpret_cast<spz_amd_deflate_static *>(carve(sizeof(spz_amd_deflate_static)));
  a->blocks = reinterpret_cast<spz_amd_deflate_block *>(carve((size_t)total_blocks * sizeof(spz_amd_deflate_block)));
  a->codes = reinterpret_cast<spz_amd_deflate_codes *>(carve((size_t)total_blocks * sizeof(spz_amd_deflate_codes)));
  a->bits = reinterpret_cast<unsigned long long *>(carve((size_t)total_blocks * sizeof(unsigned long long)));
  a->trees = reinterpret_cast<BlockTrees *>(carve((size_t)total_blocks * sizeof(BlockTrees)));
  a->plan = reinterpret_cast<spz_amd_deflate_plan *>(carve((size_t)total_blocks * sizeof(spz_amd_deflate_plan)));
  a->invalid = reinterpret_cast<uint32_t *>(carve(sizeof(uint32_t)));
  a->header_bits = reinterpret_cast<uint32_t *>(carve((size_t)total_blocks * sizeof(uint32_t)));
  if (off + 4096 > c->scratch_b_bytes) return false;
  a->header = reinterpret_cast<uint32_t *>(c->scratch_b + off);
  a->header_capacity = (c->scratch_b_bytes - off) / sizeof(uint32_t);
  return true;
}
}  // namespace

int spz_amd_zlib_encode_group(void *ctx, const spz_amd_deflate_static *tables, uint32_t block_symbols, uint32_t total_blocks,
                              uint32_t first_block, uint32_t group_blocks, const spz_amd_deflate_block *h_blocks,
                              const spz_amd_deflate_codes *h_codes, const uint32_t *h_header_words,
                              uint64_t num_header_words, uint64_t body_bytes_bound) {
  LzContext *c = static_cast<LzContext *>(ctx);
  if (c == nullptr || tables == nullptr || block_symbols == 0 || group_blocks == 0 || h_blocks == nullptr || h_codes == nullptr ||
      (num_header_words > 0 && h_header_words == nullptr) || (uint64_t)first_block + group_blocks > total_blocks) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  if ((uint64_t)total_blocks * block_symbols < c->num_symbols) return SPZ_AMD_ERR_INVALID_ARG;
  const size_t body_words = (size_t)((body_bytes_bound + 3) / 4) + 2;
  if (body_words * 4 > c->scratch_a_bytes) return SPZ_AMD_ERR_CAPACITY;
  for (uint32_t b = 0; b < group_blocks; ++b) {  // nothing the kernel writes may leave the body
    const spz_amd_deflate_block &k = h_blocks[b];
    if ((uint64_t)k.header_word_begin + k.header_words > num_header_words || (k.bit_start >> 5) + k.header_words > body_words ||
        (k.choice == 0 && (((k.bit_start + k.header_bits) & 7) != 0 ||
                           ((k.bit_start + k.header_bits) >> 3) + k.input_bytes > body_bytes_bound ||
                           (uint64_t)k.input_begin + k.input_bytes > c->size))) {
      return SPZ_AMD_ERR_INVALID_ARG;
    }
  }
  DeviceGuard guard;
  int rc = guard.enter(c->device);
  if (rc != SPZ_AMD_OK) return rc;
  EncodeArrays a;
  if (!encode_arrays(c, total_blocks, &a) || num_header_words > a.header_capacity) return SPZ_AMD_ERR_CAPACITY;
  uint32_t *d_body = reinterpret_cast<uint32_t *>(c->scratch_a);
  hipStream_t st = nullptr;
  if (first_block == 0) {
    c->validated_blocks = 0;
    SPZ_HIP_TRY(hipMemsetAsync(d_body, 0, body_words * 4, st));
    SPZ_HIP_TRY(hipMemcpyAsync(a.tables, tables, sizeof(*tables), hipMemcpyHostToDevice, st));
    SPZ_HIP_TRY(hipMemsetAsync(a.invalid, 0, sizeof(uint32_t), st));
  }
  SPZ_HIP_TRY(hipMemcpyAsync(a.blocks + first_block, h_blocks, (size_t)group_blocks * sizeof(spz_amd_deflate_block), hipMemcpyHostToDevice, st));
  SPZ_HIP_TRY(hipMemcpyAsync(a.codes + first_block, h_codes, (size_t)group_blocks * sizeof(spz_amd_deflate_codes), hipMemcpyHostToDevice, st));
  if (num_header_words) {
    SPZ_HIP_TRY(hipMemcpyAsync(a.header, h_header_words, (size_t)num_header_words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  }
  hipLaunchKernelGGL(lz_encode_kernel, dim3(group_blocks), dim3(kEncodeThreads), 0, st, c->dense_dist, c->dense_lc,
                     (unsigned long long)c->num_symbols, block_symbols, c->data, a.tables, a.blocks, a.codes, a.header, d_body, a.bits,
                     first_block);
  SPZ_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(lz_validate_kernel, dim3(group_blocks), dim3(256), 0, st, c->dense_dist, c->dense_lc, (unsigned long long)c->num_symbols,
                     block_symbols, c->data, (unsigned long long)c->size, a.blocks, first_block, a.invalid);
  SPZ_HIP_TRY(hipGetLastError());
  c->validated_blocks += group_blocks;
  return SPZ_AMD_OK;  // not waited for: the next group's trees can be built meanwhile
}

int spz_amd_zlib_encode_finish_ex(void *ctx, uint32_t total_blocks, uint64_t body_bytes, uint8_t *h_body, uint64_t *h_symbol_bits,
                                  uint32_t *h_header_bits) {
  LzContext *c = static_cast<LzContext *>(ctx);
  if (c == nullptr || h_body == nullptr || h_symbol_bits == nullptr || total_blocks == 0) return SPZ_AMD_ERR_INVALID_ARG;
  if (((body_bytes + 3) / 4 + 2) * 4 > c->scratch_a_bytes) return SPZ_AMD_ERR_CAPACITY;
  DeviceGuard guard;
  int rc = guard.enter(c->device);
  if (rc != SPZ_AMD_OK) return rc;
  EncodeArrays a;
  if (!encode_arrays(c, total_blocks, &a)) return SPZ_AMD_ERR_CAPACITY;
  hipStream_t st = nullptr;
  static const bool timing = std::getenv("SPZ_AMD_LZ_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  // lz_validate_kernel's verdict on the symbols the body was coded from: a body is handed out only when every block
  // was checked and none had a symbol that does not reproduce the input
  uint32_t invalid = 1;
  auto symbols_valid = [&]() {
    if (c->validated_blocks != total_blocks || invalid != 0) {
      if (timing) std::fprintf(stderr, "[lz77] symbol check FAILED: %u of %u blocks checked, %u findings\n", c->validated_blocks, total_blocks, invalid);
      return (int)SPZ_AMD_ERR_VERIFY;
    }
    return (int)SPZ_AMD_OK;
  };
  if (c->n_parts > 0) {  // after encode_planned: every part as soon as its launch is done
    uint64_t from = 0;
    for (int g = 0; g < c->n_parts; ++g) {
      const uint64_t to = c->part_end[g] < body_bytes ? c->part_end[g] : body_bytes;
      SPZ_HIP_TRY(hipStreamWaitEvent(c->lane->stream, c->lane->part_done[g], 0));
      if (to > from) SPZ_HIP_TRY(hipMemcpyAsync(h_body + from, c->scratch_a + from, to - from, hipMemcpyDeviceToHost, c->lane->stream));
      if (to > from) from = to;
    }
    c->n_parts = 0;
    if (from != body_bytes) {
      (void)hipDeviceSynchronize();
      return SPZ_AMD_ERR_INVALID_ARG;  // not the body encode_planned was given
    }
    SPZ_HIP_TRY(hipMemcpyAsync(h_symbol_bits, a.bits, (size_t)total_blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->lane->stream));
    if (h_header_bits != nullptr) {  // dense: a strided hipMemcpy2DAsync out of the block kept the block's memory after hipFree
      if (!c->header_bits_dense) return SPZ_AMD_ERR_INVALID_ARG;
      SPZ_HIP_TRY(hipMemcpyAsync(h_header_bits, a.header_bits, (size_t)total_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, c->lane->stream));
    }
    SPZ_HIP_TRY(hipStreamSynchronize(c->lane->stream));
    SPZ_HIP_TRY(hipMemcpyAsync(&invalid, a.invalid, sizeof(invalid), hipMemcpyDeviceToHost, st));  // behind lz_validate_kernel
    SPZ_HIP_TRY(hipStreamSynchronize(st));
    if (timing) {
      std::fprintf(stderr, "[lz77] encode + download %.4f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    return symbols_valid();
  }
  if (timing) {
    (void)hipStreamSynchronize(st);
    std::fprintf(stderr, "[lz77] encode (what was still running) %.4f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  SPZ_HIP_TRY(hipMemcpyAsync(h_body, c->scratch_a, body_bytes, hipMemcpyDeviceToHost, st));
  SPZ_HIP_TRY(hipMemcpyAsync(h_symbol_bits, a.bits, (size_t)total_blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  if (h_header_bits != nullptr) {
    if (!c->header_bits_dense) return SPZ_AMD_ERR_INVALID_ARG;  // the header lengths are the device's only after encode_planned
    SPZ_HIP_TRY(hipMemcpyAsync(h_header_bits, a.header_bits, (size_t)total_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  }
  SPZ_HIP_TRY(hipMemcpyAsync(&invalid, a.invalid, sizeof(invalid), hipMemcpyDeviceToHost, st));
  SPZ_HIP_TRY(hipStreamSynchronize(st));
  if (timing) {
    std::fprintf(stderr, "[lz77] + download %.4f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  return symbols_valid();
}

int spz_amd_zlib_encode_finish(void *ctx, uint32_t total_blocks, uint64_t body_bytes, uint8_t *h_body, uint64_t *h_symbol_bits) {
  return spz_amd_zlib_encode_finish_ex(ctx, total_blocks, body_bytes, h_body, h_symbol_bits, nullptr);
}

int spz_amd_zlib_block_trees(void *ctx, uint32_t num_blocks, spz_amd_deflate_plan *h_plan) {
  LzContext *c = static_cast<LzContext *>(ctx);
  if (c == nullptr || num_blocks == 0 || h_plan == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  if (c->stats_blocks != num_blocks) return SPZ_AMD_ERR_INVALID_ARG;  // block_stats first, for the same blocks
  DeviceGuard guard;
  int rc = guard.enter(c->device);
  if (rc != SPZ_AMD_OK) return rc;
  StatArrays sa;
  EncodeArrays a;
  if (!stat_arrays(c, num_blocks, &sa) || !encode_arrays(c, num_blocks, &a)) return SPZ_AMD_ERR_CAPACITY;
  hipStream_t st = nullptr;
  hipLaunchKernelGGL(lz_tree_kernel, dim3((num_blocks + kTreeLanes - 1) / kTreeLanes), dim3(64), 0, st, sa.lfreq, sa.dfreq, num_blocks,
                     a.codes, a.trees, a.plan);
  SPZ_HIP_TRY(hipGetLastError());
  SPZ_HIP_TRY(hipMemcpyAsync(h_plan, a.plan, (size_t)num_blocks * sizeof(spz_amd_deflate_plan), hipMemcpyDeviceToHost, st));
  SPZ_HIP_TRY(hipStreamSynchronize(st));
  c->trees_blocks = num_blocks;
  return SPZ_AMD_OK;
}

int spz_amd_zlib_encode_planned(void *ctx, const spz_amd_deflate_static *tables, uint32_t block_symbols, uint32_t num_blocks,
                                const spz_amd_deflate_block *h_blocks, uint64_t body_bytes) {
  LzContext *c = static_cast<LzContext *>(ctx);
  if (c == nullptr || tables == nullptr || block_symbols == 0 || num_blocks == 0 || h_blocks == nullptr) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  if (c->trees_blocks != num_blocks) return SPZ_AMD_ERR_INVALID_ARG;  // block_trees first: the codes are its output
  if ((uint64_t)num_blocks * block_symbols < c->num_symbols) return SPZ_AMD_ERR_INVALID_ARG;
  const size_t body_words = (size_t)((body_bytes + 3) / 4) + 2;
  if (body_words * 4 > c->scratch_a_bytes) return SPZ_AMD_ERR_CAPACITY;
  // nothing the kernels write may leave scratch_a: a header takes at most kHeaderStride words from its block's first
  // word, a block's symbols at most 48 bits each
  if (body_words * 4 + kHeaderStride * 4 + (size_t)block_symbols * 6 + 64 > c->scratch_a_bytes) return SPZ_AMD_ERR_CAPACITY;
  for (uint32_t b = 0; b < num_blocks; ++b) {
    const spz_amd_deflate_block &k = h_blocks[b];
    if (k.choice > 2u || (k.bit_start >> 5) >= body_words ||
        (k.choice == 0 && ((k.bit_start >> 3) + 8 + k.input_bytes > body_bytes + 8 || (uint64_t)k.input_begin + k.input_bytes > c->size ||
                           k.input_bytes > 0xffffu))) {
      return SPZ_AMD_ERR_INVALID_ARG;
    }
  }
  DeviceGuard guard;
  int rc = guard.enter(c->device);
  if (rc != SPZ_AMD_OK) return rc;
  EncodeArrays a;
  if (!encode_arrays(c, num_blocks, &a) || (size_t)num_blocks * kHeaderStride > a.header_capacity) return SPZ_AMD_ERR_CAPACITY;
  uint32_t *d_body = reinterpret_cast<uint32_t *>(c->scratch_a);
  hipStream_t st = nullptr;
  SPZ_HIP_TRY(hipMemsetAsync(d_body, 0, body_words * 4, st));
  SPZ_HIP_TRY(hipMemcpyAsync(a.tables, tables, sizeof(*tables), hipMemcpyHostToDevice, st));
  SPZ_HIP_TRY(hipMemcpyAsync(a.blocks, h_blocks, (size_t)num_blocks * sizeof(spz_amd_deflate_block), hipMemcpyHostToDevice, st));
  SPZ_HIP_TRY(hipMemsetAsync(a.invalid, 0, sizeof(uint32_t), st));
  hipLaunchKernelGGL(lz_header_kernel, dim3((num_blocks + 63) / 64), dim3(64), 0, st, a.blocks, a.codes, a.trees, a.header, a.header_bits,
                     num_blocks);
  c->header_bits_dense = true;
  SPZ_HIP_TRY(hipGetLastError());
  // in a few launches, so that the body's first parts are on their way to the host while the later ones are packed
  if (c->lane == nullptr) c->lane = copy_lane_acquire(c->device);
  const int parts = (c->lane != nullptr && num_blocks >= 4096u) ? LzContext::kBodyParts : 1;
  c->n_parts = 0;
  for (int g = 0; g < parts; ++g) {
    const uint32_t b0 = (uint32_t)((uint64_t)num_blocks * g / parts), b1 = (uint32_t)((uint64_t)num_blocks * (g + 1) / parts);
    hipLaunchKernelGGL(lz_encode_kernel, dim3(b1 - b0), dim3(kEncodeThreads), 0, st, c->dense_dist, c->dense_lc,
                       (unsigned long long)c->num_symbols, block_symbols, c->data, a.tables, a.blocks, a.codes, a.header, d_body, a.bits,
                       b0);
    SPZ_HIP_TRY(hipGetLastError());
    if (c->lane == nullptr) break;  // one launch, copied out by encode_finish_ex in one piece
    SPZ_HIP_TRY(hipEventRecord(c->lane->part_done[g], st));
    // a part ends with the last whole word before the next launch's first block: that block completes the word it starts in
    c->part_end[g] = b1 < num_blocks ? (h_blocks[b1].bit_start >> 5) * 4 : body_bytes;
    c->n_parts = g + 1;
  }
  // behind the bit packing, beside the copies of the body's parts: do the symbols reproduce the input?
  hipLaunchKernelGGL(lz_validate_kernel, dim3(num_blocks), dim3(256), 0, st, c->dense_dist, c->dense_lc, (unsigned long long)c->num_symbols,
                     block_symbols, c->data, (unsigned long long)c->size, a.blocks, 0u, a.invalid);
  SPZ_HIP_TRY(hipGetLastError());
  c->validated_blocks = num_blocks;
  return SPZ_AMD_OK;  // not waited for: the caller maps its output buffer meanwhile, then spz_amd_zlib_encode_finish_ex
}

int spz_amd_zlib_verify_member(void *ctx, uint64_t body_bytes) {
  LzContext *c = static_cast<LzContext *>(ctx);
  if (c == nullptr || body_bytes == 0 || body_bytes > c->scratch_a_bytes) return SPZ_AMD_ERR_INVALID_ARG;
  void *inf = nullptr;
  uint64_t out_bytes = 0;
  int rc = spz_amd_inflate_open_device(reinterpret_cast<const uint8_t *>(c->scratch_a), body_bytes, c->device, &inf, &out_bytes);
  if (rc != SPZ_AMD_OK) return rc;  // SPZ_AMD_ERR_UNSUPPORTED: the device reader declines this body; the caller checks on the host
  rc = spz_amd_inflate_equals_device(inf, c->data, c->size);
  spz_amd_inflate_close(inf);
  return rc;
}

int spz_amd_zlib_encode_blocks(void *ctx, const spz_amd_deflate_static *tables, uint32_t block_symbols, uint32_t num_blocks,
                               const spz_amd_deflate_block *h_blocks, const spz_amd_deflate_codes *h_codes,
                               const uint32_t *h_header_words, uint64_t num_header_words, uint64_t body_bytes,
                               uint8_t *h_body, uint64_t *h_symbol_bits) {
  if (h_body == nullptr || h_symbol_bits == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  const int rc = spz_amd_zlib_encode_group(ctx, tables, block_symbols, num_blocks, 0, num_blocks, h_blocks, h_codes, h_header_words,
                                           num_header_words, body_bytes);
  if (rc != SPZ_AMD_OK) return rc;
  return spz_amd_zlib_encode_finish(ctx, num_blocks, body_bytes, h_body, h_symbol_bits);
}

}  // extern "C"
