// spz_inflate_dev.hip — inflate of a single, ordinary deflate stream on the MI355X (include/spz_amd.h:
// spz_amd_inflate_*).  The files the reference writes are one zlib stream with no index (load-spz.cc:141-184 reads
// them with one inflate loop); a deflate stream can still be decoded in parallel (pugz, Kerbiriou & Chikhi 2019;
// the host version is spz_inflate.cpp), and the decoder source is the same here (spz_inflate_core.hpp):
//
//   inf_search_kernel   one workgroup per 64 KiB of compressed bytes: all threads test bit positions for a dynamic
//                       block header (type bits, code counts, a complete code-length code); thread 0 takes the
//                       candidates in order and accepts the first whose two code-length sets are valid.
//   inf_decode_kernel   one wave per chunk, its 64 lanes in lockstep (tables and 2 KiB of compressed bytes in LDS; a
//                       match is copied by as many lanes as it is long):
//                       from its block start to the next chunk's — they must link up bit-exactly — into 16-bit
//                       symbols: a byte, or 256 + k = "byte k of the 32 KiB before this chunk", which the chunk
//                       does not have.
//   inf_window_*        every chunk's final 32 KiB, first symbolically (gather), then every reference followed through the
//                       predecessors' windows to the byte it stands for (resolve: parallel over all entries).
//   inf_place_kernel    symbols -> bytes at their final offsets, references resolved from the predecessor's window.
//   inf_crc_kernel      CRC-32 of 256 KiB pieces, a wave each (the caller folds them with crc32_combine and compares with the
//                       member's trailer: the result can only be right or refused).
// Runs of stored blocks (incompressible sections) have no dynamic header to find: their blocks are found by their own
// pattern (00 LEN ~LEN at a byte boundary, leading to another header).  Raw bytes now and then read like a block header: the chunk before such a look-alike does not end on it, so the
// look-alike is dropped and the chunk resumes, from the block that ran past it, towards the next start.
// Anything else irregular (a reference where there is no predecessor, a chunk that expands more than 8x, data
// that opens with a stored block) declines: the caller's host readers take over.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "spz_amd.h"
#include "spz_common.hpp"
#include "spz_inflate_core.hpp"

namespace spz_amd_detail {
namespace {

using namespace spz::pinflate;

// Compressed bytes per chunk, at least (more when that lets every chunk's wave be resident at once).  About one block
// of zlib's (38 KB of an .spz stream): smaller chunks find the same block starts and become one job; 64 KiB — until
// round 3 — left a member of 40 - 200 MB with few, long jobs: a wave takes as long for 64 KiB whether the chip holds
// 600 of them or 5 000.
constexpr uint32_t kMinChunkBytes = 32768;
constexpr uint32_t kDecodeWavesPerCU = 23;  // what the decode kernel's registers and LDS allow
constexpr uint32_t kSearchBytes = 131072;  // how far past its chunk's first byte a block start is looked for
constexpr uint32_t kExpand = 8;            // symbols a chunk may produce per compressed byte
constexpr uint32_t kCrcPiece = 262144;
constexpr uint32_t kMaxCand = 192;
constexpr uint32_t kSearchPerThread = 64;  // bit positions a thread tests per sub-batch
constexpr uint32_t kSearchSubBatches = 4;  // sub-batches (of 256 x 64 positions) whose candidates are checked together
constexpr uint32_t kValidateSlots = 16;    // candidates whose code lengths are read at the same time (1.3 KiB of LDS each)

// Tables of the device decoder: 9- and 7-bit fast tables (4.7 + 1 KiB per wave: the decode is bound by the latency
// of its match copies, so what counts is how many waves a CU holds; longer codes take the canonical walk).
using DevLit = HuffT<9, 288>;
using DevDist = HuffT<7, 32>;
struct alignas(16) Tables {
  DevLit lit;
  DevDist dist;
};
constexpr uint32_t kWinBytes = 1024;  // compressed bytes a wave keeps in LDS around its bit position

// ---- search -----------------------------------------------------------------------------------------------
// 128 input bits in registers: what the cheap header test of 16 neighbouring bit positions needs (at most 74 bits each).
struct RegBits {
  uint64_t lo, hi;     // the bits from bit position `base` on
  uint64_t base, nbits;
  __device__ __forceinline__ uint32_t uniform(uint32_t v) const { return v; }
  __device__ __forceinline__ uint64_t peek(uint64_t at) const {
    const uint32_t s = (uint32_t)(at - base);  // < 128
    if (s == 0) return lo;
    if (s < 64) return (lo >> s) | (hi << (64u - s));
    return s < 128 ? hi >> (s - 64u) : 0ull;
  }
};

// The member's bits for one serial reader (a header check): 128 of them in registers, reloaded when the reader has
// moved 72 on — a check reads some 300 code lengths one after the other, and a load from global memory per length,
// each waited for, was three quarters of the search kernel's time.
struct SlidingBits {
  const uint8_t *p;
  uint64_t nbits;
  size_t nbytes;
  mutable uint64_t lo = 0, hi = 0, base = ~0ull;  // bits [base, base + 128), base a multiple of 8; ~0: nothing loaded
  __device__ __forceinline__ uint32_t uniform(uint32_t v) const { return v; }
  __device__ __forceinline__ uint64_t peek(uint64_t at) const {
    if (at < base || at - base > 72) {  // (the data is padded: 16 bytes can be read from any byte of it)
      base = at & ~7ull;
      __builtin_memcpy(&lo, p + (base >> 3), 8);
      __builtin_memcpy(&hi, p + (base >> 3) + 8, 8);
    }
    const uint32_t s = (uint32_t)(at - base);  // <= 72: 56 valid bits at least
    if (s == 0) return lo;
    if (s < 64) return (lo >> s) | (hi << (64u - s));
    return hi >> (s - 64u);
  }
};

__global__ __launch_bounds__(256) void inf_search_kernel(const uint8_t *__restrict__ d, uint64_t nbytes, uint32_t chunk_bytes,
                                                         uint32_t n_chunks, unsigned long long *__restrict__ starts,
                                                         unsigned long long *__restrict__ stats) {
  __shared__ unsigned long long cand[kMaxCand];
  __shared__ uint32_t ncand;
  __shared__ unsigned long long found;
  __shared__ HeaderWork s_work[kValidateSlots];  // the candidate checks' arrays (in scratch memory they were most of the kernel's time)
  const uint32_t tid = threadIdx.x, chunk = blockIdx.x + 1;  // chunk 0 starts at bit 0
  if (chunk >= n_chunks) return;
  const Bits in = {d, 8ull * nbytes, (size_t)nbytes};
  const uint64_t lo = 8ull * chunk_bytes * chunk;
  uint64_t hi = lo + 8ull * kSearchBytes;
  if (hi > in.nbits) hi = in.nbits;
  if (tid == 0) found = NONE;
  __syncthreads();
  long long t_mark = stats ? clock64() : 0, t_stored = 0, t_scan = 0, t_check = 0;
  // Inside a run of stored blocks (an incompressible section) there is no dynamic header to find, but the blocks start
  // at byte boundaries with 00 LEN ~LEN, one after the other: a position whose LEN leads to another such header (or to
  // the final one) is a block start.  Looked for in the chunk's own bytes only; the lowest wins against a dynamic
  // header found below.
  {
    const uint64_t b0 = (uint64_t)chunk_bytes * chunk;
    // sixteen bytes per lane and load (one byte per lane and load, each waited for, was a quarter of the kernel's time);
    // only a zero byte is looked at further (chunk_bytes is a multiple of 4096, the data is padded)
    for (uint32_t k16 = tid * 16u; k16 < chunk_bytes; k16 += 256u * 16u) {
      uint4 q;
      __builtin_memcpy(&q, d + b0 + k16, 16);
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
      // a byte of the word is zero: the classic (w - 0x01010101) & ~w & 0x80808080
      uint32_t zeros = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) zeros |= ((w[j] - 0x01010101u) & ~w[j] & 0x80808080u) ? (1u << j) : 0u;
      if (zeros == 0u) continue;
      for (uint32_t k = k16; k < k16 + 16u; ++k) {
        const uint64_t b = b0 + k;
        if (b + 5 > nbytes || d[b] != 0) continue;
        const uint32_t len = (uint32_t)d[b + 1] | ((uint32_t)d[b + 2] << 8), nlen = (uint32_t)d[b + 3] | ((uint32_t)d[b + 4] << 8);
        if ((len ^ nlen) != 0xffffu || len < 1024) continue;   // zlib's stored blocks carry tens of kilobytes
        const uint64_t b2 = b + 5 + len;
        if (b2 + 5 > nbytes) continue;
        const uint32_t h2 = d[b2], len2 = (uint32_t)d[b2 + 1] | ((uint32_t)d[b2 + 2] << 8),
                       nlen2 = (uint32_t)d[b2 + 3] | ((uint32_t)d[b2 + 4] << 8);
        const bool stored_next = (h2 == 0 || h2 == 1) && (len2 ^ nlen2) == 0xffffu;
        const bool dynamic_next = (h2 & 6u) == 4u && plausibleDynamicHeader(in, 8 * b2);
        if (stored_next || dynamic_next) atomicMin(&found, 8ull * b);
      }
    }
    __syncthreads();
  }
  if (stats) {
    const long long t = clock64();
    t_stored = t - t_mark;
    t_mark = t;
  }
  const unsigned long long stored_start = found;
  if (stored_start != NONE && stored_start < hi) hi = stored_start;   // a dynamic header counts only below it
  __syncthreads();
  // A batch is kSearchSubBatches x 16384 bit positions: its candidates for the full check are collected first, sorted, and
  // checked sixteen at a time in position order until one is a header — the lowest valid position is all that is wanted,
  // a round of full checks lasts as long as its longest walk (~200 k cycles) whether it has two candidates or sixteen,
  // and with a round per 16384 positions (7.5 candidates) a chunk paid ten rounds where four do.
  for (uint64_t base = lo; base < hi; base += 256ull * kSearchPerThread * kSearchSubBatches) {
    if (tid == 0) ncand = 0;
    __syncthreads();
    for (uint32_t sub = 0; sub < kSearchSubBatches; ++sub) {
      // In two passes, because the test's second part is a loop of up to 19 steps that one position in nine gets to: run
      // position by position, some lane of the wave got there at nearly every position and the other 56 waited (13 % of
      // the lanes at work, and this scan was the kernel's time).  First every lane marks which of its 64 positions have
      // the header's fields in range — 16 at a time: they start at a byte boundary, and 16 bytes from there cover the
      // fields (padded data); then the lanes go through their marked positions together, one each per step.
      const uint64_t first = base + (uint64_t)sub * 256u * kSearchPerThread + (uint64_t)tid * kSearchPerThread;
      if (first >= hi) break;  // (the same for every thread of a wave but its last positions': they test p < hi below)
      constexpr uint32_t kGroups = kSearchPerThread / 16;
      uint64_t glo[kGroups], ghi[kGroups];  // 128 bits from each group's first position: all requested before the first is used
#pragma unroll
      for (uint32_t g = 0; g < kGroups; ++g) {
        const uint64_t p0 = first + g * 16;
        __builtin_memcpy(&glo[g], d + (p0 >> 3), 8);
        __builtin_memcpy(&ghi[g], d + (p0 >> 3) + 8, 8);
      }
      uint64_t marked = 0;
#pragma unroll
      for (uint32_t g = 0; g < kGroups; ++g) {
        const uint64_t p0 = first + g * 16;
        const RegBits rb = {glo[g], ghi[g], p0, in.nbits};
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
          const uint64_t p = p0 + k;
          if (p < hi && p + 64 < in.nbits && dynamicHeaderFieldsInRange(rb.peek(p))) marked |= 1ull << (g * 16 + k);
        }
      }
      while (__any(marked != 0ull)) {
        if (marked != 0ull) {
          const uint32_t bit = (uint32_t)__builtin_ctzll(marked);
          marked &= marked - 1ull;
          const uint32_t g = bit >> 4;
          // the group's 128 bits hold the 74 from any of its positions on (15 + 74 < 128); picked by selects, not by an
          // index (an indexed register array would go to scratch memory)
          uint64_t lo = glo[0], hi2 = ghi[0];
#pragma unroll
          for (uint32_t j = 1; j < kGroups; ++j) {
            lo = g == j ? glo[j] : lo;
            hi2 = g == j ? ghi[j] : hi2;
          }
          const RegBits rb = {lo, hi2, first + (uint64_t)g * 16, in.nbits};
          const uint64_t p = first + bit;
          if (codeLengthCodeComplete(rb, p, rb.peek(p))) {
            const uint32_t slot = atomicAdd(&ncand, 1u);
            if (slot < kMaxCand) cand[slot] = p;
          }
        }
      }
    }
    __syncthreads();
    if (stats) {
      const long long t = clock64();
      t_scan += t - t_mark;
      t_mark = t;
    }
    // one candidate per thread: header and both code-length sets valid?  The lowest position that is wins.
    const uint32_t n = ncand < kMaxCand ? ncand : kMaxCand;
    {  // in position order (they were collected in any): a candidate's place is the number of lower ones
      unsigned long long mine = 0;
      uint32_t place = 0;
      if (tid < n) {
        mine = cand[tid];
        for (uint32_t j = 0; j < n; ++j) place += cand[j] < mine ? 1u : 0u;
      }
      __syncthreads();
      if (tid < n) cand[place] = mine;
      __syncthreads();
    }
    if (stats != nullptr && tid == 0) {  // SPZ_AMD_LZ_TIMING: batches of 16384 bit positions, and the candidates in them
      atomicAdd(&stats[0], 1ull);
      atomicAdd(&stats[1], (unsigned long long)ncand);
    }
    // kValidateSlots at a time (their arrays are in LDS), four to each of the workgroup's waves: a check is a serial
    // walk over up to 316 code lengths, a wave lasts as long as its slowest lane, and sixteen in ONE wave with the other
    // three waiting at the barrier was where the kernel's time went (13 % of its lanes at work)
    const uint32_t slot = (tid & 63u) < kValidateSlots / 4u ? (tid >> 6) * (kValidateSlots / 4u) + (tid & 63u) : kValidateSlots;
    for (uint32_t base2 = 0; base2 < n; base2 += kValidateSlots) {
      if (slot < kValidateSlots && base2 + slot < n) {
        const SlidingBits sb = {d, in.nbits, in.nbytes};
        s_work[slot].walked = 0;
        if (hasValidDynamicHeader(sb, cand[base2 + slot], &s_work[slot])) atomicMin(&found, cand[base2 + slot]);
        if (stats) atomicAdd(&stats[5], (unsigned long long)s_work[slot].walked);
      }
      __syncthreads();
      if (found != NONE) break;  // every later candidate is at a higher position
      if (stats != nullptr && tid == 0) atomicAdd(&stats[6], 1ull);
    }
    __syncthreads();
    if (stats) {
      const long long t = clock64();
      t_check += t - t_mark;
      t_mark = t;
    }
    if (found != NONE) break;
    if (ncand > kMaxCand) {  // too many look-alikes to be sure none was missed: no start for this chunk
      break;
    }
  }
  if (tid == 0) starts[chunk] = found;
  if (stats != nullptr && tid == 0) {
    atomicAdd(&stats[2], (unsigned long long)t_stored);
    atomicAdd(&stats[3], (unsigned long long)t_scan);
    atomicAdd(&stats[4], (unsigned long long)t_check);
  }
}

// ---- decode -----------------------------------------------------------------------------------------------
struct ChunkJob {
  unsigned long long from, to;      // bit positions: first block, the next chunk's first block (NONE: to the final block)
  unsigned long long region;        // first symbol of the chunk's region
  unsigned long long capacity;      // symbols
  unsigned long long start_n;       // symbols of the region that an earlier attempt has already produced (kept)
};
struct ChunkResult {
  unsigned long long length;        // symbols produced
  unsigned long long end_bit;       // FINAL: first bit after the final block
  uint32_t outcome;                 // Outcome
  uint32_t overflow;
  unsigned long long mark_n;        // FAILED: symbols before the block that did not work out (end_bit: where it begins)
  uint32_t n_rounds, n_matches, n_miss, pad;
};
// The wave's view of the deflate data: all 64 lanes run the decoder in lockstep (same bits, same decisions), the
// compressed bytes around the current position are kept in LDS and refilled by all lanes (16 bytes each).
struct WaveBits {
  const uint8_t *p;
  uint64_t nbits;
  size_t nbytes;
  uint32_t *win;             // LDS: kWinBytes + 16 bytes
  uint32_t lane;
  mutable uint64_t win_lo;   // byte offset of win[0], a multiple of 16; ~0: nothing loaded
  // 128 bits of the stream in (scalar) registers: a peek that still finds its 56 bits there costs no LDS round trip
  mutable uint64_t reg_lo, reg_hi, reg_base;  // reg_base: bit position of reg_lo's bit 0, a multiple of 32; ~0: empty
  __device__ __forceinline__ const uint8_t *bytes(uint64_t byte_offset) const { return p + byte_offset; }
  __device__ __forceinline__ uint32_t uniform(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
  __device__ __forceinline__ uint64_t peek(uint64_t at) const {
    if (at < reg_base || at - reg_base > 128u - 56u) {
      const uint64_t b = (at >> 5) << 2;   // byte offset of the dword `at` lies in
      if (b < win_lo || b + 16 > win_lo + kWinBytes) {
        win_lo = b & ~15ull;   // the allocation is padded: reads past the end see zeros
        const uint4 *src = reinterpret_cast<const uint4 *>(p + win_lo);
        uint4 *dst = reinterpret_cast<uint4 *>(win);
        for (uint32_t i = lane; i < kWinBytes / 16; i += 64) dst[i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      const uint32_t i = (uint32_t)(b - win_lo) >> 2;
      const uint32_t w0 = uniform(win[i]), w1 = uniform(win[i + 1]), w2 = uniform(win[i + 2]), w3 = uniform(win[i + 3]);
      reg_lo = (uint64_t)w0 | ((uint64_t)w1 << 32);
      reg_hi = (uint64_t)w2 | ((uint64_t)w3 << 32);
      reg_base = b * 8;
    }
    const uint32_t s = (uint32_t)(at - reg_base);  // <= 72
    if (s == 0) return reg_lo;
    return s < 64 ? (reg_lo >> s) | (reg_hi << (64u - s)) : reg_hi >> (s - 64u);  // >= 56 valid bits
  }
  // 64 valid bits from bit position at + lane, a different one in every lane (`at` is the wave's): the lanes of a round
  // read two or three neighbouring dwords of the window between them (broadcast reads)
  __device__ __forceinline__ uint64_t lane_peek(uint64_t at) const {
    const uint64_t b0 = (at >> 5) << 2;
    if (b0 < win_lo || b0 + 32 > win_lo + kWinBytes) {   // the round's last lane reads dwords up to b0 + 8 + 12
      win_lo = b0 & ~15ull;
      const uint4 *src = reinterpret_cast<const uint4 *>(p + win_lo);
      uint4 *dst = reinterpret_cast<uint4 *>(win);
      for (uint32_t i = lane; i < kWinBytes / 16; i += 64) dst[i] = src[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const uint64_t mine = at + lane;
    const uint32_t i = (uint32_t)((((mine >> 5) << 2) - win_lo) >> 2), sh = (uint32_t)(mine & 31u);
    const uint64_t lo = (uint64_t)win[i] | ((uint64_t)win[i + 1] << 32);
    return sh ? (lo >> sh) | ((uint64_t)win[i + 2] << (64u - sh)) : lo;
  }
};

// Symbols go straight to the chunk's region in HBM: literals by lane 0, matches by as many lanes as they are long
// (one load and one store for the wave instead of a round trip per symbol).  A ring of recent symbols in LDS with
// block-wise flushes was tried: it costs registers (occupancy 2 waves per SIMD instead of 6) and did not pay
// (profiles/r02_device_inflate.txt).
struct WaveSymbolSink {
  uint16_t *sym;   // sym[i] = symbol of the chunk's byte i
  uint64_t cap, n;
  uint32_t lane;
  bool overflow;
  uint64_t mark_n;
  uint32_t n_rounds = 0, n_matches = 0, n_miss = 0;  // what the wave met (SPZ_AMD_LZ_TIMING prints the totals)
  bool skip_copies = false;                          // experiment: matches are counted, not copied (wrong output)
  __device__ __forceinline__ void mark() { mark_n = n; }
  __device__ __forceinline__ bool lit(uint8_t b) {
    if (n >= cap) {
      overflow = true;
      return false;
    }
    if (lane == 0) sym[n] = b;
    ++n;
    return true;
  }
  // the `len` symbols from `at` on, copied from `dist` back
  __device__ __forceinline__ void copy_at(uint64_t at, uint32_t len, uint32_t dist) {
    // byte at + i is byte at + i - dist, which for an overlapping copy is byte at - dist + (i mod dist): never one of this
    // match's own; a source before the chunk's first byte is byte W + (index) of the predecessor's final window
    // i mod dist for i < 258 without an integer division: (i + 0.5) / dist is at least 0.5 / 258 away from every
    // integer, far more than the error of the reciprocal, so the truncated product is the exact quotient
    const float rd = __builtin_amdgcn_rcpf((float)dist);
    for (uint32_t base = 0; base < len; base += 64) {
      const uint32_t i = base + lane;
      if (i < len) {
        const uint32_t q = dist < len ? (uint32_t)(((float)i + 0.5f) * rd) : 0u;
        const long long src = (long long)at - (long long)dist + (long long)(i - q * dist);
        sym[at + i] = src >= 0 ? sym[src] : (uint16_t)(256 + W + src);
      }
    }
  }
  __device__ __forceinline__ bool match(uint32_t len, uint32_t dist) {
    if (n + len > cap) {
      overflow = true;
      return false;
    }
    copy_at(n, len, dist);
    n += len;
    return true;
  }
  __device__ __forceinline__ bool raw(const uint8_t *src, uint32_t len) {  // a stored block: 64 bytes per round
    if (n + len > cap) {
      overflow = true;
      return false;
    }
    // a run of stored blocks can be tens of megabytes, all met by this one wave: 64 bytes per lane and round, four
    // 16-byte loads in flight
    uint32_t done = 0;
    for (; done + 4096 <= len; done += 4096) {
      const uint8_t *s = src + done + 64u * lane;
      uint4 q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) __builtin_memcpy(&q[k], s + 16 * k, 16);
      uint16_t *dst = sym + n + done + 64u * lane;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dst[16 * k + 4 * j + 0] = (uint16_t)(w[j] & 0xffu);
          dst[16 * k + 4 * j + 1] = (uint16_t)((w[j] >> 8) & 0xffu);
          dst[16 * k + 4 * j + 2] = (uint16_t)((w[j] >> 16) & 0xffu);
          dst[16 * k + 4 * j + 3] = (uint16_t)(w[j] >> 24);
        }
      }
    }
    for (uint32_t i = done + lane; i < len; i += 64) sym[n + i] = src[i];
    n += len;
    return true;
  }
};

// ---- the symbols of one Huffman-coded block, 64 bit positions at a time ---------------------------------------------
// decodeHuffBlock (spz_inflate_core.hpp) run by a wave in lockstep spends ~85 instructions per symbol with 63 lanes
// idle.  Here a round covers the 64 bit positions from `pos`: lane l decodes the symbol that WOULD start at pos + l
// (fast-table lookups, extra bits, distance code: all per lane), a scalar walk then follows the real chain — position
// 0 is a symbol start, and each start names the next through its own length — (one readlane per symbol), the output
// offsets of the real symbols are a scan of their lengths, their literals are stored by their own lanes at once and
// their matches copied in order by the whole wave.  A code the fast tables do not hold is decoded the old way, by all
// lanes, when (and only when) the chain reaches it.  Same symbols as decodeHuffBlock by construction: the per-lane
// arithmetic is that function's, applied at every position instead of one.
constexpr uint32_t F_LIT = 1u, F_MATCH = 2u, F_EOB = 4u, F_INVALID = 8u, F_MISS = 16u;

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {  // a value all lanes agree on, moved to scalar registers
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

template <class HL, class HD>
__device__ __forceinline__ bool decodeHuffBlockWave(const WaveBits &in, uint64_t *at, const HL &L, const HD &D, WaveSymbolSink &sink,
                                                    uint64_t limit) {
  uint64_t pos = uniform64(*at);
  const uint32_t lane = in.lane;
  for (;;) {
    if (pos >= in.nbits || pos > limit) return false;
    uint64_t bits = in.lane_peek(pos);
    const uint32_t e = L.packed[bits & ((1u << HL::fastbits) - 1u)];
    uint32_t total = e & 15u, flags, outlen = 0, dist = 0;
    const uint32_t value = e >> 16;  // the literal
    if (e == 0u) {
      flags = F_MISS;
    } else if (e & ENT_LITERAL) {
      flags = F_LIT;
      outlen = 1;
    } else if (e & ENT_EOB) {
      flags = F_EOB;
    } else if (e & ENT_INVALID) {
      flags = F_INVALID;
    } else {
      bits >>= total;
      const uint32_t lextra = (e >> 4) & 15u;
      outlen = (e >> 16) + (uint32_t)(bits & ((1u << lextra) - 1u));
      bits >>= lextra;
      const uint32_t d = D.packed[bits & ((1u << HD::fastbits) - 1u)];
      if (d == 0u) {
        flags = F_MISS;
      } else if (d & ENT_INVALID) {
        flags = F_INVALID;
      } else {
        bits >>= d & 15u;
        const uint32_t dextra = (d >> 4) & 15u;
        dist = (d >> 16) + (uint32_t)(bits & ((1u << dextra) - 1u));
        total += lextra + (d & 15u) + dextra;  // <= 48
        flags = F_MATCH;
      }
    }
    // The chain of real symbol starts: position 0 is one, and each names the next through its own length.  A symbol
    // that ends the round's ordinary processing (end of block, a code the fast tables do not hold, an invalid code)
    // steps out of the round, and is looked at afterwards.
    const bool stops = (flags & (F_EOB | F_MISS | F_INVALID)) != 0u;
    const uint32_t step = stops ? 64u : total;
    uint32_t cur = 0;
    uint64_t starts = 0;
    do {
      starts |= 1ull << cur;
      cur += (uint32_t)__builtin_amdgcn_readlane((int)step, (int)cur);
    } while (cur < 64u);
    const uint64_t stopped = __ballot(stops) & starts;  // at most the chain's last start
    const bool mine = ((starts >> lane) & 1ull) != 0ull && !stops;
    const uint32_t ol = mine ? outlen : 0u;
    uint32_t incl = ol;
#pragma unroll
    for (uint32_t off = 1; off < 64u; off <<= 1) {
      const uint32_t y = __shfl_up(incl, off, 64);
      if (lane >= off) incl += y;
    }
    const uint32_t produced = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint64_t n = uniform64(sink.n);
    if (n + produced > sink.cap) {
      sink.overflow = true;
      return false;
    }
    const uint32_t before = incl - ol;
    if (mine && (flags & F_LIT)) sink.sym[n + before] = (uint16_t)value;
    // (Batching the round's matches — all their loads, then all their stores, when none reads what another writes — was
    // measured: the bookkeeping costs more than the waits it saves, 48 ms for the stage instead of 37.  Round 3: only
    // PAIRS of short matches, the second reading nothing the first writes, both reads issued before either store — no
    // bookkeeping to speak of — measured the same as one after the other, 36.5 ms.)
    uint64_t matches = __ballot(mine && (flags & F_MATCH));
    while (matches) {
      const int l = __builtin_ctzll(matches);
      matches &= matches - 1ull;
      const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)outlen, l);
      const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)dist, l);
      const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)before, l);
      ++sink.n_matches;
      if (!sink.skip_copies) sink.copy_at(n + off, len, dst);
    }
    ++sink.n_rounds;
    n += produced;
    sink.n = n;
    if (stopped == 0ull) {
      pos += cur;
      continue;
    }
    // the symbol the chain stopped at, decoded the serial way (all lanes, same bits): the end of the block almost always
    const uint32_t sl = (uint32_t)__builtin_ctzll(stopped);
    pos += sl;
    uint64_t b = in.peek(pos);
    const uint32_t e2 = in.uniform(L.lookup(b));
    if (e2 == 0u || (e2 & ENT_INVALID)) return false;
    uint32_t used = e2 & 15u;
    if (e2 & ENT_EOB) {
      pos += used;
      if (pos > in.nbits) return false;
      *at = pos;
      return true;
    }
    ++sink.n_miss;
    if (e2 & ENT_LITERAL) {
      if (!sink.lit((uint8_t)(e2 >> 16))) return false;
      pos += used;
      continue;
    }
    b >>= used;
    const uint32_t lextra = (e2 >> 4) & 15u;
    const uint32_t length = (e2 >> 16) + (uint32_t)(b & ((1u << lextra) - 1u));
    b >>= lextra;
    used += lextra;
    const uint32_t dd = in.uniform(D.lookup(b));
    if (dd == 0u || (dd & ENT_INVALID)) return false;
    b >>= dd & 15u;
    const uint32_t dextra = (dd >> 4) & 15u;
    const uint32_t distance = (dd >> 16) + (uint32_t)(b & ((1u << dextra) - 1u));
    used += (dd & 15u) + dextra;
    pos += used;
    if (pos > in.nbits) return false;
    ++sink.n_matches;
    if (sink.skip_copies) {
      sink.n += length;
    } else if (!sink.match(length, distance)) {
      return false;
    }
  }
}

__global__ __launch_bounds__(64, 6) void inf_decode_kernel(const uint8_t *__restrict__ d, uint64_t nbytes,
                                                           const ChunkJob *__restrict__ jobs, const uint32_t *__restrict__ run,
                                                           uint16_t *symbols, ChunkResult *__restrict__ results, uint32_t experiment) {
  __shared__ Tables tb;
  // The header arrays stay in scratch memory HERE (3-4 headers per wave): in LDS the kernel needs 123 registers instead of
  // 80 — four waves per SIMD instead of six — and the stage took 47 ms instead of 35; the block-start search, whose time
  // WAS these arrays, keeps them in LDS (14 -> 8 ms).
  HeaderWork s_work;
  __shared__ uint32_t s_win[kWinBytes / 4 + 4];
  const uint32_t j = run[blockIdx.x];
  const ChunkJob job = jobs[j];
  const WaveBits in = {d, 8ull * nbytes, (size_t)nbytes, s_win, threadIdx.x, ~0ull, 0ull, 0ull, ~0ull};
  WaveSymbolSink sink = {symbols + job.region, job.capacity, job.start_n, threadIdx.x, false, job.start_n};
  sink.skip_copies = (experiment & 1u) != 0u;
  uint64_t end = 0;
  const Outcome r = decodeBlocksWith(in, job.from, job.to, sink, &end, &tb.lit, &tb.dist, &s_work,
                                     [](const WaveBits &i, uint64_t *at, const DevLit &L, const DevDist &D, WaveSymbolSink &s, uint64_t limit) {
                                       return decodeHuffBlockWave(i, at, L, D, s, limit);
                                     });
  if (threadIdx.x == 0) {
    ChunkResult &o = results[j];
    o.length = sink.n;
    o.end_bit = end;
    o.outcome = (uint32_t)r;
    o.overflow = sink.overflow ? 1u : 0u;
    o.mark_n = sink.mark_n;
    o.n_rounds = sink.n_rounds;
    o.n_matches = sink.n_matches;
    o.n_miss = sink.n_miss;
  }
}

// ---- windows ----------------------------------------------------------------------------------------------
struct ChunkPlace {
  unsigned long long region, length, offset;  // symbols at `region`, bytes at `offset`
};

// The final window of a chunk, symbolically: entry k is a byte, or 256 + i = "entry i of the predecessor's final window"
// (a reference its own symbols carry, or — a chunk shorter than the window — the predecessor's bytes moving up).
__global__ __launch_bounds__(256) void inf_window_gather_kernel(const uint16_t *__restrict__ symbols, const ChunkPlace *__restrict__ chunks,
                                                                uint16_t *__restrict__ windows) {
  const uint32_t c = blockIdx.x;
  const ChunkPlace ch = chunks[c];
  const uint16_t *sym = symbols + ch.region;
  for (uint32_t k = threadIdx.x; k < W; k += 256) {
    const long long idx = (long long)ch.length - (long long)W + k;
    windows[(size_t)c * W + k] = idx >= 0 ? sym[idx] : (uint16_t)(256u + (uint32_t)((long long)ch.length + k));
  }
}

// Resolution of the references: R_c[k] = S_c[k] if that is a byte, else R_(c-1)[S_c[k] - 256] — a chain through the
// predecessors that ends where the byte was produced.  The maps compose associatively and almost every chain is one
// or two chunks long, so instead of walking the chunks in order (one workgroup, a step per chunk: 21 ms for the 5.5 k
// chunks of a 650 MB stream) every entry follows its own chain, up to kChainSteps chunks per pass, and writes the byte
// back in place; an entry read while another thread resolves it is either still the reference or already the byte, both
// right.  After pass p every chain of up to p * kChainSteps chunks is resolved (the entries it runs through were
// resolved in the passes before), so input whose bytes are carried through hundreds of chunks (long runs) takes a few
// passes more; `pending` counts what is left.
constexpr uint32_t kChainSteps = 32;
__global__ __launch_bounds__(256) void inf_window_resolve_kernel(uint16_t *__restrict__ windows, uint32_t n_chunks,
                                                                 uint32_t *__restrict__ pending, uint32_t *__restrict__ bad) {
  const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t c = (uint32_t)(e / W);
  if (c >= n_chunks) return;
  uint32_t s = windows[e];
  if (s < 256u) return;
  uint32_t steps = 0;
  while (s >= 256u && c > 0u && steps < kChainSteps) {
    --c;
    s = windows[(size_t)c * W + (s - 256u)];
    ++steps;
  }
  if (s < 256u) windows[e] = (uint16_t)s;
  else if (c == 0u) atomicOr(bad, 1u);  // the first chunk has nothing before it
  else atomicAdd(pending, 1u);
}

// ---- place ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void inf_place_kernel(const uint16_t *__restrict__ symbols, const ChunkPlace *__restrict__ chunks,
                                                        const uint16_t *__restrict__ windows, uint8_t *__restrict__ out,
                                                        uint32_t *__restrict__ bad) {
  const uint32_t c = blockIdx.y;
  const ChunkPlace ch = chunks[c];
  const uint16_t *sym = symbols + ch.region;
  const uint16_t *prev = c ? windows + (size_t)(c - 1) * W : nullptr;  // resolved: bytes
  uint8_t *dst = out + ch.offset;
  uint32_t broken = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < ch.length; i += (unsigned long long)gridDim.x * 256) {
    const uint32_t s = sym[i];
    uint32_t b = s;
    if (s >= 256) {
      if (prev) b = prev[s - 256];
      else broken = 1;
      if (b >= 256) broken = 1;  // a window entry the resolution left open
    }
    dst[i] = (uint8_t)b;
  }
  if (broken) atomicOr(bad, 1u);
}

// ---- CRC-32 (IEEE 802.3, the one gzip uses) of pieces ------------------------------------------------------------
// One wave per 256 KiB piece: lane t takes the t-th 64th of it (slice-by-4 tables in LDS), and the lanes' CRCs are
// put together the way crc32_combine does it — crc(A || B) = crc(A) * x^(8 len B) mod P  xor  crc(B), which over all
// 64 parts is the xor of every part's CRC times x^(8 * bytes after it).  (A thread per piece, 39 waves for a 650 MB
// stream, took 9.3 ms.)
constexpr uint32_t kCrcPoly = 0xedb88320u;

__device__ __forceinline__ uint32_t crc_multmodp(uint32_t a, uint32_t b) {  // a * b mod P, bit-reflected (x^0 = 1 << 31)
  uint32_t m = 1u << 31, p = 0u;
  for (;;) {
    if (a & m) {
      p ^= b;
      if ((a & (m - 1u)) == 0u) break;
    }
    m >>= 1;
    b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
  }
  return p;
}

__global__ __launch_bounds__(64) void inf_crc_kernel(const uint8_t *__restrict__ data, uint64_t nbytes, uint32_t piece,
                                                     uint32_t *__restrict__ crcs) {
  __shared__ uint32_t table[4][256];
  __shared__ uint32_t x2n[32];  // x^(2^n) mod P
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < 256; i += 64) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? kCrcPoly ^ (c >> 1) : c >> 1;
    table[0][i] = c;
  }
  if (lane == 0) {
    uint32_t p = 1u << 30;  // x^1
    x2n[0] = p;
    for (int n = 1; n < 32; ++n) x2n[n] = p = crc_multmodp(p, p);
  }
  __syncthreads();
  for (int t = 1; t < 4; ++t) {
    for (uint32_t i = lane; i < 256; i += 64) {
      const uint32_t c = table[t - 1][i];
      table[t][i] = (c >> 8) ^ table[0][c & 0xffu];
    }
    __syncthreads();
  }
  const uint64_t piece_lo = (uint64_t)blockIdx.x * piece;
  if (piece_lo >= nbytes) return;
  const uint64_t piece_hi = piece_lo + piece < nbytes ? piece_lo + piece : nbytes;
  const uint32_t sub = piece / 64u;  // a multiple of 64 bytes (pieces start 256-byte aligned)
  const uint64_t lo = piece_lo + (uint64_t)lane * sub < piece_hi ? piece_lo + (uint64_t)lane * sub : piece_hi;
  const uint64_t hi = lo + sub < piece_hi ? lo + sub : piece_hi;
  uint32_t c = 0xffffffffu;
  uint64_t p = lo;
  auto eat = [&](uint32_t w) {
    c ^= w;
    c = table[3][c & 0xffu] ^ table[2][(c >> 8) & 0xffu] ^ table[1][(c >> 16) & 0xffu] ^ table[0][c >> 24];
  };
  for (; p + 64 <= hi; p += 64) {
    const uint4 *q = reinterpret_cast<const uint4 *>(data + p);
    const uint4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
    eat(a0.x); eat(a0.y); eat(a0.z); eat(a0.w);
    eat(a1.x); eat(a1.y); eat(a1.z); eat(a1.w);
    eat(a2.x); eat(a2.y); eat(a2.z); eat(a2.w);
    eat(a3.x); eat(a3.y); eat(a3.z); eat(a3.w);
  }
  for (; p < hi; ++p) c = table[0][(c ^ data[p]) & 0xffu] ^ (c >> 8);
  c ^= 0xffffffffu;
  // this part's CRC times x^(8 * bytes after it); an empty part contributes nothing
  uint32_t part = 0u;
  if (hi > lo) {
    uint64_t n = piece_hi - hi;  // bytes after this part
    uint32_t op = 1u << 31;      // x^0
    for (uint32_t k = 3; n != 0; n >>= 1, ++k) {
      if (n & 1u) op = crc_multmodp(x2n[k & 31u], op);
    }
    part = crc_multmodp(op, c);
  }
  for (int off = 32; off > 0; off >>= 1) part ^= __shfl_xor(part, off, 64);
  if (lane == 0) crcs[blockIdx.x] = part;
}

// Byte equality of two device buffers (the writer's self-check: the member it has just made, inflated, against its input).
__global__ __launch_bounds__(256) void inf_compare_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint64_t nbytes,
                                                         uint32_t *__restrict__ diff) {
  const uint64_t groups = nbytes / 16;
  uint32_t bad = 0;
  for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (uint64_t)gridDim.x * 256) {
    uint4 x, y;
    __builtin_memcpy(&x, a + 16 * g, 16);
    __builtin_memcpy(&y, b + 16 * g, 16);
    if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) bad = 1;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (uint64_t i = groups * 16; i < nbytes; ++i) bad |= a[i] != b[i];
  }
  if (bad) atomicAdd(diff, 1u);
}

struct InfContext {
  int device = 0;
  char *block = nullptr;
  uint8_t *out = nullptr;
  uint64_t out_bytes = 0;
  uint32_t *crcs = nullptr;
  uint32_t n_pieces = 0;
  uint32_t *diff = nullptr;  // equals_device's count of differing groups
};

size_t round256(size_t v) { return (v + 255) / 256 * 256; }

// Why the device reader last stood down on this thread ("" = it did not): the caller's host readers produce the same
// bytes, so a decline shows only as time — spz_amd_inflate_last_decline() says what to look at.
thread_local const char *g_last_decline = "";
int decline(const char *reason) {
  g_last_decline = reason;
  if (std::getenv("SPZ_AMD_LZ_TIMING")) std::fprintf(stderr, "[inflate] declined: %s\n", reason);
  return SPZ_AMD_ERR_UNSUPPORTED;
}

}  // namespace
}  // namespace spz_amd_detail

using namespace spz_amd_detail;

extern "C" {

// The deflate data comes from the host (h_deflate) or is on the device already (d_deflate: the writer's own body,
// spz_amd_zlib_verify_member).
static int inflate_open_impl(const uint8_t *h_deflate, const uint8_t *d_deflate, uint64_t nbytes, int device, void **ctx,
                             uint64_t *out_bytes, void (*after_upload)(void *) = nullptr, void *after_arg = nullptr) {
  if ((h_deflate == nullptr) == (d_deflate == nullptr) || ctx == nullptr || out_bytes == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  *ctx = nullptr;
  g_last_decline = "";
  if (nbytes < 4ull * kMinChunkBytes || nbytes >= (1ull << 32)) return decline("size");  // under 128 KiB of deflate data, or 4 GiB and more
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  // (A member that opens with a stored block was declined until round 3 — "nothing to gain" — which sent every file
  // whose first bytes are incompressible to the host readers: 4 % of the coverage campaign.  Runs of stored blocks are
  // found and copied in parallel like everything else, so it is decoded here now.)
  static const bool timing = std::getenv("SPZ_AMD_LZ_TIMING") != nullptr;
  // SPZ_AMD_INFLATE_EXPERIMENT (measurements only; bit 0: matches are counted but not copied — the result is wrong and
  // the CRC check sends the caller to the host readers)
  const char *exp_env = std::getenv("SPZ_AMD_INFLATE_EXPERIMENT");
  const uint32_t experiment = exp_env ? (uint32_t)std::atoi(exp_env) : 0u;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[inflate] %-8s %.4f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  // One wave per chunk, and a decode stage that is as long as its slowest wave: with a few more chunks than the chip holds
  // waves, the stragglers' second round doubles the stage (6236 chunks on 5888 slots: 70 ms; 5700 chunks: one round).
  int cus = 0;
  SPZ_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  const uint64_t slots = (uint64_t)(cus > 0 ? cus : 256) * kDecodeWavesPerCU;
  uint32_t chunk_bytes = kMinChunkBytes;
  if (nbytes / kMinChunkBytes > slots * 95 / 100) chunk_bytes = (uint32_t)((nbytes / (slots * 95 / 100) + 4095) / 4096 * 4096);
  const uint32_t n_chunks = (uint32_t)(nbytes / chunk_bytes);  // the last one takes the remainder
  // one allocation: the deflate data, block starts, jobs/results/places, symbols, windows; the output and CRCs later
  const size_t sym_capacity = (size_t)nbytes * kExpand + (size_t)n_chunks * 128;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t at = off;
    off += round256(bytes);
    return at;
  };
  const size_t o_data = carve(nbytes + kWinBytes + 64);
  const size_t o_starts = carve((size_t)n_chunks * sizeof(unsigned long long));
  const size_t o_jobs = carve((size_t)n_chunks * sizeof(ChunkJob));
  const size_t o_res = carve((size_t)n_chunks * sizeof(ChunkResult));
  const size_t o_place = carve((size_t)n_chunks * sizeof(ChunkPlace));
  const size_t o_bad = carve(256);
  const size_t o_run = carve((size_t)n_chunks * sizeof(uint32_t));
  const size_t o_win = carve((size_t)n_chunks * W * sizeof(uint16_t));
  const size_t o_sym = carve(sym_capacity * sizeof(uint16_t));
  const size_t total = off;
  size_t free_b = 0;
  rc = device_free_bytes(device, &free_b);
  if (rc != SPZ_AMD_OK) return rc;
  if (total + (size_t)nbytes * kExpand + (size_t(512) << 20) > free_b) return decline("memory");
  char *block = nullptr;
  rc = scratch_acquire(device, total, reinterpret_cast<void **>(&block));
  if (rc != SPZ_AMD_OK) return rc;
  struct Free {
    char *p;
    int device;
    ~Free() {
      if (p) scratch_release(device, p);
    }
  } holder{block, device};
  uint8_t *d_data = reinterpret_cast<uint8_t *>(block + o_data);
  unsigned long long *d_starts = reinterpret_cast<unsigned long long *>(block + o_starts);
  ChunkJob *d_jobs = reinterpret_cast<ChunkJob *>(block + o_jobs);
  ChunkResult *d_res = reinterpret_cast<ChunkResult *>(block + o_res);
  ChunkPlace *d_place = reinterpret_cast<ChunkPlace *>(block + o_place);
  uint32_t *d_bad = reinterpret_cast<uint32_t *>(block + o_bad);
  uint32_t *d_run = reinterpret_cast<uint32_t *>(block + o_run);
  uint16_t *d_win = reinterpret_cast<uint16_t *>(block + o_win);
  uint16_t *d_sym = reinterpret_cast<uint16_t *>(block + o_sym);
  lap("alloc");
  hipStream_t st = nullptr;
  if (h_deflate != nullptr) SPZ_HIP_TRY(upload_adaptive(d_data, h_deflate, nbytes, st));
  else SPZ_HIP_TRY(hipMemcpyAsync(d_data, d_deflate, nbytes, hipMemcpyDeviceToDevice, st));
  SPZ_HIP_TRY(hipMemsetAsync(d_data + nbytes, 0, kWinBytes + 64, st));
  SPZ_HIP_TRY(hipMemsetAsync(d_bad, 0, 256, st));
  lap("upload");
  // the caller's host-side work that must not run beside the upload (mapping output pages: the two contend in the
  // kernel's memory management and the upload took 80 ms longer) but may run beside the kernels
  if (after_upload) after_upload(after_arg);
  // ---- 1. block starts.  (Searching the chunks of an uploaded prefix while the next part of the member uploads — four
  // parts, a stream of its own, low priority — was measured in round 3: upload + search 25 ms instead of 7 + 10; the
  // upload of pageable memory and a kernel that fills the chip get in each other's way on this system.)
  unsigned long long *d_search_stats = timing ? reinterpret_cast<unsigned long long *>(d_bad + 16) : nullptr;  // (zeroed above)
  hipLaunchKernelGGL(inf_search_kernel, dim3(n_chunks - 1), dim3(256), 0, st, d_data, nbytes, chunk_bytes, n_chunks, d_starts, d_search_stats);
  SPZ_HIP_TRY(hipGetLastError());
  std::vector<unsigned long long> starts(n_chunks);
  SPZ_HIP_TRY(hipMemcpyAsync(starts.data(), d_starts, (size_t)n_chunks * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  SPZ_HIP_TRY(hipStreamSynchronize(st));
  starts[0] = 0;
  lap("search");
  if (timing) {
    unsigned long long h[7] = {0, 0, 0, 0, 0, 0, 0};
    SPZ_HIP_TRY(hipMemcpy(h, d_search_stats, sizeof(h), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "[inflate] search: clock ticks per chunk: stored-block scan %.0f, header scan %.0f, full checks %.0f\n", (double)h[2] / (n_chunks - 1),
                 (double)h[3] / (n_chunks - 1), (double)h[4] / (n_chunks - 1));
    std::fprintf(stderr, "[inflate] search: %llu code-length symbols decoded by the full checks, %llu rounds of checks that found nothing\n", h[5], h[6]);
    std::fprintf(stderr, "[inflate] search: %llu batches of 65536 bit positions for %u chunks, %llu candidates for the full check (%.1f per batch)\n", h[0],
                 n_chunks - 1, h[1], h[0] ? (double)h[1] / (double)h[0] : 0.0);
  }
  std::vector<ChunkJob> jobs;
  jobs.reserve(n_chunks);
  for (uint32_t i = 0; i < n_chunks; ++i) {
    if (starts[i] == NONE) continue;
    if (!jobs.empty() && starts[i] <= jobs.back().from) continue;  // found again from the chunk before: one job
    jobs.push_back({starts[i], NONE, 0, 0, 0});
  }
  uint32_t n = (uint32_t)jobs.size();
  if (n < 2) return decline("no-block-starts");
  if (timing) std::fprintf(stderr, "[inflate] %u of %u chunks have a block start\n", n, n_chunks);
  unsigned long long region = 0;
  for (uint32_t j = 0; j < n; ++j) {
    jobs[j].to = (j + 1 < n) ? jobs[j + 1].from : NONE;
    const unsigned long long span_bits = (j + 1 < n ? jobs[j + 1].from : 8ull * nbytes) - jobs[j].from;
    jobs[j].region = region;
    jobs[j].capacity = ((span_bits / 8 + 1) * kExpand + 32 + 63) / 64 * 64;  // regions stay 128-byte aligned
    region += jobs[j].capacity;
  }
  if (region > sym_capacity) return decline("symbol-budget");
  // ---- 2. decode.  A chunk that does not end exactly at its successor's block start has met a look-alike (raw bytes
  // of a stored block that read like a header): the successor is dropped, the chunk takes its range and region and
  // is decoded again.  Chunk 0 starts at a true block start, so this is sound by induction.
  std::vector<uint32_t> live(n), todo(n);
  for (uint32_t j = 0; j < n; ++j) live[j] = todo[j] = j;
  std::vector<ChunkResult> res(n);
  bool linked = false;
  for (uint32_t round = 0; round < 8 && !linked; ++round) {
    SPZ_HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), (size_t)n * sizeof(ChunkJob), hipMemcpyHostToDevice, st));
    SPZ_HIP_TRY(hipMemcpyAsync(d_run, todo.data(), todo.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(inf_decode_kernel, dim3((unsigned)todo.size()), dim3(64), 0, st, d_data, nbytes, d_jobs, d_run, d_sym, d_res,
                       experiment);
    SPZ_HIP_TRY(hipGetLastError());
    SPZ_HIP_TRY(hipMemcpyAsync(res.data(), d_res, (size_t)n * sizeof(ChunkResult), hipMemcpyDeviceToHost, st));
    SPZ_HIP_TRY(hipStreamSynchronize(st));
    todo.clear();
    std::vector<uint32_t> next_live;
    next_live.reserve(live.size());
    for (size_t k = 0; k < live.size(); ++k) {
      const uint32_t j = live[k];
      const bool last = k + 1 == live.size();
      if (res[j].overflow) return decline("expansion");  // a chunk that expands more than 8 x
      if (res[j].outcome == (uint32_t)(last ? FINAL : LINKED)) {
        next_live.push_back(j);
        continue;
      }
      if (last) return decline("no-final-block");  // the final block was not reached: not this reader's case (or damage)
      const uint32_t drop = live[k + 1];           // its block start was not one
      jobs[j].to = jobs[drop].to;
      jobs[j].capacity += jobs[drop].capacity;     // regions are adjacent
      jobs[j].from = res[j].end_bit;               // the blocks before the one that ran past the look-alike stand
      jobs[j].start_n = res[j].mark_n;
      next_live.push_back(j);
      todo.push_back(j);
      ++k;                                         // the dropped chunk's own result does not count
    }
    live.swap(next_live);
    linked = todo.empty();
    if (timing && !linked) std::fprintf(stderr, "[inflate] round %u: %zu chunks decoded again\n", round, todo.size());
  }
  if (!linked) return decline("not-linked");  // look-alike block starts left after 8 rounds
  lap("decode");
  if (timing) {
    unsigned long long rounds = 0, matches = 0, miss = 0, syms = 0;
    for (uint32_t j : live) {
      rounds += res[j].n_rounds;
      matches += res[j].n_matches;
      miss += res[j].n_miss;
      syms += res[j].length;
    }
    std::fprintf(stderr, "[inflate] %llu bytes out, %llu rounds of 64 bit positions, %llu matches, %llu long codes\n", syms, rounds, matches, miss);
  }
  n = (uint32_t)live.size();
  std::vector<ChunkPlace> place(n);
  unsigned long long offset = 0;
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t j = live[k];
    place[k] = {jobs[j].region, res[j].length, offset};
    offset += res[j].length;
  }
  if (((res[live[n - 1]].end_bit + 7) >> 3) != nbytes) return decline("trailing-bytes");  // the stream must end exactly at the trailer
  const uint64_t total_out = offset;
  if (total_out == 0) return decline("empty");
  SPZ_HIP_TRY(hipMemcpyAsync(d_place, place.data(), (size_t)n * sizeof(ChunkPlace), hipMemcpyHostToDevice, st));
  // ---- 3. windows, 4. place, 5. piece CRCs
  const uint32_t n_pieces = (uint32_t)((total_out + kCrcPiece - 1) / kCrcPiece);
  char *outblock = nullptr;
  const size_t out_alloc = round256(total_out + 64) + round256((size_t)n_pieces * sizeof(uint32_t)) + 256;  // + equals_device's counter
  rc = scratch_acquire(device, out_alloc, reinterpret_cast<void **>(&outblock));
  if (rc != SPZ_AMD_OK) return rc;
  Free holder2{outblock, device};
  uint8_t *d_out = reinterpret_cast<uint8_t *>(outblock);
  uint32_t *d_crcs = reinterpret_cast<uint32_t *>(outblock + round256(total_out + 64));
  hipLaunchKernelGGL(inf_window_gather_kernel, dim3(n), dim3(256), 0, st, d_sym, d_place, d_win);
  SPZ_HIP_TRY(hipGetLastError());
  {
    uint32_t *d_pending = d_bad + 1;
    uint32_t pending = 1;
    const uint32_t resolve_blocks = (uint32_t)(((uint64_t)n * W + 255) / 256);
    for (uint32_t pass = 0; pass < 64 && pending != 0; ++pass) {
      SPZ_HIP_TRY(hipMemsetAsync(d_pending, 0, sizeof(uint32_t), st));
      hipLaunchKernelGGL(inf_window_resolve_kernel, dim3(resolve_blocks), dim3(256), 0, st, d_win, n, d_pending, d_bad);
      SPZ_HIP_TRY(hipGetLastError());
      SPZ_HIP_TRY(hipMemcpyAsync(&pending, d_pending, sizeof(pending), hipMemcpyDeviceToHost, st));
      SPZ_HIP_TRY(hipStreamSynchronize(st));
      if (timing && pending) std::fprintf(stderr, "[inflate] window pass %u: %u entries still open\n", pass, pending);
    }
    if (pending != 0) return decline("window-chains");  // bytes carried through more than 2048 chunks: the host readers
  }
  lap("windows");
  hipLaunchKernelGGL(inf_place_kernel, dim3(16, n), dim3(256), 0, st, d_sym, d_place, d_win, d_out, d_bad);
  SPZ_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(inf_crc_kernel, dim3(n_pieces), dim3(64), 0, st, d_out, total_out, kCrcPiece, d_crcs);
  SPZ_HIP_TRY(hipGetLastError());
  uint32_t bad = 0;
  SPZ_HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, st));
  SPZ_HIP_TRY(hipStreamSynchronize(st));
  lap("place+crc");
  if (bad) return decline("bad-reference");  // a reference where there is no predecessor
  InfContext *c = new (std::nothrow) InfContext();
  if (c == nullptr) return SPZ_AMD_ERR_HIP;
  c->device = device;
  c->block = outblock;
  holder2.p = nullptr;  // the scratch block goes now, the output stays with the context
  c->out = d_out;
  c->out_bytes = total_out;
  c->crcs = d_crcs;
  c->n_pieces = n_pieces;
  c->diff = reinterpret_cast<uint32_t *>(outblock + round256(total_out + 64) + round256((size_t)n_pieces * sizeof(uint32_t)));
  *ctx = c;
  *out_bytes = total_out;
  return SPZ_AMD_OK;
}

int spz_amd_inflate_open(const uint8_t *h_deflate, uint64_t nbytes, int device, void **ctx, uint64_t *out_bytes) {
  if (h_deflate == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  return inflate_open_impl(h_deflate, nullptr, nbytes, device, ctx, out_bytes);
}

int spz_amd_inflate_open_ex(const uint8_t *h_deflate, uint64_t nbytes, int device, void **ctx, uint64_t *out_bytes,
                            void (*after_upload)(void *), void *after_arg) {
  if (h_deflate == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  return inflate_open_impl(h_deflate, nullptr, nbytes, device, ctx, out_bytes, after_upload, after_arg);
}

int spz_amd_inflate_open_device(const uint8_t *d_deflate, uint64_t nbytes, int device, void **ctx, uint64_t *out_bytes) {
  if (d_deflate == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  return inflate_open_impl(nullptr, d_deflate, nbytes, device, ctx, out_bytes);
}

int spz_amd_stream_to_device(const uint8_t *h_stream, uint64_t nbytes, int device, void **ctx) {
  if (h_stream == nullptr || ctx == nullptr || nbytes == 0) return SPZ_AMD_ERR_INVALID_ARG;
  *ctx = nullptr;
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  const uint32_t n_pieces = (uint32_t)((nbytes + kCrcPiece - 1) / kCrcPiece);
  const size_t bytes = round256(nbytes + 64) + round256((size_t)n_pieces * sizeof(uint32_t)) + 256;
  char *block = nullptr;
  rc = scratch_acquire(device, bytes, reinterpret_cast<void **>(&block));
  if (rc != SPZ_AMD_OK) return rc;
  hipStream_t st = nullptr;
  hipError_t e = upload_adaptive(block, h_stream, nbytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(block + nbytes, 0, 64, st);
  uint32_t *d_crcs = reinterpret_cast<uint32_t *>(block + round256(nbytes + 64));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(inf_crc_kernel, dim3(n_pieces), dim3(64), 0, st, reinterpret_cast<const uint8_t *>(block), nbytes, kCrcPiece, d_crcs);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  InfContext *c = e == hipSuccess ? new (std::nothrow) InfContext() : nullptr;
  if (c == nullptr) {
    scratch_release(device, block);
    if (e != hipSuccess) g_last_hip_error = (int)e;
    return SPZ_AMD_ERR_HIP;
  }
  c->device = device;
  c->block = block;
  c->out = reinterpret_cast<uint8_t *>(block);
  c->out_bytes = nbytes;
  c->crcs = d_crcs;
  c->n_pieces = n_pieces;
  c->diff = reinterpret_cast<uint32_t *>(block + round256(nbytes + 64) + round256((size_t)n_pieces * sizeof(uint32_t)));
  *ctx = c;
  return SPZ_AMD_OK;
}

int spz_amd_inflate_equals_device(void *ctx, const uint8_t *d_expected, uint64_t nbytes) {
  InfContext *c = static_cast<InfContext *>(ctx);
  if (c == nullptr || d_expected == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  if (nbytes != c->out_bytes) return SPZ_AMD_ERR_VERIFY;
  DeviceGuard guard;
  int rc = guard.enter(c->device);
  if (rc != SPZ_AMD_OK) return rc;
  uint32_t *d_diff = c->diff;
  hipStream_t st = nullptr;
  SPZ_HIP_TRY(hipMemsetAsync(d_diff, 0, sizeof(uint32_t), st));
  hipLaunchKernelGGL(inf_compare_kernel, dim3(2048), dim3(256), 0, st, c->out, d_expected, nbytes, d_diff);
  SPZ_HIP_TRY(hipGetLastError());
  uint32_t diff = 1;
  SPZ_HIP_TRY(hipMemcpyAsync(&diff, d_diff, sizeof(diff), hipMemcpyDeviceToHost, st));
  SPZ_HIP_TRY(hipStreamSynchronize(st));
  return diff == 0 ? SPZ_AMD_OK : SPZ_AMD_ERR_VERIFY;
}

const char *spz_amd_inflate_last_decline(void) { return g_last_decline; }

uint32_t spz_amd_inflate_crc_piece_bytes(void) { return kCrcPiece; }

int spz_amd_inflate_piece_crcs(void *ctx, uint32_t *h_crcs, uint32_t capacity, uint32_t *num_pieces) {
  InfContext *c = static_cast<InfContext *>(ctx);
  if (c == nullptr || h_crcs == nullptr || num_pieces == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  *num_pieces = c->n_pieces;
  if (capacity < c->n_pieces) return SPZ_AMD_ERR_CAPACITY;
  DeviceGuard guard;
  int rc = guard.enter(c->device);
  if (rc != SPZ_AMD_OK) return rc;
  SPZ_HIP_TRY(hipMemcpy(h_crcs, c->crcs, (size_t)c->n_pieces * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return SPZ_AMD_OK;
}

int spz_amd_inflate_fetch(void *ctx, uint8_t *h_out) {
  InfContext *c = static_cast<InfContext *>(ctx);
  if (c == nullptr || h_out == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  DeviceGuard guard;
  int rc = guard.enter(c->device);
  if (rc != SPZ_AMD_OK) return rc;
  SPZ_HIP_TRY(hipMemcpy(h_out, c->out, c->out_bytes, hipMemcpyDeviceToHost));
  return SPZ_AMD_OK;
}

const uint8_t *spz_amd_inflate_device_data(void *ctx) {
  InfContext *c = static_cast<InfContext *>(ctx);
  return c ? c->out : nullptr;
}

void spz_amd_inflate_close(void *ctx) {
  InfContext *c = static_cast<InfContext *>(ctx);
  if (c == nullptr) return;
  DeviceGuard guard;
  if (guard.enter(c->device) == SPZ_AMD_OK && c->block) scratch_release(c->device, c->block);
  delete c;
}

}  // extern "C"
