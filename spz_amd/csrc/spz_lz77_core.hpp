// spz_lz77_core.hpp — zlib's level-6 LZ77 parse (deflate.c: deflate_slow + longest_match, as restated in
// spz_deflate.cpp) cut into three data-parallel stages, shared by the HIP kernels (spz_lz77.hip) and by a
// serial host model of the same stages (spz_lz77_model.cpp) that the CPU tests run:
//
//   1. links    link[p] = distance from p to the nearest earlier position with the same 3-byte hash (zlib's
//               prev[] chain), 0 when there is none within 32 KiB.  zlib inserts EVERY position into its hash
//               chains at level >= 4, so the chains are a pure function of the input.
//   2. matches  r128[p] / r32[p] = what longest_match() returns at a loop top at p with a chain budget of 128 /
//               32 (the budget is 32 when the previous match is >= good_match) and no previous match.  A
//               previous match of length L only raises the bar a candidate has to pass: longest_match with
//               prev_length = L returns the same (length, start) when that length exceeds L, and "nothing
//               better" otherwise (a candidate is taken iff it is longer than the best so far; the walk stops
//               at the first candidate of nice_match bytes, which is always taken).  So the two tables answer
//               every call the lazy matcher can make, and computing them needs no parse state at all.
//   3. lazy     deflate_slow's loop over loop tops with longest_match() replaced by the table lookup: O(1) per
//               position, serial inside a job, jobs spliced where their lazy states agree (spz_deflate.cpp).
//
// The window base (which positions zlib's sliding window has already dropped) is base_at(), a closed form
// of the position; only positions with a full 262-byte lookahead are handled here, the end of the input
// stays with spz_deflate.cpp's serial job.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define SPZ_LZ_HD __host__ __device__ __forceinline__
#else
#define SPZ_LZ_HD inline
#endif

namespace spz_lz {

// deflate.c parameters for level 6, windowBits 15, memLevel 9 (load-spz.cc:190)
constexpr uint32_t W = 32768;
constexpr uint32_t HASH_MASK = 65535;  // hash_bits = memLevel + 7, hash_shift = 6
constexpr uint32_t MIN_MATCH = 3, MAX_MATCH = 258, MIN_LOOKAHEAD = MAX_MATCH + MIN_MATCH + 1;
constexpr uint32_t MAX_DIST = W - MIN_LOOKAHEAD, TOO_FAR = 4096;
constexpr uint32_t GOOD_MATCH = 8, MAX_LAZY = 16, NICE_MATCH = 128, MAX_CHAIN = 128, SHORT_CHAIN = MAX_CHAIN >> 2;

// Geometry of the data-parallel parse.  A job is kJobBytes of input; job j+1 first parses its first
// kRecordWindow positions and records its lazy state at every loop top, job j then runs until its own state at a
// position inside that window equals the recorded one (declined when that does not happen: the caller parses
// on the host instead).
constexpr uint32_t kJobBytes = 2 * W;
constexpr uint32_t kRecordWindow = 8192;
constexpr uint32_t kJobSymbolStride = kJobBytes + kRecordWindow + MAX_MATCH + 6;  // symbols a job can emit, rounded up
// Positions past the last job's end that the tables must cover (loop tops of that job inside its successor's
// record window, plus one match), and the bytes past a position that a match compare may read.
constexpr uint32_t kTableSlack = kRecordWindow + MAX_MATCH + 6;
constexpr uint32_t kReadAhead = MAX_MATCH + 4;

SPZ_LZ_HD uint32_t hash3(uint32_t b0, uint32_t b1, uint32_t b2) { return ((b0 << 12) ^ (b1 << 6) ^ b2) & HASH_MASK; }

// Window base (absolute position of window[0]) after fill_window at a loop top at position s (the same
// function as spz_deflate.cpp's, which checks it against the simulated window in every job).
SPZ_LZ_HD uint64_t base_at(uint64_t s, uint64_t size) {
  const uint64_t kfull = size / W;
  uint64_t a = (s + 261) / W;
  a = a >= 1 ? a - 1 : 0;
  const uint64_t cap = kfull >= 1 ? kfull - 1 : 0;
  uint64_t k = a < cap ? a : cap;
  if (kfull >= 1 && k == kfull - 1 && s + 262 >= W * (kfull + 1)) k = kfull;
  return k * W;
}

SPZ_LZ_HD uint32_t encode_match(uint32_t len, uint32_t dist) { return len >= MIN_MATCH ? (len << 16) | dist : 0u; }

SPZ_LZ_HD uint32_t ctz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_ctz(x);
#else
  return (uint32_t)__builtin_ctz(x);
#endif
}

// Stage 2 for one position.  `data.load4(pos)`: the four input bytes at pos, little endian (pos up to
// p + kReadAhead); `link(pos)`: stage 1's value.  p must have a full lookahead (p + MIN_LOOKAHEAD <= size).
template <class Data, class Link>
SPZ_LZ_HD void find_matches(const Data &data, const Link &link, uint64_t p, uint64_t size, uint32_t *r128,
                            uint32_t *r32) {
  *r128 = 0;
  *r32 = 0;
  uint32_t gap = link(p);
  if (gap == 0 || gap > MAX_DIST) return;          // hash_head == NIL, or strstart - hash_head > MAX_DIST
  const uint64_t base = base_at(p, size);
  uint64_t cur = p - gap;
  if (cur <= base) return;                           // the window has dropped it (index 0 is NIL too)
  const uint64_t limit = (p - base > MAX_DIST) ? p - MAX_DIST : base;
  uint32_t best = MIN_MATCH - 1, best_dist = 0, k = 0;
  const uint32_t s4 = data.load4(p);
  for (;;) {
    const uint32_t m4 = data.load4(cur);
    if (((m4 ^ s4) & 0xffffffu) == 0u &&
        (best < MIN_MATCH || ((data.load4(cur + best) ^ data.load4(p + best)) & 0xffu) == 0u)) {
      uint32_t len = MIN_MATCH;
      while (len < MAX_MATCH) {
        const uint32_t x = data.load4(p + len) ^ data.load4(cur + len);
        if (x != 0u) {
          len += ctz32(x) >> 3;
          break;
        }
        len += 4;
      }
      if (len > MAX_MATCH) len = MAX_MATCH;
      if (len > best) {
        best = len;
        best_dist = (uint32_t)(p - cur);
        if (len >= NICE_MATCH) {
          const uint32_t r = encode_match(best, best_dist);
          if (k < SHORT_CHAIN) *r32 = r;
          *r128 = r;
          return;
        }
      }
    }
    ++k;
    if (k == SHORT_CHAIN) *r32 = encode_match(best, best_dist);
    if (k == MAX_CHAIN) break;
    gap = link(cur);
    if (gap == 0 || gap > cur) break;
    cur -= gap;
    if (cur <= limit) break;
  }
  const uint32_t r = encode_match(best, best_dist);
  if (k < SHORT_CHAIN) *r32 = r;
  *r128 = r;
}

// ---- stage 3 -------------------------------------------------------------------------------------------
struct LazyState {
  uint32_t match_available = 0;
  uint32_t match_length = MIN_MATCH - 1;
  uint64_t match_start = 0;
};

// The record spz_deflate.cpp's jobs compare when they splice (TopRec::state).
SPZ_LZ_HD uint32_t pack_state(const LazyState &st, uint64_t strstart) {
  const uint32_t dist = st.match_length >= MIN_MATCH ? (uint32_t)(strstart - st.match_start) : 0u;
  return 0x80000000u | (st.match_available << 30) | st.match_length | ((dist & 0xffffu) << 9);
}

// One iteration of deflate_slow's loop at a loop top `strstart` with a full lookahead.  `r128(pos)`, `r32(pos)`:
// stage 2's tables; `byte(pos)`: input byte; `emit(dist, lc)`: one symbol (dist 0 = literal lc, else a match of
// lc + 3 bytes).
template <class R128, class R32, class Byte, class Emit>
SPZ_LZ_HD void lazy_step(uint64_t &strstart, LazyState &st, const R128 &r128, const R32 &r32, const Byte &byte,
                         Emit &emit) {
  const uint32_t prev_length = st.match_length;
  const uint64_t prev_match = st.match_start;
  st.match_length = MIN_MATCH - 1;
  if (prev_length < MAX_LAZY) {
    const uint32_t r = prev_length >= GOOD_MATCH ? r32(strstart) : r128(strstart);
    const uint32_t len = r >> 16, dist = r & 0xffffu;
    if (len > prev_length) {
      st.match_length = (len == MIN_MATCH && dist > TOO_FAR) ? MIN_MATCH - 1 : len;
      st.match_start = strstart - dist;
    }
  }
  if (prev_length >= MIN_MATCH && st.match_length <= prev_length) {
    emit((uint32_t)(strstart - 1 - prev_match), prev_length - MIN_MATCH);
    strstart += prev_length - 1;
    st.match_available = 0;
    st.match_length = MIN_MATCH - 1;
  } else if (st.match_available) {
    emit(0u, (uint32_t)byte(strstart - 1));
    ++strstart;
  } else {
    st.match_available = 1;
    ++strstart;
  }
}

}  // namespace spz_lz
