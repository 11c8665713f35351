// spz_lz77_core.hpp — zlib's level-6 LZ77 parse (deflate.c: deflate_slow + longest_match, as restated in
// spz_deflate.cpp) cut into three data-parallel stages, shared by the HIP kernels (spz_lz77.hip) and by a
// serial host model of the same stages (spz_lz77_model.cpp) that the CPU tests run:
//
//   1. chains   zlib walks, per position, the chain of earlier positions with the same 16-bit hash of three bytes
//               (its prev[] links; EVERY position is in the chains at level >= 4, so they are a pure function of
//               the input) and spends its budget of 128 steps mostly on strangers: the .spz sh bytes are
//               multiples of 8 or 16, so whole families of byte triples share a hash.  Two arrays replace the chain:
//               link[p] = distance to the nearest earlier position with the same value of a DIFFERENT, well-mixed
//               hash of the triple (hash2), and rank[p] = how many earlier positions share p's zlib hash (mod 2^16).
//               Walking link[] visits (almost) only positions with the same three bytes, and the difference of
//               two ranks says how many steps zlib's own walk needs from one to the other — kept per link as one
//               byte, delta[p] (chain_delta below), so that a walk adds bytes up and reads ranks only after a stranger.
//   2. matches  r128[p] / r32[p] = the input byte at p and what longest_match() returns at a loop top at p with a
//               chain budget of 128 / 32 (the budget is 32 when the previous match is >= good_match) and no
//               previous match.  A
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

// Geometry of the data-parallel parse.  A job is job_bytes of input: it runs deflate_slow's loop from its first
// position (lazy state reset) to the first loop top at or past its end, and records its lazy state at every
// loop top (one word per input position).  A second, short step continues each job into its successor's range
// until its state at a loop top equals the one the successor recorded there — from that position on the two
// parses are the same parse, so the successor's symbols take over (spz_deflate.cpp's splice; how many symbols
// the successor had emitted by then is counted by running its parse up to that position once more).  The
// meeting point is usually a few positions in, a few thousand after long runs, tens of thousands when long
// stretches of the input repeat; none inside the successor's range: the stage is run again with larger jobs
// (the tables do not depend on the jobs), and declined after the largest — the caller parses on the host instead.
constexpr uint32_t kJobSizes[4] = {W / 4, W / 2, 4 * W, 32 * W};
constexpr uint32_t kSmallestJob = W / 4;
constexpr uint32_t kTailWindow = W;  // loop tops of the caller's tail parse that the last job may meet
SPZ_LZ_HD uint32_t job_symbol_stride(uint32_t job_bytes) { return job_bytes + 8; }  // at most one symbol per position
// End of the range job j's continuation may run through: its successor's own range, or the tail parse's first loop tops.
SPZ_LZ_HD uint64_t stitch_end(uint64_t j, uint64_t n_jobs, uint32_t job_bytes, uint64_t head_end) {
  const uint64_t next = (j + 1) * job_bytes < head_end ? (j + 1) * job_bytes : head_end;
  if (j + 1 < n_jobs) return next + job_bytes < head_end ? next + job_bytes : head_end;
  return head_end + (job_bytes < kTailWindow ? job_bytes : kTailWindow);
}
// Positions past the last job's end that the tables must cover (loop tops of that job inside the tail window,
// plus one match), and the bytes past a position that a match compare may read.
constexpr uint32_t kTableSlack = kTailWindow + MAX_MATCH + 6;
constexpr uint32_t kReadAhead = MAX_MATCH + 4;

SPZ_LZ_HD uint32_t hash3(uint32_t b0, uint32_t b1, uint32_t b2) { return ((b0 << 12) ^ (b1 << 6) ^ b2) & HASH_MASK; }
// The other hash of the same three bytes: equal triples have equal values, unequal ones rarely.
// (15 bits: the kernel's table of newest positions is then 64 KiB and two of its workgroups share a CU; twice the
// strangers of a 16-bit value — 3 % of an .spz stream's candidates instead of 1.5 % — cost the walks less than that
// gains: table kernel 10.5 -> 6.1 ms, match kernel 24.0 -> 24.4 ms for 650 MB; with 14 bits 4.3 and 29.8 ms)
constexpr uint32_t HASH2_BITS = 15;
SPZ_LZ_HD uint32_t hash2(uint32_t b0, uint32_t b1, uint32_t b2) { return ((b0 | (b1 << 8) | (b2 << 16)) * 0x9E3779B1u) >> (32 - HASH2_BITS); }

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

// A table entry: the input byte at the position (what a literal there is), the match distance (< 32768) and
// length - 3, and whether there is a match at all.
SPZ_LZ_HD uint32_t encode_entry(uint32_t len, uint32_t dist, uint32_t byte) {
  return len >= MIN_MATCH ? (0x80000000u | ((len - MIN_MATCH) << 23) | (dist << 8) | byte) : byte;
}
SPZ_LZ_HD uint32_t entry_length(uint32_t e) { return (e >> 31) ? ((e >> 23) & 0xffu) + MIN_MATCH : 0u; }
SPZ_LZ_HD uint32_t entry_distance(uint32_t e) { return (e >> 8) & 0x7fffu; }
SPZ_LZ_HD uint32_t entry_byte(uint32_t e) { return e & 0xffu; }

// Stage 2 for one position p.  Positions are in the caller's coordinates (Pos: absolute int64_t on the host,
// window-relative int32_t in the kernel).  `data.load4(pos)` / `data.load8(pos)`: the four / eight input bytes at pos, little endian (pos up
// to p + kReadAhead); `link(pos)`, `delta(pos)`, `rank(pos)`: stage 1's values (ranks need a common origin only among
// the positions one walk can reach, and are read only after a stranger); `base`: base_at(p) in the same coordinates (anything at or below p - W stands
// for "not in reach").  p has a full lookahead.
//
// zlib's walk, restated on the positions with p's three bytes: its first candidate (chain index 0) must lie above
// the window base and at most MAX_DIST back, every later one strictly above `limit`; positions only fall along
// the chain, so the first position that fails ends the walk for every later one too, whether it is a stranger or
// not; the budget ends it at chain index 128; strangers can never be taken (a candidate needs 3 equal bytes).
// How far along zlib's chain a step of the hash2 chain goes, in one byte per position (stage 1b): for a position c
// whose link leads to t,
//   1 ... kChainFar   rank(c) - rank(t), the steps zlib's own walk needs from c to t, when t has c's three bytes
//                     (kChainFar: that many or more — the walk's budget is MAX_CHAIN, so it ends there either way);
//   kChainStranger    t has other bytes (it shares only the hash2 value): the walk passes over it.
// A walk adds these up instead of reading two ranks per candidate; after a stranger it finds its place again from the
// 16-bit ranks (rare: 1.5 % of an .spz stream's candidates).
constexpr uint32_t kChainFar = 250, kChainStranger = 255;
SPZ_LZ_HD uint32_t chain_delta(uint32_t rank_c, uint32_t rank_t, bool same_bytes) {
  if (!same_bytes) return kChainStranger;
  const uint32_t d = (rank_c - rank_t) & 0xffffu;
  return d < kChainFar ? d : kChainFar;
}

// The walk is a state machine of one candidate per step, so that the kernel can give a lane whose walk has ended the
// next position while its neighbours are still walking (walks differ in length by two orders of magnitude); the host
// model runs the same steps in a plain loop.  A step's next position and chain index come from link[] and delta[] at the
// CURRENT position, so the candidate's bytes are off the chain of dependent reads.
template <class Pos>
struct MatchWalk {
  uint64_t s8;  // the eight bytes at p
  Pos p, base, limit, cur;
  uint32_t k;   // chain index of cur, + 1 (0 at p itself)
  uint32_t best, best_dist, best32, dist32;  // ..32: what a budget of SHORT_CHAIN had found
  bool snapped, synced;

  template <class Data>
  SPZ_LZ_HD void start(const Data &data, Pos p_, Pos base_) {
    p = p_;
    base = base_;
    s8 = data.load8(p);
    limit = (p - base > (Pos)MAX_DIST) ? p - (Pos)MAX_DIST : base;
    k = 0;
    best = MIN_MATCH - 1;
    best_dist = 0;
    best32 = 0;
    dist32 = 0;
    snapped = false;
    synced = true;
    cur = p;
  }

  // One position of the hash2 chain.  false: the walk is over.  Written as selects rather than nested conditions: on
  // the GPU every nested condition is a handful of scalar mask instructions per round for the whole wave, and those —
  // not the reads — were what a round's time went to (profiles/README.md, round 3: 100 scalar instructions per round).
  // Two real branches are left, both rare: finding the chain index again after a stranger, and matches of 8+ bytes.
  template <class Data, class Link, class Delta, class Rank>
  SPZ_LZ_HD bool step(const Data &data, const Link &link, const Delta &delta, const Rank &rank) {
    const uint32_t gap = link(cur);
    const uint32_t dk = delta(cur);
    cur -= (Pos)gap;  // gap 0: cur itself, whose bytes are read and ignored
    // ONE eight-byte read per candidate: its match length for all but the rare candidate that agrees on all eight
    // (zlib's own order — the byte at the best length so far first, then the first two, then the rest — is a shortcut
    // to the same length), and, after a stranger, whether this one has p's three bytes again.
    const uint64_t x8 = data.load8(cur) ^ s8;
    // out of reach for the first candidate and for any later one
    const bool reach = (gap != 0u) & (cur > base) & (p - cur <= (Pos)MAX_DIST);
    const bool same = (x8 & 0xffffffull) == 0ull;
    // passed over: not p's bytes (known from delta[] while the walk is among p's bytes, from the bytes after a stranger,
    // when delta[] is about the stranger's bytes)
    const bool skip = synced ? dk == kChainStranger : !same;
    if (!synced & !skip & reach) {
      const uint32_t far = (rank(p) - rank(cur)) & 0xffffu;
      k = far < kChainFar ? far : kChainFar;
    } else {
      k += (synced & !skip) ? dk : 0u;
    }
    synced = !skip;
    const bool cand = reach & !skip;
    // chain index k - 1 >= MAX_CHAIN, or a later candidate at or below the limit
    bool over = (!reach) | (cand & ((k > MAX_CHAIN) | ((k > 1u) & (cur <= limit))));
    const bool take = cand & !over;
    const bool snap = take & (k > SHORT_CHAIN) & !snapped;  // what a budget of 32 has found
    best32 = snap ? best : best32;
    dist32 = snap ? best_dist : dist32;
    snapped |= snap;
    uint32_t len = (uint32_t)__builtin_ctzll(x8 | (1ull << 63)) >> 3;  // 3 ... 7 for a candidate
    if (take & (x8 == 0ull)) {
      len = 8;
      while (len < MAX_MATCH) {
        const uint32_t x = data.load4(p + (Pos)len) ^ data.load4(cur + (Pos)len);
        if (x != 0u) {
          len += (uint32_t)__builtin_ctz(x) >> 3;
          break;
        }
        len += 4;
      }
      if (len > MAX_MATCH) len = MAX_MATCH;
    }
    const bool better = take & (len > best);
    best = better ? len : best;
    best_dist = better ? (uint32_t)(p - cur) : best_dist;
    over |= better & (len >= NICE_MATCH);
    return !over;
  }

  SPZ_LZ_HD void finish(uint32_t *r128, uint32_t *r32) const {
    const uint32_t r = encode_entry(best, best_dist, (uint32_t)s8 & 0xffu);
    *r32 = snapped ? encode_entry(best32, dist32, (uint32_t)s8 & 0xffu) : r;  // not snapped: the walk ended inside the short budget
    *r128 = r;
  }
};

template <class Pos, class Data, class Link, class Delta, class Rank>
SPZ_LZ_HD void find_matches(const Data &data, const Link &link, const Delta &delta, const Rank &rank, Pos p, Pos base, uint32_t *r128,
                            uint32_t *r32) {
  MatchWalk<Pos> w;
  w.start(data, p, base);
  while (w.step(data, link, delta, rank)) {
  }
  w.finish(r128, r32);
}

// ---- stage 3 -------------------------------------------------------------------------------------------
template <class Pos>
struct LazyState {
  uint32_t match_available = 0;
  uint32_t match_length = MIN_MATCH - 1;
  uint32_t byte_before = 0;  // input byte at strstart - 1, known whenever match_available is set
  Pos match_start = 0;
};

// The record spz_deflate.cpp's jobs compare when they splice (TopRec::state).
template <class Pos>
SPZ_LZ_HD uint32_t pack_state(const LazyState<Pos> &st, Pos strstart) {
  const uint32_t dist = st.match_length >= MIN_MATCH ? (uint32_t)(strstart - st.match_start) : 0u;
  return 0x80000000u | (st.match_available << 30) | st.match_length | ((dist & 0xffffu) << 9);
}

// One iteration of deflate_slow's loop at a loop top `strstart` with a full lookahead.  `e128(pos)`, `e32(pos)`:
// stage 2's tables; `emit(dist, lc)`: one symbol (dist 0 = literal lc, else a match of lc + 3 bytes).  The literal
// a loop top emits is the byte before it, and the loop top before it has read that position's entry.
template <class Pos, class E128, class E32, class Emit>
SPZ_LZ_HD void lazy_step(Pos &strstart, LazyState<Pos> &st, const E128 &e128, const E32 &e32, Emit &emit) {
  const uint32_t prev_length = st.match_length;
  const Pos prev_match = st.match_start;
  const uint32_t literal = st.byte_before;
  st.match_length = MIN_MATCH - 1;
  if (prev_length < MAX_LAZY) {
    const uint32_t e = prev_length >= GOOD_MATCH ? e32(strstart) : e128(strstart);
    const uint32_t len = entry_length(e), dist = entry_distance(e);
    st.byte_before = entry_byte(e);  // for the next loop top, if it is strstart + 1
    if (len > prev_length) {
      st.match_length = (len == MIN_MATCH && dist > TOO_FAR) ? MIN_MATCH - 1 : len;
      st.match_start = strstart - (Pos)dist;
    }
  }
  if (prev_length >= MIN_MATCH && st.match_length <= prev_length) {
    emit((uint32_t)(strstart - 1 - prev_match), prev_length - MIN_MATCH);
    strstart += (Pos)(prev_length - 1);
    st.match_available = 0;
    st.match_length = MIN_MATCH - 1;
  } else if (st.match_available) {
    emit(0u, literal);
    ++strstart;
  } else {
    st.match_available = 1;
    ++strstart;
  }
}

}  // namespace spz_lz
