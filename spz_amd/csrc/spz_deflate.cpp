// spz_deflate.cpp — multi-threaded gzip writer with zlib's exact output.
//
// Why: the reference's container step (compressGzipped, /root/reference/src/cc/load-spz.cc:186-214:
// zlib, default level, one deflate stream) is ~90 % of an end-to-end saveSpz once the quantise step
// runs on the GPU, and the .spz bytes have to stay identical to the
// reference's.  A pigz-style writer is fast but produces different bytes; this one produces the SAME
// bytes as zlib 1.2.11 with deflateInit2(level 6, windowBits 15 + gzip, memLevel 9, default strategy),
// in parallel.  zlib is a third-party dependency of the reference (system zlib 1.2.11);
// what follows restates its published algorithm (RFC 1951/1952 format; deflate.c's lazy matcher
// `deflate_slow` + `longest_match`, trees.c's block writer) and is pinned by byte comparison with the
// system zlib in tests/test_exact_gzip.py and by a self-check against zlib at run time.
//
// Acknowledgement: longest_match / deflate_slow / the block writer below follow deflate.c and trees.c of zlib 1.2.11,
// (C) 1995-2017 Jean-loup Gailly and Mark Adler (zlib licence), closely enough to give identical output; this is an
// altered restatement, plainly marked as such, not the original software.
//
// How it can be parallel and exact:
//  * zlib's LZ77 parse at a position depends only on (a) the previous 32 KiB of input, through hash
//    chains that contain EVERY earlier position (level >= 4 inserts all of them), (b) the phase of its
//    64 KiB sliding window, which is a pure function of the position, and (c) three words of lazy-match
//    state (match_available, match_length, match_start).  A parse job that starts at a multiple of
//    32 KiB with (a) and (b) rebuilt exactly and (c) reset therefore differs from the serial parse only
//    until the two reach the same position with the same (c) — after that they are identical forever.
//    Each job runs 32 KiB into its successor's range, both record (c) at every loop top there, and the
//    symbol streams are spliced at the first common position with equal state.
//  * Block boundaries are every 32767 symbols of the spliced stream (lit_bufsize - 1), and each block's
//    bits are a deterministic function of its symbols (trees.c), so blocks are encoded independently and
//    their bit strings concatenated.
//  * Reads past the end of the input (the matcher looks up to 258 bytes ahead) see what zlib's window
//    holds there: the bytes 32 KiB earlier (stale upper half of the window), reproduced in a padded tail.
#include "spz_deflate.hpp"
#include "spz_huff_core.hpp"
#include "spz_host_util.hpp"
#include "spz_lz77_core.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <new>
#include <system_error>
#include <thread>

namespace spz {
namespace exactgz {
namespace {

// ---- deflate.c parameters for level 6, windowBits 15, memLevel 9 ----------------------------------
constexpr uint32_t W = 32768, WMASK = W - 1;
constexpr uint32_t HASH_SIZE = 65536, HASH_MASK = HASH_SIZE - 1;  // hash_bits = memLevel + 7, hash_shift = 6
constexpr uint32_t MIN_MATCH = 3, MAX_MATCH = 258, MIN_LOOKAHEAD = MAX_MATCH + MIN_MATCH + 1;
constexpr uint32_t MAX_DIST = W - MIN_LOOKAHEAD, TOO_FAR = 4096;
constexpr uint32_t GOOD_MATCH = 8, MAX_LAZY = 16, NICE_MATCH = 128, MAX_CHAIN = 128;
constexpr size_t BLOCK_SYMS = 32767;  // lit_bufsize - 1 with lit_bufsize = 1 << (memLevel + 6)
constexpr uint32_t OVERLAP = W;

inline uint32_t hash3(const uint8_t *p) {
  return ((uint32_t(p[0]) << 12) ^ (uint32_t(p[1]) << 6) ^ uint32_t(p[2])) & HASH_MASK;
}

// Window base (absolute position of window[0]) that zlib has after fill_window at a loop top at
// position s: the window slides by 32 KiB at the first loop top with lookahead < 262 and
// strstart - base >= 32768 + MAX_DIST.
uint64_t base_at(uint64_t s, uint64_t size) {
  const uint64_t kfull = size / W;  // slides k with W*(k+1) <= size need s >= W*(k+1) - 261
  uint64_t a = (s + 261) / W;
  a = a >= 1 ? a - 1 : 0;
  uint64_t k = std::min<uint64_t>(a, kfull >= 1 ? kfull - 1 : 0);
  // the slide whose refill is cut short by the end of the input happens one position earlier
  if (kfull >= 1 && k == kfull - 1 && s + 262 >= W * (kfull + 1)) k = kfull;
  return k * W;
}

struct TopRec {  // lazy-match state at a loop top, after fill_window
  uint32_t state = 0;     // bit 31 valid, bit 30 match_available, bits 0-8 match_length, bits 9-24 distance of match_start
  uint32_t symcount = 0;  // symbols this job had emitted before this loop top
};

// ---- one parse job: deflate_slow over [begin, stop) ---------------------------------------------------
struct Job {
  const uint8_t *d = nullptr;  // d[abs] valid for every absolute position the job can touch
  uint64_t size = 0, begin = 0, stop = 0;
  bool last = false;
  uint64_t rec_succ_lo = 0, rec_pred_lo = 0;  // first positions of the two record windows
  std::vector<TopRec> rec_succ, rec_pred;     // as successor: [begin, ...); as predecessor: [next begin, ...)
  std::vector<uint16_t> sym_dist;             // 0 = literal
  std::vector<uint8_t> sym_lc;                // literal byte, or match length - 3
  bool phase_ok = true;
  bool check_phase = false;    // tests: compare the tracked window base with base_at() at every loop top
  bool tail_literal = false;   // the last symbol was the pending literal tallied after the loop (no flush check there)

  // Everything the loop touches lives in locals: the symbol stores go through uint8_t*, which may alias
  // any member, and would otherwise force reloads of every pointer after each symbol.
  void run() {
    // head: absolute position of the newest string per hash (0 = none).  prev: distance from a position to
    // the previous string with its hash, 0 when there is none within 64 KiB — a link that far back is beyond
    // MAX_DIST for this and every later position, so it ends the chain exactly where zlib's does.  16-bit
    // links keep the table the chain walk depends on at 64 KiB, like zlib's.
    std::vector<uint32_t> head_v(HASH_SIZE, 0);
    std::vector<uint16_t> prev_v(W, 0);
    uint32_t *const head = head_v.data();
    uint16_t *const prev = prev_v.data();
    const uint8_t *const dd = d;
    const uint64_t sz = size, stop_at = stop;
    const bool is_last = last, check = check_phase;
    uint64_t strstart = begin;
    uint64_t base = begin == 0 ? 0 : begin - W;
    uint64_t filled_end = std::min<uint64_t>(sz, base + 2 * W);
    // zlib's rolling hash (UPDATE_HASH): positions are inserted in increasing order without gaps, so
    // ins_h always holds the two bytes before the one INSERT_STRING shifts in.
    uint32_t ins_h = ((uint32_t(dd[base]) << 6) ^ uint32_t(dd[base + 1])) & HASH_MASK;
#define SPZ_INSERT_STRING(pos, old_head)                                   \
  do {                                                                     \
    ins_h = ((ins_h << 6) ^ uint32_t(dd[(pos) + 2])) & HASH_MASK;          \
    (old_head) = head[ins_h];                                              \
    {                                                                      \
      const uint64_t gap_ = (pos) - (old_head);                            \
      prev[(pos) & WMASK] = ((old_head) != 0 && gap_ <= 0xffffu) ? static_cast<uint16_t>(gap_) : 0; \
    }                                                                      \
    head[ins_h] = static_cast<uint32_t>(pos);                              \
  } while (0)
    uint32_t scratch;
    for (uint64_t p = base; p < begin; ++p) SPZ_INSERT_STRING(p, scratch);  // every earlier position is in the chains
    uint32_t match_available = 0, match_length = MIN_MATCH - 1;
    uint64_t match_start = 0;
    // at most one symbol per input byte; written by index, trimmed at the end
    const size_t cap = static_cast<size_t>((is_last ? sz : stop_at + MAX_MATCH + 2) - begin) + 2;
    sym_dist.resize(cap);
    sym_lc.resize(cap);
    uint16_t *const out_dist = sym_dist.data();
    uint8_t *const out_lc = sym_lc.data();
    size_t nsym = 0;
    bool ok = true, tail = false;
    TopRec *const rs = rec_succ.data();
    TopRec *const rp = rec_pred.data();
    const uint64_t rs_lo = rec_succ_lo, rp_lo = rec_pred_lo, rs_n = rec_succ.size(), rp_n = rec_pred.size();
    // loop tops before rec_from need no record; past rec_to the job is done
    const uint64_t rec_from = rs_n ? rs_lo : (rp_n ? rp_lo : ~uint64_t(0));
    for (;;) {
      uint64_t lookahead = filled_end - strstart;
      if (lookahead < MIN_LOOKAHEAD) {  // fill_window
        do {
          if (strstart - base >= W + MAX_DIST) {
            base += W;
            if (base != base_at(strstart, sz)) ok = false;
          }
          if (filled_end == sz) break;
          filled_end = std::min<uint64_t>(sz, base + 2 * W);
          lookahead = filled_end - strstart;
        } while (lookahead < MIN_LOOKAHEAD && filled_end != sz);
        lookahead = filled_end - strstart;
        if (lookahead == 0) break;
        if (check && base != base_at(strstart, sz)) ok = false;
      }
      // ---- loop top: record the lazy state where a neighbour may want to splice
      if (strstart >= rec_from) {
        const uint32_t dist = match_length >= MIN_MATCH ? static_cast<uint32_t>(strstart - match_start) : 0;
        const uint32_t st = 0x80000000u | (match_available << 30) | match_length | ((dist & 0xffffu) << 9);
        if (strstart - rs_lo < rs_n) rs[strstart - rs_lo] = {st, static_cast<uint32_t>(nsym)};  // unsigned compare
        if (strstart - rp_lo < rp_n) rp[strstart - rp_lo] = {st, static_cast<uint32_t>(nsym)};
        if (check && base != base_at(strstart, sz)) ok = false;
        if (!is_last && strstart >= stop_at) break;
      }
      uint32_t hash_head = 0;
      if (lookahead >= MIN_MATCH) SPZ_INSERT_STRING(strstart, hash_head);
      const uint32_t prev_length = match_length;
      const uint64_t prev_match = match_start;
      match_length = MIN_MATCH - 1;
      if (hash_head > base && prev_length < MAX_LAZY && strstart - hash_head <= MAX_DIST) {
        // longest_match
        uint32_t chain = MAX_CHAIN, best_len = prev_length, nice = NICE_MATCH;
        const uint64_t limit = (strstart - base > MAX_DIST) ? strstart - MAX_DIST : base;
        const uint8_t *const scan = dd + strstart;
        auto load16 = [](const uint8_t *q) {
          uint16_t v;
          std::memcpy(&v, q, 2);
          return v;
        };
        uint16_t scan_end = load16(scan + best_len - 1);  // the bytes at best_len - 1 and best_len, as one compare
        const uint16_t scan_start = load16(scan);
        if (prev_length >= GOOD_MATCH) chain >>= 2;
        if (nice > lookahead) nice = static_cast<uint32_t>(lookahead);
        uint64_t cur = hash_head;
        do {
          const uint8_t *const match = dd + cur;
          if (load16(match + best_len - 1) == scan_end && load16(match) == scan_start) {
            uint32_t len = 2;  // bytes 0 and 1 are equal; MAX_MATCH - 2 = 32 * 8
            for (; len < MAX_MATCH; len += 8) {
              uint64_t a, b;
              std::memcpy(&a, scan + len, 8);
              std::memcpy(&b, match + len, 8);
              const uint64_t x = a ^ b;
              if (x != 0) {
                len += static_cast<uint32_t>(__builtin_ctzll(x)) >> 3;
                break;
              }
            }
            if (len > best_len) {
              match_start = cur;
              best_len = len;
              if (len >= nice) break;
              scan_end = load16(scan + best_len - 1);
            }
          }
          const uint32_t gap = prev[cur & WMASK];
          if (gap == 0) break;
          cur -= gap;
        } while (cur > limit && --chain != 0);
        match_length = best_len <= lookahead ? best_len : static_cast<uint32_t>(lookahead);
        if (match_length == MIN_MATCH && strstart - match_start > TOO_FAR) match_length = MIN_MATCH - 1;
      }
      if (prev_length >= MIN_MATCH && match_length <= prev_length) {
        const uint64_t max_insert = strstart + lookahead - MIN_MATCH;
        out_dist[nsym] = static_cast<uint16_t>(strstart - 1 - prev_match);
        out_lc[nsym++] = static_cast<uint8_t>(prev_length - MIN_MATCH);
        uint32_t n = prev_length - 2;
        do {
          if (++strstart <= max_insert) SPZ_INSERT_STRING(strstart, scratch);
        } while (--n != 0);
        match_available = 0;
        match_length = MIN_MATCH - 1;
        ++strstart;
      } else if (match_available) {
        out_dist[nsym] = 0;
        out_lc[nsym++] = dd[strstart - 1];
        ++strstart;
      } else {
        match_available = 1;
        ++strstart;
      }
    }
    if (is_last && match_available) {  // end of input: the pending literal
      out_dist[nsym] = 0;
      out_lc[nsym++] = dd[strstart - 1];
      tail = true;
    }
    sym_dist.resize(nsym);
    sym_lc.resize(nsym);
    tail_literal = tail;
    if (!ok) phase_ok = false;
#undef SPZ_INSERT_STRING
  }
};

// ---- trees.c: static tables -----------------------------------------------------------------------
constexpr int L_CODES = 286, D_CODES = 30, BL_CODES = 19, HEAP_SIZE = 2 * L_CODES + 1, LITERALS = 256, END_BLOCK = 256;
constexpr int REP_3_6 = 16, REPZ_3_10 = 17, REPZ_11_138 = 18, MAX_BITS = 15, MAX_BL_BITS = 7;
const int kExtraL[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const int kExtraD[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const int kExtraBl[19] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};
const uint8_t kBlOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct StaticTables {
  uint8_t length_code[256];
  uint8_t dist_code[512];
  int base_length[29];
  int base_dist[30];
  uint16_t sl_len[L_CODES + 2], sl_code[L_CODES + 2];  // static literal/length tree
  uint16_t sd_code[D_CODES];                           // static distance tree: 5-bit codes
  StaticTables() {
    int length = 0;
    for (int code = 0; code < 28; ++code) {
      base_length[code] = length;
      for (int n = 0; n < (1 << kExtraL[code]); ++n) length_code[length++] = static_cast<uint8_t>(code);
    }
    base_length[28] = 0;
    length_code[255] = 28;  // length 258 has a code of its own
    int dist = 0;
    for (int code = 0; code < 16; ++code) {
      base_dist[code] = dist;
      for (int n = 0; n < (1 << kExtraD[code]); ++n) dist_code[dist++] = static_cast<uint8_t>(code);
    }
    dist >>= 7;
    for (int code = 16; code < D_CODES; ++code) {
      base_dist[code] = dist << 7;
      for (int n = 0; n < (1 << (kExtraD[code] - 7)); ++n) dist_code[256 + dist++] = static_cast<uint8_t>(code);
    }
    uint16_t bl_count[MAX_BITS + 1] = {};
    int n = 0;
    while (n <= 143) sl_len[n++] = 8, bl_count[8]++;
    while (n <= 255) sl_len[n++] = 9, bl_count[9]++;
    while (n <= 279) sl_len[n++] = 7, bl_count[7]++;
    while (n <= 287) sl_len[n++] = 8, bl_count[8]++;
    gen_codes(sl_len, sl_code, L_CODES + 1, bl_count);
    for (int c = 0; c < D_CODES; ++c) sd_code[c] = static_cast<uint16_t>(reverse(static_cast<unsigned>(c), 5));
  }
  static unsigned reverse(unsigned code, int len) {
    unsigned res = 0;
    do {
      res |= code & 1;
      code >>= 1;
      res <<= 1;
    } while (--len > 0);
    return res >> 1;
  }
  static void gen_codes(const uint16_t *len, uint16_t *code_out, int max_code, const uint16_t *bl_count) {
    uint16_t next_code[MAX_BITS + 1];
    unsigned code = 0;
    for (int bits = 1; bits <= MAX_BITS; ++bits) {
      code = (code + bl_count[bits - 1]) << 1;
      next_code[bits] = static_cast<uint16_t>(code);
    }
    for (int n = 0; n <= max_code; ++n) {
      const int l = len[n];
      if (l == 0) continue;
      code_out[n] = static_cast<uint16_t>(reverse(next_code[l]++, l));
    }
  }
  inline int d_code(unsigned dist) const { return dist < 256 ? dist_code[dist] : dist_code[256 + (dist >> 7)]; }
};
const StaticTables &tables() {
  static const StaticTables t;
  return t;
}

// ---- trees.c: build_tree / gen_bitlen / gen_codes for one tree: spz_huff_core.hpp (shared with the device) ----
struct Tree {
  uint16_t freq[HEAP_SIZE] = {};
  uint16_t len[HEAP_SIZE] = {};
  uint16_t dad[HEAP_SIZE] = {};
  uint16_t code[HEAP_SIZE] = {};
  int max_code = 0;
};

// Note on freq width: zlib's Freq is 16 bits as well; a block holds at most 32767 symbols + END_BLOCK and
// internal nodes sum to at most 32768 per tree, so nothing overflows.

struct BitWriter {
  std::vector<uint8_t> out;
  uint64_t acc = 0;
  int nbits = 0;
  uint64_t total = 0;
  inline void put(uint32_t value, int length) {
    acc |= static_cast<uint64_t>(value) << nbits;
    nbits += length;
    total += static_cast<uint64_t>(length);
    while (nbits >= 8) {
      out.push_back(static_cast<uint8_t>(acc));
      acc >>= 8;
      nbits -= 8;
    }
  }
  void align() {  // bi_windup
    if (nbits > 0) {
      total += static_cast<uint64_t>(8 - nbits);
      out.push_back(static_cast<uint8_t>(acc));
      acc = 0;
      nbits = 0;
    }
  }
};

struct Seg {
  const uint16_t *dist;
  const uint8_t *lc;
  size_t n;
};

enum Choice { STORED = 0, STATIC = 1, DYNAMIC = 2 };

struct BlockLayout {             // what the stored / static / dynamic choice and the bit offsets need
  size_t nsyms = 0;
  bool last = false;
  uint64_t bytes = 0;          // input bytes the symbols cover (stored_len)
  uint32_t last_sym_len = 0;   // input bytes of the last symbol
  uint64_t start = 0;          // absolute input position of the first byte
  int64_t opt_len = 0, static_len = 0;
  Choice choice = DYNAMIC;
  uint64_t bit_start = 0, bit_len = 0;
};
struct Block : BlockLayout {
  std::vector<Seg> segs;
  Tree lt, dt, bt;
  int max_blindex = 0;
  std::vector<uint8_t> bits;   // encoded block, first byte holds (bit_start & 7) leading zero bits
};

template <class F>
void walk_lengths(const Tree &t, int max_code, F emit) {
  huff::walk_lengths(t.len, max_code, emit);
}

void plan_trees(Block &b) {  // the first half of _tr_flush_block, from the tallied frequencies
  uint16_t heap[HEAP_SIZE], bl_count[MAX_BITS + 1], next_code[MAX_BITS + 1];
  uint8_t depth[HEAP_SIZE];
  huff::Work w = {heap, depth, bl_count, next_code, 0, 0, 0, 0};
  huff::TreeRef<uint16_t> lt = {b.lt.freq, b.lt.dad, b.lt.len, b.lt.code, 0}, dt = {b.dt.freq, b.dt.dad, b.dt.len, b.dt.code, 0},
                          bt = {b.bt.freq, b.bt.dad, b.bt.len, b.bt.code, 0};
  b.max_blindex = huff::plan_trees(lt, dt, bt, w, static_cast<const uint16_t *>(b.lt.len), static_cast<const uint16_t *>(b.dt.len));
  b.lt.max_code = lt.max_code;
  b.dt.max_code = dt.max_code;
  b.bt.max_code = bt.max_code;
  b.opt_len = w.opt_len;
  b.static_len = w.static_len;
}

void plan_block(Block &b) {  // the tally loop, then the trees
  const StaticTables &T = tables();
  uint64_t bytes = 0;
  uint32_t last_len = 0;
  for (const Seg &s : b.segs) {
    for (size_t i = 0; i < s.n; ++i) {
      const unsigned dist = s.dist[i], lc = s.lc[i];
      if (dist == 0) {
        b.lt.freq[lc]++;
        last_len = 1;
      } else {
        b.lt.freq[T.length_code[lc] + LITERALS + 1]++;
        b.dt.freq[T.d_code(dist - 1)]++;
        last_len = lc + MIN_MATCH;
      }
      bytes += last_len;
    }
  }
  b.bytes = bytes;
  b.last_sym_len = last_len;
  plan_trees(b);
}

// What a block writes before its first symbol: the type bits and, for a dynamic block, the trees; for a stored
// block the padding and the two length words.
void write_block_header(const Block &b, BitWriter &w) {
  const uint32_t last = b.last ? 1u : 0u;
  if (b.choice == STORED) {
    w.put((0u << 1) + last, 3);
    w.align();
    const uint32_t len = static_cast<uint32_t>(b.bytes);
    w.put(len & 0xffffu, 16);
    w.put(~len & 0xffffu, 16);
  } else if (b.choice == STATIC) {
    w.put((1u << 1) + last, 3);
  } else {
    w.put((2u << 1) + last, 3);
    const int lcodes = b.lt.max_code + 1, dcodes = b.dt.max_code + 1, blcodes = b.max_blindex + 1;
    w.put(static_cast<uint32_t>(lcodes - 257), 5);
    w.put(static_cast<uint32_t>(dcodes - 1), 5);
    w.put(static_cast<uint32_t>(blcodes - 4), 4);
    for (int rank = 0; rank < blcodes; ++rank) w.put(b.bt.len[kBlOrder[rank]], 3);
    auto send = [&](int code, int extra_value, int extra_bits) {
      w.put(b.bt.code[code], b.bt.len[code]);
      if (extra_bits) w.put(static_cast<uint32_t>(extra_value), extra_bits);
    };
    walk_lengths(b.lt, lcodes - 1, send);
    walk_lengths(b.dt, dcodes - 1, send);
  }
}

void encode_block(Block &b, const uint8_t *data) {  // the second half of _tr_flush_block
  const StaticTables &T = tables();
  BitWriter w;
  w.out.reserve(static_cast<size_t>(b.bit_len / 8 + 16));
  const int lead = static_cast<int>(b.bit_start & 7);
  w.nbits = lead;  // leading zero bits: the block is OR-ed into place
  write_block_header(b, w);
  if (b.choice == STORED) {
    w.out.insert(w.out.end(), data + b.start, data + b.start + b.bytes);
    w.total += 8 * b.bytes;
  } else {
    const uint16_t *llen, *lcode, *dcode;
    uint16_t dlen_static[D_CODES];
    const uint16_t *dlen;
    if (b.choice == STATIC) {
      llen = T.sl_len;
      lcode = T.sl_code;
      for (int i = 0; i < D_CODES; ++i) dlen_static[i] = 5;
      dlen = dlen_static;
      dcode = T.sd_code;
    } else {
      llen = b.lt.len;
      lcode = b.lt.code;
      dlen = b.dt.len;
      dcode = b.dt.code;
    }
    for (const Seg &s : b.segs) {  // compress_block
      for (size_t i = 0; i < s.n; ++i) {
        unsigned dist = s.dist[i], lc = s.lc[i];
        if (dist == 0) {
          w.put(lcode[lc], llen[lc]);
        } else {
          unsigned code = T.length_code[lc];
          w.put(lcode[code + LITERALS + 1], llen[code + LITERALS + 1]);
          int extra = kExtraL[code];
          if (extra) w.put(lc - static_cast<unsigned>(T.base_length[code]), extra);
          dist--;
          code = static_cast<unsigned>(T.d_code(dist));
          w.put(dcode[code], dlen[code]);
          extra = kExtraD[code];
          if (extra) w.put(dist - static_cast<unsigned>(T.base_dist[code]), extra);
        }
      }
    }
    w.put(lcode[END_BLOCK], llen[END_BLOCK]);
  }
  if (b.last) w.align();
  if (w.nbits > 0) w.out.push_back(static_cast<uint8_t>(w.acc));  // partial last byte, completed by the next block
  b.bits.swap(w.out);
  b.bit_len = w.total;  // compress() compares this with the planned length
}

template <class F>
void parallel_for(size_t n, int threads, F fn) {
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= n) return;
      fn(i);
    }
  };
  const int nt = static_cast<int>(std::min<size_t>(static_cast<size_t>(std::max(threads, 1)), n));
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(worker);
  worker();
  for (auto &t : pool) t.join();
}

// ---- self-check against the linked zlib: deflate the first `verify` input bytes with zlib itself and
// compare every output byte that cannot depend on where that prefix ends (whole blocks that end at least
// 1 KiB before it).  Guards against a zlib whose algorithm differs from the one restated here.
// zlib's own member for the first `verify` input bytes (empty on failure): the half of the check that needs nothing
// from the writer, so it can run beside it.
std::vector<uint8_t> zlib_prefix_member(const uint8_t *data, uint64_t verify) {
  std::vector<uint8_t> z(compressBound(static_cast<uLong>(verify)) + 64);
  z_stream zs = {};
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 16 + MAX_WBITS, 9, Z_DEFAULT_STRATEGY) != Z_OK) return {};
  zs.next_in = const_cast<Bytef *>(data);
  zs.avail_in = static_cast<uInt>(verify);
  zs.next_out = z.data();
  zs.avail_out = static_cast<uInt>(z.size());
  const int rc = deflate(&zs, Z_FINISH);
  z.resize(rc == Z_STREAM_END ? zs.total_out : 0);
  deflateEnd(&zs);
  return z;
}

template <class B>
bool matches_zlib_prefix(const std::vector<uint8_t> &z, size_t size, const std::vector<B> &blocks,
                         const std::vector<uint8_t> &member, size_t verify_prefix) {
  const uint64_t verify = std::min<uint64_t>(size, verify_prefix);
  uint64_t safe_bits = 0;
  for (const B &b : blocks) {
    const uint64_t end_pos = b.start + b.bytes;
    if (verify == size || end_pos + 1024 <= verify) safe_bits = b.bit_start + b.bit_len;
    else break;
  }
  const size_t safe_bytes = 10 + static_cast<size_t>(safe_bits / 8) + (verify == size ? 8 : 0);
  return !z.empty() && z.size() >= safe_bytes && member.size() >= safe_bytes && std::memcmp(z.data(), member.data(), safe_bytes) == 0;
}

template <class B>
bool verify_against_zlib(const uint8_t *data, size_t size, const std::vector<B> &blocks, const std::vector<uint8_t> &member,
                         size_t verify_prefix) {
  return matches_zlib_prefix(zlib_prefix_member(data, std::min<uint64_t>(size, verify_prefix)), size, blocks, member, verify_prefix);
}

// ---- serial: positions, stored / static / dynamic, bit offsets (the rest of _tr_flush_block's decision)
// Blocks [first, end) given where the ones before them ended (*pos_io input bytes, *bit_io output bits).
template <class B>
bool layout_block_range(std::vector<B> &blocks, size_t first, size_t end, size_t size, uint64_t *pos_io, uint64_t *bit_io) {
  uint64_t pos = *pos_io, bit = *bit_io;
  for (size_t bi = first; bi < end; ++bi) {
    B &b = blocks[bi];
    b.start = pos;
    pos += b.bytes;
    // loop top of the iteration that tallied the block's last symbol (Z_FINISH flush: the end of the input)
    const uint64_t s_flush = b.last ? size : (pos - b.last_sym_len + 1);
    const bool buf_in_window = b.start >= base_at(s_flush, size);
    int64_t opt_lenb = (b.opt_len + 3 + 7) >> 3;
    const int64_t static_lenb = (b.static_len + 3 + 7) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
    b.bit_start = bit;
    if (static_cast<int64_t>(b.bytes) + 4 <= opt_lenb && buf_in_window) {
      if (b.bytes > 0xffff) return false;
      b.choice = STORED;
      const uint64_t after_type = bit + 3;
      b.bit_len = 3 + ((8 - (after_type & 7)) & 7) + 32 + 8 * b.bytes;
    } else if (static_lenb == opt_lenb) {
      b.choice = STATIC;
      b.bit_len = 3 + static_cast<uint64_t>(b.static_len);
    } else {
      b.choice = DYNAMIC;
      b.bit_len = 3 + static_cast<uint64_t>(b.opt_len);
    }
    if (b.last) b.bit_len += (8 - ((bit + b.bit_len) & 7)) & 7;  // bi_windup
    bit += b.bit_len;
  }
  *pos_io = pos;
  *bit_io = bit;
  return true;
}

template <class B>
bool layout_blocks(std::vector<B> &blocks, size_t size, uint64_t *total_bits) {
  uint64_t pos = 0, bit = 0;
  if (!layout_block_range(blocks, 0, blocks.size(), size, &pos, &bit) || pos != size) return false;
  *total_bits = bit;
  return true;
}

// ---- from the spliced symbol stream to the gzip member ------------------------------------------------
// `parts`: the symbol stream in order, in pieces; `tail_literal`: its last symbol is the pending literal zlib
// tallies after the loop; `crc`: CRC-32 of the input.
template <class Lap>
bool finish_member(const uint8_t *data, size_t size, int threads, const std::vector<Seg> &parts, bool tail_literal,
                   uLong crc, std::vector<uint8_t> *out, size_t verify_prefix, Lap &lap) {
  const size_t nparts = parts.size();
  std::vector<uint64_t> goff(nparts + 1, 0);  // global symbol index of each part's first symbol
  for (size_t i = 0; i < nparts; ++i) goff[i + 1] = goff[i] + parts[i].n;
  const uint64_t total_syms = goff[nparts];

  // ---- blocks: a block is flushed when the symbol buffer fills (BLOCK_SYMS) and at Z_FINISH, where the
  // final (possibly empty) block carries the last-block flag.  The literal that is still pending when the
  // input ends is tallied without a buffer check, so if it is the one that fills the buffer, that full block
  // is the final one.
  size_t nblocks = static_cast<size_t>(total_syms / BLOCK_SYMS) + 1;
  if (total_syms > 0 && total_syms % BLOCK_SYMS == 0 && tail_literal) nblocks -= 1;
  std::vector<Block> blocks(nblocks);
  parallel_for(nblocks, threads, [&](size_t bi) {
    Block &b = blocks[bi];
    const uint64_t g0 = static_cast<uint64_t>(bi) * BLOCK_SYMS, g1 = std::min<uint64_t>(total_syms, g0 + BLOCK_SYMS);
    b.nsyms = static_cast<size_t>(g1 - g0);
    b.last = (bi + 1 == nblocks);
    size_t part = static_cast<size_t>(std::upper_bound(goff.begin(), goff.end(), g0) - goff.begin());
    part = part > 0 ? part - 1 : 0;
    uint64_t g = g0;
    while (g < g1) {
      while (goff[part + 1] <= g) ++part;
      const uint64_t take = std::min<uint64_t>(g1, goff[part + 1]) - g;
      const size_t off = static_cast<size_t>(g - goff[part]);
      b.segs.push_back({parts[part].dist + off, parts[part].lc + off, static_cast<size_t>(take)});
      g += take;
    }
    plan_block(b);
  });
  lap("plan");

  uint64_t bit = 0;
  if (!layout_blocks(blocks, size, &bit)) return false;
  std::atomic<bool> bad{false};
  parallel_for(nblocks, threads, [&](size_t bi) {
    const uint64_t planned = blocks[bi].bit_len;
    encode_block(blocks[bi], data);
    if (blocks[bi].bit_len != planned) bad = true;
  });
  if (bad) return false;
  lap("encode");

  // ---- assemble: 10-byte header (deflate.c: no flags, mtime 0, xfl 0, OS_CODE 3), bit strings, CRC-32, ISIZE
  const uint64_t deflate_bytes = bit / 8;
  out->clear();
  detail::resizeUninitialized(out, static_cast<size_t>(10 + deflate_bytes + 8));
  const uint8_t header[10] = {0x1f, 0x8b, 0x08, 0x00, 0, 0, 0, 0, 0x00, 0x03};
  std::memcpy(out->data(), header, 10);
  uint8_t *body = out->data() + 10;
  // a block's first and last byte may be shared with its neighbours (it is OR-ed into place): those are zeroed
  // and OR-ed serially, the bytes in between are copied by all threads
  for (const Block &b : blocks) {
    if (b.bits.empty()) continue;
    const size_t at = static_cast<size_t>(b.bit_start / 8);
    body[at] = 0;
    body[at + b.bits.size() - 1] = 0;
  }
  parallel_for(nblocks, threads, [&](size_t bi) {
    const Block &b = blocks[bi];
    if (b.bits.size() > 2) std::memcpy(body + b.bit_start / 8 + 1, b.bits.data() + 1, b.bits.size() - 2);
  });
  for (const Block &b : blocks) {
    if (b.bits.empty()) continue;
    const size_t at = static_cast<size_t>(b.bit_start / 8);
    body[at] |= b.bits[0];
    if (b.bits.size() > 1) body[at + b.bits.size() - 1] |= b.bits.back();
  }
  uint8_t *trailer = body + deflate_bytes;
  for (int k = 0; k < 4; ++k) trailer[k] = static_cast<uint8_t>(crc >> (8 * k));
  for (int k = 0; k < 4; ++k) trailer[4 + k] = static_cast<uint8_t>((size & 0xffffffffu) >> (8 * k));
  lap("assemble");

  if (verify_prefix > 0) {
    if (!verify_against_zlib(data, size, blocks, *out, verify_prefix)) return false;
    lap("verify");
  }
  return true;
}

// ---- the same, with the symbols on the parser's side (the device): counts come back, trees and block layout are
// made here, codes and headers go out, the finished deflate body comes back.
template <class Lap>
bool finish_member_on_parser(HeadParser &parser, const uint8_t *data, size_t size, int threads, uint64_t total_syms,
                             bool tail_literal, uLong crc, std::vector<uint8_t> *out, size_t verify_prefix,
                             const std::vector<uint8_t> *zlib_prefix, Lap &lap) {
  const StaticTables &T = tables();
  spz_amd_deflate_static st = {};
  std::memcpy(st.length_code, T.length_code, sizeof(st.length_code));
  std::memcpy(st.dist_code, T.dist_code, sizeof(st.dist_code));
  for (int i = 0; i < 29; ++i) {
    st.base_length[i] = static_cast<uint16_t>(T.base_length[i]);
    st.extra_lbits[i] = static_cast<uint8_t>(kExtraL[i]);
  }
  for (int i = 0; i < 30; ++i) {
    st.base_dist[i] = static_cast<uint16_t>(T.base_dist[i]);
    st.extra_dbits[i] = static_cast<uint8_t>(kExtraD[i]);
  }
  size_t nblocks = static_cast<size_t>(total_syms / BLOCK_SYMS) + 1;
  if (total_syms > 0 && total_syms % BLOCK_SYMS == 0 && tail_literal) nblocks -= 1;
  const bool device_trees = parser.canBuildTrees();
  std::vector<uint16_t> lfreq(device_trees ? 0 : nblocks * 286), dfreq(device_trees ? 0 : nblocks * 30);
  std::vector<uint32_t> bytes(nblocks), last_len(nblocks);
  if (!parser.blockStats(st, static_cast<uint32_t>(BLOCK_SYMS), static_cast<uint32_t>(nblocks), device_trees ? nullptr : lfreq.data(),
                         device_trees ? nullptr : dfreq.data(), bytes.data(), last_len.data())) {
    return false;
  }
  lap("stats");
  auto trailer_and_check = [&](const auto &blocks, uint64_t deflate_bytes) {
    uint8_t *trailer = out->data() + 10 + deflate_bytes;
    for (int k2 = 0; k2 < 4; ++k2) trailer[k2] = static_cast<uint8_t>(crc >> (8 * k2));
    for (int k2 = 0; k2 < 4; ++k2) trailer[4 + k2] = static_cast<uint8_t>((size & 0xffffffffu) >> (8 * k2));
    if (verify_prefix == 0) return true;
    return zlib_prefix ? matches_zlib_prefix(*zlib_prefix, size, blocks, *out, verify_prefix)
                       : verify_against_zlib(data, size, blocks, *out, verify_prefix);
  };
  if (device_trees) {  // the trees are built where the counts are; the layout of the blocks is all that is left here
    std::vector<spz_amd_deflate_plan> plan(nblocks);
    if (!parser.blockTrees(static_cast<uint32_t>(nblocks), plan.data())) return false;
    lap("trees");
    std::vector<BlockLayout> blocks(nblocks);
    for (size_t bi = 0; bi < nblocks; ++bi) {
      BlockLayout &b = blocks[bi];
      const uint64_t g0 = static_cast<uint64_t>(bi) * BLOCK_SYMS, g1 = std::min<uint64_t>(total_syms, g0 + BLOCK_SYMS);
      b.nsyms = static_cast<size_t>(g1 - g0);
      b.last = (bi + 1 == nblocks);
      b.bytes = bytes[bi];
      b.last_sym_len = last_len[bi];
      b.opt_len = plan[bi].opt_len;
      b.static_len = plan[bi].static_len;
    }
    uint64_t bit = 0;
    if (!layout_blocks(blocks, size, &bit)) return false;
    std::vector<spz_amd_deflate_block> desc(nblocks);
    for (size_t bi = 0; bi < nblocks; ++bi) {
      desc[bi] = {};
      desc[bi].bit_start = blocks[bi].bit_start;
      desc[bi].choice = static_cast<uint32_t>(blocks[bi].choice);
      desc[bi].input_begin = static_cast<uint32_t>(blocks[bi].start);
      desc[bi].input_bytes = static_cast<uint32_t>(blocks[bi].bytes);
    }
    const uint64_t deflate_bytes = bit / 8;
    if (!parser.encodePlanned(st, static_cast<uint32_t>(BLOCK_SYMS), static_cast<uint32_t>(nblocks), desc.data(), deflate_bytes)) {
      return false;
    }
    // while the device writes headers and symbols: the output buffer, its pages mapped.  (Mapping a buffer of a guessed
    // size from the start of the writer, beside the device's parse, takes these 3 - 4 ms away and costs 10: the small
    // copies of the stages in between wait for the mapping threads in the kernel's memory management.  Round 3.)
    out->clear();
    detail::resizeUninitialized(out, static_cast<size_t>(10 + deflate_bytes + 8));
    {
      detail::Prefault pf;
      pf.add(out->data(), out->size());
      pf.start();
      pf.join();
    }
    const uint8_t header[10] = {0x1f, 0x8b, 0x08, 0x00, 0, 0, 0, 0, 0x00, 0x03};
    std::memcpy(out->data(), header, 10);
    lap("layout");
    std::vector<uint64_t> symbol_bits(nblocks);
    std::vector<uint32_t> header_bits(nblocks);
    if (!parser.encodeFinish(static_cast<uint32_t>(nblocks), deflate_bytes, out->data() + 10, symbol_bits.data(), header_bits.data())) {
      return false;
    }
    lap("encode");
    for (size_t bi = 0; bi < nblocks; ++bi) {  // what the device wrote against what was planned
      const BlockLayout &b = blocks[bi];
      uint64_t planned = b.bit_len - header_bits[bi];
      if (b.last) planned -= (8 - ((b.bit_start + header_bits[bi] + symbol_bits[bi]) & 7)) & 7;
      if (header_bits[bi] > b.bit_len || symbol_bits[bi] != planned) return false;
    }
    if (!trailer_and_check(blocks, deflate_bytes)) return false;
    lap("verify");
    return true;
  }
  // Trees and codes are made group by group: while the host builds the next group's trees the device packs the bits
  // of the one before.
  std::vector<Block> blocks(nblocks);
  std::vector<spz_amd_deflate_block> desc(nblocks);
  std::vector<spz_amd_deflate_codes> codes(nblocks);
  const uint64_t body_bound = static_cast<uint64_t>(size) + size / 8 + static_cast<uint64_t>(nblocks) * 512 + 4096;
  const size_t group = std::max<size_t>(64, (nblocks + 5) / 6);
  uint64_t pos = 0, bit = 0;
  std::vector<std::vector<uint32_t>> group_words;  // kept until encodeFinish: the copies out of them are only enqueued
  group_words.reserve(nblocks / group + 1);
  for (size_t g0b = 0; g0b < nblocks; g0b += group) {
    const size_t gn = std::min(group, nblocks - g0b);
    parallel_for(gn, threads, [&](size_t k) {
      const size_t bi = g0b + k;
      Block &b = blocks[bi];
      const uint64_t g0 = static_cast<uint64_t>(bi) * BLOCK_SYMS, g1 = std::min<uint64_t>(total_syms, g0 + BLOCK_SYMS);
      b.nsyms = static_cast<size_t>(g1 - g0);
      b.last = (bi + 1 == nblocks);
      for (int i = 0; i < 286; ++i) b.lt.freq[i] = lfreq[bi * 286 + i];
      for (int i = 0; i < 30; ++i) b.dt.freq[i] = dfreq[bi * 30 + i];
      b.bytes = bytes[bi];
      b.last_sym_len = last_len[bi];
      plan_trees(b);
    });
    if (!layout_block_range(blocks, g0b, g0b + gn, size, &pos, &bit)) return false;
    if (bit / 8 + 8 > body_bound) return false;
    // codes and headers: a block's header is written on the 32-bit grid of the body, starting at the word its first bit is in
    std::vector<std::vector<uint8_t>> hdr(gn);
    parallel_for(gn, threads, [&](size_t k) {
      const size_t bi = g0b + k;
      const Block &b = blocks[bi];
      BitWriter w;
      const unsigned lead = static_cast<unsigned>(b.bit_start & 31);
      for (unsigned i = 0; i < lead / 8; ++i) w.out.push_back(0);
      w.nbits = static_cast<int>(lead & 7);
      write_block_header(b, w);
      spz_amd_deflate_block &d = desc[bi];
      d.bit_start = b.bit_start;
      d.header_bits = static_cast<uint32_t>(w.total);
      if (w.nbits > 0) w.out.push_back(static_cast<uint8_t>(w.acc));
      while (w.out.size() % 4 != 0) w.out.push_back(0);
      d.header_words = static_cast<uint32_t>(w.out.size() / 4);
      d.choice = static_cast<uint32_t>(b.choice);
      d.input_begin = static_cast<uint32_t>(b.start);
      d.input_bytes = static_cast<uint32_t>(b.bytes);
      hdr[k].swap(w.out);
      spz_amd_deflate_codes &c = codes[bi];
      if (b.choice == STATIC) {
        for (int i = 0; i < 286; ++i) {
          c.lcode[i] = T.sl_code[i];
          c.llen[i] = static_cast<uint8_t>(T.sl_len[i]);
        }
        for (int i = 0; i < 30; ++i) {
          c.dcode[i] = T.sd_code[i];
          c.dlen[i] = 5;
        }
      } else {
        for (int i = 0; i < 286; ++i) {
          c.lcode[i] = b.lt.code[i];
          c.llen[i] = static_cast<uint8_t>(b.lt.len[i]);
        }
        for (int i = 0; i < 30; ++i) {
          c.dcode[i] = b.dt.code[i];
          c.dlen[i] = static_cast<uint8_t>(b.dt.len[i]);
        }
      }
    });
    uint64_t nwords = 0;  // counted from the group's first header word
    for (size_t k = 0; k < gn; ++k) {
      desc[g0b + k].header_word_begin = static_cast<uint32_t>(nwords);
      nwords += desc[g0b + k].header_words;
    }
    if (nwords >= (uint64_t(1) << 32)) return false;
    group_words.emplace_back(static_cast<size_t>(nwords) + 1);
    std::vector<uint32_t> &words = group_words.back();
    parallel_for(gn, threads, [&](size_t k) {
      if (!hdr[k].empty()) std::memcpy(&words[desc[g0b + k].header_word_begin], hdr[k].data(), hdr[k].size());
    });
    if (!parser.encodeGroup(st, static_cast<uint32_t>(BLOCK_SYMS), static_cast<uint32_t>(nblocks), static_cast<uint32_t>(g0b),
                            static_cast<uint32_t>(gn), &desc[g0b], &codes[g0b], words.data(), nwords, body_bound)) {
      return false;
    }
  }
  if (pos != size) return false;
  lap("trees+codes");
  const uint64_t deflate_bytes = bit / 8;
  out->clear();
  detail::resizeUninitialized(out, static_cast<size_t>(10 + deflate_bytes + 8));
  {
    detail::Prefault pf;
    pf.add(out->data(), out->size());
    pf.start();
    pf.join();
  }
  const uint8_t header[10] = {0x1f, 0x8b, 0x08, 0x00, 0, 0, 0, 0, 0x00, 0x03};
  std::memcpy(out->data(), header, 10);
  std::vector<uint64_t> symbol_bits(nblocks);
  if (!parser.encodeFinish(static_cast<uint32_t>(nblocks), deflate_bytes, out->data() + 10, symbol_bits.data(), nullptr)) return false;
  lap("encode");
  for (size_t bi = 0; bi < nblocks; ++bi) {  // what the encoder wrote against what was planned
    const Block &b = blocks[bi];
    uint64_t planned = b.bit_len - desc[bi].header_bits;
    if (b.last) planned -= (8 - ((b.bit_start + desc[bi].header_bits + symbol_bits[bi]) & 7)) & 7;
    if (symbol_bits[bi] != planned) return false;
  }
  uint8_t *trailer = out->data() + 10 + deflate_bytes;
  for (int k2 = 0; k2 < 4; ++k2) trailer[k2] = static_cast<uint8_t>(crc >> (8 * k2));
  for (int k2 = 0; k2 < 4; ++k2) trailer[4 + k2] = static_cast<uint8_t>((size & 0xffffffffu) >> (8 * k2));
  if (verify_prefix > 0) {
    const bool same = zlib_prefix ? matches_zlib_prefix(*zlib_prefix, size, blocks, *out, verify_prefix)
                                  : verify_against_zlib(data, size, blocks, *out, verify_prefix);
    if (!same) return false;
  }
  lap("verify");
  return true;
}

}  // namespace

bool compress(const uint8_t *data, size_t size, int threads, int windows_per_chunk, std::vector<uint8_t> *out,
              size_t verify_prefix) {
  if (data == nullptr || out == nullptr || threads < 1 || windows_per_chunk < 4) return false;
  const uint64_t C = static_cast<uint64_t>(windows_per_chunk) * W;
  if (size < 4 * W || size >= (uint64_t(1) << 32) - 2 * W) return false;
  const size_t njobs = std::max<size_t>(1, static_cast<size_t>(size / C));  // the last job takes the remainder too

  // The last job reads up to 258 + 1 bytes past the end; zlib's window holds there what was 32 KiB earlier.
  const uint64_t last_begin = (njobs - 1) * C;
  const uint64_t tail_lo = last_begin == 0 ? 0 : last_begin - W;
  std::vector<uint8_t> tail(static_cast<size_t>(size - tail_lo) + MAX_MATCH + 8);
  std::memcpy(tail.data(), data + tail_lo, static_cast<size_t>(size - tail_lo));
  for (uint64_t abs = size; abs < size + MAX_MATCH + 8; ++abs) tail[static_cast<size_t>(abs - tail_lo)] = data[abs - W];

  static const bool check_phase = [] {
    const char *e = std::getenv("SPZ_AMD_EXACT_GZIP_CHECK");
    return e && e[0] == '1';
  }();
  static const bool timing = std::getenv("SPZ_AMD_EXACT_GZIP_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[exactgz] %-10s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  std::vector<Job> jobs(njobs);
  for (size_t i = 0; i < njobs; ++i) {
    Job &j = jobs[i];
    j.size = size;
    j.begin = i * C;
    j.last = (i + 1 == njobs);
    j.stop = j.last ? size : (i + 1) * C + OVERLAP;
    j.d = j.last ? tail.data() - tail_lo : data;
    j.check_phase = check_phase;
    if (i > 0) {
      j.rec_succ_lo = j.begin;
      j.rec_succ.assign(OVERLAP + MAX_MATCH + 2, TopRec{});
    }
    if (!j.last) {
      j.rec_pred_lo = (i + 1) * C;
      j.rec_pred.assign(OVERLAP + MAX_MATCH + 2, TopRec{});
    }
  }
  // CRC-32 of the input rides along with the parse jobs (folded with crc32_combine afterwards).
  std::vector<uLong> crcs(njobs);
  parallel_for(njobs, threads, [&](size_t i) {
    jobs[i].run();
    const uint64_t lo = jobs[i].begin, hi = jobs[i].last ? size : (i + 1) * C;
    uLong c = crc32(0L, Z_NULL, 0);
    for (uint64_t p = lo; p < hi;) {
      const uint64_t n = std::min<uint64_t>(hi - p, uint64_t(1) << 30);
      c = crc32(c, data + p, static_cast<uInt>(n));
      p += n;
    }
    crcs[i] = c;
  });
  lap("parse+crc");
  for (const Job &j : jobs) {
    if (!j.phase_ok) return false;
  }

  // ---- splice: job i contributes symbols [lo[i], hi[i])
  std::vector<size_t> lo(njobs, 0), hi(njobs, 0);
  for (size_t i = 0; i < njobs; ++i) hi[i] = jobs[i].sym_lc.size();
  for (size_t i = 0; i + 1 < njobs; ++i) {
    const Job &a = jobs[i], &b = jobs[i + 1];
    bool found = false;
    for (size_t k = 0; k < a.rec_pred.size() && k < b.rec_succ.size(); ++k) {
      if (a.rec_pred[k].state != 0 && a.rec_pred[k].state == b.rec_succ[k].state) {
        hi[i] = a.rec_pred[k].symcount;
        lo[i + 1] = b.rec_succ[k].symcount;
        found = true;
        break;
      }
    }
    if (!found || hi[i] < lo[i]) return false;
  }
  std::vector<Seg> parts(njobs);
  for (size_t i = 0; i < njobs; ++i) parts[i] = {jobs[i].sym_dist.data() + lo[i], jobs[i].sym_lc.data() + lo[i], hi[i] - lo[i]};
  lap("splice");
  uLong crc = crcs[0];
  for (size_t i = 1; i < njobs; ++i) {
    const uint64_t n = (jobs[i].last ? size : (i + 1) * C) - jobs[i].begin;
    crc = crc32_combine(crc, crcs[i], static_cast<z_off_t>(n));
  }
  return finish_member(data, size, threads, parts, jobs[njobs - 1].tail_literal, crc, out, verify_prefix, lap);
}

// the serial tail job: the last 64 ... 96 KiB, with the reads past the end of the input as zlib's window has them
struct TailAhead {
  std::vector<uint8_t> tail;
  Job job;
  std::thread thread;
  uint64_t size = 0;
  std::chrono::steady_clock::time_point asked = std::chrono::steady_clock::now();
  void prepare(const uint8_t *data, uint64_t size_) {
    size = size_;
    const uint64_t tail_begin = (size - 2 * W) / W * W;
    const uint64_t tail_lo = tail_begin - W;
    tail.resize(static_cast<size_t>(size - tail_lo) + MAX_MATCH + 8);
    std::memcpy(tail.data(), data + tail_lo, static_cast<size_t>(size - tail_lo));
    for (uint64_t abs = size; abs < size + MAX_MATCH + 8; ++abs) tail[static_cast<size_t>(abs - tail_lo)] = data[abs - W];
    job.size = size;
    job.begin = tail_begin;
    job.last = true;
    job.stop = size;
    job.d = tail.data() - tail_lo;
    job.rec_succ_lo = tail_begin;
    job.rec_succ.assign(OVERLAP + MAX_MATCH + 2, TopRec{});
  }
  void finish() {  // the job has run when this returns
    if (thread.joinable()) thread.join();
    else if (!ran) job.run();
    ran = true;
  }
  bool ran = false;
};

static bool sizeInRange(uint64_t size) { return size >= 16 * W && size < (uint64_t(1) << 32) - 2 * W; }

size_t tailAheadBytes(size_t size) {
  if (!sizeInRange(size)) return 0;
  const uint64_t tail_lo = (size - 2 * W) / W * W - W;
  return static_cast<size_t>(size - tail_lo);
}

TailAhead *tailAheadStart(const uint8_t *data, size_t size) {
  if (data == nullptr || !sizeInRange(size)) return nullptr;
  TailAhead *a = new (std::nothrow) TailAhead();
  if (a == nullptr) return nullptr;
  try {
    a->prepare(data, size);
  } catch (const std::bad_alloc &) {  // (called from a C callback: nothing may leave it)
    delete a;
    return nullptr;
  }
  try {
    a->thread = std::thread([a]() {
      const auto t0 = std::chrono::steady_clock::now();
      a->job.run();
      a->ran = true;
      if (std::getenv("SPZ_AMD_EXACT_GZIP_TIMING")) {
        std::fprintf(stderr, "[exactgz] tail job, started ahead: %.4f s on its thread, which began %.4f s after it was asked for\n",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(),
                     std::chrono::duration<double>(t0 - a->asked).count());
      }
    });
  } catch (const std::system_error &) {
    // no thread to be had: the job runs when the writer asks for it
  }
  return a;
}

void tailAheadDrop(TailAhead *ahead) {
  if (ahead == nullptr) return;
  if (ahead->thread.joinable()) ahead->thread.join();
  delete ahead;
}

bool compressWithHeadParser(const uint8_t *data, size_t size, int threads, HeadParser &parser,
                            std::vector<uint8_t> *out, size_t verify_prefix, TailAhead *ahead) {
  static_assert(sizeof(TopRec) == 8, "records travel as {state, symcount} pairs");
  struct DropAhead {
    TailAhead *a;
    ~DropAhead() { tailAheadDrop(a); }
  } drop_ahead{ahead};
  if (data == nullptr || out == nullptr || threads < 1) return false;
  if (!sizeInRange(size)) return false;
  if (ahead != nullptr && ahead->size != size) return false;
  static const bool timing = std::getenv("SPZ_AMD_EXACT_GZIP_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[exactgz] %-10s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  // CRC-32 of the input on the host's threads while the parse runs elsewhere
  const size_t ncrc = static_cast<size_t>((size + (size_t(4) << 20) - 1) / (size_t(4) << 20));
  std::vector<uLong> crcs(ncrc);
  std::thread crc_thread([&]() {
    parallel_for(ncrc, threads, [&](size_t i) {
      const size_t lo = i * (size_t(4) << 20), hi = std::min(size, lo + (size_t(4) << 20));
      crcs[i] = crc32(crc32(0L, Z_NULL, 0), data + lo, static_cast<uInt>(hi - lo));
    });
  });
  struct Joiner {
    std::thread &t;
    ~Joiner() {
      if (t.joinable()) t.join();
    }
  } joiner{crc_thread};
  // zlib's own member of the input's first bytes, for the writer's standing self-check: computed meanwhile as well
  std::vector<uint8_t> zprefix;
  std::thread zlib_thread;
  if (verify_prefix > 0) {
    try {
      zlib_thread = std::thread([&]() { zprefix = zlib_prefix_member(data, std::min<uint64_t>(size, verify_prefix)); });
    } catch (const std::system_error &) {
      // no thread to be had: computed at the end
    }
  }
  Joiner zlib_joiner{zlib_thread};

  const uint64_t tail_begin = (size - 2 * W) / W * W;
  TailAhead own_tail;
  TailAhead &ta = ahead ? *ahead : own_tail;
  if (!ahead) own_tail.prepare(data, size);
  Job &tj = ta.job;
  // the tail job runs when the parser says its records can wait no longer: beside the device's first kernels (or has
  // been running since the caller had the input's last bytes)
  uint64_t nhead = 0;
  uint32_t tail_first = 0;
  const uint32_t n_rec = static_cast<uint32_t>(std::min<size_t>(tj.rec_succ.size(), spz_lz::kTailWindow));
  auto run_tail = [](void *p) { static_cast<TailAhead *>(p)->finish(); };
  if (!parser.parseLate(data, size, tail_begin, reinterpret_cast<const uint32_t *>(tj.rec_succ.data()), n_rec, run_tail, &ta,
                        &nhead, &tail_first)) {
    ta.finish();  // (nothing of it may still run when it goes)
    return false;
  }
  if (!tj.phase_ok) return false;
  lap("head parse");
  if (tail_first > tj.sym_lc.size() || nhead > size) return false;
  if (parser.canFinish()) {
    const size_t ntail = tj.sym_lc.size() - tail_first;
    if (!parser.append(tj.sym_dist.data() + tail_first, tj.sym_lc.data() + tail_first, ntail)) return false;
    crc_thread.join();
    uLong crc = crcs[0];
    for (size_t i = 1; i < ncrc; ++i) {
      const size_t lo = i * (size_t(4) << 20), hi = std::min(size, lo + (size_t(4) << 20));
      crc = crc32_combine(crc, crcs[i], static_cast<z_off_t>(hi - lo));
    }
    const bool have_prefix = zlib_thread.joinable();
    if (have_prefix) zlib_thread.join();
    return finish_member_on_parser(parser, data, size, threads, nhead + ntail, tj.tail_literal, crc, out, verify_prefix,
                                   have_prefix ? &zprefix : nullptr, lap);
  }
  std::vector<uint16_t> hd;
  std::vector<uint8_t> hl;
  detail::resizeUninitialized(&hd, static_cast<size_t>(nhead));
  detail::resizeUninitialized(&hl, static_cast<size_t>(nhead));
  {
    detail::Prefault pf;
    pf.add(hd.data(), hd.size() * sizeof(uint16_t));
    pf.add(hl.data(), hl.size());
    pf.start();
    pf.join();
  }
  if (!parser.fetch(hd.data(), hl.data())) return false;
  lap("fetch");
  std::vector<Seg> parts(2);
  parts[0] = {hd.data(), hl.data(), static_cast<size_t>(nhead)};
  parts[1] = {tj.sym_dist.data() + tail_first, tj.sym_lc.data() + tail_first, tj.sym_lc.size() - tail_first};
  crc_thread.join();
  uLong crc = crcs[0];
  for (size_t i = 1; i < ncrc; ++i) {
    const size_t lo = i * (size_t(4) << 20), hi = std::min(size, lo + (size_t(4) << 20));
    crc = crc32_combine(crc, crcs[i], static_cast<z_off_t>(hi - lo));
  }
  lap("crc join");
  const bool ok = finish_member(data, size, threads, parts, tj.tail_literal, crc, out, verify_prefix, lap);
  lap("blocks freed");
  return ok;
}

}  // namespace exactgz
}  // namespace spz
