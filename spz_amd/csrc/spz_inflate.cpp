// spz_inflate.cpp — multi-threaded inflate of a single, ordinary deflate stream.
//
// Why: once the dequantise step runs on the GPU, gunzip is ~80 % of a loadSpz, and the
// files the reference writes are one zlib deflate stream with no index.  A deflate stream can still be
// decoded in parallel (the idea of pugz, Kerbiriou & Chikhi 2019), and here every result is verified
// against the member's CRC-32, so the fast path can only ever be faster, not different:
//   1. cut the compressed bytes into chunks; in each, search bit by bit for the start of a dynamic-Huffman
//      block (header fields in range, both code-length sets complete, the block decodes to its end-of-block
//      and a plausible header follows);
//   2. decode, parallel, ONCE: every chunk from its block start up to the next chunk's block start (they must
//      link up bit-exactly) WITHOUT its 32 KiB of left context.  The sliding window is kept in 16-bit symbols,
//      where 256 + k stands for "byte k of the predecessor's final window"; the bytes go to a buffer of the
//      chunk's own, and every position whose symbol is still such a reference goes to a patch list;
//   3. serial, tiny: resolve each chunk's final window against its predecessor's, prefix-sum the lengths;
//   4. place, parallel: copy every chunk's bytes to their final offset and apply its patches from the resolved
//      context.  (A chunk whose patch list outgrows a quarter of its length — data that keeps copying from the
//      unknown context — is decoded a second time with its context instead, as every chunk was before.)
//   5. CRC-32 (parallel, crc32_combine) and ISIZE must match the gzip trailer.
// Any irregularity returns false and the caller's serial reader decides.
#include "spz_inflate.hpp"
#include "spz_host_util.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <system_error>
#include <thread>

namespace spz {
namespace pinflate {
namespace {

constexpr uint32_t W = 32768;
constexpr int FAST_L = 11, FAST_D = 9;
constexpr uint64_t NONE = ~uint64_t(0);

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Bits {  // the deflate data of the member
  const uint8_t *p;
  uint64_t nbits;
  size_t nbytes;
  // >= 56 valid bits starting at bit position `at` (zeros past the end)
  inline uint64_t peek(uint64_t at) const {
    const size_t b = static_cast<size_t>(at >> 3);
    uint64_t v = 0;
    if (b + 8 <= nbytes) {
      std::memcpy(&v, p + b, 8);
    } else {
      for (size_t k = 0; b + k < nbytes && k < 8; ++k) v |= static_cast<uint64_t>(p[b + k]) << (8 * k);
    }
    return v >> (at & 7);
  }
};

// Canonical Huffman decoder: a direct table for codes up to `fast` bits, canonical walk for longer ones.
// Packed decode entry of a literal/length or distance code: everything the inner loop needs in one 32-bit load.
//   bits 0-3  code length (0: the code is longer than the fast table, take the canonical walk)
//   bits 4-7  number of extra bits (length / distance codes)
//   bit  8    literal, bit 9 end of block, bit 10 invalid symbol (286, 287 / distance 30, 31)
//   bits 16-31 the literal byte, or the base length / base distance
constexpr uint32_t ENT_LITERAL = 1u << 8, ENT_EOB = 1u << 9, ENT_INVALID = 1u << 10;

struct Huff {
  uint16_t fast[1 << FAST_L];
  uint32_t packed[1 << FAST_L];  // filled by pack(): same index as fast[]
  uint32_t ent[288];             // per symbol, without the code length (slow path)
  uint16_t count[16];
  uint16_t symbol[288];
  int fastbits = 0;
  int ncodes = 0;

  // returns false for an over-subscribed set, or an incomplete one that is not a single code
  bool build(const uint8_t *lens, int n, int fast_bits) {
    fastbits = fast_bits;
    std::memset(count, 0, sizeof(count));
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    ncodes = n - count[0];
    count[0] = 0;
    int left = 1;
    for (int len = 1; len <= 15; ++len) {
      left <<= 1;
      left -= count[len];
      if (left < 0) return false;
    }
    if (left > 0 && ncodes != 1 && ncodes != 0) return false;
    uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = static_cast<uint16_t>(offs[len] + count[len]);
    for (int i = 0; i < n; ++i) {
      if (lens[i]) symbol[offs[lens[i]]++] = static_cast<uint16_t>(i);
    }
    std::memset(fast, 0, sizeof(uint16_t) << fastbits);
    unsigned code = 0;
    int idx = 0;
    for (int len = 1; len <= fastbits; ++len) {
      for (int k = 0; k < count[len]; ++k, ++code, ++idx) {
        unsigned rev = 0;  // codes are sent most significant bit first
        for (int b = 0; b < len; ++b) rev |= ((code >> b) & 1u) << (len - 1 - b);
        const uint16_t e = static_cast<uint16_t>((symbol[idx] << 4) | len);
        for (unsigned j = rev; j < (1u << fastbits); j += (1u << len)) fast[j] = e;
      }
      code <<= 1;
    }
    return true;
  }
  // Builds the packed tables from fast[]; `dist` selects the distance alphabet's bases.
  void pack(bool dist) {
    const int nsym = dist ? 32 : 288;
    for (int sym = 0; sym < nsym; ++sym) {
      uint32_t e;
      if (dist) e = sym < 30 ? (static_cast<uint32_t>(kDistExtra[sym]) << 4) | (static_cast<uint32_t>(kDistBase[sym]) << 16) : ENT_INVALID;
      else if (sym < 256) e = ENT_LITERAL | (static_cast<uint32_t>(sym) << 16);
      else if (sym == 256) e = ENT_EOB;
      else if (sym < 286) e = (static_cast<uint32_t>(kLenExtra[sym - 257]) << 4) | (static_cast<uint32_t>(kLenBase[sym - 257]) << 16);
      else e = ENT_INVALID;
      ent[sym] = e;
    }
    const uint32_t n = 1u << fastbits;
    for (uint32_t i = 0; i < n; ++i) {
      const uint16_t f = fast[i];
      packed[i] = f ? (ent[f >> 4] | (f & 15u)) : 0u;
    }
  }
  // packed entry of the next code (code length in its low 4 bits); 0 if there is no such code
  inline uint32_t lookup(uint64_t bits) const {
    const uint32_t e = packed[bits & ((1u << fastbits) - 1)];
    if (e) return e;
    int len;
    const int sym = decode(bits, &len);
    if (sym < 0) return 0u;
    return ent[sym] | static_cast<uint32_t>(len);  // len <= 15
  }
  // returns the symbol and sets *len, or -1
  inline int decode(uint64_t bits, int *len) const {
    const uint16_t e = fast[bits & ((1u << fastbits) - 1)];
    if (e) {
      *len = e & 15;
      return e >> 4;
    }
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
      code |= static_cast<int>(bits & 1);
      bits >>= 1;
      const int c = count[l];
      if (code - c < first) {
        *len = l;
        return symbol[index + (code - first)];
      }
      index += c;
      first += c;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  }
};

struct StaticHuff {
  Huff lit, dist;
  StaticHuff() {
    uint8_t l[288];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    lit.build(l, 288, FAST_L);
    lit.pack(false);
    uint8_t d[30];
    for (int i = 0; i < 30; ++i) d[i] = 5;
    dist.build(d, 30, FAST_D);
    dist.pack(true);
  }
};
const StaticHuff &staticHuff() {
  static const StaticHuff s;
  return s;
}

// Reads a dynamic block's code lengths (after the 3 header bits) and builds both decoders.
bool readDynamic(const Bits &in, uint64_t *at, Huff *lit, Huff *dist) {
  uint64_t pos = *at;
  if (pos + 14 > in.nbits) return false;
  uint64_t v = in.peek(pos);
  const int hlit = static_cast<int>(v & 31) + 257, hdist = static_cast<int>((v >> 5) & 31) + 1,
            hclen = static_cast<int>((v >> 10) & 15) + 4;
  if (hlit > 286 || hdist > 30) return false;
  pos += 14;
  uint8_t cl[19] = {};
  if (pos + 3 * static_cast<uint64_t>(hclen) > in.nbits) return false;
  v = in.peek(pos);
  for (int i = 0; i < hclen; ++i) {  // 19 * 3 = 57 bits: one peek is not always enough
    if (i == 16) v = in.peek(pos + 48);
    cl[kClOrder[i]] = static_cast<uint8_t>((v >> (3 * (i & 15))) & 7);
  }
  pos += 3 * static_cast<uint64_t>(hclen);
  Huff clh;
  if (!clh.build(cl, 19, 7)) return false;
  if (clh.ncodes < 1) return false;
  uint8_t lens[286 + 30] = {};
  int n = 0;
  const int total = hlit + hdist;
  while (n < total) {
    if (pos >= in.nbits) return false;
    v = in.peek(pos);
    int len;
    const int sym = clh.decode(v, &len);
    if (sym < 0) return false;
    v >>= len;
    pos += static_cast<uint64_t>(len);
    if (sym < 16) {
      lens[n++] = static_cast<uint8_t>(sym);
    } else {
      int rep, val = 0;
      if (sym == 16) {
        if (n == 0) return false;
        val = lens[n - 1];
        rep = 3 + static_cast<int>(v & 3);
        pos += 2;
      } else if (sym == 17) {
        rep = 3 + static_cast<int>(v & 7);
        pos += 3;
      } else {
        rep = 11 + static_cast<int>(v & 127);
        pos += 7;
      }
      if (n + rep > total) return false;
      while (rep--) lens[n++] = static_cast<uint8_t>(val);
    }
  }
  if (pos > in.nbits || lens[256] == 0) return false;
  if (!lit->build(lens, hlit, FAST_L)) return false;
  if (!dist->build(lens + hlit, hdist, FAST_D)) return false;
  if (lit->ncodes < 2) return false;  // zlib never writes fewer; keeps the block-start search strict
  lit->pack(false);
  dist->pack(true);
  *at = pos;
  return true;
}

// ---- sinks ---------------------------------------------------------------------------------------------
struct NullSink {  // block-start validation
  uint64_t n = 0;
  inline bool lit(uint8_t) { ++n; return true; }
  inline bool match(uint32_t len, uint32_t) { n += len; return true; }
  inline bool raw(const uint8_t *, uint32_t len) { n += len; return true; }
};

// The one-pass form: the whole chunk in 16-bit symbols (a byte, or 256 + k for "byte k of the predecessor's final
// window").  The buffer opens with the W references to that window, so a match that reaches back before the chunk
// copies them like any other symbols and the chunk's own tail is the sliding window: no window of its own, no slide.
struct SymbolSink {
  std::vector<uint16_t> *store;
  std::atomic<size_t> *budget_used;  // symbol bytes held by all chunks of this call
  size_t budget;                     // give up (the serial readers take over) rather than outgrow the machine
  uint16_t *sym = nullptr;  // sym[W + i] = symbol of the chunk's byte i
  size_t cap = 0;           // in symbols, the W leading references included
  uint64_t n = 0;
  inline bool grow(uint64_t need) {
    if (W + n + need + 8 <= cap) return true;
    const size_t want = std::max<size_t>(cap * 2, static_cast<size_t>(W + n + need) + (size_t(1) << 20));
    if (budget_used->fetch_add((want - cap) * sizeof(uint16_t)) + (want - cap) * sizeof(uint16_t) > budget) return false;
    std::vector<uint16_t> bigger;
    detail::resizeUninitialized(&bigger, want);
    detail::adviseHugePages(bigger.data(), want * sizeof(uint16_t));  // written once, front to back: 2 MiB faults
    if (sym) std::memcpy(bigger.data(), sym, (W + static_cast<size_t>(n)) * sizeof(uint16_t));
    else for (uint32_t k = 0; k < W; ++k) bigger[k] = static_cast<uint16_t>(256 + k);
    store->swap(bigger);
    sym = store->data();
    cap = want;
    return true;
  }
  inline bool lit(uint8_t b) {
    if (W + n + 9 > cap && !grow(1)) return false;
    sym[W + n++] = b;
    return true;
  }
  inline bool match(uint32_t len, uint32_t dist) {
    if (W + n + len + 8 > cap && !grow(len)) return false;
    uint16_t *d = sym + W + n;
    const uint16_t *s = d - dist;  // dist <= 32768 = W: never before the buffer
    if (dist >= 4) {
      uint32_t k = 0;
      for (; k < len; k += 4) std::memcpy(d + k, s + k, 8);  // may write up to 3 symbols past len: room is kept
    } else {
      for (uint32_t k = 0; k < len; ++k) d[k] = s[k];
    }
    n += len;
    return true;
  }
  inline bool raw(const uint8_t *src, uint32_t len) {
    if (!grow(len)) return false;
    uint16_t *d = sym + W + n;
    for (uint32_t k = 0; k < len; ++k) d[k] = src[k];
    n += len;
    return true;
  }
  // symbol of the byte `back` positions before the end (1 <= back <= W)
  inline uint16_t tail(uint32_t back) const { return sym[W + n - back]; }
};

template <class Sink>
bool decodeHuffBlock(const Bits &in, uint64_t *at, const Huff &L, const Huff &D, Sink &sink) {
  uint64_t pos = *at;
  for (;;) {
    if (pos >= in.nbits) return false;
    uint64_t bits = in.peek(pos);  // >= 56 valid bits
    uint32_t e = L.lookup(bits);
    // up to three literals per refill (3 x 15 bits <= 56)
    if (e & ENT_LITERAL) {
      if (!sink.lit(static_cast<uint8_t>(e >> 16))) return false;
      pos += e & 15u;
      bits >>= e & 15u;
      e = L.lookup(bits);
      if (e & ENT_LITERAL) {
        if (!sink.lit(static_cast<uint8_t>(e >> 16))) return false;
        pos += e & 15u;
        bits >>= e & 15u;
        e = L.lookup(bits);
        if (e & ENT_LITERAL) {
          if (!sink.lit(static_cast<uint8_t>(e >> 16))) return false;
          pos += e & 15u;
        }
      }
      continue;  // refill before anything that needs more than a code
    }
    if (e == 0u || (e & ENT_INVALID)) return false;
    uint32_t used = e & 15u;
    if (e & ENT_EOB) {
      pos += used;
      if (pos > in.nbits) return false;
      *at = pos;
      return true;
    }
    bits >>= used;
    const uint32_t lextra = (e >> 4) & 15u;
    const uint32_t length = (e >> 16) + static_cast<uint32_t>(bits & ((1u << lextra) - 1u));
    bits >>= lextra;
    used += lextra;  // <= 20
    const uint32_t d = D.lookup(bits);
    if (d == 0u || (d & ENT_INVALID)) return false;
    bits >>= d & 15u;
    const uint32_t dextra = (d >> 4) & 15u;
    const uint32_t dist = (d >> 16) + static_cast<uint32_t>(bits & ((1u << dextra) - 1u));
    used += (d & 15u) + dextra;  // <= 48 of the >= 56 bits
    pos += used;
    if (pos > in.nbits) return false;
    if (!sink.match(length, dist)) return false;
  }
}

enum Outcome { FAILED, LINKED, FINAL };

// Decodes whole blocks from `start` until a block would start at `stop` (LINKED) or the final block ends
// (FINAL, *end = first bit after it).
template <class Sink>
Outcome decodeBlocks(const Bits &in, uint64_t start, uint64_t stop, Sink &sink, uint64_t *end) {
  uint64_t pos = start;
  Huff lit, dist;
  for (;;) {
    if (pos == stop) return LINKED;
    if (pos > stop || pos + 3 > in.nbits) return FAILED;
    const uint64_t v = in.peek(pos);
    const bool final_block = v & 1;
    const int type = static_cast<int>((v >> 1) & 3);
    pos += 3;
    if (type == 0) {
      pos = (pos + 7) & ~uint64_t(7);
      if (pos + 32 > in.nbits) return FAILED;
      const uint64_t h = in.peek(pos);
      const uint32_t len = static_cast<uint32_t>(h & 0xffff), nlen = static_cast<uint32_t>((h >> 16) & 0xffff);
      if ((len ^ nlen) != 0xffff) return FAILED;
      pos += 32;
      if (pos + 8 * static_cast<uint64_t>(len) > in.nbits) return FAILED;
      if (!sink.raw(in.p + (pos >> 3), len)) return FAILED;
      pos += 8 * static_cast<uint64_t>(len);
    } else if (type == 1) {
      const StaticHuff &s = staticHuff();
      if (!decodeHuffBlock(in, &pos, s.lit, s.dist, sink)) return FAILED;
    } else if (type == 2) {
      if (!readDynamic(in, &pos, &lit, &dist)) return FAILED;
      if (!decodeHuffBlock(in, &pos, lit, dist, sink)) return FAILED;
    } else {
      return FAILED;
    }
    if (final_block) {
      *end = pos;
      return FINAL;
    }
  }
}

// First bit position in [lo, hi) where a non-final dynamic block starts, decodes to its end-of-block and is
// followed by a plausible header.
uint64_t findBlockStart(const Bits &in, uint64_t lo, uint64_t hi) {
  Huff lit, dist;
  for (uint64_t p = lo; p < hi && p + 64 < in.nbits; ++p) {
    const uint64_t v = in.peek(p);
    if ((v & 7) != 4) continue;                          // BFINAL = 0, BTYPE = 2
    if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;
    {  // the code-length code must be complete (zlib's always is): Kraft sum over its 3-bit lengths
      const int hclen = static_cast<int>((v >> 13) & 15) + 4;
      uint64_t c = v >> 17;  // 39+ valid bits = 13 lengths; the rest from a second peek
      unsigned kraft = 0;
      for (int i = 0; i < hclen; ++i) {
        if (i == 13) c = in.peek(p + 17 + 39);
        const unsigned l = static_cast<unsigned>(c & 7);
        c >>= 3;
        if (l) kraft += 128u >> l;
      }
      if (kraft != 128) continue;
    }
    uint64_t pos = p + 3;
    if (!readDynamic(in, &pos, &lit, &dist)) continue;
    NullSink sink;
    if (!decodeHuffBlock(in, &pos, lit, dist, sink)) continue;
    if (sink.n < 64) continue;                           // real blocks carry thousands of bytes
    if (pos + 3 > in.nbits || ((in.peek(pos) >> 1) & 3) == 3) continue;
    return p;
  }
  return NONE;
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

std::atomic<uint64_t> g_successes{0};

}  // namespace

uint64_t successCount() { return g_successes.load(); }

bool inflate(const uint8_t *gz, size_t size, size_t header_len, int threads, std::vector<uint8_t> *out) {
  if (gz == nullptr || out == nullptr || threads < 2 || size < header_len + 8 + (size_t(1) << 20)) return false;
  const size_t dbytes = size - header_len - 8;
  Bits in{gz + header_len, 8 * static_cast<uint64_t>(dbytes), dbytes};
  auto le32 = [&](const uint8_t *q) {
    return static_cast<uint32_t>(q[0]) | (static_cast<uint32_t>(q[1]) << 8) | (static_cast<uint32_t>(q[2]) << 16) |
           (static_cast<uint32_t>(q[3]) << 24);
  };
  const uint32_t want_crc = le32(gz + size - 8);
  const uint64_t isize = le32(gz + size - 4);
  // ISIZE is the length modulo 2^32; it is compared with what the decode produced, never used to allocate

  static const bool timing = std::getenv("SPZ_AMD_PINFLATE_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[pinflate] %-8s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  const size_t nchunks = std::min<size_t>(static_cast<size_t>(threads) * 4, dbytes / (size_t(256) << 10));  // four per thread: the spans between block starts are uneven
  if (nchunks < 2) return false;
  if (((in.peek(0) >> 1) & 3) == 0) return false;  // opens with a stored block: incompressible data, nothing to gain
  // ---- 1. block starts
  std::vector<uint64_t> start(nchunks, NONE);
  start[0] = 0;
  parallel_for(nchunks - 1, threads, [&](size_t k) {
    const size_t i = k + 1;
    const uint64_t lo = 8 * static_cast<uint64_t>(dbytes / nchunks * i), hi = 8 * static_cast<uint64_t>(dbytes / nchunks * (i + 1));
    // zlib's Huffman blocks are at most ~60 KiB of compressed data, so a start that exists is near; the cap
    // keeps the cost of finding out that there is none (stored data) small
    const uint64_t cap = lo + 8 * (uint64_t(128) << 10);
    start[i] = findBlockStart(in, lo, std::min<uint64_t>(cap, i + 1 == nchunks ? in.nbits : hi));
  });
  lap("search");
  std::vector<size_t> live;  // chunks that have a block start
  for (size_t i = 0; i < nchunks; ++i) {
    if (start[i] != NONE) live.push_back(i);
  }
  if (live.size() < 2 || live.size() * 2 < nchunks) return false;  // mostly stored / static data: no gain
  const size_t n = live.size();
  if (timing) {
    uint64_t span = 0;
    for (size_t j = 0; j < n; ++j) span = std::max<uint64_t>(span, (j + 1 < n ? start[live[j + 1]] : in.nbits) - start[live[j]]);
    std::fprintf(stderr, "[pinflate] %zu chunks, %zu with a block start, largest span %.1f%% of the data\n", nchunks, n,
                 100.0 * static_cast<double>(span) / static_cast<double>(in.nbits));
  }

  // ---- 2. decode every chunk once, into 16-bit symbols.  Nothing is allocated from the trailer's ISIZE: the
  // output size is what the decode itself produces.  Memory: 2 bytes per output byte until step 4 has placed a chunk.
  std::vector<uint64_t> length(n, 0);
  std::vector<std::vector<uint16_t>> symbols(n);
  std::vector<uint64_t> end_bit(n, 0);
  std::atomic<bool> ok{true};
  std::atomic<size_t> symbol_bytes{0};
  std::vector<double> chunk_s(n, 0.0);
  // at most half of the memory the machine has free right now (unknown: 64 GiB) goes into symbol buffers
  const long avail_pages = sysconf(_SC_AVPHYS_PAGES), page_size = sysconf(_SC_PAGESIZE);
  const size_t budget = (avail_pages > 0 && page_size > 0) ? static_cast<size_t>(avail_pages) / 2 * static_cast<size_t>(page_size)
                                                           : (size_t(64) << 30);
  parallel_for(n, threads, [&](size_t j) {
    const uint64_t from = start[live[j]], to = (j + 1 < n) ? start[live[j + 1]] : NONE;
    uint64_t end = 0;
    SymbolSink sink{&symbols[j], &symbol_bytes, budget};
    if (!sink.grow(static_cast<uint64_t>(((to == NONE ? in.nbits : to) - from) / 8 * 2))) {  // a first guess: ratio 2
      ok = false;
      return;
    }
    const auto t_chunk = std::chrono::steady_clock::now();
    const Outcome r = decodeBlocks(in, from, to, sink, &end);
    if (timing) chunk_s[j] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_chunk).count();
    length[j] = sink.n;
    end_bit[j] = end;
    if (r != (j + 1 < n ? LINKED : FINAL)) ok = false;
  });
  lap("decode");
  if (timing) {
    double sum = 0, mx = 0;
    for (double t : chunk_s) {
      sum += t;
      mx = std::max(mx, t);
    }
    std::fprintf(stderr, "[pinflate] chunk decode times: sum %.3f s, longest %.3f s, %d threads\n", sum, mx, threads);
  }
  if (!ok) return false;
  if (((end_bit[n - 1] + 7) >> 3) != dbytes) return false;  // the member must end exactly at the trailer

  // ---- 3. offsets and contexts
  std::vector<uint64_t> offset(n + 1, 0);
  for (size_t j = 0; j < n; ++j) offset[j + 1] = offset[j] + length[j];
  if ((offset[n] & 0xffffffffull) != isize) return false;  // ISIZE is the length modulo 2^32
  out->clear();
  detail::resizeUninitialized(out, static_cast<size_t>(offset[n]));  // step 4 writes every byte; the CRC-32 decides
  {
    detail::Prefault prefault;  // 650 MB of fresh 4 KiB pages faulted in by 16 writers cost more than the copy itself
    prefault.add(out->data(), out->size());
    prefault.start();
    prefault.join();
  }
  std::vector<std::vector<uint8_t>> ctx(n);  // ctx[j] = the W bytes before chunk j + 1's first byte
  for (size_t j = 0; j + 1 < n; ++j) {
    ctx[j].assign(W, 0);
    // the last W symbols of the chunk; a chunk shorter than W reaches into its leading references, as it should
    const uint16_t *win = symbols[j].data() + length[j];
    for (uint32_t k = 0; k < W; ++k) {
      const uint16_t s = win[k];
      if (s >= 256 && j == 0) return false;  // the first chunk has no left context: a reference there is invalid data
      ctx[j][k] = s < 256 ? static_cast<uint8_t>(s) : ctx[j - 1][s - 256];
    }
  }

  // ---- 4. place: symbols -> bytes at the final offset, references resolved from the predecessor's context
  parallel_for(n, threads, [&](size_t j) {
    uint8_t *dst = out->data() + offset[j];
    const uint16_t *src = symbols[j].data() + W;
    const uint8_t *c = j ? ctx[j - 1].data() : nullptr;
    const uint64_t len = length[j];
    uint64_t i = 0;
    for (; i + 32 <= len; i += 32) {
      unsigned any = 0;
      for (int k = 0; k < 32; ++k) {
        dst[i + k] = static_cast<uint8_t>(src[i + k]);
        any |= src[i + k];
      }
      if (any >= 256) {
        if (!c) {
          ok = false;
          return;
        }
        for (int k = 0; k < 32; ++k) {
          if (src[i + k] >= 256) dst[i + k] = c[src[i + k] - 256];
        }
      }
    }
    for (; i < len; ++i) {
      const uint16_t sy = src[i];
      if (sy >= 256 && !c) {
        ok = false;
        return;
      }
      dst[i] = sy < 256 ? static_cast<uint8_t>(sy) : c[sy - 256];
    }
  });
  lap("place");

  // ---- 5. CRC-32 and ISIZE
  const uint64_t total = offset[n];
  const size_t pieces = std::min<size_t>(static_cast<size_t>(threads), static_cast<size_t>(total / (size_t(1) << 20)) + 1);
  std::vector<uLong> crcs(pieces);
  std::vector<uint64_t> cut(pieces + 1);
  for (size_t i = 0; i <= pieces; ++i) cut[i] = total / pieces * i;
  cut[pieces] = total;
  parallel_for(pieces, threads, [&](size_t i) {
    uLong c = crc32(0L, Z_NULL, 0);
    for (uint64_t p = cut[i]; p < cut[i + 1];) {
      const uint64_t m = std::min<uint64_t>(cut[i + 1] - p, uint64_t(1) << 30);
      c = crc32(c, out->data() + p, static_cast<uInt>(m));
      p += m;
    }
    crcs[i] = c;
  });
  lap("crc");
  uLong crc = crcs[0];
  for (size_t i = 1; i < pieces; ++i) crc = crc32_combine(crc, crcs[i], static_cast<z_off_t>(cut[i + 1] - cut[i]));
  // Returning the symbol buffers (2 bytes per output byte) to the system costs ~50 ms per GB (munmap of huge pages) and
  // nobody waits for it: a detached thread does it — after the CRC pass, whose threads its TLB shoot-downs would
  // otherwise interrupt (27 ms became 90), and not inside the placing loop either, where an munmap's hold on the
  // address space's lock stalls the other threads' page faults.
  try {
    std::thread([dead = std::move(symbols)]() mutable { dead.clear(); }).detach();
  } catch (const std::system_error &) {
    // no thread to be had: the buffers go when `symbols` does
  }
  symbols.clear();
  if (static_cast<uint32_t>(crc) != want_crc) return false;
  g_successes.fetch_add(1);
  return true;
}

}  // namespace pinflate
}  // namespace spz
