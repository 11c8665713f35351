// spz_inflate_core.hpp — the deflate decoder of the parallel readers, shared by the host reader
// (spz_inflate.cpp) and the device reader (spz_inflate_dev.hip): bit access, canonical Huffman tables with a
// packed fast table, dynamic-block headers, the block decode loop over a symbol sink, and the tests that make a
// bit position a plausible block start.  RFC 1951; nothing here allocates.
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define SPZ_INF_HD __host__ __device__ __forceinline__
#else
#define SPZ_INF_HD inline
#endif
// header loops stay rolled in the kernels (unrolled they cost registers the decode loop needs); gcc has no such pragma
#if defined(__clang__)
#define SPZ_INF_NO_UNROLL _Pragma("clang loop unroll(disable)")
#else
#define SPZ_INF_NO_UNROLL
#endif

namespace spz {
namespace pinflate {

constexpr uint32_t W = 32768;
constexpr int FAST_L = 11, FAST_D = 9;
constexpr uint64_t NONE = ~uint64_t(0);

SPZ_INF_HD uint32_t lenBase(int i) {
  constexpr uint16_t t[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  return t[i];
}
SPZ_INF_HD uint32_t lenExtra(int i) {
  constexpr uint8_t t[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  return t[i];
}
SPZ_INF_HD uint32_t distBase(int i) {
  constexpr uint16_t t[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  return t[i];
}
SPZ_INF_HD uint32_t distExtra(int i) {
  constexpr uint8_t t[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  return t[i];
}
SPZ_INF_HD int clOrder(int i) {
  constexpr uint8_t t[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  return t[i];
}

struct Bits {  // the deflate data of the member
  const uint8_t *p;
  uint64_t nbits;
  size_t nbytes;
  // >= 56 valid bits starting at bit position `at` (zeros past the end)
  SPZ_INF_HD const uint8_t *bytes(uint64_t byte_offset) const { return p + byte_offset; }
  // a value every caller of one decoder agrees on (the device's lockstep decoder moves it to a scalar register)
  SPZ_INF_HD uint32_t uniform(uint32_t v) const { return v; }
  SPZ_INF_HD uint64_t peek(uint64_t at) const {
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

template <int FAST, int NSYM>
struct HuffT {
  uint16_t fast[1 << FAST];
  uint32_t packed[1 << FAST];  // filled by pack(): same index as fast[]
  uint32_t ent[NSYM];          // per symbol, without the code length (slow path)
  uint16_t count[16];
  uint16_t offs[16];           // build()'s running offsets (a member: a local array indexed by a code length lives in
                               // scratch memory on the device, 500 cycles an access)
  uint16_t symbol[NSYM];
  int ncodes;
  static constexpr int fastbits = FAST;

  // returns false for an over-subscribed set, or an incomplete one that is not a single code
  SPZ_INF_HD bool build(const uint8_t *lens, int n) {
    SPZ_INF_NO_UNROLL
    for (int i = 0; i < 16; ++i) count[i] = 0;
    SPZ_INF_NO_UNROLL
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    ncodes = n - count[0];
    count[0] = 0;
    int left = 1;
    SPZ_INF_NO_UNROLL
    for (int len = 1; len <= 15; ++len) {
      left <<= 1;
      left -= count[len];
      if (left < 0) return false;
    }
    if (left > 0 && ncodes != 1 && ncodes != 0) return false;
    offs[0] = 0;
    offs[1] = 0;
    SPZ_INF_NO_UNROLL
    for (int len = 1; len < 15; ++len) offs[len + 1] = static_cast<uint16_t>(offs[len] + count[len]);
    SPZ_INF_NO_UNROLL
    for (int i = 0; i < n; ++i) {
      if (lens[i]) symbol[offs[lens[i]]++] = static_cast<uint16_t>(i);
    }
    SPZ_INF_NO_UNROLL
    for (int i = 0; i < (1 << FAST); ++i) fast[i] = 0;
    unsigned code = 0;
    int idx = 0;
    SPZ_INF_NO_UNROLL
    for (int len = 1; len <= fastbits; ++len) {
      SPZ_INF_NO_UNROLL
      for (int k = 0; k < count[len]; ++k, ++code, ++idx) {
        unsigned rev = 0;  // codes are sent most significant bit first
        SPZ_INF_NO_UNROLL
        for (int b = 0; b < len; ++b) rev |= ((code >> b) & 1u) << (len - 1 - b);
        const uint16_t e = static_cast<uint16_t>((symbol[idx] << 4) | len);
        SPZ_INF_NO_UNROLL
        for (unsigned j = rev; j < (1u << fastbits); j += (1u << len)) fast[j] = e;
      }
      code <<= 1;
    }
    return true;
  }
  // Builds the packed tables from fast[]; `dist` selects the distance alphabet's bases.
  SPZ_INF_HD void pack(bool dist) {
    const int nsym = dist ? 32 : 288;
    SPZ_INF_NO_UNROLL
    for (int sym = 0; sym < nsym; ++sym) {
      uint32_t e;
      if (dist) e = sym < 30 ? (distExtra(sym) << 4) | (distBase(sym) << 16) : ENT_INVALID;
      else if (sym < 256) e = ENT_LITERAL | (static_cast<uint32_t>(sym) << 16);
      else if (sym == 256) e = ENT_EOB;
      else if (sym < 286) e = (lenExtra(sym - 257) << 4) | (lenBase(sym - 257) << 16);
      else e = ENT_INVALID;
      ent[sym] = e;
    }
    const uint32_t n = 1u << fastbits;
    SPZ_INF_NO_UNROLL
    for (uint32_t i = 0; i < n; ++i) {
      const uint16_t f = fast[i];
      packed[i] = f ? (ent[f >> 4] | (f & 15u)) : 0u;
    }
  }
  // packed entry of the next code (code length in its low 4 bits); 0 if there is no such code
  SPZ_INF_HD uint32_t lookup(uint64_t bits) const {
    const uint32_t e = packed[bits & ((1u << fastbits) - 1)];
    if (e) return e;
    int len;
    const int sym = decode(bits, &len);
    if (sym < 0) return 0u;
    return ent[sym] | static_cast<uint32_t>(len);  // len <= 15
  }
  // returns the symbol and sets *len, or -1
  SPZ_INF_HD int decode(uint64_t bits, int *len) const {
    const uint16_t e = fast[bits & ((1u << fastbits) - 1)];
    if (e) {
      *len = e & 15;
      return e >> 4;
    }
    int code = 0, first = 0, index = 0;
    SPZ_INF_NO_UNROLL
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

using HuffLit = HuffT<FAST_L, 288>;
using HuffDist = HuffT<FAST_D, 32>;

// The fixed Huffman codes of block type 1 (RFC 1951 §3.2.6), built into the caller's tables.
template <class HL, class HD>
SPZ_INF_HD void buildStatic(HL *lit, HD *dist, uint8_t *lens /* 316 bytes of the caller's */) {
  uint8_t *l = lens;
  SPZ_INF_NO_UNROLL
  for (int i = 0; i < 144; ++i) l[i] = 8;
  SPZ_INF_NO_UNROLL
  for (int i = 144; i < 256; ++i) l[i] = 9;
  SPZ_INF_NO_UNROLL
  for (int i = 256; i < 280; ++i) l[i] = 7;
  SPZ_INF_NO_UNROLL
  for (int i = 280; i < 288; ++i) l[i] = 8;
  lit->build(l, 288);
  lit->pack(false);
  uint8_t *d = lens;
  SPZ_INF_NO_UNROLL
  for (int i = 0; i < 30; ++i) d[i] = 5;
  dist->build(d, 30);
  dist->pack(true);
}

// What reading a dynamic block's header needs besides the two decoders: the code-length code, its lengths, the
// literal/length + distance code lengths.  The caller's memory — a local on the host, LDS on the device (as locals these
// arrays, indexed by data, are scratch memory there: the device's block-start search spent most of its time in them).
struct HeaderWork {
  HuffT<7, 19> clh;
  uint8_t lens[286 + 30];
  uint8_t cl[19];
  uint16_t count[16];
  uint32_t walked;  // code-length symbols the last header check decoded (statistics of the device's search)
};

// Same acceptance rule as HuffT::build(): not over-subscribed, and complete unless it has at most one code.
SPZ_INF_HD bool completeCode(const uint8_t *lens, int n, int *ncodes, uint16_t *count) {
  SPZ_INF_NO_UNROLL
  for (int i = 0; i < 16; ++i) count[i] = 0;
  SPZ_INF_NO_UNROLL
  for (int i = 0; i < n; ++i) count[lens[i]]++;
  const int codes = n - count[0];
  int left = 1;
  SPZ_INF_NO_UNROLL
  for (int len = 1; len <= 15; ++len) {
    left <<= 1;
    left -= count[len];
    if (left < 0) return false;
  }
  *ncodes = codes;
  return !(left > 0 && codes != 1 && codes != 0);
}

// Reads a dynamic block's code lengths (after the 3 header bits): lens[0 .. hlit) literal/length, lens[hlit .. hlit + hdist)
// distance codes.
template <class In>
SPZ_INF_HD bool readCodeLengths(const In &in, uint64_t *at, HeaderWork *w, int *hlit_out, int *hdist_out) {
  uint8_t *lens = w->lens;
  uint8_t *cl = w->cl;
  HuffT<7, 19> &clh = w->clh;
  uint64_t pos = *at;
  if (pos + 14 > in.nbits) return false;
  uint64_t v = in.peek(pos);
  const int hlit = static_cast<int>(v & 31) + 257, hdist = static_cast<int>((v >> 5) & 31) + 1,
            hclen = static_cast<int>((v >> 10) & 15) + 4;
  if (hlit > 286 || hdist > 30) return false;
  pos += 14;
  SPZ_INF_NO_UNROLL
  for (int i = 0; i < 19; ++i) cl[i] = 0;
  if (pos + 3 * static_cast<uint64_t>(hclen) > in.nbits) return false;
  v = in.peek(pos);
  SPZ_INF_NO_UNROLL
  for (int i = 0; i < hclen; ++i) {  // 19 * 3 = 57 bits: one peek is not always enough
    if (i == 16) v = in.peek(pos + 48);
    cl[clOrder(i)] = static_cast<uint8_t>((v >> (3 * (i & 15))) & 7);
  }
  pos += 3 * static_cast<uint64_t>(hclen);
  if (!clh.build(cl, 19)) return false;
  if (clh.ncodes < 1) return false;
  int n = 0;
  const int total = hlit + hdist;
  // (lens[] is not cleared: every entry below hlit + hdist is written before it is read — a repeat code reads the one
  // before it, the tables are built from exactly that many — and clearing all 316 was a fifth of a header check's work)
  // Kraft sums of the two sets as they come, in units of 2^-15: an over-subscribed set (which build() would refuse
  // in the end anyway) ends the reading at once — random bits that look like a header get there within a few
  // dozen lengths, and the block-start searches spend most of their time on those.
  uint32_t kraft_lit = 0, kraft_dist = 0;
  auto add = [&](int from, int to, int len) {
    if (len == 0) return true;
    SPZ_INF_NO_UNROLL
    for (int i = from; i < to; ++i) {
      if (i < hlit) kraft_lit += 32768u >> len;
      else kraft_dist += 32768u >> len;
    }
    return kraft_lit <= 32768u && kraft_dist <= 32768u;
  };
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
      if (!add(n - 1, n, sym)) return false;
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
      if (!add(n, n + rep, val)) return false;
      while (rep--) lens[n++] = static_cast<uint8_t>(val);
    }
  }
  if (pos > in.nbits || lens[256] == 0) return false;
  *hlit_out = hlit;
  *hdist_out = hdist;
  *at = pos;
  return true;
}

// ... and builds both decoders.
template <class In, class HL, class HD>
SPZ_INF_HD bool readDynamic(const In &in, uint64_t *at, HL *lit, HD *dist, HeaderWork *w) {
  int hlit = 0, hdist = 0;
  uint64_t pos = *at;
  if (!readCodeLengths(in, &pos, w, &hlit, &hdist)) return false;
  const uint8_t *lens = w->lens;
  if (!lit->build(lens, hlit)) return false;
  if (!dist->build(lens + hlit, hdist)) return false;
  if (lit->ncodes < 2) return false;  // zlib never writes fewer; keeps the block-start search strict
  lit->pack(false);
  dist->pack(true);
  *at = pos;
  return true;
}

// ---- sinks ---------------------------------------------------------------------------------------------
struct NullSink {  // block-start validation
  SPZ_INF_HD void mark() {}
  uint64_t n;
  SPZ_INF_HD bool lit(uint8_t) { ++n; return true; }
  SPZ_INF_HD bool match(uint32_t len, uint32_t) { n += len; return true; }
  SPZ_INF_HD bool raw(const uint8_t *, uint32_t len) { n += len; return true; }
};


template <class In, class HL, class HD, class Sink>
SPZ_INF_HD bool decodeHuffBlock(const In &in, uint64_t *at, const HL &L, const HD &D, Sink &sink, uint64_t limit = NONE) {
  // `limit`: a bit position at which a block is known (or believed) to start: a block that runs past it is not one
  // that ends there, and a decoder on a look-alike start must not run on to the end of the data
  uint64_t pos = *at;
  for (;;) {
    if (pos >= in.nbits || pos > limit) return false;
    uint64_t bits = in.peek(pos);  // >= 56 valid bits
    uint32_t e = in.uniform(L.lookup(bits));
    // up to three literals per refill (3 x 15 bits <= 56)
    if (e & ENT_LITERAL) {
      if (!sink.lit(static_cast<uint8_t>(e >> 16))) return false;
      pos += e & 15u;
      bits >>= e & 15u;
      e = in.uniform(L.lookup(bits));
      if (e & ENT_LITERAL) {
        if (!sink.lit(static_cast<uint8_t>(e >> 16))) return false;
        pos += e & 15u;
        bits >>= e & 15u;
        e = in.uniform(L.lookup(bits));
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
    const uint32_t d = in.uniform(D.lookup(bits));
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
// `lit`, `dist`: the caller's table memory (14 KiB + 3 KiB).
// `huff`: the decoder of one Huffman-coded block's symbols, (in, &pos, lit, dist, sink, stop) -> bool: decodeHuffBlock
// above, or the device's wave-wide form of it (spz_inflate_dev.hip).
template <class In, class HL, class HD, class Sink, class HuffFn>
SPZ_INF_HD Outcome decodeBlocksWith(const In &in, uint64_t start, uint64_t stop, Sink &sink, uint64_t *end, HL *lit, HD *dist,
                                    HeaderWork *work, HuffFn huff) {
  uint64_t pos = start;
  for (;;) {
    if (pos == stop) return LINKED;
    if (pos > stop || pos + 3 > in.nbits) return FAILED;
    *end = pos;   // on FAILED: where the block that did not work out begins (with sink.mark(): how far the output was)
    sink.mark();
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
      if (!sink.raw(in.bytes(pos >> 3), len)) return FAILED;
      pos += 8 * static_cast<uint64_t>(len);
    } else if (type == 1) {
      buildStatic(lit, dist, work->lens);
      if (!huff(in, &pos, *lit, *dist, sink, stop)) return FAILED;
    } else if (type == 2) {
      if (!readDynamic(in, &pos, lit, dist, work)) return FAILED;
      if (!huff(in, &pos, *lit, *dist, sink, stop)) return FAILED;
    } else {
      return FAILED;
    }
    if (final_block) {
      *end = pos;
      return FINAL;
    }
  }
}

template <class In, class HL, class HD, class Sink>
SPZ_INF_HD Outcome decodeBlocks(const In &in, uint64_t start, uint64_t stop, Sink &sink, uint64_t *end, HL *lit, HD *dist) {
  HeaderWork work;
  return decodeBlocksWith(in, start, stop, sink, end, lit, dist, &work,
                          [](const In &i, uint64_t *at, const HL &L, const HD &D, Sink &s, uint64_t limit) {
                            return decodeHuffBlock(i, at, L, D, s, limit);
                          });
}

// The cheap part of the block-start test at bit position p: BFINAL = 0, BTYPE = 2, code counts in range and a
// complete code-length code (zlib's always is: Kraft sum over its 3-bit lengths).
// ... in two parts, for a caller that tests many positions in lockstep (the device's search): the fields of the first 13
// bits, which one position in nine passes, and the code-length code's Kraft sum, a loop of up to 19 steps.
SPZ_INF_HD bool dynamicHeaderFieldsInRange(uint64_t v) {  // v: the bits from the block's first on
  return (v & 7) == 4 && ((v >> 3) & 31) <= 29 && ((v >> 8) & 31) <= 29;
}
template <class In>
SPZ_INF_HD bool codeLengthCodeComplete(const In &in, uint64_t p, uint64_t v) {  // v = in.peek(p)
  const int hclen = static_cast<int>((v >> 13) & 15) + 4;
  uint64_t c = v >> 17;  // 39+ valid bits = 13 lengths; the rest from a second peek
  unsigned kraft = 0;
  for (int i = 0; i < hclen; ++i) {
    if (i == 13) c = in.peek(p + 17 + 39);
    const unsigned l = static_cast<unsigned>(c & 7);
    c >>= 3;
    if (l) kraft += 128u >> l;
  }
  return kraft == 128;
}
template <class In>
SPZ_INF_HD bool plausibleDynamicHeader(const In &in, uint64_t p) {
  const uint64_t v = in.peek(p);
  return dynamicHeaderFieldsInRange(v) && codeLengthCodeComplete(in, p, v);
}

// Header and both code-length sets valid — exactly what readDynamic() accepts, without building tables or decoding
// the block: what the device's search takes; the chunk before it has to end exactly there, which is the rest of
// the proof.
// The same verdict as readCodeLengths() + completeCode() on both sets + "two literal / length codes at least", without
// storing a single code length: a check only needs the two Kraft sums (units of 2^-15: lengths are at most 15), the
// numbers of codes, the last length (what a repeat code repeats) and whether the end-of-block symbol has a code.  The
// block-start searches spend most of their time here — a position that passes the cheap test is usually walked to its
// last code length, random bits decoding mostly to zero runs that never over-subscribe — and the 316 byte stores and
// two counting passes per candidate were most of that.
template <class In>
SPZ_INF_HD bool hasValidDynamicHeader(const In &in, uint64_t p, HeaderWork *w) {
  if (p + 64 >= in.nbits || !plausibleDynamicHeader(in, p)) return false;
  uint8_t *cl = w->cl;
  HuffT<7, 19> &clh = w->clh;
  uint64_t pos = p + 3;
  if (pos + 14 > in.nbits) return false;
  uint64_t v = in.peek(pos);
  const int hlit = static_cast<int>(v & 31) + 257, hdist = static_cast<int>((v >> 5) & 31) + 1,
            hclen = static_cast<int>((v >> 10) & 15) + 4;
  if (hlit > 286 || hdist > 30) return false;
  pos += 14;
  SPZ_INF_NO_UNROLL
  for (int i = 0; i < 19; ++i) cl[i] = 0;
  if (pos + 3 * static_cast<uint64_t>(hclen) > in.nbits) return false;
  v = in.peek(pos);
  SPZ_INF_NO_UNROLL
  for (int i = 0; i < hclen; ++i) {
    if (i == 16) v = in.peek(pos + 48);
    cl[clOrder(i)] = static_cast<uint8_t>((v >> (3 * (i & 15))) & 7);
  }
  pos += 3 * static_cast<uint64_t>(hclen);
  if (!clh.build(cl, 19)) return false;
  if (clh.ncodes < 1) return false;
  const int total = hlit + hdist;
  int n = 0, prev = 0, codes_lit = 0, codes_dist = 0;
  uint32_t kraft_lit = 0, kraft_dist = 0;
  bool eob_has_code = false;
  w->walked = 0;
  // One code length symbol per step, written as selects: the device's search runs sixteen of these walks side by side in
  // lockstep lanes, and with a branch per kind of symbol the lanes took turns (measured: the checks' time was that of
  // all their symbols one after the other).  The code-length code is complete (the cheap test's Kraft sum) and at most
  // 7 bits long, so every pattern is in its direct table: an empty entry cannot happen for a header that passes, and is
  // the end of the walk if it does.
  bool bad = false;
  while (n < total && !bad) {
    ++w->walked;
    bad |= pos >= in.nbits;
    v = in.peek(pos);
    const uint32_t e = clh.fast[v & 127u];
    bad |= e == 0u;
    const int len = static_cast<int>(e & 15u), sym = static_cast<int>(e >> 4);
    v >>= len;
    const bool is16 = sym == 16, is17 = sym == 17, is18 = sym == 18;
    const int extra_bits = is16 ? 2 : (is17 ? 3 : (is18 ? 7 : 0));
    const int extra = static_cast<int>(v & ((1u << extra_bits) - 1u));
    const int rep = sym < 16 ? 1 : (is18 ? 11 : 3) + extra;
    const int val = sym < 16 ? sym : (is16 ? prev : 0);
    bad |= is16 & (n == 0);
    pos += static_cast<uint64_t>(len + extra_bits);
    bad |= n + rep > total;
    // entries [n, n + rep): those below hlit are literal / length codes, the rest distance codes
    const int below = hlit - n;
    const int in_lit = below <= 0 ? 0 : (rep < below ? rep : below);
    const uint32_t unit = val != 0 ? 32768u >> val : 0u;
    kraft_lit += unit * static_cast<uint32_t>(in_lit);
    kraft_dist += unit * static_cast<uint32_t>(rep - in_lit);
    bad |= (kraft_lit > 32768u) | (kraft_dist > 32768u);  // over-subscribed
    codes_lit += val != 0 ? in_lit : 0;
    codes_dist += val != 0 ? rep - in_lit : 0;
    eob_has_code |= (val != 0) & (n <= 256) & (256 < n + rep);
    prev = val;
    n += rep;
  }
  if (bad) return false;
  if (pos > in.nbits || !eob_has_code) return false;
  // complete, unless it has at most one code (HuffT::build()'s rule); two literal / length codes at least
  const bool lit_ok = kraft_lit == 32768u || codes_lit <= 1, dist_ok = kraft_dist == 32768u || codes_dist <= 1;
  return lit_ok && dist_ok && codes_lit >= 2;
}

// The full test: a non-final dynamic block starts at p, decodes to its end-of-block and is followed by a
// plausible header.
template <class In, class HL, class HD>
SPZ_INF_HD bool isBlockStart(const In &in, uint64_t p, HL *lit, HD *dist) {
  if (p + 64 >= in.nbits || !plausibleDynamicHeader(in, p)) return false;
  uint64_t pos = p + 3;
  HeaderWork work;
  if (!readDynamic(in, &pos, lit, dist, &work)) return false;
  NullSink sink;
  sink.n = 0;
  if (!decodeHuffBlock(in, &pos, *lit, *dist, sink)) return false;
  if (sink.n < 64) return false;                           // real blocks carry thousands of bytes
  if (pos + 3 > in.nbits || ((in.peek(pos) >> 1) & 3) == 3) return false;
  return true;
}

}  // namespace pinflate
}  // namespace spz
