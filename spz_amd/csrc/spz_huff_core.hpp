// spz_huff_core.hpp — zlib 1.2.11's tree construction for one deflate block (trees.c: build_tree, gen_bitlen,
// gen_codes, scan_tree / send_tree, build_bl_tree), restated once for the host writer (spz_deflate.cpp) and for the
// device's tree kernel (spz_lz77.hip).  The result has to be zlib's, bit for bit: the heap's tie-breaks (equal
// frequencies ordered by depth, then by what the heap moves happened to leave where), the overflow repair of
// gen_bitlen and the run-length walk of the code lengths are all kept as they are there.
// Storage is handed in by pointer: the host keeps a tree in its Block, the device keeps it in LDS (where dad[] and
// len[] are the same array, as in zlib's ct_data union: a node's length replaces its parent link once that is read).
//
// Acknowledgement: this file follows the structure of trees.c of zlib 1.2.11, (C) 1995-2017 Jean-loup Gailly and Mark
// Adler, statement by statement where the tie-breaks decide the output, and keeps its function names so that the two
// can be read side by side.  zlib is distributed under the zlib licence ("This software is provided 'as-is' ...
// altered source versions must be plainly marked as such, and must not be misrepresented as being the original
// software"); this is such an altered restatement, not the original software.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define SPZ_HUFF_HD __host__ __device__ inline
#else
#define SPZ_HUFF_HD inline
#endif

namespace spz {
namespace huff {

constexpr int L_CODES = 286, D_CODES = 30, BL_CODES = 19, HEAP_SIZE = 2 * L_CODES + 1, LITERALS = 256, END_BLOCK = 256;
constexpr int REP_3_6 = 16, REPZ_3_10 = 17, REPZ_11_138 = 18, MAX_BITS = 15, MAX_BL_BITS = 7;

enum Kind { LIT = 0, DIST = 1, BL = 2 };

// extra_lbits {0 x8, 1 x4, 2 x4, 3 x4, 4 x4, 5 x4, 0}, extra_dbits {0 x4, 1, 1, 2, 2, ... 13, 13}, extra_blbits
SPZ_HUFF_HD int extra_lbits(int code) { return (code < 8 || code == 28) ? 0 : (code >> 2) - 1; }
SPZ_HUFF_HD int extra_dbits(int code) { return code < 4 ? 0 : (code >> 1) - 1; }
SPZ_HUFF_HD int extra_blbits(int code) { return code == REP_3_6 ? 2 : code == REPZ_3_10 ? 3 : code == REPZ_11_138 ? 7 : 0; }
SPZ_HUFF_HD int bl_order(int rank) {  // {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15}
  return rank < 3 ? 16 + rank : rank == 3 ? 0 : (rank & 1) ? 8 - ((rank - 3) >> 1) : 8 + ((rank - 4) >> 1);
}
SPZ_HUFF_HD int static_llen(int n) { return n <= 143 ? 8 : n <= 255 ? 9 : n <= 279 ? 7 : 8; }

template <Kind K>
SPZ_HUFF_HD int extra_bits_of(int n) {
  return K == LIT ? (n > LITERALS ? extra_lbits(n - LITERALS - 1) : 0) : K == DIST ? extra_dbits(n) : extra_blbits(n);
}
template <Kind K>
SPZ_HUFF_HD int static_len_of(int n) {
  return K == LIT ? static_llen(n) : K == DIST ? 5 : 0;
}

SPZ_HUFF_HD unsigned bit_reverse(unsigned code, int len) {
  unsigned res = 0;
  do {
    res |= code & 1;
    code >>= 1;
    res <<= 1;
  } while (--len > 0);
  return res >> 1;
}

// len[0 .. max_code] and the count of codes per length -> code[]; next_code: 16 entries of scratch.
template <class LenPtr, class CodePtr>
SPZ_HUFF_HD void gen_codes(LenPtr len, CodePtr code_out, int max_code, const uint16_t *bl_count, uint16_t *next_code) {
  unsigned code = 0;
  for (int bits = 1; bits <= MAX_BITS; ++bits) {
    code = (code + bl_count[bits - 1]) << 1;
    next_code[bits] = static_cast<uint16_t>(code);
  }
  for (int n = 0; n <= max_code; ++n) {
    const int l = len[n];
    if (l == 0) continue;
    code_out[n] = static_cast<uint16_t>(bit_reverse(next_code[l]++, l));
  }
}

template <class LenT>
struct TreeRef {
  uint16_t *freq;  // 2 * elems + 1 entries: the symbols' counts in, the inner nodes' sums while building
  uint16_t *dad;   // 2 * elems + 1
  LenT *len;       // 2 * elems + 1 (may be the same memory as dad)
  uint16_t *code;  // elems
  int max_code;
};

struct Work {
  uint16_t *heap;      // HEAP_SIZE
  uint8_t *depth;      // HEAP_SIZE
  uint16_t *bl_count;  // MAX_BITS + 1
  uint16_t *next_code; // MAX_BITS + 1
  int heap_len, heap_max;
  long long opt_len, static_len;
};

template <class LenT>
SPZ_HUFF_HD bool smaller(const TreeRef<LenT> &t, const Work &w, int n, int m) {
  const unsigned fn = t.freq[n], fm = t.freq[m];
  return fn < fm || (fn == fm && w.depth[n] <= w.depth[m]);
}

template <class LenT>
SPZ_HUFF_HD void pqdownheap(const TreeRef<LenT> &t, Work &w, int k) {
  const int v = w.heap[k];
  int j = k << 1;
  while (j <= w.heap_len) {
    if (j < w.heap_len && smaller(t, w, w.heap[j + 1], w.heap[j])) j++;
    if (smaller(t, w, v, w.heap[j])) break;
    w.heap[k] = w.heap[j];
    k = j;
    j <<= 1;
  }
  w.heap[k] = static_cast<uint16_t>(v);
}

// build_tree + gen_bitlen + gen_codes.  K selects the extra-bit and static-length tables, `elems` and `max_length` the
// tree's size and depth limit (trees.c's static_tree_desc); the heap always has HEAP_SIZE slots, as zlib's does.
template <Kind K, class LenT>
SPZ_HUFF_HD void build(TreeRef<LenT> &t, Work &w, int elems, int max_length) {
  constexpr bool has_static = (K != BL);
  const int base = (K == LIT) ? LITERALS + 1 : 0;
  (void)base;
  w.heap_len = 0;
  w.heap_max = HEAP_SIZE;
  int max_code = -1;
  for (int n = 0; n < elems; ++n) {
    if (t.freq[n] != 0) {
      w.heap[++w.heap_len] = static_cast<uint16_t>(max_code = n);
      w.depth[n] = 0;
    } else {
      t.len[n] = 0;
    }
  }
  while (w.heap_len < 2) {
    const int node = (max_code < 2 ? ++max_code : 0);
    w.heap[++w.heap_len] = static_cast<uint16_t>(node);
    t.freq[node] = 1;
    w.depth[node] = 0;
    w.opt_len--;
    if (has_static) w.static_len -= static_len_of<K>(node);
  }
  t.max_code = max_code;
  for (int n = w.heap_len / 2; n >= 1; --n) pqdownheap(t, w, n);
  int node = elems;
  do {
    const int n = w.heap[1];
    w.heap[1] = w.heap[w.heap_len--];
    pqdownheap(t, w, 1);
    const int m = w.heap[1];
    w.heap[--w.heap_max] = static_cast<uint16_t>(n);
    w.heap[--w.heap_max] = static_cast<uint16_t>(m);
    t.freq[node] = static_cast<uint16_t>(t.freq[n] + t.freq[m]);
    const unsigned dn = w.depth[n], dm = w.depth[m];
    w.depth[node] = static_cast<uint8_t>((dn >= dm ? dn : dm) + 1);
    t.dad[n] = t.dad[m] = static_cast<uint16_t>(node);
    w.heap[1] = static_cast<uint16_t>(node++);
    pqdownheap(t, w, 1);
  } while (w.heap_len >= 2);
  w.heap[--w.heap_max] = w.heap[1];
  // gen_bitlen
  for (int bits = 0; bits <= MAX_BITS; ++bits) w.bl_count[bits] = 0;
  int overflow = 0, h;
  t.len[w.heap[w.heap_max]] = 0;
  for (h = w.heap_max + 1; h < HEAP_SIZE; ++h) {
    const int n = w.heap[h];
    int bits = static_cast<int>(t.len[t.dad[n]]) + 1;
    if (bits > max_length) bits = max_length, overflow++;
    t.len[n] = static_cast<LenT>(bits);
    if (n > max_code) continue;
    w.bl_count[bits]++;
    const int xbits = extra_bits_of<K>(n);
    const long long f = t.freq[n];
    w.opt_len += f * (bits + xbits);
    if (has_static) w.static_len += f * (static_len_of<K>(n) + xbits);
  }
  if (overflow != 0) {
    do {
      int bits = max_length - 1;
      while (w.bl_count[bits] == 0) bits--;
      w.bl_count[bits]--;
      w.bl_count[bits + 1] += 2;
      w.bl_count[max_length]--;
      overflow -= 2;
    } while (overflow > 0);
    for (int bits = max_length; bits != 0; --bits) {
      int n = w.bl_count[bits];
      while (n != 0) {
        const int m = w.heap[--h];
        if (m > max_code) continue;
        if (static_cast<int>(t.len[m]) != bits) {
          w.opt_len += (static_cast<long long>(bits) - static_cast<long long>(t.len[m])) * t.freq[m];
          t.len[m] = static_cast<LenT>(bits);
        }
        n--;
      }
    }
  }
  gen_codes(t.len, t.code, max_code, w.bl_count, w.next_code);
}

// scan_tree + send_tree share this walk; `emit(code, extra_value, extra_bits)` is called per bl symbol.
template <class LenPtr, class F>
SPZ_HUFF_HD void walk_lengths(LenPtr len, int max_code, F emit) {
  int prevlen = -1, nextlen = len[0], count = 0, max_count = 7, min_count = 4;
  if (nextlen == 0) max_count = 138, min_count = 3;
  for (int n = 0; n <= max_code; ++n) {
    const int curlen = nextlen;
    nextlen = (n == max_code) ? 0xffff : static_cast<int>(len[n + 1]);  // the guard zlib stores at tree[max_code + 1]
    if (++count < max_count && curlen == nextlen) continue;
    if (count < min_count) {
      for (int i = 0; i < count; ++i) emit(curlen, 0, 0);
    } else if (curlen != 0) {
      if (curlen != prevlen) {
        emit(curlen, 0, 0);
        count--;
      }
      emit(REP_3_6, count - 3, 2);
    } else if (count <= 10) {
      emit(REPZ_3_10, count - 3, 3);
    } else {
      emit(REPZ_11_138, count - 11, 7);
    }
    count = 0;
    prevlen = curlen;
    if (nextlen == 0) max_count = 138, min_count = 3;
    else if (curlen == nextlen) max_count = 6, min_count = 3;
    else max_count = 7, min_count = 4;
  }
}

// The first half of _tr_flush_block from the tallied frequencies (lt.freq[0 .. 285] without END_BLOCK, dt.freq[0 .. 29]):
// the three trees, opt_len / static_len (in w) and max_blindex (returned).  llen / dlen: where the finished literal and
// distance lengths are read from for the run-length walk (the trees' own len arrays, or wherever `publish` put them).
template <class LenT, class LenPtrL, class LenPtrD>
SPZ_HUFF_HD int plan_trees(TreeRef<LenT> &lt, TreeRef<LenT> &dt, TreeRef<LenT> &bt, Work &w, LenPtrL llen, LenPtrD dlen) {
  w.opt_len = 0;
  w.static_len = 0;
  lt.freq[END_BLOCK] = 1;  // init_block
  build<LIT>(lt, w, L_CODES, MAX_BITS);
  build<DIST>(dt, w, D_CODES, MAX_BITS);
  // build_bl_tree
  for (int i = 0; i < BL_CODES; ++i) bt.freq[i] = 0;
  uint16_t *bf = bt.freq;
  auto count = [bf](int code, int, int) { bf[code]++; };
  walk_lengths(llen, lt.max_code, count);
  walk_lengths(dlen, dt.max_code, count);
  build<BL>(bt, w, BL_CODES, MAX_BL_BITS);
  int max_blindex = BL_CODES - 1;
  for (; max_blindex >= 3; --max_blindex) {
    if (bt.len[bl_order(max_blindex)] != 0) break;
  }
  w.opt_len += 3 * (static_cast<long long>(max_blindex) + 1) + 5 + 5 + 4;
  return max_blindex;
}

}  // namespace huff
}  // namespace spz
