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
#include "spz_inflate_core.hpp"

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
  inline void mark() {}
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

// First bit position in [lo, hi) where a non-final dynamic block starts, decodes to its end-of-block and is
// followed by a plausible header.
uint64_t findBlockStart(const Bits &in, uint64_t lo, uint64_t hi) {
  HuffLit lit;
  HuffDist dist;
  for (uint64_t p = lo; p < hi && p + 64 < in.nbits; ++p) {
    if (isBlockStart(in, p, &lit, &dist)) return p;
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
    HuffLit lit_tables;
    HuffDist dist_tables;
    const Outcome r = decodeBlocks(in, from, to, sink, &end, &lit_tables, &dist_tables);
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
