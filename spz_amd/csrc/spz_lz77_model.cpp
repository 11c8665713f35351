// spz_lz77_model.cpp — serial host model of the data-parallel zlib parse (spz_lz77_core.hpp), with the SAME
// stage functions and job geometry as the HIP kernels of spz_lz77.hip.  Test infrastructure: it lets the CPU
// suite prove that links -> match tables -> lazy state machine -> record-window splice reproduces zlib's
// symbols before a GPU is involved (tests/test_exact_gzip.py); nothing in the product path calls it.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "spz_deflate.hpp"
#include "spz_lz77_core.hpp"

namespace spz {
namespace exactgz {
namespace {

using namespace spz_lz;

struct ModelParser final : HeadParser {
  std::vector<uint16_t> dist;
  std::vector<uint8_t> lc;

  bool parse(const uint8_t *data, size_t size, uint64_t tail_begin, const uint32_t *tail_rec, uint32_t n_rec,
             uint64_t *num_symbols, uint32_t *tail_first_symbol) override {
    if (tail_begin == 0 || tail_begin % W != 0 || n_rec < kRecordWindow) return false;
    const uint64_t n_pos = tail_begin + kTableSlack;        // positions the tables cover
    if (n_pos + kReadAhead + MIN_LOOKAHEAD > size) return false;
    // ---- stage 1: links (zlib's head[] / prev[] as one pass with a 32-bit head table)
    std::vector<uint16_t> link(n_pos, 0);
    {
      std::vector<uint64_t> head(HASH_MASK + 1, ~uint64_t(0));
      for (uint64_t p = 0; p < n_pos; ++p) {
        const uint32_t h = hash3(data[p], data[p + 1], data[p + 2]);
        const uint64_t q = head[h];
        link[p] = (q != ~uint64_t(0) && p - q < W) ? static_cast<uint16_t>(p - q) : 0;
        head[h] = p;
      }
    }
    // ---- stage 2: match tables
    std::vector<uint32_t> r128(n_pos), r32(n_pos);
    struct Data {
      const uint8_t *d;
      uint32_t load4(int64_t pos) const {
        uint32_t v;
        std::memcpy(&v, d + pos, 4);
        return v;
      }
    } dacc{data};
    auto lacc = [&](int64_t pos) { return static_cast<uint32_t>(link[pos]); };
    for (uint64_t p = 0; p < n_pos; ++p) {
      find_matches<int64_t>(dacc, lacc, static_cast<int64_t>(p), static_cast<int64_t>(base_at(p, size)), &r128[p], &r32[p]);
    }
    // ---- stage 3: every job parses its own range and records its lazy state at every loop top
    const uint64_t njobs = tail_begin / kJobBytes;  // tail_begin is a multiple of W = 2 jobs
    std::vector<uint32_t> rec(tail_begin + kRecordWindow, 0);  // per position; the tail job's first loop tops at the end
    for (uint32_t k = 0; k < kRecordWindow; ++k) rec[tail_begin + k] = tail_rec[2 * k];
    auto a128 = [&](uint64_t pos) { return r128[pos]; };
    auto a32 = [&](uint64_t pos) { return r32[pos]; };
    std::vector<std::vector<uint16_t>> jd(njobs), xd(njobs);
    std::vector<std::vector<uint8_t>> jl(njobs), xl(njobs);
    std::vector<uint64_t> end_s(njobs);
    std::vector<LazyState<uint64_t>> end_st(njobs);
    for (uint64_t j = 0; j < njobs; ++j) {
      const uint64_t begin = j * kJobBytes, next = begin + kJobBytes;
      uint64_t s = begin;
      LazyState<uint64_t> st;
      auto emit = [&](uint32_t d, uint32_t l) {
        jd[j].push_back(static_cast<uint16_t>(d));
        jl[j].push_back(static_cast<uint8_t>(l));
      };
      while (s < next) {
        rec[s] = pack_state(st, s);
        lazy_step(s, st, a128, a32, emit);
      }
      if (jl[j].size() > kJobSymbolStride) return false;
      end_s[j] = s;
      end_st[j] = st;
    }
    // ---- stitch: every job continues into its successor's range until the states agree
    std::vector<uint32_t> lo(njobs + 1, 0);
    for (uint64_t j = 0; j < njobs; ++j) {
      const uint64_t next = (j + 1) * kJobBytes;
      uint64_t s = end_s[j];
      LazyState<uint64_t> st = end_st[j];
      auto emit = [&](uint32_t d, uint32_t l) {
        xd[j].push_back(static_cast<uint16_t>(d));
        xl[j].push_back(static_cast<uint8_t>(l));
      };
      bool spliced = false;
      while (s - next < kRecordWindow) {
        if (rec[s] == pack_state(st, s)) {
          // symbols the successor's own parse had emitted before this loop top
          uint64_t s2 = next;
          LazyState<uint64_t> st2;
          uint32_t cnt = 0;
          auto count = [&](uint32_t, uint32_t) { ++cnt; };
          while (s2 < s) lazy_step(s2, st2, a128, a32, count);
          if (s2 != s) return false;
          lo[j + 1] = cnt;
          spliced = true;
          break;
        }
        lazy_step(s, st, a128, a32, emit);
      }
      if (!spliced) return false;
      if (j + 1 == njobs && tail_rec[2 * (s - next) + 1] != lo[njobs]) return false;  // the host's own count
    }
    dist.clear();
    lc.clear();
    for (uint64_t j = 0; j < njobs; ++j) {
      if (lo[j] > jl[j].size()) return false;
      dist.insert(dist.end(), jd[j].begin() + lo[j], jd[j].end());
      lc.insert(lc.end(), jl[j].begin() + lo[j], jl[j].end());
      dist.insert(dist.end(), xd[j].begin(), xd[j].end());
      lc.insert(lc.end(), xl[j].begin(), xl[j].end());
    }
    *num_symbols = lc.size();
    *tail_first_symbol = lo[njobs];
    return true;
  }

  bool fetch(uint16_t *d, uint8_t *l) override {
    std::memcpy(d, dist.data(), dist.size() * sizeof(uint16_t));
    std::memcpy(l, lc.data(), lc.size());
    return true;
  }
};

}  // namespace

HeadParser *newModelHeadParser() { return new ModelParser(); }

}  // namespace exactgz
}  // namespace spz
