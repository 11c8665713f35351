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
  static bool declined(int where) {
    if (std::getenv("SPZ_LZ_MODEL_TRACE")) std::fprintf(stderr, "[model] declined at check %d\n", where);
    return false;
  }
  std::vector<uint16_t> dist;
  std::vector<uint8_t> lc;

  // Stage 3 for one job size: every job parses its own range and records its lazy state at every loop top, then
  // continues into its successor's range until the states agree.  false: some pair of neighbours did not meet.
  template <class A128, class A32>
  bool run_jobs(uint32_t job_bytes, uint64_t tail_begin, const uint32_t *tail_rec, const A128 &a128, const A32 &a32,
                uint32_t *tail_lo) {
    const uint64_t njobs = (tail_begin + job_bytes - 1) / job_bytes;
    std::vector<uint32_t> rec(tail_begin + kTailWindow, 0);  // per position; the tail job's first loop tops at the end
    for (uint32_t k = 0; k < kTailWindow; ++k) rec[tail_begin + k] = tail_rec[2 * k];
    std::vector<std::vector<uint16_t>> jd(njobs), xd(njobs);
    std::vector<std::vector<uint8_t>> jl(njobs), xl(njobs);
    std::vector<uint64_t> end_s(njobs);
    std::vector<LazyState<uint64_t>> end_st(njobs);
    for (uint64_t j = 0; j < njobs; ++j) {
      const uint64_t begin = j * job_bytes, next = std::min<uint64_t>(begin + job_bytes, tail_begin);
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
      if (jl[j].size() > job_symbol_stride(job_bytes)) return declined(3);
      end_s[j] = s;
      end_st[j] = st;
    }
    std::vector<uint32_t> lo(njobs + 1, 0);
    for (uint64_t j = 0; j < njobs; ++j) {
      const uint64_t next = std::min<uint64_t>((j + 1) * job_bytes, tail_begin);
      const uint64_t stop = stitch_end(j, njobs, job_bytes, tail_begin);
      uint64_t s = end_s[j];
      LazyState<uint64_t> st = end_st[j];
      auto emit = [&](uint32_t d, uint32_t l) {
        xd[j].push_back(static_cast<uint16_t>(d));
        xl[j].push_back(static_cast<uint8_t>(l));
      };
      bool spliced = false;
      while (s < stop) {
        if (rec[s] == pack_state(st, s)) {
          // symbols the successor's own parse had emitted before this loop top
          uint64_t s2 = next;
          LazyState<uint64_t> st2;
          uint32_t cnt = 0;
          auto count = [&](uint32_t, uint32_t) { ++cnt; };
          while (s2 < s) lazy_step(s2, st2, a128, a32, count);
          if (s2 != s) return declined(4);
          lo[j + 1] = cnt;
          spliced = true;
          if (std::getenv("SPZ_LZ_MODEL_TRACE") && s - next > 1000) {
            std::fprintf(stderr, "[model] job %llu (of %u bytes) met at +%llu\n", (unsigned long long)j, job_bytes,
                         (unsigned long long)(s - next));
          }
          break;
        }
        lazy_step(s, st, a128, a32, emit);
      }
      if (!spliced) {
        if (std::getenv("SPZ_LZ_MODEL_TRACE")) std::fprintf(stderr, "[model] jobs of %u bytes: no meeting point after job %llu\n", job_bytes, (unsigned long long)j);
        return false;
      }
      if (j + 1 == njobs && tail_rec[2 * (s - next) + 1] != lo[njobs]) return declined(6);  // the host's own count
    }
    dist.clear();
    lc.clear();
    for (uint64_t j = 0; j < njobs; ++j) {
      if (lo[j] > jl[j].size()) return declined(7);
      dist.insert(dist.end(), jd[j].begin() + lo[j], jd[j].end());
      lc.insert(lc.end(), jl[j].begin() + lo[j], jl[j].end());
      dist.insert(dist.end(), xd[j].begin(), xd[j].end());
      lc.insert(lc.end(), xl[j].begin(), xl[j].end());
    }
    *tail_lo = lo[njobs];
    return true;
  }

  bool parse(const uint8_t *data, size_t size, uint64_t tail_begin, const uint32_t *tail_rec, uint32_t n_rec,
             uint64_t *num_symbols, uint32_t *tail_first_symbol) override {
    if (tail_begin == 0 || tail_begin % W != 0 || n_rec < kTailWindow) return declined(1);
    const uint64_t n_pos = tail_begin + kTableSlack;        // positions the tables cover
    if (n_pos + kReadAhead + MIN_LOOKAHEAD > size) return declined(2);
    // ---- stage 1: link[] along hash2, rank[] along zlib's hash
    std::vector<uint16_t> link(n_pos, 0), rank(n_pos, 0);
    {
      std::vector<uint64_t> head(size_t(1) << HASH2_BITS, ~uint64_t(0));
      std::vector<uint16_t> count(HASH_MASK + 1, 0);
      for (uint64_t p = 0; p < n_pos; ++p) {
        const uint32_t h2 = hash2(data[p], data[p + 1], data[p + 2]);
        const uint64_t q = head[h2];
        link[p] = (q != ~uint64_t(0) && p - q < W) ? static_cast<uint16_t>(p - q) : 0;
        head[h2] = p;
        rank[p] = count[hash3(data[p], data[p + 1], data[p + 2])]++;
      }
    }
    // ---- stage 1b: zlib's chain steps per link
    std::vector<uint8_t> delta(n_pos, 0);
    for (uint64_t p = 0; p < n_pos; ++p) {
      if (link[p] == 0) continue;
      const uint64_t t = p - link[p];
      delta[p] = static_cast<uint8_t>(chain_delta(rank[p], rank[t], data[p] == data[t] && data[p + 1] == data[t + 1] && data[p + 2] == data[t + 2]));
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
      uint64_t load8(int64_t pos) const {
        uint64_t v;
        std::memcpy(&v, d + pos, 8);
        return v;
      }
    } dacc{data};
    auto lacc = [&](int64_t pos) { return static_cast<uint32_t>(link[pos]); };
    auto racc = [&](int64_t pos) { return static_cast<uint32_t>(rank[pos]); };
    auto kacc = [&](int64_t pos) { return static_cast<uint32_t>(delta[pos]); };
    for (uint64_t p = 0; p < n_pos; ++p) {
      find_matches<int64_t>(dacc, lacc, kacc, racc, static_cast<int64_t>(p), static_cast<int64_t>(base_at(p, size)), &r128[p], &r32[p]);
    }
    // ---- stage 3, with larger jobs if two neighbours do not meet
    auto a128 = [&](uint64_t pos) { return r128[pos]; };
    auto a32 = [&](uint64_t pos) { return r32[pos]; };
    uint32_t tail_lo = 0;
    bool done = false;
    for (uint32_t job_bytes : kJobSizes) {
      if (run_jobs(job_bytes, tail_begin, tail_rec, a128, a32, &tail_lo)) {
        done = true;
        break;
      }
    }
    if (!done) return declined(5);
    *num_symbols = lc.size();
    *tail_first_symbol = tail_lo;
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
