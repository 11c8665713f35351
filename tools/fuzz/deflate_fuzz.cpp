// Sanitizer harness for the byte-identical gzip writer: random inputs of several textures and sizes,
// several chunkings; every output must equal zlib's.  Built with -fsanitize=address,undefined by `make fuzz`.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../spz_amd/csrc/spz_deflate.hpp"

static std::vector<uint8_t> zlibGzip(const std::vector<uint8_t> &d) {
  std::vector<uint8_t> z(compressBound(d.size()) + 64);
  z_stream s = {};
  deflateInit2(&s, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 16 + MAX_WBITS, 9, Z_DEFAULT_STRATEGY);
  s.next_in = const_cast<Bytef *>(d.data());
  s.avail_in = static_cast<uInt>(d.size());
  s.next_out = z.data();
  s.avail_out = static_cast<uInt>(z.size());
  deflate(&s, Z_FINISH);
  z.resize(s.total_out);
  deflateEnd(&s);
  return z;
}

int main() {
  std::mt19937 rng(99);
  int cases = 0, declined = 0;
  for (int it = 0; it < 60; ++it) {
    const size_t n = 131072 + rng() % 900000;
    std::vector<uint8_t> d(n);
    const int kind = it % 5;
    size_t i = 0;
    while (i < n) {
      if (kind == 0) d[i++] = rng() & 15;
      else if (kind == 1) d[i++] = static_cast<uint8_t>(rng());
      else if (kind == 2) { size_t run = 1 + rng() % 3000; uint8_t v = rng() & 3; while (run-- && i < n) d[i++] = v; }
      else if (kind == 3) d[i++] = static_cast<uint8_t>(128 + 8 * (static_cast<int>(rng() % 9) - 4));
      else { size_t back = 1 + rng() % 40000, len = 3 + rng() % 400; for (size_t k = 0; k < len && i < n; ++k, ++i) d[i] = i >= back ? d[i - back] : static_cast<uint8_t>(rng()); if (rng() % 3 == 0 && i < n) d[i++] = static_cast<uint8_t>(rng()); }
    }
    const std::vector<uint8_t> want = zlibGzip(d);
    for (int windows : {4, 6, 4096}) {
      std::vector<uint8_t> got;
      const bool ok = spz::exactgz::compress(d.data(), d.size(), 3, windows, &got, it % 2 ? 200000 : 0);
      ++cases;
      if (!ok) { ++declined; continue; }
      if (got != want) {
        std::printf("MISMATCH it=%d kind=%d n=%zu windows=%d\n", it, kind, n, windows);
        return 1;
      }
    }
  }
  std::printf("deflate_fuzz: %d cases equal to zlib, %d declined\n", cases, declined);
  return 0;
}
