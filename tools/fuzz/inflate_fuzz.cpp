// Sanitizer harness for the parallel single-stream inflater: a 7 MB member, mutated (bit flips in the
// deflate data and in the trailer, truncation, splices of foreign bytes); whenever the parallel reader
// accepts, zlib must accept too and the bytes must be equal.  Built with ASan + UBSan by `make fuzz`.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../spz_amd/csrc/spz_inflate.hpp"

static bool zlibInflate(const std::vector<uint8_t> &gz, std::vector<uint8_t> *out) {
  z_stream s = {};
  if (inflateInit2(&s, 16 + MAX_WBITS) != Z_OK) return false;
  out->assign(40u << 20, 0);
  s.next_in = const_cast<Bytef *>(gz.data());
  s.avail_in = static_cast<uInt>(gz.size());
  s.next_out = out->data();
  s.avail_out = static_cast<uInt>(out->size());
  const int rc = inflate(&s, Z_FINISH);
  out->resize(s.total_out);
  inflateEnd(&s);
  return rc == Z_STREAM_END;
}

int main() {
  std::mt19937 rng(31);
  std::vector<uint8_t> raw(12u << 20);
  for (size_t i = 0; i < raw.size(); ++i) {
    const size_t sec = i / (raw.size() / 4);
    raw[i] = sec == 1 ? static_cast<uint8_t>(rng()) : static_cast<uint8_t>(128 + 8 * (static_cast<int>(rng() % 11) - 5));
  }
  std::vector<uint8_t> gz(compressBound(raw.size()) + 64);
  z_stream s = {};
  deflateInit2(&s, 6, Z_DEFLATED, 16 + MAX_WBITS, 9, Z_DEFAULT_STRATEGY);
  s.next_in = raw.data(); s.avail_in = static_cast<uInt>(raw.size());
  s.next_out = gz.data(); s.avail_out = static_cast<uInt>(gz.size());
  deflate(&s, Z_FINISH);
  gz.resize(s.total_out);
  deflateEnd(&s);
  int accepted = 0, declined = 0;
  for (int it = 0; it < 120; ++it) {
    std::vector<uint8_t> m = gz;
    const int kind = it == 0 ? -1 : it % 5;
    if (kind == 0) for (int k = 0; k < 1 + (int)(rng() % 3); ++k) m[10 + rng() % (m.size() - 18)] ^= 1u << (rng() % 8);
    else if (kind == 1) m.resize(m.size() - 1 - rng() % 5000);
    else if (kind == 2) m[m.size() - 1 - rng() % 8] ^= 1u << (rng() % 8);
    else if (kind == 3) { size_t a = 10 + rng() % (m.size() - 4000); for (int k = 0; k < 2000; ++k) m[a + k] = static_cast<uint8_t>(rng()); }
    else if (kind == 4) { size_t a = 10 + rng() % (m.size() - 100000), b = 10 + rng() % (m.size() - 100000); std::memmove(m.data() + a, m.data() + b, 50000); }
    std::vector<uint8_t> got, want;
    const bool ok = spz::pinflate::inflate(m.data(), m.size(), 10, 4, &got);
    if (ok) {
      ++accepted;
      if (!zlibInflate(m, &want) || want != got) {
        std::printf("inflate_fuzz: accepted what zlib does not, it=%d kind=%d\n", it, kind);
        return 1;
      }
    } else {
      ++declined;
    }
  }
  std::printf("inflate_fuzz: %d accepted (all equal to zlib), %d declined\n", accepted, declined);
  return accepted >= 1 ? 0 : 1;
}
