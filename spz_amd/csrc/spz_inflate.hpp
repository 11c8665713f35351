// spz_inflate.hpp — multi-threaded inflate of ONE ordinary deflate stream (a gzip member written by zlib,
// e.g. every .spz file the reference writes).  Internal to libspz_host.so (see spz_inflate.cpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace spz {
namespace pinflate {

// `gz` is a complete gzip member of `size` bytes whose header is `header_len` bytes long; the deflate
// data must end exactly 8 bytes (CRC-32, ISIZE) before the end.  On success *out holds the inflated bytes,
// verified against the member's CRC-32 and ISIZE.  Returns false — leaving the decision to a serial
// reader — for anything else: too small to split, block starts that cannot be found or do not link up,
// invalid data, trailing bytes, a CRC mismatch.
bool inflate(const uint8_t *gz, size_t size, size_t header_len, int threads, std::vector<uint8_t> *out);

// Number of members this process has inflated through the parallel reader (for tests and diagnostics).
uint64_t successCount();

}  // namespace pinflate
}  // namespace spz
