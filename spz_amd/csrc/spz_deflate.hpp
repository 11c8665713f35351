// spz_deflate.hpp — a multi-threaded gzip writer whose output is BYTE-IDENTICAL to zlib's single stream.
// Internal to libspz_host.so (see spz_deflate.cpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace spz {
namespace exactgz {

// gzip member equal, byte for byte, to what zlib 1.2.11 writes for
//   deflateInit2(Z_DEFAULT_COMPRESSION, Z_DEFLATED, 16 + MAX_WBITS, /*memLevel*/ 9, Z_DEFAULT_STRATEGY)
// fed the whole input and finished with Z_FINISH (the reference's compressGzipped, load-spz.cc:186-214).
// `windows_per_chunk` = 32 KiB windows of input per parse job (>= 4).  Returns false when the input is
// outside what this writer handles (size < 128 KiB or >= 4 GiB - 64 KiB, threads < 1) or when two
// neighbouring parse jobs did not re-synchronise inside their overlap (the caller then uses zlib).
// verify_prefix > 0: as a last step the first `verify_prefix` input bytes are deflated with the linked
// zlib and the output bytes that cannot depend on where that prefix ends are compared; false on a mismatch.
bool compress(const uint8_t *data, size_t size, int threads, int windows_per_chunk, std::vector<uint8_t> *out,
              size_t verify_prefix = 0);

}  // namespace exactgz
}  // namespace spz
