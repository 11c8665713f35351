// spz_deflate.hpp — a multi-threaded gzip writer whose output is BYTE-IDENTICAL to zlib's single stream.
// Internal to libspz_host.so (see spz_deflate.cpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "spz_amd.h"

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

// The same member with the LZ77 parse of everything but the last 64-96 KiB done by `parser` — the MI355X
// (spz_lz77.hip behind the C ABI's spz_amd_zlib_parse_*) or, in tests, the serial host model of the same
// three stages (spz_lz77_model.cpp).  The end of the input, where zlib's lookahead runs out, stays with the
// serial job of spz_deflate.cpp; the two symbol streams are spliced where their lazy-match states agree.
struct HeadParser {
  virtual ~HeadParser() = default;
  // Parses the loop tops before the splice.  `tail_rec`: n_rec pairs {state, symbol count} the tail job recorded
  // at its loop tops tail_begin + k (state 0: not a loop top).  Out: the number of head symbols and the index of the
  // tail job's first contributed symbol.  false = declined (the caller parses on the host).
  virtual bool parse(const uint8_t *data, size_t size, uint64_t tail_begin, const uint32_t *tail_rec, uint32_t n_rec,
                     uint64_t *num_symbols, uint32_t *tail_first_symbol) = 0;
  // The same for a caller whose tail records do not exist yet: produce(arg) fills tail_rec and is called by the
  // parser when it has work of its own under way that does not need them (the device: its table and match kernels).
  virtual bool parseLate(const uint8_t *data, size_t size, uint64_t tail_begin, const uint32_t *tail_rec, uint32_t n_rec,
                         void (*produce)(void *), void *arg, uint64_t *num_symbols, uint32_t *tail_first_symbol) {
    produce(arg);
    return parse(data, size, tail_begin, tail_rec, n_rec, num_symbols, tail_first_symbol);
  }
  // Copies the head symbols out: distance (0 = literal) and literal byte / match length - 3.
  virtual bool fetch(uint16_t *dist, uint8_t *lc) = 0;
  // Optional: the Huffman stage where the symbols already are (include/spz_amd.h: spz_amd_zlib_parse_append /
  // _block_stats / _encode_group / _encode_finish have the same meaning).  canFinish() false: the caller fetches the symbols and
  // does all of it on the host.
  virtual bool canFinish() const { return false; }
  virtual bool append(const uint16_t *, const uint8_t *, size_t) { return false; }
  virtual bool blockStats(const spz_amd_deflate_static &, uint32_t, uint32_t, uint16_t *, uint16_t *, uint32_t *,
                          uint32_t *) {
    return false;
  }
  // encodeGroup: blocks [first, first + n) of `total`, enqueued (the caller goes on to the next group's trees);
  // encodeFinish waits for all of them and copies the body and the blocks' bit counts out.
  virtual bool encodeGroup(const spz_amd_deflate_static &, uint32_t /*block_syms*/, uint32_t /*total*/, uint32_t /*first*/,
                           uint32_t /*n*/, const spz_amd_deflate_block *, const spz_amd_deflate_codes *, const uint32_t *,
                           uint64_t /*nwords*/, uint64_t /*body_bytes_bound*/) {
    return false;
  }
  virtual bool encodeFinish(uint32_t /*total*/, uint64_t /*body_bytes*/, uint8_t *, uint64_t * /*symbol_bits*/,
                            uint32_t * /*header_bits, may be null*/) {
    return false;
  }
  // Optional: the trees there as well (spz_amd_zlib_block_trees / _encode_planned); blockStats is then called with
  // null frequency arrays.
  virtual bool canBuildTrees() const { return false; }
  virtual bool blockTrees(uint32_t /*total*/, spz_amd_deflate_plan *) { return false; }
  virtual bool encodePlanned(const spz_amd_deflate_static &, uint32_t /*block_syms*/, uint32_t /*total*/,
                             const spz_amd_deflate_block *, uint64_t /*body_bytes*/) {  // enqueued; then encodeFinish
    return false;
  }
};
// The writer's serial tail job (the last 64 ... 96 KiB of the input, 6 - 9 ms of one core on .spz data) started ahead
// of the call, for a caller whose input's END exists before the rest does (saveSpz: the stream's last bytes come
// with the first sh piece on purpose).  tailAheadBytes(size): how many bytes at the end of data[0, size) must be final
// when tailAheadStart is called (0: no tail job for this size); the job runs on a thread of its own and is consumed by
// compressWithHeadParser(..., ahead) — whatever that returns — or by tailAheadDrop.
struct TailAhead;
size_t tailAheadBytes(size_t size);
TailAhead *tailAheadStart(const uint8_t *data, size_t size);
void tailAheadDrop(TailAhead *ahead);
bool compressWithHeadParser(const uint8_t *data, size_t size, int threads, HeadParser &parser,
                            std::vector<uint8_t> *out, size_t verify_prefix = 0, TailAhead *ahead = nullptr);

// Serial host model of the data-parallel parse (stages and job geometry of spz_lz77_core.hpp), for tests.
HeadParser *newModelHeadParser();

}  // namespace exactgz
}  // namespace spz
