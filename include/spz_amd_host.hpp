// spz_amd_host.hpp — C++ drop-in layer over the C ABI (include/spz_amd.h).
//
// Keeps the public C++ surface of lanxinger/spz for the save/load path so that
// code written against the reference compiles against this header unchanged:
//   namespace spz, GaussianCloud / PackedGaussians / PackOptions / UnpackOptions /
//   CoordinateSystem (reference: src/cc/splat-types.h:24-34,90-186, src/cc/load-spz.h:42-67),
//   saveSpz / loadSpz / loadSpzPacked / serializePackedGaussians / compressGzipped
//   (reference: src/cc/load-spz.h:69-100) with the same argument meaning, ownership
//   (everything by value / caller-owned vectors) and error behaviour (never throws; save ->
//   false, load -> default-constructed cloud plus one "[SPZ ERROR] ..." line on stdout).
//
// The per-Gaussian quantise / dequantise work runs on the GPU through libspz_amd.so; gzip
// stays on the host (zlib, same deflate parameters as load-spz.cc:186-214, so the .spz bytes
// are identical).  There is no CPU fallback: without a usable HIP device saveSpz returns
// false and loadSpz returns an empty cloud, each after logging
// "[SPZ ERROR] spz_amd: <status>".  The device used is $SPZ_AMD_DEVICE (default 0).
#pragma once

#include <array>
#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

#include "spz_amd_c_types.h"

namespace spz {

// splat-types.h:24-34
enum class CoordinateSystem {
  UNSPECIFIED = 0,
  LDB = 1,
  RDB = 2,
  LUB = 3,
  RUB = 4,
  LDF = 5,
  RDF = 6,
  LUF = 7,
  RUF = 8,
};

// splat-types.h:36-41: the three sign tables of a conversion.
struct CoordinateConverter {
  std::array<float, 3> flipP = {1.0f, 1.0f, 1.0f};
  std::array<float, 3> flipQ = {1.0f, 1.0f, 1.0f};
  std::array<float, 15> flipSh = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f,
                                  1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f};
};

// splat-types.h:43-81
CoordinateConverter coordinateConverter(CoordinateSystem from, CoordinateSystem to);

// splat-types.h:90-186.  Same field names and layout semantics as the reference struct.
struct GaussianCloud {
  int32_t numPoints = 0;
  int32_t shDegree = 0;
  bool antialiased = false;
  std::vector<float> positions;  // xyz
  std::vector<float> scales;     // log scale xyz
  std::vector<float> rotations;  // xyzw
  std::vector<float> alphas;     // pre-sigmoid
  std::vector<float> colors;     // SH DC rgb
  std::vector<float> sh;         // [point][coeff][rgb]

  // In-place flip between coordinate systems (splat-types.h:134-164); runs the GPU flip pass.
  void convertCoordinates(CoordinateSystem from, CoordinateSystem to);
  void rotate180DegAboutX() { convertCoordinates(CoordinateSystem::RUB, CoordinateSystem::RDF); }
  // splat-types.h:170-185 (host utility, not on the hot path).
  float medianVolume() const;
  // splat-types.h:117-130: copies of the six arrays in new float[] buffers the CALLER frees.
  GaussianCloudData data() const;
};

// load-spz.h:11-22: one inflated Gaussian (236 bytes).
struct UnpackedGaussian {
  std::array<float, 3> position;  // x, y, z
  std::array<float, 4> rotation;  // x, y, z, w
  std::array<float, 3> scale;     // log scale
  std::array<float, 3> color;     // rgb sh0 encoding
  float alpha;                    // inverse logistic
  std::array<float, 15> shR;
  std::array<float, 15> shG;
  std::array<float, 15> shB;
};

// load-spz.h:24-38: one packed Gaussian, always 65 bytes (missing sh coefficients are 128 = 0.0).
struct PackedGaussian {
  std::array<uint8_t, 9> position{};
  std::array<uint8_t, 4> rotation{};
  std::array<uint8_t, 3> scale{};
  std::array<uint8_t, 3> color{};
  uint8_t alpha = 0;
  std::array<uint8_t, 15> shR{};
  std::array<uint8_t, 15> shG{};
  std::array<uint8_t, 15> shB{};

  // load-spz.cc:383-431.  Runs the decode kernel on this one point (a 1-point stream through
  // spz_amd_decode_host): bit-identical to the reference, but a device round trip per call — use
  // unpackIndices / spz_amd_decode_gather_* for more than a handful of points.  `c` must be a table
  // that coordinateConverter() can produce (every from/to pair is; hand-edited tables are rejected
  // with a log line and a zeroed result).
  UnpackedGaussian unpack(bool usesFloat16, bool usesQuaternionSmallestThree, int32_t fractionalBits,
                          const CoordinateConverter &c) const;
};

// load-spz.h:42-59.
struct PackedGaussians {
  int32_t numPoints = 0;
  int32_t shDegree = 0;
  int32_t fractionalBits = 0;
  bool antialiased = false;
  bool usesQuaternionSmallestThree = true;
  std::vector<uint8_t> positions;
  std::vector<uint8_t> scales;
  std::vector<uint8_t> rotations;
  std::vector<uint8_t> alphas;
  std::vector<uint8_t> colors;
  std::vector<uint8_t> sh;

  bool usesFloat16() const;  // load-spz.cc:465
  PackedGaussian at(int32_t i) const;                                         // load-spz.cc:433-459 (bytes only)
  UnpackedGaussian unpack(int32_t i, const CoordinateConverter &c) const;     // load-spz.cc:461-463
};

// load-spz.h:61-67
struct PackOptions {
  CoordinateSystem from = CoordinateSystem::UNSPECIFIED;
};
struct UnpackOptions {
  CoordinateSystem to = CoordinateSystem::UNSPECIFIED;
};

// load-spz.h:69-100 -------------------------------------------------------------------------
bool saveSpz(const GaussianCloud &gaussians, const PackOptions &options, std::vector<uint8_t> *output);
bool saveSpz(const GaussianCloud &gaussians, const PackOptions &options, const std::string &filename);
GaussianCloud loadSpz(const std::vector<uint8_t> &data, const UnpackOptions &options);
GaussianCloud loadSpz(const uint8_t *data, int32_t size, const UnpackOptions &options);
GaussianCloud loadSpz(const std::string &filename, const UnpackOptions &options);
PackedGaussians loadSpzPacked(const std::string &filename);
PackedGaussians loadSpzPacked(const uint8_t *data, int32_t size);
PackedGaussians loadSpzPacked(const std::vector<uint8_t> &data);
// .ply pair (load-spz.cc:691-934): binary little-endian 3DGS layout, RDF on disk.
bool saveSplatToPly(const GaussianCloud &gaussians, const PackOptions &options, const std::string &filename);
GaussianCloud loadSplatFromPly(const std::string &filename, const UnpackOptions &options);
void serializePackedGaussians(const PackedGaussians &packed, std::ostream *out);
// load-spz.cc:186-214.  The bytes are zlib's (level 6, one deflate stream, gzip wrapper): for inputs of
// 1 MiB and more they are produced by a multi-threaded writer that reproduces zlib 1.2.11's output exactly
// (SPZ_AMD_GZIP_EXACT_THREADS, default min(usable CPUs, 32); 1 = zlib itself), otherwise by zlib — and by zlib
// too when the machine has less free memory than 4.5 x the input (the writer holds ~3 bytes per input byte
// until it assembles the member).  SPZ_AMD_GZIP_VERIFY=1 inflates the finished member and compares it with the
// input before returning it, =2 also compares it byte for byte with zlib's own member; a failed check returns zlib's.
bool compressGzipped(const uint8_t *data, size_t size, std::vector<uint8_t> *out);

// External-linkage internals of the reference (load-spz.cc:257,467,548), kept because
// downstream code forward-declares them to skip gzip.
PackedGaussians packGaussians(const GaussianCloud &g, const PackOptions &o);
GaussianCloud unpackGaussians(const PackedGaussians &packed, const UnpackOptions &o);
PackedGaussians deserializePackedGaussians(std::istream &in);

// Extras of this implementation -------------------------------------------------------------
// Inverse of compressGzipped (the reference keeps it file-local, load-spz.cc:141-182).  Ordinary members of
// 4 MiB and more are inflated in parallel and verified by CRC-32 (from 8 threads up).  Members written
// by compressGzippedParallel are inflated piece-parallel (SPZ_AMD_GUNZIP_THREADS, default min(cores, 32));
// other members go through libdeflate when the system has libdeflate.so.0 (SPZ_AMD_NO_LIBDEFLATE=1 turns
// that off); zlib's streaming inflate, what the reference uses, is the fallback and decides every case
// the fast readers decline, so acceptance and output are zlib's.
bool decompressGzipped(const uint8_t *compressed, size_t size, std::vector<uint8_t> *out);
// Opt-in multi-threaded gzip (pigz's independent-blocks construction: 1 MiB raw-deflate pieces in one
// gzip member, their sizes listed in an FEXTRA subfield "SZ" that other readers skip).  Readable by
// every gzip reader including the reference's loadSpz, NOT byte-identical to compressGzipped.  saveSpz uses it when the environment sets SPZ_AMD_GZIP_THREADS > 1.
bool compressGzippedParallel(const uint8_t *data, size_t size, std::vector<uint8_t> *out, int threads);
// Bulk random access: the points `indices` of a packed cloud, decoded by one gather
// launch into a GaussianCloud of indices.size() points (same arithmetic as unpackGaussians).  An index
// past the end is an error: empty cloud + log line, lastDeviceStatus() = SPZ_AMD_ERR_INVALID_ARG (the
// device-pointer form spz_amd_decode_gather_device clamps instead, as its header says).  Empty cloud on failure.
GaussianCloud unpackIndices(const PackedGaussians &packed, const std::vector<uint32_t> &indices,
                            const UnpackOptions &o);
// Raw (pre-gzip) stream <-> cloud, i.e. saveSpz / loadSpz without the zlib step.
bool packToStream(const GaussianCloud &g, const PackOptions &o, std::vector<uint8_t> *stream);
GaussianCloud unpackFromStream(const uint8_t *stream, size_t size, const UnpackOptions &o);
// CPUs this process may use (online count, affinity mask, cgroup quota): what the thread-count defaults of the
// gzip writer / readers derive from.
unsigned effectiveCpuCount();
// Members compressGzipped has written in this process with their LZ77 parse done on the device (spz_lz77.hip;
// environment SPZ_AMD_GZIP_DEVICE = 0 never, 1 always, unset: inputs of 8 MiB and more).  The bytes are zlib's
// whichever way the parse ran; this says which way it was.
uint64_t deviceGzipParseCount();
// Members of the device writer that failed one of its checks (symbols that do not reproduce the input; with
// SPZ_AMD_GZIP_VERIFY=1 a member that does not inflate back to it) and were discarded: the caller got the host writer's
// bytes instead, and a "[SPZ ERROR] spz_amd: the device gzip writer ..." line.  Anything but 0 is a defect to report.
uint64_t deviceGzipRejectCount();
// Members decompressGzipped has inflated on the device in this process (spz_inflate_dev.hip; SPZ_AMD_GUNZIP_DEVICE = 0
// never, 1 from 1 MiB, unset: from 8 MiB); believed only after the CRC-32 and ISIZE of the trailer matched.
uint64_t deviceInflateCount();
// saveSpz keeps its transient stream buffer (65 bytes per Gaussian, at most 1 GiB) for the next save, because returning
// memory that device copies have pinned costs ~80 ms per GB; this drops it.  (The device side: spz_amd_release_device_memory().)
void releaseHostMemory();
// Status (spz_amd.h codes) of the last device call made by this thread; 0 = ok.
int lastDeviceStatus();
void setLastDeviceStatus(int status);

}  // namespace spz
