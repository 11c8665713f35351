// spz_host.cpp — C++ drop-in layer (namespace spz) over the C ABI of libspz_amd.so.
//
// Host-side mirror of the reference's save/load plumbing (/root/reference/src/cc/load-spz.cc
// :141-214 gzip, :533-596 (de)serialise, :598-668 saveSpz/loadSpz overloads), re-plumbed so
// that the per-Gaussian work goes through spz_amd_encode_host / spz_amd_decode_host and the
// stream is written/read in place (no stringstream double copy).
#include "spz_amd_host.hpp"

#include <dlfcn.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <cerrno>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <istream>
#include <mutex>
#include <ostream>
#include <thread>

#include "spz_amd.h"
#include "spz_deflate.hpp"
#include "spz_host_util.hpp"
#include "spz_inflate.hpp"

namespace spz {
namespace {

thread_local int g_last_status = SPZ_AMD_OK;

// SpzLog, load-spz.cc:29-35: printf to stdout + newline + flush.
void logLine(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vprintf(fmt, ap);
  va_end(ap);
  std::printf("\n");
  std::fflush(stdout);
}

int deviceIndex() {
  const char *e = std::getenv("SPZ_AMD_DEVICE");
  return e ? std::atoi(e) : 0;
}

int dimForDegree(int degree) {  // load-spz.cc:58-72
  switch (degree) {
    case 0: return 0;
    case 1: return 3;
    case 2: return 8;
    case 3: return 15;
    default:
      logLine("[SPZ: ERROR] Unsupported SH degree: %d\n", degree);
      return 0;
  }
}

#define SPZ_CHECK(x)                                             \
  if (!(x)) {                                                    \
    logLine("[SPZ: ERROR] Check failed: %s:%d: %s", __FILE__, __LINE__, #x); \
    return false;                                                \
  }

// checkSizes(const GaussianCloud&), load-spz.cc:106-117
bool checkSizes(const GaussianCloud &g) {
  SPZ_CHECK(g.numPoints >= 0);
  SPZ_CHECK(g.shDegree >= 0);
  SPZ_CHECK(g.shDegree <= 3);
  const size_t n = static_cast<size_t>(g.numPoints);
  SPZ_CHECK(g.positions.size() == n * 3);
  SPZ_CHECK(g.scales.size() == n * 3);
  SPZ_CHECK(g.rotations.size() == n * 4);
  SPZ_CHECK(g.alphas.size() == n);
  SPZ_CHECK(g.colors.size() == n * 3);
  SPZ_CHECK(g.sh.size() == n * dimForDegree(g.shDegree) * 3);
  return true;
}

// checkSizes(const PackedGaussians&, ...), load-spz.cc:119-127
bool checkSizes(const PackedGaussians &p, int32_t numPoints, int32_t shDim, bool usesFloat16) {
  const size_t n = static_cast<size_t>(numPoints);
  SPZ_CHECK(p.positions.size() == n * 3 * (usesFloat16 ? 2 : 3));
  SPZ_CHECK(p.scales.size() == n * 3);
  SPZ_CHECK(p.rotations.size() == n * (p.usesQuaternionSmallestThree ? 4 : 3));
  SPZ_CHECK(p.alphas.size() == n);
  SPZ_CHECK(p.colors.size() == n * 3);
  SPZ_CHECK(p.sh.size() == n * shDim * 3);
  return true;
}

bool deviceFailed(int rc, const char *what) {
  g_last_status = rc;
  if (rc == SPZ_AMD_OK) return false;
  logLine("[SPZ ERROR] spz_amd: %s: %s", what, spz_amd_status_string(rc));
  return true;
}

// Header checks of deserializePackedGaussians (load-spz.cc:551-568,591-594) with its log lines.
bool peekHeaderLogged(const uint8_t *stream, size_t size, spz_amd_header *hdr) {
  const int rc = spz_amd_peek_header(stream, size, hdr);
  switch (rc) {
    case SPZ_AMD_OK: return true;
    case SPZ_AMD_ERR_HEADER_NOT_FOUND:
      logLine("[SPZ ERROR] deserializePackedGaussians: header not found");
      break;
    case SPZ_AMD_ERR_VERSION: {
      uint32_t v = 0;
      std::memcpy(&v, stream + 4, 4);
      logLine("[SPZ ERROR] deserializePackedGaussians: version not supported: %d", v);
      break;
    }
    case SPZ_AMD_ERR_TOO_MANY_POINTS: {
      uint32_t n = 0;
      std::memcpy(&n, stream + 8, 4);
      logLine("[SPZ ERROR] deserializePackedGaussians: Too many points: %d", n);
      break;
    }
    case SPZ_AMD_ERR_SH_DEGREE:
      logLine("[SPZ ERROR] deserializePackedGaussians: Unsupported SH degree: %d", stream[12]);
      break;
    case SPZ_AMD_ERR_SHORT_STREAM:
      logLine("[SPZ ERROR] deserializePackedGaussians: read error");
      break;
    default:
      logLine("[SPZ ERROR] deserializePackedGaussians: %s", spz_amd_status_string(rc));
      break;
  }
  return false;
}

// File I/O of the overloads that take a file name (load-spz.cc:634-650, 652-668 use an ifstream / ofstream).  A 409 MB
// .spz through a zero-filled vector and one read() takes 0.15 s — more than the load itself — and a vector of 4 KiB
// pages then uploads at a tenth of the link's rate: the buffer is sized without being written, mapped with huge pages by
// several threads, and filled in 8 MiB pieces by several threads with pread: 409 MB 0.247 -> 0.133 s for the whole
// loadSpz(name).  Anything that is not a regular file of 16 MiB or more takes the reference's stream route.
using detail::fileIoThreads;
using detail::kParallelIoMin;
using detail::parallelRead;

bool readFileStream(const std::string &filename, std::vector<uint8_t> *data, bool log) {
  std::ifstream in(filename, std::ios::binary | std::ios::ate);
  if (!in.good()) {
    if (log) logLine("[SPZ ERROR] Unable to open: %s", filename.c_str());
    return false;
  }
  data->resize(static_cast<size_t>(in.tellg()));
  in.seekg(0, std::ios::beg);
  in.read(reinterpret_cast<char *>(data->data()), static_cast<std::streamsize>(data->size()));
  if (!in.good()) {
    if (log) logLine("[SPZ ERROR] Unable to load data from: %s", filename.c_str());
    return false;
  }
  return true;
}

bool readFile(const std::string &filename, std::vector<uint8_t> *data, bool log) {
  const int fd = fileIoThreads() == 0 ? -1 : ::open(filename.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd < 0) return readFileStream(filename, data, log);  // (its message for a file that cannot be opened)
  struct stat st;
  if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || static_cast<size_t>(st.st_size) < kParallelIoMin) {
    ::close(fd);
    return readFileStream(filename, data, log);
  }
  const size_t size = static_cast<size_t>(st.st_size);
  data->clear();
  detail::resizeUninitialized(data, size);
  {
    detail::Prefault pf;
    pf.add(data->data(), size);
    pf.start();
    pf.join();
  }
  const bool ok = parallelRead(fd, data->data(), size);
  ::close(fd);
  if (!ok && log) logLine("[SPZ ERROR] Unable to load data from: %s", filename.c_str());
  return ok;
}

// The whole buffer into a new (or truncated) file; the result is what the reference's out.good() after close() says.
// (In pieces by several threads with pwrite, as the read: measured slower — 409 MB into a file in memory 127 ms with
// 8 threads, 183 with 4, against 82 ms for the one write() of the stream — the threads wait for each other where the
// file's pages are allocated.  So the reference's way.)
bool writeFile(const std::string &filename, const std::vector<uint8_t> &data) {
  std::ofstream out(filename, std::ios::binary | std::ios::out);
  out.write(reinterpret_cast<const char *>(data.data()), static_cast<std::streamsize>(data.size()));
  out.close();
  return out.good();
}

}  // namespace

unsigned effectiveCpuCount() { return detail::effectiveCpuCount(); }
int lastDeviceStatus() { return g_last_status; }
void setLastDeviceStatus(int status) { g_last_status = status; }

// ---- splat-types.h:43-81 ---------------------------------------------------------------------
CoordinateConverter coordinateConverter(CoordinateSystem from, CoordinateSystem to) {
  const int a = static_cast<int>(from) - 1, b = static_cast<int>(to) - 1;
  bool mx = true, my = true, mz = true;
  if (a >= 0 && b >= 0) {
    mx = ((a >> 0) & 1) == ((b >> 0) & 1);
    my = ((a >> 1) & 1) == ((b >> 1) & 1);
    mz = ((a >> 2) & 1) == ((b >> 2) & 1);
  }
  const float x = mx ? 1.0f : -1.0f, y = my ? 1.0f : -1.0f, z = mz ? 1.0f : -1.0f;
  CoordinateConverter c;
  c.flipP = {x, y, z};
  c.flipQ = {y * z, x * z, x * y};
  c.flipSh = {y, z, x, x * y, y * z, 1.0f, x * z, 1.0f, y, x * y * z, y, z, x, z, x};
  return c;
}

// ---- GaussianCloud methods ---------------------------------------------------------------------
void GaussianCloud::convertCoordinates(CoordinateSystem from, CoordinateSystem to) {
  g_last_status = SPZ_AMD_OK;
  if (numPoints == 0) return;  // splat-types.h:135-138
  // The reference derives the per-point sh count from the array sizes (splat-types.h:151-152).
  const size_t coeffs = sh.size() / 3 / static_cast<size_t>(numPoints);
  int degree = -1;
  for (int d = 0; d <= 3; ++d) {
    if (static_cast<size_t>(dimForDegree(d)) == coeffs) degree = d;
  }
  if (degree < 0) {
    logLine("[SPZ ERROR] spz_amd: convertCoordinates: unsupported sh size %zu for %d points", sh.size(),
            numPoints);
    return;
  }
  const int f = static_cast<int>(from), t = static_cast<int>(to), dev = deviceIndex();
  const size_t n = static_cast<size_t>(numPoints);
  if (positions.size() == n * 3 && rotations.size() == n * 4 && sh.size() == n * coeffs * 3) {
    deviceFailed(spz_amd_convert_coordinates_host(positions.empty() ? nullptr : positions.data(),
                                                  rotations.empty() ? nullptr : rotations.data(),
                                                  sh.empty() ? nullptr : sh.data(), n, degree, f, t, dev),
                 "convertCoordinates");
    return;
  }
  // Inconsistent sizes: each array is walked over its own length, like the reference loops.
  if (positions.size() >= 3) {
    deviceFailed(spz_amd_convert_coordinates_host(positions.data(), nullptr, nullptr, positions.size() / 3, 0, f,
                                                  t, dev), "convertCoordinates");
  }
  if (rotations.size() >= 4) {
    deviceFailed(spz_amd_convert_coordinates_host(nullptr, rotations.data(), nullptr, rotations.size() / 4, 0, f,
                                                  t, dev), "convertCoordinates");
  }
  if (coeffs > 0) {
    deviceFailed(spz_amd_convert_coordinates_host(nullptr, nullptr, sh.data(), n, degree, f, t, dev),
                 "convertCoordinates");
  }
}

float GaussianCloud::medianVolume() const {  // splat-types.h:170-185
  g_last_status = SPZ_AMD_OK;
  if (numPoints == 0 || scales.size() < 3) return 0.01f;
  // Selection of the middle scale sum on the device (no sort); the volume is one scalar on top of it.
  float median = 0.0f;
  if (deviceFailed(spz_amd_median_scale_sum_host(scales.data(), scales.size() / 3, &median, deviceIndex()),
                   "medianVolume")) {
    return 0.01f;
  }
  return static_cast<float>((M_PI * 4 / 3) * std::exp(median));
}

GaussianCloudData GaussianCloud::data() const {  // splat-types.h:14-22,117-130
  auto copy = [](const std::vector<float> &v) {
    SpzFloatBuffer b = {0, nullptr};
    if (!v.empty()) {
      b.count = v.size();
      b.data = new float[b.count];
      std::memcpy(b.data, v.data(), b.count * sizeof(float));
    }
    return b;
  };
  GaussianCloudData d;
  d.numPoints = numPoints;
  d.shDegree = shDegree;
  d.antialiased = antialiased;
  d.positions = copy(positions);
  d.scales = copy(scales);
  d.rotations = copy(rotations);
  d.alphas = copy(alphas);
  d.colors = copy(colors);
  d.sh = copy(sh);
  return d;
}

// ---- per-splat access, load-spz.cc:383-463 --------------------------------------------------------
PackedGaussian PackedGaussians::at(int32_t i) const {  // byte moves only
  PackedGaussian r;
  const size_t k = static_cast<size_t>(i);
  const size_t positionBytes = usesFloat16() ? 6 : 9;
  std::copy_n(positions.data() + k * positionBytes, positionBytes, r.position.data());
  std::copy_n(scales.data() + k * 3, 3, r.scale.data());
  const size_t rotationBytes = usesQuaternionSmallestThree ? 4 : 3;
  std::copy_n(rotations.data() + k * rotationBytes, rotationBytes, r.rotation.data());
  std::copy_n(colors.data() + k * 3, 3, r.color.data());
  r.alpha = alphas[k];
  const size_t shDim = static_cast<size_t>(dimForDegree(shDegree));
  const uint8_t *p = sh.data() + k * shDim * 3;
  for (size_t j = 0; j < 15; ++j) {
    const bool have = j < shDim;
    r.shR[j] = have ? p[3 * j + 0] : 128;
    r.shG[j] = have ? p[3 * j + 1] : 128;
    r.shB[j] = have ? p[3 * j + 2] : 128;
  }
  return r;
}

UnpackedGaussian PackedGaussians::unpack(int32_t i, const CoordinateConverter &c) const {
  return at(i).unpack(usesFloat16(), usesQuaternionSmallestThree, fractionalBits, c);
}

namespace {

// The coordinate system X for which coordinateConverter(RUB, X) equals `c`, or -1 when `c` is not a
// table coordinateConverter can produce.  (enum - 1: bit 0 = R, bit 1 = U, bit 2 = F; RUB = 0b011.)
int targetSystemOf(const CoordinateConverter &c) {
  for (float v : c.flipP) {
    if (v != 1.0f && v != -1.0f) return -1;
  }
  const int bits = (c.flipP[0] > 0 ? 1 : 0) | (c.flipP[1] > 0 ? 2 : 0) | (c.flipP[2] > 0 ? 0 : 4);
  const CoordinateSystem to = static_cast<CoordinateSystem>(bits + 1);
  const CoordinateConverter want = coordinateConverter(CoordinateSystem::RUB, to);
  if (want.flipP != c.flipP || want.flipQ != c.flipQ || want.flipSh != c.flipSh) return -1;
  return static_cast<int>(to);
}

// One point as a complete SH3 stream: header + the six sections (load-spz.cc:540-545).
size_t onePointStream(const PackedGaussian &g, uint32_t version, int32_t fractionalBits, uint8_t *out) {
  spz_amd_header h = {};
  h.version = version;
  h.num_points = 1;
  h.sh_degree = 3;
  h.fractional_bits = static_cast<uint8_t>(fractionalBits);
  spz_amd_write_header(&h, out);
  uint8_t *p = out + 16;
  const size_t positionBytes = version == 1 ? 6 : 9, rotationBytes = version >= 3 ? 4 : 3;
  p = std::copy_n(g.position.data(), positionBytes, p);
  *p++ = g.alpha;
  p = std::copy_n(g.color.data(), 3, p);
  p = std::copy_n(g.scale.data(), 3, p);
  p = std::copy_n(g.rotation.data(), rotationBytes, p);
  for (size_t j = 0; j < 15; ++j) {
    *p++ = g.shR[j];
    *p++ = g.shG[j];
    *p++ = g.shB[j];
  }
  return static_cast<size_t>(p - out);
}

struct OnePoint {
  float position[3], scale[3], rotation[4], alpha[1], color[3], sh[45];
};

bool decodeOnePoint(const PackedGaussian &g, uint32_t version, int32_t fractionalBits, int to, OnePoint *r) {
  uint8_t stream[16 + 9 + 1 + 3 + 3 + 4 + 45];
  const size_t size = onePointStream(g, version, fractionalBits, stream);
  spz_amd_cloud_out out = {r->position, r->scale, r->rotation, r->alpha, r->color, r->sh};
  return !deviceFailed(spz_amd_decode_host_ex(stream, size, 0, to, &out, deviceIndex()), "PackedGaussian::unpack");
}

}  // namespace

UnpackedGaussian PackedGaussian::unpack(bool usesFloat16, bool usesQuaternionSmallestThree, int32_t fractionalBits,
                                        const CoordinateConverter &c) const {
  g_last_status = SPZ_AMD_OK;
  UnpackedGaussian result = {};
  const int to = targetSystemOf(c);
  if (to < 0) {
    logLine("[SPZ ERROR] spz_amd: PackedGaussian::unpack: the CoordinateConverter is not one coordinateConverter() makes");
    g_last_status = SPZ_AMD_ERR_INVALID_ARG;
    return result;
  }
  // stream versions: 1 = float16 positions + first-three rotations, 2 = 24-bit + first-three,
  // 3 = 24-bit + smallest-three.  The fourth combination the signature allows (float16 with
  // smallest-three, which no stream version encodes) takes positions from a v1 and the rest from a v3 decode.
  OnePoint a;
  const uint32_t version = usesQuaternionSmallestThree ? 3u : (usesFloat16 ? 1u : 2u);
  if (!decodeOnePoint(*this, version, fractionalBits, to, &a)) return result;
  if (usesFloat16 && usesQuaternionSmallestThree) {
    OnePoint b;
    if (!decodeOnePoint(*this, 1u, fractionalBits, to, &b)) return result;
    std::copy_n(b.position, 3, a.position);
  }
  std::copy_n(a.position, 3, result.position.data());
  std::copy_n(a.rotation, 4, result.rotation.data());
  std::copy_n(a.scale, 3, result.scale.data());
  std::copy_n(a.color, 3, result.color.data());
  result.alpha = a.alpha[0];
  for (size_t j = 0; j < 15; ++j) {
    result.shR[j] = a.sh[3 * j + 0];
    result.shG[j] = a.sh[3 * j + 1];
    result.shB[j] = a.sh[3 * j + 2];
  }
  return result;
}

bool PackedGaussians::usesFloat16() const {  // load-spz.cc:465
  return positions.size() == static_cast<size_t>(numPoints) * 3 * 2;
}

// ---- gzip, load-spz.cc:141-214 -------------------------------------------------------------------
namespace {
// Threads for the byte-identical parallel writer (spz_deflate.cpp): SPZ_AMD_GZIP_EXACT_THREADS, default
// min(cores, 32); 0 or 1 keeps everything in zlib.
int exactGzipThreads() {
  const char *e = std::getenv("SPZ_AMD_GZIP_EXACT_THREADS");
  if (e) return std::max(0, std::atoi(e));
  return static_cast<int>(std::min<unsigned>(detail::effectiveCpuCount(), 32u));
}
}  // namespace

namespace {

// Physical memory the machine has free right now (0 if unknown).
size_t availablePhysicalBytes() {
  const long pages = sysconf(_SC_AVPHYS_PAGES), page = sysconf(_SC_PAGESIZE);
  return (pages > 0 && page > 0) ? static_cast<size_t>(pages) * static_cast<size_t>(page) : 0;
}

// What a member of the parallel writers has been through before it is returned:
//   always        the first 64-256 KiB compared with zlib's own output; every Huffman block's bit count compared with the
//                 layout's plan; and on the device route every symbol checked against the input on the device
//                 (lz_validate_kernel: literals are their input bytes, matches copy equal bytes, blocks cover their
//                 ranges) — the LZ77 stage can only be lossless, whatever the parse kernels did.
//   SPZ_AMD_GZIP_VERIFY=1  additionally the finished member is inflated and compared with the input byte for byte
//                 before it is returned: on the device route by the device reader on the body still in HBM
//                 (spz_amd_zlib_verify_member, ~0.1 s for a 650 MB stream), otherwise by the host readers.
//   SPZ_AMD_GZIP_VERIFY=2  additionally zlib itself runs over the whole input and the two members are compared byte for
//                 byte (the identity claim itself, at zlib's price).
// A failed check discards the member, logs, and the next writer down (host writer, then zlib) produces it.
int gzipVerifyLevel() {
  const char *e = std::getenv("SPZ_AMD_GZIP_VERIFY");
  return e ? std::max(0, std::atoi(e)) : 0;
}

bool compressGzippedZlib(const uint8_t *data, size_t size, std::vector<uint8_t> *out);

bool verifiedExact(const uint8_t *data, size_t size, const std::vector<uint8_t> &member, int level, bool inflated_already = false) {
  if (level >= 1 && !inflated_already) {
    std::vector<uint8_t> back;
    if (!decompressGzipped(member.data(), member.size(), &back) || back.size() != size ||
        std::memcmp(back.data(), data, size) != 0) {
      logLine("[SPZ ERROR] spz_amd: the parallel gzip writer's member does not inflate to its input; using zlib");
      return false;
    }
  }
  if (level >= 2) {
    std::vector<uint8_t> z;
    if (!compressGzippedZlib(data, size, &z) || z != member) {
      logLine("[SPZ ERROR] spz_amd: the parallel gzip writer's bytes differ from zlib's; using zlib");
      return false;
    }
  }
  return true;
}

std::atomic<uint64_t> g_device_rejects{0};  // members of the device writer that failed a check and were not returned

// The LZ77 parse of the exact writer on the MI355X (spz_lz77.hip): SPZ_AMD_GZIP_DEVICE = 0 never, 1 whenever a
// device answers, unset: for inputs of 2 MiB and more when a device answers.
struct DeviceHeadParser final : exactgz::HeadParser {
  void *ctx = nullptr;
  int status = SPZ_AMD_OK;
  const uint8_t *d_copy = nullptr;  // the input's bytes on the device already (saveSpz), or null
  void *session = nullptr;          // a parse whose first stages were fed while the input was produced (saveSpz), or null
  ~DeviceHeadParser() override {
    spz_amd_zlib_parse_close(ctx);
    spz_amd_zlib_session_close(session);
  }
  int open(const uint8_t *data, size_t size, uint64_t tail_begin, const uint32_t *tail_rec, uint32_t n_rec, void (*produce)(void *),
           void *arg, uint64_t *num_symbols, uint32_t *tail_first_symbol) {
    if (session != nullptr) {  // consumed by the call, whatever it returns
      void *q = session;
      session = nullptr;
      return spz_amd_zlib_parse_open_session(q, data, d_copy, size, tail_begin, tail_rec, n_rec, &ctx, num_symbols, tail_first_symbol,
                                             produce, arg);
    }
    return spz_amd_zlib_parse_open_dev(data, d_copy, size, tail_begin, tail_rec, n_rec, deviceIndex(), &ctx, num_symbols,
                                       tail_first_symbol, produce, arg);
  }
  bool parse(const uint8_t *data, size_t size, uint64_t tail_begin, const uint32_t *tail_rec, uint32_t n_rec,
             uint64_t *num_symbols, uint32_t *tail_first_symbol) override {
    status = open(data, size, tail_begin, tail_rec, n_rec, nullptr, nullptr, num_symbols, tail_first_symbol);
    return status == SPZ_AMD_OK;
  }
  bool parseLate(const uint8_t *data, size_t size, uint64_t tail_begin, const uint32_t *tail_rec, uint32_t n_rec,
                 void (*produce)(void *), void *arg, uint64_t *num_symbols, uint32_t *tail_first_symbol) override {
    status = open(data, size, tail_begin, tail_rec, n_rec, produce, arg, num_symbols, tail_first_symbol);
    return status == SPZ_AMD_OK;
  }
  bool fetch(uint16_t *dist, uint8_t *lc) override {
    status = spz_amd_zlib_parse_fetch(ctx, dist, lc);
    return status == SPZ_AMD_OK;
  }
  // SPZ_AMD_GZIP_DEVICE_HUFFMAN=0: symbols come back and the host codes them (the parse alone on the device)
  bool canFinish() const override {
    const char *e = std::getenv("SPZ_AMD_GZIP_DEVICE_HUFFMAN");
    return !(e && e[0] == '0');
  }
  bool append(const uint16_t *dist, const uint8_t *lc, size_t n) override {
    status = spz_amd_zlib_parse_append(ctx, dist, lc, n);
    return status == SPZ_AMD_OK;
  }
  bool blockStats(const spz_amd_deflate_static &t, uint32_t block_syms, uint32_t nblocks, uint16_t *lfreq, uint16_t *dfreq,
                  uint32_t *bytes, uint32_t *last_len) override {
    status = spz_amd_zlib_block_stats(ctx, &t, block_syms, nblocks, lfreq, dfreq, bytes, last_len);
    return status == SPZ_AMD_OK;
  }
  bool canBuildTrees() const override {
    const char *e = std::getenv("SPZ_AMD_GZIP_DEVICE_TREES");
    return !(e && e[0] == '0');
  }
  bool blockTrees(uint32_t total, spz_amd_deflate_plan *plan) override {
    status = spz_amd_zlib_block_trees(ctx, total, plan);
    return status == SPZ_AMD_OK;
  }
  bool encodePlanned(const spz_amd_deflate_static &t, uint32_t block_syms, uint32_t total, const spz_amd_deflate_block *blocks,
                     uint64_t body_bytes) override {
    status = spz_amd_zlib_encode_planned(ctx, &t, block_syms, total, blocks, body_bytes);
    return status == SPZ_AMD_OK;
  }
  bool encodeGroup(const spz_amd_deflate_static &t, uint32_t block_syms, uint32_t total, uint32_t first, uint32_t n,
                   const spz_amd_deflate_block *blocks, const spz_amd_deflate_codes *codes, const uint32_t *words, uint64_t nwords,
                   uint64_t body_bytes_bound) override {
    status = spz_amd_zlib_encode_group(ctx, &t, block_syms, total, first, n, blocks, codes, words, nwords, body_bytes_bound);
    return status == SPZ_AMD_OK;
  }
  bool encodeFinish(uint32_t total, uint64_t body_bytes, uint8_t *body, uint64_t *symbol_bits, uint32_t *header_bits) override {
    status = spz_amd_zlib_encode_finish_ex(ctx, total, body_bytes, body, symbol_bits, header_bits);
    return status == SPZ_AMD_OK;
  }
  // SPZ_AMD_GZIP_VERIFY >= 1: the body, still in device memory, inflated there and compared with the input byte for byte
  int verifyMember(uint64_t body_bytes) { return spz_amd_zlib_verify_member(ctx, body_bytes); }
};

bool deviceParseWanted(size_t size) {
  const char *e = std::getenv("SPZ_AMD_GZIP_DEVICE");
  if (e && e[0] == '0') return false;
  const bool forced = e && e[0] == '1';
  // profiles/r03_size_sweep.json: the device writer wins from the smallest stream measured (60 k points SH3, 3.9 MB:
  // 13 ms against 53 ms for the multi-threaded host writer on the box's 16 CPUs)
  if (!forced && size < (size_t(2) << 20)) return false;
  return spz_amd_device_count() > 0;
}

std::atomic<uint64_t> g_device_parses{0};

}  // namespace

uint64_t deviceGzipParseCount() { return g_device_parses.load(); }
uint64_t deviceGzipRejectCount() { return g_device_rejects.load(); }

namespace {
// A saveSpz whose container stage has been fed beside the upload: the session, and the writer's tail job started as soon
// as the stream's last bytes were on the host
struct SaveAhead {
  void *session = nullptr;
  exactgz::TailAhead *tail = nullptr;
  const uint8_t *stream = nullptr;
  size_t size = 0;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~SaveAhead() {
    if (session) spz_amd_zlib_session_close(session);
    exactgz::tailAheadDrop(tail);
  }
};
bool compressGzippedWithCopy(const uint8_t *data, size_t size, std::vector<uint8_t> *out, const uint8_t *d_copy, SaveAhead *ahead = nullptr);
}
bool compressGzipped(const uint8_t *data, size_t size, std::vector<uint8_t> *out) {
  return compressGzippedWithCopy(data, size, out, nullptr);
}
namespace {
// d_copy: the same bytes on the device (spz_amd_encode_host_keep), or null; ahead: a parse already fed with them and the
// tail job already started on them (both taken; what is not consumed here goes with *ahead)
bool compressGzippedWithCopy(const uint8_t *data, size_t size, std::vector<uint8_t> *out, const uint8_t *d_copy, SaveAhead *ahead) {
  // Large inputs: the writer that reproduces zlib's bytes exactly with its parse on the device or on all
  // cores (it checks itself against zlib on a prefix, and declines inputs it cannot split); zlib itself
  // otherwise and as the fallback.
  constexpr size_t kExactMinBytes = size_t(1) << 20;
  if (size >= kExactMinBytes && std::strcmp(zlibVersion(), "1.2.11") == 0) {
    const int threads = exactGzipThreads();
    if (threads >= 1 && deviceParseWanted(size)) {
      const size_t verify = std::min<size_t>(size_t(256) << 10, std::max<size_t>(size_t(64) << 10, size / 16));
      const size_t avail = availablePhysicalBytes();
      if (avail == 0 || size / 2 * 9 < avail) {
        static const bool timing = std::getenv("SPZ_AMD_EXACT_GZIP_TIMING") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        bool ok;
        const int level = gzipVerifyLevel();
        bool inflated_on_device = false;
        {
          DeviceHeadParser parser;
          parser.d_copy = d_copy;
          exactgz::TailAhead *tail = nullptr;
          if (ahead && d_copy) {
            parser.session = ahead->session;
            ahead->session = nullptr;
            tail = ahead->tail;
            ahead->tail = nullptr;
          }
          ok = exactgz::compressWithHeadParser(data, size, std::max(threads, 1), parser, out, verify, tail);
          if (timing) {
            std::fprintf(stderr, "[exactgz] writer     %.3f s in all\n",
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
          }
          if (!ok && parser.status == SPZ_AMD_ERR_VERIFY) {
            g_device_rejects.fetch_add(1);
            logLine("[SPZ ERROR] spz_amd: the device gzip writer's symbols do not reproduce the input; using the host writer");
          }
          if (ok && level >= 1 && out->size() > 18) {
            const int rc = parser.verifyMember(out->size() - 18);
            if (rc == SPZ_AMD_OK) {
              inflated_on_device = true;
            } else if (rc == SPZ_AMD_ERR_VERIFY) {
              g_device_rejects.fetch_add(1);
              logLine("[SPZ ERROR] spz_amd: the device gzip writer's member does not inflate to its input; using the host writer");
              ok = false;
            }  // anything else: the device reader declined, the host inflates below
            if (timing) {
              std::fprintf(stderr, "[exactgz] + verify   %.3f s in all (device reader: %s)\n",
                           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), spz_amd_status_string(rc));
            }
          }
        }
        if (timing) {
          std::fprintf(stderr, "[exactgz] + release  %.3f s in all\n",
                       std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        if (ok) {
          // level 1 is done when the device has inflated and compared; level 2 (zlib's own member) is the host's either way
          if (level == 0 || (level == 1 && inflated_on_device) || verifiedExact(data, size, *out, inflated_on_device ? 2 : level, inflated_on_device)) {
            g_device_parses.fetch_add(1);
            return true;
          }
        }
      }
    }
    // parse jobs of 1 MiB; smaller ones (down to 128 KiB) when that is what it takes to give every thread two
    const int windows = static_cast<int>(std::min<size_t>(32, std::max<size_t>(4, size / (size_t(threads > 0 ? threads : 1) * 2 * 32768))));
    const size_t verify = std::min<size_t>(size_t(256) << 10, std::max<size_t>(size_t(64) << 10, size / 16));
    // The writer keeps every job's symbols until assembly: about 3 bytes per input byte plus the output, where
    // zlib needs under 1 MB.  A host that does not have that to spare gets zlib's pace instead of the OOM killer.
    const size_t avail = availablePhysicalBytes();
    const bool fits = avail == 0 || size / 2 * 9 < avail;   // 4.5 x the input
    if (threads > 1 && fits && exactgz::compress(data, size, threads, windows, out, verify)) {
      const int level = gzipVerifyLevel();
      if (level == 0 || verifiedExact(data, size, *out, level)) return true;
    }
  }
  return compressGzippedZlib(data, size, out);
}
}  // namespace

namespace {
bool compressGzippedZlib(const uint8_t *data, size_t size, std::vector<uint8_t> *out) {
  z_stream stream = {};
  // Same parameters as load-spz.cc:190: default level, gzip wrapper, memLevel 9.
  if (deflateInit2(&stream, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 16 + MAX_WBITS, 9, Z_DEFAULT_STRATEGY) != Z_OK) {
    return false;
  }
  out->clear();
  std::vector<uint8_t> buffer(1u << 20);
  const uint8_t *next = data;
  size_t left = size;
  bool success = false;
  while (true) {
    // deflate output does not depend on how the input is chunked; feed at most 1 GiB per call.
    const size_t chunk = std::min<size_t>(left, size_t(1) << 30);
    stream.next_in = const_cast<Bytef *>(reinterpret_cast<const Bytef *>(next));
    stream.avail_in = static_cast<uInt>(chunk);
    next += chunk;
    left -= chunk;
    const int flush = (left == 0) ? Z_FINISH : Z_NO_FLUSH;
    int res = Z_OK;
    do {
      stream.next_out = buffer.data();
      stream.avail_out = static_cast<uInt>(buffer.size());
      res = deflate(&stream, flush);
      if (res != Z_OK && res != Z_STREAM_END && res != Z_BUF_ERROR) break;
      out->insert(out->end(), buffer.data(), buffer.data() + (buffer.size() - stream.avail_out));
    } while (stream.avail_out == 0 || (flush == Z_FINISH && res != Z_STREAM_END));
    if (res == Z_STREAM_END) {
      success = true;
      break;
    }
    if ((res != Z_OK && res != Z_BUF_ERROR) || left == 0) break;
  }
  deflateEnd(&stream);
  return success;
}
}  // namespace

// ---- opt-in parallel gzip ---------------------------------------------------------
// zlib is ~87 % of an end-to-end saveSpz.  This is pigz's "independent blocks"
// construction: the input is cut into blocks, every block is deflated on its own (raw deflate, ended
// on a byte boundary with Z_SYNC_FLUSH; the last one with Z_FINISH), and the pieces are concatenated
// inside ONE gzip member whose CRC-32 is folded together with crc32_combine.  Any gzip reader, the
// reference's loadSpz included, reads the result; the bytes differ from single-stream deflate, so it
// is used only when asked for (threads > 1).  The member's header carries an FEXTRA subfield "SZ"
// (RFC 1952 §2.3.1.1; readers that do not know it skip it) listing the compressed size of every piece,
// which is what lets decompressGzipped below inflate the pieces concurrently.
namespace {

constexpr uint8_t kIndexId1 = 'S', kIndexId2 = 'Z';
constexpr uint32_t kIndexVersion = 1;

struct GzipIndex {
  uint32_t blockBytes = 0;            // uncompressed bytes per piece (the last may be shorter)
  uint64_t totalBytes = 0;            // uncompressed size of the member
  std::vector<uint32_t> pieceBytes;   // compressed size of every piece
};

void putLe(std::vector<uint8_t> *v, uint64_t x, int bytes) {
  for (int k = 0; k < bytes; ++k) v->push_back(static_cast<uint8_t>(x >> (8 * k)));
}
uint64_t getLe(const uint8_t *p, int bytes) {
  uint64_t x = 0;
  for (int k = 0; k < bytes; ++k) x |= static_cast<uint64_t>(p[k]) << (8 * k);
  return x;
}

// Walks a gzip member header (RFC 1952 §2.3).  Returns its length, 0 if it is not one; fills *idx when
// an "SZ" subfield of the known version is present.
size_t parseGzipHeader(const uint8_t *p, size_t n, GzipIndex *idx) {
  if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return 0;
  const uint8_t flg = p[3];
  size_t pos = 10;
  if (flg & 4) {  // FEXTRA
    if (pos + 2 > n) return 0;
    const size_t xlen = getLe(p + pos, 2);
    pos += 2;
    if (pos + xlen > n) return 0;
    size_t q = pos;
    while (q + 4 <= pos + xlen) {
      const size_t len = getLe(p + q + 2, 2);
      if (q + 4 + len > pos + xlen) break;
      const uint8_t *d = p + q + 4;
      if (p[q] == kIndexId1 && p[q + 1] == kIndexId2 && len >= 20 && getLe(d, 4) == kIndexVersion) {
        const uint64_t nb = getLe(d + 16, 4);
        if (20 + 4 * nb == len) {
          idx->blockBytes = static_cast<uint32_t>(getLe(d + 4, 4));
          idx->totalBytes = getLe(d + 8, 8);
          idx->pieceBytes.resize(nb);
          for (uint64_t i = 0; i < nb; ++i) idx->pieceBytes[i] = static_cast<uint32_t>(getLe(d + 20 + 4 * i, 4));
        }
      }
      q += 4 + len;
    }
    pos += xlen;
  }
  for (int f = 8; f <= 16; f <<= 1) {  // FNAME, FCOMMENT: zero-terminated
    if (flg & f) {
      while (pos < n && p[pos] != 0) ++pos;
      if (pos >= n) return 0;
      ++pos;
    }
  }
  // FHCRC: zlib checks the header CRC16 and rejects a member whose CRC16 is wrong; the fast readers start
  // behind the header and would not.  The reference writer never sets it, so such members go to zlib.
  if (flg & 2) return 0;
  return pos + 8 <= n ? pos : 0;
}

int gunzipThreads(size_t pieces) {
  const char *e = std::getenv("SPZ_AMD_GUNZIP_THREADS");
  size_t t = e ? static_cast<size_t>(std::max(1, std::atoi(e))) : std::min<size_t>(detail::effectiveCpuCount(), 32);
  return static_cast<int>(std::max<size_t>(1, std::min(t, pieces)));
}

// Inflates the pieces of an indexed member concurrently, straight into their final positions.  Every
// size, the CRC-32 and ISIZE are checked; on ANY inconsistency the caller falls back to the serial
// reader, which is the authority on what is and is not a valid gzip file.
bool inflateIndexed(const uint8_t *p, size_t n, size_t headerLen, const GzipIndex &idx, std::vector<uint8_t> *out) {
  const size_t nb = idx.pieceBytes.size();
  if (nb == 0 || idx.blockBytes == 0) return false;
  if ((idx.totalBytes + idx.blockBytes - 1) / idx.blockBytes != nb) return false;
  std::vector<size_t> off(nb + 1, headerLen);
  for (size_t i = 0; i < nb; ++i) off[i + 1] = off[i] + idx.pieceBytes[i];
  if (off[nb] + 8 != n) return false;  // exactly one member, nothing after it
  if (idx.totalBytes > (n - headerLen) * 1032 + 1024) return false;  // beyond deflate's maximum expansion
  out->clear();
  detail::resizeUninitialized(out, idx.totalBytes);  // every piece is checked to fill its range exactly
  std::vector<uLong> crcs(nb);
  std::atomic<size_t> next{0};
  std::atomic<bool> failed{false};
  auto worker = [&]() {
    for (;;) {
      const size_t b = next.fetch_add(1);
      if (b >= nb || failed.load()) return;
      const size_t uoff = b * static_cast<size_t>(idx.blockBytes);
      const size_t ulen = std::min<size_t>(idx.blockBytes, idx.totalBytes - uoff);
      z_stream zs = {};
      if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) {
        failed = true;
        return;
      }
      zs.next_in = const_cast<Bytef *>(p + off[b]);
      zs.avail_in = idx.pieceBytes[b];
      zs.next_out = out->data() + uoff;
      zs.avail_out = static_cast<uInt>(ulen);
      int rc = inflate(&zs, Z_SYNC_FLUSH);
      if (rc == Z_OK && zs.avail_in > 0) rc = inflate(&zs, Z_SYNC_FLUSH);  // the empty stored block after a full buffer
      const bool last = (b + 1 == nb);
      const bool ok = zs.total_out == ulen && zs.avail_in == 0 && (last ? rc == Z_STREAM_END : (rc == Z_OK || rc == Z_BUF_ERROR));
      inflateEnd(&zs);
      if (!ok) {
        failed = true;
        return;
      }
      crcs[b] = crc32(crc32(0L, Z_NULL, 0), out->data() + uoff, static_cast<uInt>(ulen));
    }
  };
  std::vector<std::thread> pool;
  const int nt = gunzipThreads(nb);
  for (int t = 1; t < nt; ++t) pool.emplace_back(worker);
  worker();
  for (auto &t : pool) t.join();
  if (failed) return false;
  uLong crc = crc32(0L, Z_NULL, 0);
  for (size_t b = 0; b < nb; ++b) {
    const size_t ulen = std::min<size_t>(idx.blockBytes, idx.totalBytes - b * static_cast<size_t>(idx.blockBytes));
    crc = crc32_combine(crc, crcs[b], static_cast<z_off_t>(ulen));
  }
  return getLe(p + n - 8, 4) == static_cast<uint32_t>(crc) && getLe(p + n - 4, 4) == (idx.totalBytes & 0xffffffffu);
}

// libdeflate (a whole-buffer inflater, about twice zlib's speed) is used when the system has it; it is
// looked up at run time so that nothing depends on it being there.  SPZ_AMD_NO_LIBDEFLATE=1 disables it.
struct LibDeflate {
  void *(*alloc)() = nullptr;
  int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
  void (*release)(void *) = nullptr;
  LibDeflate() {
    const char *e = std::getenv("SPZ_AMD_NO_LIBDEFLATE");
    if (e && std::atoi(e) != 0) return;
    void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    alloc = reinterpret_cast<void *(*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
    gunzip = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *)>(
        dlsym(h, "libdeflate_gzip_decompress"));
    release = reinterpret_cast<void (*)(void *)>(dlsym(h, "libdeflate_free_decompressor"));
    if (!alloc || !gunzip || !release) alloc = nullptr;
  }
  bool usable() const { return alloc != nullptr; }
};

bool inflateWholeBuffer(const uint8_t *p, size_t n, std::vector<uint8_t> *out) {
  static const LibDeflate lib;
  if (!lib.usable()) return false;
  const uint64_t isize = getLe(p + n - 4, 4);  // mod 2^32; a mismatch sends the caller to the serial reader
  if (isize == 0 || isize > n * 1032 + 1024) return false;
  void *d = lib.alloc();
  if (!d) return false;
  out->clear();
  detail::resizeUninitialized(out, isize);  // accepted only when libdeflate wrote all of it (got == isize)
  size_t got = 0;
  const int rc = lib.gunzip(d, p, n, out->data(), out->size(), &got);
  lib.release(d);
  return rc == 0 && got == isize;
}

}  // namespace

bool compressGzippedParallel(const uint8_t *data, size_t size, std::vector<uint8_t> *out, int threads) {
  if (threads <= 1 || size < (1u << 20)) return compressGzipped(data, size, out);
  size_t blockBytes = size_t(1) << 20;  // 1 MiB of input per deflate job
  while ((size + blockBytes - 1) / blockBytes > 16000) blockBytes <<= 1;  // the index must fit FEXTRA's 64 KiB
  const size_t kBlock = blockBytes;
  const size_t nblocks = (size + kBlock - 1) / kBlock;
  std::vector<std::vector<uint8_t>> pieces(nblocks);
  std::vector<uLong> crcs(nblocks);
  std::atomic<size_t> next{0};
  std::atomic<bool> failed{false};
  auto worker = [&]() {
    std::vector<uint8_t> buf;
    for (;;) {
      const size_t b = next.fetch_add(1);
      if (b >= nblocks || failed.load()) return;
      const size_t off = b * kBlock, len = std::min(kBlock, size - off);
      z_stream zs = {};
      if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -MAX_WBITS, 9, Z_DEFAULT_STRATEGY) != Z_OK) {
        failed = true;
        return;
      }
      buf.resize(deflateBound(&zs, static_cast<uLong>(len)) + 16);
      zs.next_in = const_cast<Bytef *>(data + off);
      zs.avail_in = static_cast<uInt>(len);
      zs.next_out = buf.data();
      zs.avail_out = static_cast<uInt>(buf.size());
      const bool last = (b + 1 == nblocks);
      const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
      const bool ok = last ? (rc == Z_STREAM_END) : (rc == Z_OK && zs.avail_in == 0 && zs.avail_out > 0);
      if (!ok) failed = true;
      pieces[b].assign(buf.data(), buf.data() + (buf.size() - zs.avail_out));
      deflateEnd(&zs);
      crcs[b] = crc32(crc32(0L, Z_NULL, 0), data + off, static_cast<uInt>(len));
    }
  };
  std::vector<std::thread> pool;
  const int nt = static_cast<int>(std::min<size_t>(static_cast<size_t>(threads), nblocks));
  for (int t = 0; t < nt; ++t) pool.emplace_back(worker);
  for (auto &t : pool) t.join();
  if (failed) return false;
  const size_t indexBytes = 20 + 4 * nblocks;
  size_t total = 10 + 2 + 4 + indexBytes + 8;
  for (const auto &p : pieces) total += p.size();
  out->clear();
  out->reserve(total);
  // zlib's own 10-byte header (magic, deflate, mtime 0, xfl 0, OS 3 = Unix) with FLG.FEXTRA set
  const uint8_t header[10] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0x00, 0x03};
  out->insert(out->end(), header, header + 10);
  putLe(out, 4 + indexBytes, 2);  // XLEN
  out->push_back(kIndexId1);
  out->push_back(kIndexId2);
  putLe(out, indexBytes, 2);
  putLe(out, kIndexVersion, 4);
  putLe(out, kBlock, 4);
  putLe(out, size, 8);
  putLe(out, nblocks, 4);
  for (const auto &p : pieces) putLe(out, p.size(), 4);
  uLong crc = crc32(0L, Z_NULL, 0);
  for (size_t b = 0; b < nblocks; ++b) {
    out->insert(out->end(), pieces[b].begin(), pieces[b].end());
    crc = crc32_combine(crc, crcs[b], static_cast<z_off_t>(std::min(kBlock, size - b * kBlock)));
  }
  putLe(out, static_cast<uint32_t>(crc), 4);
  putLe(out, size & 0xffffffffu, 4);
  return true;
}

// gunzip (load-spz.cc:141-182).  Three readers with one result: members written by
// compressGzippedParallel are inflated piece-parallel through their index; anything else goes through
// libdeflate when the system has it; and the zlib loop the reference uses is the fallback for every
// case the fast readers decline or fail on, so acceptance and rejection are exactly zlib's.
namespace {
// The device reader (spz_inflate_dev.hip): SPZ_AMD_GUNZIP_DEVICE = 0 never, 1 for members of 1 MiB and more, unset:
// of 8 MiB and more on hosts with fewer than 32 usable CPUs.  The result is believed only when its length matches ISIZE and its CRC-32 the trailer's.
std::atomic<uint64_t> g_device_inflates{0};

thread_local const char *g_inflate_decline = "";  // deviceInflateLastDecline()

// CRC-32 (the reflected polynomial 0xedb88320) as polynomial arithmetic: a * b mod P, and x^(8 n) mod P (bit 31 = x^0).
uint32_t crcMultModP(uint32_t a, uint32_t b) {
  uint32_t m = 1u << 31, p = 0u;
  for (;;) {
    if (a & m) {
      p ^= b;
      if ((a & (m - 1u)) == 0u) break;
    }
    m >>= 1;
    b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
  }
  return p;
}
uint32_t crcShiftFactor(uint64_t nbytes) {  // x^(8 nbytes) mod P
  uint32_t sq = 1u << 30, f = 1u << 31;     // x^1, x^0
  for (uint64_t n = nbytes << 3; n != 0; n >>= 1) {
    if (n & 1u) f = crcMultModP(sq, f);
    sq = crcMultModP(sq, sq);
  }
  return f;
}

// Is a member of this size one the device reader will be asked about (SPZ_AMD_GUNZIP_DEVICE, size, core count)?
bool deviceInflateWanted(size_t size, size_t header_len) {
  const char *e = std::getenv("SPZ_AMD_GUNZIP_DEVICE");
  if (e && e[0] == '0') return false;
  const bool forced = e && e[0] == '1';
  if (size < header_len + 8 + (size_t(1) << 20)) return false;
  if (!forced) {
    // profiles/r03_load_sweep.txt (16 usable CPUs; file MB: device / host ms): 10: 26 / 15, 20: 32 / 21, 41: 25 / 41,
    // 82: 37 / 71, 163: 80 / 165, 409: 99 / 353.  The device reader costs ~20 ms whatever the size (a wave takes that
    // long for one of zlib's blocks) and 0.2 ms per MB; the parallel host reader 13.9 ms per MB and core, if it scales
    // (it does to 16).  The faster of the two by that account: from ~30 MB with 16 cores, ~85 MB with 32.
    const double mb = static_cast<double>(size) / 1e6, cores = static_cast<double>(std::max(1u, detail::effectiveCpuCount()));
    if (20.0 + 0.2 * mb >= 13.9 * mb / cores) return false;
  }
  return spz_amd_device_count() > 0;
}

// The first `want` bytes a raw deflate stream inflates to (zlib); returns how many it got.
size_t inflatePrefix(const uint8_t *deflate, size_t n, uint8_t *out, size_t want) {
  z_stream zs = {};
  if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) return 0;
  zs.next_in = const_cast<Bytef *>(deflate);
  zs.avail_in = static_cast<uInt>(std::min<size_t>(n, size_t(1) << 16));
  zs.next_out = out;
  zs.avail_out = static_cast<uInt>(want);
  (void)inflate(&zs, Z_SYNC_FLUSH);
  const size_t got = want - zs.avail_out;
  inflateEnd(&zs);
  return got;
}

// Inflates the member on the device and checks length and CRC-32 against its trailer; nullptr: declined or wrong.
void *openVerifiedDeviceInflate(const uint8_t *gz, size_t size, size_t header_len, uint64_t *out_bytes,
                                void (*after_upload)(void *) = nullptr, void *after_arg = nullptr) {
  if (!deviceInflateWanted(size, header_len)) return nullptr;
  const size_t nbytes = size - header_len - 8;
  auto le32 = [&](const uint8_t *q) {
    return static_cast<uint32_t>(q[0]) | (static_cast<uint32_t>(q[1]) << 8) | (static_cast<uint32_t>(q[2]) << 16) |
           (static_cast<uint32_t>(q[3]) << 24);
  };
  const uint32_t want_crc = le32(gz + size - 8), isize = le32(gz + size - 4);
  void *ctx = nullptr;
  g_inflate_decline = "";
  const int open_rc = spz_amd_inflate_open_ex(gz + header_len, nbytes, deviceIndex(), &ctx, out_bytes, after_upload, after_arg);
  if (open_rc != SPZ_AMD_OK) {
    g_inflate_decline = open_rc == SPZ_AMD_ERR_UNSUPPORTED ? spz_amd_inflate_last_decline() : spz_amd_status_string(open_rc);
    return nullptr;
  }
  bool ok = static_cast<uint32_t>(*out_bytes & 0xffffffffull) == isize;
  if (ok) {
    const uint32_t piece = spz_amd_inflate_crc_piece_bytes();
    std::vector<uint32_t> crcs(static_cast<size_t>((*out_bytes + piece - 1) / piece));
    uint32_t n_pieces = 0;
    ok = spz_amd_inflate_piece_crcs(ctx, crcs.data(), static_cast<uint32_t>(crcs.size()), &n_pieces) == SPZ_AMD_OK &&
         n_pieces == crcs.size();
    if (ok) {
      // crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B).  zlib 1.2.11's crc32_combine rebuilds that operator (a
      // 32 x 32 matrix, squared log |B| times) on every call: 2 480 calls for a 650 MB stream were several
      // milliseconds of a loadSpz; every piece but the last has the same length, so the factor is computed twice.
      uint32_t crc = crcs.empty() ? 0u : crcs[0];
      const uint32_t full = crcShiftFactor(piece);
      for (size_t i = 1; i < crcs.size(); ++i) {
        const uint64_t len = std::min<uint64_t>(piece, *out_bytes - static_cast<uint64_t>(i) * piece);
        crc = crcMultModP(len == piece ? full : crcShiftFactor(len), crc) ^ crcs[i];
      }
      ok = crc == want_crc;
    }
  }
  if (!ok) {
    g_inflate_decline = "crc";
    spz_amd_inflate_close(ctx);
    return nullptr;
  }
  return ctx;
}

bool inflateOnDevice(const uint8_t *gz, size_t size, size_t header_len, std::vector<uint8_t> *out) {
  uint64_t out_bytes = 0;
  void *ctx = openVerifiedDeviceInflate(gz, size, header_len, &out_bytes);
  if (ctx == nullptr) return false;
  struct Close {
    void *c;
    ~Close() { spz_amd_inflate_close(c); }
  } closer{ctx};
  out->clear();
  detail::resizeUninitialized(out, static_cast<size_t>(out_bytes));
  {
    detail::Prefault pf;
    pf.add(out->data(), out->size());
    pf.start();
    pf.join();
  }
  if (spz_amd_inflate_fetch(ctx, out->data()) != SPZ_AMD_OK) return false;
  g_device_inflates.fetch_add(1);
  return true;
}

}  // namespace

uint64_t deviceInflateCount() { return g_device_inflates.load(); }
const char *deviceInflateLastDecline() { return g_inflate_decline; }

namespace {
bool decompressGzippedWith(const uint8_t *compressed, size_t size, std::vector<uint8_t> *out, bool try_device);
}  // namespace

bool decompressGzipped(const uint8_t *compressed, size_t size, std::vector<uint8_t> *out) {
  return decompressGzippedWith(compressed, size, out, true);
}

namespace {
bool decompressGzippedWith(const uint8_t *compressed, size_t size, std::vector<uint8_t> *out, bool try_device) {
  if (compressed != nullptr) {
    GzipIndex idx;
    const size_t headerLen = parseGzipHeader(compressed, size, &idx);
    if (headerLen != 0) {
      if (!idx.pieceBytes.empty() && inflateIndexed(compressed, size, headerLen, idx, out)) return true;
      // an ordinary single deflate stream (what the reference writes): on the device when it is large and one answers
      if (try_device && inflateOnDevice(compressed, size, headerLen, out)) return true;
      // ... or decoded in parallel on the host
      if (size >= (size_t(4) << 20)) {
        const int threads = gunzipThreads(size_t(1) << 20);
        // two decoding passes with a plain decoder: pays off from about 8 threads (measured), see spz_inflate.cpp
        if (threads >= 8 && pinflate::inflate(compressed, size, headerLen, threads, out)) return true;
      }
      if (inflateWholeBuffer(compressed, size, out)) return true;
    }
  }
  z_stream stream = {};
  // 16 | MAX_WBITS: gzip wrapper only (load-spz.cc:172).
  if (inflateInit2(&stream, 16 | MAX_WBITS) != Z_OK) return false;
  out->clear();
  std::vector<uint8_t> buffer(1u << 20);
  const uint8_t *next = compressed;
  size_t left = size;
  bool success = false;
  int res = Z_OK;
  while (res != Z_STREAM_END) {
    if (stream.avail_in == 0) {
      const size_t chunk = std::min<size_t>(left, size_t(1) << 30);
      stream.next_in = const_cast<Bytef *>(next);
      stream.avail_in = static_cast<uInt>(chunk);
      next += chunk;
      left -= chunk;
    }
    stream.next_out = buffer.data();
    stream.avail_out = static_cast<uInt>(buffer.size());
    res = inflate(&stream, Z_NO_FLUSH);
    if (res != Z_OK && res != Z_STREAM_END) break;  // incl. Z_BUF_ERROR on truncated input
    out->insert(out->end(), buffer.data(), buffer.data() + (buffer.size() - stream.avail_out));
    if (res == Z_STREAM_END) success = true;
  }
  inflateEnd(&stream);
  return success;
}
}  // namespace

// ---- (de)serialise, load-spz.cc:533-596 ----------------------------------------------------------
void serializePackedGaussians(const PackedGaussians &packed, std::ostream *out) {
  spz_amd_header h = {};
  h.version = 3;  // the reference writer never sets another version (load-spz.cc:133,534-539)
  h.num_points = static_cast<uint32_t>(packed.numPoints);
  h.sh_degree = static_cast<uint8_t>(packed.shDegree);
  h.fractional_bits = static_cast<uint8_t>(packed.fractionalBits);
  h.flags = static_cast<uint8_t>(packed.antialiased ? 1 : 0);
  uint8_t raw[16];
  spz_amd_write_header(&h, raw);
  out->write(reinterpret_cast<const char *>(raw), 16);
  auto put = [&](const std::vector<uint8_t> &v) {
    out->write(reinterpret_cast<const char *>(v.data()), static_cast<std::streamsize>(v.size()));
  };
  put(packed.positions);
  put(packed.alphas);
  put(packed.colors);
  put(packed.scales);
  put(packed.rotations);
  put(packed.sh);
}

namespace {

PackedGaussians slicePacked(const uint8_t *stream, const spz_amd_header &hdr) {
  spz_amd_layout lay;
  spz_amd_stream_layout(hdr.num_points, hdr.sh_degree, static_cast<int>(hdr.version), &lay);
  PackedGaussians r;
  r.numPoints = static_cast<int32_t>(hdr.num_points);
  r.shDegree = hdr.sh_degree;
  r.fractionalBits = hdr.fractional_bits;
  r.antialiased = (hdr.flags & 1) != 0;
  r.usesQuaternionSmallestThree = hdr.version >= 3;
  auto take = [&](int s) {
    return std::vector<uint8_t>(stream + lay.offset[s], stream + lay.offset[s] + lay.bytes[s]);
  };
  r.positions = take(SPZ_AMD_SEC_POSITIONS);
  r.alphas = take(SPZ_AMD_SEC_ALPHAS);
  r.colors = take(SPZ_AMD_SEC_COLORS);
  r.scales = take(SPZ_AMD_SEC_SCALES);
  r.rotations = take(SPZ_AMD_SEC_ROTATIONS);
  r.sh = take(SPZ_AMD_SEC_SH);
  return r;
}

}  // namespace

PackedGaussians deserializePackedGaussians(std::istream &in) {
  std::vector<uint8_t> data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  spz_amd_header hdr;
  if (!peekHeaderLogged(data.data(), data.size(), &hdr)) return {};
  return slicePacked(data.data(), hdr);
}

// ---- pack / unpack -------------------------------------------------------------------------------
namespace {
// Sizes the six arrays of a cloud that a device copy is about to fill completely.
void sizeCloudArrays(GaussianCloud *r, size_t n, size_t shDim, detail::Prefault *prefault) {
  std::vector<float> *arrays[6] = {&r->positions, &r->scales, &r->rotations, &r->alphas, &r->colors, &r->sh};
  const size_t counts[6] = {n * 3, n * 3, n * 4, n, n * 3, n * shDim * 3};
  for (int i = 0; i < 6; ++i) {
    detail::resizeUninitialized(arrays[i], counts[i]);
    prefault->add(arrays[i]->data(), counts[i] * sizeof(float));
  }
}
}  // namespace

namespace {
bool packToStreamKeep(const GaussianCloud &g, const PackOptions &o, std::vector<uint8_t> *stream, const uint8_t **d_copy,
                      SaveAhead *ahead = nullptr);
}
bool packToStream(const GaussianCloud &g, const PackOptions &o, std::vector<uint8_t> *stream) {
  return packToStreamKeep(g, o, stream, nullptr);
}
namespace {
// d_copy != null: *d_copy receives the device's copy of the stream (or null), to be given back with spz_amd_kept_stream_release
bool packToStreamKeep(const GaussianCloud &g, const PackOptions &o, std::vector<uint8_t> *stream, const uint8_t **d_copy,
                      SaveAhead *ahead) {
  if (d_copy) *d_copy = nullptr;
  g_last_status = SPZ_AMD_OK;
  if (!checkSizes(g)) {
    // packGaussians returns an empty PackedGaussians{} (load-spz.cc:258-260) and saveSpz goes on
    // to write it: a 0-point stream whose header carries the defaults (fractionalBits 0).
    spz_amd_header h = {};
    h.version = 3;
    stream->assign(16, 0);
    spz_amd_write_header(&h, stream->data());
    return true;
  }
  spz_amd_layout lay;
  if (spz_amd_stream_layout(static_cast<uint64_t>(g.numPoints), g.shDegree, 3, &lay) != SPZ_AMD_OK) return false;
  // every byte of the stream is written by the device copy: no zero fill, pages mapped in the background
  stream->clear();
  detail::resizeUninitialized(stream, lay.total_bytes);
  detail::Prefault prefault;
  prefault.add(stream->data(), stream->size());
  prefault.start();
  spz_amd_cloud_in in = {g.positions.data(), g.scales.data(), g.rotations.data(),
                         g.alphas.data(),    g.colors.data(), g.sh.empty() ? nullptr : g.sh.data()};
  void *zlib_session = ahead ? ahead->session : nullptr;
  auto tail_ready = [](void *p) {
    SaveAhead *a = static_cast<SaveAhead *>(p);
    try {
      a->tail = exactgz::tailAheadStart(a->stream, a->size);
    } catch (...) {  // on the C library's download thread: the writer then runs its tail job itself
      a->tail = nullptr;
    }
    if (std::getenv("SPZ_AMD_EXACT_GZIP_TIMING")) {
      std::fprintf(stderr, "[saveSpz] the stream's end is on the host %.4f s into the pack\n",
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - a->t0).count());
    }
  };
  if (ahead) {
    ahead->stream = stream->data();
    ahead->size = stream->size();
    ahead->t0 = std::chrono::steady_clock::now();
  }
  const size_t tail_bytes = zlib_session ? exactgz::tailAheadBytes(stream->size()) : 0;
  const int rc = d_copy ? spz_amd_encode_host_keep_session_tail(&in, static_cast<uint64_t>(g.numPoints), g.shDegree, g.antialiased ? 1 : 0,
                                                                static_cast<int>(o.from), 3, stream->data(), stream->size(), deviceIndex(),
                                                                d_copy, zlib_session, tail_bytes, tail_bytes ? +tail_ready : nullptr, ahead)
                        : spz_amd_encode_host(&in, static_cast<uint64_t>(g.numPoints), g.shDegree, g.antialiased ? 1 : 0,
                                              static_cast<int>(o.from), 3, stream->data(), stream->size(), deviceIndex());
  return !deviceFailed(rc, "encode");
}
}  // namespace

GaussianCloud unpackFromStream(const uint8_t *stream, size_t size, const UnpackOptions &o) {
  g_last_status = SPZ_AMD_OK;
  spz_amd_header hdr;
  if (!peekHeaderLogged(stream, size, &hdr)) return {};
  GaussianCloud r;
  r.numPoints = static_cast<int32_t>(hdr.num_points);
  r.shDegree = hdr.sh_degree;
  r.antialiased = (hdr.flags & 1) != 0;
  const size_t n = hdr.num_points;
  detail::Prefault prefault;
  sizeCloudArrays(&r, n, static_cast<size_t>(dimForDegree(hdr.sh_degree)), &prefault);
  prefault.start();
  spz_amd_cloud_out out = {r.positions.data(), r.scales.data(), r.rotations.data(),
                           r.alphas.data(),    r.colors.data(), r.sh.empty() ? nullptr : r.sh.data()};
  const int rc = spz_amd_decode_host(stream, size, static_cast<int>(o.to), &out, deviceIndex());
  prefault.join();
  if (deviceFailed(rc, "decode")) return {};
  return r;
}

PackedGaussians packGaussians(const GaussianCloud &g, const PackOptions &o) {
  if (!checkSizes(g)) return {};
  std::vector<uint8_t> stream;
  if (!packToStream(g, o, &stream)) return {};
  spz_amd_header hdr;
  if (spz_amd_peek_header_ex(stream.data(), stream.size(), 0, &hdr) != SPZ_AMD_OK) return {};
  return slicePacked(stream.data(), hdr);
}

GaussianCloud unpackGaussians(const PackedGaussians &packed, const UnpackOptions &o) {
  g_last_status = SPZ_AMD_OK;
  const int32_t shDim = dimForDegree(packed.shDegree);
  const bool f16 = packed.usesFloat16();
  if (!checkSizes(packed, packed.numPoints, shDim, f16)) return {};
  // Re-assemble the stream (host memcpy) so that one fused decode launch handles it.
  spz_amd_header h = {};
  h.version = f16 ? 1u : (packed.usesQuaternionSmallestThree ? 3u : 2u);
  if (packed.numPoints == 0) h.version = packed.usesQuaternionSmallestThree ? 3u : 2u;
  h.num_points = static_cast<uint32_t>(packed.numPoints);
  h.sh_degree = static_cast<uint8_t>(packed.shDegree);
  h.fractional_bits = static_cast<uint8_t>(packed.fractionalBits);
  h.flags = packed.antialiased ? 1 : 0;
  std::vector<uint8_t> stream(16);
  spz_amd_write_header(&h, stream.data());
  stream.reserve(16 + packed.positions.size() + packed.alphas.size() + packed.colors.size() + packed.scales.size() +
                 packed.rotations.size() + packed.sh.size());
  for (const auto *v : {&packed.positions, &packed.alphas, &packed.colors, &packed.scales, &packed.rotations,
                        &packed.sh}) {
    stream.insert(stream.end(), v->begin(), v->end());
  }
  spz_amd_header hdr;
  const int prc = spz_amd_peek_header_ex(stream.data(), stream.size(), 0, &hdr);
  if (prc != SPZ_AMD_OK) {
    deviceFailed(prc, "unpackGaussians");
    return {};
  }
  GaussianCloud r;
  r.numPoints = packed.numPoints;
  r.shDegree = packed.shDegree;
  r.antialiased = packed.antialiased;
  const size_t n = static_cast<size_t>(packed.numPoints);
  if (n == 0) return r;
  detail::Prefault prefault;
  sizeCloudArrays(&r, n, static_cast<size_t>(shDim), &prefault);
  prefault.start();
  spz_amd_cloud_out out = {r.positions.data(), r.scales.data(), r.rotations.data(),
                           r.alphas.data(),    r.colors.data(), r.sh.empty() ? nullptr : r.sh.data()};
  // no point limit here: the reference's 10 M cap lives in deserializePackedGaussians (load-spz.cc:549,561),
  // unpackGaussians (:467-531) has none
  const int rc = spz_amd_decode_host_ex(stream.data(), stream.size(), 0, static_cast<int>(o.to), &out, deviceIndex());
  prefault.join();
  if (deviceFailed(rc, "decode")) return {};
  return r;
}

GaussianCloud unpackIndices(const PackedGaussians &packed, const std::vector<uint32_t> &indices,
                            const UnpackOptions &o) {
  g_last_status = SPZ_AMD_OK;
  const int32_t shDim = dimForDegree(packed.shDegree);
  const bool f16 = packed.usesFloat16();
  if (!checkSizes(packed, packed.numPoints, shDim, f16)) return {};
  GaussianCloud r;
  r.shDegree = packed.shDegree;
  r.antialiased = packed.antialiased;
  if (indices.empty()) return r;
  if (packed.numPoints == 0) {
    logLine("[SPZ ERROR] spz_amd: unpackIndices: the packed cloud is empty");
    return {};
  }
  spz_amd_header h = {};
  h.version = f16 ? 1u : (packed.usesQuaternionSmallestThree ? 3u : 2u);
  h.num_points = static_cast<uint32_t>(packed.numPoints);
  h.sh_degree = static_cast<uint8_t>(packed.shDegree);
  h.fractional_bits = static_cast<uint8_t>(packed.fractionalBits);
  h.flags = packed.antialiased ? 1 : 0;
  std::vector<uint8_t> stream(16);
  spz_amd_write_header(&h, stream.data());
  for (const auto *v : {&packed.positions, &packed.alphas, &packed.colors, &packed.scales, &packed.rotations,
                        &packed.sh}) {
    stream.insert(stream.end(), v->begin(), v->end());
  }
  const size_t n = indices.size();
  r.numPoints = static_cast<int32_t>(n);
  r.positions.resize(n * 3);
  r.scales.resize(n * 3);
  r.rotations.resize(n * 4);
  r.alphas.resize(n);
  r.colors.resize(n * 3);
  r.sh.resize(n * shDim * 3);
  spz_amd_cloud_out out = {r.positions.data(), r.scales.data(), r.rotations.data(),
                           r.alphas.data(),    r.colors.data(), r.sh.empty() ? nullptr : r.sh.data()};
  const int rc = spz_amd_decode_gather_host(stream.data(), stream.size(), 0, indices.data(), n, static_cast<int>(o.to),
                                            &out, deviceIndex());
  if (deviceFailed(rc, "unpackIndices")) return {};
  return r;
}

// ---- saveSpz / loadSpz overloads, load-spz.cc:598-668 ---------------------------------------------
namespace {
std::mutex g_stream_cache_mutex;
std::vector<uint8_t> g_stream_cache;
}  // namespace

bool saveSpz(const GaussianCloud &g, const PackOptions &o, std::vector<uint8_t> *out) {
  static const bool timing = std::getenv("SPZ_AMD_EXACT_GZIP_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  struct Report {
    std::chrono::steady_clock::time_point t0;
    bool on;
    ~Report() {
      if (on) std::fprintf(stderr, "[saveSpz] %.3f s in all (stream freed)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
  } report{t0, timing};
  // The stream is a transient of this call, 65 bytes per Gaussian.  Giving such a buffer back costs ~80 ms per GB once
  // device copies have pinned it (measured: 52 ms of a 0.36 s save), in the foreground or — contending with
  // whatever the caller does next — in the background; so one buffer of up to 1 GiB is kept for the next save
  // (spz::releaseHostMemory() drops it).  A second save at the same time uses a buffer of its own.
  std::vector<uint8_t> local;
  static const bool keep = [] {
    const char *e = std::getenv("SPZ_AMD_KEEP_STREAM_BUFFER");
    return !(e && e[0] == '0');
  }();
  std::unique_lock<std::mutex> cache_lock(g_stream_cache_mutex, std::defer_lock);
  if (keep) (void)cache_lock.try_lock();
  std::vector<uint8_t> &stream = cache_lock.owns_lock() ? g_stream_cache : local;
  struct Trim {
    std::vector<uint8_t> &v;
    bool cached;
    ~Trim() {
      if (cached && v.capacity() > (size_t(1) << 30)) std::vector<uint8_t>().swap(v);
    }
  } trim{stream, cache_lock.owns_lock()};
  // the stream stays on the device as well when its container stage is going to run there: no second upload
  const uint8_t *d_copy = nullptr;
  struct Kept {
    const uint8_t **p;
    ~Kept() { spz_amd_kept_stream_release(deviceIndex(), *p); }
  } kept{&d_copy};
  spz_amd_layout lay;
  const bool keep_on_device = g.numPoints > 0 && spz_amd_stream_layout(static_cast<uint64_t>(g.numPoints), g.shDegree, 3, &lay) == SPZ_AMD_OK &&
                              lay.total_bytes >= (size_t(1) << 20) && exactGzipThreads() >= 1 && deviceParseWanted(lay.total_bytes);
  // Default: the reference's single deflate stream (byte-identical files).  SPZ_AMD_GZIP_THREADS=n>1
  // opts into the parallel container (same content, different bytes, n x faster).
  const char *e = std::getenv("SPZ_AMD_GZIP_THREADS");
  const int threads = e ? std::atoi(e) : 1;
  // ... and its first stages (hash chains, match tables: a pure function of the stream's bytes) run on the finished
  // sections while the rest of the floats still upload: SPZ_AMD_GZIP_OVERLAP=0 starts them after the pack, as before
  SaveAhead ahead;
  static const bool overlap = [] {
    const char *v = std::getenv("SPZ_AMD_GZIP_OVERLAP");
    return !(v && v[0] == '0');
  }();
  if (keep_on_device && overlap && threads <= 1 && std::strcmp(zlibVersion(), "1.2.11") == 0) {
    const size_t avail = availablePhysicalBytes();
    if ((avail == 0 || lay.total_bytes / 2 * 9 < avail) &&
        spz_amd_zlib_session_open(lay.total_bytes, deviceIndex(), &ahead.session) != SPZ_AMD_OK) {
      ahead.session = nullptr;  // declined (size, memory): the stage starts after the pack, or runs on the host
    }
  }
  if (!packToStreamKeep(g, o, &stream, keep_on_device ? &d_copy : nullptr, &ahead)) return false;
  if (timing) std::fprintf(stderr, "[saveSpz] pack %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  if (threads <= 1) return compressGzippedWithCopy(stream.data(), stream.size(), out, d_copy, &ahead);
  return compressGzippedParallel(stream.data(), stream.size(), out, threads);
}

void releaseHostMemory() {
  std::lock_guard<std::mutex> lock(g_stream_cache_mutex);
  std::vector<uint8_t>().swap(g_stream_cache);
}

bool saveSpz(const GaussianCloud &g, const PackOptions &o, const std::string &filename) {
  std::vector<uint8_t> data;
  if (!saveSpz(g, o, &data)) return false;
  // The file is opened only after encoding succeeded; result is out.good() (load-spz.cc:647-650).
  return writeFile(filename, data);
}

PackedGaussians loadSpzPacked(const uint8_t *data, int32_t size) {
  std::vector<uint8_t> stream;
  if (!decompressGzipped(data, static_cast<size_t>(size), &stream)) return {};
  spz_amd_header hdr;
  if (!peekHeaderLogged(stream.data(), stream.size(), &hdr)) return {};
  return slicePacked(stream.data(), hdr);
}

PackedGaussians loadSpzPacked(const std::vector<uint8_t> &data) {
  return loadSpzPacked(data.data(), static_cast<int>(data.size()));
}

PackedGaussians loadSpzPacked(const std::string &filename) {
  std::vector<uint8_t> data;
  if (!readFile(filename, &data, /*log=*/false)) return {};
  return loadSpzPacked(data);
}

// ---- device-resident packed load -----------------------------------------------------------------------------
DevicePackedGaussians::DevicePackedGaussians(DevicePackedGaussians &&o) noexcept { *this = std::move(o); }

DevicePackedGaussians &DevicePackedGaussians::operator=(DevicePackedGaussians &&o) noexcept {
  if (this == &o) return *this;
  release();
  numPoints = o.numPoints;
  shDegree = o.shDegree;
  fractionalBits = o.fractionalBits;
  antialiased = o.antialiased;
  usesQuaternionSmallestThree = o.usesQuaternionSmallestThree;
  version = o.version;
  positions = o.positions; alphas = o.alphas; colors = o.colors; scales = o.scales; rotations = o.rotations; sh = o.sh;
  positionsBytes = o.positionsBytes; alphasBytes = o.alphasBytes; colorsBytes = o.colorsBytes;
  scalesBytes = o.scalesBytes; rotationsBytes = o.rotationsBytes; shBytes = o.shBytes;
  stream = o.stream;
  streamBytes = o.streamBytes;
  device = o.device;
  inflatedOnDevice = o.inflatedOnDevice;
  owner_ = o.owner_;
  o.owner_ = nullptr;
  o.release();
  return *this;
}

DevicePackedGaussians::~DevicePackedGaussians() { release(); }

void DevicePackedGaussians::release() {
  if (owner_ != nullptr) spz_amd_inflate_close(owner_);
  owner_ = nullptr;
  numPoints = shDegree = fractionalBits = 0;
  antialiased = false;
  usesQuaternionSmallestThree = true;
  version = 0;
  positions = alphas = colors = scales = rotations = sh = stream = nullptr;
  positionsBytes = alphasBytes = colorsBytes = scalesBytes = rotationsBytes = shBytes = streamBytes = 0;
  inflatedOnDevice = false;
}

namespace {
spz_amd_header headerOf(const DevicePackedGaussians &p) {
  spz_amd_header h = {};
  h.version = p.version;
  h.num_points = static_cast<uint32_t>(p.numPoints);
  h.sh_degree = static_cast<uint8_t>(p.shDegree);
  h.fractional_bits = static_cast<uint8_t>(p.fractionalBits);
  h.flags = p.antialiased ? 1 : 0;
  return h;
}
}  // namespace

GaussianCloud DevicePackedGaussians::unpack(const UnpackOptions &o) const {
  g_last_status = SPZ_AMD_OK;
  if (!valid()) return {};
  const spz_amd_header hdr = headerOf(*this);
  GaussianCloud r;
  r.numPoints = numPoints;
  r.shDegree = shDegree;
  r.antialiased = antialiased;
  detail::Prefault prefault;
  sizeCloudArrays(&r, hdr.num_points, static_cast<size_t>(dimForDegree(shDegree)), &prefault);
  prefault.start();
  spz_amd_cloud_out out = {r.positions.data(), r.scales.data(), r.rotations.data(),
                           r.alphas.data(),    r.colors.data(), r.sh.empty() ? nullptr : r.sh.data()};
  const int rc = spz_amd_decode_host_from_device(stream, streamBytes, &hdr, static_cast<int>(o.to), &out, device);
  prefault.join();
  if (deviceFailed(rc, "DevicePackedGaussians::unpack")) return {};
  return r;
}

GaussianCloud DevicePackedGaussians::unpackIndices(const std::vector<uint32_t> &indices, const UnpackOptions &o) const {
  g_last_status = SPZ_AMD_OK;
  if (!valid()) return {};
  GaussianCloud r;
  r.shDegree = shDegree;
  r.antialiased = antialiased;
  if (indices.empty()) return r;
  if (numPoints == 0) {
    logLine("[SPZ ERROR] spz_amd: unpackIndices: the packed cloud is empty");
    return {};
  }
  const spz_amd_header hdr = headerOf(*this);
  const size_t n = indices.size(), shDim = static_cast<size_t>(dimForDegree(shDegree));
  r.numPoints = static_cast<int32_t>(n);
  r.positions.resize(n * 3);
  r.scales.resize(n * 3);
  r.rotations.resize(n * 4);
  r.alphas.resize(n);
  r.colors.resize(n * 3);
  r.sh.resize(n * shDim * 3);
  spz_amd_cloud_out out = {r.positions.data(), r.scales.data(), r.rotations.data(),
                           r.alphas.data(),    r.colors.data(), r.sh.empty() ? nullptr : r.sh.data()};
  const int rc = spz_amd_decode_gather_host_from_device(stream, streamBytes, &hdr, indices.data(), n, static_cast<int>(o.to), &out, device);
  if (deviceFailed(rc, "DevicePackedGaussians::unpackIndices")) return {};
  return r;
}

DevicePackedGaussians loadSpzPackedDevice(const uint8_t *data, int32_t size) {
  g_last_status = SPZ_AMD_OK;
  DevicePackedGaussians r;
  if (data == nullptr || size <= 0) return r;
  GzipIndex idx;
  const size_t headerLen = parseGzipHeader(data, static_cast<size_t>(size), &idx);
  uint64_t stream_bytes = 0;
  void *ctx = (headerLen != 0 && idx.pieceBytes.empty()) ? openVerifiedDeviceInflate(data, static_cast<size_t>(size), headerLen, &stream_bytes)
                                                         : nullptr;
  bool on_device = ctx != nullptr;
  if (ctx == nullptr) {  // the host readers (same bytes), then one upload
    std::vector<uint8_t> stream;
    if (!decompressGzippedWith(data, static_cast<size_t>(size), &stream, false)) return r;  // silently, load-spz.cc:609-612
    spz_amd_header hdr;
    if (!peekHeaderLogged(stream.data(), stream.size(), &hdr)) return r;
    const int rc = spz_amd_stream_to_device(stream.data(), stream.size(), deviceIndex(), &ctx);
    if (deviceFailed(rc, "loadSpzPackedDevice")) return r;
    stream_bytes = stream.size();
  }
  struct Close {
    void *c;
    ~Close() {
      if (c) spz_amd_inflate_close(c);
    }
  } closer{ctx};
  const uint8_t *d_stream = spz_amd_inflate_device_data(ctx);
  uint8_t first16[16] = {};
  spz_amd_header hdr;
  const int prc = stream_bytes >= 16 ? spz_amd_peek_header_device(d_stream, static_cast<size_t>(stream_bytes), 0, &hdr, nullptr) : SPZ_AMD_ERR_HEADER_NOT_FOUND;
  if (prc != SPZ_AMD_OK || spz_amd_write_header(&hdr, first16) != SPZ_AMD_OK) {
    // a stream the device inflated but whose header is not one: the reference's log line from the host-side check
    std::vector<uint8_t> whole;
    detail::resizeUninitialized(&whole, static_cast<size_t>(stream_bytes));
    if (spz_amd_inflate_fetch(ctx, whole.data()) == SPZ_AMD_OK) (void)peekHeaderLogged(whole.data(), whole.size(), &hdr);
    return r;
  }
  if (!peekHeaderLogged(first16, static_cast<size_t>(stream_bytes), &hdr)) return r;  // limits and short streams, with the log lines
  spz_amd_layout lay;
  if (spz_amd_stream_layout(hdr.num_points, hdr.sh_degree, static_cast<int>(hdr.version), &lay) != SPZ_AMD_OK) return r;
  r.numPoints = static_cast<int32_t>(hdr.num_points);
  r.shDegree = hdr.sh_degree;
  r.fractionalBits = hdr.fractional_bits;
  r.antialiased = (hdr.flags & 1) != 0;
  r.version = hdr.version;
  r.usesQuaternionSmallestThree = hdr.version >= 3;
  auto sec = [&](int s) { return lay.bytes[s] ? d_stream + lay.offset[s] : nullptr; };
  r.positions = sec(SPZ_AMD_SEC_POSITIONS);
  r.alphas = sec(SPZ_AMD_SEC_ALPHAS);
  r.colors = sec(SPZ_AMD_SEC_COLORS);
  r.scales = sec(SPZ_AMD_SEC_SCALES);
  r.rotations = sec(SPZ_AMD_SEC_ROTATIONS);
  r.sh = sec(SPZ_AMD_SEC_SH);
  r.positionsBytes = lay.bytes[SPZ_AMD_SEC_POSITIONS];
  r.alphasBytes = lay.bytes[SPZ_AMD_SEC_ALPHAS];
  r.colorsBytes = lay.bytes[SPZ_AMD_SEC_COLORS];
  r.scalesBytes = lay.bytes[SPZ_AMD_SEC_SCALES];
  r.rotationsBytes = lay.bytes[SPZ_AMD_SEC_ROTATIONS];
  r.shBytes = lay.bytes[SPZ_AMD_SEC_SH];
  r.stream = d_stream;
  r.streamBytes = static_cast<size_t>(stream_bytes);
  r.device = deviceIndex();
  r.inflatedOnDevice = on_device;
  r.owner_ = ctx;
  closer.c = nullptr;
  if (on_device) g_device_inflates.fetch_add(1);
  return r;
}

DevicePackedGaussians loadSpzPackedDevice(const std::vector<uint8_t> &data) {
  return loadSpzPackedDevice(data.data(), static_cast<int32_t>(data.size()));
}

DevicePackedGaussians loadSpzPackedDevice(const std::string &filename) {
  std::vector<uint8_t> data;
  if (!readFile(filename, &data, /*log=*/false)) return {};
  return loadSpzPackedDevice(data);
}

GaussianCloud loadSpz(const uint8_t *data, int32_t size, const UnpackOptions &o) {
  g_last_status = SPZ_AMD_OK;
  static const bool timing = std::getenv("SPZ_AMD_EXACT_GZIP_TIMING") != nullptr;
  const auto t_load = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (timing) std::fprintf(stderr, "[loadSpz] %-22s %.3f s since the call\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load).count());
  };
  // An ordinary member that the device inflates stays there: the decode kernels read the stream where it is and only
  // the floats cross PCIe.
  bool device_declined = false;
  if (data != nullptr && size > 0) {
    GzipIndex idx;
    const size_t headerLen = parseGzipHeader(data, static_cast<size_t>(size), &idx);
    uint64_t stream_bytes = 0;
    // The output arrays can be sized — and their pages mapped, 25-30 ms for a 10 M-point cloud — while the device
    // inflates: the stream's first 16 bytes say how many points are coming, and zlib has them in microseconds.  Only
    // for members the device reader is going to be asked about; what it finds later must agree with this peek.
    GaussianCloud r;
    detail::Prefault prefault;
    spz_amd_header early = {};
    bool sized_early = false;
    if (headerLen != 0 && idx.pieceBytes.empty() && deviceInflateWanted(static_cast<size_t>(size), headerLen)) {
      uint8_t first[16];
      if (inflatePrefix(data + headerLen, static_cast<size_t>(size) - headerLen, first, sizeof(first)) == sizeof(first)) {
        // the header's own fields (spz_amd_peek_header would also want the whole stream behind them); the gzip trailer's
        // ISIZE must be what a stream of that many points takes, or the peek is not believed
        auto u32 = [&](const uint8_t *q) {
          return static_cast<uint32_t>(q[0]) | (static_cast<uint32_t>(q[1]) << 8) | (static_cast<uint32_t>(q[2]) << 16) |
                 (static_cast<uint32_t>(q[3]) << 24);
        };
        early.version = u32(first + 4);
        early.num_points = u32(first + 8);
        early.sh_degree = first[12];
        spz_amd_layout lay;
        if (u32(first) == 0x5053474eu && early.version >= 1 && early.version <= 3 && early.sh_degree <= 3 &&
            early.num_points > 0 && early.num_points <= SPZ_AMD_REFERENCE_MAX_POINTS &&
            spz_amd_stream_layout(early.num_points, early.sh_degree, static_cast<int>(early.version), &lay) == SPZ_AMD_OK &&
            static_cast<uint32_t>(lay.total_bytes & 0xffffffffull) == u32(data + size - 4)) {
          sizeCloudArrays(&r, early.num_points, static_cast<size_t>(dimForDegree(early.sh_degree)), &prefault);
          sized_early = true;  // mapped once the member is on the device: beside the upload the two contend (+80 ms)
        }
      }
      lap("sized from a peek");
    }
    auto map_pages = [](void *p) { static_cast<detail::Prefault *>(p)->startBeside(); };
    void *ctx = (headerLen != 0 && idx.pieceBytes.empty())
                    ? openVerifiedDeviceInflate(data, static_cast<size_t>(size), headerLen, &stream_bytes,
                                                sized_early ? +map_pages : nullptr, &prefault)
                    : nullptr;
    if (ctx != nullptr) {
      lap("inflated on the device");
      struct Close {
        void *c;
        ~Close() { spz_amd_inflate_close(c); }
      } closer{ctx};
      const uint8_t *d_stream = spz_amd_inflate_device_data(ctx);
      uint8_t first16[16] = {};
      spz_amd_header hdr;
      if (stream_bytes >= 16 && spz_amd_peek_header_device(d_stream, static_cast<size_t>(stream_bytes), 0, &hdr, nullptr) == SPZ_AMD_OK &&
          spz_amd_write_header(&hdr, first16) == SPZ_AMD_OK) {
        // the reference's checks and log lines (load-spz.cc:553-568), on the 16 header bytes and the stream's size
        if (!peekHeaderLogged(first16, static_cast<size_t>(stream_bytes), &hdr)) return {};
        r.numPoints = static_cast<int32_t>(hdr.num_points);
        r.shDegree = hdr.sh_degree;
        r.antialiased = (hdr.flags & 1) != 0;
        prefault.join();   // mapped before the downloads begin (beside them the two contend: spz_host_util.hpp)
        lap("pages mapped");
        detail::Prefault late;
        if (!sized_early || early.num_points != hdr.num_points || early.sh_degree != hdr.sh_degree) {
          sizeCloudArrays(&r, hdr.num_points, static_cast<size_t>(dimForDegree(hdr.sh_degree)), &late);
          late.start();
        }
        spz_amd_cloud_out out = {r.positions.data(), r.scales.data(), r.rotations.data(),
                                 r.alphas.data(),    r.colors.data(), r.sh.empty() ? nullptr : r.sh.data()};
        const int rc = spz_amd_decode_host_from_device(d_stream, static_cast<size_t>(stream_bytes), &hdr, static_cast<int>(o.to),
                                                       &out, deviceIndex());
        late.join();
        lap("decoded and downloaded");
        if (rc == SPZ_AMD_OK) {
          g_device_inflates.fetch_add(1);
          return r;
        }
      }
      // anything unusual about the stream (bad header, short stream): the reference's log line and result come from
      // the ordinary decode of the bytes this context already holds — the member is not inflated a second time
      std::vector<uint8_t> inflated;
      detail::resizeUninitialized(&inflated, static_cast<size_t>(stream_bytes));
      if (spz_amd_inflate_fetch(ctx, inflated.data()) == SPZ_AMD_OK) return unpackFromStream(inflated.data(), inflated.size(), o);
      device_declined = true;
    } else if (headerLen != 0 && idx.pieceBytes.empty()) {
      device_declined = true;  // not asked a second time by decompressGzipped
    }
  }
  std::vector<uint8_t> stream;
  // A failed gunzip yields an empty PackedGaussians and hence an empty cloud, silently
  // (load-spz.cc:609-612).
  if (!decompressGzippedWith(data, static_cast<size_t>(size), &stream, !device_declined)) return {};
  return unpackFromStream(stream.data(), stream.size(), o);
}

GaussianCloud loadSpz(const std::vector<uint8_t> &data, const UnpackOptions &o) {
  return loadSpz(data.data(), static_cast<int32_t>(data.size()), o);
}

GaussianCloud loadSpz(const std::string &filename, const UnpackOptions &o) {
  g_last_status = SPZ_AMD_OK;
  std::vector<uint8_t> data;
  if (!readFile(filename, &data, /*log=*/true)) return {};
  return loadSpz(data, o);
}

}  // namespace spz
