// ref_shim.cc — extern "C" handles onto the REFERENCE's own compiled C++.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled TOGETHER with the
// reference's sources where they lie (/root/reference/src/cc/*.cc, never copied
// into this repo) by oracle/Makefile into oracle/_ref/libspz_ref.so.  It is used
//   * to validate oracle/spz_oracle.c (the restatement) bit-for-bit,
//   * to generate the golden vectors in tests/golden/ (make_golden.py),
//   * as bench.py's cpu_baseline ("kind": "reference") on the GPU box.
// The product (spz_amd/csrc) never loads it.
//
// packGaussians / unpackGaussians / deserializePackedGaussians and the quaternion
// helpers have external linkage in load-spz.cc but are not declared in
// load-spz.h, so they are forward-declared here with the reference's signatures
// (load-spz.cc:216,257,333,347,467,548).
#include <chrono>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "src/cc/load-spz.h"
#include "src/cc/splat-types.h"

namespace spz {
PackedGaussians packGaussians(const GaussianCloud &g, const PackOptions &o);
GaussianCloud unpackGaussians(const PackedGaussians &packed, const UnpackOptions &o);
PackedGaussians deserializePackedGaussians(std::istream &in);
void packQuaternionSmallestThree(uint8_t r[4], const float rotation[4], const CoordinateConverter &c);
void unpackQuaternionFirstThree(float rotation[4], const uint8_t r[3], const CoordinateConverter &c);
void unpackQuaternionSmallestThree(float rotation[4], const uint8_t r[4], const CoordinateConverter &c);
}  // namespace spz

namespace {

int dimForDegree(int d) { return d == 1 ? 3 : d == 2 ? 8 : d == 3 ? 15 : 0; }

spz::GaussianCloud makeCloud(const float *pos, const float *scales, const float *rot,
                             const float *alphas, const float *colors, const float *sh,
                             int32_t n, int shDegree, int antialiased) {
  spz::GaussianCloud g;
  g.numPoints = n;
  g.shDegree = shDegree;
  g.antialiased = antialiased != 0;
  const size_t N = n;
  g.positions.assign(pos, pos + N * 3);
  g.scales.assign(scales, scales + N * 3);
  g.rotations.assign(rot, rot + N * 4);
  g.alphas.assign(alphas, alphas + N);
  g.colors.assign(colors, colors + N * 3);
  const size_t shN = N * dimForDegree(shDegree) * 3;
  if (shN) g.sh.assign(sh, sh + shN);
  return g;
}

void copyOut(const spz::GaussianCloud &g, float *pos, float *scales, float *rot, float *alphas,
             float *colors, float *sh) {
  auto cp = [](float *dst, const std::vector<float> &v) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(float));
  };
  cp(pos, g.positions);
  cp(scales, g.scales);
  cp(rot, g.rotations);
  cp(alphas, g.alphas);
  cp(colors, g.colors);
  cp(sh, g.sh);
}

double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

// packGaussians + serializePackedGaussians -> raw (pre-gzip) stream. Returns bytes written
// (0 if capacity too small).
size_t ref_pack(const float *pos, const float *scales, const float *rot, const float *alphas,
                const float *colors, const float *sh, int32_t n, int shDegree, int antialiased,
                int from, uint8_t *out, size_t capacity) {
  spz::GaussianCloud g = makeCloud(pos, scales, rot, alphas, colors, sh, n, shDegree, antialiased);
  spz::PackOptions o;
  o.from = static_cast<spz::CoordinateSystem>(from);
  spz::PackedGaussians packed = spz::packGaussians(g, o);
  std::stringstream ss;
  spz::serializePackedGaussians(packed, &ss);
  const std::string data = ss.str();
  if (data.size() > capacity) return 0;
  std::memcpy(out, data.data(), data.size());
  return data.size();
}

// deserializePackedGaussians + unpackGaussians on a raw stream.
// info[0..2] = numPoints, shDegree, antialiased.  Returns 0, or -1 if the reference returned an
// empty PackedGaussians for a non-trivial stream (its error convention).
int ref_unpack(const uint8_t *stream, size_t size, int to, float *pos, float *scales, float *rot,
               float *alphas, float *colors, float *sh, int32_t *info) {
  std::stringstream ss(std::string(reinterpret_cast<const char *>(stream), size));
  spz::PackedGaussians packed = spz::deserializePackedGaussians(ss);
  spz::UnpackOptions o;
  o.to = static_cast<spz::CoordinateSystem>(to);
  spz::GaussianCloud g = spz::unpackGaussians(packed, o);
  if (info) {
    info[0] = g.numPoints;
    info[1] = g.shDegree;
    info[2] = g.antialiased ? 1 : 0;
  }
  copyOut(g, pos, scales, rot, alphas, colors, sh);
  return 0;
}

// Full saveSpz (pack + serialize + gzip) to memory.  Returns gz size, 0 on failure/too small.
size_t ref_save_spz(const float *pos, const float *scales, const float *rot, const float *alphas,
                    const float *colors, const float *sh, int32_t n, int shDegree,
                    int antialiased, int from, uint8_t *out, size_t capacity) {
  spz::GaussianCloud g = makeCloud(pos, scales, rot, alphas, colors, sh, n, shDegree, antialiased);
  spz::PackOptions o;
  o.from = static_cast<spz::CoordinateSystem>(from);
  std::vector<uint8_t> gz;
  if (!spz::saveSpz(g, o, &gz)) return 0;
  if (gz.size() > capacity) return 0;
  std::memcpy(out, gz.data(), gz.size());
  return gz.size();
}

// Full loadSpz from gz bytes in memory.
int ref_load_spz(const uint8_t *gz, int32_t size, int to, float *pos, float *scales, float *rot,
                 float *alphas, float *colors, float *sh, int32_t *info) {
  spz::UnpackOptions o;
  o.to = static_cast<spz::CoordinateSystem>(to);
  spz::GaussianCloud g = spz::loadSpz(gz, size, o);
  if (info) {
    info[0] = g.numPoints;
    info[1] = g.shDegree;
    info[2] = g.antialiased ? 1 : 0;
  }
  copyOut(g, pos, scales, rot, alphas, colors, sh);
  return 0;
}

size_t ref_compress_gzipped(const uint8_t *data, size_t size, uint8_t *out, size_t capacity) {
  std::vector<uint8_t> gz;
  if (!spz::compressGzipped(data, size, &gz) || gz.size() > capacity) return 0;
  std::memcpy(out, gz.data(), gz.size());
  return gz.size();
}

void ref_converter(int from, int to, float *flipP, float *flipQ, float *flipSh) {
  spz::CoordinateConverter c = spz::coordinateConverter(static_cast<spz::CoordinateSystem>(from),
                                                        static_cast<spz::CoordinateSystem>(to));
  std::memcpy(flipP, c.flipP.data(), 3 * sizeof(float));
  std::memcpy(flipQ, c.flipQ.data(), 3 * sizeof(float));
  std::memcpy(flipSh, c.flipSh.data(), 15 * sizeof(float));
}

void ref_convert_coordinates(float *pos, float *rot, float *sh, int32_t n, int shDegree, int from,
                             int to) {
  spz::GaussianCloud g;
  g.numPoints = n;
  g.shDegree = shDegree;
  const size_t N = n;
  g.positions.assign(pos, pos + N * 3);
  g.rotations.assign(rot, rot + N * 4);
  const size_t shN = N * dimForDegree(shDegree) * 3;
  if (shN) g.sh.assign(sh, sh + shN);
  g.convertCoordinates(static_cast<spz::CoordinateSystem>(from),
                       static_cast<spz::CoordinateSystem>(to));
  copyOut(g, pos, nullptr, rot, nullptr, nullptr, sh);
}

void ref_pack_quat(const float *q, int32_t n, int from, uint8_t *out) {
  spz::CoordinateConverter c = spz::coordinateConverter(static_cast<spz::CoordinateSystem>(from),
                                                        spz::CoordinateSystem::RUB);
  for (int32_t i = 0; i < n; i++) spz::packQuaternionSmallestThree(out + 4 * i, q + 4 * i, c);
}

void ref_unpack_quat_smallest_three(const uint8_t *r, int32_t n, int to, float *out) {
  spz::CoordinateConverter c = spz::coordinateConverter(spz::CoordinateSystem::RUB,
                                                        static_cast<spz::CoordinateSystem>(to));
  for (int32_t i = 0; i < n; i++) spz::unpackQuaternionSmallestThree(out + 4 * i, r + 4 * i, c);
}

void ref_unpack_quat_first_three(const uint8_t *r, int32_t n, int to, float *out) {
  spz::CoordinateConverter c = spz::coordinateConverter(spz::CoordinateSystem::RUB,
                                                        static_cast<spz::CoordinateSystem>(to));
  for (int32_t i = 0; i < n; i++) spz::unpackQuaternionFirstThree(out + 4 * i, r + 3 * i, c);
}

float ref_half_to_float(uint16_t h) { return spz::halfToFloat(h); }

// GaussianCloud::medianVolume (splat-types.h:170-185) of a cloud that has these scales.
float ref_median_volume(const float *scales, int32_t n) {
  spz::GaussianCloud g;
  g.numPoints = n;
  g.scales.assign(scales, scales + static_cast<size_t>(n) * 3);
  return g.medianVolume();
}

// deserializePackedGaussians on a raw stream, then PackedGaussians::at(i) and ::unpack(i, converter) for every
// index: bytes65 gets 65 bytes per index (PackedGaussian field order), floats59 gets 59 floats per index
// (UnpackedGaussian field order: position, rotation, scale, color, alpha, shR, shG, shB).
int ref_packed_unpack(const uint8_t *stream, size_t size, const int32_t *indices, int32_t count, int from, int to,
                      uint8_t *bytes65, float *floats59) {
  std::stringstream ss(std::string(reinterpret_cast<const char *>(stream), size));
  spz::PackedGaussians packed = spz::deserializePackedGaussians(ss);
  if (packed.numPoints == 0) return -1;
  const spz::CoordinateConverter c = spz::coordinateConverter(static_cast<spz::CoordinateSystem>(from),
                                                              static_cast<spz::CoordinateSystem>(to));
  for (int32_t k = 0; k < count; ++k) {
    const int32_t i = indices[k];
    if (i < 0 || i >= packed.numPoints) return -2;
    const spz::PackedGaussian one = packed.at(i);
    uint8_t *b = bytes65 + static_cast<size_t>(k) * 65;
    std::memcpy(b, one.position.data(), 9);
    std::memcpy(b + 9, one.rotation.data(), 4);
    std::memcpy(b + 13, one.scale.data(), 3);
    std::memcpy(b + 16, one.color.data(), 3);
    b[19] = one.alpha;
    std::memcpy(b + 20, one.shR.data(), 15);
    std::memcpy(b + 35, one.shG.data(), 15);
    std::memcpy(b + 50, one.shB.data(), 15);
    const spz::UnpackedGaussian u = packed.unpack(i, c);
    float *f = floats59 + static_cast<size_t>(k) * 59;
    std::memcpy(f, u.position.data(), 12);
    std::memcpy(f + 3, u.rotation.data(), 16);
    std::memcpy(f + 7, u.scale.data(), 12);
    std::memcpy(f + 10, u.color.data(), 12);
    f[13] = u.alpha;
    std::memcpy(f + 14, u.shR.data(), 60);
    std::memcpy(f + 29, u.shG.data(), 60);
    std::memcpy(f + 44, u.shB.data(), 60);
  }
  return 0;
}

// saveSplatToPly / loadSplatFromPly through real files (the reference has no in-memory form).
int ref_save_ply(const float *pos, const float *scales, const float *rot, const float *alphas,
                 const float *colors, const float *sh, int32_t n, int shDegree, int from, const char *filename) {
  spz::GaussianCloud g = makeCloud(pos, scales, rot, alphas, colors, sh, n, shDegree, 0);
  spz::PackOptions o;
  o.from = static_cast<spz::CoordinateSystem>(from);
  return spz::saveSplatToPly(g, o, filename) ? 0 : -1;
}

int ref_load_ply(const char *filename, int to, float *pos, float *scales, float *rot, float *alphas, float *colors,
                 float *sh, int32_t *info) {
  spz::UnpackOptions o;
  o.to = static_cast<spz::CoordinateSystem>(to);
  spz::GaussianCloud g = spz::loadSplatFromPly(filename, o);
  if (info) {
    info[0] = g.numPoints;
    info[1] = g.shDegree;
    info[2] = static_cast<int32_t>(g.sh.size());
  }
  copyOut(g, pos, scales, rot, alphas, colors, sh);
  return 0;
}

// CPU baseline: time the reference's packGaussians and unpackGaussians (gzip-free, as SURVEY §8d)
// on caller-provided arrays.  The vector copies happen outside the timed regions.  Optionally
// returns the serialized stream (for a parity cross-check of what was timed).
int ref_bench_pack_unpack(const float *pos, const float *scales, const float *rot,
                          const float *alphas, const float *colors, const float *sh, int32_t n,
                          int shDegree, int from, int to, double *t_pack_s, double *t_unpack_s,
                          uint8_t *stream_out, size_t capacity, uint64_t *decoded_bit_sums) {
  spz::GaussianCloud g = makeCloud(pos, scales, rot, alphas, colors, sh, n, shDegree, 0);
  spz::PackOptions po;
  po.from = static_cast<spz::CoordinateSystem>(from);
  spz::UnpackOptions uo;
  uo.to = static_cast<spz::CoordinateSystem>(to);
  double t0 = now();
  spz::PackedGaussians packed = spz::packGaussians(g, po);
  double t1 = now();
  spz::GaussianCloud back = spz::unpackGaussians(packed, uo);
  double t2 = now();
  *t_pack_s = t1 - t0;
  *t_unpack_s = t2 - t1;
  if (back.numPoints != n) return -1;
  if (decoded_bit_sums) {
    // order-independent checksum of what unpackGaussians produced: sum of the float bit patterns per array
    const std::vector<float> *arrs[6] = {&back.positions, &back.scales, &back.rotations,
                                         &back.alphas,    &back.colors, &back.sh};
    for (int a = 0; a < 6; ++a) {
      uint64_t acc = 0;
      for (float f : *arrs[a]) {
        uint32_t b;
        std::memcpy(&b, &f, 4);
        acc += b;
      }
      decoded_bit_sums[a] = acc;
    }
  }
  if (stream_out) {
    std::stringstream ss;
    spz::serializePackedGaussians(packed, &ss);
    const std::string data = ss.str();
    if (data.size() > capacity) return -2;
    std::memcpy(stream_out, data.data(), data.size());
  }
  return 0;
}

}  // extern "C"
