// spz_ply.cpp — binary little-endian .ply reader/writer for 3DGS splats (host side).
//
// Mirrors the behaviour of loadSplatFromPly / saveSplatToPly of the reference
// (/root/reference/src/cc/load-spz.cc:670-934): same header grammar, same field names, same
// [N,C,S] <-> [N,S,C] spherical-harmonics transpose, same log lines and failure results.  This is
// the step on the far side of the hot path.  Header parsing and file I/O are
// host work; the AoS<->SoA shuffle of the vertex rows, the sh [channel][coeff]<->[coeff][channel]
// transpose and the coordinate flips run on the GPU through spz_amd_ply_rows_to_cloud_host /
// spz_amd_cloud_to_ply_rows_host (spz_ply_kernels.hip).  No CPU fallback.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "spz_amd.h"
#include "spz_amd_host.hpp"
#include "spz_host_util.hpp"

namespace spz {
namespace {

void plyLog(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vprintf(fmt, ap);
  va_end(ap);
  std::printf("\n");
  std::fflush(stdout);
}

// Next header line that is neither blank nor a comment, with leading whitespace removed
// (load-spz.cc:670-689).
bool nextHeaderLine(std::istream &in, std::string *line) {
  std::string raw;
  while (std::getline(in, raw)) {
    const size_t start = raw.find_first_not_of(" \t\n\r\f\v");
    if (start == std::string::npos) continue;
    if (raw.compare(start, 7, "comment") == 0) continue;
    *line = raw.substr(start);
    return true;
  }
  return false;
}

int plyDevice() {
  const char *e = std::getenv("SPZ_AMD_DEVICE");
  return e ? std::atoi(e) : 0;
}

bool startsWith(const std::string &s, const char *prefix) { return s.rfind(prefix, 0) == 0; }

int degreeForDim(int dim) {  // load-spz.cc:48-56
  if (dim < 3) return 0;
  if (dim < 8) return 1;
  if (dim < 15) return 2;
  return 3;
}

}  // namespace

GaussianCloud loadSplatFromPly(const std::string &filename, const UnpackOptions &o) {
  plyLog("[SPZ] Loading: %s", filename.c_str());
  std::ifstream in(filename, std::ios::binary);
  if (!in.good()) {
    plyLog("[SPZ ERROR] Unable to open: %s", filename.c_str());
    return {};
  }
  std::string line;
  std::getline(in, line);
  if (line != "ply") {
    plyLog("[SPZ ERROR] %s: not a .ply file", filename.c_str());
    return {};
  }
  if (!nextHeaderLine(in, &line) || line != "format binary_little_endian 1.0") {
    plyLog("[SPZ ERROR] %s: unsupported .ply format", filename.c_str());
    return {};
  }
  static const char kVertex[] = "element vertex ";
  if (!nextHeaderLine(in, &line) || !startsWith(line, kVertex)) {
    plyLog("[SPZ ERROR] %s: missing vertex count", filename.c_str());
    return {};
  }
  int32_t numPoints = 0;
  try {
    numPoints = std::stoi(line.substr(sizeof(kVertex) - 1));
  } catch (...) {
    numPoints = 0;  // the reference lets std::stoi throw here; report it as an invalid count instead
  }
  if (numPoints <= 0 || numPoints > 10 * 1024 * 1024) {
    plyLog("[SPZ ERROR] %s: invalid vertex count: %d", filename.c_str(), numPoints);
    return {};
  }
  plyLog("[SPZ] Loading %d points", numPoints);

  static const char kProp[] = "property float ";
  std::map<std::string, int> column;  // property name -> column index
  for (int i = 0;; ++i) {
    if (!nextHeaderLine(in, &line)) {
      plyLog("[SPZ ERROR] %s: unexpected EOF while reading header properties.", filename.c_str());
      return {};
    }
    if (line == "end_header") break;
    if (!startsWith(line, kProp)) {
      plyLog("[SPZ ERROR] %s: unsupported property data type: %s", filename.c_str(), line.c_str());
      return {};
    }
    column[line.substr(sizeof(kProp) - 1)] = i;
  }
  const size_t stride = column.size();

  bool missing = false;
  auto col = [&](const char *name) {
    auto it = column.find(name);
    if (it == column.end()) {
      plyLog("[SPZ ERROR] Missing field: %s", name);
      missing = true;
      return -1;
    }
    return it->second;
  };
  const int cPos[3] = {col("x"), col("y"), col("z")};
  const int cScale[3] = {col("scale_0"), col("scale_1"), col("scale_2")};
  const int cRot[4] = {col("rot_1"), col("rot_2"), col("rot_3"), col("rot_0")};  // file is wxyz, cloud xyzw
  const int cAlpha = col("opacity");
  const int cColor[3] = {col("f_dc_0"), col("f_dc_1"), col("f_dc_2")};
  if (missing) return {};

  std::vector<int> cRest;  // f_rest_0.. consecutive, optional
  for (int i = 0; i < 45; ++i) {
    auto it = column.find("f_rest_" + std::to_string(i));
    if (it == column.end()) break;
    cRest.push_back(it->second);
  }
  const int shDim = static_cast<int>(cRest.size() / 3);

  // The vertex rows (load-spz.cc:803-812: one read into a resized vector).  Large files — 2.5 GB for 10 M SH3 points,
  // whose zero fill and single read() took 1.4 s where the device needs 0.1 — go into a buffer that is sized without
  // being written and mapped with huge pages, by pread from several threads.
  std::vector<float> rows;
  const size_t row_bytes = static_cast<size_t>(numPoints) * stride * sizeof(float);
  const std::streamoff data_at = in.tellg();
  bool have_rows = false;
  if (row_bytes >= detail::kParallelIoMin && detail::fileIoThreads() > 0 && data_at > 0) {
    const int fd = ::open(filename.c_str(), O_RDONLY | O_CLOEXEC);
    struct stat st;
    if (fd >= 0 && ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
      detail::resizeUninitialized(&rows, row_bytes / sizeof(float));
      {
        detail::Prefault pf;
        pf.add(rows.data(), row_bytes);
        pf.start();
        pf.join();
      }
      const bool ok = detail::parallelRead(fd, reinterpret_cast<uint8_t *>(rows.data()), row_bytes, static_cast<size_t>(data_at));
      ::close(fd);
      if (!ok) {
        plyLog("[SPZ ERROR] Unable to load data from: %s", filename.c_str());
        return {};
      }
      have_rows = true;
    } else if (fd >= 0) {
      ::close(fd);
    }
  }
  if (!have_rows) {
    rows.resize(static_cast<size_t>(numPoints) * stride);
    in.read(reinterpret_cast<char *>(rows.data()), static_cast<std::streamsize>(rows.size() * sizeof(float)));
    if (!in.good()) {
      plyLog("[SPZ ERROR] Unable to load data from: %s", filename.c_str());
      return {};
    }
  }

  if (stride > 255) {
    plyLog("[SPZ ERROR] spz_amd: %s: more than 255 properties per vertex are not supported", filename.c_str());
    return {};
  }
  spz_amd_ply_columns cols = {};
  cols.stride = static_cast<int32_t>(stride);
  cols.sh_dim = shDim;
  for (int k = 0; k < 3; ++k) {
    cols.position[k] = cPos[k];
    cols.scale[k] = cScale[k];
    cols.color[k] = cColor[k];
  }
  for (int k = 0; k < 4; ++k) cols.rotation[k] = cRot[k];
  cols.alpha = cAlpha;
  for (int k = 0; k < shDim * 3; ++k) cols.sh[k] = cRest[k];

  GaussianCloud g;
  g.numPoints = numPoints;
  g.shDegree = degreeForDim(shDim);
  const size_t n = static_cast<size_t>(numPoints);
  {  // every element is written by the device copy: no zero fill, pages mapped by several threads
    std::vector<float> *arrays[6] = {&g.positions, &g.scales, &g.rotations, &g.alphas, &g.colors, &g.sh};
    const size_t counts[6] = {n * 3, n * 3, n * 4, n, n * 3, n * static_cast<size_t>(shDim) * 3};
    detail::Prefault pf;
    for (int i = 0; i < 6; ++i) {
      detail::resizeUninitialized(arrays[i], counts[i]);
      pf.add(arrays[i]->data(), counts[i] * sizeof(float));
    }
    pf.start();
    pf.join();
  }
  spz_amd_cloud_out out = {g.positions.data(), g.scales.data(), g.rotations.data(),
                           g.alphas.data(),    g.colors.data(), g.sh.empty() ? nullptr : g.sh.data()};
  // rows -> cloud with convertCoordinates(RDF, o.to) fused (load-spz.cc:814-842)
  const int rc = spz_amd_ply_rows_to_cloud_host(rows.data(), n, &cols, static_cast<int>(o.to), &out, plyDevice());
  if (rc != SPZ_AMD_OK) {
    plyLog("[SPZ ERROR] spz_amd: loadSplatFromPly: %s", spz_amd_status_string(rc));
    setLastDeviceStatus(rc);
    return {};
  }
  return g;
}

bool saveSplatToPly(const GaussianCloud &data, const PackOptions &o, const std::string &filename) {
  const int32_t N = data.numPoints;
  const size_t n = static_cast<size_t>(N < 0 ? 0 : N);
  auto sizeOk = [&](const std::vector<float> &v, size_t want, const char *what) {
    if (v.size() == want) return true;
    plyLog("[SPZ: ERROR] Check failed: %s:%d: %s", __FILE__, __LINE__, what);
    return false;
  };
  if (!sizeOk(data.positions, n * 3, "data.positions.size() == N * 3") ||
      !sizeOk(data.scales, n * 3, "data.scales.size() == N * 3") ||
      !sizeOk(data.rotations, n * 4, "data.rotations.size() == N * 4") ||
      !sizeOk(data.alphas, n, "data.alphas.size() == N") ||
      !sizeOk(data.colors, n * 3, "data.colors.size() == N * 3")) {
    return false;
  }
  const int shDim = n ? static_cast<int>(data.sh.size() / n / 3) : 0;
  const size_t D = 17 + static_cast<size_t>(shDim) * 3;
  if (shDim > 15) {
    plyLog("[SPZ ERROR] spz_amd: saveSplatToPly: %d sh coefficients per channel (max 15)", shDim);
    return false;
  }
  // every field of every row, the three normals' zeros included, is written by the device copy: no zero fill of what is
  // 2.5 GB for 10 M SH3 points, its pages mapped by several threads
  std::vector<float> rows;
  detail::resizeUninitialized(&rows, n * D);
  {
    detail::Prefault pf;
    pf.add(rows.data(), rows.size() * sizeof(float));
    pf.start();
    pf.join();
  }
  if (n) {
    spz_amd_cloud_in in = {data.positions.data(), data.scales.data(), data.rotations.data(),
                           data.alphas.data(),    data.colors.data(), data.sh.empty() ? nullptr : data.sh.data()};
    // row assembly with the from -> RDF flips (load-spz.cc:856-893)
    const int rc = spz_amd_cloud_to_ply_rows_host(&in, n, shDim, static_cast<int>(o.from), rows.data(), plyDevice());
    if (rc != SPZ_AMD_OK) {
      plyLog("[SPZ ERROR] spz_amd: saveSplatToPly: %s", spz_amd_status_string(rc));
      setLastDeviceStatus(rc);
      return false;
    }
  }

  std::ofstream out(filename, std::ios::binary);
  if (!out.good()) {
    plyLog("[SPZ ERROR] Unable to open for writing: %s", filename.c_str());
    return false;
  }
  out << "ply\nformat binary_little_endian 1.0\nelement vertex " << N << "\n";
  for (const char *name : {"x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"}) {
    out << "property float " << name << "\n";
  }
  for (int i = 0; i < shDim * 3; ++i) out << "property float f_rest_" << i << "\n";
  for (const char *name : {"opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"}) {
    out << "property float " << name << "\n";
  }
  out << "end_header\n";
  out.write(reinterpret_cast<const char *>(rows.data()), static_cast<std::streamsize>(rows.size() * sizeof(float)));
  out.close();
  if (!out.good()) {
    plyLog("[SPZ ERROR] Failed to write to: %s", filename.c_str());
    return false;
  }
  return true;
}

}  // namespace spz
