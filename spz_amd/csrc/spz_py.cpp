// spz_py.cpp — Python module `spz` (imported as spz_amd.spz) over the C++ drop-in layer.
//
// Same surface as the reference's nanobind shim (/root/reference/src/python/spz/spz.cc:110-362):
// CoordinateSystem enum with exported values, PackOptions.from_coord, UnpackOptions.to_coord,
// GaussianCloud with copying float32 array properties and the shim's validation messages,
// load_spz / save_spz / load_splat_from_ply / save_splat_to_ply.  nanobind is not available in
// this image; pybind11 is, so the dtype/ndim gate that nanobind's ndarray caster applies
// (numeric dtypes convert to float32, anything else -> TypeError "incompatible function
// arguments") is written out by hand in `toFloatVector`.
#include <chrono>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <sstream>
#include <memory>
#include <string>
#include <vector>

#include "spz_amd.h"
#include "spz_amd_host.hpp"
#include "spz_deflate.hpp"
#include "spz_inflate.hpp"

namespace py = pybind11;

namespace {
// The bytes of a Python bytes object, in place (the std::string conversion would copy them: 409 MB for a 10 M-point file).
struct BytesView {
  const uint8_t *p;
  size_t n;
};
BytesView viewOf(const py::bytes &b) {
  char *buf = nullptr;
  Py_ssize_t len = 0;
  if (PyBytes_AsStringAndSize(b.ptr(), &buf, &len) != 0) throw py::error_already_set();
  return {reinterpret_cast<const uint8_t *>(buf), static_cast<size_t>(len)};
}
}  // namespace

namespace {

// Accepts what nb::ndarray<numpy, float, ndim<1>, c_contig, device::cpu> accepts: a 1-D array (or
// array-like) of a bool/int/uint/float dtype, converted to float32.  Everything else is the
// overload-resolution failure nanobind reports.
std::vector<float> toFloatVector(const py::object &obj, const char *prop) {
  auto reject = [&]() {
    throw py::type_error(std::string("__set__(): incompatible function arguments. ") + prop +
                         " expects a 1-D numeric numpy array convertible to float32");
  };
  py::array arr;
  try {
    arr = py::array::ensure(obj);
  } catch (const py::error_already_set &) {
    PyErr_Clear();
  }
  if (!arr) reject();
  const char kind = arr.dtype().kind();
  if (!(kind == 'b' || kind == 'i' || kind == 'u' || kind == 'f')) reject();
  if (arr.ndim() != 1) reject();
  py::array_t<float, py::array::c_style | py::array::forcecast> f(arr);
  std::vector<float> out(static_cast<size_t>(f.size()));
  if (!out.empty()) std::memcpy(out.data(), f.data(), out.size() * sizeof(float));
  return out;
}

// New owning float32 1-D copy (spz.cc:47-79).
py::array_t<float> toArray(const std::vector<float> &v) {
  py::array_t<float> a(static_cast<py::ssize_t>(v.size()));
  if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(float));
  return a;
}

void ensureMultiple(const char *name, size_t size, size_t k) {  // spz.cc:27-37
  if (k == 0) throw py::value_error("internal error: divisor cannot be zero");
  if (size % k != 0) {
    throw py::value_error(std::string(name) + " length must be a multiple of " + std::to_string(k) + ", got " +
                          std::to_string(size));
  }
}

// The product has no CPU fallback: an unusable device is raised, not returned as an empty cloud.
void raiseIfDeviceUnusable() {
  const int st = spz::lastDeviceStatus();
  if (st == SPZ_AMD_ERR_NO_DEVICE || st == SPZ_AMD_ERR_HIP) {
    throw std::runtime_error(std::string("spz_amd: ") + spz_amd_status_string(st) +
                             " (the SPZ hot path runs on the GPU only)");
  }
}

}  // namespace

PYBIND11_MODULE(spz, m) {
  m.doc() = "MI355X-native drop-in for the `spz` Python bindings (Gaussian splat .spz codec).";

  py::enum_<spz::CoordinateSystem>(m, "CoordinateSystem",
                                   "Axis conventions: Right/Left, Up/Down, Front/Back (RDF = PLY, RUB = three.js, "
                                   "LUF = glTF, RUF = Unity).")
      .value("UNSPECIFIED", spz::CoordinateSystem::UNSPECIFIED)
      .value("LDB", spz::CoordinateSystem::LDB)
      .value("RDB", spz::CoordinateSystem::RDB)
      .value("LUB", spz::CoordinateSystem::LUB)
      .value("RUB", spz::CoordinateSystem::RUB)
      .value("LDF", spz::CoordinateSystem::LDF)
      .value("RDF", spz::CoordinateSystem::RDF)
      .value("LUF", spz::CoordinateSystem::LUF)
      .value("RUF", spz::CoordinateSystem::RUF)
      .export_values();

  py::class_<spz::PackOptions>(m, "PackOptions")
      .def(py::init<>())
      .def_readwrite("from_coord", &spz::PackOptions::from, "Coordinate system of the input splat");
  py::class_<spz::UnpackOptions>(m, "UnpackOptions")
      .def(py::init<>())
      .def_readwrite("to_coord", &spz::UnpackOptions::to, "Desired coordinate system of the output splat");

  using Cloud = spz::GaussianCloud;
  py::class_<Cloud>(m, "GaussianCloud")
      .def(py::init<>(), "Construct an empty GaussianCloud.")
      .def_property_readonly("num_points",
                             [](const Cloud &c) { return static_cast<int32_t>(c.positions.size() / 3); })
      .def("__len__", [](const Cloud &c) { return static_cast<int32_t>(c.positions.size() / 3); })
      .def("__repr__",
           [](const Cloud &c) {
             return py::str("GaussianCloud(num_points={}, sh_degree={}, antialiased={})")
                 .format(static_cast<int32_t>(c.positions.size() / 3), c.shDegree, c.antialiased);
           })
      .def_property(
          "sh_degree", [](const Cloud &c) { return c.shDegree; },
          [](Cloud &c, int32_t deg) {
            if (deg < 0 || deg > 3) throw py::value_error("sh_degree must be in [0, 3]");
            c.shDegree = deg;
          })
      .def_readwrite("antialiased", &Cloud::antialiased)
      .def_property(
          "positions", [](const Cloud &c) { return toArray(c.positions); },
          [](Cloud &c, const py::object &o) {
            std::vector<float> v = toFloatVector(o, "positions");
            ensureMultiple("positions", v.size(), 3);
            c.positions = std::move(v);
            c.numPoints = static_cast<int32_t>(c.positions.size() / 3);  // positions define num_points
          })
      .def_property(
          "scales", [](const Cloud &c) { return toArray(c.scales); },
          [](Cloud &c, const py::object &o) {
            std::vector<float> v = toFloatVector(o, "scales");
            ensureMultiple("scales", v.size(), 3);
            c.scales = std::move(v);
            if (c.numPoints > 0 && c.scales.size() != static_cast<size_t>(c.numPoints) * 3) {
              throw py::value_error("scales length must equal num_points * 3");
            }
          })
      .def_property(
          "rotations", [](const Cloud &c) { return toArray(c.rotations); },
          [](Cloud &c, const py::object &o) {
            std::vector<float> v = toFloatVector(o, "rotations");
            ensureMultiple("rotations", v.size(), 4);
            c.rotations = std::move(v);
            if (c.numPoints > 0 && c.rotations.size() != static_cast<size_t>(c.numPoints) * 4) {
              throw py::value_error("rotations length must equal num_points * 4");
            }
          })
      .def_property(
          "alphas", [](const Cloud &c) { return toArray(c.alphas); },
          [](Cloud &c, const py::object &o) {
            c.alphas = toFloatVector(o, "alphas");
            if (c.numPoints > 0 && c.alphas.size() != static_cast<size_t>(c.numPoints)) {
              throw py::value_error("alphas length must equal num_points");
            }
          })
      .def_property(
          "colors", [](const Cloud &c) { return toArray(c.colors); },
          [](Cloud &c, const py::object &o) {
            std::vector<float> v = toFloatVector(o, "colors");
            ensureMultiple("colors", v.size(), 3);
            c.colors = std::move(v);
            if (c.numPoints > 0 && c.colors.size() != static_cast<size_t>(c.numPoints) * 3) {
              throw py::value_error("colors length must equal num_points * 3");
            }
          })
      .def_property(
          "sh", [](const Cloud &c) { return toArray(c.sh); },
          [](Cloud &c, const py::object &o) {
            std::vector<float> v = toFloatVector(o, "sh");
            ensureMultiple("sh", v.size(), 3);
            const int deg = c.shDegree;
            const size_t perChannel = (deg == 0) ? 0 : static_cast<size_t>((deg + 1) * (deg + 1) - 1);
            if (perChannel == 0) {
              if (!v.empty()) throw py::value_error("sh must be empty when sh_degree == 0");
            } else {
              ensureMultiple("sh", v.size(), perChannel * 3);
            }
            c.sh = std::move(v);
            if (c.numPoints > 0 && c.sh.size() != static_cast<size_t>(c.numPoints) * perChannel * 3) {
              throw py::value_error("sh length must equal num_points * ((sh_degree+1)^2 - 1) * 3");
            }
          })
      .def("convert_coordinates",
           [](Cloud &c, spz::CoordinateSystem from, spz::CoordinateSystem to) {
             c.convertCoordinates(from, to);
             if (c.numPoints) raiseIfDeviceUnusable();
           },
           py::arg("from_coord"), py::arg("to_coord"), "Convert between two coordinate systems in-place.")
      .def("rotate_180_deg_about_x",
           [](Cloud &c) {
             c.rotate180DegAboutX();
             if (c.numPoints) raiseIfDeviceUnusable();
           },
           "RUB <-> RDF conversion (180 degrees about X).")
      .def("median_volume", [](const Cloud &c) {
             spz::setLastDeviceStatus(SPZ_AMD_OK);
             const float v = c.medianVolume();   // selection runs on the device
             raiseIfDeviceUnusable();
             return v;
           }, "Return the median Gaussian volume.");

  // Device-resident packed load (an extra of this implementation): the stream stays in HBM.
  py::class_<spz::DevicePackedGaussians>(m, "DevicePackedGaussians",
                                         "A .spz file's packed sections left in device memory (spz::loadSpzPackedDevice). "
                                         "Pointers are device addresses (ints); the object owns the memory until release().")
      .def_readonly("num_points", &spz::DevicePackedGaussians::numPoints)
      .def_readonly("sh_degree", &spz::DevicePackedGaussians::shDegree)
      .def_readonly("fractional_bits", &spz::DevicePackedGaussians::fractionalBits)
      .def_readonly("antialiased", &spz::DevicePackedGaussians::antialiased)
      .def_readonly("version", &spz::DevicePackedGaussians::version)
      .def_readonly("uses_quaternion_smallest_three", &spz::DevicePackedGaussians::usesQuaternionSmallestThree)
      .def_readonly("device", &spz::DevicePackedGaussians::device)
      .def_readonly("inflated_on_device", &spz::DevicePackedGaussians::inflatedOnDevice)
      .def_property_readonly("uses_float16", [](const spz::DevicePackedGaussians &d) { return d.usesFloat16(); })
      .def_property_readonly("valid", [](const spz::DevicePackedGaussians &d) { return d.valid(); })
      .def_property_readonly("stream_ptr", [](const spz::DevicePackedGaussians &d) { return reinterpret_cast<uintptr_t>(d.stream); })
      .def_property_readonly("stream_bytes", [](const spz::DevicePackedGaussians &d) { return d.streamBytes; })
      .def_property_readonly("sections", [](const spz::DevicePackedGaussians &d) {
             py::dict r;
             auto put = [&](const char *name, const uint8_t *p, size_t n) { r[name] = py::make_tuple(reinterpret_cast<uintptr_t>(p), n); };
             put("positions", d.positions, d.positionsBytes);
             put("alphas", d.alphas, d.alphasBytes);
             put("colors", d.colors, d.colorsBytes);
             put("scales", d.scales, d.scalesBytes);
             put("rotations", d.rotations, d.rotationsBytes);
             put("sh", d.sh, d.shBytes);
             return r;
           }, "name -> (device address, bytes) of the six sections, in PackedGaussians' naming.")
      .def("release", &spz::DevicePackedGaussians::release, "Return the device memory; the object becomes empty.")
      .def("unpack", [](const spz::DevicePackedGaussians &d, const spz::UnpackOptions &o) {
             spz::GaussianCloud g;
             {
               py::gil_scoped_release release;
               g = d.unpack(o);
             }
             if (g.numPoints == 0) raiseIfDeviceUnusable();
             return g;
           }, py::arg("options") = spz::UnpackOptions(), "unpackGaussians of all points, from where the stream lies.")
      .def("unpack_indices", [](const spz::DevicePackedGaussians &d, const std::vector<uint32_t> &indices, const spz::UnpackOptions &o) {
             spz::GaussianCloud g = d.unpackIndices(indices, o);
             if (g.numPoints == 0 && !indices.empty()) raiseIfDeviceUnusable();
             return g;
           }, py::arg("indices"), py::arg("options") = spz::UnpackOptions(), "One gather launch over the resident stream.");
  m.def("load_spz_packed_device", [](const std::string &filename) {
          spz::DevicePackedGaussians d;
          {
            py::gil_scoped_release release;
            d = spz::loadSpzPackedDevice(filename);
          }
          if (!d.valid()) raiseIfDeviceUnusable();
          return d;
        }, py::arg("filename"), "loadSpzPacked with the packed sections left in device memory.");
  m.def("_load_spz_packed_device_bytes", [](const py::bytes &data) {
          const BytesView in = viewOf(data);
          spz::DevicePackedGaussians d = spz::loadSpzPackedDevice(in.p, static_cast<int32_t>(in.n));
          if (!d.valid()) raiseIfDeviceUnusable();
          return d;
        }, py::arg("data"), "The same from .spz bytes in memory.");

  m.def("load_spz",
        [](const std::string &filename, const spz::UnpackOptions &o) {
          spz::GaussianCloud g;
          {
            py::gil_scoped_release release;  // other Python threads run meanwhile (the library is re-entrant)
            g = spz::loadSpz(filename, o);
          }
          if (g.numPoints == 0) raiseIfDeviceUnusable();
          return g;
        },
        py::arg("filename"), py::arg("options") = spz::UnpackOptions(), "Load a *.spz* file and return a GaussianCloud.");
  m.def("save_spz",
        [](const spz::GaussianCloud &g, const spz::PackOptions &o, const std::string &filename) {
          bool ok;
          {
            py::gil_scoped_release release;
            ok = spz::saveSpz(g, o, filename);
          }
          if (!ok) raiseIfDeviceUnusable();
          return ok;
        },
        py::arg("gaussians"), py::arg("options"), py::arg("filename"), "Save a GaussianCloud to a *.spz* file.");
  // Extras of this implementation (underscore-prefixed: not part of the reference surface).
  m.def("_compress_gzipped", [](const py::bytes &data) {
    const BytesView in = viewOf(data);
    std::vector<uint8_t> out;
    bool ok;
    {
      py::gil_scoped_release release;
      ok = spz::compressGzipped(in.p, in.n, &out);
    }
    if (!ok) throw std::runtime_error("compressGzipped failed");
    return py::bytes(reinterpret_cast<const char *>(out.data()), out.size());
  }, "gzip wrapper of saveSpz (host zlib, parameters of load-spz.cc:190).");
  m.def("_compress_gzipped_parallel", [](const py::bytes &data, int threads) {
    const BytesView in = viewOf(data);
    std::vector<uint8_t> out;
    bool ok;
    {
      py::gil_scoped_release release;
      ok = spz::compressGzippedParallel(in.p, in.n, &out, threads);
    }
    if (!ok) throw std::runtime_error("compressGzippedParallel failed");
    return py::bytes(reinterpret_cast<const char *>(out.data()), out.size());
  }, py::arg("data"), py::arg("threads"), "Opt-in multi-threaded gzip (one member, independent deflate blocks).");
  m.def("_compress_gzipped_exact", [](const py::bytes &data, int threads, int windows_per_chunk, size_t verify_prefix) -> py::object {
    const BytesView in = viewOf(data);
    std::vector<uint8_t> out;
    bool ok;
    {
      py::gil_scoped_release release;
      ok = spz::exactgz::compress(in.p, in.n, threads, windows_per_chunk, &out,
                                  verify_prefix);
    }
    if (!ok) return py::none();
    return py::bytes(reinterpret_cast<const char *>(out.data()), out.size());
  }, py::arg("data"), py::arg("threads") = 8, py::arg("windows_per_chunk") = 32, py::arg("verify_prefix") = 0,
     "The multi-threaded writer with zlib's exact bytes (None when it declines the input).");
  m.def("_compress_gzipped_exact_model", [](const py::bytes &data, int threads, size_t verify_prefix) -> py::object {
    const BytesView in = viewOf(data);
    std::vector<uint8_t> out;
    bool ok;
    {
      py::gil_scoped_release release;
      std::unique_ptr<spz::exactgz::HeadParser> parser(spz::exactgz::newModelHeadParser());
      ok = spz::exactgz::compressWithHeadParser(in.p, in.n, threads, *parser, &out, verify_prefix);
    }
    if (!ok) return py::none();
    return py::bytes(reinterpret_cast<const char *>(out.data()), out.size());
  }, py::arg("data"), py::arg("threads") = 4, py::arg("verify_prefix") = 0,
     "Test hook: the exact writer with its parse done by the serial host model of the device stages "
     "(links, match tables, lazy state machine, record-window splice); None when declined.");
  m.def("_device_inflate_count", []() { return spz::deviceInflateCount(); },
        "gzip members inflated on the device so far.");
  m.def("_device_inflate_last_decline", []() { return std::string(spz::deviceInflateLastDecline()); },
        "Why the device reader last stood down on this thread ('' = it did not).");
  m.def("_device_gzip_parse_count", []() { return spz::deviceGzipParseCount(); },
        "gzip members written so far with their LZ77 parse done on the device.");
  m.def("_device_gzip_reject_count", []() { return spz::deviceGzipRejectCount(); },
        "members of the device gzip writer discarded because one of its self-checks failed.");
  m.def("_effective_cpu_count", []() { return spz::effectiveCpuCount(); },
        "CPUs the worker pools of the container stage size themselves by (online, affinity mask, cgroup quota).");
  m.def("_parallel_inflate_count", []() { return spz::pinflate::successCount(); },
        "Members inflated by the parallel single-stream reader so far in this process.");
  m.def("_decompress_gzipped", [](const py::bytes &data) -> py::object {
    const BytesView in = viewOf(data);
    std::vector<uint8_t> out;
    bool ok;
    {
      py::gil_scoped_release release;
      ok = spz::decompressGzipped(in.p, in.n, &out);
    }
    if (!ok) return py::none();
    return py::bytes(reinterpret_cast<const char *>(out.data()), out.size());
  }, "Inverse of _compress_gzipped; None on failure.");
  m.def("_save_spz_bytes", [](const spz::GaussianCloud &g, const spz::PackOptions &o) -> py::object {
    std::vector<uint8_t> out;
    bool ok;
    {
      py::gil_scoped_release release;
      ok = spz::saveSpz(g, o, &out);
    }
    if (!ok) {
      raiseIfDeviceUnusable();
      return py::none();
    }
    return py::bytes(reinterpret_cast<const char *>(out.data()), out.size());
  }, py::arg("gaussians"), py::arg("options"), "saveSpz(cloud, options, &vector) -> .spz bytes in memory.");
  m.def("_load_spz_bytes", [](const py::bytes &data, const spz::UnpackOptions &o) {
    const BytesView in = viewOf(data);
    spz::GaussianCloud g;
    {
      py::gil_scoped_release release;
      g = spz::loadSpz(in.p, static_cast<int32_t>(in.n), o);
    }
    if (g.numPoints == 0) raiseIfDeviceUnusable();
    return g;
  }, py::arg("data"), py::arg("options") = spz::UnpackOptions(), "loadSpz(ptr, size, options) from .spz bytes.");
  m.def("_packed_unpack", [](const py::bytes &data, int32_t index, spz::CoordinateSystem from, spz::CoordinateSystem to,
                              bool gzipped) -> py::object {
    // loadSpzPacked (or deserializePackedGaussians for a raw stream) + PackedGaussians::at / unpack:
    // returns (65 packed bytes, 59 floats) in the field order of PackedGaussian / UnpackedGaussian.
    const BytesView in = viewOf(data);
    spz::PackedGaussians packed;
    if (gzipped) {
      packed = spz::loadSpzPacked(in.p, static_cast<int32_t>(in.n));
    } else {
      std::istringstream ss(std::string(reinterpret_cast<const char *>(in.p), in.n));
      packed = spz::deserializePackedGaussians(ss);
    }
    if (index < 0 || index >= packed.numPoints) return py::none();
    spz::setLastDeviceStatus(SPZ_AMD_OK);
    const spz::PackedGaussian one = packed.at(index);
    const spz::UnpackedGaussian u = packed.unpack(index, spz::coordinateConverter(from, to));
    raiseIfDeviceUnusable();
    std::string b;
    auto putb = [&](const uint8_t *p, size_t n) { b.append(reinterpret_cast<const char *>(p), n); };
    putb(one.position.data(), 9); putb(one.rotation.data(), 4); putb(one.scale.data(), 3); putb(one.color.data(), 3);
    putb(&one.alpha, 1); putb(one.shR.data(), 15); putb(one.shG.data(), 15); putb(one.shB.data(), 15);
    std::vector<float> f;
    auto putf = [&](const float *p, size_t n) { f.insert(f.end(), p, p + n); };
    putf(u.position.data(), 3); putf(u.rotation.data(), 4); putf(u.scale.data(), 3); putf(u.color.data(), 3);
    putf(&u.alpha, 1); putf(u.shR.data(), 15); putf(u.shG.data(), 15); putf(u.shB.data(), 15);
    return py::make_tuple(py::bytes(b), toArray(f));
  }, py::arg("data"), py::arg("index"), py::arg("from_coord"), py::arg("to_coord"), py::arg("gzipped") = true,
     "PackedGaussians::at(i) bytes and PackedGaussians::unpack(i, coordinateConverter(from, to)) floats.");
  m.def("_unpack_indices", [](const py::bytes &data, const std::vector<uint32_t> &indices, const spz::UnpackOptions &o,
                               bool gzipped) {
    const BytesView in = viewOf(data);
    spz::PackedGaussians packed;
    if (gzipped) {
      packed = spz::loadSpzPacked(in.p, static_cast<int32_t>(in.n));
    } else {
      std::istringstream ss(std::string(reinterpret_cast<const char *>(in.p), in.n));
      packed = spz::deserializePackedGaussians(ss);
    }
    spz::setLastDeviceStatus(SPZ_AMD_OK);
    spz::GaussianCloud g = spz::unpackIndices(packed, indices, o);
    if (g.numPoints == 0) raiseIfDeviceUnusable();
    return g;
  }, py::arg("data"), py::arg("indices"), py::arg("options") = spz::UnpackOptions(), py::arg("gzipped") = true,
     "unpackIndices(loadSpzPacked(data), indices, options): one gather launch.");
  m.def("_unpack_from_stream", [](const py::bytes &data, const spz::UnpackOptions &o) {
    const BytesView in = viewOf(data);
    spz::GaussianCloud g = spz::unpackFromStream(in.p, in.n, o);
    if (g.numPoints == 0) raiseIfDeviceUnusable();
    return g;
  }, py::arg("data"), py::arg("options") = spz::UnpackOptions(), "Cloud of a raw (pre-gzip) stream: loadSpz without the gunzip step.");
  m.def("_pack_unpack_seconds", [](const spz::GaussianCloud &g, const spz::PackOptions &o, const spz::UnpackOptions &u) {
    // spz::packToStream / spz::unpackFromStream timed around the C++ calls themselves (fresh vectors), without the copy
    // into a Python bytes object that _pack_to_stream pays on top
    std::vector<uint8_t> stream;
    auto t0 = std::chrono::steady_clock::now();
    const bool ok = spz::packToStream(g, o, &stream);
    const double pack_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!ok) raiseIfDeviceUnusable();
    t0 = std::chrono::steady_clock::now();
    spz::GaussianCloud back = spz::unpackFromStream(stream.data(), stream.size(), u);
    const double unpack_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return py::make_tuple(pack_s, unpack_s, static_cast<int64_t>(back.numPoints));
  }, py::arg("gaussians"), py::arg("pack_options"), py::arg("unpack_options"), "Seconds of packToStream and unpackFromStream at the C++ boundary.");
  m.def("_save_load_seconds", [](const spz::GaussianCloud &g, const spz::PackOptions &o, const spz::UnpackOptions &u) {
    // spz::saveSpz / spz::loadSpz (the reference's vector overloads, load-spz.h) timed around the C++ calls themselves
    std::vector<uint8_t> file;
    const uint64_t parses_before = spz::deviceGzipParseCount();
    auto t0 = std::chrono::steady_clock::now();
    const bool ok = spz::saveSpz(g, o, &file);
    const double save_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!ok) raiseIfDeviceUnusable();
    const bool gzip_on_device = spz::deviceGzipParseCount() != parses_before;
    t0 = std::chrono::steady_clock::now();
    spz::GaussianCloud back = spz::loadSpz(file, u);
    const double load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return py::make_tuple(save_s, load_s, static_cast<int64_t>(file.size()), static_cast<int64_t>(back.numPoints), gzip_on_device);
  }, py::arg("gaussians"), py::arg("pack_options"), py::arg("unpack_options"),
     "Seconds of saveSpz and loadSpz at the C++ boundary, file bytes, points read back, whether the gzip stage ran on the device.");
  m.def("_pack_to_stream", [](const spz::GaussianCloud &g, const spz::PackOptions &o) -> py::object {
    std::vector<uint8_t> out;
    if (!spz::packToStream(g, o, &out)) {
      raiseIfDeviceUnusable();
      return py::none();
    }
    return py::bytes(reinterpret_cast<const char *>(out.data()), out.size());
  }, py::arg("gaussians"), py::arg("options"), "Raw (pre-gzip) stream of a cloud.");

  m.def("load_splat_from_ply",
        [](const std::string &filename, const spz::UnpackOptions &o) {
          spz::setLastDeviceStatus(SPZ_AMD_OK);
          spz::GaussianCloud g = spz::loadSplatFromPly(filename, o);
          if (g.numPoints == 0) raiseIfDeviceUnusable();
          return g;
        },
        py::arg("filename"), py::arg("options") = spz::UnpackOptions(), "Read GaussianCloud data from a *.ply* file.");
  m.def("save_splat_to_ply",
        [](const spz::GaussianCloud &g, const spz::PackOptions &o, const std::string &filename) {
          spz::setLastDeviceStatus(SPZ_AMD_OK);
          const bool ok = spz::saveSplatToPly(g, o, filename);
          if (!ok) raiseIfDeviceUnusable();
          return ok;
        },
        py::arg("gaussians"), py::arg("options"), py::arg("filename"), "Write GaussianCloud data to a *.ply* file.");
}
