// host_bench — the C++ boundary users actually call (spz::packToStream / unpackFromStream / saveSpz /
// loadSpz with host vectors in and out), timed in C++ so that no language binding's copies are counted.
//   g++ -O2 -std=c++17 -Iinclude -o spz_amd/bin/host_bench tools/host_bench.cpp -Lspz_amd/lib -lspz_host -lspz_amd
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <string>
#include <vector>
#include <unistd.h>

#include "spz_amd.h"
#include "spz_amd_host.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static uint64_t checksum(const void *p, size_t bytes) {  // order-dependent, cheap: detects any differing byte
  const uint64_t *q = static_cast<const uint64_t *>(p);
  uint64_t h = 0x9e3779b97f4a7c15ull;
  for (size_t i = 0; i < bytes / 8; ++i) h = (h ^ q[i]) * 0x100000001b3ull + (h >> 29);
  for (size_t i = bytes & ~size_t(7); i < bytes; ++i) h = (h ^ static_cast<const uint8_t *>(p)[i]) * 0x100000001b3ull;
  return h;
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? static_cast<size_t>(atoll(argv[1])) : 10000000;
  const int deg = argc > 2 ? atoi(argv[2]) : 3;
  const int reps = argc > 3 ? atoi(argv[3]) : 4;
  const bool gzip = argc > 4 && atoi(argv[4]) != 0;
  const size_t shDim = deg == 0 ? 0 : deg == 1 ? 3 : deg == 2 ? 8 : 15;
  spz::GaussianCloud g;
  g.numPoints = static_cast<int32_t>(n);
  g.shDegree = deg;
  g.positions.resize(n * 3);
  g.scales.resize(n * 3);
  g.rotations.resize(n * 4);
  g.alphas.resize(n);
  g.colors.resize(n * 3);
  g.sh.resize(n * shDim * 3);
  {  // SURVEY §8(d) distributions, filled by a few threads (std::mt19937 per slice)
    auto fill = [](std::vector<float> &v, float lo, float hi, bool normal, float sigma, unsigned seed) {
      const int T = 8;
      std::vector<std::thread> pool;
      for (int t = 0; t < T; ++t) {
        pool.emplace_back([&, t]() {
          std::mt19937 rng(seed * 131u + t);
          std::uniform_real_distribution<float> u(lo, hi);
          std::normal_distribution<float> nd(0.0f, sigma);
          const size_t a = v.size() * t / T, b = v.size() * (t + 1) / T;
          for (size_t i = a; i < b; ++i) v[i] = normal ? nd(rng) : u(rng);
        });
      }
      for (auto &th : pool) th.join();
    };
    fill(g.positions, -10, 10, false, 0, 1);
    fill(g.scales, -8, 0, false, 0, 2);
    fill(g.rotations, 0, 0, true, 1.0f, 3);
    fill(g.alphas, 0, 0, true, 3.0f, 4);
    fill(g.colors, 0, 0, true, 1.0f, 5);
    fill(g.sh, 0, 0, true, 0.25f, 6);
  }
  spz::PackOptions po;
  po.from = spz::CoordinateSystem::RDF;
  spz::UnpackOptions uo;
  uo.to = spz::CoordinateSystem::RDF;
  double pack_fresh = 1e30, pack_reuse = 1e30, unpack = 1e30, first_pack = 0, first_unpack = 0;
  uint64_t sum_stream = 0, sum_floats = 0;
  std::vector<uint8_t> kept;
  for (int r = 0; r < reps; ++r) {
    std::vector<uint8_t> stream;  // a fresh vector every time: allocation and first touch are in the figure
    double t0 = now();
    if (!spz::packToStream(g, po, &stream)) { printf("{\"error\": \"packToStream failed, status %d\"}\n", spz::lastDeviceStatus()); return 1; }
    double t = now() - t0;
    if (r == 0) first_pack = t; else pack_fresh = std::min(pack_fresh, t);
    const uint64_t cs = checksum(stream.data(), stream.size());
    if (r && cs != sum_stream) { printf("{\"error\": \"stream differs between repetitions\"}\n"); return 1; }
    sum_stream = cs;
    if (r == 0) kept = stream;
    t0 = now();
    if (!spz::packToStream(g, po, &kept)) return 1;  // a vector that already has the capacity
    pack_reuse = std::min(pack_reuse, now() - t0);
    if (checksum(kept.data(), kept.size()) != sum_stream) { printf("{\"error\": \"reused vector differs\"}\n"); return 1; }
    t0 = now();
    spz::GaussianCloud back = spz::unpackFromStream(stream.data(), stream.size(), uo);
    t = now() - t0;
    if (back.numPoints != static_cast<int32_t>(n)) { printf("{\"error\": \"unpackFromStream failed, status %d\"}\n", spz::lastDeviceStatus()); return 1; }
    if (r == 0) first_unpack = t; else unpack = std::min(unpack, t);
    const uint64_t cf = checksum(back.sh.empty() ? back.positions.data() : back.sh.data(), (back.sh.empty() ? back.positions.size() : back.sh.size()) * 4) ^
                        checksum(back.rotations.data(), back.rotations.size() * 4) ^ checksum(back.alphas.data(), back.alphas.size() * 4);
    if (r && cf != sum_floats) { printf("{\"error\": \"decoded floats differ between repetitions\"}\n"); return 1; }
    sum_floats = cf;
  }
  // the raw C ABI under it, into buffers that are already mapped (what the C++ layer would cost with no allocation at all)
  double abi_enc = 1e30, abi_dec = 1e30;
  {
    spz_amd_layout lay;
    spz_amd_stream_layout(n, deg, 3, &lay);
    std::vector<uint8_t> s(lay.total_bytes, 1);
    spz::GaussianCloud b = g;
    const spz_amd_cloud_in in = {g.positions.data(), g.scales.data(), g.rotations.data(), g.alphas.data(), g.colors.data(), g.sh.empty() ? nullptr : g.sh.data()};
    const spz_amd_cloud_out out = {b.positions.data(), b.scales.data(), b.rotations.data(), b.alphas.data(), b.colors.data(), b.sh.empty() ? nullptr : b.sh.data()};
    for (int r = 0; r < reps; ++r) {
      double t0 = now();
      if (spz_amd_encode_host(&in, n, deg, 0, SPZ_AMD_RDF, 3, s.data(), s.size(), 0) != 0) return 1;
      abi_enc = std::min(abi_enc, now() - t0);
      t0 = now();
      if (spz_amd_decode_host(s.data(), s.size(), SPZ_AMD_RDF, &out, 0) != 0) return 1;
      abi_dec = std::min(abi_dec, now() - t0);
    }
    if (checksum(s.data(), s.size()) != sum_stream) { printf("{\"error\": \"ABI stream differs from packToStream's\"}\n"); return 1; }
  }
  const double fb = (14 + shDim * 3) * 4.0 * n, sb = (20 + shDim * 3) * 1.0 * n + 16;
  printf("{\"points\": %zu, \"sh_degree\": %d, \"float_GB\": %.3f, \"stream_GB\": %.3f, \"host_cores\": %u, "
         "\"pack_to_stream_first_s\": %.4f, \"pack_to_stream_fresh_vector_s\": %.4f, \"pack_to_stream_reused_vector_s\": %.4f, "
         "\"unpack_from_stream_first_s\": %.4f, \"unpack_from_stream_s\": %.4f, \"abi_encode_host_s\": %.4f, \"abi_decode_host_s\": %.4f, "
         "\"pack_GBps_over_pcie\": %.1f, \"unpack_GBps_over_pcie\": %.1f, \"stream_checksum\": \"%016llx\", \"floats_checksum\": \"%016llx\"",
         n, deg, fb / 1e9, sb / 1e9, std::thread::hardware_concurrency(), first_pack, pack_fresh, pack_reuse, first_unpack, unpack,
         abi_enc, abi_dec, (fb + sb) / 1e9 / pack_fresh, (fb + sb) / 1e9 / unpack, (unsigned long long)sum_stream, (unsigned long long)sum_floats);
  {  // GaussianCloud::convertCoordinates on the host arrays (positions, rotations, sh up, flipped, down): there and back
    spz::GaussianCloud c = g;
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
      const double t0 = now();
      c.convertCoordinates(spz::CoordinateSystem::RDF, spz::CoordinateSystem::LUF);
      c.convertCoordinates(spz::CoordinateSystem::LUF, spz::CoordinateSystem::RDF);
      best = std::min(best, (now() - t0) / 2);
    }
    const bool same = std::memcmp(c.sh.data(), g.sh.data(), g.sh.size() * 4) == 0 && std::memcmp(c.positions.data(), g.positions.data(), g.positions.size() * 4) == 0;
    printf(", \"convert_coordinates_s\": %.4f, \"convert_coordinates_round_trip_identical\": %s", best, same ? "true" : "false");
  }
  if (gzip) {
    // the first save and load allocate what later ones reuse (device blocks, the stream buffer): both are reported
    std::vector<uint8_t> file;
    double t_save_first = 0, t_save = 1e30, t_load_first = 0, t_load = 1e30;
    bool load_ok = true;
    for (int r = 0; r < std::max(reps, 2); ++r) {
      std::vector<uint8_t> fresh;
      double t0 = now();
      if (!spz::saveSpz(g, po, &fresh)) return 1;
      const double dt = now() - t0;
      if (r == 0) t_save_first = dt;
      t_save = std::min(t_save, dt);
      file.swap(fresh);
    }
    for (int r = 0; r < std::max(reps, 2); ++r) {
      const double t0 = now();
      spz::GaussianCloud back = spz::loadSpz(file, uo);
      const double dt = now() - t0;
      if (r == 0) t_load_first = dt;
      t_load = std::min(t_load, dt);
      load_ok = load_ok && back.numPoints == static_cast<int32_t>(n);
    }
    printf(", \"save_spz_first_s\": %.3f, \"save_spz_s\": %.3f, \"load_spz_first_s\": %.3f, \"load_spz_s\": %.3f, \"spz_bytes\": %zu, \"load_ok\": %s",
           t_save_first, t_save, t_load_first, t_load, file.size(), load_ok ? "true" : "false");
    {  // the overloads that take a file name (what most callers use): a file in memory (/dev/shm) unless SPZ_BENCH_FILE says where
      const char *e = std::getenv("SPZ_BENCH_FILE");
      const std::string path = e ? std::string(e) : "/dev/shm/spz_host_bench_" + std::to_string(static_cast<long>(getpid())) + ".spz";
      double t_fsave = 1e30, t_fload = 1e30;
      bool ok = true;
      for (int r = 0; r < std::max(reps, 2); ++r) {
        const double t0 = now();
        ok = spz::saveSpz(g, po, path) && ok;
        t_fsave = std::min(t_fsave, now() - t0);
      }
      for (int r = 0; r < std::max(reps, 2); ++r) {
        const double t0 = now();
        spz::GaussianCloud back = spz::loadSpz(path, uo);
        t_fload = std::min(t_fload, now() - t0);
        ok = ok && back.numPoints == static_cast<int32_t>(n);
      }
      std::remove(path.c_str());
      printf(", \"save_spz_file_s\": %.3f, \"load_spz_file_s\": %.3f, \"file_ok\": %s", t_fsave, t_fload, ok ? "true" : "false");
    }
  }
  printf("}\n");
  return 0;
}
