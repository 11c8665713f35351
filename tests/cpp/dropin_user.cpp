// A user program written against the REFERENCE's public C++ API only (load-spz.h / splat-types.h:
// GaussianCloud incl. data(), PackOptions, UnpackOptions, CoordinateSystem, coordinateConverter, saveSpz,
// loadSpz, loadSpzPacked, PackedGaussians::at / unpack, convertCoordinates).
// It is compiled twice from this one source:
//   * against /root/reference/src/cc (the reference itself)         -> expected output (golden)
//   * against include/compat + libspz_host.so (this implementation) -> must print the same lines
// Prints FNV-1a hashes of the .spz bytes and of the decoded floats' bit patterns.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "load-spz.h"
#include "splat-types.h"

static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
  const uint8_t *b = static_cast<const uint8_t *>(p);
  for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

static uint64_t cloudHash(const spz::GaussianCloud &g) {
  uint64_t h = fnv(&g.numPoints, sizeof(g.numPoints));
  h = fnv(&g.shDegree, sizeof(g.shDegree), h);
  for (const auto *v : {&g.positions, &g.scales, &g.rotations, &g.alphas, &g.colors, &g.sh}) {
    h = fnv(v->data(), v->size() * sizeof(float), h);
  }
  return h;
}

int main() {
  for (int degree = 0; degree <= 3; ++degree) {
    const int n = 20000 + degree;
    const int shDim = degree == 0 ? 0 : (degree + 1) * (degree + 1) - 1;
    std::mt19937 rng(1234 + degree);
    std::uniform_real_distribution<float> pos(-10.f, 10.f), scale(-8.f, 0.f);
    std::normal_distribution<float> nrm(0.f, 1.f);
    spz::GaussianCloud g;
    g.numPoints = n;
    g.shDegree = degree;
    g.antialiased = (degree & 1) != 0;
    for (int i = 0; i < n * 3; ++i) g.positions.push_back(pos(rng));
    for (int i = 0; i < n * 3; ++i) g.scales.push_back(scale(rng));
    for (int i = 0; i < n * 4; ++i) g.rotations.push_back(nrm(rng));
    for (int i = 0; i < n; ++i) g.alphas.push_back(3.f * nrm(rng));
    for (int i = 0; i < n * 3; ++i) g.colors.push_back(nrm(rng));
    for (int i = 0; i < n * shDim * 3; ++i) g.sh.push_back(0.25f * nrm(rng));

    spz::PackOptions po;
    po.from = spz::CoordinateSystem::RDF;
    std::vector<uint8_t> bytes;
    const bool ok = spz::saveSpz(g, po, &bytes);
    std::printf("degree %d save %d bytes %zu hash %016llx\n", degree, ok ? 1 : 0, bytes.size(),
                (unsigned long long)fnv(bytes.data(), bytes.size()));

    spz::UnpackOptions uo;
    uo.to = spz::CoordinateSystem::LUF;
    spz::GaussianCloud back = spz::loadSpz(bytes, uo);
    std::printf("degree %d load points %d sh %d aa %d hash %016llx\n", degree, back.numPoints, back.shDegree,
                back.antialiased ? 1 : 0, (unsigned long long)cloudHash(back));

    // the C bridge view (caller frees) and per-splat access on the packed form
    GaussianCloudData cd = back.data();
    uint64_t dh = fnv(&cd.numPoints, sizeof(cd.numPoints));
    for (const SpzFloatBuffer *b : {&cd.positions, &cd.scales, &cd.rotations, &cd.alphas, &cd.colors, &cd.sh}) {
      dh = fnv(b->data, b->count * sizeof(float), dh);
      delete[] b->data;
    }
    std::printf("degree %d data() counts %zu %zu %zu hash %016llx\n", degree, cd.positions.count, cd.rotations.count,
                cd.sh.count, (unsigned long long)dh);
    spz::PackedGaussians packed = spz::loadSpzPacked(bytes);
    const spz::CoordinateConverter conv = spz::coordinateConverter(spz::CoordinateSystem::RUB, spz::CoordinateSystem::LDF);
    uint64_t ph = 0, uh = 0;
    for (int i : {0, 1, 777, n / 2, n - 1}) {
      const spz::PackedGaussian one = packed.at(i);
      ph = fnv(one.position.data(), 9, ph ? ph : 1469598103934665603ull);
      ph = fnv(one.rotation.data(), 4, ph);
      ph = fnv(&one.alpha, 1, ph);
      ph = fnv(one.shR.data(), 15, fnv(one.shG.data(), 15, fnv(one.shB.data(), 15, ph)));
      const spz::UnpackedGaussian u = packed.unpack(i, conv);
      uh = fnv(u.position.data(), 12, uh ? uh : 1469598103934665603ull);
      uh = fnv(u.rotation.data(), 16, uh);
      uh = fnv(u.scale.data(), 12, uh);
      uh = fnv(u.color.data(), 12, uh);
      uh = fnv(&u.alpha, 4, uh);
      uh = fnv(u.shR.data(), 60, fnv(u.shG.data(), 60, fnv(u.shB.data(), 60, uh)));
    }
    std::printf("degree %d packed points %d f16 %d at hash %016llx unpack hash %016llx\n", degree, packed.numPoints,
                packed.usesFloat16() ? 1 : 0, (unsigned long long)ph, (unsigned long long)uh);

    back.convertCoordinates(spz::CoordinateSystem::LUF, spz::CoordinateSystem::RUB);
    back.rotate180DegAboutX();
    std::printf("degree %d converted hash %016llx median %.9g\n", degree, (unsigned long long)cloudHash(back),
                back.medianVolume());
  }
  return 0;
}
