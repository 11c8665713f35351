#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "spz_amd_host.hpp"
int main() {
  std::mt19937 rng(11);
  std::string hdr = "ply\nformat binary_little_endian 1.0\nelement vertex 7\n";
  const char *props[] = {"x","y","z","nx","ny","nz","f_dc_0","f_dc_1","f_dc_2","f_rest_0","f_rest_1","f_rest_2","f_rest_3","f_rest_4","f_rest_5","f_rest_6","f_rest_7","f_rest_8","opacity","scale_0","scale_1","scale_2","rot_0","rot_1","rot_2","rot_3"};
  for (auto p : props) hdr += std::string("property float ") + p + "\n";
  hdr += "end_header\n";
  std::string body(7 * 26 * 4, '\1');
  size_t empties = 0, loaded = 0;
  for (int it = 0; it < 4000; ++it) {
    std::string f = hdr + body;
    int kind = rng() % 6;
    if (kind == 0) f.resize(rng() % (f.size() + 1));
    else if (kind == 1) for (int k = 0; k < 1 + (int)(rng() % 3); ++k) f[rng() % hdr.size()] = " \n0123456789abcxyz_-"[rng() % 19];
    else if (kind == 2) { size_t a = rng() % hdr.size(); f.erase(a, 1 + rng() % 12); }
    else if (kind == 3) { size_t a = rng() % hdr.size(); f.insert(a, "element vertex 99999999999\n"); }
    else if (kind == 4) { size_t a = f.find("vertex 7"); f.replace(a, 8, "vertex " + std::to_string((long long)(int32_t)rng())); }
    else { size_t a = rng() % hdr.size(); f.insert(a, std::string(1 + rng() % 300, "property float q\n"[rng() % 17])); }
    { std::ofstream o("/tmp/spz_fuzz.ply", std::ios::binary); o.write(f.data(), f.size()); }
    if (!freopen("/dev/null", "w", stdout)) return 2;
    spz::GaussianCloud g = spz::loadSplatFromPly("/tmp/spz_fuzz.ply", spz::UnpackOptions{});
    if (g.numPoints == 0) ++empties; else ++loaded;
  }
  fprintf(stderr, "empty %zu loaded %zu\n", empties, loaded);
  return 0;
}
