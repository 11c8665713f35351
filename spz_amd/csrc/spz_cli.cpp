// spz_cli.cpp — the three command-line tools of the reference (cli_tools/src/*.cpp) over the
// MI355X drop-in layer: ply_to_spz, spz_to_ply, spz_info.  One binary, dispatched on argv[0]
// (the Makefile installs it under the three names) or on a first argument naming the tool.
// Same behaviour as the reference mains: default (UNSPECIFIED) pack/unpack options, exit code 0
// once the arguments are there (the reference ignores the save/load results), usage -> 1.
#include <algorithm>
#include <cstring>
#include <exception>
#include <iostream>
#include <string>

#include "spz_amd_host.hpp"

namespace {

int plyToSpz(int argc, char **argv) {
  if (argc < 3) {
    std::cerr << "Usage: ply_to_spz <input.ply> <output.spz>" << std::endl;
    return 1;
  }
  spz::GaussianCloud splat = spz::loadSplatFromPly(argv[1], spz::UnpackOptions{});
  spz::saveSpz(splat, spz::PackOptions{}, std::string(argv[2]));
  return 0;
}

int spzToPly(int argc, char **argv) {
  if (argc < 3) {
    std::cerr << "Usage: spz_to_ply <input.spz> <output.ply>" << std::endl;
    return 1;
  }
  spz::GaussianCloud splat = spz::loadSpz(std::string(argv[1]), spz::UnpackOptions{});
  spz::saveSplatToPly(splat, spz::PackOptions{}, argv[2]);
  return 0;
}

int spzInfo(int argc, char **argv) {
  if (argc < 2) {
    std::cerr << "Usage: spz_info <input.spz>" << std::endl;
    return 1;
  }
  const spz::GaussianCloud cloud = spz::loadSpz(std::string(argv[1]), spz::UnpackOptions{});
  std::cout << "Number of points: " << cloud.positions.size() / 3 << std::endl;
  if (!cloud.positions.empty()) {
    float lo[3] = {cloud.positions[0], cloud.positions[1], cloud.positions[2]};
    float hi[3] = {lo[0], lo[1], lo[2]};
    for (size_t i = 0; i + 2 < cloud.positions.size(); i += 3) {
      for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(lo[a], cloud.positions[i + a]);
        hi[a] = std::max(hi[a], cloud.positions[i + a]);
      }
    }
    std::cout << "Bounding box:" << std::endl;
    const char *axis = "XYZ";
    for (int a = 0; a < 3; ++a) std::cout << "  " << axis[a] << ": " << lo[a] << " to " << hi[a] << std::endl;
  }
  return 0;
}

int dispatch(const std::string &tool, int argc, char **argv) {
  if (tool == "ply_to_spz") return plyToSpz(argc, argv);
  if (tool == "spz_to_ply") return spzToPly(argc, argv);
  if (tool == "spz_info") return spzInfo(argc, argv);
  return -1;
}

}  // namespace

int main(int argc, char **argv) {
  try {
    std::string self = argc > 0 ? argv[0] : "";
    const size_t slash = self.find_last_of('/');
    if (slash != std::string::npos) self = self.substr(slash + 1);
    int rc = dispatch(self, argc, argv);
    if (rc >= 0) return rc;
    if (argc >= 2) {
      rc = dispatch(argv[1], argc - 1, argv + 1);
      if (rc >= 0) return rc;
    }
    std::cerr << "Usage: spz_tool {ply_to_spz|spz_to_ply|spz_info} <args...>" << std::endl;
    return 1;
  } catch (const std::exception &e) {
    std::cerr << "Error: " << e.what() << std::endl;
    return 1;
  }
}
