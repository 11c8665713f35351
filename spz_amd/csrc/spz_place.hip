// spz_place.hip — device buffers for a resident cloud, placed so that the codec kernels run at their fast rate
// (include/spz_amd.h: spz_amd_cloud_buffers_alloc / _free).
//
// What round 3's placement experiments found (tools/placement_scan.hip, profiles/r03_placement_*.jsonl): the
// ±10 % of an sh3 decode (0.46 ms or 0.55 ms for the same binary and the same data) is decided by ONE thing — whether
// the sh float array (three quarters of the bytes a decode writes, of the bytes an encode reads) lies in the same
// physical region of HBM as the other arrays of the launch.  In one hipMalloc block together with them: always the slow
// rate.  In an allocation of its own: fast for about three pairs of allocations out of four, slow for the fourth, the
// same verdict every time for a given pair; offsets inside an allocation, the virtual address, the allocation API
// (hipMalloc or hipMemCreate) and the tile order of the kernels do not change it, and neither array is slow alone
// (one buffer streams at the same rate wherever it lies).  Physical addresses are not visible from user space, so the
// fast kind cannot be asked for — but it can be recognised in half a millisecond: time the launch.  This file does
// that once, when the buffers are made: the five small arrays (and the stream, unless the caller brings one) go into
// one block, the sh array into an allocation of its own, and up to `max_candidates` such allocations are tried
// (spacers of growing size between them move the allocator on to other regions) until one is at least 13 % faster than
// the slowest seen, i.e. both kinds have been seen and this one is of the fast kind.  Unchosen candidates and
// spacers are freed before the call returns.  Cost: a few launches of the real kernel per candidate, tens of
// milliseconds in all, once per set of long-lived buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "spz_amd.h"
#include "spz_common.hpp"

namespace spz_amd_detail {
namespace {

struct PlacedOwner {
  int device = 0;
  void *rest = nullptr;  // positions, scales, rotations, alphas, colours (+ the stream)
  void *sh = nullptr;
};

size_t up2m(size_t v) { return (v + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1); }

}  // namespace
}  // namespace spz_amd_detail

using namespace spz_amd_detail;

extern "C" {

int spz_amd_cloud_buffers_alloc(uint64_t num_points, int sh_degree, int version, uint8_t *d_stream, int probe, int max_candidates,
                                void *hip_stream, spz_amd_cloud_buffers *out) {
  if (out == nullptr || probe < 0 || probe > 2) return SPZ_AMD_ERR_INVALID_ARG;
  *out = spz_amd_cloud_buffers();
  spz_amd_layout lay;
  int rc = spz_amd_stream_layout(num_points, sh_degree, version, &lay);
  if (rc != SPZ_AMD_OK) return rc;
  if (num_points == 0 || num_points > 0xffffffffull) return SPZ_AMD_ERR_INVALID_ARG;
  if (probe == 2 && version == 1) return SPZ_AMD_ERR_UNSUPPORTED;
  int device = 0;
  rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  const int sd = sh_dim_for_degree(sh_degree);
  const size_t n = (size_t)num_points;
  const size_t floats[6] = {3 * n, 3 * n, 4 * n, n, 3 * n, (size_t)sd * 3 * n};
  size_t off[7], at = 0;
  for (int k = 0; k < 5; ++k) {
    off[k] = at;
    at = up2m(at + floats[k] * sizeof(float));
  }
  off[6] = at;
  const bool own_stream = d_stream == nullptr;
  if (own_stream) at = up2m(at + lay.total_bytes + 64);
  PlacedOwner *owner = new (std::nothrow) PlacedOwner();
  if (owner == nullptr) return SPZ_AMD_ERR_HIP;
  owner->device = device;
  std::vector<void *> held;  // spacers, unchosen candidates, unchosen blocks
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto fail = [&](int code) {
    for (void *p : held) (void)hipFree(p);
    if (owner->sh) (void)hipFree(owner->sh);
    if (owner->rest) (void)hipFree(owner->rest);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    delete owner;
    return code;
  };
  auto cloud_in = [&](void *rest_block) {
    char *rest = static_cast<char *>(rest_block);
    return spz_amd_cloud_out{reinterpret_cast<float *>(rest + off[0]), reinterpret_cast<float *>(rest + off[1]),
                             reinterpret_cast<float *>(rest + off[2]), reinterpret_cast<float *>(rest + off[3]),
                             reinterpret_cast<float *>(rest + off[4]), nullptr};
  };
  auto stream_in = [&](void *rest_block) { return own_stream ? reinterpret_cast<uint8_t *>(static_cast<char *>(rest_block) + off[6]) : d_stream; };
  if (hipMalloc(&owner->rest, at) != hipSuccess) return fail(SPZ_AMD_ERR_HIP);
  out->stream_capacity = own_stream ? lay.total_bytes + 64 : lay.total_bytes;
  if (sd == 0) {  // no sh array: nothing to place
    out->cloud = cloud_in(owner->rest);
    out->stream = stream_in(owner->rest);
    out->owner = owner;
    return SPZ_AMD_OK;
  }
  const size_t sh_bytes = floats[5] * sizeof(float);
  // Below a few hundred megabytes the launch is over before placement shows; above, time the real kernel.
  const bool timed = probe != 0 && sh_bytes >= (size_t(256) << 20) && max_candidates > 1;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (timed && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) return fail(SPZ_AMD_ERR_HIP);
  const spz_amd_header hdr = {(uint32_t)version, (uint32_t)num_points, (uint8_t)sh_degree, 12, 0, 0};
  auto probe_ms = [&](void *rest_block, float *sh, float *ms) {  // the launch the buffers are for, on zeroed buffers
    spz_amd_cloud_out c = cloud_in(rest_block);
    c.sh = sh;
    uint8_t *stream = stream_in(rest_block);
    const spz_amd_cloud_in cin = {c.positions, c.scales, c.rotations, c.alphas, c.colors, c.sh};
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      if (hipEventRecord(e0, st) != hipSuccess) return (int)SPZ_AMD_ERR_HIP;
      const int prc = probe == 1 ? spz_amd_decode_device(stream, lay.total_bytes, &hdr, SPZ_AMD_UNSPECIFIED, &c, st)
                                 : spz_amd_encode_device(&cin, num_points, sh_degree, 0, SPZ_AMD_UNSPECIFIED, version, stream, lay.total_bytes, st);
      if (prc != SPZ_AMD_OK) return prc;
      float t = 0;
      if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) {
        return (int)SPZ_AMD_ERR_HIP;
      }
      if (r > 0 && t < best) best = t;  // the first launch warms up
    }
    *ms = best;
    return (int)SPZ_AMD_OK;
  };
  auto spacer = [&](int k) {  // moves the allocator on: what comes next should come from another region
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    const size_t want = std::min<size_t>((size_t)(k + 1) * (size_t(6) << 30), free_b / 6);
    void *sp = nullptr;
    if (want >= (size_t(1) << 30) && hipMalloc(&sp, want) == hipSuccess) held.push_back(sp);
  };
  const int tries = timed ? std::min(max_candidates, 12) : 1;
  // Up to three blocks for the small arrays, each tried with up to `tries` sh allocations: when every sh candidate next to
  // one block is of the slow kind (about one case in ten with six candidates: neighbours in the allocator's order tend to
  // be neighbours in memory), another block elsewhere is an independent draw.
  const int rounds = timed ? 3 : 1;
  float best_ms = 1e30f, worst_ms = 0.0f, first_ms = 0.0f;
  void *best_sh = nullptr, *best_rest = nullptr, *rest = owner->rest;
  owner->rest = nullptr;  // tracked in rest / best_rest / held from here on
  int tried = 0;
  bool accepted = false;
  auto give_up = [&](int code) {
    if (rest && rest != best_rest) held.push_back(rest);
    if (best_rest) held.push_back(best_rest);
    if (best_sh) held.push_back(best_sh);
    return fail(code);
  };
  for (int round = 0; round < rounds && !accepted; ++round) {
    if (round > 0) {
      spacer(round + 2);
      rest = nullptr;
      if (hipMalloc(&rest, at) != hipSuccess) break;  // the card is filling up: what has been found will do
    }
    if (timed) {
      // zeros on both sides: every candidate is timed on the same values (no denormal / NaN paths from stale memory);
      // quaternions (1, 1, 1, 1) rather than zero ones: the encode's ordinary path, not its zero-norm one
      (void)hipMemsetAsync(rest, 0, own_stream ? at : off[6], st);
      (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cloud_in(rest).rotations), 0x3f800000, floats[2], st);
    }
    for (int c = 0; c < tries && !accepted; ++c) {
      void *cand = nullptr;
      if (hipMalloc(&cand, sh_bytes) != hipSuccess) {
        if (best_sh != nullptr) break;
        return give_up(SPZ_AMD_ERR_HIP);
      }
      float ms = 0.0f;
      if (timed) {
        (void)hipMemsetAsync(cand, 0, sh_bytes, st);
        rc = probe_ms(rest, static_cast<float *>(cand), &ms);
        if (rc != SPZ_AMD_OK) {
          held.push_back(cand);
          return give_up(rc);
        }
      }
      ++tried;
      if (tried == 1) first_ms = ms;
      worst_ms = std::max(worst_ms, ms);
      if (best_sh == nullptr || ms < best_ms) {
        if (best_sh) held.push_back(best_sh);
        if (best_rest && best_rest != rest) held.push_back(best_rest);
        best_sh = cand;
        best_rest = rest;
        best_ms = ms;
      } else {
        held.push_back(cand);
      }
      // Both kinds seen and the best clearly of the fast one (0.46 against 0.54 ms for 10 M sh3 points: a ratio of 0.85);
      // a ratio of ~0.9 is a placement in between (part of the traffic conflicts), worth another try.
      accepted = !timed || (tried >= 2 && best_ms <= 0.87f * worst_ms);
      // an encode reads its floats once and is hardly moved by where they lie (every candidate within 3-4 % in every
      // session): three alike are the answer
      if (probe == 2 && tried >= 3 && worst_ms <= 1.06f * best_ms) accepted = true;
      if (!accepted) spacer(c);
    }
    if (rest != best_rest) held.push_back(rest);
    rest = nullptr;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  e0 = e1 = nullptr;
  (void)hipStreamSynchronize(st);
  for (void *p : held) (void)hipFree(p);
  held.clear();
  owner->rest = best_rest;
  owner->sh = best_sh;
  spz_amd_cloud_out cloud = cloud_in(best_rest);
  cloud.sh = static_cast<float *>(best_sh);
  out->cloud = cloud;
  out->stream = stream_in(best_rest);
  out->owner = owner;
  out->candidates = tried;
  out->probe_ms_first = first_ms;
  out->probe_ms_chosen = timed ? best_ms : 0.0f;
  out->probe_ms_worst = worst_ms;
  if (std::getenv("SPZ_AMD_LZ_TIMING")) {
    std::fprintf(stderr, "[place] %d sh placement(s) timed (%s launch): first %.3f ms, chosen %.3f ms, slowest %.3f ms\n", tried,
                 probe == 1 ? "decode" : probe == 2 ? "encode" : "no", first_ms, best_ms, worst_ms);
  }
  return SPZ_AMD_OK;
}

int spz_amd_cloud_buffers_free(spz_amd_cloud_buffers *b) {
  if (b == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  PlacedOwner *o = static_cast<PlacedOwner *>(b->owner);
  if (o != nullptr) {
    DeviceGuard guard;
    if (guard.enter(o->device) == SPZ_AMD_OK) {
      if (o->sh) (void)hipFree(o->sh);
      if (o->rest) (void)hipFree(o->rest);
    }
    delete o;
  }
  *b = spz_amd_cloud_buffers();
  return SPZ_AMD_OK;
}

}  // extern "C"
