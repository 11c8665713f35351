// spz_abi.hip — host side of libspz_amd.so: the C ABI of include/spz_amd.h over the kernels of
// spz_kernels.hip.  Stream geometry and header (load-spz.cc:131-139,533-596), coordinate-flip masks
// (splat-types.h:43-81), the host-computed decode tables / alpha thresholds, section-table
// construction and launches.  The host-pointer (PCIe) variants and their cached device workspace are
// spz_hostpath.hip, which calls the device entry points below.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "spz_amd.h"
#include "spz_common.hpp"
#include "spz_kernel_params.hpp"

namespace spz_amd_detail {
thread_local int g_last_hip_error = 0;

}  // namespace spz_amd_detail

namespace {

using namespace spz_amd_detail;

// ==========================================================================================
// Host side
// ==========================================================================================
constexpr uint32_t kMagic = 0x5053474eu;  // load-spz.cc:132

// Per-element mask over a point's D = 3*shDim sh floats, with elements 0..2 repeated at D..D+2
// so that a 4-element window starting anywhere in [0, D) can be read with one shift.
unsigned long long sh_elem_mask_ext(uint32_t sh15, int sh_dim) {
  unsigned long long m = 0;
  const int D = sh_dim * 3;
  for (int j = 0; j < D; ++j) {
    if ((sh15 >> (j / 3)) & 1u) m |= 1ull << j;
  }
  if (D > 0) m |= (m & 7ull) << D;
  return m;
}

// ---- tables -----------------------------------------------------------------------------
struct Tables {
  float host[kTableFloats];
  float *dev[kMaxDevices];
  bool host_ready;
  bool monotone;
};
Tables g_tables = {};
std::mutex g_tables_mutex;

// The reference's own expressions (load-spz.cc:74,85,87,301,518,522), evaluated with the host
// libm the reference itself links against; volatile stops the host compiler from folding or
// contracting anything.
float ref_sigmoid_host(float x) {
  volatile float e = std::exp(-x);
  volatile float d = 1 + e;
  volatile float s = 1 / d;
  return s;
}
uint8_t ref_alpha_byte_host(float a) {
  volatile float t = ref_sigmoid_host(a) * 255.0f;
  float r = std::round(t);
  r = (r < 0.0f) ? 0.0f : ((255.0f < r) ? 255.0f : r);
  return (uint8_t)(int)r;
}
uint32_t float_key(float f) {  // monotone map float -> uint32 (total order)
  uint32_t b;
  std::memcpy(&b, &f, 4);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
float key_float(uint32_t k) {
  uint32_t b = (k >> 31) ? (k & 0x7fffffffu) : ~k;
  float f;
  std::memcpy(&f, &b, 4);
  return f;
}

void build_host_tables() {
  float *t = g_tables.host;
  for (int b = 0; b < 256; ++b) {
    volatile float x = b / 255.0f;
    volatile float om = 1.0f - x;
    volatile float ratio = x / om;
    t[kTableAlphaDec + b] = std::log(ratio);  // invSigmoid, load-spz.cc:87
    volatile float c0 = x - 0.5f;
    t[kTableColorDec + b] = c0 / 0.15f;       // load-spz.cc:522
  }
  // thresholds: thr[v-1] = smallest float a (total order) with alpha_byte(a) >= v
  const uint32_t klo = float_key(-INFINITY), khi = float_key(INFINITY);
  for (int v = 1; v <= 255; ++v) {
    uint32_t lo = klo, hi = khi;  // byte(lo) = 0 < v <= 255 = byte(hi)
    while (hi - lo > 1) {
      uint32_t mid = lo + (hi - lo) / 2;
      if (ref_alpha_byte_host(key_float(mid)) >= v) hi = mid;
      else lo = mid;
    }
    t[kTableAlphaThr + v - 1] = key_float(hi);
  }
  t[kTableAlphaThr + 255] = NAN;
  // The threshold form is exact only if toUint8(sigmoid(a) * 255) is monotone in a with this libm.
  // Check the step at every threshold over +-8 ulp; a libm that breaks it makes the alpha encode
  // fail loudly (SPZ_AMD_ERR_UNSUPPORTED) instead of silently differing from the reference.
  g_tables.monotone = true;
  for (int v = 1; v <= 255 && g_tables.monotone; ++v) {
    const uint32_t k = float_key(t[kTableAlphaThr + v - 1]);
    for (int d = -8; d <= 8; ++d) {
      const int byte = ref_alpha_byte_host(key_float(k + (uint32_t)d));
      if ((d < 0 && byte >= v) || (d >= 0 && byte < v)) g_tables.monotone = false;
    }
    if (v > 1 && !(t[kTableAlphaThr + v - 2] < t[kTableAlphaThr + v - 1])) g_tables.monotone = false;
  }
  g_tables.host_ready = true;
}

int ensure_tables(int device, const float **dev_tables) {
  if (device < 0 || device >= kMaxDevices) return SPZ_AMD_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  if (!g_tables.host_ready) build_host_tables();
  if (!g_tables.monotone) return SPZ_AMD_ERR_UNSUPPORTED;
  if (g_tables.dev[device] == nullptr) {
    float *d = nullptr;
    SPZ_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(float) * kTableFloats));
    hipError_t e = hipMemcpy(d, g_tables.host, sizeof(float) * kTableFloats, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      g_last_hip_error = (int)e;
      (void)hipFree(d);
      return SPZ_AMD_ERR_HIP;
    }
    g_tables.dev[device] = d;
  }
  *dev_tables = g_tables.dev[device];
  return SPZ_AMD_OK;
}

int grid_for(int device, uint32_t total_tiles, uint32_t *grid) {
  uint32_t cap = 0x7fffffffu;  // hipGridDim.x limit; the kernels grid-stride beyond it
  if (kMaxBlocksPerCU > 0) {
    static int cus[kMaxDevices] = {0};
    if (cus[device] == 0) {
      int n = 0;
      SPZ_HIP_TRY(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
      cus[device] = (n > 0) ? n : 256;
    }
    cap = (uint32_t)cus[device] * (uint32_t)kMaxBlocksPerCU;
  }
  uint32_t g = total_tiles < cap ? total_tiles : cap;
#if SPZ_XCD_REMAP
  g = ((total_tiles + 7u) / 8u) * 8u;  // every XCD gets ceil(tiles / 8) blocks; surplus blocks exit
#endif
  *grid = g > 0 ? g : 1;
  return SPZ_AMD_OK;
}

// Appends a section to the fused grid (skips empty ones).
void add_section(KParams *p, uint32_t tile_units, uint32_t kind, uint8_t *bytes, float *floats,
                 unsigned long long n_elems) {
  if (n_elems == 0) return;
  SecDesc &s = p->sec[p->n_sec++];
  s.bytes = bytes;
  s.floats = floats;
  s.n_elems = n_elems;
  s.n_units = (n_elems + 3) / 4;
  s.tile_begin = p->total_tiles;
  s.tile_skip = 0;
  s.kind = kind;
  p->total_tiles += (uint32_t)((s.n_units + tile_units - 1) / tile_units);
}

// Turns the sequential grid (section after section) into: an interleaved part, in which every period of
// kIlPeriod tiles holds each section's slots in proportion to its tile count and evenly spread, followed by
// a sequential part with what is left of each section.
void interleave_sections(KParams *p) {
  const uint32_t n = p->n_sec, total = p->total_tiles;
  if (n < 2 || total < 4 * kIlPeriod) return;
  constexpr uint32_t G = SPZ_IL_GROUP;  // tiles per pattern slot
  uint32_t tiles[SPZ_AMD_NUM_SECTIONS], count[SPZ_AMD_NUM_SECTIONS], used = 0;
  uint32_t big = 0, part_total = 0;
  for (uint32_t k = 0; k < n; ++k) {
    tiles[k] = (k + 1 < n ? p->sec[k + 1].tile_begin : total) - p->sec[k].tile_begin;
    big = tiles[k] > tiles[big] ? k : big;
  }
  auto takes_part = [&](uint32_t k) {
    return !SPZ_IL_ONLY_ROT || k == big || p->sec[k].kind == KIND_ROT_S3 || p->sec[k].kind == KIND_ROT_F3;
  };
  for (uint32_t k = 0; k < n; ++k) part_total += takes_part(k) ? tiles[k] : 0u;
  for (uint32_t k = 0; k < n; ++k) {
    count[k] = takes_part(k) ? (uint32_t)(((unsigned long long)tiles[k] * kIlPeriod + part_total / 2) / part_total) : 0u;
    used += count[k];
  }
  // make the slot counts add up to the period: give to / take from the largest section
  if (used > kIlPeriod && count[big] <= used - kIlPeriod) return;
  count[big] = count[big] + kIlPeriod - used;
  uint32_t reps = 0xffffffffu;
  for (uint32_t k = 0; k < n; ++k) {
    if (count[k]) reps = tiles[k] / (count[k] * G) < reps ? tiles[k] / (count[k] * G) : reps;
  }
  if (reps == 0 || reps == 0xffffffffu) return;
  // slot order: the i-th slot of section k wants position (i + 1/2) * period / count[k]
  uint32_t given[SPZ_AMD_NUM_SECTIONS] = {};
  for (uint32_t slot = 0; slot < kIlPeriod; ++slot) {
    uint32_t best = n;
    unsigned long long best_pos = ~0ull;
    for (uint32_t k = 0; k < n; ++k) {
      if (given[k] >= count[k]) continue;
      const unsigned long long pos = (2ull * given[k] + 1ull) * kIlPeriod * 1024ull / (2ull * count[k]);
      if (pos < best_pos) {
        best_pos = pos;
        best = k;
      }
    }
    p->il_sec[slot] = (uint8_t)best;
    p->il_rank[slot] = (uint8_t)given[best];
    ++given[best];
  }
  p->il_tiles = reps * kIlPeriod * G;
  uint32_t begin = p->il_tiles;
  for (uint32_t k = 0; k < n; ++k) {
    p->il_count[k] = (uint8_t)count[k];
    p->sec[k].tile_skip = reps * count[k] * G;
    p->sec[k].tile_begin = begin;
    begin += tiles[k] - reps * count[k] * G;
  }
}

// SPZ_AMD_GRID_ORDER (read at every launch): "sequential" / "interleaved" override the per-launch policy of the tile
// order — which of the two is faster for an sh3 decode depends on where the buffers lie, so a caller that
// can measure its own placement may want to choose.  Anything else: the policy.
int grid_order_override() {
  const char *e = std::getenv("SPZ_AMD_GRID_ORDER");
  if (e == nullptr) return 0;
  if (std::strcmp(e, "sequential") == 0) return 1;
  if (std::strcmp(e, "interleaved") == 0) return 2;
  return 0;
}

int layout_impl(uint64_t n, int sh_degree, int version, spz_amd_layout *out) {
  const int sd = sh_dim_for_degree(sh_degree);
  if (sd < 0 || version < 1 || version > 3 || out == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  const uint32_t bpp[SPZ_AMD_NUM_SECTIONS] = {version == 1 ? 6u : 9u, 1u, 3u, 3u, version >= 3 ? 4u : 3u,
                                               (uint32_t)sd * 3u};
  uint64_t off = 16;
  for (int s = 0; s < SPZ_AMD_NUM_SECTIONS; ++s) {
    out->bytes_per_point[s] = bpp[s];
    out->offset[s] = off;
    out->bytes[s] = n * bpp[s];
    off += out->bytes[s];
  }
  out->total_bytes = off;
  return SPZ_AMD_OK;
}

int encode_impl(const spz_amd_cloud_in *cl, uint64_t first, uint64_t count, uint64_t n_total, int sh_degree,
                int antialiased, int from_coord, int version, int write_header, uint8_t *d_stream,
                size_t capacity, void *hip_stream, unsigned section_mask = 0x3fu) {
  if (cl == nullptr || d_stream == nullptr || !valid_coord(from_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  if (version == 1) return SPZ_AMD_ERR_UNSUPPORTED;
  spz_amd_layout lay;
  int rc = layout_impl(n_total, sh_degree, version, &lay);
  if (rc != SPZ_AMD_OK) return rc;
  if (n_total > 0xffffffffull || first > n_total || count > n_total - first) return SPZ_AMD_ERR_INVALID_ARG;
  if (capacity < lay.total_bytes) return SPZ_AMD_ERR_CAPACITY;
  const int sd = sh_dim_for_degree(sh_degree);
  if (count > 0 && (!cl->positions || !cl->scales || !cl->rotations || !cl->alphas || !cl->colors ||
                    (sd > 0 && !cl->sh))) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  int device = 0;
  rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  KParams p = {};
  rc = ensure_tables(device, &p.tables);
  if (rc != SPZ_AMD_OK) return rc;

  const FlipMasks fm = flip_masks(from_coord, SPZ_AMD_RUB);  // load-spz.cc:263
  p.flip_p = fm.p;
  p.flip_q = fm.q;
  p.sh_mask_ext = sh_elem_mask_ext(fm.sh15, sd);
  p.sh_d = (uint32_t)sd * 3u;
  p.pos_scale = 0.0f;
  auto frag = [&](int s) { return d_stream + lay.offset[s] + first * lay.bytes_per_point[s]; };
  auto fl = [](const float *q) { return const_cast<float *>(q); };
  auto want = [&](int s) { return ((section_mask >> s) & 1u) != 0u; };
  // Largest sections first so the tail of the grid is made of the small ones.
  if (want(SPZ_AMD_SEC_SH)) add_section(&p, EncGeom::kTileUnits, KIND_SH, frag(SPZ_AMD_SEC_SH), fl(cl->sh), count * (uint64_t)sd * 3u);
  if (want(SPZ_AMD_SEC_POSITIONS)) add_section(&p, EncGeom::kTileUnits, KIND_POS24, frag(SPZ_AMD_SEC_POSITIONS), fl(cl->positions), count * 3u);
  if (want(SPZ_AMD_SEC_ROTATIONS)) {
    add_section(&p, EncGeom::kTileUnits, version >= 3 ? KIND_ROT_S3 : KIND_ROT_F3, frag(SPZ_AMD_SEC_ROTATIONS), fl(cl->rotations),
                count * 4u);
  }
  if (want(SPZ_AMD_SEC_SCALES)) add_section(&p, EncGeom::kTileUnits, KIND_SCALE, frag(SPZ_AMD_SEC_SCALES), fl(cl->scales), count * 3u);
  if (want(SPZ_AMD_SEC_COLORS)) add_section(&p, EncGeom::kTileUnits, KIND_COLOR, frag(SPZ_AMD_SEC_COLORS), fl(cl->colors), count * 3u);
  if (want(SPZ_AMD_SEC_ALPHAS)) add_section(&p, EncGeom::kTileUnits, KIND_ALPHA, frag(SPZ_AMD_SEC_ALPHAS), fl(cl->alphas), count);
  if (write_header) {
    p.header_dst = d_stream;
    p.header_words[0] = kMagic;
    p.header_words[1] = (uint32_t)version;
    p.header_words[2] = (uint32_t)n_total;
    p.header_words[3] = (uint32_t)sh_degree | (12u << 8) | ((antialiased ? 1u : 0u) << 16);
  }
  if (p.total_tiles == 0 && !write_header) return SPZ_AMD_OK;
  {
    const int order = grid_order_override();
    if (order == 2 || (order == 0 && (SPZ_ENC_INTERLEAVE == 1 || (SPZ_ENC_INTERLEAVE == 2 && sd == 0)))) interleave_sections(&p);
  }
  uint32_t grid = 1;
  rc = grid_for(device, p.total_tiles, &grid);
  if (rc != SPZ_AMD_OK) return rc;
  hipLaunchKernelGGL(spz_encode_kernel, dim3(grid), dim3(EncGeom::kBlock), 0, static_cast<hipStream_t>(hip_stream), p);
  SPZ_HIP_TRY(hipGetLastError());
  return SPZ_AMD_OK;
}

int check_header_fields(const spz_amd_header *h) {
  if (h == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  if (h->version < 1 || h->version > 3) return SPZ_AMD_ERR_VERSION;
  if (h->sh_degree > 3) return SPZ_AMD_ERR_SH_DEGREE;
  return SPZ_AMD_OK;
}

int decode_impl(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr, uint64_t first, uint64_t count,
                int to_coord, const spz_amd_cloud_out *cl, void *hip_stream) {
  if (d_stream == nullptr || cl == nullptr || !valid_coord(to_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  int rc = check_header_fields(hdr);
  if (rc != SPZ_AMD_OK) return rc;
  const uint64_t n_total = hdr->num_points;
  if (first > n_total || count > n_total - first) return SPZ_AMD_ERR_INVALID_ARG;
  spz_amd_layout lay;
  rc = layout_impl(n_total, hdr->sh_degree, (int)hdr->version, &lay);
  if (rc != SPZ_AMD_OK) return rc;
  if (size < lay.total_bytes) return SPZ_AMD_ERR_SHORT_STREAM;  // load-spz.cc:591-594
  const int sd = sh_dim_for_degree(hdr->sh_degree);
  if (count > 0 && (!cl->positions || !cl->scales || !cl->rotations || !cl->alphas || !cl->colors ||
                    (sd > 0 && !cl->sh))) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  if (count == 0) return SPZ_AMD_OK;
  int device = 0;
  rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  KParams p = {};
  rc = ensure_tables(device, &p.tables);
  if (rc != SPZ_AMD_OK) return rc;

  const FlipMasks fm = flip_masks(SPZ_AMD_RUB, to_coord);  // load-spz.cc:529
  p.flip_p = fm.p;
  p.flip_q = fm.q;
  p.sh_mask_ext = sh_elem_mask_ext(fm.sh15, sd);
  p.sh_d = (uint32_t)sd * 3u;
  // float scale = 1.0 / (1 << fractionalBits) (load-spz.cc:495); x86 masks the shift count to 5 bits
  p.pos_scale = (float)(1.0 / (double)(int32_t)(1u << (hdr->fractional_bits & 31)));
  auto frag = [&](int s) { return const_cast<uint8_t *>(d_stream) + lay.offset[s] + first * lay.bytes_per_point[s]; };
  add_section(&p, DecGeom::kTileUnits, KIND_SH, frag(SPZ_AMD_SEC_SH), cl->sh, count * (uint64_t)sd * 3u);
  add_section(&p, DecGeom::kTileUnits, hdr->version == 1 ? KIND_POS16 : KIND_POS24, frag(SPZ_AMD_SEC_POSITIONS), cl->positions,
              count * 3u);
  add_section(&p, DecGeom::kTileUnits, hdr->version >= 3 ? KIND_ROT_S3 : KIND_ROT_F3, frag(SPZ_AMD_SEC_ROTATIONS), cl->rotations,
              count * 4u);
  add_section(&p, DecGeom::kTileUnits, KIND_SCALE, frag(SPZ_AMD_SEC_SCALES), cl->scales, count * 3u);
  add_section(&p, DecGeom::kTileUnits, KIND_COLOR, frag(SPZ_AMD_SEC_COLORS), cl->colors, count * 3u);
  add_section(&p, DecGeom::kTileUnits, KIND_ALPHA, frag(SPZ_AMD_SEC_ALPHAS), cl->alphas, count);
#if SPZ_DEC_REVERSE && SPZ_DEC_HOT_MIB > 0
  {  // tiles that cover the last-written SPZ_DEC_HOT_MIB of the stream, counted from the last tile
    unsigned long long budget = (unsigned long long)SPZ_DEC_HOT_MIB << 20, plain = 0;
    for (int k = (int)p.n_sec - 1; k >= 0 && budget > 0; --k) {
      const SecDesc &sd_ = p.sec[k];
      const unsigned long long tiles = (k + 1 < (int)p.n_sec ? p.sec[k + 1].tile_begin : p.total_tiles) - sd_.tile_begin;
      const unsigned unit_bytes = sd_.kind == KIND_POS24 ? 12u : sd_.kind == KIND_POS16 ? 8u : sd_.kind == KIND_ROT_F3 ? 3u : 4u;
      const unsigned long long bytes = sd_.n_units * unit_bytes;
      if (bytes <= budget) {
        plain += tiles;
        budget -= bytes;
      } else {
        plain += tiles * budget / bytes;
        budget = 0;
      }
    }
    p.plain_tiles = (uint32_t)plain;
  }
#endif
  // policy: with an sh section (every box measured gains at degrees 1 and 2; at degree 3 with runs of 8 tiles, see
  // SPZ_IL_GROUP); not without sh (loses up to 13 %)
  if (!SPZ_DEC_REVERSE) {
    const int order = grid_order_override();
    if (order == 2 || (order == 0 && (SPZ_DEC_INTERLEAVE == 1 || (SPZ_DEC_INTERLEAVE == 2 && sd > 0)))) interleave_sections(&p);
  }
  uint32_t grid = 1;
  rc = grid_for(device, p.total_tiles, &grid);
  if (rc != SPZ_AMD_OK) return rc;
  hipLaunchKernelGGL(spz_decode_kernel, dim3(grid), dim3(DecGeom::kBlock), 0, static_cast<hipStream_t>(hip_stream), p);
  SPZ_HIP_TRY(hipGetLastError());
  return SPZ_AMD_OK;
}


}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int spz_amd_abi_version(void) { return SPZ_AMD_ABI_VERSION; }

const char *spz_amd_status_string(int status) {
  switch (status) {
    case SPZ_AMD_OK: return "ok";
    case SPZ_AMD_ERR_INVALID_ARG: return "invalid argument";
    case SPZ_AMD_ERR_HEADER_NOT_FOUND: return "header not found";
    case SPZ_AMD_ERR_VERSION: return "version not supported";
    case SPZ_AMD_ERR_TOO_MANY_POINTS: return "too many points";
    case SPZ_AMD_ERR_SH_DEGREE: return "unsupported SH degree";
    case SPZ_AMD_ERR_SHORT_STREAM: return "read error";
    case SPZ_AMD_ERR_CAPACITY: return "output buffer too small";
    case SPZ_AMD_ERR_NO_DEVICE: return "no usable HIP device";
    case SPZ_AMD_ERR_HIP: return "HIP runtime error";
    case SPZ_AMD_ERR_UNSUPPORTED: return "unsupported operation";
    case SPZ_AMD_ERR_COMM: return "RCCL error";
    case SPZ_AMD_ERR_VERIFY: return "self-check failed";
    default: return "unknown status";
  }
}

int spz_amd_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return 0;
  }
  return n;
}

int spz_amd_last_hip_error(void) { return g_last_hip_error; }

int spz_amd_release_device_memory(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SPZ_AMD_OK;  // nothing can have been allocated
  workspace_free_all();
  int prev = 0;
  SPZ_HIP_TRY(hipGetDevice(&prev));
  {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    for (int d = 0; d < kMaxDevices && d < n; ++d) {
      if (g_tables.dev[d]) {
        if (hipSetDevice(d) == hipSuccess) (void)hipFree(g_tables.dev[d]);
        g_tables.dev[d] = nullptr;
      }
    }
  }
  SPZ_HIP_TRY(hipSetDevice(prev));
  return SPZ_AMD_OK;
}

int spz_amd_stream_layout(uint64_t num_points, int sh_degree, int version, spz_amd_layout *out) {
  return layout_impl(num_points, sh_degree, version, out);
}

int spz_amd_write_header(const spz_amd_header *hdr, uint8_t out16[16]) {
  if (hdr == nullptr || out16 == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  const uint32_t w[3] = {kMagic, hdr->version, hdr->num_points};
  for (int i = 0; i < 12; ++i) out16[i] = (uint8_t)(w[i >> 2] >> ((i & 3) * 8));
  out16[12] = hdr->sh_degree;
  out16[13] = hdr->fractional_bits;
  out16[14] = hdr->flags;
  out16[15] = hdr->reserved;
  return SPZ_AMD_OK;
}

int spz_amd_peek_header_ex(const uint8_t *stream, size_t size, uint64_t max_points, spz_amd_header *out) {
  if (out == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  if (stream == nullptr) return size == 0 ? SPZ_AMD_ERR_HEADER_NOT_FOUND : SPZ_AMD_ERR_INVALID_ARG;
  auto u32 = [&](int o) {
    return (uint32_t)stream[o] | ((uint32_t)stream[o + 1] << 8) | ((uint32_t)stream[o + 2] << 16) |
           ((uint32_t)stream[o + 3] << 24);
  };
  if (size < 16 || u32(0) != kMagic) return SPZ_AMD_ERR_HEADER_NOT_FOUND;
  spz_amd_header h;
  h.version = u32(4);
  h.num_points = u32(8);
  h.sh_degree = stream[12];
  h.fractional_bits = stream[13];
  h.flags = stream[14];
  h.reserved = stream[15];
  if (h.version < 1 || h.version > 3) return SPZ_AMD_ERR_VERSION;
  if (max_points != 0 && h.num_points > max_points) return SPZ_AMD_ERR_TOO_MANY_POINTS;
  if (h.sh_degree > 3) return SPZ_AMD_ERR_SH_DEGREE;
  spz_amd_layout lay;
  int rc = layout_impl(h.num_points, h.sh_degree, (int)h.version, &lay);
  if (rc != SPZ_AMD_OK) return rc;
  if (size < lay.total_bytes) return SPZ_AMD_ERR_SHORT_STREAM;
  *out = h;
  return SPZ_AMD_OK;
}

int spz_amd_peek_header_device(const uint8_t *d_stream, size_t size, uint64_t max_points, spz_amd_header *out,
                               void *hip_stream) {
  if (out == nullptr || d_stream == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  if (size < 16) return SPZ_AMD_ERR_HEADER_NOT_FOUND;
  int device = 0;
  int rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  uint8_t raw[16];
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  SPZ_HIP_TRY(hipMemcpyAsync(raw, d_stream, 16, hipMemcpyDeviceToHost, st));
  SPZ_HIP_TRY(hipStreamSynchronize(st));
  // validate the fields on the 16 bytes, then the size against the full device stream
  spz_amd_header h;
  rc = spz_amd_peek_header_ex(raw, 16, max_points, &h);
  if (rc != SPZ_AMD_OK && rc != SPZ_AMD_ERR_SHORT_STREAM) return rc;
  if (rc == SPZ_AMD_ERR_SHORT_STREAM) {
    // the 16-byte probe is always "short" unless the stream is empty: redo the size check with `size`
    auto u32 = [&](int o) {
      return (uint32_t)raw[o] | ((uint32_t)raw[o + 1] << 8) | ((uint32_t)raw[o + 2] << 16) | ((uint32_t)raw[o + 3] << 24);
    };
    h.version = u32(4);
    h.num_points = u32(8);
    h.sh_degree = raw[12];
    h.fractional_bits = raw[13];
    h.flags = raw[14];
    h.reserved = raw[15];
    spz_amd_layout lay;
    rc = layout_impl(h.num_points, h.sh_degree, (int)h.version, &lay);
    if (rc != SPZ_AMD_OK) return rc;
    if (size < lay.total_bytes) return SPZ_AMD_ERR_SHORT_STREAM;
  }
  *out = h;
  return SPZ_AMD_OK;
}

int spz_amd_peek_header(const uint8_t *stream, size_t size, spz_amd_header *out) {
  return spz_amd_peek_header_ex(stream, size, SPZ_AMD_REFERENCE_MAX_POINTS, out);
}

int spz_amd_encode_device(const spz_amd_cloud_in *d_cloud, uint64_t num_points, int sh_degree, int antialiased,
                          int from_coord, int version, uint8_t *d_stream, size_t capacity, void *hip_stream) {
  return encode_impl(d_cloud, 0, num_points, num_points, sh_degree, antialiased, from_coord, version, 1, d_stream,
                     capacity, hip_stream);
}

int spz_amd_encode_shard_device(const spz_amd_cloud_in *d_cloud, uint64_t first, uint64_t count,
                                uint64_t num_points_total, int sh_degree, int antialiased, int from_coord,
                                int version, int write_header, uint8_t *d_stream, size_t capacity,
                                void *hip_stream) {
  return encode_impl(d_cloud, first, count, num_points_total, sh_degree, antialiased, from_coord, version,
                     write_header, d_stream, capacity, hip_stream);
}

int spz_amd_encode_shard_sections_device(const spz_amd_cloud_in *d_cloud, uint64_t first, uint64_t count,
                                         uint64_t num_points_total, int sh_degree, int antialiased, int from_coord,
                                         int version, int write_header, unsigned section_mask, uint8_t *d_stream,
                                         size_t capacity, void *hip_stream) {
  if ((section_mask & ~0x3fu) != 0u) return SPZ_AMD_ERR_INVALID_ARG;
  return encode_impl(d_cloud, first, count, num_points_total, sh_degree, antialiased, from_coord, version,
                     write_header, d_stream, capacity, hip_stream, section_mask);
}

int spz_amd_decode_device(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr, int to_coord,
                          const spz_amd_cloud_out *d_cloud, void *hip_stream) {
  if (hdr == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  return decode_impl(d_stream, size, hdr, 0, hdr->num_points, to_coord, d_cloud, hip_stream);
}

int spz_amd_decode_shard_device(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr, uint64_t first,
                                uint64_t count, int to_coord, const spz_amd_cloud_out *d_cloud,
                                void *hip_stream) {
  return decode_impl(d_stream, size, hdr, first, count, to_coord, d_cloud, hip_stream);
}

int spz_amd_decode_gather_device(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr,
                                 const uint32_t *d_indices, uint64_t count, int to_coord,
                                 const spz_amd_cloud_out *cl, void *hip_stream) {
  if (d_stream == nullptr || cl == nullptr || !valid_coord(to_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  int rc = check_header_fields(hdr);
  if (rc != SPZ_AMD_OK) return rc;
  spz_amd_layout lay;
  rc = layout_impl(hdr->num_points, hdr->sh_degree, (int)hdr->version, &lay);
  if (rc != SPZ_AMD_OK) return rc;
  if (size < lay.total_bytes) return SPZ_AMD_ERR_SHORT_STREAM;
  if (count == 0) return SPZ_AMD_OK;
  const int sd = sh_dim_for_degree(hdr->sh_degree);
  if (hdr->num_points == 0 || d_indices == nullptr || !cl->positions || !cl->scales || !cl->rotations ||
      !cl->alphas || !cl->colors || (sd > 0 && !cl->sh)) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  int device = 0;
  rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  GatherParams p = {};
  rc = ensure_tables(device, &p.tables);
  if (rc != SPZ_AMD_OK) return rc;
  p.positions = d_stream + lay.offset[SPZ_AMD_SEC_POSITIONS];
  p.alphas = d_stream + lay.offset[SPZ_AMD_SEC_ALPHAS];
  p.colors = d_stream + lay.offset[SPZ_AMD_SEC_COLORS];
  p.scales = d_stream + lay.offset[SPZ_AMD_SEC_SCALES];
  p.rotations = d_stream + lay.offset[SPZ_AMD_SEC_ROTATIONS];
  p.sh = d_stream + lay.offset[SPZ_AMD_SEC_SH];
  p.out_positions = cl->positions;
  p.out_scales = cl->scales;
  p.out_rotations = cl->rotations;
  p.out_alphas = cl->alphas;
  p.out_colors = cl->colors;
  p.out_sh = cl->sh;
  p.indices = d_indices;
  p.count = count;
  p.num_points = hdr->num_points;
  p.version = hdr->version;
  p.sh_dim = (uint32_t)sd;
  const FlipMasks fm = flip_masks(SPZ_AMD_RUB, to_coord);
  p.flip_p = fm.p;
  p.flip_q = fm.q;
  p.flip_sh15 = fm.sh15;
  p.sh_elem_mask = sh_elem_mask_ext(fm.sh15, sd);
  p.pos_scale = (float)(1.0 / (double)(int32_t)(1u << (hdr->fractional_bits & 31)));
  const unsigned long long items = count * (unsigned long long)(sd > 0 ? sd * 3 : 3);
  unsigned long long blocks = (items + 255) / 256;
  if (blocks > 65536ull * 16) blocks = 65536ull * 16;  // grid-stride beyond that
  hipLaunchKernelGGL(spz_decode_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), p);
  SPZ_HIP_TRY(hipGetLastError());
  return SPZ_AMD_OK;
}

int spz_amd_convert_coordinates_device(float *d_positions, float *d_rotations, float *d_sh, uint64_t num_points,
                                       int sh_degree, int from_coord, int to_coord, void *hip_stream) {
  const int sd = sh_dim_for_degree(sh_degree);
  if (sd < 0 || !valid_coord(from_coord) || !valid_coord(to_coord)) return SPZ_AMD_ERR_INVALID_ARG;
  if (num_points == 0) return SPZ_AMD_OK;  // splat-types.h:135-138
  int device = 0;
  int rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  KParams p = {};
  const FlipMasks fm = flip_masks(from_coord, to_coord);
  p.flip_p = fm.p;
  p.flip_q = fm.q;
  p.sh_mask_ext = sh_elem_mask_ext(fm.sh15, sd);
  p.sh_d = (uint32_t)sd * 3u;
  if (d_sh) add_section(&p, FlipGeom::kTileUnits, KIND_FLIP_SH, nullptr, d_sh, num_points * (uint64_t)sd * 3u);
  if (d_positions) add_section(&p, FlipGeom::kTileUnits, KIND_FLIP_POS, nullptr, d_positions, num_points * 3u);
  if (d_rotations) add_section(&p, FlipGeom::kTileUnits, KIND_FLIP_ROT, nullptr, d_rotations, num_points * 4u);
  if (p.total_tiles == 0) return SPZ_AMD_OK;
  uint32_t grid = 1;
  rc = grid_for(device, p.total_tiles, &grid);
  if (rc != SPZ_AMD_OK) return rc;
  hipLaunchKernelGGL(spz_flip_kernel, dim3(grid), dim3(FlipGeom::kBlock), 0, static_cast<hipStream_t>(hip_stream), p);
  SPZ_HIP_TRY(hipGetLastError());
  return SPZ_AMD_OK;
}


int spz_amd_selftest_device(int mode, uint64_t begin, uint64_t count, uint64_t result[3], void *hip_stream) {
  if (result == nullptr || mode < 0 || mode > (int)SELFTEST_UNPACK_S3) return SPZ_AMD_ERR_INVALID_ARG;
  int device = 0;
  int rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  unsigned long long *d = nullptr;
  SPZ_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), 3 * sizeof(unsigned long long)));
  const unsigned long long init[3] = {0ull, ~0ull, 0ull};
  hipError_t e = hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, st);
  if (e == hipSuccess && count > 0) {
    SelfTestParams p = {(uint32_t)mode, begin, count, d};
    unsigned long long blocks = (count + 255) / 256;
    if (blocks > 256ull * 64) blocks = 256ull * 64;  // grid-stride beyond that
    hipLaunchKernelGGL(spz_selftest_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
    e = hipGetLastError();
  }
  unsigned long long out[3] = {0, 0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, sizeof(out), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return SPZ_AMD_ERR_HIP;
  }
  for (int i = 0; i < 3; ++i) result[i] = out[i];
  return SPZ_AMD_OK;
}

int spz_amd_get_tables(float alpha_decode[256], float color_decode[256], float alpha_thresholds[255]) {
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  if (!g_tables.host_ready) build_host_tables();
  if (alpha_decode) std::memcpy(alpha_decode, g_tables.host + kTableAlphaDec, 256 * sizeof(float));
  if (color_decode) std::memcpy(color_decode, g_tables.host + kTableColorDec, 256 * sizeof(float));
  if (alpha_thresholds) std::memcpy(alpha_thresholds, g_tables.host + kTableAlphaThr, 255 * sizeof(float));
  return SPZ_AMD_OK;
}

}  // extern "C"
