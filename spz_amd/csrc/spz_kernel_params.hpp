// spz_kernel_params.hpp — what the host side (spz_abi.hip) and the kernels (spz_kernels.hip) share:
// launch geometry, the section table passed by value to the fused kernels, table offsets, and the
// kernel prototypes.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "spz_amd.h"

namespace spz_amd_detail {

// ------------------------------------------------------------------------------------------
// Geometry
// ------------------------------------------------------------------------------------------
// Launch geometry, per kernel.  A tile is BLOCK threads x UNROLL units.  The defaults are the
// measured best on MI355X (profiles/README.md); the macros exist so that tools/tune.py can build
// variants.  WC: a wave owns one contiguous span of its tile instead of UNROLL strided 1 KiB spans.
// NTL / NTS: non-temporal loads / stores (every byte on this path is touched exactly once).  What a
// kernel leaves in the write-back caches is paid for by the NEXT kernel: with plain stores in decode
// the following encode ran 0.537 ms, with non-temporal ones 0.484 ms (profiles/r01_tune_h_*.jsonl),
// so pairs are tuned in their own steady state, not kernel by kernel.
#ifndef SPZ_DEC_BLOCK
#define SPZ_DEC_BLOCK 256
#endif
#ifndef SPZ_DEC_UNROLL
#define SPZ_DEC_UNROLL 4
#endif
#ifndef SPZ_DEC_WC
#define SPZ_DEC_WC 0
#endif
#ifndef SPZ_DEC_NTL
#define SPZ_DEC_NTL 1
#endif
#ifndef SPZ_DEC_NTS
#define SPZ_DEC_NTS 1
#endif
#ifndef SPZ_ENC_BLOCK
#define SPZ_ENC_BLOCK 256
#endif
#ifndef SPZ_ENC_UNROLL
#define SPZ_ENC_UNROLL 4
#endif
#ifndef SPZ_ENC_WC
#define SPZ_ENC_WC 1
#endif
#ifndef SPZ_ENC_NTL
#define SPZ_ENC_NTL 1
#endif
#ifndef SPZ_ENC_NTS
#define SPZ_ENC_NTS 0
#endif
// 0: one tile per block ("flat" grid, measured faster than a persistent grid-stride loop: a wave's
// next loads would queue behind its own stores in the in-order vmcnt).  k > 0: at most k blocks per CU.
// Second __launch_bounds__ argument (minimum waves per SIMD) for the encode kernel.  6: the allocator then fits
// the kernel into 78 VGPRs without scratch (left alone it takes 89-90 = 5 waves per SIMD): SH0 encode +3 %, the
// other workloads unchanged (profiles/r02_tune_o_*).  8 spills (round 1).
#ifndef SPZ_ENC_MIN_WAVES
#define SPZ_ENC_MIN_WAVES 6
#endif
// 1: a scheduling barrier after every unit of an encode tile, so the compiler does not interleave the
// arithmetic of the four units (fewer live registers, more waves per SIMD).
#ifndef SPZ_ENC_SCHED_BARRIER
#define SPZ_ENC_SCHED_BARRIER 1
#endif
// 1: XCD-contiguous tile assignment (flat grids only), see first_tile() in spz_kernels.hip.
// Decode walks the tiles from the last to the first: the reverse of the order in which an encode launch
// wrote the stream, so a decode that follows an encode meets the most recently written bytes first.
#ifndef SPZ_DEC_REVERSE
#define SPZ_DEC_REVERSE 0
#endif
// With SPZ_DEC_REVERSE: the part of the stream that was written last (this many MiB, counted from its end in
// encode's writing order) is read with ordinary loads, which hit what is still in the caches; the rest
// with the non-temporal loads of DecGeom.  0 = non-temporal everywhere.
#ifndef SPZ_DEC_HOT_MIB
#define SPZ_DEC_HOT_MIB 0
#endif
#ifndef SPZ_XCD_REMAP
#define SPZ_XCD_REMAP 0
#endif
// 1: quaternion divisions without the IEEE expansion's operand scaling inside the exponent window where
// that is exact (spz_kernels.hip, "Correctly rounded divisions ..."); 0: the plain `/` everywhere.
// Interleaved grid: the tiles of the sections are dealt out in a repeating pattern (each section gets slots in
// proportion to its tile count) instead of section after section, so that at any moment the resident
// blocks are the launch's average mix of arithmetic-heavy (rotations) and traffic-heavy (sh) tiles.
// 0: never, 1: always, 2: by the measured policy (profiles/r02_tune_*.jsonl, 10 M points, six boxes): encode
// gains without sh (+4...6 %: the rotation tiles no longer run as one arithmetic-bound phase) and loses with sh
// (-3...5 %); decode gains with sh (+8...16 %; degree 3: see SPZ_IL_GROUP) and loses without (up to -13 %: five
// equal streams at once).
#ifndef SPZ_ENC_INTERLEAVE
#define SPZ_ENC_INTERLEAVE 2
#endif
#ifndef SPZ_DEC_INTERLEAVE
#define SPZ_DEC_INTERLEAVE 2
#endif
// The interleaved grid deals RUNS of SPZ_IL_GROUP consecutive tiles (a power of two) per pattern slot.  8 = one tile
// per XCD (the dispatcher deals consecutive blocks round-robin over the eight XCDs), so inside a run the eight
// XCDs write one contiguous 128 KiB window exactly as in the sequential order.  With single tiles (1) the gain of
// the sh3 decode depended on the box (+7 %, +9 %, +12 % on three, -5 %, -6 %, -7 % on three others); with runs
// of 8 it was +12 % on a box of the first kind and -0.5 % on one of the second (profiles/r02_tune_f_*, r02_tune_g_*).
// SPZ_IL_ONLY_ROT = 1 interleaves only the rotation tiles into the largest section, the rest stays sequential.
#ifndef SPZ_IL_GROUP
#define SPZ_IL_GROUP 8
#endif
#ifndef SPZ_IL_ONLY_ROT
#define SPZ_IL_ONLY_ROT 0
#endif
#ifndef SPZ_QUAT_FAST
#define SPZ_QUAT_FAST 1
#endif
#ifndef SPZ_BLOCKS_PER_CU
#define SPZ_BLOCKS_PER_CU 0
#endif

template <int BLOCK_, int UNROLL_, bool WC_, bool NTL_, bool NTS_>
struct Geom {
  static constexpr int kBlock = BLOCK_;
  static constexpr int kUnroll = UNROLL_;
  static constexpr int kTileUnits = BLOCK_ * UNROLL_;  // one unit = 4 float elements
  static constexpr bool kWaveContig = WC_;
  static constexpr bool kNtLoad = NTL_;
  static constexpr bool kNtStore = NTS_;
  // Unit handled by thread `tid` in round `r` of a tile.
  __device__ static __forceinline__ uint32_t local_unit(int r, uint32_t tid) {
    if constexpr (WC_) return ((tid >> 6) * (uint32_t)UNROLL_ + (uint32_t)r) * 64u + (tid & 63u);
    else return (uint32_t)(r * BLOCK_) + tid;
  }
};
using DecGeom = Geom<SPZ_DEC_BLOCK, SPZ_DEC_UNROLL, SPZ_DEC_WC != 0, SPZ_DEC_NTL != 0, SPZ_DEC_NTS != 0>;
using DecGeomHot = Geom<SPZ_DEC_BLOCK, SPZ_DEC_UNROLL, SPZ_DEC_WC != 0, false, SPZ_DEC_NTS != 0>;  // ordinary loads
using EncGeom = Geom<SPZ_ENC_BLOCK, SPZ_ENC_UNROLL, SPZ_ENC_WC != 0, SPZ_ENC_NTL != 0, SPZ_ENC_NTS != 0>;
using FlipGeom = Geom<256, 4, false, false, false>;
constexpr int kMaxBlocksPerCU = SPZ_BLOCKS_PER_CU;

enum SecKind : uint32_t {
  KIND_POS24 = 0,  // 24-bit fixed point, 12 bytes per unit
  KIND_POS16,      // legacy float16 positions (decode only), 8 bytes per unit
  KIND_ALPHA,      // 4 bytes per unit, table / threshold search
  KIND_COLOR,      // 4 bytes per unit
  KIND_SCALE,      // 4 bytes per unit
  KIND_ROT_S3,     // smallest-three quaternion: 1 point = 4 floats <-> 4 bytes
  KIND_ROT_F3,     // first-three quaternion:    1 point = 4 floats <-> 3 bytes
  KIND_SH,         // 4 bytes per unit, flip + bucket depend on (element % D)
  KIND_FLIP_POS,   // in-place convertCoordinates passes (floats only)
  KIND_FLIP_ROT,
  KIND_FLIP_SH,
};

struct SecDesc {
  uint8_t *bytes;              // stream side (encode writes, decode reads); unused by flip kinds
  float *floats;               // float side
  unsigned long long n_elems;  // float elements in this section
  unsigned long long n_units;  // ceil(n_elems / 4)
  uint32_t tile_begin;         // first tile of this section in the sequential part of the fused grid
  uint32_t tile_skip;          // tiles of this section that the interleaved part of the grid covers
  uint32_t kind;
};

// Interleaved part of the grid: tiles [0, il_tiles) repeat a pattern of kIlPeriod slots; slot k belongs to
// section il_sec[k] and is that section's il_rank[k]-th slot of the period.  The period is odd, so a
// section's slots move over all eight XCDs (blocks b and b + 8 share one).
constexpr uint32_t kIlPeriod = 61;

struct KParams {
  SecDesc sec[SPZ_AMD_NUM_SECTIONS];
  uint32_t n_sec;
  uint32_t total_tiles;
  uint32_t flip_p;                // bit a set: axis a is negated
  uint32_t flip_q;                // bit i set: quaternion component i (x,y,z) is negated
  unsigned long long sh_mask_ext; // bit j: element j of a point's D sh floats is negated; bits D..D+2 repeat 0..2
  uint32_t sh_d;                  // 0, 9, 24, 45
  float pos_scale;                // decode: 1 / (1 << fractionalBits)
  const float *tables;            // device tables, see kTable* below
  uint8_t *header_dst;            // encode: where the 16 header bytes go (nullptr: none)
  uint32_t header_words[4];
  uint32_t plain_tiles;           // decode: this many tiles, counted from the last one, use ordinary loads
  uint32_t il_tiles;              // 0: no interleaved part
  uint8_t il_count[8];            // slots per period of section k
  uint8_t il_sec[64];
  uint8_t il_rank[64];
};

constexpr int kTableAlphaDec = 0;    // 256 floats
constexpr int kTableColorDec = 256;  // 256 floats
constexpr int kTableAlphaThr = 512;  // 255 floats + 1 NaN pad
constexpr int kTableFloats = 768;

// Random-access decode: the points named by an index list, out of a stream that
// stays packed in HBM.  Sections / outputs in the order of the spz_amd_cloud_out fields.
struct GatherParams {
  const uint8_t *positions, *alphas, *colors, *scales, *rotations, *sh;  // section bases in the stream
  float *out_positions, *out_scales, *out_rotations, *out_alphas, *out_colors, *out_sh;
  const uint32_t *indices;
  unsigned long long count;       // number of indices
  uint32_t num_points;            // points in the stream (indices are clamped to num_points - 1)
  uint32_t version;               // 1, 2, 3
  uint32_t sh_dim;                // 0, 3, 8, 15
  uint32_t flip_p, flip_q, flip_sh15;
  unsigned long long sh_elem_mask;  // bit j: element j of a point's D sh floats is negated
  float pos_scale;
  const float *tables;
};

// Proof obligations of the fast quaternion arithmetic, checked on the device itself (spz_selftest_kernel):
// every mode compares the fast form with the plain IEEE form it replaces over a range of inputs and
// counts the inputs on which the bits differ.
enum SelfTestMode : uint32_t {
  SELFTEST_DIV_SQRT1_2 = 0,  // x / 0.70710677f, x = every float bit pattern in [begin, begin + count) inside the window
  SELFTEST_DIV_511 = 1,      // x / 511.0f, likewise
  SELFTEST_SQRT = 2,         // sqrt_cr(x) vs sqrtf, x in [2^-80, 2^82]
  SELFTEST_QUOTIENT = 3,     // quat_quotient(a, b) vs a / b on hashed operand pairs inside the window
  SELFTEST_PACK_S3 = 4,      // whole smallest-three encoder, fast vs general, hashed quaternions inside the window
  SELFTEST_PACK_F3 = 5,      // whole first-three encoder, likewise
  SELFTEST_UNPACK_S3 = 6,    // v3 decoder with div_by_const vs with `/ 511.0f`, comp = every 32-bit word in range
};
struct SelfTestParams {
  uint32_t mode;
  unsigned long long begin, count;
  unsigned long long *mismatches;  // [0] = count, [1] = smallest mismatching index (init ~0), [2] = inputs compared
};

// The kernels (spz_kernels.hip).
__global__ void spz_selftest_kernel(const SelfTestParams p);
__global__ void spz_decode_gather_kernel(const GatherParams p);
__global__ void spz_decode_kernel(const KParams p);
__global__ void spz_encode_kernel(const KParams p);
__global__ void spz_flip_kernel(const KParams p);

}  // namespace spz_amd_detail
