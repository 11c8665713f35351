/*
 * spz_oracle.h — CPU restatement of the reference SPZ pack/unpack hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / the timed CPU baseline.
 * The product path (spz_amd/csrc) never links, loads or calls it.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against
 * the reference's own C++ compiled from /root/reference/src/cc by
 * oracle/Makefile (oracle/_ref/libspz_ref.so) and against the committed
 * golden vectors in tests/golden/ that were produced by that build
 * (tests/golden/make_golden.py).  The one exception is spzo_pack's version==2
 * rotation encoder: the reference tree has no v2 encoder (load-spz.cc:272
 * always writes smallest-three), so that branch is "parity unpinned" and is
 * checked only by round-trip through the reference's v2 decoder.
 */
#ifndef SPZ_ORACLE_H_
#define SPZ_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* splat-types.h:55-59 */
typedef struct {
  float flipP[3];
  float flipQ[3];
  float flipSh[15];
} spzo_converter_t;

/* splat-types.h:43-81 */
void spzo_coordinate_converter(int from, int to, spzo_converter_t *c);

/* load-spz.cc:58-72 */
int spzo_dim_for_degree(int degree);

/* Bytes of the serialized (pre-gzip) stream: 16-byte header + 6 sections
 * (load-spz.cc:533-546).  version 1: 6-byte positions, 3-byte rotations;
 * version 2: 9-byte positions, 3-byte rotations; version 3: 9 / 4. */
size_t spzo_stream_size(int64_t num_points, int sh_degree, int version);

/* packGaussians (load-spz.cc:257-331) + serializePackedGaussians (:533-546).
 * Writes header + sections into `stream` (capacity >= spzo_stream_size).
 * version must be 3 (reference behaviour) or 2 (parity unpinned, see above).
 * Returns the number of bytes written, 0 on invalid arguments. */
size_t spzo_pack(const float *positions, const float *scales, const float *rotations,
                 const float *alphas, const float *colors, const float *sh,
                 int32_t num_points, int sh_degree, int antialiased, int from_coord,
                 int version, uint8_t *stream);

/* deserializePackedGaussians (load-spz.cc:548-596) header checks.
 * Returns 0 on success, negative on the reference's rejection conditions:
 * -1 header not found, -2 version, -3 too many points, -4 sh degree, -5 short read. */
int spzo_peek(const uint8_t *stream, size_t size, uint32_t *version, uint32_t *num_points,
              int *sh_degree, int *fractional_bits, int *antialiased);

/* deserializePackedGaussians + unpackGaussians (load-spz.cc:467-531) including
 * the trailing convertCoordinates(RUB, to) pass (splat-types.h:134-164).
 * Output arrays must hold 3N,3N,4N,N,3N,N*shDim*3 floats.  Returns 0 / negative
 * as spzo_peek. */
int spzo_unpack(const uint8_t *stream, size_t size, int to_coord, float *positions,
                float *scales, float *rotations, float *alphas, float *colors, float *sh);

/* GaussianCloud::convertCoordinates (splat-types.h:134-164), in place. */
void spzo_convert_coordinates(float *positions, float *rotations, float *sh,
                              int32_t num_points, int sh_dim, int from_coord, int to_coord);

/* Scalar helpers exposed for table / known-answer tests. */
uint8_t spzo_to_uint8(float x);                 /* load-spz.cc:74 */
uint8_t spzo_quantize_sh(float x, int bucket);  /* load-spz.cc:77-81 */
float spzo_unquantize_sh(uint8_t x);            /* load-spz.cc:83 */
float spzo_sigmoid(float x);                    /* load-spz.cc:85 */
float spzo_inv_sigmoid(float x);                /* load-spz.cc:87 */
uint8_t spzo_alpha_byte(float a);               /* load-spz.cc:301 */
float spzo_alpha_value(uint8_t b);              /* load-spz.cc:518 */
float spzo_color_value(uint8_t b);              /* load-spz.cc:522 */
float spzo_half_to_float(uint16_t h);           /* splat-types.cc:8-27 */
/* GaussianCloud::medianVolume, splat-types.h:170-185: 0.01 for n == 0, else 4/3*pi*expf(middle scale sum). */
float spzo_median_volume(const float *scales, int32_t n);
void spzo_pack_quat_smallest_three(uint8_t r[4], const float q[4],
                                   const spzo_converter_t *c); /* load-spz.cc:216-255 */
void spzo_unpack_quat_smallest_three(float q[4], const uint8_t r[4],
                                     const spzo_converter_t *c); /* load-spz.cc:347-381 */
void spzo_unpack_quat_first_three(float q[4], const uint8_t r[3],
                                  const spzo_converter_t *c); /* load-spz.cc:333-345 */

/* ---- .ply rows <-> cloud -------------------------------------------------- */
/* Column map of a vertex row; same field order as spz_amd_ply_columns. */
typedef struct {
  int32_t stride, sh_dim;
  int32_t position[3], scale[3], rotation[4], alpha, color[3], sh[45];
} spzo_ply_columns_t;

/* The value-shuffling part of loadSplatFromPly (load-spz.cc:814-839) followed by
 * convertCoordinates(RDF, to) (:842). */
void spzo_ply_rows_to_cloud(const float *rows, int32_t num_points, const spzo_ply_columns_t *cols, int to_coord,
                            float *positions, float *scales, float *rotations, float *alphas, float *colors,
                            float *sh);
/* The row assembly of saveSplatToPly (load-spz.cc:846-893); rows has num_points*(17+3*sh_dim) floats. */
void spzo_cloud_to_ply_rows(const float *positions, const float *scales, const float *rotations,
                            const float *alphas, const float *colors, const float *sh, int32_t num_points,
                            int sh_dim, int from_coord, float *rows);

#ifdef __cplusplus
}
#endif
#endif  /* SPZ_ORACLE_H_ */
