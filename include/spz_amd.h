/*
 * spz_amd.h — C ABI of the MI355X-native SPZ pack/unpack path (libspz_amd.so).
 *
 * This is the drop-in boundary for the hot path of lanxinger/spz: everything a
 * binding of the reference's packGaussians / unpackGaussians / (de)serialize
 * step would need, as plain C: pointers, sizes, ints.  No C++ types, no torch
 * types, no exceptions, no caller-visible allocation.  Citations are to
 * /root/reference/src/cc (the interface each entry point replaces).
 *
 * Byte stream ("raw stream", pre-gzip), little-endian, load-spz.cc:131-139,533-546:
 *   16-byte header | positions | alphas | colors | scales | rotations | sh
 * gzip stays on the host (libspz_host.so, include/spz_amd_host.hpp).
 *
 * Float side: the six flat float32 arrays of GaussianCloud (splat-types.h:90-115):
 *   positions[3N] xyz, scales[3N], rotations[4N] xyzw, alphas[N], colors[3N],
 *   sh[N*shDim*3] laid out [point][coeff][rgb], shDim = 0,3,8,15.
 *
 * Memory spaces: *_device entry points take DEVICE pointers valid on the current
 * HIP device and enqueue on `hip_stream` (a hipStream_t passed as void*, NULL =
 * default stream) without synchronising.  *_host entry points take HOST pointers,
 * stage through device memory on `device`, and return when the result is in
 * host memory.  Every compute entry point fails with SPZ_AMD_ERR_NO_DEVICE when
 * no HIP device is usable: there is no CPU fallback in this library.
 *
 * Alignment: float arrays need 4-byte alignment; stream pointers need none.
 */
#ifndef SPZ_AMD_H_
#define SPZ_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPZ_AMD_ABI_VERSION 1

/* Status codes.  The header-validation codes mirror, one for one, the rejection
 * branches of deserializePackedGaussians (load-spz.cc:553-568,591-594). */
enum {
  SPZ_AMD_OK = 0,
  SPZ_AMD_ERR_INVALID_ARG = -1,     /* NULL pointer, bad degree/version/coord, checkSizes load-spz.cc:106-127 */
  SPZ_AMD_ERR_HEADER_NOT_FOUND = -2, /* short header or wrong magic, load-spz.cc:553-556 */
  SPZ_AMD_ERR_VERSION = -3,         /* version outside [1,3], load-spz.cc:557-560 */
  SPZ_AMD_ERR_TOO_MANY_POINTS = -4, /* numPoints > limit, load-spz.cc:561-564 */
  SPZ_AMD_ERR_SH_DEGREE = -5,       /* shDegree > 3, load-spz.cc:565-568 */
  SPZ_AMD_ERR_SHORT_STREAM = -6,    /* "read error", load-spz.cc:591-594 */
  SPZ_AMD_ERR_CAPACITY = -7,        /* output buffer too small */
  SPZ_AMD_ERR_NO_DEVICE = -8,       /* no usable HIP device / runtime */
  SPZ_AMD_ERR_HIP = -9,             /* a HIP call failed (see spz_amd_last_hip_error) */
  SPZ_AMD_ERR_UNSUPPORTED = -10,    /* e.g. encode of version 1; RCCL not loadable */
  SPZ_AMD_ERR_COMM = -11,           /* an RCCL call failed (see spz_amd_last_rccl_error) */
  SPZ_AMD_ERR_VERIFY = -12          /* a result failed the library's own check (container stage: symbols that do not
                                       reproduce their input, a member that does not inflate back to it) */
};

/* CoordinateSystem values, splat-types.h:24-34. */
enum {
  SPZ_AMD_UNSPECIFIED = 0, SPZ_AMD_LDB = 1, SPZ_AMD_RDB = 2, SPZ_AMD_LUB = 3, SPZ_AMD_RUB = 4,
  SPZ_AMD_LDF = 5, SPZ_AMD_RDF = 6, SPZ_AMD_LUF = 7, SPZ_AMD_RUF = 8
};

/* Reader limit of the reference, load-spz.cc:549. */
#define SPZ_AMD_REFERENCE_MAX_POINTS 10000000u

/* PackedGaussiansHeader minus the magic, load-spz.cc:131-139. */
typedef struct {
  uint32_t version;        /* 1, 2 or 3 */
  uint32_t num_points;
  uint8_t sh_degree;       /* 0..3 */
  uint8_t fractional_bits; /* writer: 12, load-spz.cc:270 */
  uint8_t flags;           /* bit0 = antialiased, load-spz.cc:129 */
  uint8_t reserved;
} spz_amd_header;

/* Section order of the stream, load-spz.cc:540-545. */
enum { SPZ_AMD_SEC_POSITIONS = 0, SPZ_AMD_SEC_ALPHAS, SPZ_AMD_SEC_COLORS, SPZ_AMD_SEC_SCALES,
       SPZ_AMD_SEC_ROTATIONS, SPZ_AMD_SEC_SH, SPZ_AMD_NUM_SECTIONS };

typedef struct {
  uint64_t total_bytes;                      /* 16 + sum(bytes) */
  uint64_t offset[SPZ_AMD_NUM_SECTIONS];     /* byte offset of each section from stream start */
  uint64_t bytes[SPZ_AMD_NUM_SECTIONS];      /* byte length of each section */
  uint32_t bytes_per_point[SPZ_AMD_NUM_SECTIONS];
} spz_amd_layout;

/* GaussianCloud's arrays (splat-types.h:101-115) as raw pointers. sh may be NULL iff sh_degree==0. */
typedef struct {
  const float *positions, *scales, *rotations, *alphas, *colors, *sh;
} spz_amd_cloud_in;
typedef struct {
  float *positions, *scales, *rotations, *alphas, *colors, *sh;
} spz_amd_cloud_out;

/* ---- introspection ---------------------------------------------------------------------- */
int spz_amd_abi_version(void);
const char *spz_amd_status_string(int status);
/* Number of usable HIP devices (0 when there is none or the runtime fails to initialise). */
int spz_amd_device_count(void);
/* hipError_t value of the last failing HIP call on this thread (0 if none). */
int spz_amd_last_hip_error(void);

/* Frees what the library keeps on the devices between calls (decode tables / thresholds, the cached
 * workspace of the *_host entry points, up to three large blocks of the gzip container stage: ~25 bytes per byte of
 * the largest stream compressed so far).  Optional: everything is re-created on demand.  Must not
 * run concurrently with other calls into the library. */
int spz_amd_release_device_memory(void);

/* ---- stream geometry: replaces the size arithmetic spread over packGaussians
 *      (load-spz.cc:273-278), serializePackedGaussians (:540-545) and
 *      deserializePackedGaussians (:578-590).  Pure host function. ------------------------ */
int spz_amd_stream_layout(uint64_t num_points, int sh_degree, int version, spz_amd_layout *out);

/* ---- header: PackedGaussiansHeader write (load-spz.cc:534-539) / checks (:551-568).
 *      Pure host functions on HOST memory.  spz_amd_peek_header applies the reference's
 *      10 M point limit; the _ex form takes the limit (0 = none) for shards that are
 *      reassembled into a stream larger than the reference itself can read. --------------- */
int spz_amd_write_header(const spz_amd_header *hdr, uint8_t out16[16]);
int spz_amd_peek_header(const uint8_t *stream, size_t size, spz_amd_header *out);
int spz_amd_peek_header_ex(const uint8_t *stream, size_t size, uint64_t max_points, spz_amd_header *out);
/* Same checks for a stream that lives in DEVICE memory (e.g. fragments reassembled over RCCL): copies
 * the 16 header bytes to the host on `hip_stream` and waits for that copy. */
int spz_amd_peek_header_device(const uint8_t *d_stream, size_t size, uint64_t max_points, spz_amd_header *out,
                               void *hip_stream);

/* ---- encode: packGaussians (load-spz.cc:257-331) + serializePackedGaussians (:533-546)
 *      fused: float SoA -> header + six sections written in place.
 *      version: 3 (what the reference writes, :272) or 2 (first-three quaternions; the
 *      reference has no v2 encoder, parity unpinned).  from_coord: PackOptions::from. -------- */
int spz_amd_encode_device(const spz_amd_cloud_in *d_cloud, uint64_t num_points, int sh_degree,
                          int antialiased, int from_coord, int version, uint8_t *d_stream,
                          size_t capacity, void *hip_stream);

/* ---- decode: the body of deserializePackedGaussians (section slicing, :569-590) +
 *      unpackGaussians (:467-531) with the trailing convertCoordinates(RUB, to)
 *      pass (:529, splat-types.h:134-164) fused into the same kernel.
 *      `hdr` is the header the caller obtained from spz_amd_peek_header (the first 16
 *      bytes of d_stream are not re-read).  to_coord: UnpackOptions::to. -------------------- */
int spz_amd_decode_device(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr,
                          int to_coord, const spz_amd_cloud_out *d_cloud, void *hip_stream);

/* ---- point-range shards (multi-GPU / multi-stream).  The stream is attribute-major, so a
 *      shard [first, first+count) of a num_points-total stream is six fragments at
 *      offset[s] + first*bytes_per_point[s].  d_cloud holds only the shard's points.
 *      encode_shard writes the fragments (and the header iff write_header != 0) into the
 *      FULL stream buffer d_stream; decode_shard reads them from it. ------------------------ */
int spz_amd_encode_shard_device(const spz_amd_cloud_in *d_cloud, uint64_t first, uint64_t count,
                                uint64_t num_points_total, int sh_degree, int antialiased,
                                int from_coord, int version, int write_header, uint8_t *d_stream,
                                size_t capacity, void *hip_stream);
int spz_amd_decode_shard_device(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr,
                                uint64_t first, uint64_t count, int to_coord,
                                const spz_amd_cloud_out *d_cloud, void *hip_stream);

/* Same as spz_amd_encode_shard_device for a subset of the sections: bit s of section_mask selects section s
 * (SPZ_AMD_SEC_*).  Lets a multi-GPU caller encode the five small sections (20 B/point), hand their
 * fragments to the exchange, and encode the sh section (up to 45 B/point) while they travel. */
int spz_amd_encode_shard_sections_device(const spz_amd_cloud_in *d_cloud, uint64_t first, uint64_t count,
                                         uint64_t num_points_total, int sh_degree, int antialiased,
                                         int from_coord, int version, int write_header, unsigned section_mask,
                                         uint8_t *d_stream, size_t capacity, void *hip_stream);

/* ---- the exchange step of the multi-GPU path: the stream of R point-range shards lands on one
 *      device.  spz_amd_shard_fragments gives the six byte ranges of a shard: where they sit in the full
 *      stream (global_offset) and in a stream that holds the shard alone (local_offset).
 *
 *      RCCL route: spz_amd_gatherv_rccl issues ONE ncclGroupStart/End with the six ncclSend of every
 *      non-root rank matched by ncclRecv on the root straight at the final offsets (RCCL has no gatherv).
 *      comm is an ncclComm_t the caller owns (spz_amd_rccl_comm_init wraps ncclCommInitRank for callers that
 *      have none; the 128-byte id comes from spz_amd_rccl_unique_id on one rank and travels by any means).
 *      first[r], count[r]: the contiguous point ranges in rank order.  d_local_stream: this rank's own
 *      stream (header + its fragments); the root passes NULL when it encoded straight into
 *      d_global_stream (spz_amd_encode_shard_device).  Enqueued on hip_stream, no synchronisation.
 *      RCCL is looked up at run time (dlopen): SPZ_AMD_ERR_UNSUPPORTED when it cannot be found.
 *
 *      IPC route: the root allocates the stream with spz_amd_ipc_alloc and passes the 64-byte handle to its
 *      peers (same node); a peer maps it with spz_amd_ipc_open and hands the mapped pointer to
 *      spz_amd_encode_shard_device as d_stream: the encode kernel's stores write the fragments into the
 *      root's memory over xGMI, no second pass.  The root may read after the peer's stream has completed. -- */
typedef struct {
  uint64_t global_offset[SPZ_AMD_NUM_SECTIONS], local_offset[SPZ_AMD_NUM_SECTIONS], bytes[SPZ_AMD_NUM_SECTIONS];
} spz_amd_fragments;
int spz_amd_shard_fragments(uint64_t first, uint64_t count, uint64_t num_points_total, int sh_degree, int version,
                            spz_amd_fragments *out);
#define SPZ_AMD_RCCL_UNIQUE_ID_BYTES 128
int spz_amd_rccl_available(void);
int spz_amd_last_rccl_error(void); /* ncclResult_t of the last failing RCCL call on this thread */
int spz_amd_rccl_unique_id(uint8_t id[SPZ_AMD_RCCL_UNIQUE_ID_BYTES]);
int spz_amd_rccl_comm_init(const uint8_t id[SPZ_AMD_RCCL_UNIQUE_ID_BYTES], int world, int rank, void **comm);
int spz_amd_rccl_comm_destroy(void *comm);
int spz_amd_gatherv_rccl(void *comm, int rank, int world, int root, const uint64_t *first, const uint64_t *count,
                         int sh_degree, int version, const uint8_t *d_local_stream, uint8_t *d_global_stream,
                         unsigned section_mask, void *hip_stream);
/* The mirror image for the decode direction when the stream starts on the root only: every rank receives its six
 * fragments into a stream of its own (d_local_stream: layout of spz_amd_stream_layout(count[rank]); the header is the
 * caller's to write) and decodes them with spz_amd_decode_device.  Same group construction, 65 B/point for SH3. */
int spz_amd_scatterv_rccl(void *comm, int rank, int world, int root, const uint64_t *first, const uint64_t *count,
                          int sh_degree, int version, const uint8_t *d_global_stream, uint8_t *d_local_stream,
                          unsigned section_mask, void *hip_stream);
#define SPZ_AMD_IPC_HANDLE_BYTES 64
int spz_amd_ipc_alloc(size_t bytes, void **d_ptr, uint8_t handle[SPZ_AMD_IPC_HANDLE_BYTES]);
int spz_amd_ipc_free(void *d_ptr);
int spz_amd_ipc_open(const uint8_t handle[SPZ_AMD_IPC_HANDLE_BYTES], void **d_ptr);
int spz_amd_ipc_close(void *d_ptr);

/* ---- random access: decode only the points d_indices[0..count) out of a stream that
 *      stays packed in device memory — the bulk form of PackedGaussians::unpack(i, converter)
 *      (load-spz.cc:383-463), output in the GaussianCloud array layout (d_cloud holds `count` points).
 *      Indices >= hdr->num_points are clamped to the last point.  Same arithmetic as decode. ------- */
int spz_amd_decode_gather_device(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr,
                                 const uint32_t *d_indices, uint64_t count, int to_coord,
                                 const spz_amd_cloud_out *d_cloud, void *hip_stream);

/* ---- device buffers for a resident cloud, placed for speed (spz_place.hip).  Whether an sh3 decode runs
 *      at 0.46 ms or at 0.55 ms is decided by whether the sh float array shares a physical region of HBM with the other
 *      arrays of the launch; that cannot be asked for, but it shows in one launch.  alloc: the five small arrays (and a
 *      stream buffer, unless d_stream brings one) in one block, the sh array in an allocation of its own, chosen among
 *      up to max_candidates (1 = the first) by timing the launch the buffers are for — probe 1: a decode (d_stream
 *      read, cloud written), 2: an encode (cloud read, the stream OVERWRITTEN), 0: no timing — on `hip_stream`, with the
 *      buffers' contents zeroed.  Blocking; tens of milliseconds, once per set of long-lived buffers.  The reference has
 *      no counterpart (it holds clouds in host vectors); callers of the *_device entry points own their buffers and
 *      may use this to make them. ------------------------------------------------------------------------------- */
typedef struct {
  spz_amd_cloud_out cloud;      /* six float arrays in device memory (sh NULL for degree 0) */
  uint8_t *stream;              /* the stream buffer: this call's (stream_capacity bytes) or the caller's d_stream */
  size_t stream_capacity;
  void *owner;                  /* what spz_amd_cloud_buffers_free releases */
  int32_t candidates;           /* sh placements timed (1 when nothing was timed) */
  float probe_ms_first, probe_ms_chosen, probe_ms_worst;
} spz_amd_cloud_buffers;
int spz_amd_cloud_buffers_alloc(uint64_t num_points, int sh_degree, int version, uint8_t *d_stream, int probe,
                                int max_candidates, void *hip_stream, spz_amd_cloud_buffers *out);
int spz_amd_cloud_buffers_free(spz_amd_cloud_buffers *b);

/* ---- GaussianCloud::convertCoordinates (splat-types.h:134-164) as a standalone in-place
 *      device pass (the reference-shaped, un-fused second pass; kept for API parity and
 *      for the fused-vs-unfused measurement).  Any of the three pointers may be NULL. ------- */
int spz_amd_convert_coordinates_device(float *d_positions, float *d_rotations, float *d_sh,
                                       uint64_t num_points, int sh_degree, int from_coord,
                                       int to_coord, void *hip_stream);

/* ---- host-pointer entry points: H2D, kernel, D2H on `device`; blocking.  Calls of more than 160 MiB of
 *      floats run as a pipeline of point-range chunks (upload of chunk k+1, kernel on chunk k and
 *      download of chunk k-1 overlap; environment SPZ_AMD_HOST_CHUNK_MIB sets the chunk size).  The device
 *      staging memory is one grow-only allocation per device, kept until spz_amd_release_device_memory().
 *      spz_amd_decode_host applies the reference reader's 10 M point limit (load-spz.cc:549,561), which
 *      belongs to deserializePackedGaussians; the _ex form takes the limit (0 = none), for callers that
 *      mirror unpackGaussians (:467-531), which has none. --------------------------------------------- */
int spz_amd_encode_host(const spz_amd_cloud_in *h_cloud, uint64_t num_points, int sh_degree,
                        int antialiased, int from_coord, int version, uint8_t *h_stream,
                        size_t capacity, int device);
/* The same for a caller that goes on to the container stage (saveSpz, load-spz.cc:639-645: packGaussians then
 * compressGzipped): *d_stream = a device copy of the stream that stays valid until spz_amd_kept_stream_release — what
 * spz_amd_zlib_parse_open_dev takes instead of uploading the stream again — or NULL when the one such buffer per
 * device is in use (or num_points is 0). */
int spz_amd_encode_host_keep(const spz_amd_cloud_in *h_cloud, uint64_t num_points, int sh_degree,
                             int antialiased, int from_coord, int version, uint8_t *h_stream,
                             size_t capacity, int device, const uint8_t **d_stream);
void spz_amd_kept_stream_release(int device, const uint8_t *d_stream);
/* The same with the stream already in device memory (what spz_amd_inflate_device_data() returns): no upload, the
 * decoded floats come back to the host arrays through the same chunked pipeline.  `hdr`: the stream's header
 * (spz_amd_peek_header_device, or _ex on its first 16 bytes). */
int spz_amd_decode_host_from_device(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr, int to_coord,
                                    const spz_amd_cloud_out *h_cloud, int device);
int spz_amd_decode_host(const uint8_t *h_stream, size_t size, int to_coord,
                        const spz_amd_cloud_out *h_cloud, int device);
int spz_amd_decode_host_ex(const uint8_t *h_stream, size_t size, uint64_t max_points, int to_coord,
                           const spz_amd_cloud_out *h_cloud, int device);
int spz_amd_convert_coordinates_host(float *h_positions, float *h_rotations, float *h_sh,
                                     uint64_t num_points, int sh_degree, int from_coord,
                                     int to_coord, int device);
/* Host form of spz_amd_decode_gather_device: the packed stream and the index list are in host memory,
 * the `count` decoded points land in host arrays.  The header is read from the stream with the checks
 * of spz_amd_peek_header_ex(max_points).  Unlike the device form, which clamps, an index >= num_points
 * is rejected here with SPZ_AMD_ERR_INVALID_ARG before anything is copied. */
int spz_amd_decode_gather_host(const uint8_t *h_stream, size_t size, uint64_t max_points,
                               const uint32_t *h_indices, uint64_t count, int to_coord,
                               const spz_amd_cloud_out *h_cloud, int device);
/* The same for a stream that is in device memory already (spz_amd_inflate_device_data): only the index list goes
 * up and the `count` decoded points come down.  `hdr`: the stream's header. */
int spz_amd_decode_gather_host_from_device(const uint8_t *d_stream, size_t size, const spz_amd_header *hdr,
                                           const uint32_t *h_indices, uint64_t count, int to_coord,
                                           const spz_amd_cloud_out *h_cloud, int device);

/* ---- GaussianCloud::medianVolume's selection step (splat-types.h:170-185): the
 *      element of rank num_points/2 among the per-point sums (s0 + s1) + s2 of the log scales, found by
 *      radix selection instead of a sort (4 streaming passes over d_scales, 48 B read per point).
 *      d_workspace: SPZ_AMD_MEDIAN_WORKSPACE_BYTES of device memory owned by the caller for the
 *      duration of the call's work on hip_stream; d_median: one float in device memory.  The volume
 *      itself, 4/3*pi*exp(median), is one scalar the caller computes.  num_points must be >= 1. ---- */
#define SPZ_AMD_MEDIAN_WORKSPACE_BYTES 8192
int spz_amd_median_scale_sum_device(const float *d_scales, uint64_t num_points, void *d_workspace,
                                    float *d_median, void *hip_stream);
int spz_amd_median_scale_sum_host(const float *h_scales, uint64_t num_points, float *h_median, int device);

/* ---- .ply vertex rows <-> GaussianCloud arrays (the step on the far side of
 *      the hot path).  A binary-LE 3DGS .ply stores one row of `property float` columns per
 *      Gaussian (load-spz.cc:728-740); the column map below is what the header parse yields
 *      (:742-786).  rows_to_cloud replaces the AoS->SoA loop of loadSplatFromPly incl. the
 *      [channel][coeff] -> [coeff][channel] sh transpose (:814-839) and the trailing
 *      convertCoordinates(RDF, to) (:842); cloud_to_rows replaces the row assembly of
 *      saveSplatToPly with its from->RDF flips (:858-893).  sh_dim is the number of sh
 *      coefficients per channel actually present (0..15; 0,3,8,15 for degrees 0..3). ------------ */
typedef struct {
  int32_t stride;      /* floats per row (= number of properties), 14 + 3*sh_dim .. 255 */
  int32_t sh_dim;      /* 0..15 */
  int32_t position[3]; /* x, y, z */
  int32_t scale[3];    /* scale_0, scale_1, scale_2 */
  int32_t rotation[4]; /* rot_1, rot_2, rot_3, rot_0  (cloud order x y z w; the file is w x y z) */
  int32_t alpha;       /* opacity */
  int32_t color[3];    /* f_dc_0, f_dc_1, f_dc_2 */
  int32_t sh[45];      /* f_rest_i, i < 3*sh_dim, file order [channel][coeff] */
} spz_amd_ply_columns;

/* The layout saveSplatToPly writes (load-spz.cc:900-922): x y z nx ny nz f_dc_0..2 f_rest_* opacity
 * scale_0..2 rot_0..3; stride = 17 + 3*sh_dim. */
int spz_amd_ply_default_columns(int sh_dim, spz_amd_ply_columns *out);
int spz_amd_ply_rows_to_cloud_device(const float *d_rows, uint64_t num_points, const spz_amd_ply_columns *cols,
                                     int to_coord, const spz_amd_cloud_out *d_cloud, void *hip_stream);
int spz_amd_cloud_to_ply_rows_device(const spz_amd_cloud_in *d_cloud, uint64_t num_points, int sh_dim,
                                     int from_coord, float *d_rows, void *hip_stream);
int spz_amd_ply_rows_to_cloud_host(const float *h_rows, uint64_t num_points, const spz_amd_ply_columns *cols,
                                   int to_coord, const spz_amd_cloud_out *h_cloud, int device);
int spz_amd_cloud_to_ply_rows_host(const spz_amd_cloud_in *h_cloud, uint64_t num_points, int sh_dim,
                                   int from_coord, float *h_rows, int device);

/* ---- tables.  The alpha / colour decode tables (invSigmoid(b/255) load-spz.cc:87,518;
 *      ((b/255)-0.5)/0.15 :522) and the 255 alpha-encode thresholds (smallest float whose
 *      toUint8(sigmoid(a)*255) (:85,301) is >= v) are computed once on the host with the
 *      reference's own expressions and libm, then kept in device memory.  This accessor
 *      returns the host copies (for tests); any pointer may be NULL. ------------------------ */
int spz_amd_get_tables(float alpha_decode[256], float color_decode[256], float alpha_thresholds[255]);

/* ---- self test of the kernels' arithmetic.  Inside an exponent window the quaternion kernels divide
 *      without the IEEE expansion's operand scaling (reciprocal multiply + fma residual corrections,
 *      spz_kernels.hip); this runs, on the device, the comparison of each such form with the plain IEEE
 *      operation it replaces over inputs begin .. begin+count of `mode`:
 *        0  x / 0.70710677f (load-spz.cc:46,244)   x = the float with bit pattern i, window [2^-100, 2^126] and 0
 *        1  x / 511.0f      (load-spz.cc:366)      likewise
 *        2  sqrt            (splat-types.cc:72)    x = bit pattern i, window [2^-80, 2^82]
 *        3  x_i / norm      (splat-types.cc:73)    operand pair hashed from i
 *        4  packQuaternionSmallestThree, 5 the v2 first-three encoder: quaternion hashed from i
 *        6  unpackQuaternionSmallestThree: comp = low 32 bits of i, flips = bits 32..34
 *      result[0] = inputs whose bits differ (must be 0), result[1] = the smallest such i (~0 if none),
 *      result[2] = inputs actually compared (those inside the mode's window).  Blocking. -------------- */
int spz_amd_selftest_device(int mode, uint64_t begin, uint64_t count, uint64_t result[3], void *hip_stream);

/* ---- container stage: zlib's LZ77 parse on the device, exact.  compressGzipped
 *      (load-spz.cc:186-214: one zlib stream, Z_DEFAULT_COMPRESSION, memLevel 9) is what is left of a saveSpz once
 *      the quantise step runs on the GPU, and its output has to stay the reference's bytes.  These entry points
 *      produce the literal/match symbols zlib 1.2.11's deflate_slow + longest_match produce for h_data[0 .. ) up to
 *      the point where they meet the caller's own serial parse of the input's end (`tail_begin`, a multiple of
 *      32768 with 64-96 KiB after it; h_tail_rec: 32768 pairs {lazy-match state, symbols emitted so far} recorded at the
 *      loop tops tail_begin + k of that parse, state 0 where k is not a loop top — spz_deflate.cpp's TopRec).
 *      open: uploads, runs the stages (spz_lz77.hip), returns the symbol count and the index of the first symbol
 *      the tail parse contributes; fetch: copies the symbols out (distance, 0 = literal; literal byte or
 *      length - 3); close: frees the device memory.  SPZ_AMD_ERR_UNSUPPORTED = declined (two neighbouring jobs did not
 *      meet even with parse jobs of 1 MiB, or not enough free device memory: ~25 bytes per input byte): the caller parses on
 *      the host, with the same result.  Blocking; Huffman coding and the gzip framing stay on the host. ---------- */
int spz_amd_zlib_parse_open(const uint8_t *h_data, uint64_t size, uint64_t tail_begin, const uint32_t *h_tail_rec,
                            uint32_t n_rec, int device, void **ctx, uint64_t *num_symbols,
                            uint32_t *tail_first_symbol);
/* The same; produce_tail_rec(arg), if given, is called on the calling thread once the input is on the device and the
 * table and match kernels are running (~90 ms of device work for 650 MB): h_tail_rec need not be filled before it
 * returns, so the caller's serial parse of the input's end can run there instead of before the call. */
int spz_amd_zlib_parse_open_ex(const uint8_t *h_data, uint64_t size, uint64_t tail_begin, const uint32_t *h_tail_rec,
                               uint32_t n_rec, int device, void **ctx, uint64_t *num_symbols,
                               uint32_t *tail_first_symbol, void (*produce_tail_rec)(void *), void *produce_arg);
/* The same for a caller whose input is what spz_amd_encode_host_keep just produced: d_copy (may be NULL) = the
 * device copy of h_data's bytes, on `device`; the upload is then a device-to-device copy. */
int spz_amd_zlib_parse_open_dev(const uint8_t *h_data, const uint8_t *d_copy, uint64_t size, uint64_t tail_begin,
                                const uint32_t *h_tail_rec, uint32_t n_rec, int device, void **ctx,
                                uint64_t *num_symbols, uint32_t *tail_first_symbol, void (*produce_tail_rec)(void *),
                                void *produce_arg);
int spz_amd_zlib_parse_fetch(void *ctx, uint16_t *h_dist, uint8_t *h_lc);
void spz_amd_zlib_parse_close(void *ctx);

/* The Huffman stage of the same member, with the symbols still on the device (trees.c: _tr_tally's counts and
 * compress_block's bit string; the trees — build_tree / gen_bitlen / gen_codes, the stored / static / dynamic
 * choice and the tree headers — are built by the caller from the counts, spz_deflate.cpp).
 *   append        the caller's own symbols of the input's end behind the device's;
 *   block_stats   block b = symbols [b * block_symbols, ...): literal/length and distance frequencies
 *                 ([num_blocks][286], [num_blocks][30]; END_BLOCK not counted), input bytes covered, length of the
 *                 last symbol;
 *   encode_blocks writes every block's header words (placed by the caller on the 32-bit grid of the deflate body,
 *                 which starts at bit 0) and its symbols with the block's codes (or, choice 0, the stored input
 *                 bytes) into a body of body_bytes and copies it to h_body; h_symbol_bits[b] = bits the block's
 *                 symbols + END_BLOCK took (for the caller's check against its plan).
 * The static tables of trees.c (length_code, dist_code, base_length, base_dist, extra bits) come from the caller. */
typedef struct {
  uint8_t length_code[256];
  uint8_t dist_code[512];
  uint16_t base_length[29];
  uint16_t base_dist[30];
  uint8_t extra_lbits[29];
  uint8_t extra_dbits[30];
  uint8_t pad_[3];
} spz_amd_deflate_static;
typedef struct {
  uint64_t bit_start;          /* of the block in the deflate body */
  uint32_t header_word_begin;  /* index into the header word array */
  uint32_t header_words;       /* words, the first one aligned down to the 32-bit grid at bit_start */
  uint32_t header_bits;        /* bits from bit_start to the first symbol (stored: to the first input byte) */
  uint32_t choice;             /* 0 stored, 1 static, 2 dynamic */
  uint32_t input_begin;        /* stored blocks: the input range */
  uint32_t input_bytes;
} spz_amd_deflate_block;
typedef struct {
  uint16_t lcode[286];
  uint16_t dcode[30];
  uint8_t llen[286];
  uint8_t dlen[30];
} spz_amd_deflate_codes;
int spz_amd_zlib_parse_append(void *ctx, const uint16_t *h_dist, const uint8_t *h_lc, uint64_t n);
int spz_amd_zlib_block_stats(void *ctx, const spz_amd_deflate_static *tables, uint32_t block_symbols,
                             uint32_t num_blocks, uint16_t *h_lfreq, uint16_t *h_dfreq, uint32_t *h_bytes,
                             uint32_t *h_last_len);
/* The trees on the device as well (same source as the caller's: spz_huff_core.hpp).  block_trees (after block_stats,
 * whose h_lfreq / h_dfreq may then both be NULL) builds every block's three trees and returns the two lengths
 * _tr_flush_block chooses by; the caller lays the blocks out (bit_start, choice, stored input range; the header_*
 * fields are filled on the device), encode_planned enqueues the writing of headers and symbols and
 * encode_finish_ex (below) copies out the body and each block's symbol bits and header bits.  The order is
 * enforced: block_trees returns SPZ_AMD_ERR_INVALID_ARG unless block_stats has run for the same num_blocks, and
 * encode_planned unless block_trees has. */
typedef struct {
  int64_t opt_len;
  int64_t static_len;
} spz_amd_deflate_plan;
int spz_amd_zlib_block_trees(void *ctx, uint32_t num_blocks, spz_amd_deflate_plan *h_plan);
int spz_amd_zlib_encode_planned(void *ctx, const spz_amd_deflate_static *tables, uint32_t block_symbols,
                                uint32_t num_blocks, const spz_amd_deflate_block *h_blocks, uint64_t body_bytes);
/* encode_blocks in pieces: encode_group enqueues blocks [first_block, first_block + group_blocks) of total_blocks (the
 * arrays hold the group's entries, header_word_begin counts from the group's first header word; the first group zeroes
 * a body of body_bytes_bound) and returns without waiting — the caller builds the next group's trees meanwhile;
 * encode_finish waits and copies the body and every block's bit count out. */
int spz_amd_zlib_encode_group(void *ctx, const spz_amd_deflate_static *tables, uint32_t block_symbols,
                              uint32_t total_blocks, uint32_t first_block, uint32_t group_blocks,
                              const spz_amd_deflate_block *h_blocks, const spz_amd_deflate_codes *h_codes,
                              const uint32_t *h_header_words, uint64_t num_header_words, uint64_t body_bytes_bound);
int spz_amd_zlib_encode_finish(void *ctx, uint32_t total_blocks, uint64_t body_bytes, uint8_t *h_body,
                               uint64_t *h_symbol_bits);
int spz_amd_zlib_encode_finish_ex(void *ctx, uint32_t total_blocks, uint64_t body_bytes, uint8_t *h_body,
                                  uint64_t *h_symbol_bits, uint32_t *h_header_bits /* may be NULL */);
/* The same parse with its table and match stages fed while the input is still being produced (saveSpz: the quantise step's
 * sections become final one after the other while the floats upload, and the device would idle through that upload).
 * session_open(size): the parse's device memory for an input of `size` bytes (SPZ_AMD_ERR_UNSUPPORTED: declined, as
 * parse_open_dev would).  session_feed: bytes [0, final_upto) of d_stream are final once the work queued on
 * `producer_stream` so far is done — they are copied and the table / match kernels for what they cover are enqueued on
 * the session's own stream; returns at once; final_upto must not decrease.  parse_open_session consumes the session
 * (also on failure): whatever has not been fed comes from d_stream (all of it final by now) and the call goes on as
 * parse_open_dev.  session_close: for a session that is not going to be consumed.  spz_amd_encode_host_keep_session is
 * spz_amd_encode_host_keep feeding such a session: the five small sections of all points first, then sh chunk by chunk. */
int spz_amd_zlib_session_open(uint64_t size, int device, void **session);
int spz_amd_zlib_session_feed(void *session, const uint8_t *d_stream, uint64_t final_upto, void *producer_stream);
void spz_amd_zlib_session_close(void *session);
int spz_amd_zlib_parse_open_session(void *session, const uint8_t *h_data, const uint8_t *d_stream, uint64_t size, uint64_t tail_begin,
                                    const uint32_t *h_tail_rec, uint32_t n_rec, void **ctx, uint64_t *num_symbols,
                                    uint32_t *tail_first_symbol, void (*produce_tail_rec)(void *), void *produce_arg);
int spz_amd_encode_host_keep_session(const spz_amd_cloud_in *h_cloud, uint64_t num_points, int sh_degree, int antialiased,
                                     int from_coord, int version, uint8_t *h_stream, size_t capacity, int device,
                                     const uint8_t **d_stream, void *zlib_session);
/* The same for a caller with work of its own on the END of the stream (the container stage's serial tail job: ~8 ms of one
 * host core on the last 64 ... 128 KiB): the sh of the last points go up first, and tail_ready(tail_arg) is called — on
 * the call's download thread; it must return quickly, e.g. after starting a thread — as soon as the stream's last
 * tail_bytes (at least) are in h_stream, long before the call returns.  Not called when tail_bytes is 0, the cloud has
 * no sh, or it is too small for the reordering to pay (the caller then does that work after the call, as without this). */
int spz_amd_encode_host_keep_session_tail(const spz_amd_cloud_in *h_cloud, uint64_t num_points, int sh_degree, int antialiased,
                                          int from_coord, int version, uint8_t *h_stream, size_t capacity, int device,
                                          const uint8_t **d_stream, void *zlib_session, size_t tail_bytes,
                                          void (*tail_ready)(void *), void *tail_arg);
/* Every encode_finish(_ex) returns SPZ_AMD_ERR_VERIFY instead of a body when the symbols it was coded from do not
 * reproduce the input (checked on the device for every block, always: each literal is its input byte, each match
 * copies equal bytes from at most 32 KiB back, each block covers exactly its input range).
 * verify_member (after encode_finish(_ex), before close): inflates the body where it still lies in device memory with
 * the device reader below and compares the result with the input byte for byte: SPZ_AMD_OK = equal, SPZ_AMD_ERR_VERIFY =
 * not, SPZ_AMD_ERR_UNSUPPORTED = the device reader declines this body (the caller checks on the host). */
int spz_amd_zlib_verify_member(void *ctx, uint64_t body_bytes);
int spz_amd_zlib_encode_blocks(void *ctx, const spz_amd_deflate_static *tables, uint32_t block_symbols,
                               uint32_t num_blocks, const spz_amd_deflate_block *h_blocks,
                               const spz_amd_deflate_codes *h_codes, const uint32_t *h_header_words,
                               uint64_t num_header_words, uint64_t body_bytes, uint8_t *h_body,
                               uint64_t *h_symbol_bits);

/* ---- container stage, reading: inflate of one ordinary deflate stream on the device (spz_inflate_dev.hip).  The
 *      reference's files are a single zlib stream (decompressGzipped, load-spz.cc:141-184); it is cut into 64 KiB
 *      chunks whose block starts are found by search, decoded in parallel without their left context and resolved
 *      afterwards (the scheme of spz_inflate.cpp).  h_deflate: the raw deflate data of a gzip member (after its
 *      header, before its 8-byte trailer).  open: decodes; *out_bytes = size of the result, which stays on the
 *      device.  piece_crcs: CRC-32 of consecutive pieces of crc_piece_bytes() of the result — the caller folds them
 *      (crc32_combine) and compares with the trailer's CRC-32 and ISIZE before it believes the result.  fetch: the
 *      bytes; device_data: the device pointer (a decode can read the stream where it is).  close: frees.
 *      SPZ_AMD_ERR_UNSUPPORTED = declined (no usable block starts, chunks that do not link up, a chunk that expands
 *      more than 8 x, not enough device memory): the caller's host readers take over.  Blocking. ------------------- */
int spz_amd_inflate_open(const uint8_t *h_deflate, uint64_t nbytes, int device, void **ctx, uint64_t *out_bytes);
/* inflate_open with a callback that runs on the calling thread once the deflate data is on the device (an upload that is
 * blocking: the data is there when it runs) and before the kernels are waited for: host-side work of the caller that would
 * contend with the upload but not with the kernels (loadSpz maps its output pages there).  Not called when the open
 * fails before or during the upload. */
int spz_amd_inflate_open_ex(const uint8_t *h_deflate, uint64_t nbytes, int device, void **ctx, uint64_t *out_bytes,
                            void (*after_upload)(void *), void *after_arg);
/* the same for deflate data that is in device memory already; equals_device: is the result these nbytes (device memory)? */
int spz_amd_inflate_open_device(const uint8_t *d_deflate, uint64_t nbytes, int device, void **ctx, uint64_t *out_bytes);
int spz_amd_inflate_equals_device(void *ctx, const uint8_t *d_expected, uint64_t nbytes);
/* A context of the same kind around a stream the CALLER has inflated (a member the device reader declines, a raw
 * stream): the bytes are uploaded and stay in device memory until spz_amd_inflate_close; device_data / piece_crcs /
 * fetch work as after inflate_open.  With it "file -> packed sections left in HBM" (loadSpzPacked for renderers,
 * load-spz.cc:609-632) has one shape whichever reader inflated the member. */
int spz_amd_stream_to_device(const uint8_t *h_stream, uint64_t nbytes, int device, void **ctx);
uint32_t spz_amd_inflate_crc_piece_bytes(void);
/* Why the last inflate_open(_device) of this thread returned SPZ_AMD_ERR_UNSUPPORTED ("" when it did not): one of
 * size, stored-first, memory, no-block-starts, symbol-budget, expansion, no-final-block, not-linked, trailing-bytes,
 * empty, window-chains, bad-reference.  The caller's host readers give the same bytes; this tells a slowdown's cause. */
const char *spz_amd_inflate_last_decline(void);
int spz_amd_inflate_piece_crcs(void *ctx, uint32_t *h_crcs, uint32_t capacity, uint32_t *num_pieces);
int spz_amd_inflate_fetch(void *ctx, uint8_t *h_out);
const uint8_t *spz_amd_inflate_device_data(void *ctx);
void spz_amd_inflate_close(void *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SPZ_AMD_H_ */
