// spz_exchange.hip — the one exchange step of the multi-GPU path: reassembling the byte stream of
// point-range shards on one device (BASELINE config 5).
//
// The stream is attribute-major (load-spz.cc:540-545), so rank r's shard [first_r, first_r + count_r) is six
// fragments, and section s of the result is the concatenation over ranks of fragment (r, s).  Two routes,
// both native (no torch in the way):
//
//  * RCCL: ONE ncclGroupStart .. ncclGroupEnd per call holding, for every non-root rank, the six ncclSend of
//    its fragments matched by six ncclRecv on the root DIRECTLY at their final offsets (RCCL has no
//    gatherv; counts differ per section).  A section mask lets a caller send the small sections while the
//    sh section is still being encoded.  RCCL is resolved at run time (dlopen of librccl.so.1: the copy the
//    process already holds, e.g. torch's, else the one in /opt/rocm/lib), so libspz_amd.so has no link
//    dependency on it and there is never a second copy in the process.
//  * IPC: the root exports its stream buffer (hipIpcGetMemHandle), every peer maps it and passes the
//    mapped pointer as `d_stream` to spz_amd_encode_shard_device: the encode kernel's own stores land the
//    fragments at their final offsets over xGMI, there is no second pass over the bytes.  Completion is
//    the peer's stream completion plus whatever barrier the caller already has.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstring>
#include <mutex>

#include "spz_amd.h"
#include "spz_common.hpp"

namespace {

using namespace spz_amd_detail;

struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};

const Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, []() {
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // the copy the process already has
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
  });
  return r;
}

thread_local int g_last_rccl_error = 0;
#define SPZ_RCCL_TRY(expr)               \
  do {                                   \
    ncclResult_t r_ = (expr);            \
    if (r_ != ncclSuccess) {             \
      g_last_rccl_error = (int)r_;       \
      return SPZ_AMD_ERR_COMM;           \
    }                                    \
  } while (0)

static_assert(sizeof(ncclUniqueId) == SPZ_AMD_RCCL_UNIQUE_ID_BYTES, "unique id size");
static_assert(sizeof(hipIpcMemHandle_t) == SPZ_AMD_IPC_HANDLE_BYTES, "ipc handle size");

}  // namespace

extern "C" {

int spz_amd_shard_fragments(uint64_t first, uint64_t count, uint64_t num_points_total, int sh_degree, int version,
                            spz_amd_fragments *out) {
  if (out == nullptr || first > num_points_total || count > num_points_total - first) return SPZ_AMD_ERR_INVALID_ARG;
  spz_amd_layout g, l;
  int rc = spz_amd_stream_layout(num_points_total, sh_degree, version, &g);
  if (rc != SPZ_AMD_OK) return rc;
  rc = spz_amd_stream_layout(count, sh_degree, version, &l);
  if (rc != SPZ_AMD_OK) return rc;
  for (int s = 0; s < SPZ_AMD_NUM_SECTIONS; ++s) {
    out->global_offset[s] = g.offset[s] + first * g.bytes_per_point[s];
    out->local_offset[s] = l.offset[s];
    out->bytes[s] = count * g.bytes_per_point[s];
  }
  return SPZ_AMD_OK;
}

int spz_amd_rccl_available(void) { return rccl().ok ? 1 : 0; }
int spz_amd_last_rccl_error(void) { return g_last_rccl_error; }

int spz_amd_rccl_unique_id(uint8_t id[SPZ_AMD_RCCL_UNIQUE_ID_BYTES]) {
  if (id == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  if (!rccl().ok) return SPZ_AMD_ERR_UNSUPPORTED;
  ncclUniqueId u;
  SPZ_RCCL_TRY(rccl().GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return SPZ_AMD_OK;
}

int spz_amd_rccl_comm_init(const uint8_t id[SPZ_AMD_RCCL_UNIQUE_ID_BYTES], int world, int rank, void **comm) {
  if (id == nullptr || comm == nullptr || world < 1 || rank < 0 || rank >= world) return SPZ_AMD_ERR_INVALID_ARG;
  if (!rccl().ok) return SPZ_AMD_ERR_UNSUPPORTED;
  int device = 0;
  int rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  ncclComm_t c = nullptr;
  SPZ_RCCL_TRY(rccl().CommInitRank(&c, world, u, rank));
  *comm = c;
  return SPZ_AMD_OK;
}

int spz_amd_rccl_comm_destroy(void *comm) {
  if (comm == nullptr) return SPZ_AMD_OK;
  if (!rccl().ok) return SPZ_AMD_ERR_UNSUPPORTED;
  SPZ_RCCL_TRY(rccl().CommDestroy(static_cast<ncclComm_t>(comm)));
  return SPZ_AMD_OK;
}

namespace {

// One ncclGroupStart/End moving every selected fragment between the root's global stream and the ranks' own streams:
// toward the root (gather) or away from it (scatter).
static int exchange_rccl(bool gather, void *comm, int rank, int world, int root, const uint64_t *first, const uint64_t *count,
                  int sh_degree, int version, uint8_t *d_local_stream, uint8_t *d_global_stream, unsigned section_mask,
                  void *hip_stream) {
  if (comm == nullptr || first == nullptr || count == nullptr || world < 1 || rank < 0 || rank >= world || root < 0 ||
      root >= world) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  if (!rccl().ok) return SPZ_AMD_ERR_UNSUPPORTED;
  uint64_t total = 0;
  for (int r = 0; r < world; ++r) {
    if (first[r] != total) return SPZ_AMD_ERR_INVALID_ARG;  // contiguous ranges in rank order
    total += count[r];
  }
  if (rank == root && d_global_stream == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  if (rank != root && d_local_stream == nullptr && count[rank] > 0) return SPZ_AMD_ERR_INVALID_ARG;
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  spz_amd_fragments f;
  // every fragment table is validated before the group is opened: nothing may fail between Start and End
  for (int r = 0; r < world; ++r) {
    const int rc = spz_amd_shard_fragments(first[r], count[r], total, sh_degree, version, &f);
    if (rc != SPZ_AMD_OK) return rc;
  }
  SPZ_RCCL_TRY(rccl().GroupStart());
  ncclResult_t bad = ncclSuccess;
  auto send = [&](const void *p, size_t n, int peer) {
    if (bad == ncclSuccess) bad = rccl().Send(p, n, ncclUint8, peer, c, st);
  };
  auto recv = [&](void *p, size_t n, int peer) {
    if (bad == ncclSuccess) bad = rccl().Recv(p, n, ncclUint8, peer, c, st);
  };
  for (int r = 0; r < world; ++r) {
    // the root's own fragments travel too (to / from itself) when it keeps a stream of its own besides the global one
    const bool local_on_root = (r == root && d_local_stream != nullptr);
    if (r == root && !local_on_root) continue;
    if (rank != root && rank != r) continue;
    (void)spz_amd_shard_fragments(first[r], count[r], total, sh_degree, version, &f);
    for (int s = 0; s < SPZ_AMD_NUM_SECTIONS; ++s) {
      if (!((section_mask >> s) & 1u) || f.bytes[s] == 0) continue;
      if (gather) {
        if (rank == r) send(d_local_stream + f.local_offset[s], f.bytes[s], root);
        if (rank == root) recv(d_global_stream + f.global_offset[s], f.bytes[s], r);
      } else {
        if (rank == root) send(d_global_stream + f.global_offset[s], f.bytes[s], r);
        if (rank == r) recv(d_local_stream + f.local_offset[s], f.bytes[s], root);
      }
    }
  }
  const ncclResult_t end = rccl().GroupEnd();
  SPZ_RCCL_TRY(bad);
  SPZ_RCCL_TRY(end);
  return SPZ_AMD_OK;
}

}  // namespace

int spz_amd_gatherv_rccl(void *comm, int rank, int world, int root, const uint64_t *first, const uint64_t *count,
                         int sh_degree, int version, const uint8_t *d_local_stream, uint8_t *d_global_stream,
                         unsigned section_mask, void *hip_stream) {
  return exchange_rccl(true, comm, rank, world, root, first, count, sh_degree, version,
                       const_cast<uint8_t *>(d_local_stream), d_global_stream, section_mask, hip_stream);
}

int spz_amd_scatterv_rccl(void *comm, int rank, int world, int root, const uint64_t *first, const uint64_t *count,
                          int sh_degree, int version, const uint8_t *d_global_stream, uint8_t *d_local_stream,
                          unsigned section_mask, void *hip_stream) {
  return exchange_rccl(false, comm, rank, world, root, first, count, sh_degree, version, d_local_stream,
                       const_cast<uint8_t *>(d_global_stream), section_mask, hip_stream);
}

int spz_amd_ipc_alloc(size_t bytes, void **d_ptr, uint8_t handle[SPZ_AMD_IPC_HANDLE_BYTES]) {
  if (d_ptr == nullptr || handle == nullptr || bytes == 0) return SPZ_AMD_ERR_INVALID_ARG;
  int device = 0;
  int rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  void *p = nullptr;
  SPZ_HIP_TRY(hipMalloc(&p, bytes));
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    (void)hipFree(p);
    return SPZ_AMD_ERR_HIP;
  }
  std::memcpy(handle, &h, sizeof(h));
  *d_ptr = p;
  return SPZ_AMD_OK;
}

int spz_amd_ipc_free(void *d_ptr) {
  if (d_ptr == nullptr) return SPZ_AMD_OK;
  SPZ_HIP_TRY(hipFree(d_ptr));
  return SPZ_AMD_OK;
}

int spz_amd_ipc_open(const uint8_t handle[SPZ_AMD_IPC_HANDLE_BYTES], void **d_ptr) {
  if (handle == nullptr || d_ptr == nullptr) return SPZ_AMD_ERR_INVALID_ARG;
  int device = 0;
  int rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  void *p = nullptr;
  SPZ_HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  *d_ptr = p;
  return SPZ_AMD_OK;
}

int spz_amd_ipc_close(void *d_ptr) {
  if (d_ptr == nullptr) return SPZ_AMD_OK;
  SPZ_HIP_TRY(hipIpcCloseMemHandle(d_ptr));
  return SPZ_AMD_OK;
}

}  // extern "C"
