// spz_median.hip — the median of the per-Gaussian log-scale sums, i.e. the selection step of
// GaussianCloud::medianVolume (/root/reference/src/cc/splat-types.h:170-185: sum = s0 + s1 + s2 per
// point, sort, take element size/2), without the sort.
//
// Exact k-th smallest by most-significant-digit radix selection on order-preserving integer keys:
// four passes, each a 256-bin histogram of the next 8 key bits over the points whose higher bits
// already match, followed by a one-wave kernel that picks the bin holding rank k.  The scale sums are
// recomputed from the scales in every pass (12 B/point/pass, no key buffer), so the state is 4 KiB and
// the whole selection reads 48 B per point: HBM-bound, 4 launches of streaming reads.  No host
// synchronisation between passes; the result is a float in device memory.
#include "spz_common.hpp"

namespace spz_amd_detail {

struct SelectState {
  uint32_t prefix;  // key bits decided so far
  uint32_t mask;    // which bits those are
  uint32_t k;       // rank still to find among the keys that match prefix
  uint32_t pad;
  uint32_t hist[4][256];
};
static_assert(sizeof(SelectState) <= SPZ_AMD_MEDIAN_WORKSPACE_BYTES, "workspace constant too small");

// (s0 + s1) + s2 in f32, left to right like the reference's expression, as an unsigned key whose
// integer order is the float order (-0.0 sorts directly below +0.0; the volume is the same for both).
__device__ __forceinline__ uint32_t scale_sum_key(const float *__restrict__ s, uint64_t i) {
  const float *p = s + 3 * i;
  float sum = (p[0] + p[1]) + p[2];
  const uint32_t u = __float_as_uint(sum);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

__global__ __launch_bounds__(256) void spz_select_hist_kernel(const float *__restrict__ scales, uint64_t n,
                                                              SelectState *__restrict__ st, int pass) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t prefix = st->prefix, mask = st->mask;
  const int shift = 24 - 8 * pass;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint32_t key = scale_sum_key(scales, i);
    if ((key & mask) == prefix) atomicAdd(&h[(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&st->hist[pass][threadIdx.x], h[threadIdx.x]);
}

// One wave: exclusive prefix sum over the 256 bins (4 per lane), pick the bin that holds rank k.
__global__ __launch_bounds__(64) void spz_select_pick_kernel(SelectState *__restrict__ st, int pass, uint32_t k0,
                                                             float *__restrict__ out) {
  const uint32_t lane = threadIdx.x;
  uint32_t c[4], mine = 0;
  for (int j = 0; j < 4; ++j) {
    c[j] = st->hist[pass][lane * 4 + j];
    mine += c[j];
  }
  uint32_t incl = mine;  // inclusive scan across the wave
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += up;
  }
  const uint32_t k = pass == 0 ? k0 : st->k;  // the rank sought arrives as an argument: no host copy to wait for
  uint32_t below = incl - mine;
  if (k >= below && k < incl) {  // exactly one lane (bins are disjoint rank intervals)
    for (int j = 0; j < 4; ++j) {
      if (k < below + c[j]) {
        const int shift = 24 - 8 * pass;
        const uint32_t prefix = st->prefix | ((lane * 4 + j) << shift);
        st->prefix = prefix;
        st->mask = st->mask | (0xffu << shift);
        st->k = k - below;
        if (pass == 3) {
          const uint32_t bits = prefix ^ ((prefix >> 31) ? 0x80000000u : 0xffffffffu);
          *out = __uint_as_float(bits);
        }
        break;
      }
      below += c[j];
    }
  }
}

}  // namespace spz_amd_detail

using namespace spz_amd_detail;

extern "C" {

int spz_amd_median_scale_sum_device(const float *d_scales, uint64_t num_points, void *d_workspace,
                                    float *d_median, void *hip_stream) {
  if (d_scales == nullptr || d_workspace == nullptr || d_median == nullptr || num_points == 0 ||
      num_points > 0xffffffffull) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  int device = 0;
  int rc = current_device(&device);
  if (rc != SPZ_AMD_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  SelectState *st = static_cast<SelectState *>(d_workspace);
  SPZ_HIP_TRY(hipMemsetAsync(st, 0, sizeof(SelectState), s));
  const uint32_t k = (uint32_t)(num_points / 2);  // rank size/2 of the sorted sums (splat-types.h:182)
  unsigned long long blocks = (num_points + 255) / 256;
  if (blocks > 8192) blocks = 8192;  // grid-stride beyond 2 M points: 32 blocks per CU
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(spz_select_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d_scales, num_points, st, pass);
    hipLaunchKernelGGL(spz_select_pick_kernel, dim3(1), dim3(64), 0, s, st, pass, k, d_median);
  }
  SPZ_HIP_TRY(hipGetLastError());
  return SPZ_AMD_OK;
}

int spz_amd_median_scale_sum_host(const float *h_scales, uint64_t num_points, float *h_median, int device) {
  if (h_scales == nullptr || h_median == nullptr || num_points == 0 || num_points > 0xffffffffull) {
    return SPZ_AMD_ERR_INVALID_ARG;
  }
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != SPZ_AMD_OK) return rc;
  const size_t bytes = (size_t)num_points * 3 * sizeof(float);
  Workspace ws;
  rc = ws.open(device, Workspace::aligned(bytes) + Workspace::aligned(SPZ_AMD_MEDIAN_WORKSPACE_BYTES) +
                           Workspace::aligned(sizeof(float)));
  if (rc != SPZ_AMD_OK) return rc;
  float *d_scales = static_cast<float *>(ws.take(bytes));
  void *d_state = ws.take(SPZ_AMD_MEDIAN_WORKSPACE_BYTES);
  float *d_out = static_cast<float *>(ws.take(sizeof(float)));
  SPZ_HIP_TRY(hipMemcpyAsync(d_scales, h_scales, bytes, hipMemcpyHostToDevice, nullptr));
  rc = spz_amd_median_scale_sum_device(d_scales, num_points, d_state, d_out, nullptr);
  if (rc != SPZ_AMD_OK) return rc;
  SPZ_HIP_TRY(hipMemcpyAsync(h_median, d_out, sizeof(float), hipMemcpyDeviceToHost, nullptr));
  SPZ_HIP_TRY(hipStreamSynchronize(nullptr));
  return SPZ_AMD_OK;
}

}  // extern "C"
