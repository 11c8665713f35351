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
__device__ __forceinline__ uint32_t scale_sum_key(float s0, float s1, float s2) {
  float sum = (s0 + s1) + s2;
  const uint32_t u = __float_as_uint(sum);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

// h[bin] += 1 for every lane with `valid`, without the same-address serialisation of 64 LDS atomics: real
// scale sums put most of a wave into two or three bins of the first pass (sign + exponent bits).  Up to four
// rounds of "take the first pending lane's bin, count the lanes that share it with one ballot, one lane adds
// the count"; lanes still pending after that are spread over many bins and add for themselves.
__device__ __forceinline__ void wave_histogram_add(uint32_t *h, uint32_t bin, bool valid) {
  unsigned long long todo = __ballot(valid);
#pragma unroll 1
  for (int round = 0; round < 4 && todo != 0ull; ++round) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t b = (uint32_t)__shfl((int)bin, leader, 64);
    const bool mine = valid && bin == b;
    const unsigned long long same = __ballot(mine);
    if ((int)(threadIdx.x & 63u) == leader) atomicAdd(&h[b], (uint32_t)__popcll(same));
    valid = valid && !mine;
    todo &= ~same;
  }
  if (valid) atomicAdd(&h[bin], 1u);
}

constexpr int kSelectPointsPerThread = 8;  // 24 floats = six 16-byte loads in flight per lane, 6 KiB contiguous per wave

__global__ __launch_bounds__(256) void spz_select_hist_kernel(const float *__restrict__ scales, uint64_t n,
                                                              SelectState *__restrict__ st, int pass) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t prefix = st->prefix, mask = st->mask;
  const int shift = 24 - 8 * pass;
  const uint64_t stride = (uint64_t)gridDim.x * 256 * kSelectPointsPerThread;
  const uint64_t n_round = (n + stride - 1) / stride * stride;  // every lane of a wave makes the same number of trips
  for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * kSelectPointsPerThread; i < n_round; i += stride) {
    float v[3 * kSelectPointsPerThread];
    if (i + kSelectPointsPerThread <= n) {
      const F32x4 *q = reinterpret_cast<const F32x4 *>(scales + 3 * i);
#pragma unroll
      for (int r = 0; r < 3 * kSelectPointsPerThread / 4; ++r) {
        const F32x4 t = q[r];
        v[4 * r + 0] = t.x; v[4 * r + 1] = t.y; v[4 * r + 2] = t.z; v[4 * r + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 3 * kSelectPointsPerThread; ++r) v[r] = (3 * i + r < 3 * n) ? scales[3 * i + r] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < kSelectPointsPerThread; ++r) {
      const uint32_t key = scale_sum_key(v[3 * r], v[3 * r + 1], v[3 * r + 2]);
      wave_histogram_add(h, (key >> shift) & 255u, i + r < n && (key & mask) == prefix);
    }
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
  // 2048 points per block and trip; at most 8 blocks per CU (the per-block flush into the 256 global bins stays
  // at ~2000 adds per bin), every block the same number of trips
  unsigned long long blocks = (num_points + 256 * kSelectPointsPerThread - 1) / (256 * kSelectPointsPerThread);
  if (blocks > 2048) {
    const unsigned long long trips = (blocks + 2047) / 2048;
    blocks = (blocks + trips - 1) / trips;
  }
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
