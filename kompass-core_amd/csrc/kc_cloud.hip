// Raw point cloud -> laserscan on gfx950.
//
// Reference: pointCloudToLaserScanFromRaw, utils/pointcloud.h:116-177 (angle
// step) and :205-259 (bin count); the SYCL kernel of the reference
// (local_mapper_gpu.cpp:59-164) is a float re-implementation with its own
// atan2 and is NOT what this reproduces: the contract here is the CPU loop.
//
// One lane per point record.  The bin of a point depends on the host libm's
// atan2f (std::atan2(float, float)); the device evaluates atan2 in double and
// keeps a point only when its angle is more than 1e-6 rad away from both edges
// of its bin -- the float result of a < 1 ulp atan2f (2.4e-7 rad at pi) cannot
// fall into another bin then.  (A float atan2 settles the points farther than
// 3e-6 rad from an edge first; only the rest pays for the double one.)  The few points closer to an edge go to a list
// the host re-bins with the reference expression.  Distances are
// sqrtf(x*x + y*y) (correctly rounded on both sides) and the per-bin minimum
// is a 64-bit atomic min on the bits of the non-negative double.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include "kc_internal.h"
#include "kompass_hip.h"

namespace kc {

struct CloudArgs {
  const uint8_t *data;
  size_t nbytes;
  long long n_records;   // height * ceil(row_step / point_step)
  int per_row;           // records per row
  int point_step, row_step;
  int x_off, y_off, z_off, max_off;
  double min_z, max_z;
  double angle_step;     // > 0: bin = int(angle / angle_step); else int(angle / 2pi * num_bins)
  int num_bins;
  double edge;           // 1e-6 rad expressed in bins
  unsigned long long max_bits;   // bits of max_range
  unsigned long long *bins;      // [num_bins] double bits (kLds = false: armed with max_range)
  unsigned long long *partial;   // [gridDim.x][num_bins] per-workgroup minima (kLds = true)
  unsigned int *list_count;
  float2 *list;                  // (x, y) of the points the host has to bin
  int packed16;                  // aligned 16-byte records with x, y, z in front, rows contiguous
  long long n_packed;            // ... and how many of them fit the buffer
  int ftype, fsize;              // PointFieldType of the x / y / z fields (utils/pointcloud.h:37-46) and its size
};

// load_and_cast_val, utils/pointcloud.h:49-87: a field of any PointCloud2 datatype as float, byte by byte
__device__ __forceinline__ float load_field(const uint8_t *p, int t) {
  auto u16 = [&]() { return static_cast<uint16_t>(static_cast<uint16_t>(p[0]) | (static_cast<uint16_t>(p[1]) << 8)); };
  auto u32 = [&]() {
    return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) | (static_cast<uint32_t>(p[2]) << 16) |
           (static_cast<uint32_t>(p[3]) << 24);
  };
  switch (t) {
    case KC_FIELD_INT8: return static_cast<float>(static_cast<int8_t>(p[0]));
    case KC_FIELD_UINT8: return static_cast<float>(p[0]);
    case KC_FIELD_INT16: return static_cast<float>(static_cast<int16_t>(u16()));
    case KC_FIELD_UINT16: return static_cast<float>(u16());
    case KC_FIELD_INT32: return static_cast<float>(static_cast<int32_t>(u32()));
    case KC_FIELD_UINT32: return static_cast<float>(u32());
    case KC_FIELD_FLOAT32: return __uint_as_float(u32());
    case KC_FIELD_FLOAT64: {
      const uint64_t lo = u32();
      p += 4;
      const uint64_t hi = u32();
      return static_cast<float>(__longlong_as_double(static_cast<long long>(lo | (hi << 32))));
    }
    default: return 0.0f;
  }
}

__device__ __forceinline__ float load_f32(const uint8_t *p) {
  uint32_t u;
  if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) {
    u = *reinterpret_cast<const uint32_t *>(p);
  } else {
    u = static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) |
        (static_cast<uint32_t>(p[2]) << 16) | (static_cast<uint32_t>(p[3]) << 24);
  }
  return __uint_as_float(u);
}

template <bool kLds>
__device__ __forceinline__ void cloud_point(const CloudArgs &a, unsigned long long *lbins, float x,
                                            float y, float z) {
  const double two_pi = 2.0 * M_PI;
  const float xx = x * x, yy = y * y;
  const float range_sq = xx + yy;                       // :153
  if (static_cast<double>(range_sq) < 1e-6) return;
  if (static_cast<double>(z) < a.min_z || (a.max_z >= 0.0 && static_cast<double>(z) > a.max_z))
    return;                                             // :159
  if (!isfinite(x) || !isfinite(y)) return;             // reference: int(NaN) index
  // Two tiers: a float atan2 decides the bin of a point more than 3e-6 rad
  // (six times the largest float-vs-float atan2 discrepancy at pi) away from
  // an edge; closer ones are redone in double against the 1e-6 rad margin.
  double ang = static_cast<double>(atan2f(y, x));
  if (ang < 0.0) ang += two_pi;
  double t = a.angle_step > 0.0 ? ang / a.angle_step : (ang / two_pi) * a.num_bins;
  double fl = floor(t);
  double frac = t - fl;
  if (frac < 3.0 * a.edge || frac > 1.0 - 3.0 * a.edge) {
    ang = atan2(static_cast<double>(y), static_cast<double>(x));
    if (ang < 0.0) ang += two_pi;
    t = a.angle_step > 0.0 ? ang / a.angle_step : (ang / two_pi) * a.num_bins;
    fl = floor(t);
    frac = t - fl;
    if (frac < a.edge || frac > 1.0 - a.edge) {
      // too close to a bin edge to trust a different atan2: the host decides
      a.list[atomicAdd(a.list_count, 1u)] = make_float2(x, y);
      return;
    }
  }
  int bin = static_cast<int>(fl);
  bin = min(bin, a.num_bins - 1);
  const float distf = kc::sqrt_rn(range_sq);  // std::sqrt(float)
  if (!(distf >= 0.0f)) return;
  if (kLds) {
    // the minimum of the widened floats is the widened minimum: the workgroup keeps FLOAT bits (non-negative
    // floats order like their bit patterns; half the LDS and half the row traffic), the merge widens once
    // and applies max_range
    uint32_t *l32 = reinterpret_cast<uint32_t *>(lbins);
    const uint32_t b32 = __float_as_uint(distf);
    if (b32 < l32[bin]) atomicMin(&l32[bin], b32);
  } else {
    const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(static_cast<double>(distf)));
    atomicMin(&a.bins[bin], bits);
  }
}

constexpr int kCloudBlock = 1024;
constexpr int kCloudMaxLdsBins = 8192;  // 64 KB of LDS bins per workgroup

// kLds: one workgroup per CU strides over the records and keeps its minima in
// LDS (ds_min_u64), then writes them out as one row of `partial` -- no global
// atomics; cloud_merge_kernel takes the column minima.  Otherwise (more bins
// than LDS holds) every point goes to a global atomic min.
template <bool kLds>
__global__ __launch_bounds__(kCloudBlock) void cloud_bins_kernel(CloudArgs a) {
  extern __shared__ unsigned long long lbins[];
  if (kLds) {
    uint32_t *l32 = reinterpret_cast<uint32_t *>(lbins);
    for (int b = threadIdx.x; b < a.num_bins; b += kCloudBlock) l32[b] = 0x7F800000u;  // +inf: no point yet
    __syncthreads();
  }
  const long long stride = static_cast<long long>(gridDim.x) * kCloudBlock;
  const long long first = static_cast<long long>(blockIdx.x) * kCloudBlock + threadIdx.x;
  if (a.packed16) {
    // x, y, z are the first 12 bytes of aligned 16-byte records: one 128-bit
    // load per point, four points in flight per lane
    const float4 *rec = reinterpret_cast<const float4 *>(a.data);
    const long long n = a.n_packed;  // records that lie completely inside the buffer
    for (long long i = first; i < n; i += 4 * stride) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long j = i + u * stride;
        v[u] = rec[j < n ? j : i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + u * stride < n) cloud_point<kLds>(a, lbins, v[u].x, v[u].y, v[u].z);
    }
  } else {
    for (long long i = first; i < a.n_records; i += stride) {
      const long long row = i / a.per_row;
      const long long col = (i - row * a.per_row) * a.point_step;
      const size_t start = static_cast<size_t>(row) * a.row_step + static_cast<size_t>(col);
      if (start + static_cast<size_t>(a.max_off) + static_cast<size_t>(a.fsize) > a.nbytes) continue;  // :139-146
      if (a.ftype == KC_FIELD_FLOAT32)
        cloud_point<kLds>(a, lbins, load_f32(a.data + start + a.x_off), load_f32(a.data + start + a.y_off),
                          load_f32(a.data + start + a.z_off));
      else
        cloud_point<kLds>(a, lbins, load_field(a.data + start + a.x_off, a.ftype),
                          load_field(a.data + start + a.y_off, a.ftype), load_field(a.data + start + a.z_off, a.ftype));
    }
  }
  if (kLds) {
    __syncthreads();
    uint32_t *rowp = reinterpret_cast<uint32_t *>(a.partial) + static_cast<size_t>(blockIdx.x) * a.num_bins;
    const uint32_t *l32 = reinterpret_cast<const uint32_t *>(lbins);
    for (int b = threadIdx.x; b < a.num_bins; b += kCloudBlock) rowp[b] = l32[b];
  }
}

// column minima of the per-workgroup rows
// ... and hands the result to the host: bins, edge-list count and (last
// workgroup) the sequence number go straight into pinned host memory, which
// the host polls instead of copying and waiting on the stream
constexpr int kMergeBins = 16;  // bins per workgroup of the merge: 64 row lanes each
__global__ __launch_bounds__(1024) void cloud_merge_kernel(const uint32_t *partial, int rows, int num_bins,
                                                           double max_range, unsigned long long *bins,
                                                           const unsigned int *count_now,
                                                           unsigned int *next_count, unsigned int *ticket,
                                                           unsigned long long *host_out, long long seq) {
  // 16 bins per workgroup (a 64-byte segment of every row), 64 lanes per bin over the rows -- every load of a
  // lane in flight at once (512 rows: eight) --, the minimum over the lanes of a bin by two DPP-free shuffles
  // inside the wavefront (4 row lanes x 16 bins) and an LDS pass over the 16 wavefronts
  __shared__ uint32_t part[16][kMergeBins];
  const int bl = threadIdx.x & (kMergeBins - 1), rg = threadIdx.x >> 4;  // rg: 0 .. 63
  const int b = blockIdx.x * kMergeBins + bl;
  uint32_t m = 0x7F800000u;
  if (b < num_bins) {
#pragma unroll 8
    for (int r = rg; r < rows; r += 64) {
      const uint32_t v = partial[static_cast<size_t>(r) * num_bins + b];
      m = v < m ? v : m;
    }
  }
  {
    const uint32_t o = static_cast<uint32_t>(__shfl_xor(static_cast<int>(m), 16, 64));
    m = o < m ? o : m;
    const uint32_t o2 = static_cast<uint32_t>(__shfl_xor(static_cast<int>(m), 32, 64));
    m = o2 < m ? o2 : m;
  }
  if ((threadIdx.x & 63) < kMergeBins) part[threadIdx.x >> 6][bl] = m;
  __syncthreads();
  if (threadIdx.x < kMergeBins && b < num_bins) {
#pragma unroll
    for (int k = 1; k < 16; ++k) m = part[k][bl] < m ? part[k][bl] : m;
    // widen once; `distance < ranges[bin]` against the initial max_range (pointcloud.h:170-175)
    const double d = static_cast<double>(__uint_as_float(m));
    const double out = d < max_range ? d : max_range;
    const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(out));
    bins[b] = bits;
    host_out[2 + b] = bits;
    __threadfence_system();
  }
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {
    *ticket = 0u;
    host_out[1] = *count_now;
    *next_count = 0u;  // the edge-list counter of the NEXT call (the two alternate)
    __threadfence_system();
    *reinterpret_cast<volatile unsigned long long *>(host_out) = static_cast<unsigned long long>(seq);
  }
}

__global__ void cloud_arm_kernel(unsigned long long *bins, int n, double max_range,
                                 unsigned int *counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (bins && i < n) bins[i] = static_cast<unsigned long long>(__double_as_longlong(max_range));
  if (i == 0) counts[0] = counts[1] = 0u;  // both edge-list counters
}

}  // namespace kc

using namespace kc;

struct kc_cloud {
  int device = 0;
  hipStream_t stream = nullptr;
  Timing timing;
  DevBuf<uint8_t> d_data;
  DevBuf<unsigned long long> d_bins;
  DevBuf<unsigned long long> d_partial;
  DevBuf<unsigned int> d_count;   // [2]: edge-list counters, alternating between calls; [2]: ticket
  unsigned calls = 0;
  PinBuf<unsigned long long> h_out;  // [0] sequence, [1] edge count, [2..] bins (written by the merge kernel)
  long long seq = 0;
  DevBuf<float2> d_list;         // (x, y) of the edge points
  PinBuf<unsigned long long> h_bins;
  PinBuf<unsigned int> h_count;
  PinBuf<float2> h_list;
  bool lds_ok = false;
  size_t last_rebinned = 0;
};

extern "C" {

int kc_cloud_create(size_t max_bytes, size_t max_bins, int device, kc_cloud **out) {
  if (!out) KC_FAIL(KC_ERR_INVALID, "null argument");
  *out = nullptr;
  int ndev = 0;
  KC_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev)
    KC_FAIL(KC_ERR_HIP, "HIP device %d not available (%d visible)", device, ndev);
  auto *c = new kc_cloud();
  c->device = device;
  auto fail = [&](int rc) {
    kc_cloud_destroy(c);
    return rc;
  };
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("HIP stream creation failed on device %d", device);
    return fail(KC_ERR_HIP);
  }
  int rc;
  if ((rc = c->d_data.reserve(std::max<size_t>(max_bytes, 64))) ||
      (rc = c->d_bins.reserve(std::max<size_t>(max_bins, 16))) ||
      (rc = c->h_bins.reserve(std::max<size_t>(max_bins, 16))) ||
      (rc = c->d_count.reserve(3)) || (rc = c->h_count.reserve(1)) ||
      (rc = c->h_out.reserve(std::max<size_t>(max_bins, 16) + 2)))
    return fail(rc);
  c->h_out.p[0] = 0;
  if (hipMemset(c->d_count.p, 0, 3 * sizeof(unsigned int)) != hipSuccess) {
    set_error("counter initialisation failed");
    return fail(KC_ERR_HIP);
  }
  c->lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(cloud_bins_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kCloudMaxLdsBins * 8) == hipSuccess;
  if (!c->lds_ok) (void)hipGetLastError();
  *out = c;
  return KC_OK;
}

void kc_cloud_destroy(kc_cloud *c) {
  if (!c) return;
  hipError_t e = hipSetDevice(c->device);
  if (c->stream) {
    e = hipStreamSynchronize(c->stream);
    e = hipStreamDestroy(c->stream);
  }
  (void)e;
  c->timing.release();
  c->d_data.release();
  c->d_bins.release();
  c->d_list.release();
  c->d_partial.release();
  c->d_count.release();
  c->h_bins.release();
  c->h_count.release();
  c->h_out.release();
  c->h_list.release();
  delete c;
}

int kc_cloud_to_laserscan(kc_cloud *c, const int8_t *data, size_t nbytes,
                          int data_on_device, int point_step, int row_step,
                          int height, int width, int x_offset, int y_offset,
                          int z_offset, double max_range, double min_z,
                          double max_z, double angle_step, int num_bins,
                          double *ranges_out, double *angles_out, size_t cap,
                          size_t *bins_out) {
  return kc_cloud_to_laserscan_typed(c, data, nbytes, data_on_device, point_step, row_step, height, width, x_offset,
                                     y_offset, z_offset, KC_FIELD_FLOAT32, max_range, min_z, max_z, angle_step, num_bins,
                                     ranges_out, angles_out, cap, bins_out);
}

int kc_cloud_to_laserscan_typed(kc_cloud *c, const int8_t *data, size_t nbytes,
                                int data_on_device, int point_step, int row_step,
                                int height, int width, int x_offset, int y_offset,
                                int z_offset, int field_type, double max_range, double min_z,
                                double max_z, double angle_step, int num_bins,
                                double *ranges_out, double *angles_out, size_t cap,
                                size_t *bins_out) {
  (void)width;  // pointcloud.h:137-138: the loops use row_step / point_step only
  if (!c || !ranges_out || (nbytes && !data)) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (field_type < KC_FIELD_INT8 || field_type > KC_FIELD_FLOAT64)
    KC_FAIL(KC_ERR_INVALID, "Invalid integer for PointFieldType. Must be 1-8.");
  static const int kFieldSize[9] = {0, 1, 1, 2, 2, 4, 4, 4, 8};
  if (point_step <= 0 || row_step < 0 || height < 0 || x_offset < 0 || y_offset < 0 ||
      z_offset < 0)
    KC_FAIL(KC_ERR_INVALID, "point_step must be positive, sizes and offsets non-negative");
  const double two_pi = 2.0 * M_PI;
  const bool by_step = angle_step > 0.0;
  if (by_step) num_bins = static_cast<int>(std::ceil(two_pi / angle_step));  // :124
  if (num_bins <= 0) KC_FAIL(KC_ERR_INVALID, "no angular bins");
  if (static_cast<size_t>(num_bins) > cap)
    KC_FAIL(KC_ERR_RANGE, "%d bins do not fit the output capacity %zu", num_bins, cap);
  if (bins_out) *bins_out = static_cast<size_t>(num_bins);
  for (int i = 0; i < num_bins; ++i) {
    if (by_step && angles_out) angles_out[i] = i * angle_step;  // :127-132
    ranges_out[i] = max_range;
  }
  c->last_rebinned = 0;
  const long long per_row = (static_cast<long long>(row_step) + point_step - 1) / point_step;
  const long long n_rec = per_row * height;
  // nothing can be closer than a negative / NaN max_range: the reference leaves
  // every bin at max_range
  if (n_rec == 0 || nbytes == 0 || !(max_range >= 0.0)) return KC_OK;
  if (n_rec > 0x7FFFFFFFll) KC_FAIL(KC_ERR_RANGE, "more than 2^31 point records");
  KC_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  c->timing.begin_cycle();
  const uint8_t *dev = reinterpret_cast<const uint8_t *>(data);
  if (!data_on_device) {
    KC_TRY(c->d_data.reserve(nbytes));
    KC_HIP(hipMemcpyAsync(c->d_data.p, data, nbytes, hipMemcpyHostToDevice, s));
    dev = c->d_data.p;
  }
  KC_TRY(c->d_bins.reserve(num_bins));
  KC_TRY(c->h_bins.reserve(num_bins));
  KC_TRY(c->d_list.reserve(static_cast<size_t>(n_rec)));
  const bool in_lds = c->lds_ok && num_bins <= kCloudMaxLdsBins;
  const unsigned grid = static_cast<unsigned>(
      std::min<long long>(256, (n_rec + kCloudBlock - 1) / kCloudBlock));
  if (in_lds) KC_TRY(c->d_partial.reserve(static_cast<size_t>(grid) * num_bins));
  double max_r = max_range;
  CloudArgs a{};
  a.data = dev;
  a.nbytes = nbytes;
  a.n_records = n_rec;
  a.per_row = static_cast<int>(per_row);
  a.point_step = point_step;
  a.row_step = row_step;
  a.x_off = x_offset;
  a.y_off = y_offset;
  a.z_off = z_offset;
  a.max_off = std::max(std::max(x_offset, y_offset), z_offset);
  a.min_z = min_z;
  a.max_z = max_z;
  a.angle_step = by_step ? angle_step : 0.0;
  a.num_bins = num_bins;
  a.edge = 1e-6 / (by_step ? angle_step : two_pi / num_bins);
  std::memcpy(&a.max_bits, &max_r, sizeof(double));
  a.ftype = field_type;
  a.fsize = kFieldSize[field_type];
  a.packed16 = field_type == KC_FIELD_FLOAT32 && point_step == 16 && x_offset == 0 && y_offset == 4 && z_offset == 8 &&
               row_step % 16 == 0 && (reinterpret_cast<uintptr_t>(dev) & 15u) == 0;
  a.n_packed = std::min<long long>(n_rec, static_cast<long long>(nbytes / 16));
  a.bins = c->d_bins.p;
  a.partial = c->d_partial.p;
  unsigned int *const count_now = c->d_count.p + (c->calls & 1u);
  unsigned int *const count_next = c->d_count.p + ((c->calls + 1u) & 1u);
  ++c->calls;
  a.list_count = count_now;
  // the edge list goes straight to pinned host memory (a few hundred 8-byte
  // PCIe writes) when the polled hand-off is used
  if (in_lds) KC_TRY(c->h_list.reserve(static_cast<size_t>(n_rec)));
  a.list = in_lds ? c->h_list.p : c->d_list.p;
  if (!in_lds)  // global-atomic path: the bins start at max_range (and both counters at 0)
    hipLaunchKernelGGL(cloud_arm_kernel, dim3((num_bins + 255) / 256), dim3(256), 0, s,
                       c->d_bins.p, num_bins, max_range, c->d_count.p);
  KC_TRY(c->timing.start("cloud_bins_kernel", s));
  if (in_lds)
    hipLaunchKernelGGL(cloud_bins_kernel<true>, dim3(grid), dim3(kCloudBlock),
                       static_cast<size_t>(num_bins) * 4, s, a);
  else
    hipLaunchKernelGGL(cloud_bins_kernel<false>, dim3(grid), dim3(kCloudBlock), 0, s, a);
  KC_TRY(c->timing.stop(s));
  size_t nl = 0;
  if (in_lds) {
    KC_TRY(c->h_out.reserve(static_cast<size_t>(num_bins) + 2));
    const long long seq = ++c->seq;
    KC_TRY(c->timing.start("cloud_merge_kernel", s));
    hipLaunchKernelGGL(cloud_merge_kernel, dim3((num_bins + kMergeBins - 1) / kMergeBins), dim3(1024), 0, s,
                       reinterpret_cast<const uint32_t *>(c->d_partial.p), static_cast<int>(grid), num_bins, max_range,
                       c->d_bins.p, count_now, count_next, c->d_count.p + 2, c->h_out.p, seq);
    KC_TRY(c->timing.stop(s));
    KC_HIP(hipGetLastError());
    // poll the sequence word (bounded: then wait on the stream)
    volatile unsigned long long *hp = c->h_out.p;
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0; hp[0] != static_cast<unsigned long long>(seq); ++spins) {
      if ((spins & 1023) == 1023 &&
          std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
        KC_HIP(hipStreamSynchronize(s));
        break;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < num_bins; ++i) std::memcpy(&ranges_out[i], &c->h_out.p[2 + i], sizeof(double));
    nl = static_cast<size_t>(c->h_out.p[1]);
  } else {
    KC_HIP(hipGetLastError());
    KC_HIP(hipMemcpyAsync(c->h_bins.p, c->d_bins.p, num_bins * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost, s));
    KC_HIP(hipMemcpyAsync(c->h_count.p, count_now, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    KC_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < num_bins; ++i) std::memcpy(&ranges_out[i], &c->h_bins.p[i], sizeof(double));
    nl = c->h_count.p[0];
    if (nl) {
      KC_TRY(c->h_list.reserve(nl));
      KC_HIP(hipMemcpyAsync(c->h_list.p, c->d_list.p, nl * sizeof(float2), hipMemcpyDeviceToHost, s));
      KC_HIP(hipStreamSynchronize(s));
    }
  }
  c->last_rebinned = nl;
  // the edge cases: the reference expression on the host (pointcloud.h:148-175)
  for (size_t k = 0; k < nl; ++k) {
    const float x = c->h_list.p[k].x, y = c->h_list.p[k].y;
    const float xx = x * x, yy = y * y;
    const float range_sq = xx + yy;
    double angle = static_cast<double>(::atan2f(y, x));  // std::atan2(float, float)
    if (angle < 0.0) angle += two_pi;
    int bin = by_step ? static_cast<int>(angle / angle_step)
                      : static_cast<int>((angle / two_pi) * num_bins);
    bin = std::min(bin, num_bins - 1);
    const double distance = static_cast<double>(::sqrtf(range_sq));
    if (distance < ranges_out[bin]) ranges_out[bin] = distance;
  }
  return KC_OK;
}

int kc_cloud_last_rebinned(kc_cloud *c, size_t *count) {
  if (!c || !count) KC_FAIL(KC_ERR_INVALID, "null argument");
  *count = c->last_rebinned;
  return KC_OK;
}

int kc_cloud_timing_enable(kc_cloud *c, int enable) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  c->timing.enabled = enable != 0;
  return KC_OK;
}

int kc_cloud_timing_get(kc_cloud *c, const char **names, float *ms, size_t cap,
                        size_t *count) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  KC_HIP(hipSetDevice(c->device));
  return c->timing.get(names, ms, cap, count);
}

}  // extern "C"
