// Error reporting, device query, HIP-event helpers and small layout kernels of libpartner_hip.
#include "pn_common.h"
#include <algorithm>
#include <cstring>

namespace pn {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

ProfileSlot* profile_slot() {
  static thread_local ProfileSlot slot = {nullptr, nullptr};
  return &slot;
}

namespace {
__global__ void zero_fill_kernel(uint4* __restrict__ p16, size_t n16, unsigned char* __restrict__ tail, int ntail) {
  const uint4 z = {0u, 0u, 0u, 0u};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p16[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
}  // namespace

int zero_async(void* ptr, size_t bytes, hipStream_t st) {
  if (bytes == 0) return PN_OK;
  unsigned char* b = static_cast<unsigned char*>(ptr);
  // head bytes up to 16-byte alignment are handled as a tail of a first tiny launch
  const size_t mis = (16 - (reinterpret_cast<uintptr_t>(b) & 15)) & 15;
  const size_t head = mis < bytes ? mis : bytes;
  if (head) hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(64), 0, st, (uint4*)nullptr, (size_t)0, b, (int)head);
  b += head;
  bytes -= head;
  const size_t n16 = bytes / 16;
  const int ntail = (int)(bytes - n16 * 16);
  if (n16 || ntail) {
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>(4096, (n16 + 255) / 256));
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<uint4*>(b), n16, b + n16 * 16, ntail);
  }
  return check_launch("zero_fill_kernel");
}

}  // namespace pn

// ---- opaque per-device handle: device properties + the one loud check that the code objects match the GPU -------
struct pn_handle_s {
  int device;
  int compute_units;
  int lds_bytes;
  size_t hbm_bytes;
  char arch[64];
};

extern "C" int pn_handle_create(int device, pn_handle_t* out) {
  PN_REQUIRE(out != nullptr, "handle_create: null output");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return pn::fail(PN_ERR_INVALID, "handle_create: no HIP device %d (%d visible)", device, n);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return pn::fail(PN_ERR_LAUNCH, "handle_create: hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return pn::fail(PN_ERR_INVALID, "handle_create: device %d is %s; libpartner_hip holds gfx950 (MI355X) code objects only", device, prop.gcnArchName);
  pn_handle_s* h = new pn_handle_s();
  h->device = device;
  h->compute_units = prop.multiProcessorCount;
  h->lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
  h->hbm_bytes = prop.totalGlobalMem;
  snprintf(h->arch, sizeof(h->arch), "%s", prop.gcnArchName);
  *out = h;
  return PN_OK;
}

extern "C" int pn_handle_destroy(pn_handle_t h) {
  delete h;
  return PN_OK;
}

extern "C" int pn_handle_info(pn_handle_t h, int* device, int* compute_units, int* lds_bytes_per_cu, unsigned long long* hbm_bytes, char* arch,
                              size_t arch_len) {
  PN_REQUIRE(h != nullptr, "handle_info: null handle");
  if (device) *device = h->device;
  if (compute_units) *compute_units = h->compute_units;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = h->lds_bytes;
  if (hbm_bytes) *hbm_bytes = (unsigned long long)h->hbm_bytes;
  if (arch && arch_len) snprintf(arch, arch_len, "%s", h->arch);
  return PN_OK;
}

extern "C" int pn_handle_pci_bus_id(pn_handle_t h, char* buf, size_t buf_len) {
  PN_REQUIRE(h != nullptr && buf && buf_len >= 13, "handle_pci_bus_id: null handle or a buffer under 13 bytes");
  hipError_t rc = hipDeviceGetPCIBusId(buf, (int)buf_len, h->device);
  if (rc != hipSuccess) return pn::fail(PN_ERR_LAUNCH, "hipDeviceGetPCIBusId: %s", hipGetErrorString(rc));
  return PN_OK;
}

namespace {

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, int c, int hw, float* __restrict__ out, size_t total) {
  // out index order (b, p, c); 32x32 LDS transpose tiles keep both sides coalesced
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int cc = c0 + i, pp = p0 + threadIdx.x;
    tile[i][threadIdx.x] = (cc < c && pp < hw) ? in[((size_t)b * c + cc) * hw + pp] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int pp = p0 + i, cc = c0 + threadIdx.x;
    if (pp < hw && cc < c) out[((size_t)b * hw + pp) * c + cc] = tile[threadIdx.x][i];
  }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, int c, int hw, int ps, int co, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int pp = p0 + i, cc = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (cc < c && pp < hw) ? in[((size_t)b * hw + pp) * ps + co + cc] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int cc = c0 + i, pp = p0 + threadIdx.x;
    if (pp < hw && cc < c) out[((size_t)b * c + cc) * hw + pp] = tile[threadIdx.x][i];
  }
}

// (B,H,W,C) -> (B,W,H,C): swap the two spatial axes, channel vectors stay contiguous
__global__ void transpose_hw_kernel(const float* __restrict__ in, int h, int w, int c, float* __restrict__ out, size_t total4) {
  const int c4 = c / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int k = (int)(r % c4); r /= c4;
    const int y = (int)(r % h); r /= h;   // output is (b, x, y, c): y fastest after c
    const int x = (int)(r % w);
    const size_t b = r / w;
    const float4 v = *reinterpret_cast<const float4*>(in + (((b * h + y) * w + x) * (size_t)c) + k * 4);
    *reinterpret_cast<float4*>(out + i * 4) = v;
  }
}

}  // namespace

extern "C" {

int pn_transpose_hw_f32(const float* in, int b, int h, int w, int c, float* out, pn_stream_t stream) {
  PN_REQUIRE(in && out && b > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "transpose_hw: bad arguments (c must be a multiple of 4)");
  const size_t total4 = (size_t)b * h * w * (c / 4);
  hipLaunchKernelGGL(transpose_hw_kernel, dim3((unsigned)std::min<size_t>(8192, (total4 + 255) / 256)), dim3(256), 0, pn::S(stream),
                     in, h, w, c, out, total4);
  return pn::check_launch("transpose_hw_kernel");
}

int pn_version(void) { return 100; }

int pn_last_error(char* buf, size_t buf_len) {
  const char* e = pn::err_buf();
  const size_t n = strlen(e);
  if (buf && buf_len) {
    const size_t k = n < buf_len - 1 ? n : buf_len - 1;
    memcpy(buf, e, k);
    buf[k] = 0;
  }
  return (int)n;
}

int pn_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

int pn_fill_zero(void* ptr, size_t bytes, pn_stream_t stream) {
  PN_REQUIRE(ptr || bytes == 0, "fill_zero: null pointer");
  if (bytes == 0) return PN_OK;
  return pn::zero_async(ptr, bytes, pn::S(stream));
}

int pn_nchw_to_nhwc_f32(const float* in, int b, int c, int h, int w, float* out, pn_stream_t stream) {
  PN_REQUIRE(in && out && b > 0 && c > 0 && h > 0 && w > 0, "nchw_to_nhwc: bad arguments");
  const int hw = h * w;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(pn::cdiv(hw, 32), pn::cdiv(c, 32), b), dim3(32, 8), 0, pn::S(stream), in,
                     c, hw, out, (size_t)b * c * hw);
  return pn::check_launch("nchw_to_nhwc_kernel");
}

int pn_nhwc_to_nchw_f32(const float* in, int b, int c, int h, int w, int pixel_stride, int channel_offset, float* out,
                        pn_stream_t stream) {
  PN_REQUIRE(in && out && b > 0 && c > 0 && h > 0 && w > 0 && pixel_stride >= c, "nhwc_to_nchw: bad arguments");
  const int hw = h * w;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(pn::cdiv(hw, 32), pn::cdiv(c, 32), b), dim3(32, 8), 0, pn::S(stream), in,
                     c, hw, pixel_stride, channel_offset, out);
  return pn::check_launch("nhwc_to_nchw_kernel");
}

int pn_event_create(pn_event_t* ev) {
  PN_REQUIRE(ev, "event_create: null");
  hipEvent_t e;
  hipError_t rc = hipEventCreate(&e);
  if (rc != hipSuccess) return pn::fail(PN_ERR_LAUNCH, "hipEventCreate: %s", hipGetErrorString(rc));
  *ev = e;
  return PN_OK;
}
int pn_event_destroy(pn_event_t ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
  return PN_OK;
}
int pn_event_record(pn_event_t ev, pn_stream_t stream) {
  hipError_t rc = hipEventRecord((hipEvent_t)ev, pn::S(stream));
  if (rc != hipSuccess) return pn::fail(PN_ERR_LAUNCH, "hipEventRecord: %s", hipGetErrorString(rc));
  return PN_OK;
}
int pn_event_elapsed_ms(pn_event_t start, pn_event_t stop, float* ms) {
  PN_REQUIRE(ms, "event_elapsed: null");
  hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
  if (rc == hipSuccess) rc = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
  if (rc != hipSuccess) return pn::fail(PN_ERR_LAUNCH, "hipEventElapsedTime: %s", hipGetErrorString(rc));
  return PN_OK;
}

int pn_profile_next_launch(pn_event_t start, pn_event_t stop) {
  PN_REQUIRE((start && stop) || (!start && !stop), "profile_next_launch: both events or none");
  pn::ProfileSlot* s = pn::profile_slot();
  s->start = (hipEvent_t)start;
  s->stop = (hipEvent_t)stop;
  return PN_OK;
}

}  // extern "C"
