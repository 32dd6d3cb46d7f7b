// Shared helpers for libpartner_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdint>

#include "../../include/partner_hip.h"

namespace pn {

// per-thread last error text
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return PN_OK;
}

// one-shot per-thread timing slot armed by pn_profile_next_launch(): the launcher that finds it attaches the events to
// its dispatch (hipExtLaunchKernelGGL) and clears it
struct ProfileSlot { hipEvent_t start, stop; };
ProfileSlot* profile_slot();
inline bool take_profile_slot(ProfileSlot& out) {
  ProfileSlot* s = profile_slot();
  if (!s->start) return false;
  out = *s;
  s->start = s->stop = nullptr;
  return true;
}

// hipFuncSetAttribute belongs to the CURRENT device's copy of the kernel: true the first time the calling site runs on a device
// (one-shot statics would leave the second GPU of a process with the default dynamic-LDS limit).  `flags` = a static array of 64.
inline bool first_use_on_device(bool* flags) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  if (flags[dev]) return false;
  flags[dev] = true;
  return true;
}

inline hipStream_t S(pn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

constexpr int kWave = 64;

// Zero-fill as an ordinary kernel (16-byte stores).  Used instead of hipMemsetAsync everywhere
// in the library: memset *nodes* of a captured hipGraph were observed (ROCm 7.2) not to be
// re-executed reliably between replays, kernel nodes always are.
int zero_async(void* ptr, size_t bytes, hipStream_t st);

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide exclusive scan of one value per thread (THREADS = multiple of 64): returns the exclusive prefix and leaves
// the block total in *total.  Wave-level shuffles + one LDS hop; fixed association order.  Shared by the bitmap
// unique-rank (voxelize.hip), the sparse-convolution site index (sparse_conv.hip) and the sweep compaction (assign.hip).
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[THREADS / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < THREADS / 64; ++k) {
    if (k < w) base += wsum[k];
    tot += wsum[k];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == PN_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == PN_ACT_TANH) return tanhf(v);
  if (act == PN_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));  // exact (erf) GELU
  return v;
}

}  // namespace pn

#define PN_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) return pn::fail(PN_ERR_INVALID, __VA_ARGS__); \
  } while (0)
