// Shared helpers for the mmEgo HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MMEGO_OK 0
#define MMEGO_EBADARG (-1)

// Launch check: returns the HIP error code (positive) from the enclosing extern "C" function.
#define MMEGO_LAUNCH_CHECK()                        \
  do {                                              \
    hipError_t e_ = hipGetLastError();              \
    if (e_ != hipSuccess) return (int)e_;           \
  } while (0)

#define MMEGO_REQUIRE(cond) \
  do {                      \
    if (!(cond)) return MMEGO_EBADARG; \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// 64-lane wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Counter-based dropout RNG: element i of a call keeps its value when a 24-bit hash of (i, call key) is >= p.  The call key
// mixes the device-side seed counter (advanced once per training forward by mmego_inc_i64) with a per-call-site salt.
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned dropout_key(unsigned long long seed, unsigned salt) {
  seed = (seed + (unsigned long long)salt * 0x9E3779B97F4A7C15ULL) * 6364136223846793005ULL + 1442695040888963407ULL;
  return hash32((unsigned)(seed & 0xffffffffU) + 0x9e3779b9U * (unsigned)(seed >> 32));
}
__device__ __forceinline__ bool dropout_keep(unsigned key, unsigned i, float p) {
  return (hash32(i ^ key) >> 8) * (1.0f / 16777216.0f) >= p;
}

// In-kernel clock stamps: compiled ONLY into the diagnostic probe (scripts/clock_probe.hip defines MMEGO_STAMP); the
// product library contains no stamp.  Slot s of workgroup b holds {s_memtime, s_memrealtime} taken by one lane.
#ifdef MMEGO_STAMP
#define MMEGO_STAMP_SLOTS 4
__device__ unsigned long long mmego_stamp_buf[8192 * MMEGO_STAMP_SLOTS * 2];
__device__ unsigned int mmego_stamp_hw[8192 * 2];   // {HW_ID, XCC_ID} of the stamping wave (where the workgroup ran)
#define MMEGO_STAMP_AT(id, slot, cond)                                                                 \
  do {                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    if (cond) {                                                                                      \
      unsigned long long t_ = __builtin_amdgcn_s_memtime(), r_ = __builtin_amdgcn_s_memrealtime();   \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                                            \
      mmego_stamp_buf[(((id) & 8191) * MMEGO_STAMP_SLOTS + (slot)) * 2] = t_;                  \
      mmego_stamp_buf[(((id) & 8191) * MMEGO_STAMP_SLOTS + (slot)) * 2 + 1] = r_;              \
      if ((slot) == 0) {                                                                             \
        mmego_stamp_hw[((id) & 8191) * 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);         \
        mmego_stamp_hw[((id) & 8191) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    \
      }                                                                                              \
    }                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  } while (0)
#else
#define MMEGO_STAMP_AT(id, slot, cond) do { } while (0)
#endif
