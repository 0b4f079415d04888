// Test aid: fill the LDS of every CU with a bit pattern (a kernel that reads LDS it never wrote then sees NaNs instead of whatever the
// previous workgroup on its CU left behind).  scripts/find_lds_uninit.py launches it in front of chosen kernels of a stage step.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned pattern, unsigned* sink) {
  extern __shared__ unsigned lds[];
  const int n = 160 * 1024 / 4;
  for (int i = threadIdx.x; i < n; i += 256) lds[i] = pattern;
  __syncthreads();
  if (sink && lds[(threadIdx.x * 97) % n] == 0x12345678u) sink[0] = 1;      // (keep the stores)
}
extern "C" int lds_poison(void* stream, unsigned pattern, unsigned* sink) {
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute((const void*)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
    set = true;
  }
  lds_poison_kernel<<<1024, 256, 160 * 1024, (hipStream_t)stream>>>(pattern, sink);
  return (int)hipGetLastError();
}
