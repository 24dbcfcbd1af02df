// How many 64-lane workgroups fit a CU for a given dynamic LDS size (allocation granularity of gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(float* o) { extern __shared__ float s[]; s[threadIdx.x] = 1; o[threadIdx.x] = s[63 - threadIdx.x]; }
int main() {
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
  int prev = -1;
  for (int bytes = 16384; bytes <= 160 * 1024 - 1024; bytes += 256) {
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, bytes);
    if (n != prev) { printf("dynamic LDS %6d B -> %d workgroups per CU\n", bytes, n); prev = n; }
  }
  return 0;
}
