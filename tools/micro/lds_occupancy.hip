// How many single-wavefront blocks with a given dynamic LDS size are resident per CU on this device: the occupancy API's
// answer, and a measurement (k blocks per CU of a fixed-latency kernel: the time doubles once they no longer fit).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lds_occupancy.hip -o /tmp/lds_occ && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(64) void spin(double* out, int iters) {
  extern __shared__ double sm[];
  double x = threadIdx.x * 1e-3 + 1.0;
  for (int i = 0; i < iters; i++) x = fma(x, 0.999999, 1e-9);
  sm[threadIdx.x] = x;
  if (x == 12345.0) out[blockIdx.x] = sm[(threadIdx.x + 1) & 63];
}

int main() {
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  double* d = nullptr;
  hipMalloc((void**)&d, 1 << 20);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  const int sizes[] = {13312, 13653, 14336, 14848, 15360, 16384, 16640, 17920, 18176, 18204, 18432, 18464, 20480};
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int lds : sizes) {
    int nblk = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)spin, 64, lds);
    printf("lds %6d  api %2d  ms(k blocks/CU):", lds, nblk);
    for (int k = 6; k <= 13; k++) {
      hipLaunchKernelGGL(spin, dim3(ncu * k), dim3(64), lds, 0, d, 200000);
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(spin, dim3(ncu * k), dim3(64), lds, 0, d, 200000);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      printf(" %d:%.2f", k, ms);
    }
    printf("\n");
  }
  return 0;
}
