// stream_bw.hip -- what HBM gives a plain streaming kernel on this box: write-only, read-only and copy, 16 B per lane,
// grid-stride, sizes of the FK kernel's output (2^20 frames x 456 B = 478 MB).  The ceilings the FK kernel's 0.29-0.34
// of "8 TB/s" should be read against.   hipcc -O3 --offload-arch=gfx950 stream_bw.hip -o stream_bw && ./stream_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_write(float4* __restrict__ d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_read(const float4* __restrict__ s, size_t n, float* out) {
  float a = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = s[i];
    a += v.x + v.y + v.z + v.w;
  }
  if (a == 12345.678f) out[0] = a;
}
__global__ void k_copy(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
int main() {
  const size_t bytes = (size_t)(1 << 20) * 456, n = bytes / 16;
  float4 *a, *b; float* o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {1024, 4096, 16384}) {
    for (int which = 0; which < 4; which++) {
      float best = 1e9f;
      for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, a, n);
        if (which == 1) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, o);
        if (which == 2) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n);
        if (which == 3) hipMemsetAsync(a, 0, bytes, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const char* nm[] = {"write", "read", "copy(r+w)", "hipMemset"};
      printf("blocks %5d %-10s %.3f ms  %.2f TB/s\n", blocks, nm[which], best, (which == 2 ? 2.0 : 1.0) * bytes / best * 1e-9);
    }
  }
  return 0;
}
