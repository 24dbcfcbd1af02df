// Microbenchmark: cost of a workgroup barrier / LDS round trip / FP64 chain on gfx950 for a 1- and 4-wavefront
// workgroup alone on its CU (the IK kernel's latency shape).  hipcc --offload-arch=gfx950 -O3 barrier_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(unsigned long long* out, double* sink, int iters) {
  __shared__ double buf[1024];
  const int t = threadIdx.x;
  buf[t] = t;
  double x = 1.0 + t * 1e-9, y = 0.5, z = 0.25 + t * 1e-9, w = 0.125 + t * 1e-9, v = 0.3 + t * 1e-9;
  float f = 0.5f + t * 1e-6f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) { __syncthreads(); }
    if (MODE == 1) { buf[t] = x; __syncthreads(); x += buf[(t + 64) & (blockDim.x - 1)]; }
    if (MODE == 2) { buf[t] = x; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); x += buf[(t + 1) & 63]; }
    if (MODE == 3) {                                                         // 16 dependent FP64 FMAs
#pragma unroll
      for (int u = 0; u < 16; u++) x = fma(x, y, 1.0);
    }
    if (MODE == 4) {                                                         // 2 x 16 FMAs, two independent chains
#pragma unroll
      for (int u = 0; u < 16; u++) { x = fma(x, y, 1.0); z = fma(z, 0.999, 0.001); }
    }
    if (MODE == 8) {                                                         // 4 x 16 FMAs, four independent chains
#pragma unroll
      for (int u = 0; u < 16; u++) { x = fma(x, y, 1.0); z = fma(z, 0.999, 0.001); w = fma(w, 0.998, 0.002); v = fma(v, 0.997, 0.003); }
    }
    if (MODE == 9) {                                                         // 16 dependent FP32 FMAs
#pragma unroll
      for (int u = 0; u < 16; u++) f = fmaf(f, 0.999f, 0.001f);
    }
    if (MODE == 5) { x = sin(x); }                                           // dependent libm sin
    if (MODE == 6) { x = 1.0 / sqrt(x + 2.0); }                              // rsqrt via sqrt+div
    if (MODE == 7) { x = __shfl(x, (t + 1) & 63, 64); }                      // ds_bpermute round trip
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + t] = x + y + z + w + v + f;
}
int main() {
  unsigned long long* d; double* s;
  hipMalloc(&d, 64 * 8); hipMalloc(&s, 64 * 256 * 8);
  const char* names[] = {"barrier", "lds write + barrier + read", "lds write + fence + read (1 wave)", "16 dependent f64 fma",
                         "32 f64 fma, two independent chains", "dependent f64 sin", "1/sqrt f64", "ds_bpermute shuffle",
                         "64 f64 fma, four independent chains", "16 dependent f32 fma"};
  const int iters = 2000;
  for (int nw : {1, 4}) {
    for (int m = 0; m < 10; m++) {
      if (m == 2 && nw != 1) continue;
      switch (m) {
        case 0: hipLaunchKernelGGL(k<0>, 1, 64 * nw, 0, 0, d, s, iters); break;
        case 1: hipLaunchKernelGGL(k<1>, 1, 64 * nw, 0, 0, d, s, iters); break;
        case 2: hipLaunchKernelGGL(k<2>, 1, 64 * nw, 0, 0, d, s, iters); break;
        case 3: hipLaunchKernelGGL(k<3>, 1, 64 * nw, 0, 0, d, s, iters); break;
        case 4: hipLaunchKernelGGL(k<4>, 1, 64 * nw, 0, 0, d, s, iters); break;
        case 5: hipLaunchKernelGGL(k<5>, 1, 64 * nw, 0, 0, d, s, iters); break;
        case 6: hipLaunchKernelGGL(k<6>, 1, 64 * nw, 0, 0, d, s, iters); break;
        case 7: hipLaunchKernelGGL(k<7>, 1, 64 * nw, 0, 0, d, s, iters); break;
        case 8: hipLaunchKernelGGL(k<8>, 1, 64 * nw, 0, 0, d, s, iters); break;
        case 9: hipLaunchKernelGGL(k<9>, 1, 64 * nw, 0, 0, d, s, iters); break;
      }
      unsigned long long h = 0;
      hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      printf("waves=%d  %-36s %8.1f cycles/iter\n", nw, names[m], (double)h / iters);
    }
  }
  return 0;
}
