// accuracy of v_rsq_f64 + one / two Newton steps against 1/sqrt(x) in long double (host)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, double* y3, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i], y = __builtin_amdgcn_rsq(v), h = 0.5 * v;
  y0[i] = y;
  y = y * fma(-h * y, y, 1.5); y1[i] = y;
  y = y * fma(-h * y, y, 1.5); y2[i] = y;
  {  // one third-order step from the seed: e = 1 - x y^2, y (1 + e/2 + 3 e^2 / 8)
    double z = y0[i], e = fma(-v, z * z, 1.0);
    y3[i] = fma(z * e, fma(0.375, e, 0.5), z);
  }
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n), d(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (s >> 11) * 0x1.0p-53, (int)(s % 40) - 20); }
  double *dx, *d0, *d1, *d2, *d3;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), d3, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  for (int i = 0; i < n; i++) {
    long double r = 1.0L / sqrtl((long double)x[i]);
    e0 = std::fmax(e0, (double)fabsl((a[i] - r) / r)); e1 = std::fmax(e1, (double)fabsl((b[i] - r) / r)); e2 = std::fmax(e2, (double)fabsl((c[i] - r) / r)); e3 = std::fmax(e3, (double)fabsl((d[i] - r) / r));
  }
  printf("max relative error: seed %.3e, one Newton step %.3e, two steps %.3e, one third-order step %.3e (2^-53 = 1.11e-16)\n", e0, e1, e2, e3);
  return 0;
}
