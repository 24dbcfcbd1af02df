// math_check.hip -- lean device math (gmr_device_math.h: se3_log_rel5, se3_jlinv_aux5, sincos_small, fast_rcp) against
// the closed-form versions and host libm on random inputs.   hipcc -O3 --offload-arch=gfx950 -o math_check math_check.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../general_motion_retargeting_amd/csrc/gmr_device_math.h"
using namespace gmr;

__global__ void k_log(int n, const double* in, double* oa, double* ob, double* ja, double* jb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* p = in + 14 * i;
  double e[6], aux[3], e5[6], aux5[5];
  se3_log_rel(d3{p[0], p[1], p[2]}, d4{p[3], p[4], p[5], p[6]}, d3{p[7], p[8], p[9]}, d4{p[10], p[11], p[12], p[13]}, e, aux);
  se3_log_rel5(d3{p[0], p[1], p[2]}, d4{p[3], p[4], p[5], p[6]}, d3{p[7], p[8], p[9]}, d4{p[10], p[11], p[12], p[13]}, e5, aux5);
  for (int r = 0; r < 6; r++) { oa[9 * i + r] = e[r]; ob[9 * i + r] = e5[r]; }
  for (int r = 0; r < 3; r++) { oa[9 * i + 6 + r] = aux[r]; ob[9 * i + 6 + r] = aux5[r]; }
  m3 A, B, A5, B5;
  se3_jlinv_aux(e, aux, A, B);
  se3_jlinv_aux5(e5, aux5, A5, B5);
  for (int r = 0; r < 9; r++) { ja[18 * i + r] = A.a[r]; ja[18 * i + 9 + r] = B.a[r]; jb[18 * i + r] = A5.a[r]; jb[18 * i + 9 + r] = B5.a[r]; }
}
__global__ void k_sc(int n, const double* in, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  sincos_small(in[i], &s, &c);
  out[3 * i] = s; out[3 * i + 1] = c; out[3 * i + 2] = fast_rcp(1.0 + fabs(in[i]));
}
__global__ void k_at(int n, const double* in, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = atan2_q1(in[2 * i], in[2 * i + 1]);
}

static double urand() { return rand() / (double)RAND_MAX; }

int main() {
  const int n = 1 << 18;
  std::vector<double> in(14 * (size_t)n);
  srand(1);
  for (int i = 0; i < n; i++) {
    double* p = &in[14 * (size_t)i];
    // relative rotation angle from 1e-9 to pi, log-uniform in half of the cases
    double ang = (i & 1) ? exp(log(1e-9) + urand() * (log(3.14159) - log(1e-9))) : urand() * 3.14159265;
    if (i % 97 == 0) ang = 3.14159265358979 - 1e-7 * urand();
    double ax[3] = {urand() - 0.5, urand() - 0.5, urand() - 0.5};
    double an = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    double qb[4] = {urand() - 0.5, urand() - 0.5, urand() - 0.5, urand() - 0.5};
    double qn = sqrt(qb[0] * qb[0] + qb[1] * qb[1] + qb[2] * qb[2] + qb[3] * qb[3]);
    for (int k = 0; k < 4; k++) qb[k] /= qn;
    double qr[4] = {cos(ang / 2), sin(ang / 2) * ax[0] / an, sin(ang / 2) * ax[1] / an, sin(ang / 2) * ax[2] / an};
    if (i % 5 == 0) for (int k = 0; k < 4; k++) qr[k] = -qr[k];   // the other hemisphere
    // qt = qb * qr
    double qt[4] = {qb[0] * qr[0] - qb[1] * qr[1] - qb[2] * qr[2] - qb[3] * qr[3], qb[0] * qr[1] + qb[1] * qr[0] + qb[2] * qr[3] - qb[3] * qr[2],
                    qb[0] * qr[2] - qb[1] * qr[3] + qb[2] * qr[0] + qb[3] * qr[1], qb[0] * qr[3] + qb[1] * qr[2] - qb[2] * qr[1] + qb[3] * qr[0]};
    for (int k = 0; k < 3; k++) { p[k] = urand() - 0.5; p[7 + k] = p[k] + (urand() - 0.5) * ((i & 2) ? 1.0 : 1e-3); }
    for (int k = 0; k < 4; k++) { p[3 + k] = qb[k]; p[10 + k] = qt[k]; }
  }
  double *d_in, *d_a, *d_b, *d_ja, *d_jb;
  hipMalloc(&d_in, in.size() * 8); hipMalloc(&d_a, 9 * (size_t)n * 8); hipMalloc(&d_b, 9 * (size_t)n * 8);
  hipMalloc(&d_ja, 18 * (size_t)n * 8); hipMalloc(&d_jb, 18 * (size_t)n * 8);
  hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice);
  k_log<<<n / 256, 256>>>(n, d_in, d_a, d_b, d_ja, d_jb);
  std::vector<double> a(9 * (size_t)n), b(9 * (size_t)n), ja(18 * (size_t)n), jb(18 * (size_t)n);
  hipMemcpy(a.data(), d_a, a.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d_b, b.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(ja.data(), d_ja, ja.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(jb.data(), d_jb, jb.size() * 8, hipMemcpyDeviceToHost);
  double de = 0, da = 0, dj = 0; int nan = 0;
  for (int i = 0; i < n; i++) {
    for (int r = 0; r < 6; r++) { double d = fabs(a[9 * i + r] - b[9 * i + r]); if (!(d == d)) nan++; if (d > de) de = d; }
    { double d = fabs(a[9 * i + 6] - b[9 * i + 6]); if (d > da) da = d; }
    for (int r = 0; r < 18; r++) { double d = fabs(ja[18 * i + r] - jb[18 * i + r]); if (!(d == d)) nan++; if (d > dj) dj = d; }
  }
  printf("se3_log_rel5 vs se3_log_rel: max |de| = %.3e, max |da| = %.3e; se3_jlinv_aux5 vs se3_jlinv_aux: max |dJ| = %.3e; nan = %d (n = %d)\n", de, da, dj, nan, n);
  // sincos_small vs host libm
  std::vector<double> x(n), o(3 * (size_t)n);
  for (int i = 0; i < n; i++) x[i] = (i & 1) ? (urand() - 0.5) * 20.0 : (urand() - 0.5) * 600.0;
  x[0] = 0.0; x[1] = 1.5707963267948966; x[2] = -3.141592653589793; x[3] = 1e-300; x[4] = 0.7853981633974483;
  hipMemcpy(d_in, x.data(), n * 8, hipMemcpyHostToDevice);
  k_sc<<<n / 256, 256>>>(n, d_in, d_a);
  hipMemcpy(o.data(), d_a, o.size() * 8, hipMemcpyDeviceToHost);
  double ds = 0, dc = 0, dr = 0;
  for (int i = 0; i < n; i++) {
    ds = fmax(ds, fabs(o[3 * i] - sin(x[i]))); dc = fmax(dc, fabs(o[3 * i + 1] - cos(x[i])));
    dr = fmax(dr, fabs(o[3 * i + 2] * (1.0 + fabs(x[i])) - 1.0));
  }
  printf("sincos_small vs libm on [-300, 300]: max |ds| = %.3e, max |dc| = %.3e; fast_rcp max rel err = %.3e\n", ds, dc, dr);
  // atan2_q1 vs host libm on the first quadrant: unit vectors (sin, cos of a half angle), tiny and huge ratios
  std::vector<double> yx(2 * (size_t)n), at(n);
  for (int i = 0; i < n; i++) {
    double h = (i & 1) ? urand() * 1.5707963267948966 : exp(log(1e-12) + urand() * (log(1.5707963) - log(1e-12)));
    if (i % 101 == 0) h = 1.5707963267948966 - 1e-9 * urand();
    yx[2 * i] = sin(h); yx[2 * i + 1] = cos(h);
    if (i % 7 == 0) { yx[2 * i] *= 0.37; yx[2 * i + 1] *= 0.37; }      // the scale must not matter
  }
  yx[0] = 0.0; yx[1] = 1.0; yx[2] = 1.0; yx[3] = 0.0; yx[4] = 1.0; yx[5] = 1.0;
  hipMemcpy(d_in, yx.data(), 2 * (size_t)n * 8, hipMemcpyHostToDevice);
  k_at<<<n / 256, 256>>>(n, d_in, d_a);
  hipMemcpy(at.data(), d_a, n * 8, hipMemcpyDeviceToHost);
  double dat = 0;
  for (int i = 0; i < n; i++) { double ref = atan2(yx[2 * i], yx[2 * i + 1]); dat = fmax(dat, fabs(at[i] - ref) / fmax(ref, 1e-300)); }
  printf("atan2_q1 vs libm: max relative error = %.3e\n", dat);
  return (de < 1e-12 && dj < 1e-10 && ds < 4e-16 && dc < 4e-16 && dat < 1e-15 && nan == 0) ? 0 : 1;
}
