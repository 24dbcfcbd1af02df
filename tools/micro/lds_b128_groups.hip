// Which lanes of a wavefront does the LDS serve together for ds_read_b128 / ds_read_b64?  Pair test: lanes i and j read the
// SAME four banks at different addresses, every other lane reads other banks -- the access costs an extra pass exactly when
// i and j are served in the same pass.  Prints, for lane 0 .. 63, the set of lanes it conflicts with (the H schedule's
// conflict model, gmr_ik_wide_layout.h, assumes groups of 16 consecutive lanes for b128).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lds_b128_groups.hip -o tools/micro/lds_groups && tools/micro/lds_groups
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

template <int W>   // 16: b128, 8: b64
__global__ __launch_bounds__(64) void probe(const int* __restrict__ addr, long long* __restrict__ cycles, int npat, int iters) {
  __shared__ __attribute__((aligned(16))) double sm[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) sm[i] = i;
  __syncthreads();
  for (int p = 0; p < npat; p++) {
    const unsigned a = (unsigned)addr[p * 64 + threadIdx.x] + (unsigned)(size_t)sm;
    double acc = 0.0;
    __builtin_amdgcn_s_waitcnt(0);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {      // eight reads in flight, one wait: the LDS pipe's throughput, not its latency
      typedef double d2 __attribute__((ext_vector_type(2)));
      if (W == 16) {
        d2 x0, x1, x2, x3, x4, x5, x6, x7;
        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8\n ds_read_b128 %2, %8\n ds_read_b128 %3, %8\n"
                     "ds_read_b128 %4, %8\n ds_read_b128 %5, %8\n ds_read_b128 %6, %8\n ds_read_b128 %7, %8\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7) : "v"(a) : "memory");
        acc += x0.x + x7.y;
      } else {
        double x0, x1, x2, x3, x4, x5, x6, x7;
        asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8\n"
                     "ds_read_b64 %4, %8\n ds_read_b64 %5, %8\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7) : "v"(a) : "memory");
        acc += x0 + x7;
      }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[p] = t1 - t0;
    if (acc == 12345.678) cycles[p] = 0;
  }
}

template <int W>
static void run(const char* name) {
  const int npat = 64 * 64 + 1, iters = 200;
  std::vector<int> h((size_t)npat * 64);
  auto other = [](int l) { return W * (1 + (l % 15)); };          // bank groups 1 .. 15, one address per group (broadcast)
  for (int i = 0; i < 64; i++)
    for (int j = 0; j < 64; j++) {
      int* a = &h[(size_t)(i * 64 + j) * 64];
      for (int l = 0; l < 64; l++) a[l] = other(l);
      a[i] = 0;                                                   // bank group 0, address 0
      if (j != i) a[j] = 256;                                     // the same banks again (64 banks x 4 B further), another address
    }
  for (int l = 0; l < 64; l++) h[(size_t)(npat - 1) * 64 + l] = other(l);
  int* d_a; long long* d_c;
  hipMalloc((void**)&d_a, h.size() * 4); hipMalloc((void**)&d_c, npat * 8);
  hipMemcpy(d_a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe<W>, dim3(1), dim3(64), 0, 0, d_a, d_c, npat, iters);
  hipDeviceSynchronize();
  std::vector<long long> c(npat);
  hipMemcpy(c.data(), d_c, npat * 8, hipMemcpyDeviceToHost);
  const long long base = c[npat - 1];
  printf("%s: baseline %lld cycles per %d x 8 reads; lane 0 with 1 / 16 / 32 / 48: %lld %lld %lld %lld\n", name, base, iters, c[1], c[16], c[32], c[48]);
  for (int i = 0; i < 64; i++) {
    printf("lane %2d conflicts with:", i);
    for (int j = 0; j < 64; j++)
      if (j != i && c[i * 64 + j] > base + base / 64) printf(" %d", j);
    printf("\n");
  }
}

int main() {
  run<16>("ds_read_b128");
  run<8>("ds_read_b64");
  return 0;
}
