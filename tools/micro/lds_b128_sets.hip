// Companion of lds_b128_groups.hip: cost of an n-way bank conflict among a chosen SET of lanes for ds_read_b128
// (the lanes of the set read distinct addresses in the same four banks, all other lanes read other banks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>

__global__ __launch_bounds__(64) void probe(const int* __restrict__ addr, long long* __restrict__ cycles, int npat, int iters) {
  __shared__ __attribute__((aligned(16))) double sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = i;
  __syncthreads();
  for (int p = 0; p < npat; p++) {
    const unsigned a = (unsigned)addr[p * 64 + threadIdx.x] + (unsigned)(size_t)sm;
    double acc = 0.0;
    __builtin_amdgcn_s_waitcnt(0);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 x0, x1, x2, x3, x4, x5, x6, x7;
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8\n ds_read_b128 %2, %8\n ds_read_b128 %3, %8\n"
                   "ds_read_b128 %4, %8\n ds_read_b128 %5, %8\n ds_read_b128 %6, %8\n ds_read_b128 %7, %8\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7) : "v"(a) : "memory");
      acc += x0.x + x7.y;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[p] = t1 - t0;
    if (acc == 12345.678) cycles[p] = 0;
  }
}

int main() {
  struct Pat { std::string name; std::vector<int> set; };
  auto range = [](std::initializer_list<std::pair<int, int>> rs) { std::vector<int> v; for (auto r : rs) for (int i = r.first; i <= r.second; i++) v.push_back(i); return v; };
  std::vector<Pat> pats = {
      {"none", {}},
      {"0-15", range({{0, 15}})},
      {"0-7", range({{0, 7}})},
      {"4-11", range({{4, 11}})},
      {"0-3,12-15", range({{0, 3}, {12, 15}})},
      {"0-3,12-15,20-27", range({{0, 3}, {12, 15}, {20, 27}})},
      {"4-11,16-19,28-31", range({{4, 11}, {16, 19}, {28, 31}})},
      {"0-3,32-35", range({{0, 3}, {32, 35}})},
      {"4-7,36-39", range({{4, 7}, {36, 39}})},
      {"4-7,8-11", range({{4, 7}, {8, 11}})},
      {"4-7,16-19", range({{4, 7}, {16, 19}})},
      {"4-7,28-31", range({{4, 7}, {28, 31}})},
      {"even lanes 0-30", {0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30}},
      {"lanes 0,4,8,..,60", {0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60}},
      {"0-31", range({{0, 31}})},
      {"0-63", range({{0, 63}})},
  };
  const int npat = (int)pats.size(), iters = 200;
  std::vector<int> h((size_t)npat * 64);
  for (int p = 0; p < npat; p++) {
    for (int l = 0; l < 64; l++) h[p * 64 + l] = 16 * (1 + (l % 15));
    int k = 0;
    for (int l : pats[p].set) h[p * 64 + l] = 256 * (k++);
  }
  int* d_a; long long* d_c;
  hipMalloc((void**)&d_a, h.size() * 4); hipMalloc((void**)&d_c, npat * 8);
  hipMemcpy(d_a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_a, d_c, npat, iters);
  hipDeviceSynchronize();
  std::vector<long long> c(npat);
  hipMemcpy(c.data(), d_c, npat * 8, hipMemcpyDeviceToHost);
  for (int p = 0; p < npat; p++) printf("%-22s %2d lanes in one bank group: %7.2f cycles per read\n", pats[p].name.c_str(), (int)pats[p].set.size(), (double)c[p] / (8.0 * iters));
  return 0;
}
