// Microbenchmark: cost of one wave64 global_load_dwordx2 (8 B per lane) from an L2-resident table as a function of
// how the 64 lanes are spread over two table rows.  Patterns (row of lane, column of lane):
//   0 contiguous      row A only, col = lane                                  (512 B contiguous)
//   1 pair-interleave row = lane&1,      col = lane>>1   [A B A B ..]          (the lane-pair kernels' layout)
//   2 quad            row = (lane>>1)&1, col = (lane>>2)*2 + (lane&1) [A A B B ..]
//   3 row16           row = (lane>>4)&1, col = (lane&15) + 16*(lane>>5)
//   4 half-wave       row = lane>>5,     col = lane&31
//   5 row8            row = (lane>>3)&1, col = (lane&7) + 8*(lane>>4)   (partner = lane^8: one DPP row_ror:8)
// Every wave issues `iters` x 20 independent loads (like one layer of the repwvl optical-depth stage) and sums them.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/tcp_pattern.hip -o /tmp/tcp_pattern && /tmp/tcp_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PAT>
__global__ __launch_bounds__(256) void k(const double *tab, long ld, int nrows, int iters, double *out) {
  const int lane = threadIdx.x & 63;
  int row, col;
  if (PAT == 0) row = 0, col = lane;
  else if (PAT == 1) row = lane & 1, col = lane >> 1;
  else if (PAT == 2) row = (lane >> 1) & 1, col = (lane >> 2) * 2 + (lane & 1);
  else if (PAT == 3) row = (lane >> 4) & 1, col = (lane & 15) + 16 * (lane >> 5);
  else if (PAT == 4) row = lane >> 5, col = lane & 31;
  else row = (lane >> 3) & 1, col = (lane & 7) + 8 * (lane >> 4);  // 5: half-rows of 8 lanes alternate
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  double acc = 0.0;
  unsigned r0 = (unsigned)(wave * 7) % (unsigned)(nrows - 40);
  for (int it = 0; it < iters; ++it) {
    double v[20];
#pragma unroll
    for (int j = 0; j < 20; ++j) v[j] = tab[(size_t)(r0 + 2 * j + row * 1) * ld + (size_t)(row ? 17 : 0) * ld + col + (it & 1) * 32];
#pragma unroll
    for (int j = 0; j < 20; ++j) acc += v[j];
    r0 = (r0 + 41) % (unsigned)(nrows - 40 - 17);
  }
  if (acc == 1.2345) out[0] = acc;
}

template <int PAT>
float run(const double *tab, long ld, int nrows, int iters, double *out, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, tab, ld, nrows, iters, out);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, tab, ld, nrows, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int bench(long ld, int shift);

int main() {
  bench(128, 0);  // 1 KiB rows, 128-B aligned row segments
  bench(100, 0);  // Reduced100 as it is: 800-B rows, segments start at any multiple of 32 B
  bench(112, 0);  // rows padded to a multiple of 128 B
  bench(128, 3);  // aligned rows, but the wave's columns start 24 B into a line (a flattened item list)
  return 0;
}

int bench(long ld, int shift) {
  const int nrows = 3321;  // 9*9*41 rows: ~3 MB table, L2-resident like xsec of Reduced100
  std::vector<double> h((size_t)nrows * ld, 1.0);
  double *tab, *out;
  hipMalloc(&tab, h.size() * 8), hipMalloc(&out, 8);
  hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  const double *tabs = tab + shift;
  printf("---- row length %ld doubles, column shift %d doubles\n", ld, shift);
  const int iters = 200, blocks = 256 * 3;  // 12 waves per CU
  const char *names[6] = {"contiguous", "pair-interleave [ABAB]", "quad [AABB]", "row16", "half-wave", "row8"};
  float ms[6];
  ms[0] = run<0>(tabs, ld, nrows, iters, out, blocks);
  ms[1] = run<1>(tabs, ld, nrows, iters, out, blocks);
  ms[2] = run<2>(tabs, ld, nrows, iters, out, blocks);
  ms[3] = run<3>(tabs, ld, nrows, iters, out, blocks);
  ms[4] = run<4>(tabs, ld, nrows, iters, out, blocks);
  ms[5] = run<5>(tabs, ld, nrows, iters, out, blocks);
  for (int p = 0; p < 6; ++p) {
    const double loads_per_cu = 12.0 * iters * 20;  // wave-level load instructions per CU
    printf("%-24s %8.3f ms  %6.1f ns per wave-load per CU (%5.1f cycles at 2.4 GHz)\n", names[p], ms[p],
           ms[p] * 1e6 / loads_per_cu, ms[p] * 1e6 / loads_per_cu * 2.4);
  }
  hipFree(tab), hipFree(out);
  (void)hipFree(tab), (void)hipFree(out);
  return 0;
}
