// Microbenchmark behind rcm_measure_hbm_bandwidth: which launch geometry / unroll streams HBM fastest on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/stream_bw.hip -o tools/ubench/stream_bw && tools/ubench/stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const d2 *__restrict__ src, d2 *__restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const d2 *__restrict__ src, double *sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double a = 0, b = 0;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) a += v[u].x, b += v[u].y;
  }
  for (; i < n; i += stride) a += src[i].x;
  if (a + b == 1.2345e300) sink[0] = a;
}

template <int U, bool NT>
void run(const d2 *a, d2 *b, size_t n, int blocks) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  float best_c = 1e9f, best_r = 1e9f, ms;
  for (int r = 0; r < 6; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_copy<U, NT>), dim3(blocks), dim3(256), 0, 0, a, b, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best_c) best_c = ms;
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_read<U, NT>), dim3(blocks), dim3(256), 0, 0, a, (double *)b, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best_r) best_r = ms;
  }
  printf("unroll %d nt %d blocks %6d: copy %7.1f GB/s (read+write)   read-only %7.1f GB/s\n", U, (int)NT, blocks,
         2.0 * n * 16 / best_c / 1e6, 1.0 * n * 16 / best_r / 1e6);
}

int main() {
  const size_t bytes = (size_t)2 << 30, n = bytes / 16;
  d2 *a, *b;
  (void)hipMalloc(&a, bytes), (void)hipMalloc(&b, bytes);
  (void)hipMemset(a, 0, bytes), (void)hipMemset(b, 0, bytes);
  for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64}) {
    run<1, false>(a, b, n, blocks);
    run<4, false>(a, b, n, blocks);
    run<4, true>(a, b, n, blocks);
    run<8, true>(a, b, n, blocks);
  }
  run<1, false>(a, b, n, (int)(n / 256));  // one 16-byte piece per lane, no loop
  return 0;
}
