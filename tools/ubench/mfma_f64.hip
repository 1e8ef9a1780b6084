// Microbenchmark for VERDICT r03 item 5: can the fp64 matrix pipe carry the angular quadrature sums of the lane-pair
// sweep while the VALU keeps the recurrences?  Measures, on one wave per SIMD and on 3 waves per SIMD (the lbl kernel's
// occupancy), with s_memtime around an unrolled loop:
//   A  v_fma_f64 alone                         (the VALU stream: 4 independent chains)
//   B  v_mfma_f64_16x16x4_f64 alone            (4 independent accumulator tiles, back to back)
//   C  v_mfma_f64_4x4x4_4b_f64 alone           (8 independent accumulators)
//   D  A and B interleaved 8 : 1               (8 VALU FMAs per 16x16x4 MFMA)
//   E  A and C interleaved 4 : 1
//   F  A and C interleaved 6 : 1
// and prints cycles per instruction of each stream and the cycles of the mixes against the VALU stream alone: the
// matrix pipe is "free" only where mix cycles == VALU-alone cycles.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_f64.hip -o tools/ubench/mfma_f64 && tools/ubench/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double dvec4 __attribute__((ext_vector_type(4)));

#define FMA4()                                   \
  x0 = __builtin_fma(x0, m, a), x1 = __builtin_fma(x1, m, a), x2 = __builtin_fma(x2, m, a), \
  x3 = __builtin_fma(x3, m, a)

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, double seed, double *out, unsigned long long *cyc) {
  double x0 = seed + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const double m = 0.999999, a = 1e-9;
  dvec4 t0 = {0, 0, 0, 0}, t1 = t0, t2 = t0, t3 = t0;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  const double av = seed * 0.5 + (threadIdx.x & 15), bv = (threadIdx.x & 15) == 3 ? 0.25 : 0.0;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) FMA4();
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, t1, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, t2, 0, 0, 0);
        t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, t3, 0, 0, 0);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, s1, 0, 0, 0);
        s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, s2, 0, 0, 0);
        s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, s3, 0, 0, 0);
        s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, s4, 0, 0, 0);
        s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, s5, 0, 0, 0);
        s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, s6, 0, 0, 0);
        s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, s7, 0, 0, 0);
      }
    } else if (MODE == 3) {  // 64 VALU FMAs + 8 MFMA 16x16x4
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, bv, t0, 0, 0, 0); FMA4(); FMA4();
        t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, bv, t1, 0, 0, 0); FMA4(); FMA4();
        t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, bv, t2, 0, 0, 0); FMA4(); FMA4();
        t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, bv, t3, 0, 0, 0); FMA4(); FMA4();
      }
    } else if (MODE == 4) {  // 64 VALU FMAs + 16 MFMA 4x4x4_4b
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, bv, s0, 0, 0, 0); FMA4();
        s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, bv, s1, 0, 0, 0); FMA4();
        s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, bv, s2, 0, 0, 0); FMA4();
        s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x3, bv, s3, 0, 0, 0); FMA4();
        s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, bv, s4, 0, 0, 0); FMA4();
        s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, bv, s5, 0, 0, 0); FMA4();
        s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, bv, s6, 0, 0, 0); FMA4();
        s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(x3, bv, s7, 0, 0, 0); FMA4();
      }
    } else {  // 96 VALU FMAs + 16 MFMA 4x4x4_4b  (the sweep's ratio: ~3 570 VALU per 630 sums = 5.7 : 1)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, bv, s0, 0, 0, 0); FMA4(); FMA4();
        s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, bv, s1, 0, 0, 0); FMA4();
        s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, bv, s2, 0, 0, 0); FMA4(); FMA4();
        s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x3, bv, s3, 0, 0, 0); FMA4();
        s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, bv, s4, 0, 0, 0); FMA4(); FMA4();
        s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, bv, s5, 0, 0, 0); FMA4();
        s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, bv, s6, 0, 0, 0); FMA4(); FMA4();
        s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(x3, bv, s7, 0, 0, 0); FMA4();
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  double r = x0 + x1 + x2 + x3 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
  for (int q = 0; q < 4; ++q) r += t0[q] + t1[q] + t2[q] + t3[q];
  if (r == 1.2345) out[0] = r;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = c1 - c0;
}

template <int MODE>
double run(int blocks, int threads, int iters, double *out, unsigned long long *dcyc, float *ms_out) {
  const int nw = blocks * threads / 64;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, iters, 1.0, out, dcyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, iters, 1.0, out, dcyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(ms_out, e0, e1);
  std::vector<unsigned long long> c(nw);
  (void)hipMemcpy(c.data(), dcyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : c) s += (double)v;
  return s / nw / iters;  // s_memtime ticks (100 MHz reference on gfx950: convert with the wall time below)
}

int main() {
  double *out;
  unsigned long long *dcyc;
  hipMalloc(&out, 8);
  hipMalloc(&dcyc, 1 << 20);
  const int iters = 20000;
  const char *names[6] = {"A valu fma_f64 x64", "B mfma 16x16x4 x8", "C mfma 4x4x4_4b x16", "D 64 valu + 8 mfma16",
                          "E 64 valu + 16 mfma4", "F 96 valu + 16 mfma4"};
  for (int wps = 1; wps <= 3; wps += 2) {  // waves per SIMD: 1 (256-thread block per CU), 3 (three blocks per CU)
    const int blocks = 256 * wps, threads = 256;
    printf("--- %d wave(s) per SIMD, %d blocks x %d threads, %d iterations; ns per iteration per wave from the launch time\n",
           wps, blocks, threads, iters);
    float ms[6];
    run<0>(blocks, threads, iters, out, dcyc, &ms[0]);
    run<1>(blocks, threads, iters, out, dcyc, &ms[1]);
    run<2>(blocks, threads, iters, out, dcyc, &ms[2]);
    run<3>(blocks, threads, iters, out, dcyc, &ms[3]);
    run<4>(blocks, threads, iters, out, dcyc, &ms[4]);
    run<5>(blocks, threads, iters, out, dcyc, &ms[5]);
    for (int m = 0; m < 6; ++m) {
      // all waves of a SIMD run concurrently: launch time / iters = time of one iteration of every wave on its SIMD
      const double ns = ms[m] * 1e6 / iters;
      printf("%-22s %8.1f ns per iteration of all %d wave(s) of a SIMD = %7.1f cycles at 2.4 GHz (%5.1f per wave)\n",
             names[m], ns, wps, ns * 2.4, ns * 2.4 / wps);
    }
  }
  return 0;
}
