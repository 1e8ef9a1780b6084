// Issue cost of the fp64 / integer VALU instructions the RT kernels are made of, relative to v_fma_f64: 3 waves per
// SIMD (the lbl kernel's occupancy), 8 independent chains per wave, straight-line blocks of 64 instructions.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/insn_rate.hip -o tools/ubench/insn_rate && tools/ubench/insn_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BLOCK8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int OP>
__global__ __launch_bounds__(256) void k(int iters, double seed, double *out) {
  double v[8];
  int w[8];
  unsigned long long msk = 0x5555555555555555ull;
  for (int i = 0; i < 8; ++i) w[i] = threadIdx.x + i;
  int e = (int)seed;
  for (int i = 0; i < 8; ++i) v[i] = seed + 0.001 * (threadIdx.x + i);
  const double m = 0.9999999, a = 1e-9;
  for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(a));
#define FMAS(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "s"(m), "v"(a));
#define FMASS(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[i]) : "s"(m));
#define FMAD(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 2) & 7]));
#define MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(m));
#define ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(a));
#define LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[i]) : "v"(e));
#define MAXF(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[i]) : "v"(a));
#define MOV64(i) asm volatile("v_mov_b64 %0, %1" : "+v"(v[i]) : "v"(m));
#define RNDNE(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[i]));
#define CVT(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(e) : "v"(v[i]));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(e) : "v"(it));
#define RCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[i]));
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[i]) : "v"(it));
#define CNDS(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(w[i]) : "v"(it), "s"(msk));
#define CMP(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(v[i]), "v"(m) : "vcc");
#define CMPS(i) asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(msk) : "v"(v[i]), "v"(m));
#define ADDU8(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[i]) : "v"(it));
#define DPP8(i) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(w[i]));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(e));
    if (OP == 0) { BLOCK8(FMA) }
    if (OP == 1) { BLOCK8(MUL) }
    if (OP == 2) { BLOCK8(ADD) }
    if (OP == 3) { BLOCK8(LDEXP) }
    if (OP == 4) { BLOCK8(MAXF) }
    if (OP == 5) { BLOCK8(MOV64) }
    if (OP == 6) { BLOCK8(RNDNE) }
    if (OP == 7) { BLOCK8(CVT) }
    if (OP == 8) { BLOCK8(ADDU) }
    if (OP == 9) { BLOCK8(RCP) }
    if (OP == 10) { BLOCK8(CNDM) }
    if (OP == 11) { BLOCK8(DPP) }
    if (OP == 12) { BLOCK8(CNDS) }
    if (OP == 13) { BLOCK8(CMP) }
    if (OP == 14) { BLOCK8(CMPS) }
    if (OP == 15) { BLOCK8(ADDU8) }
    if (OP == 16) { BLOCK8(DPP8) }
    if (OP == 17) { BLOCK8(FMAS) }
    if (OP == 18) { BLOCK8(FMASS) }
    if (OP == 19) { BLOCK8(FMAD) }
  }
  double r = e + (double)(msk & 1);
  for (int i = 0; i < 8; ++i) r += v[i] + w[i];
  if (r == 1.2345) out[0] = r;
}

template <int OP>
float run(int iters, double *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(768), dim3(256), 0, 0, iters, 1.0, out);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(768), dim3(256), 0, 0, iters, 1.0, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double *out;
  (void)hipMalloc(&out, 8);
  const int iters = 20000;
  const char *names[20] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_ldexp_f64", "v_max_f64", "v_mov_b64", "v_rndne_f64",
                           "v_cvt_i32_f64", "v_add_u32 (1 chain)", "v_rcp_f64", "v_cndmask vcc (8 regs)", "v_mov_b32_dpp (1 chain)",
                           "v_cndmask sgpr (8 regs)", "v_cmp_lt_f64 vcc", "v_cmp_lt_f64 sgpr", "v_add_u32 (8 regs)", "v_mov_b32_dpp (8 regs)", "v_fma_f64 v,v,s,v", "v_fma_f64 v,v,s,s", "v_fma_f64 3 distinct vgprs"};
  float ms[20] = {run<0>(iters, out), run<1>(iters, out), run<2>(iters, out), run<3>(iters, out), run<4>(iters, out), run<5>(iters, out),
                  run<6>(iters, out), run<7>(iters, out), run<8>(iters, out), run<9>(iters, out), run<10>(iters, out), run<11>(iters, out),
                  run<12>(iters, out), run<13>(iters, out), run<14>(iters, out), run<15>(iters, out), run<16>(iters, out),
                  run<17>(iters, out), run<18>(iters, out), run<19>(iters, out)};
  for (int i = 0; i < 20; ++i)  // 3 waves per SIMD x 64 instructions per iteration
    printf("%-26s %8.3f ms  = %5.2f ns per instruction per SIMD = %5.2f x v_fma_f64\n", names[i], ms[i],
           ms[i] * 1e6 / iters / (3 * 64), ms[i] / ms[0]);
  return 0;
}
