// Issue rate of the VALU instructions the quantising epilogues are made of, per SIMD (gfx950): cycles per wave64 instruction
// with W waves resident on one SIMD, 8 independent chains per wave.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float v[8];
  f2 w[8];
  for (int i = 0; i < 8; ++i) { v[i] = 0.5f + 0.001f * (threadIdx.x + i); w[i] = f2{v[i], v[i] + 0.25f}; }
  unsigned u = threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
#define PKF(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[i]));
#define PKM(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(w[i]));
#define RND(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[i]));
#define MED(i) asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(v[i]));
#define CVT(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u) : "v"(v[i]));
#define CVI(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(v[i]));
#define SQR(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
    if (OP == 0) { REP8(EXP) }
    if (OP == 1) { REP8(RCP) }
    if (OP == 2) { REP8(FMA) }
    if (OP == 3) { REP8(PKF) }
    if (OP == 4) { REP8(PKM) }
    if (OP == 5) { REP8(RND) }
    if (OP == 6) { REP8(MED) }
    if (OP == 7) { REP8(CVT) }
    if (OP == 8) { REP8(CVI) }
    if (OP == 9) { REP8(SQR) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long c1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i] + w[i].x + w[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + u;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = c1 - c0; }
}
template <int OP>
void run(const char* name, float* out, unsigned long long* cyc) {
  for (int waves : {1, 2, 4}) {      // waves per SIMD: block of 256 * waves threads on one CU
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256 * waves), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256 * waves), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double ninstr = double(iters) * 8;
    printf("%-16s waves/SIMD %d: %7.2f ns per instr per wave (event), readcyclecounter %.2f ticks/instr, clock64 %.2f ticks/instr -> per SIMD %.2f ns/instr\n",
           name, waves, ms * 1e6 / ninstr, double(h[0]) / ninstr, double(h[1]) / ninstr, ms * 1e6 / ninstr / waves);
  }
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 4 * 4); hipMalloc(&cyc, 64);
  run<0>("v_exp_f32", out, cyc); run<1>("v_rcp_f32", out, cyc); run<9>("v_sqrt_f32", out, cyc); run<2>("v_fma_f32", out, cyc);
  run<3>("v_pk_fma_f32", out, cyc); run<4>("v_pk_mul_f32", out, cyc); run<5>("v_rndne_f32", out, cyc); run<6>("v_med3_f32", out, cyc);
  run<7>("v_cvt_pk_u8_f32", out, cyc); run<8>("v_cvt_f32_i32", out, cyc);
  return 0;
}
