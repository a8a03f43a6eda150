// Two waves per SIMD, one s_barrier per step: does a wave's VALU block hide under its SIMD partner's MFMA block when the two groups run the
// step's halves in OPPOSITE program order (group A: MFMAs then VALU, group B: VALU then MFMAs), compared with both in the same order?
// Per step and wave: NM v_mfma_i32_32x32x32_i8 (32 cycles each) and NV dependent-free VALU instructions (v_fma_f32 / v_exp_f32 mix).
//   hipcc --offload-arch=gfx950 -O3 pingpong_mfma_valu.hip -o pingpong_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NM, int NV, int MODE>      // MODE 0: same order in both groups, 1: opposite order, 2: MFMAs only, 3: VALU only
__global__ __launch_bounds__(512, 2) void k(int steps, float* out, int* outi) {
  const int wid = threadIdx.x >> 6;
  v16i acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = threadIdx.x + r;
  v4i a = {static_cast<int>(threadIdx.x), 1, 2, 3}, b = {4, 5, 6, static_cast<int>(threadIdx.x)};
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.001f + i;
  auto mfmas = [&]() {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i & 3], 0, 0, 0);
  };
  auto valus = [&]() {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if ((i & 7) == 7) f[i & 7] = __builtin_amdgcn_exp2f(f[i & 7]);
      else f[i & 7] = __builtin_fmaf(f[i & 7], 1.0001f, 0.5f);
    }
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
  };
  const bool second = MODE == 1 && wid >= 4;
  for (int s = 0; s < steps; ++s) {
    asm volatile("s_barrier" ::: "memory");
    if (second) {
      if (MODE != 2) valus();
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 3) mfmas();
    } else {
      if (MODE != 3) mfmas();
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 2) valus();
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float t = 0;
  for (int i = 0; i < 8; ++i) t += f[i];
  int ti = 0;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) ti += acc[j][r];
  out[blockIdx.x * 512 + threadIdx.x] = t;
  outi[blockIdx.x * 512 + threadIdx.x] = ti;
}

template <int NM, int NV, int MODE>
float run(float* o, int* oi) {
  const int steps = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, NV, MODE>), dim3(256), dim3(512), 0, 0, 100, o, oi);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NM, NV, MODE>), dim3(256), dim3(512), 0, 0, steps, o, oi);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / steps;      // ns per step
}

template <int NM, int NV>
void row(float* o, int* oi) {
  const float m = run<NM, NV, 2>(o, oi), v = run<NM, NV, 3>(o, oi), same = run<NM, NV, 0>(o, oi), opp = run<NM, NV, 1>(o, oi);
  printf("NM=%2d NV=%3d: MFMA only %6.1f ns  VALU only %6.1f ns  same order %6.1f ns  opposite order %6.1f ns   (sum %6.1f, max %6.1f)\n", NM, NV, m, v, same, opp,
         m + v, m > v ? m : v);
}

int main() {
  float* o;
  int* oi;
  hipMalloc(reinterpret_cast<void**>(&o), 256 * 512 * 4);
  hipMalloc(reinterpret_cast<void**>(&oi), 256 * 512 * 4);
  printf("per step and wave: NM MFMAs (i8 32x32x32), NV VALU (7 v_fma : 1 v_exp); 8 waves per CU, 2 per SIMD, one s_barrier per step\n");
  row<8, 16>(o, oi);
  row<8, 32>(o, oi);
  row<8, 64>(o, oi);
  row<8, 96>(o, oi);
  row<16, 64>(o, oi);
  row<16, 128>(o, oi);
  row<4, 32>(o, oi);
  return 0;
}
