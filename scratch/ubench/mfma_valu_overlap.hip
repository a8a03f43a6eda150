#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_v_fma_0(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_fma_2(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_fma_4(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_fma_6(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_fma_8(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_fma_12(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_exp_2(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_exp_4(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_exp_6(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_exp_8(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_exp_12(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_cvt_2(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_cvt_4(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_cvt_6(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_cvt_8(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_cvt_12(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_max3_2(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_max3_4(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_max3_6(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_max3_8(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_v_max3_12(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_fma_0(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_fma_2(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_fma_4(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_fma_6(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_fma_8(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_fma_12(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\tv_fma_f32 %8, %8, %8, %8\n\tv_fma_f32 %9, %9, %9, %9\n\tv_fma_f32 %10, %10, %10, %10\n\tv_fma_f32 %11, %11, %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_exp_2(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_exp_4(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_exp_6(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_exp_8(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_exp_12(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_cvt_2(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_cvt_4(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_cvt_6(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_cvt_8(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_cvt_12(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11\n\tv_cvt_pk_f16_f32 %4, %4, %4\n\tv_cvt_pk_f16_f32 %5, %5, %5\n\tv_cvt_pk_f16_f32 %6, %6, %6\n\tv_cvt_pk_f16_f32 %7, %7, %7\n\tv_cvt_pk_f16_f32 %8, %8, %8\n\tv_cvt_pk_f16_f32 %9, %9, %9\n\tv_cvt_pk_f16_f32 %10, %10, %10\n\tv_cvt_pk_f16_f32 %11, %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_max3_2(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_max3_4(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_max3_6(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_max3_8(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_a_max3_12(float* out, int iters) {
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11\n\tv_max3_f32 %4, %4, %4, %4\n\tv_max3_f32 %5, %5, %5, %5\n\tv_max3_f32 %6, %6, %6, %6\n\tv_max3_f32 %7, %7, %7, %7\n\tv_max3_f32 %8, %8, %8, %8\n\tv_max3_f32 %9, %9, %9, %9\n\tv_max3_f32 %10, %10, %10, %10\n\tv_max3_f32 %11, %11, %11, %11" : "+a"(a0), "+a"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}

int main(int argc, char** argv) {
  const int iters = 4000; float* out; hipMalloc(&out, (1 << 22) * 4 + 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-16s %5s %9s %12s %12s\n", "kernel", "w/SIMD", "us", "ns/MFMA/SIMD", "clk/MFMA");
  for (int occ = 1; occ <= 3; ++occ) {
    { k_v_fma_0<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_fma_0<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_fma_0", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_fma_2<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_fma_2<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_fma_2", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_fma_4<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_fma_4<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_fma_4", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_fma_6<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_fma_6<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_fma_6", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_fma_8<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_fma_8<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_fma_8", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_fma_12<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_fma_12<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_fma_12", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_exp_2<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_exp_2<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_exp_2", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_exp_4<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_exp_4<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_exp_4", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_exp_6<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_exp_6<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_exp_6", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_exp_8<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_exp_8<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_exp_8", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_exp_12<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_exp_12<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_exp_12", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_cvt_2<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_cvt_2<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_cvt_2", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_cvt_4<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_cvt_4<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_cvt_4", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_cvt_6<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_cvt_6<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_cvt_6", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_cvt_8<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_cvt_8<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_cvt_8", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_cvt_12<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_cvt_12<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_cvt_12", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_max3_2<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_max3_2<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_max3_2", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_max3_4<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_max3_4<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_max3_4", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_max3_6<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_max3_6<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_max3_6", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_max3_8<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_max3_8<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_max3_8", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_v_max3_12<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_v_max3_12<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_v_max3_12", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_fma_0<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_fma_0<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_fma_0", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_fma_2<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_fma_2<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_fma_2", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_fma_4<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_fma_4<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_fma_4", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_fma_6<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_fma_6<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_fma_6", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_fma_8<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_fma_8<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_fma_8", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_fma_12<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_fma_12<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_fma_12", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_exp_2<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_exp_2<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_exp_2", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_exp_4<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_exp_4<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_exp_4", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_exp_6<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_exp_6<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_exp_6", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_exp_8<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_exp_8<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_exp_8", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_exp_12<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_exp_12<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_exp_12", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_cvt_2<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_cvt_2<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_cvt_2", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_cvt_4<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_cvt_4<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_cvt_4", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_cvt_6<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_cvt_6<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_cvt_6", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_cvt_8<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_cvt_8<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_cvt_8", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_cvt_12<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_cvt_12<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_cvt_12", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_max3_2<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_max3_2<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_max3_2", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_max3_4<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_max3_4<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_max3_4", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_max3_6<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_max3_6<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_max3_6", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_max3_8<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_max3_8<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_max3_8", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
    { k_a_max3_12<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); k_a_max3_12<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\n", "k_a_max3_12", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }
  }
  return 0;
}