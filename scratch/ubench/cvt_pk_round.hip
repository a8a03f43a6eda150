// Does v_cvt_pk_u8_f32 round to nearest-even by itself (then the v_rndne_f32 in front of it in the quantising epilogues is redundant)?
// hipcc --offload-arch=gfx950 -O2 cvt_pk_round.hip -o /tmp/cvt_pk_round && /tmp/cvt_pk_round
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* x, unsigned* direct, unsigned* rounded, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  direct[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
  rounded[i] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(x[i]), 0, 0u);
}
int main() {
  const int n = 1 << 16;
  float* hx = new float[n];
  for (int i = 0; i < n; ++i) hx[i] = -8.0f + i * (272.0f / n);          // includes every k + 0.5 exactly (step 2^-8 ... 272/65536 = 0.00415: not exact)
  for (int k = 0; k < 600; ++k) hx[k] = -20.0f + 0.5f * k;               // exact halves and integers from -20 to 280
  for (int k = 0; k < 600; ++k) hx[600 + k] = -20.0f + 0.5f * k + 1e-4f;
  for (int k = 0; k < 600; ++k) hx[1200 + k] = -20.0f + 0.5f * k - 1e-4f;
  float* dx; unsigned *dd, *dr;
  hipMalloc(&dx, n * 4); hipMalloc(&dd, n * 4); hipMalloc(&dr, n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dd, dr, n);
  unsigned* hd = new unsigned[n]; unsigned* hr = new unsigned[n];
  hipMemcpy(hd, dd, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hr, dr, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) if (hd[i] != hr[i]) { if (bad < 12) printf("x = %.6f: cvt_pk alone %u, rint + cvt_pk %u\n", hx[i], hd[i], hr[i]); ++bad; }
  printf("cvt_pk_u8_f32 vs rint + cvt_pk_u8_f32: %d of %d differ\n", bad, n);
  return 0;
}
