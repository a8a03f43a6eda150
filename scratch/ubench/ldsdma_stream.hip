// How fast does a CU stream global -> LDS by LDS-DMA (global_load_lds_dwordx4), as a function of the waves issuing and the 1-KiB pieces each
// keeps in flight?  Decides whether the w4a8 GEMM family's "~22-27 B/clk/CU from L2" is a throughput cap (only fewer bytes help) or
// latency x in-flight depth (deeper rings help).  Standalone: hipcc --offload-arch=gfx950 -O3 ldsdma_stream.hip -o ldsdma_stream
//   mode 0: every block reads the SAME region (weights: L2 hits after the first touch), 1: a private region per block sized to stay in L2,
//   2: a private region per block far larger than the caches (HBM stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int D>
__global__ __launch_bounds__(1024) void k_stream(const unsigned char* src, size_t region, int iters, int priv, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const unsigned char* base = src + (priv ? static_cast<size_t>(blockIdx.x) * region : 0);
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds)) + wid * D * 1024;
  const size_t mask = region - 1;        // power of two
  size_t off = static_cast<size_t>(wid) * 1024 + lane * 16;
  const size_t stride = static_cast<size_t>(nw) * 1024;
  int slot = 0;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    glds16(base + (off & mask), lds0 + i * 1024);
    off += stride;
  }
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
    glds16(base + (off & mask), lds0 + slot * 1024);
    off += stride;
    slot = slot + 1 == D ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (iters < 0) sink[threadIdx.x] = lds[threadIdx.x];
}

template <int D>
double run(const unsigned char* buf, size_t region, int mode, int nw, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t shm = static_cast<size_t>(nw) * D * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream<D>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shm));
  hipLaunchKernelGGL(k_stream<D>, dim3(blocks), dim3(nw * 64), shm, 0, buf, region, iters / 4, mode != 0, nullptr);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_stream<D>, dim3(blocks), dim3(nw * 64), shm, 0, buf, region, iters, mode != 0, nullptr);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) return -1.0;
  const double bytes = static_cast<double>(blocks) * nw * (iters + D) * 1024.0;
  return bytes / (ms * 1e-3);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  unsigned char* buf = nullptr;
  const size_t total = static_cast<size_t>(2) << 30;
  if (hipMalloc(reinterpret_cast<void**>(&buf), total) != hipSuccess) return 1;
  hipMemset(buf, 1, total);
  const char* names[3] = {"shared 1 MiB region (weights in L2)", "private 64 KiB region per block (L2-resident)", "private 8 MiB region per block (HBM stream)"};
  const size_t regions[3] = {1u << 20, 64u << 10, 8u << 20};
  for (int mode = 0; mode < 3; ++mode) {
    printf("== %s, %d blocks: aggregate TB/s  [GB/s per CU]\n", names[mode], blocks);
    printf("waves/block  D=1          D=2          D=4          D=8          D=16\n");
    for (int nw : {4, 8, 12, 16}) {
      printf("%5d      ", nw);
      const int iters = 4096;
      double r[5];
      r[0] = run<1>(buf, regions[mode], mode, nw, blocks, iters);
      r[1] = run<2>(buf, regions[mode], mode, nw, blocks, iters);
      r[2] = run<4>(buf, regions[mode], mode, nw, blocks, iters);
      r[3] = nw * 8 <= 152 ? run<8>(buf, regions[mode], mode, nw, blocks, iters) : -1;
      r[4] = nw * 16 <= 152 ? run<16>(buf, regions[mode], mode, nw, blocks, iters) : -1;
      for (int i = 0; i < 5; ++i) {
        if (r[i] < 0) printf("   --        ");
        else printf(" %5.2f [%5.1f]", r[i] / 1e12, r[i] / 1e9 / (blocks < 256 ? blocks : 256));
      }
      printf("\n");
    }
  }
  return 0;
}
