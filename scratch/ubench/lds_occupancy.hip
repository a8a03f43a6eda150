// How many 256-thread blocks with N bytes of static LDS does a gfx950 CU take?  (hipOccupancyMaxActiveBlocksPerMultiprocessor)
//   hipcc --offload-arch=gfx950 -O2 lds_occupancy.hip -o /tmp/lds_occ && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ __launch_bounds__(256, 2) void k(float* o) {
  __shared__ unsigned char lds[BYTES];
  lds[threadIdx.x] = static_cast<unsigned char>(threadIdx.x);
  __syncthreads();
  o[threadIdx.x] = lds[(threadIdx.x * 7) % BYTES];
}
template <int BYTES>
void report() {
  int n = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BYTES>, 256, 0);
  printf("static LDS %6d B, 256 threads: %d blocks per CU\n", BYTES, n);
}
int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("%s: sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.gcnArchName, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
  report<65536>(); report<73728>(); report<77824>(); report<79872>(); report<80896>(); report<81920>(); report<83968>();
  return 0;
}
