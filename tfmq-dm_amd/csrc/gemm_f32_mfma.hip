// K15: the reconstruction GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2f32: exact fp32 products, fp32
// accumulation -- the same arithmetic as the FMA kernel in recon_kernels.hip, 4-5x its rate).  Calibration wall-clock
// is reconstruction-GEMM time: 20 000 Adam iterations per unit, each a block forward + backward.
//
// 128 x BN tile (BN = 128 or 64), 4 waves, K-step 16, double-buffered LDS stored K-major ([k][m], [k][n]) so that an
// MFMA fragment (lane = row/col l%32, k = l/32) is one conflict-free ds_read_b32; global -> register prefetch of the next
// K-step while the current one is multiplied; 16-byte global loads along whichever dimension of an operand is
// contiguous (any of the four transpose combinations of the strided-GEMM ABI), scalar loads otherwise.
#include "common.hpp"
#include <cstdlib>

typedef float v16f __attribute__((ext_vector_type(16)));


// One operand tile [R rows][16 k] -> LDS [16][R + 4].  (rs, ks) = element strides of the row / k index.
template <int R>
struct TileLoader {
  static constexpr int ITEMS = R * 16 / 4 / 256;   // float4 items per thread (R = 128 -> 2, R = 64 -> 1)
  float4 reg[ITEMS];
  int mode;                                        // 0 scalar, 1 vector along k, 2 vector along rows
  __device__ __forceinline__ void init(long rs, long ks, const float* base, int rows, int K) {
    const bool al = (reinterpret_cast<uintptr_t>(base) & 15) == 0;
    if (ks == 1 && al && (rs & 3) == 0 && (K & 3) == 0) mode = 1;
    else if (rs == 1 && al && (ks & 3) == 0 && (rows & 3) == 0) mode = 2;
    else mode = 0;
  }
  __device__ __forceinline__ void load(const float* base, long rs, long ks, int r0, int k0, int rows, int K) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = tid + it * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mode == 1) {            // 4 consecutive k of one row
        const int r = e >> 2, kq = (e & 3) * 4;
        const int gr = r0 + r, gk = k0 + kq;
        if (gr < rows && gk < K) v = *reinterpret_cast<const float4*>(base + gr * rs + gk);   // K % 4 == 0: whole or nothing
      } else if (mode == 2) {     // 4 consecutive rows of one k
        const int k = e / (R / 4), rq = (e % (R / 4)) * 4;
        const int gr = r0 + rq, gk = k0 + k;
        if (gr < rows && gk < K) v = *reinterpret_cast<const float4*>(base + gk * ks + gr);    // rows % 4 == 0
      } else {
        const int r = e >> 2, kq = (e & 3) * 4;
        const int gr = r0 + r;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (gr < rows) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (k0 + kq + q < K) t[q] = base[gr * rs + (k0 + kq + q) * ks];
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
      }
      reg[it] = v;
    }
  }
  __device__ __forceinline__ void store(float* lds) {    // lds: [16][R + 4]
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = tid + it * 256;
      const float4 v = reg[it];
      if (mode == 2) {
        const int k = e / (R / 4), rq = (e % (R / 4)) * 4;
        *reinterpret_cast<float4*>(lds + k * (R + 4) + rq) = v;
      } else {
        const int r = e >> 2, kq = (e & 3) * 4;
        lds[(kq + 0) * (R + 4) + r] = v.x;
        lds[(kq + 1) * (R + 4) + r] = v.y;
        lds[(kq + 2) * (R + 4) + r] = v.z;
        lds[(kq + 3) * (R + 4) + r] = v.w;
      }
    }
  }
};

template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES>
__global__ __launch_bounds__(256) void k_gemm_f32_mfma(GemmP p) {
  constexpr int BM = WAVES_M * WM_TILES * 32, BN = WAVES_N * WN_TILES * 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float sA[2][16 * (BM + 4)];
  __shared__ __attribute__((aligned(16))) float sB[2][16 * (BN + 4)];
  // split-K (skinny outputs with a long reduction: weight gradients dW = X^T dY, K = B*T): blockIdx.z also carries the
  // K slice; slices write raw partial tiles, k_gemm_splitk_reduce adds them in slice order (deterministic)
  int bz = blockIdx.z, kb = 0, ke = p.K;
  if (p.ksplit > 1) {
    const int sp = bz % p.ksplit;
    bz /= p.ksplit;
    kb = sp * p.kchunk;
    ke = kb + p.kchunk < p.K ? kb + p.kchunk : p.K;
  }
  const float* A = p.A + bz * p.bsa;
  const float* B = p.B + bz * p.bsb;
  float* C = p.C + bz * p.bsc;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int l32 = lane & 31, hh = lane >> 5;

  TileLoader<BM> la;
  TileLoader<BN> lb;
  la.init(p.sam, p.sak, A, p.M, p.K);
  lb.init(p.sbn, p.sbk, B, p.N, p.K);

  v16f acc[WM_TILES][WN_TILES];
#pragma unroll
  for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
    for (int j = 0; j < WN_TILES; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = (ke - kb + 15) / 16;
  la.load(A, p.sam, p.sak, m0, kb, p.M, ke);
  lb.load(B, p.sbn, p.sbk, n0, kb, p.N, ke);
  la.store(sA[0]);
  lb.store(sB[0]);
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) {
      la.load(A, p.sam, p.sak, m0, kb + (s + 1) * 16, p.M, ke);
      lb.load(B, p.sbn, p.sbk, n0, kb + (s + 1) * 16, p.N, ke);
    }
    const float* a_l = sA[buf] + (wm * WM_TILES * 32) + l32;
    const float* b_l = sB[buf] + (wn * WN_TILES * 32) + l32;
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {       // 8 MFMA k-pairs per 16-wide K-step
      float af[WM_TILES], bf[WN_TILES];
#pragma unroll
      for (int i = 0; i < WM_TILES; ++i) af[i] = a_l[(kp * 2 + hh) * (BM + 4) + i * 32];
#pragma unroll
      for (int j = 0; j < WN_TILES; ++j) bf[j] = b_l[(kp * 2 + hh) * (BN + 4) + j * 32];
#pragma unroll
      for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
        for (int j = 0; j < WN_TILES; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nk) {
      la.store(sA[buf ^ 1]);
      lb.store(sB[buf ^ 1]);
    }
    __syncthreads();
  }

  // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
    for (int j = 0; j < WN_TILES; ++j) {
      const int n = n0 + (wn * WN_TILES + j) * 32 + l32;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * WM_TILES + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (m >= p.M) continue;
        if (p.ksplit > 1) {
          p.partial[(static_cast<size_t>(blockIdx.z % p.ksplit) * gridDim.z / p.ksplit + bz) * p.M * p.N +
                    static_cast<size_t>(m) * p.N + n] = acc[i][j][r];
          continue;
        }
        float v = p.alpha * acc[i][j][r];
        if (p.bias) v += bv;
        if (p.rowadd) v += p.rowadd[static_cast<long>(m / p.rows_per_img) * p.rowadd_ld + n];
        if (p.residual) v += p.residual[bz * p.bsc + m * p.scm + n];
        float* c = C + m * p.scm + n;
        *c = p.accumulate ? *c + v : v;
      }
    }
}

// C(bz, m, n) = alpha * sum_s partial[s][bz][m][n]  (+ bias / rowadd / residual, accumulate) -- the epilogue of the slices
__global__ void k_gemm_splitk_reduce(GemmP p, int batch) {
  const size_t per = static_cast<size_t>(p.M) * p.N, total = per * batch;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int bz = static_cast<int>(i / per);
    const size_t r = i - static_cast<size_t>(bz) * per;
    const int m = static_cast<int>(r / p.N), n = static_cast<int>(r - static_cast<size_t>(m) * p.N);
    float a = 0.0f;
    for (int sp = 0; sp < p.ksplit; ++sp) a += p.partial[static_cast<size_t>(sp) * total + i];
    float v = p.alpha * a;
    if (p.bias) v += p.bias[n];
    if (p.rowadd) v += p.rowadd[static_cast<long>(m / p.rows_per_img) * p.rowadd_ld + n];
    if (p.residual) v += p.residual[bz * p.bsc + m * p.scm + n];
    float* c = p.C + bz * p.bsc + m * p.scm + n;
    *c = p.accumulate ? *c + v : v;
  }
}

// called from tfmq_gemm_f32 (recon_kernels.hip) when the problem is large enough for 128-row tiles
int tfmq_gemm_f32_mfma_launch(tfmq_handle h, GemmP& p, int batch, hipStream_t st) {
  const int M = p.M, N = p.N;
  // 128 x 64 tiles (4 waves along M) measured faster than 128 x 128 at every SD unit shape (no column waste at
  // N = 320 / 640, twice the blocks for the mid-sized problems); TFMQ_GEMM_BN128 keeps the wide tile for A/B runs
  const int BN = (N > 64 && getenv("TFMQ_GEMM_BN128")) ? 128 : 64;
  // tiles of ONE batch item: the slicing (hence the summation order) must not depend on how many items share the
  // launch -- results stay bit-identical whatever else is in the batch
  const long tiles = static_cast<long>((N + BN - 1) / BN) * ((M + 127) / 128);
  // fewer tiles than CUs and a long reduction: slice K so that ~2 blocks per CU exist, >= 256 elements per slice
  int ks = 1;
  if (tiles < 2L * h->cu_count && p.K >= 1024) {
    ks = static_cast<int>((3L * h->cu_count + tiles - 1) / tiles);
    if (ks > p.K / 256) ks = p.K / 256;
    if (ks > 64) ks = 64;
    if (static_cast<long>(batch) * ks > 65535) ks = 65535 / batch;
  }
  if (ks > 1) {
    p.kchunk = ((p.K + ks - 1) / ks + 15) / 16 * 16;
    ks = (p.K + p.kchunk - 1) / p.kchunk;
    const size_t need = static_cast<size_t>(ks) * batch * M * N * sizeof(float);
    if (need > h->gemm_ws_bytes) {
      if (h->gemm_ws) (void)hipFree(h->gemm_ws);   // stream-ordered users of the old block have been enqueued: hipFree syncs
      h->gemm_ws = nullptr;
      h->gemm_ws_bytes = 0;
      if (hipMalloc(reinterpret_cast<void**>(&h->gemm_ws), need) != hipSuccess) {
        h->err = "gemm_f32: split-K workspace allocation failed";
        return TFMQ_ERR_HIP;
      }
      h->gemm_ws_bytes = need;
    }
    p.ksplit = ks;
    p.partial = h->gemm_ws;
  }
  dim3 grid((N + BN - 1) / BN, (M + 127) / 128, batch * (ks > 1 ? ks : 1));
  if (BN == 128) hipLaunchKernelGGL((k_gemm_f32_mfma<2, 2, 2, 2>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2>), grid, dim3(256), 0, st, p);
  if (ks > 1) {
    const size_t total = static_cast<size_t>(M) * N * batch;
    hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, p, batch);
  }
  return TFMQ_OK;
}
