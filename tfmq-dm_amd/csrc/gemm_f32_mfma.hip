// K15: the reconstruction GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2f32: exact fp32 products, fp32
// accumulation -- the same arithmetic as the FMA kernel in recon_kernels.hip, 4-5x its rate).  Calibration wall-clock
// is reconstruction-GEMM time: 20 000 Adam iterations per unit, each a block forward + backward.
//
// 128 x BN tile (BN = 64, or 128 for A/B runs), 4 waves, K-step 16, double-buffered LDS.  Operands reach the CU by
// 16-byte buffer loads along whichever dimension is contiguous (any of the four transpose combinations of the
// strided-GEMM ABI; out-of-range items read zeros from beyond the descriptor's extent: no branch), two K-steps ahead
// of the MFMAs in two register sets.  A k-contiguous operand is stored as swizzled 64-byte rows and read back with
// one ds_read_b128 per four MFMA k-pairs; a row-contiguous one is stored K-major ([k][rows]) and read with ds_read_b32.
// Operands that are neither contiguous nor aligned take the generic loader (scalar loads, run-time mode).
// Measured (SD unit shapes): 97-113 TFLOP/s of the 157 peak; the same loop without its global loads runs at 110-125,
// the library's sgemm at 93-134.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

typedef float v16f __attribute__((ext_vector_type(16)));


// byte offset of 16-byte slot `slot` of row `row` in a [rows][BK floats] tile: 64-byte rows (BK 16) swizzle their 4 slots
// with row bits 2-3, 128-byte rows (BK 32) their 8 slots with row bits 0-2 -- conflict free for the 16-byte stores
// (consecutive lanes = consecutive slots of a row) and the fragment reads (consecutive lanes = consecutive rows)
template <int BK>
__device__ __forceinline__ int swz_rk(int row, int slot) {
  if constexpr (BK == 16) return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
  else return row * 128 + ((slot ^ (row & 7)) << 4);
}

// One operand tile [R rows][16 k] -> LDS [16][R + 4].  (rs, ks) = element strides of the row / k index.
// MODE 0: any strides / alignment, decided at run time (scalar loads where nothing is contiguous and aligned).
template <int R, int MODE, int BK = 16>
struct TileLoader {
  static_assert(BK == 16, "the generic loader is built for 16-wide K-steps");
  static constexpr int ITEMS = R * 16 / 4 / 256;   // float4 items per thread (R = 128 -> 2, R = 64 -> 1)
  float4 reg[2][ITEMS];                            // two register sets: the loads run two K-steps ahead
  int mode;                                        // 0 scalar, 1 vector along k, 2 vector along rows
  const float* base;
  long rs, ks;
  int r0, rows;
  __device__ __forceinline__ void init(const float* b, long rs_, long ks_, int r0_, int kb, int rows_, int K) {
    base = b; rs = rs_; ks = ks_; r0 = r0_; rows = rows_;
    const bool al = (reinterpret_cast<uintptr_t>(base) & 15) == 0;
    if (ks == 1 && al && (rs & 3) == 0 && (K & 3) == 0) mode = 1;
    else if (rs == 1 && al && (ks & 3) == 0 && (rows & 3) == 0) mode = 2;
    else mode = 0;
  }
  template <int S>
  __device__ __forceinline__ void load(int k0, int K) {
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
      reg[S][it] = v;
    }
  }
  template <int S>
  __device__ __forceinline__ void store(float* lds) {    // lds: [16][R + 4]
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = tid + it * 256;
      const float4 v = reg[S][it];
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

// MODE 1 (16-byte loads along k) / MODE 2 (16-byte loads along rows), chosen by the launcher: buffer loads with a
// per-item byte offset that advances by one K-step; an out-of-range item (row past the operand, k past this slice)
// gets an offset beyond the descriptor's extent and the hardware returns zeros.  No branch and no dependent select on
// the loaded value: the generic loader's control flow made the compiler wait for every load where it was issued
// (vmcnt(0) at each merge); here the loads of K-step s+2 stay in flight under the MFMAs of steps s and s+1.
typedef int v4i32 __attribute__((ext_vector_type(4)));
template <int R, int BK>
struct TileLoaderBase {
  static constexpr int ITEMS = R * BK / 4 / 256;
  float4 reg[2][ITEMS];
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned off[ITEMS];
  int kq[ITEMS];
  bool okr[ITEMS];
  unsigned kstep;
  __device__ __forceinline__ void make(const float* b, long extent_elems) {
    // raw buffer (stride 0), 32-bit data format; the launcher guarantees extent < 2^31 bytes
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, static_cast<int>(extent_elems * 4), 0x00020000);
  }
  template <int S>
  __device__ __forceinline__ void load(int k0, int K) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const bool ok = okr[it] && (k0 + kq[it] < K);
      const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? off[it] : 0x80000000u, 0, 0);
      reg[S][it] = __builtin_bit_cast(float4, v);
      off[it] += kstep;
    }
  }
};
template <int R, int BK>
struct TileLoader<R, 1, BK> : TileLoaderBase<R, BK> {
  using TileLoaderBase<R, BK>::ITEMS;
  static constexpr int SPR = BK / 4;               // 16-byte slots per row
  __device__ __forceinline__ void init(const float* b, long rs, long, int r0, int kb, int rows, int K) {
    this->make(b, (rows - 1) * rs + K);
    this->kstep = BK * 4;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = threadIdx.x + it * 256;
      const int gr = r0 + e / SPR;
      this->kq[it] = (e % SPR) * 4;
      this->okr[it] = gr < rows;
      this->off[it] = static_cast<unsigned>((this->okr[it] ? gr * rs : 0) + kb + this->kq[it]) * 4u;
    }
  }
  // LDS layout of a k-contiguous operand: [row][16 k] = 64-byte rows, the four 16-byte slots of a row XOR-swizzled
  // (swz_rk) so that both this store and the fragments' ds_read_b128 (lane = row l%32, slot 2*half + l/32) are
  // conflict free: one 16-byte store per loaded float4, one 16-byte read per FOUR MFMA k-pairs
  template <int S>
  __device__ __forceinline__ void store(float* lds) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = threadIdx.x + it * 256;
      *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(lds) + swz_rk<BK>(e / SPR, e % SPR)) = this->reg[S][it];
    }
  }
};
template <int R, int BK>
struct TileLoader<R, 2, BK> : TileLoaderBase<R, BK> {
  using TileLoaderBase<R, BK>::ITEMS;
  __device__ __forceinline__ void init(const float* b, long, long ks, int r0, int kb, int rows, int K) {
    this->make(b, (K - 1) * ks + rows);
    this->kstep = static_cast<unsigned>(BK * ks) * 4u;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = threadIdx.x + it * 256;
      const int k = e / (R / 4), gr = r0 + (e % (R / 4)) * 4;
      this->kq[it] = k;
      this->okr[it] = gr < rows;
      this->off[it] = static_cast<unsigned>((kb + k) * ks + (this->okr[it] ? gr : 0)) * 4u;
    }
  }
  template <int S>
  __device__ __forceinline__ void store(float* lds) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = threadIdx.x + it * 256;
      const int k = e / (R / 4), rq = (e % (R / 4)) * 4;
      *reinterpret_cast<float4*>(lds + k * (R + 4) + rq) = this->reg[S][it];
    }
  }
};

// PREC (tfmq_set_gemm_precision; the reconstruction iterations only -- everything that must be exact keeps 0):
//   0  v_mfma_f32_32x32x2f32: exact fp32 products (157 TFLOP/s matrix peak)
//   1  "bf16x3": every fp32 operand value split as hi = bf16(a), lo = bf16(a - hi) while it sits in registers between the LDS read and
//      the MFMA; a b ~ hi hi' + hi lo' + lo hi' on v_mfma_f32_32x32x16_bf16 (3 MFMAs per 16 k instead of 8 fp32 ones at twice the
//      cycles each: 5.3x fewer matrix-pipe cycles), fp32 accumulation.  Relative error per product <= 2^-16 (the dropped lo lo' term and
//      lo's own rounding) against 2^-24: SURVEY section 7-1's "split-bf16 operand MFMA, check loss-curve parity".
//   2  fp16 operands (one MFMA per 16 k), relative error per product 2^-11
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8hf __attribute__((ext_vector_type(8)));
template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES, int MA = 0, int MB = 0, int BK = 16, int PREC = 0>
__global__ __launch_bounds__(256, (WM_TILES * WN_TILES > 2) ? 2 : (BK == 32 ? 3 : (PREC ? 4 : 5))) void k_gemm_f32_mfma(GemmP p) {
  constexpr int BM = WAVES_M * WM_TILES * 32, BN = WAVES_N * WN_TILES * 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  __shared__ __attribute__((aligned(128))) float sA[2][BK * (BM + 4)];
  __shared__ __attribute__((aligned(128))) float sB[2][BK * (BN + 4)];
  // split-K (skinny outputs with a long reduction: weight gradients dW = X^T dY, K = B*T): blockIdx.z also carries the
  // K slice; slices write raw partial tiles, k_gemm_splitk_reduce adds them in slice order (deterministic)
  int bz = blockIdx.z, kb = 0, ke = p.K;
  if (p.ksplit > 1) {
    const int sp = bz % p.ksplit;
    bz /= p.ksplit;
    kb = sp * p.kchunk;
    ke = kb + p.kchunk < p.K ? kb + p.kchunk : p.K;
  }
  const float* A = p.A + p.off_a(bz);
  const float* B = p.B + p.off_b(bz);
  float* C = p.C + p.off_c(bz);
  // XCD-aware tile order: the dispatcher places block b on XCD b % 8; each XCD gets a contiguous range of tiles (column
  // tiles fastest), so the column tiles that share an A row panel -- and the row tiles that share B -- meet in ONE L2
  // instead of fetching the panel into all eight.  Placement only: any order gives the same result.
  int tile = blockIdx.x;
  if (p.tiles_n > 0) {
    const int nb = gridDim.x, xcd = tile & 7, q = nb >> 3, r = nb & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (tile >> 3);
  }
  const int tn = p.tiles_n > 0 ? p.tiles_n : -p.tiles_n;
  const int m0 = (tile / tn) * BM, n0 = (tile % tn) * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int l32 = lane & 31, hh = lane >> 5;

  TileLoader<BM, MA, BK> la;
  TileLoader<BN, MB, BK> lb;
  la.init(A, p.sam, p.sak, m0, kb, p.M, p.K);
  lb.init(B, p.sbn, p.sbk, n0, kb, p.N, p.K);

  v16f acc[WM_TILES][WN_TILES];
#pragma unroll
  for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
    for (int j = 0; j < WN_TILES; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = (ke - kb + BK - 1) / BK;
  la.template load<0>(kb, ke);
  lb.template load<0>(kb, ke);
  la.template load<1>(kb + BK, ke);
  lb.template load<1>(kb + BK, ke);
  la.template store<0>(sA[0]);
  lb.template store<0>(sB[0]);
  __syncthreads();
  // K-step s multiplies LDS buffer s & 1.  Its loads were issued two steps earlier (register set s & 1), stored to LDS
  // during step s - 1; the set it frees takes the loads of step s + 2 (an L2 miss has two steps to land: with one
  // step of lookahead the mid-step store waited on HBM).  Past the last K-step every item is out of range: zeros,
  // stored to the buffer nobody reads again -- no branch around loads or stores.
  auto step = [&](int s, auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    const int buf = PAR;
#ifndef TFMQ_DBG_GEMM_NO_LOAD      // diagnostics build (results are garbage): the K loop without its global loads
    la.template load<PAR>(kb + (s + 2) * BK, ke);
    lb.template load<PAR>(kb + (s + 2) * BK, ke);
    __builtin_amdgcn_sched_barrier(0);   // the loads stay first in the step (the scheduler sank them below the MFMAs)
#endif
    const float* a_l = sA[buf] + (wm * WM_TILES * 32) + l32;
    const float* b_l = sB[buf] + (wn * WN_TILES * 32) + l32;
    const unsigned char* a_b = reinterpret_cast<const unsigned char*>(sA[buf]);
    const unsigned char* b_b = reinterpret_cast<const unsigned char*>(sB[buf]);
    // an MFMA multiplies the k-pair (lanes 0-31: first k, lanes 32-63: second k); any pairing is a valid order of the
    // k sum as long as A and B use the same one: MFMA j of half h takes k = 8h + j (lanes 0-31) and 8h + 4 + j
    // (lanes 32-63), so a k-contiguous operand feeds four MFMAs from one 16-byte read
    auto frag_a = [&](int i, int half, float (&o)[4]) {
      if constexpr (MA == 1) {
        const float4 v = *reinterpret_cast<const float4*>(a_b + swz_rk<BK>((wm * WM_TILES + i) * 32 + l32, half * 2 + hh));
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a_l[(half * 8 + hh * 4 + j) * (BM + 4) + i * 32];
      }
    };
    auto frag_b = [&](int jn, int half, float (&o)[4]) {
      if constexpr (MB == 1) {
        const float4 v = *reinterpret_cast<const float4*>(b_b + swz_rk<BK>((wn * WN_TILES + jn) * 32 + l32, half * 2 + hh));
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = b_l[(half * 8 + hh * 4 + j) * (BN + 4) + jn * 32];
      }
    };
    if constexpr (PREC == 0) {
#pragma unroll
    for (int half = 0; half < BK / 8; ++half) {
      float af[WM_TILES][4], bf[WN_TILES][4];
#pragma unroll
      for (int i = 0; i < WM_TILES; ++i) frag_a(i, half, af[i]);
#pragma unroll
      for (int jn = 0; jn < WN_TILES; ++jn) frag_b(jn, half, bf[jn]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
          for (int jn = 0; jn < WN_TILES; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j], bf[jn][j], acc[i][jn], 0, 0, 0);
#ifndef TFMQ_DBG_GEMM_NO_STORE
      // the next K-step's tile goes to the other buffer (nobody reads it during this step) under the second half's
      // MFMAs: its loads were issued a half step ago, and the barrier below then has no LDS write left to wait for
      if (half == BK / 16 - 1) {
        la.template store<PAR ^ 1>(sA[buf ^ 1]);
        lb.template store<PAR ^ 1>(sB[buf ^ 1]);
      }
#endif
    }
    } else {
    // 16 k per MFMA: the 8 values a lane holds of two consecutive "halves" (k = 8h + 4hh + j) are ONE operand of the 16-wide MFMA --
    // any assignment of k to operand slots is a valid order of the k sum as long as A and B use the same one
#pragma unroll
    for (int grp = 0; grp < BK / 16; ++grp) {
      float a8[WM_TILES][8], b8[WN_TILES][8];
#pragma unroll
      for (int i = 0; i < WM_TILES; ++i) {
        float t0[4], t1[4];
        frag_a(i, 2 * grp, t0);
        frag_a(i, 2 * grp + 1, t1);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a8[i][j] = t0[j]; a8[i][4 + j] = t1[j]; }
      }
#pragma unroll
      for (int jn = 0; jn < WN_TILES; ++jn) {
        float t0[4], t1[4];
        frag_b(jn, 2 * grp, t0);
        frag_b(jn, 2 * grp + 1, t1);
#pragma unroll
        for (int j = 0; j < 4; ++j) { b8[jn][j] = t0[j]; b8[jn][4 + j] = t1[j]; }
      }
      if constexpr (PREC == 1) {
        v8bf ah[WM_TILES], al[WM_TILES], bh[WN_TILES], bl[WN_TILES];
#pragma unroll
        for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            ah[i][e] = static_cast<__bf16>(a8[i][e]);
            al[i][e] = static_cast<__bf16>(a8[i][e] - static_cast<float>(ah[i][e]));
          }
#pragma unroll
        for (int jn = 0; jn < WN_TILES; ++jn)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            bh[jn][e] = static_cast<__bf16>(b8[jn][e]);
            bl[jn][e] = static_cast<__bf16>(b8[jn][e] - static_cast<float>(bh[jn][e]));
          }
#pragma unroll
        for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
          for (int jn = 0; jn < WN_TILES; ++jn) {
            // small terms first: they are not absorbed by the large partial sum's rounding
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[jn], acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[jn], acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[jn], acc[i][jn], 0, 0, 0);
          }
      } else {
        v8hf ah[WM_TILES], bh[WN_TILES];
#pragma unroll
        for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) ah[i][e] = static_cast<_Float16>(a8[i][e]);
#pragma unroll
        for (int jn = 0; jn < WN_TILES; ++jn)
#pragma unroll
          for (int e = 0; e < 8; ++e) bh[jn][e] = static_cast<_Float16>(b8[jn][e]);
#pragma unroll
        for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
          for (int jn = 0; jn < WN_TILES; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[jn], acc[i][jn], 0, 0, 0);
      }
#ifndef TFMQ_DBG_GEMM_NO_STORE
      if (grp == 0) {
        la.template store<PAR ^ 1>(sA[buf ^ 1]);
        lb.template store<PAR ^ 1>(sB[buf ^ 1]);
      }
#endif
    }
    }
    __syncthreads();
  };
  for (int s = 0; s < nk; s += 2) {
    step(s, std::integral_constant<int, 0>{});
    if (s + 1 < nk) step(s + 1, std::integral_constant<int, 1>{});
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
        if (p.residual) v += p.residual[p.off_c(bz) + m * p.scm + n];
        float* c = C + m * p.scm + n;
        *c = p.accumulate ? *c + v : v;
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K15b (round 5): the bf16x3 GEMM with the hi / lo split done ONCE PER BLOCK, on the way into LDS.
//
// k_gemm_f32_mfma<..., PREC = 1> keeps fp32 tiles in LDS and splits every fragment in registers in front of its MFMAs: with 1 x 2 MFMA tiles per
// wave that is 72 conversion instructions beside 6 MFMAs per K-step -- 12 VALU per MFMA where about four hide -- and the
// kernel ran at 165-195 TFLOP/s = 0.20-0.23 of the 833 TFLOP/s three bf16 MFMAs per product allow.  Here
//   * a 128 x 128 tile on 4 waves, 2 x 2 MFMA tiles per wave, K-step 32 (two MFMA k-steps): 24 MFMAs per wave and step;
//   * the loading thread splits its 16 values per operand and step (hi = bf16(a), lo = bf16(a - hi): v_cvt_pk_bf16_f32, shift, subtract,
//     v_cvt_pk_bf16_f32 = 3 instructions per value, 96 per thread and step beside the 24 MFMAs) and stores the two planes of a row side by
//     side: LDS row = [hi k0..31 | lo k0..31] = 128 bytes, the eight 16-byte slots XOR-swizzled with row bits 0-2 (swz_rk<32>'s scheme);
//   * a fragment is one ds_read_b128 per plane (8 bf16 = the k-octet lane half hh owes the 16-wide MFMA): 16 reads per 24 MFMAs, no VALU;
//   * loads as in the kernel above: 16-byte buffer loads two K-steps ahead in two register sets (out-of-range items read zeros), along k
//     (MODE 1: a thread owns two k-octets of a row -> two ds_write_b128 per plane) or along rows (MODE 2: a thread owns 4 rows x 4 k, the
//     4 x 4 transpose is register naming -> four ds_write_b64 per plane).
// Same split, same three products per k-octet (small terms first), fp32 accumulation: the error bound of PREC = 1 (2^-16 per product);
// the k order inside a 16-wide MFMA differs from the kernel above, so results agree to fp32 rounding, not bit for bit.
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bx3_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const v2bf h = __builtin_convertvector(v2f{x0, x1}, v2bf);
  hi = __builtin_bit_cast(unsigned, h);
  const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
  const v2bf l = __builtin_convertvector(v2f{x0 - h0, x1 - h1}, v2bf);
  lo = __builtin_bit_cast(unsigned, l);
}
// byte offset of 16-byte slot `slot` (0-3 hi octets, 4-7 lo octets) of row `row` in a [128 rows][128 B] bx3 tile
// The swizzle term f(row) = ((row >> 1) & 7) ^ ((row & 1) << 2): the 16 rows of a ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...; banks =
// (a / 4) mod 64, two 128-byte rows per bank line) land on 16 distinct 16-byte positions, and the two rows an 8-lane ds_write_b128 group
// touches (2 rows x 4 octets; banks = (a / 4) mod 32) use disjoint slot sets -- row & 7 alone is two-way conflicted on both sides.
__device__ __forceinline__ int bx3_off(int row, int slot) { return row * 128 + ((slot ^ (((row >> 1) & 7) ^ ((row & 1) << 2))) << 4); }

template <int MODE, int R = 128>
struct Bx3Loader {
  // tile: R rows x 32 k.  MODE 1: units (row, k-octet), R * 4 of them -> R / 64 per thread; MODE 2: units (row quad, k quad), 2 R of them ->
  // one per thread at R = 128; at R = 64 threads 128 .. 255 repeat the work of threads 0 .. 127 (same loads, same values to the same LDS bytes)
  static constexpr int NU = MODE == 1 ? R / 64 : 1;
  static constexpr int NI = MODE == 1 ? 2 * NU : 4;      // 16-byte items per thread and K-step
  static constexpr int RQ = R / 4;                       // row quads (MODE 2)
  float4 reg[2][NI];
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned off[NU];                                // byte offset of the unit at the slice's first K-step; 0x80000000 (out of range: zeros) for rows past the operand
  unsigned kstep, kone;                            // bytes per K-step; MODE 2: bytes per k
  __device__ __forceinline__ int m2_rq() const { return (threadIdx.x & (2 * R - 1)) & (RQ - 1); }
  __device__ __forceinline__ int m2_kq4() const { return (threadIdx.x & (2 * R - 1)) / RQ; }
  __device__ __forceinline__ void init(const float* b, long rs, long ks, int r0, int kb, int rows, int K) {
    const int tid = threadIdx.x;
    if constexpr (MODE == 1) {
      rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, static_cast<int>(((rows - 1) * rs + K) * 4), 0x00020000);
      kstep = 32 * 4;
      kone = 4;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int e = tid + 256 * u, gr = r0 + (e >> 2);
        off[u] = gr < rows ? static_cast<unsigned>(gr * rs + kb + (e & 3) * 8) * 4u : 0x80000000u;
      }
    } else {
      rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, static_cast<int>(((K - 1) * ks + rows) * 4), 0x00020000);
      kstep = static_cast<unsigned>(32 * ks) * 4u;
      kone = static_cast<unsigned>(ks) * 4u;
      const int gr = r0 + m2_rq() * 4;
      off[0] = gr < rows ? static_cast<unsigned>((kb + m2_kq4() * 4) * ks + gr) * 4u : 0x80000000u;
    }
  }
  // k (relative to the K-step's first k) of the first value of item j
  __device__ __forceinline__ int item_k(int j) const {
    if constexpr (MODE == 1) return (threadIdx.x & 3) * 8 + (j & 1) * 4;
    else return m2_kq4() * 4 + j;
  }
  // K-step `step` of the slice.  CHECKED = false: every k of the step is inside the slice -- the per-thread offsets never change, the step
  // rides in the SCALAR offset of the buffer instruction (no VALU address arithmetic in the loop; with it the compiler put the temporaries
  // into the destination registers of the loads still in flight and waited vmcnt(0) at the top of every step).  CHECKED = true (the last
  // steps of a slice and the prefetches past it): items whose k lies outside get the out-of-range offset and read zeros.
  template <int S, bool CHECKED>
  __device__ __forceinline__ void load(int step, int klen) {
    const unsigned sb = static_cast<unsigned>(step) * kstep;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int u = MODE == 1 ? (j >> 1) : 0;
      const unsigned o = off[u] + (MODE == 1 ? static_cast<unsigned>((j & 1) * 16) : static_cast<unsigned>(j) * kone);
      v4i32 v;
      if constexpr (CHECKED) {
        const bool ok = step * 32 + item_k(j) < klen;
        v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? o + sb : 0x80000000u, 0, 0);
      } else {
        v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, static_cast<int>(sb), 0);
      }
      reg[S][j] = __builtin_bit_cast(float4, v);
    }
  }
  template <int S>
  __device__ __forceinline__ void store(unsigned char* lds) {
    const int tid = threadIdx.x;
    if constexpr (MODE == 1) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int e = tid + 256 * u, row = e >> 2, oct = e & 3;
        const float4 a = reg[S][2 * u], b = reg[S][2 * u + 1];
        uint4 hi, lo;
        bx3_split2(a.x, a.y, hi.x, lo.x);
        bx3_split2(a.z, a.w, hi.y, lo.y);
        bx3_split2(b.x, b.y, hi.z, lo.z);
        bx3_split2(b.z, b.w, hi.w, lo.w);
        *reinterpret_cast<uint4*>(lds + bx3_off(row, oct)) = hi;
        *reinterpret_cast<uint4*>(lds + bx3_off(row, 4 + oct)) = lo;
      }
    } else {
      const int rq = m2_rq(), kq4 = m2_kq4();
      const float r[4][4] = {{reg[S][0].x, reg[S][1].x, reg[S][2].x, reg[S][3].x}, {reg[S][0].y, reg[S][1].y, reg[S][2].y, reg[S][3].y},
                             {reg[S][0].z, reg[S][1].z, reg[S][2].z, reg[S][3].z}, {reg[S][0].w, reg[S][1].w, reg[S][2].w, reg[S][3].w}};
#pragma unroll
      for (int ri = 0; ri < 4; ++ri) {
        const int row = rq * 4 + ri;
        uint2 hi, lo;
        bx3_split2(r[ri][0], r[ri][1], hi.x, lo.x);
        bx3_split2(r[ri][2], r[ri][3], hi.y, lo.y);
        *reinterpret_cast<uint2*>(lds + bx3_off(row, kq4 >> 1) + (kq4 & 1) * 8) = hi;
        *reinterpret_cast<uint2*>(lds + bx3_off(row, 4 + (kq4 >> 1)) + (kq4 & 1) * 8) = lo;
      }
    }
  }
};

// BN = 128: 2 x 2 waves of 64 x 64 (the long-K form); BN = 64: 4 x 1 waves of 32 x 64, 48 KiB of LDS and three blocks per CU -- the short-K
// form (K = 320 ... 1280: the Linears of the transformer units), 6 splitting instructions per MFMA against the 12 of k_gemm_f32_mfma<PREC = 1>
template <int MA, int MB, int BN = 128>
__global__ __launch_bounds__(256, BN == 128 ? 2 : 3) void k_gemm_bx3(GemmP p) {
  constexpr int BM = 128, BK = 32, TILE = 128 * 128, TILE_B = BN * 128;
  constexpr int WMT = BN == 128 ? 2 : 1, WNT = 2;     // MFMA tiles per wave along M / N
  __shared__ __attribute__((aligned(128))) unsigned char smem[2 * TILE + 2 * TILE_B];      // A[2] | B[2]: 64 KiB (two blocks per CU) / 48 KiB (three)
  int bz = blockIdx.z, kb = 0, ke = p.K;
  if (p.ksplit > 1) {
    const int sp = bz % p.ksplit;
    bz /= p.ksplit;
    kb = sp * p.kchunk;
    ke = kb + p.kchunk < p.K ? kb + p.kchunk : p.K;
  }
  const float* A = p.A + p.off_a(bz);
  const float* B = p.B + p.off_b(bz);
  float* C = p.C + p.off_c(bz);
  int tile = blockIdx.x;
  if (p.tiles_n > 0) {      // XCD-aware tile order (as above)
    const int nb = gridDim.x, xcd = tile & 7, q = nb >> 3, r = nb & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (tile >> 3);
  }
  const int tn = p.tiles_n > 0 ? p.tiles_n : -p.tiles_n;
  const int m0 = (tile / tn) * BM, n0 = (tile % tn) * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = BN == 128 ? (wid >> 1) : wid, wn = BN == 128 ? (wid & 1) : 0;
  const int l32 = lane & 31, hh = lane >> 5;

  Bx3Loader<MA, 128> la;
  Bx3Loader<MB, BN> lb;
  la.init(A, p.sam, p.sak, m0, kb, p.M, p.K);
  lb.init(B, p.sbn, p.sbk, n0, kb, p.N, p.K);

  v16f acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int klen = ke - kb, nk = (klen + BK - 1) / BK, nfull = klen / BK;
  la.template load<0, true>(0, klen);
  lb.template load<0, true>(0, klen);
  la.template load<1, true>(1, klen);
  lb.template load<1, true>(1, klen);
  la.template store<0>(smem);
  lb.template store<0>(smem + 2 * TILE);
  __syncthreads();
  // fragment rows of this lane (row bits 0-3 = l32 bits 0-3 for every tile: one swizzle term)
  const int ra = (wm * WMT) * 32 + l32, rb = (wn * WNT) * 32 + l32;
  auto step = [&](int s, auto par_tag, auto chk_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    constexpr bool CHK = decltype(chk_tag)::value;
#ifndef TFMQ_DBG_GEMM_NO_LOAD
    la.template load<PAR, CHK>(s + 2, klen);
    lb.template load<PAR, CHK>(s + 2, klen);
    __builtin_amdgcn_sched_barrier(0);   // the loads stay first in the step
#endif
    const unsigned char* a_b = smem + PAR * TILE;
    const unsigned char* b_b = smem + 2 * TILE + PAR * TILE_B;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v8bf ah[WMT], al[WMT], bh[WNT], bl[WNT];
#pragma unroll
      for (int i = 0; i < WMT; ++i) {
        ah[i] = *reinterpret_cast<const v8bf*>(a_b + bx3_off(ra + 32 * i, 2 * ks + hh));
        al[i] = *reinterpret_cast<const v8bf*>(a_b + bx3_off(ra + 32 * i, 4 + 2 * ks + hh));
      }
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        bh[i] = *reinterpret_cast<const v8bf*>(b_b + bx3_off(rb + 32 * i, 2 * ks + hh));
        bl[i] = *reinterpret_cast<const v8bf*>(b_b + bx3_off(rb + 32 * i, 4 + 2 * ks + hh));
      }
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
#ifdef TFMQ_DBG_GEMM_NO_MFMA      // diagnostics build (results are garbage): the K loop without its MFMAs
          acc[i][j][0] += static_cast<float>(al[i][0]) + static_cast<float>(bh[j][1]) + static_cast<float>(ah[i][2]) + static_cast<float>(bl[j][3]);
#else
          // small terms first: they are not absorbed by the large partial sum's rounding
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#endif
        }
#ifndef TFMQ_DBG_GEMM_NO_STORE
      // the next K-step's tile (loaded one step ago) is split and stored to the other buffer under the second half's MFMAs
      if (ks == 0) {
        la.template store<PAR ^ 1>(smem + (PAR ^ 1) * TILE);
        lb.template store<PAR ^ 1>(smem + 2 * TILE + (PAR ^ 1) * TILE_B);
      }
#endif
    }
    __syncthreads();
  };
  int s = 0;
  for (; s + 3 < nfull; s += 2) {        // both steps' prefetches (s + 2, s + 3) are whole K-steps of the slice
    step(s, std::integral_constant<int, 0>{}, std::false_type{});
    step(s + 1, std::integral_constant<int, 1>{}, std::false_type{});
  }
  for (; s < nk; s += 2) {
    step(s, std::integral_constant<int, 0>{}, std::true_type{});
    if (s + 1 < nk) step(s + 1, std::integral_constant<int, 1>{}, std::true_type{});
  }

  // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)   (the epilogue of k_gemm_f32_mfma)
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int n = n0 + (wn * WNT + j) * 32 + l32;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * WMT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (m >= p.M) continue;
        if (p.ksplit > 1) {
          p.partial[(static_cast<size_t>(blockIdx.z % p.ksplit) * gridDim.z / p.ksplit + bz) * p.M * p.N +
                    static_cast<size_t>(m) * p.N + n] = acc[i][j][r];
          continue;
        }
        float v = p.alpha * acc[i][j][r];
        if (p.bias) v += bv;
        if (p.rowadd) v += p.rowadd[static_cast<long>(m / p.rows_per_img) * p.rowadd_ld + n];
        if (p.residual) v += p.residual[p.off_c(bz) + m * p.scm + n];
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
    if (p.residual) v += p.residual[p.off_c(bz) + m * p.scm + n];
    float* c = p.C + p.off_c(bz) + m * p.scm + n;
    *c = p.accumulate ? *c + v : v;
  }
}

// called from tfmq_gemm_f32 (recon_kernels.hip) when the problem is large enough for 128-row tiles
int tfmq_gemm_f32_mfma_launch(tfmq_handle h, GemmP& p, int batch, hipStream_t st) {
  const int M = p.M, N = p.N;
  // loader modes (the rules of the generic loader, evaluated once here; batch / split offsets keep the alignment
  // only if the strides do, which the rules check through bsa / bsb and kchunk % 16 == 0)
  auto mode_of = [&](const float* base, long rs, long ks, int rows, long bs) {
    // (+ buffer addressing of the fast loaders: one batch item's extent below 2^31 bytes, offsets in 32 bits)
    const bool al = (reinterpret_cast<uintptr_t>(base) & 15) == 0 && (bs & 3) == 0 && rs >= 0 && ks >= 0 &&
                    ((rows - 1) * rs + (static_cast<long>(p.K) - 1) * ks + 1) * 4 + 64L * (rs > ks ? rs : ks) < (1L << 31);
    if (ks == 1 && al && (rs & 3) == 0 && (p.K & 3) == 0) return 1;
    if (rs == 1 && al && (ks & 3) == 0 && (rows & 3) == 0) return 2;
    return 0;
  };
  int ma = mode_of(p.A, p.sam, p.sak, M, batch > 1 ? (p.bsa | p.bsa2) : 0),
      mb = mode_of(p.B, p.sbn, p.sbk, N, batch > 1 ? (p.bsb | p.bsb2) : 0);
  if (getenv("TFMQ_GEMM_GENERIC_LOADER")) ma = mb = 0;
  const int prec = h->gemm_prec;
  // round 5: bf16x3 operands with 16-byte loads on both sides and a LONG reduction take k_gemm_bx3 (hi / lo split once per block, 128 x 128
  // tiles): +20 ... 28 % on the 3x3-conv forward shapes (K = 2880 ... 11520), +6 % on the weight-gradient shape (K = 32768); with K = 320
  // (ten K-steps per tile) the smaller tiles of k_gemm_f32_mfma<PREC = 1> and their five blocks per CU win by 10-25 % and keep those
  // launches (same-box A/B: profiles/r05_ab_gemm_bx3.txt).  TFMQ_GEMM_BX3=0 switches the kernel off, =2 forces it for every K >= 64
  // (A/B runs, tests).  Skinny outputs (N <= 64: the per-head attention products) stay on the 128 x 64 tiles.
  static const int bx3_mode = getenv("TFMQ_GEMM_BX3") ? atoi(getenv("TFMQ_GEMM_BX3")) : 1;
  const bool bx3_ok = bx3_mode != 0 && prec == 1 && ma >= 1 && mb >= 1 && N > 64 && M > 64 && p.K >= 64;
  const bool bx3 = bx3_ok && p.K >= (bx3_mode == 2 ? 64 : 1024);
  // the short-K form (128 x 64 tiles, three blocks per CU) for what is left: TFMQ_GEMM_BX3=3 (A/B, round 5)
  const bool bx3s = bx3_ok && !bx3 && bx3_mode == 3;
  // 128 x 64 tiles (4 waves along M) measured faster than 128 x 128 at every SD unit shape (no column waste at
  // N = 320 / 640, twice the blocks for the mid-sized problems); TFMQ_GEMM_BN128 keeps the wide tile for A/B runs
  const int BN = bx3 ? 128 : ((N > 64 && getenv("TFMQ_GEMM_BN128")) ? 128 : 64);
  // tiles of ONE batch item: the slicing (hence the summation order) must not depend on how many items share the
  // launch -- results stay bit-identical whatever else is in the batch
  const long tiles = static_cast<long>((N + BN - 1) / BN) * ((M + 127) / 128);
  // Split K so that the blocks divide evenly over the CUs.  All blocks of these problems are resident at once (up to 5
  // per CU), a CU's time grows with the number it holds, and 640 tiles on 256 CUs leave half the chip with 3 and half
  // with 2: 83 % busy.  In units of one tile's K loop a launch takes ceil(tiles * ks / CUs) / ks; the slices' extra
  // pass (write + re-read of ks partial outputs) costs about 66 (ks + 1) / K of the GEMM.  Smallest cost wins, ties to
  // the smaller ks; every slice keeps >= 256 k.  PMC (8192 x 640 x 5760): MFMA busy 71 % unsplit.
  int ks = 1;
  if (p.K >= 512 && !getenv("TFMQ_GEMM_NO_SPLITK")) {
    const double cu = h->cu_count, t = static_cast<double>(tiles);
    double best = 0.0;
    for (int c = 1; c <= 32 && p.K / c >= 256; ++c) {
      const double rounds = static_cast<double>((tiles * c + h->cu_count - 1) / h->cu_count) / c;
      const double cost = rounds + (c > 1 ? 66.0 * (c + 1) / p.K * (t / cu) : 0.0);
      if (c == 1 || cost < best * 0.97) {     // a split has to buy at least 3 %
        best = cost;
        ks = c;
      }
    }
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
  p.tiles_n = (N + BN - 1) / BN;
  if (getenv("TFMQ_GEMM_NO_XCD")) p.tiles_n = -p.tiles_n;     // A/B runs: plain row-major tile order
  dim3 grid(((N + BN - 1) / BN) * ((M + 127) / 128), 1, batch * (ks > 1 ? ks : 1));
  const bool bk32 = getenv("TFMQ_GEMM_BK32") != nullptr;
  if (bx3s) {
    if (ma == 1 && mb == 1) hipLaunchKernelGGL((k_gemm_bx3<1, 1, 64>), grid, dim3(256), 0, st, p);
    else if (ma == 1 && mb == 2) hipLaunchKernelGGL((k_gemm_bx3<1, 2, 64>), grid, dim3(256), 0, st, p);
    else if (ma == 2 && mb == 1) hipLaunchKernelGGL((k_gemm_bx3<2, 1, 64>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_bx3<2, 2, 64>), grid, dim3(256), 0, st, p);
  } else
  if (bx3) {
    if (ma == 1 && mb == 1) hipLaunchKernelGGL((k_gemm_bx3<1, 1>), grid, dim3(256), 0, st, p);
    else if (ma == 1 && mb == 2) hipLaunchKernelGGL((k_gemm_bx3<1, 2>), grid, dim3(256), 0, st, p);
    else if (ma == 2 && mb == 1) hipLaunchKernelGGL((k_gemm_bx3<2, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_bx3<2, 2>), grid, dim3(256), 0, st, p);
  } else
  if (prec != 0 && BN == 128 && !bk32) {
    // (A/B runs, TFMQ_GEMM_BN128=1) 128 x 128 tiles, 2 x 2 MFMA tiles per wave: half the hi / lo splitting work per MFMA of the 1 x 2 form
#define TFMQ_GEMM_PREC_W(PR)                                                                                                      \
    if (ma == 1 && mb == 1) hipLaunchKernelGGL((k_gemm_f32_mfma<2, 2, 2, 2, 1, 1, 16, PR>), grid, dim3(256), 0, st, p);           \
    else if (ma == 1 && mb == 2) hipLaunchKernelGGL((k_gemm_f32_mfma<2, 2, 2, 2, 1, 2, 16, PR>), grid, dim3(256), 0, st, p);      \
    else if (ma == 2 && mb == 1) hipLaunchKernelGGL((k_gemm_f32_mfma<2, 2, 2, 2, 2, 1, 16, PR>), grid, dim3(256), 0, st, p);      \
    else if (ma == 2 && mb == 2) hipLaunchKernelGGL((k_gemm_f32_mfma<2, 2, 2, 2, 2, 2, 16, PR>), grid, dim3(256), 0, st, p);      \
    else hipLaunchKernelGGL((k_gemm_f32_mfma<2, 2, 2, 2, 0, 0, 16, PR>), grid, dim3(256), 0, st, p);
    if (prec == 1) { TFMQ_GEMM_PREC_W(1) } else { TFMQ_GEMM_PREC_W(2) }
#undef TFMQ_GEMM_PREC_W
  } else
  if (prec != 0 && BN == 64 && !bk32) {
#define TFMQ_GEMM_PREC(PR)                                                                                                        \
    if (ma == 1 && mb == 1) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 1, 1, 16, PR>), grid, dim3(256), 0, st, p);           \
    else if (ma == 1 && mb == 2) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 1, 2, 16, PR>), grid, dim3(256), 0, st, p);      \
    else if (ma == 2 && mb == 1) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 2, 1, 16, PR>), grid, dim3(256), 0, st, p);      \
    else if (ma == 2 && mb == 2) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 2, 2, 16, PR>), grid, dim3(256), 0, st, p);      \
    else hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 0, 0, 16, PR>), grid, dim3(256), 0, st, p);
    if (prec == 1) { TFMQ_GEMM_PREC(1) } else { TFMQ_GEMM_PREC(2) }
#undef TFMQ_GEMM_PREC
  } else
  if (BN == 128 && ma == 1 && mb == 1) hipLaunchKernelGGL((k_gemm_f32_mfma<2, 2, 2, 2, 1, 1>), grid, dim3(256), 0, st, p);
  else if (BN == 128 && ma == 1 && mb == 2) hipLaunchKernelGGL((k_gemm_f32_mfma<2, 2, 2, 2, 1, 2>), grid, dim3(256), 0, st, p);
  else if (BN == 128 && ma == 2 && mb == 2) hipLaunchKernelGGL((k_gemm_f32_mfma<2, 2, 2, 2, 2, 2>), grid, dim3(256), 0, st, p);
  else if (BN == 128) hipLaunchKernelGGL((k_gemm_f32_mfma<2, 2, 2, 2>), grid, dim3(256), 0, st, p);
  else if (bk32 && ma == 1 && mb == 1) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 1, 1, 32>), grid, dim3(256), 0, st, p);
  else if (bk32 && ma == 1 && mb == 2) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 1, 2, 32>), grid, dim3(256), 0, st, p);
  else if (bk32 && ma == 2 && mb == 2) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 2, 2, 32>), grid, dim3(256), 0, st, p);
  else if (ma == 1 && mb == 1) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 1, 1>), grid, dim3(256), 0, st, p);
  else if (ma == 1 && mb == 2) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 1, 2>), grid, dim3(256), 0, st, p);
  else if (ma == 2 && mb == 1) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 2, 1>), grid, dim3(256), 0, st, p);
  else if (ma == 2 && mb == 2) hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2, 2, 2>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_gemm_f32_mfma<4, 1, 1, 2>), grid, dim3(256), 0, st, p);
  if (ks > 1) {
    const size_t total = static_cast<size_t>(M) * N * batch;
    hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, p, batch);
  }
  return TFMQ_OK;
}
