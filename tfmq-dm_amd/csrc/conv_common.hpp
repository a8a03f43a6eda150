// Shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_slab.hip).
#pragma once
#include "common.hpp"
#ifdef TFMQ_PHASE_TIMERS
#include <cstdio>
#endif

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

struct ConvP {
  tfmq_conv_desc d;
  int M;        // B*Ho*Wo
  int chunks;   // K-steps per tap
  int nsteps;   // KH*KW*chunks
  int Ktot;     // KH*KW*Cin
  int cin_pad;  // f16 path: padded Cin of the weight layout
  int Hv, Wv;   // virtual input size (2H,2W when up2x)
  int tiles_n;
  int cout_pad;                    // w4a8: rows of the expanded weight operand (multiple of 32)
  const unsigned char* pad_table;  // 256 x 64 B, row v = byte v (tfmq_ctx::pad_table)
  int ksplit;                      // >= 1: workgroups per output tile (k_conv_dma, w4a8)
  int* ks_ws;                      // tfmq_ctx::ksplit_ws: [tile][slice][BM * BN] int32 partial sums
  int* ks_cnt;                     // tfmq_ctx::ksplit_cnt: arrival tickets, zero between launches
  int issue_split;                 // k_lin_direct, TFMQ_LIN_ISSUE_SPLIT=1 (round 6 A/B): waves 2-3 issue their LDS-DMA pieces behind the first half's MFMAs
#ifdef TFMQ_PHASE_TIMERS
  unsigned long long* dbg;         // [blocks][4] shader-clock stamps: start, loop start, loop end, end
  unsigned long long* dbg2;        // [blocks][8] shader cycles wave 0 spent in the parts of its K loop (k_lin_direct)
#endif
};

// Diagnostics build (TFMQ_EXTRA_HIPCC_FLAGS=-DTFMQ_PHASE_TIMERS python tfmq-dm_amd/build.py): every w4a8 DMA launch
// is followed by a device sync and prints the mean cycles a block spends in prologue / K loop / epilogue.
#ifdef TFMQ_PHASE_TIMERS
#define TFMQ_MARK(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[blockIdx.x * 4 + (i)] = clock64(); } while (0)
#else
#define TFMQ_MARK(i) do { } while (0)
#endif

// Workgroup barrier that only waits for LDS traffic.  __syncthreads() also drains vmcnt(0), i.e. it would wait for
// the global prefetch loads of the NEXT K-steps at every barrier.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ int swz(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// One LDS-DMA wave instruction: lane l moves 16 bytes from its own global pointer to LDS byte lds_dst + 16*l
// (the destination is wave-uniform base + lane*16; M0 carries the base and is restored afterwards).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// The same with a uniform base (SGPR pair) and a per-lane 32-bit byte offset: half the address VGPRs of the 64-bit form.
__device__ __forceinline__ void glds16_sv(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

// XCD-aware tile order (T1): the dispatcher places block b on XCD b % 8; give each XCD a contiguous
// range of tiles so the 3x3 taps / neighbouring rows of one image hit that XCD's private L2.
// Bijective for any grid size; placement is a speed matter only.
__device__ __forceinline__ int xcd_tile_id() {
  const int bid = blockIdx.x, nb = gridDim.x, xcd = bid & 7, q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}


__device__ __forceinline__ unsigned pack_q4(float a, float b, float c, float e, float2 qp) {
  return quant_pack4(a, b, c, e, make_quantp(qp));
}
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
  const __half2 v = __floats2half2_rn(a, b);
  return *reinterpret_cast<const unsigned*>(&v);
}

// Channel owned by an accumulator register.  The weight rows of a 32-channel MFMA tile are read from LDS in a PERMUTED
// order (lin_brow: MFMA row i <- tile channel (i & 3) + 4 (i >> 3) + 16 ((i >> 2) & 1)), so that lane half h -- which owns
// the MFMA rows with (i >> 2) & 1 == h -- holds the 16 CONSECUTIVE channels 16h .. 16h+15 of the tile, register r = channel
// 16h + r.  A lane therefore moves 16 contiguous bytes per store / residual load (8 fp16 channels; 8 int8 channels = 8
// bytes): the write path of a CU retires roughly one touched 128-byte line per 4 cycles whatever the bytes, and 8-byte
// fp16 / 4-byte int8 pieces made the epilogue's stores the longest phase of a short-K tile.
// ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32 for the upper half): with this
// permutation each group still reads one row of every (row % 4, swizzle class) combination -- conflict-free.  (A first
// version that assumed groups of 16 consecutive lanes rotated the upper half's octets and cost the K loop 6-10 %.)
__device__ __forceinline__ int lin_brow(int i) {      // LDS row (tile channel) feeding MFMA row i of a 32-channel tile
  return (i & 3) + ((i >> 3) << 2) + (((i >> 2) & 1) << 4);
}

// DPP lane exchanges for the GroupNorm statistics of the register-direct epilogues (lane = pixel)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// Sum over 8 consecutive lanes (= 8 consecutive pixel rows) in the ONE order every kernel uses for the statistics:
// ((r0 + r1) + r2) + r3 added to ((r4 + r5) + r6) + r7.  Valid in the lanes with (lane & 7) == 0.
__device__ __forceinline__ float group8_sum(float v) {
  float a = v + dpp_f<0x55>(v);      // quad_perm [1,1,1,1]
  a = a + dpp_f<0xAA>(v);            // quad_perm [2,2,2,2]
  a = a + dpp_f<0xFF>(v);            // quad_perm [3,3,3,3]
  return a + dpp_f<0x104>(a);        // row_shl:4 -- lane i reads lane i + 4
}

// four consecutive channels of the residual tensor: fp32, or fp16 when it belongs to the fp16 activation stream
__device__ __forceinline__ float4 load_res4(const tfmq_conv_desc& d, int m, int n) {
  if (d.res_f16) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(d.residual) + static_cast<size_t>(m) * d.Cout + n);
    const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&u.x)), hi = __half22float2(*reinterpret_cast<const __half2*>(&u.y));
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  }
  return *reinterpret_cast<const float4*>(d.residual + static_cast<size_t>(m) * d.Cout + n);
}

// 3x3 / stride 1 / pad 1 w4a8 convolutions on the slab kernel (conv_slab.hip): true when the launch was taken
bool launch_conv_slab(tfmq_handle h, ConvP& p, hipStream_t st, bool forced, bool f16 = false, bool half_m = false);
// pointwise w4a8 layers with fp16 / int8 / GEGLU-int8 output on the register-direct-epilogue kernel (conv_lin.hip)
bool launch_conv_lin(tfmq_handle h, ConvP& p, hipStream_t st, bool m256 = false);      // m256: 256 x 128 tiles (TFMQ_TILE_DIRECT256), layers without a residual
// fp16-operand pointwise layers (tfmq_conv2d_f16 with x_f16) on the same kernel
bool launch_conv_lin_f16(tfmq_handle h, ConvP& p, hipStream_t st);
