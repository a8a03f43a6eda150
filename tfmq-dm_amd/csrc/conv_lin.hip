// K5p: pointwise w4a8 GEMM (token Linears and 1x1 convs) with a register-direct epilogue.
//
// These layers have short reductions (K = 320 ... 5120: 5 ... 80 K-steps) and large M: a block of the tile kernel
// (conv_igemm.hip: k_conv_dma) spends 60-75 % of its life in the LDS-staged epilogue (two passes of ds_write / barrier /
// ds_read) and, with 50 KiB of LDS a block, only three blocks share a CU to hide each other's DMA and store latencies
// (DESIGN.md section 4).  Here
//   * the MFMA operands are swapped -- acc = W_tile . X_tile^T -- so that a lane owns ONE output pixel and, per
//     accumulator quad, FOUR consecutive output channels: the epilogue runs out of the accumulator registers, per lane
//     8-byte fp16 / 4-byte int8 stores and 8-byte residual loads, no LDS staging, no barrier after the K loop's last;
//   * the only LDS besides the three DMA stages is a 1.5 KiB table of per-column constants (scale, zero-point
//     correction, bias) written once per block -> 49.5 KiB a block, and a block's life is the K loop plus a short,
//     latency-free tail: the residual rows and constants are requested before the K loop's MFMAs are done;
//   * GEGLU projections put the value tile and the gate tile of the same channels into one wave (column mapping
//     j * 64 + wn * 32), so value * gelu(gate) -> the consumer's 8-bit bins is register arithmetic.
// Output modes: TFMQ_OUT_F16 (+ bias, + fp16 / fp32 residual; no transposed region, no statistics), TFMQ_OUT_Q8,
// TFMQ_OUT_GEGLU_Q8.  Same int32 sums and the same epilogue arithmetic as k_conv_dma: bit-identical outputs.
#include "conv_common.hpp"
#include <type_traits>
#ifdef TFMQ_PHASE_TIMERS
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif

namespace {

#ifndef LIN_PREFETCH
#define LIN_PREFETCH 1
#endif
enum { LIN_F16 = 0, LIN_Q8 = 1, LIN_GEGLU = 2, LIN_GEGLU_FAST = 3 };
template <int MODE> constexpr bool lin_is_geglu = MODE == LIN_GEGLU || MODE == LIN_GEGLU_FAST;

// diagnostics build (-DTFMQ_PHASE_TIMERS): cycles a wave of k_lin_stream spends in each phase of its loop, per block
#ifdef TFMQ_PHASE_TIMERS
#define LIN_MARK(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[static_cast<size_t>(blockIdx.x) * 4 + (i)] = wall_clock64(); } while (0)
#define TFMQ_T0() unsigned long long t_acc[3] = {0, 0, 0}; unsigned long long t_last = wall_clock64()
#define TFMQ_TACC(i) do { const unsigned long long t_now = wall_clock64(); t_acc[i] += t_now - t_last; t_last = t_now; } while (0)
#define TFMQ_TDUMP(off, n) do { if (p.dbg && lane == 0) for (int i_ = 0; i_ < (n); ++i_) p.dbg[blockIdx.x * 8 + (off) + i_] = t_acc[i_]; } while (0)
#else
#define TFMQ_T0() do { } while (0)
#define LIN_MARK(i) do { } while (0)
#define TFMQ_TACC(i) do { } while (0)
#define TFMQ_TDUMP(off, n) do { } while (0)
#endif

// Epilogue out of the accumulator registers (both pointwise kernels).  cs = this tile's table {scale[BN], zero-point
// correction[BN] (int bits), bias[BN]} in LDS.  Packed fp32 arithmetic (two outputs per VALU instruction); same operations
// as k_conv_dma's epilogue: bit-identical.
// PHASE 0: everything.  PHASE 1: affine map + residual only, results left in `acc` as float bits.  PHASE 2: conversion
// and stores of what phase 1 left (k_lin_stream puts the next tile's residual loads between the two).
// F16OP: fp16 operands (un-quantised layers): the accumulators are fp32 bit patterns and value = scale * acc + bias
// (scale = 1 unless the weight-only integer grid carries one), the arithmetic of k_conv_dma<true>.
constexpr int LIN_STG_ROW_Q8 = 80;        // int8 rows: 64 + 16
constexpr int LIN_STG_ROW = 144;          // bytes per staged pixel row: 64 fp16 + 16 (rows 16 bytes apart in the banks: conflict-free)
template <int MODE, int PHASE = 0, bool F16OP = false, int NI = 2>
__device__ __forceinline__ void lin_epilogue(const ConvP& p, v16i (&acc)[NI][2], const float* cs, uint4 (&rres)[NI][2][2], bool has_res,
                                             int m0, int n0, int wm, int wn, int lane, float2 oqp, float2* ldsP = nullptr,
                                             unsigned char* stg = nullptr) {
  constexpr int BN = 128;
  const tfmq_conv_desc& d = p.d;
  const int h = lane >> 5;
  auto ncol0 = [&](int j) { return lin_is_geglu<MODE> ? j * 64 + wn * 32 : (wn * 2 + j) * 32; };
  auto affine2 = [&](int a0, int a1, float sx, float sy, int kx, int ky, float bx, float by) -> f2 {
    return f2{sx, sy} * f2{static_cast<float>(a0 + kx), static_cast<float>(a1 + ky)} + f2{bx, by};
  };
  // the eight values of one register octet: scale * float(acc + k) + bias
  auto affine8 = [&](const v16i& a, int u, int ct, f2 (&v)[4]) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float4 sc = *reinterpret_cast<const float4*>(cs + ct + 4 * e);
      const float4 bb = *reinterpret_cast<const float4*>(cs + 2 * BN + ct + 4 * e);
      if constexpr (F16OP) {
        v[2 * e] = f2{sc.x, sc.y} * f2{__int_as_float(a[8 * u + 4 * e + 0]), __int_as_float(a[8 * u + 4 * e + 1])} + f2{bb.x, bb.y};
        v[2 * e + 1] = f2{sc.z, sc.w} * f2{__int_as_float(a[8 * u + 4 * e + 2]), __int_as_float(a[8 * u + 4 * e + 3])} + f2{bb.z, bb.w};
      } else {
        const int4 kc = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(cs) + BN + ct + 4 * e);
        v[2 * e] = affine2(a[8 * u + 4 * e + 0], a[8 * u + 4 * e + 1], sc.x, sc.y, kc.x, kc.y, bb.x, bb.y);
        v[2 * e + 1] = affine2(a[8 * u + 4 * e + 2], a[8 * u + 4 * e + 3], sc.z, sc.w, kc.z, kc.w, bb.z, bb.w);
      }
    }
  };
  const bool transposed = MODE == LIN_F16 && d.yt != nullptr && n0 >= d.t_col0;      // tile-uniform (t_col0 % 128 == 0)
  auto epi = [&](auto exact_div) {
    constexpr bool EX = decltype(exact_div)::value;
    const QuantP qP = make_quantp(oqp);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int m = m0 + (wm * NI + i) * 32 + (lane & 31);
      const bool mok = m < p.M;
      const int thw = d.Ho * d.Wo;
      const int tb = transposed ? m / thw : 0, tt = m - tb * thw;
      if constexpr (MODE == LIN_GEGLU_FAST) {
        // value' = (scale / delta_o) float(acc) + bias', gate = scale float(acc) + bias'' (zero-point corrections folded into the biases
        // by the table), G = gelu_fast2(gate), bin = rint(value' G + zp): 13.75 issue slots per output instead of 22.75
        const int inner = d.Cout >> 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int co = 16 * h + 8 * u;
          f2 a[4], g[4];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float4 sa = *reinterpret_cast<const float4*>(cs + ncol0(0) + co + 4 * e), ba = *reinterpret_cast<const float4*>(cs + 2 * BN + ncol0(0) + co + 4 * e);
            const float4 sg = *reinterpret_cast<const float4*>(cs + ncol0(1) + co + 4 * e), bg = *reinterpret_cast<const float4*>(cs + 2 * BN + ncol0(1) + co + 4 * e);
            const v16i& av = acc[i][0];
            const v16i& gv = acc[i][1];
            a[2 * e] = pk_fma(f2{sa.x, sa.y}, f2{static_cast<float>(av[8 * u + 4 * e]), static_cast<float>(av[8 * u + 4 * e + 1])}, f2{ba.x, ba.y});
            a[2 * e + 1] = pk_fma(f2{sa.z, sa.w}, f2{static_cast<float>(av[8 * u + 4 * e + 2]), static_cast<float>(av[8 * u + 4 * e + 3])}, f2{ba.z, ba.w});
            g[2 * e] = pk_fma(f2{sg.x, sg.y}, f2{static_cast<float>(gv[8 * u + 4 * e]), static_cast<float>(gv[8 * u + 4 * e + 1])}, f2{bg.x, bg.y});
            g[2 * e + 1] = pk_fma(f2{sg.z, sg.w}, f2{static_cast<float>(gv[8 * u + 4 * e + 2]), static_cast<float>(gv[8 * u + 4 * e + 3])}, f2{bg.z, bg.w});
          }
          const unsigned w0 = geglu_fast_pack4(a[0], gelu_fast2(g[0]), a[1], gelu_fast2(g[1]), oqp.y);
          const unsigned w1 = geglu_fast_pack4(a[2], gelu_fast2(g[2]), a[3], gelu_fast2(g[3]), oqp.y);
          const int oc = (n0 >> 1) + wn * 32 + co;
          if (mok && oc < inner) *reinterpret_cast<uint2*>(d.yq + static_cast<size_t>(m) * inner + oc) = make_uint2(w0, w1);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (MODE == LIN_GEGLU) {
        const int inner = d.Cout >> 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int co = 16 * h + 8 * u;                 // channel offset of this octet inside a 32-channel tile
          f2 a[4], g[4];
          affine8(acc[i][0], u, ncol0(0) + co, a);
          affine8(acc[i][1], u, ncol0(1) + co, g);
          const unsigned w0 = quant_pack4_t<EX>(a[0] * gelu2(g[0]), a[1] * gelu2(g[1]), qP);
          const unsigned w1 = quant_pack4_t<EX>(a[2] * gelu2(g[2]), a[3] * gelu2(g[3]), qP);
          const int oc = (n0 >> 1) + wn * 32 + co;                // output channel (of Cout / 2)
          if (mok && oc < inner) *reinterpret_cast<uint2*>(d.yq + static_cast<size_t>(m) * inner + oc) = make_uint2(w0, w1);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int ct = ncol0(j) + 16 * h + 8 * u, n = n0 + ct;
            f2 v[4];
            if (PHASE != 2) {
              affine8(acc[i][j], u, ct, v);
              if (has_res) {
                if (d.res_f16) {
                  const uint4 rr = rres[i][j][u];
                  const float2 r0 = __half22float2(*reinterpret_cast<const __half2*>(&rr.x)), r1 = __half22float2(*reinterpret_cast<const __half2*>(&rr.y));
                  const float2 r2 = __half22float2(*reinterpret_cast<const __half2*>(&rr.z)), r3 = __half22float2(*reinterpret_cast<const __half2*>(&rr.w));
                  v[0] += f2{r0.x, r0.y};
                  v[1] += f2{r1.x, r1.y};
                  v[2] += f2{r2.x, r2.y};
                  v[3] += f2{r3.x, r3.y};
                } else if (mok && n < d.Cout) {
                  const float4 a0 = *reinterpret_cast<const float4*>(d.residual + static_cast<size_t>(m) * d.Cout + n);
                  const float4 a1 = *reinterpret_cast<const float4*>(d.residual + static_cast<size_t>(m) * d.Cout + n + 4);
                  v[0] += f2{a0.x, a0.y};
                  v[1] += f2{a0.z, a0.w};
                  v[2] += f2{a1.x, a1.y};
                  v[3] += f2{a1.z, a1.w};
                }
              }
            }
            if (MODE == LIN_F16 && PHASE == 0 && ldsP != nullptr) {
              // GroupNorm statistics of the consumer: per channel the 8-row group's (sum, sum of squares) of the fp32 values,
              // DPP sums over the 8 lanes of a pixel-row group in the canonical order (conv_common.hpp: group8_sum)
              const bool ok = mok && n < d.Cout;
              const int grp = (wm * NI + i) * 4 + ((lane & 31) >> 3);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float x0 = ok ? v[e].x : 0.0f, x1 = ok ? v[e].y : 0.0f;
                const float s0 = group8_sum(x0), s1 = group8_sum(x1);
                const float q0 = group8_sum(x0 * x0), q1 = group8_sum(x1 * x1);
                if ((lane & 7) == 0) *reinterpret_cast<float4*>(ldsP + grp * BN + ct + 2 * e) = make_float4(s0, q0, s1, q1);
              }
            }
            if (PHASE == 1) {        // combined values parked in the accumulator registers (the int32 sums are dead)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc[i][j][8 * u + 2 * e] = __float_as_int(v[e].x);
                acc[i][j][8 * u + 2 * e + 1] = __float_as_int(v[e].y);
              }
              __builtin_amdgcn_sched_barrier(0);
              continue;
            }
            if (PHASE == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = f2{__int_as_float(acc[i][j][8 * u + 2 * e]), __int_as_float(acc[i][j][8 * u + 2 * e + 1])};
            }
            if (mok && n < d.Cout) {
              if constexpr (MODE == LIN_F16) {
                if (transposed) {
                  // yt[b][n - t_col0][t]: the 32 lanes of a half-wave hold 32 consecutive pixels of one channel -> every
                  // 2-byte store instruction writes two contiguous 64-byte runs
                  __half* dst = reinterpret_cast<__half*>(d.yt) + (static_cast<size_t>(tb) * (d.Cout - d.t_col0) + (n - d.t_col0)) * thw + tt;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    dst[(2 * e) * static_cast<size_t>(thw)] = __float2half_rn(v[e].x);
                    dst[(2 * e + 1) * static_cast<size_t>(thw)] = __float2half_rn(v[e].y);
                  }
                } else if (!(PHASE == 0 && stg != nullptr)) {
                  *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(d.y) + static_cast<size_t>(m) * d.ldy + d.y_coff + n) =
                      make_uint4(pack_h2(v[0].x, v[0].y), pack_h2(v[1].x, v[1].y), pack_h2(v[2].x, v[2].y), pack_h2(v[3].x, v[3].y));
                }
              } else if (!(PHASE == 0 && stg != nullptr)) {
                *reinterpret_cast<uint2*>(d.yq + static_cast<size_t>(m) * d.Cout + n) =
                    make_uint2(quant_pack4_t<EX>(v[0], v[1], qP), quant_pack4_t<EX>(v[2], v[3], qP));
              }
            }
            if constexpr (MODE == LIN_Q8 && PHASE == 0) {      // int8 output: the same transpose with 80-byte staged rows (64 + 16)
              if (stg != nullptr)
                *reinterpret_cast<uint2*>(stg + (lane & 31) * LIN_STG_ROW_Q8 + j * 32 + 16 * h + 8 * u) =
                    make_uint2(quant_pack4_t<EX>(v[0], v[1], qP), quant_pack4_t<EX>(v[2], v[3], qP));
            }
            if constexpr (MODE == LIN_F16 && PHASE == 0) {
              // row-major fp16 output through a wave-private LDS transpose (see below): this lane's 8 channels of pixel row lane % 32
              if (stg != nullptr && !transposed)
                *reinterpret_cast<uint4*>(stg + (lane & 31) * LIN_STG_ROW + (j * 32 + 16 * h + 8 * u) * 2) =
                    make_uint4(pack_h2(v[0].x, v[0].y), pack_h2(v[1].x, v[1].y), pack_h2(v[2].x, v[2].y), pack_h2(v[3].x, v[3].y));
            }
            __builtin_amdgcn_sched_barrier(0);      // one octet at a time: interleaving them all spilled the accumulators
          }
        if constexpr (MODE == LIN_F16 && PHASE == 0) {
          // The register layout gives a store instruction 32 pixel rows x two 16-byte pieces: 32 partially written lines and 64
          // 16-byte write requests to the L2 per instruction (PMC: 16 bytes per L2 write request; a CU retires about one touched
          // line per 4 cycles).  Through LDS the wave's 32 x 64 fp16 block goes out as whole 128-byte row segments: 8 lanes per
          // row, 8 rows = 8 full lines per instruction.  Wave-private region, DS operations of a wave execute in order: no barrier.
          if (stg != nullptr && !transposed) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int row = t * 8 + (lane >> 3), pc = lane & 7;
              const uint4 w = *reinterpret_cast<const uint4*>(stg + row * LIN_STG_ROW + pc * 16);
              const int m2 = m0 + (wm * NI + i) * 32 + row, n2 = n0 + wn * 64 + pc * 8;
              if (m2 < p.M && n2 < d.Cout)
                *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(d.y) + static_cast<size_t>(m2) * d.ldy + d.y_coff + n2) = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
        }
        if constexpr (MODE == LIN_Q8 && PHASE == 0) {
          if (stg != nullptr) {            // 4 lanes per 64-byte row segment, 16 rows per store instruction
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int row = t * 16 + (lane >> 2), pc = lane & 3;
              const uint4 w = *reinterpret_cast<const uint4*>(stg + row * LIN_STG_ROW_Q8 + pc * 16);
              const int m2 = m0 + (wm * NI + i) * 32 + row, n2 = n0 + wn * 64 + pc * 16;
              if (m2 < p.M && n2 < d.Cout) *reinterpret_cast<uint4*>(d.yq + static_cast<size_t>(m2) * d.Cout + n2) = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
        }
      }
    }
  };
  if constexpr (MODE == LIN_F16 || MODE == LIN_GEGLU_FAST) {
    epi(std::false_type{});
  } else {
    if (__builtin_expect((__float_as_uint(oqp.x) & 0x7fffffu) == 0x7fffffu, 0)) epi(std::true_type{});
    else epi(std::false_type{});
  }
}

// this lane's residual values (fp16 stream) of tile (m0, n0), 16 bytes per register octet, branch-free: clamped addresses
template <int MODE>
__device__ __forceinline__ void lin_load_res(const ConvP& p, uint4 (&rres)[2][2][2], int m0, int n0, int wm, int wn, int lane) {
  const tfmq_conv_desc& d = p.d;
  const int h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + (wm * 2 + i) * 32 + (lane & 31);
    const int mc = m < p.M ? m : p.M - 1;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int n = n0 + (lin_is_geglu<MODE> ? j * 64 + wn * 32 : (wn * 2 + j) * 32) + 16 * h + 8 * u;
        const int nc = n < d.Cout ? n : 0;
        rres[i][j][u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(d.residual) + static_cast<size_t>(mc) * d.Cout + nc);
      }
  }
}

// NI (round 6): 32-row tiles per wave along M.  NI = 2: 128 x 128 output tiles, three blocks per CU (the form of rounds 2-5).  NI = 4: 256 x 128
// tiles, two blocks per CU -- 24 KB of operands per 64 MFMAs of a block instead of 16 KB per 32: a quarter fewer LDS-DMA pieces per MFMA on
// K loops that run at the DMA issue rate of their waves (profiles/r06_kstep_lin_direct.txt); layers WITHOUT a residual only (the residual
// octets of a 64 x 64 ... 128 x 64 wave tile would not fit beside 128 accumulator registers).  Same int32 sums, same epilogue: same bits.
// NST2 (round 6, GEGLU modes on 128-row tiles): two operand stages instead of three -- 34 KB of LDS and <= 128 VGPRs, FOUR blocks per CU; the
// pieces of step s + 1 are issued behind the barrier of step s (one step of prefetch instead of two).
template <int MODE, bool F16OP = false, int NI = 2, bool NST2 = false>
__global__ __launch_bounds__(256, NST2 ? 4 : (NI == 2 ? 3 : 2)) void k_lin_direct(ConvP p) {
  static_assert(!NST2 || (NI == 2 && lin_is_geglu<MODE>), "two stages: the GEGLU forms on 128-row tiles");
  constexpr int BM = 64 * NI, BN = 128;
  constexpr int STAGE = (BM + BN) * 64;
  constexpr int NST = NST2 ? 2 : 3;
  constexpr int NLOAD = NI + 2;                  // DMA pieces per wave per K-step: NI of A, 2 of B
  constexpr int CONST_OFF = NST * STAGE;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE + 3 * BN * 4];

  const tfmq_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  LIN_MARK(0);
  // Tile order inside an XCD's contiguous range: panels of GM row tiles, inside a panel the row tile runs fastest and the
  // column tile slowest -- the ~96 blocks an XCD holds at once then share GM activation tiles (GM * 128 * Cin bytes, sized
  // to stay in the 4 MB L2 for the whole panel) and 96 / GM weight tiles, instead of one or two activation tiles and EVERY
  // weight tile: with N fastest the 13 MB weight matrix of the 1280 -> 10240 projection was re-streamed through the fabric
  // for every row tile (PMC: 14x the algorithmic fetch bytes on the GEGLU launches).
  const int bid = xcd_tile_id();
  const int tiles_m = (p.M + BM - 1) / BM;
  int gm = (3 << 19) / (BM * d.Cin);                       // 1.5 MB of activation rows
  gm = gm < 1 ? 1 : (gm > 16 ? 16 : gm);
  // weights that fit beside the activations keep N fastest (gm = 1): the column tiles of a row tile then run at the same
  // moment and its activation rows are fetched once (1280 -> 320 with three column tiles: panels were 16 % slower)
  if (static_cast<long>(p.cout_pad) * d.Cin < (2L << 20)) gm = 1;
  const int per_panel = gm * p.tiles_n;
  const int panel = bid / per_panel, rp = bid - panel * per_panel;
  const int gml = (tiles_m - panel * gm) < gm ? (tiles_m - panel * gm) : gm;      // rows of this (possibly last, shorter) panel
  const int tile_n = rp / gml, tile_m = panel * gm + (rp - tile_n * gml);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- DMA sources (pointwise: pixel m reads input pixel m; rows past M read the zero row of the pad table)
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));
  const unsigned char* xb = static_cast<const unsigned char*>(d.x);
  const int dcol = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
  const unsigned char* a_ptr[NI];
  const unsigned char* a2_ptr[NI];
  const unsigned char* b_ptr[2];
  bool a_live[NI];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int piece = wid * NI + it;
    const int m = m0 + piece * 16 + (lane >> 2);
    a_live[it] = m < p.M;
    a_ptr[it] = m < p.M ? xb + static_cast<size_t>(m) * (F16OP && d.x2 ? d.cin1 : d.Cin) * (F16OP ? 2 : 1) + dcol : p.pad_table + dcol;
    a2_ptr[it] = nullptr;
    if constexpr (F16OP) {         // second source of a virtual channel concat (tfmq_conv_desc.x2): K-steps >= cin1 / 32 read it
      a2_ptr[it] = (d.x2 && m < p.M) ? static_cast<const unsigned char*>(d.x2) + static_cast<size_t>(m) * (d.Cin - d.cin1) * 2 + dcol : a_ptr[it];
    }
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int piece = wid * 2 + it;
    int n = n0 + piece * 16 + (lane >> 2);
    if constexpr (F16OP) {           // fp16 weights [cout][cin_pad] row-major (tfmq_pack_w_f16), fp16 activations: 32 channels per K-step
      n = n < d.Cout ? n : d.Cout - 1;
      b_ptr[it] = static_cast<const unsigned char*>(d.w) + static_cast<size_t>(n) * p.cin_pad * 2 + dcol;
    } else {
      n = n < p.cout_pad ? n : p.cout_pad - 1;
      b_ptr[it] = static_cast<const unsigned char*>(d.w) + (static_cast<size_t>(n / 32) * p.nsteps * 32 + (n % 32)) * 64 + dcol;
    }
  }
  constexpr size_t BSTEP = F16OP ? 64 : 2048;
  const int s_split = (F16OP && d.x2) ? d.cin1 / 32 : (1 << 30);
  auto issue = [&](int s, int stage) {
    const unsigned sbase = lds0 + stage * STAGE;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      if (F16OP && s >= s_split) glds16(a2_ptr[it] + (a_live[it] ? (s - s_split) * 64 : 0), sbase + __builtin_amdgcn_readfirstlane((wid * NI + it) * 1024));
      else glds16(a_ptr[it] + (a_live[it] ? s * 64 : 0), sbase + __builtin_amdgcn_readfirstlane((wid * NI + it) * 1024));
    }
    glds16(b_ptr[0] + static_cast<size_t>(s) * BSTEP, sbase + __builtin_amdgcn_readfirstlane(BM * 64 + (wid * 2) * 1024));
    glds16(b_ptr[1] + static_cast<size_t>(s) * BSTEP, sbase + __builtin_amdgcn_readfirstlane(BM * 64 + (wid * 2 + 1) * 1024));
  };

  // fragment rows: pixels (wm * 2 + i) * 32 + lane % 32; channels of N-tile j: plain (wn * 2 + j) * 32, GEGLU j * 64 + wn * 32
  // (tile columns [0, 64) = value, [64, 128) = gate of the same 64 output channels)
  auto ncol0 = [&](int j) { return lin_is_geglu<MODE> ? j * 64 + wn * 32 : (wn * 2 + j) * 32; };
  const int fsw = (h ^ ((lane >> 2) & 3)) << 4;           // physical 16-byte slot of k-slot h in this lane's row
  const int brow = lin_brow(lane & 31);                    // weight rows in the permuted order (see lin_brow)
  const int bsw = (h ^ ((brow >> 2) & 3)) << 4;

  v16i acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  issue(0, 0);
  if (!NST2 && p.nsteps > 1) issue(1, 1);

  // ---- requested now, consumed after the K loop: per-column constants (threads < BN), the quantizer parameters, and the
  // residual values of this lane's outputs (branch-free: clamped addresses).  They are younger than the first DMA
  // pieces, so the loop's counted waits stay correct (only its first step waits for more than it needs).
  // (Measured and dropped, same-box A/B: writing the table before step 0 and STARTING the accumulators at the zero-point
  // correction -- no zero fill, no integer add per output, 12 % fewer VALU instructions -- GEGLU -1.5 %, residual Linears +3 %.)
  // (Writing the table before step 0 and starting the accumulators at the zero-point correction -- no zero fill, no integer
  // add per output -- was measured: 4-7 % SLOWER on the GEGLU projections; the LDS round trip sits on the block's latency chain.)
  float c_ws = 1.0f, c_bias = 0.0f;
  int c_zp = 0, c_rs = 0;
  if (tid < BN) {
    const int n = n0 + tid;
    if (n < d.Cout) {
      if constexpr (!F16OP) {
        const int4 wmv = reinterpret_cast<const int4*>(d.wmeta)[n];
        c_zp = wmv.x;
        c_rs = wmv.y;
        c_ws = d.wscale[n];
      } else if (d.wscale) {
        c_ws = d.wscale[n];
      }
      c_bias = d.bias ? d.bias[n] : 0.0f;
    }
  }
  float2 aqp = make_float2(1.0f, 128.0f);
  if constexpr (!F16OP) aqp = load_qparam(d.aq);
  float2 oqp = make_float2(1.0f, 0.0f);
  if constexpr (MODE != LIN_F16) oqp = load_qparam(d.oq);
  uint4 rres[NI][2][2];                    // MODE F16 / Q8 with a residual: 8 channels x fp16 per (i, j, register octet)
  const bool has_res = NI == 2 && !lin_is_geglu<MODE> && d.residual != nullptr;      // (the launcher keeps residual layers on NI = 2)
  if constexpr (NI == 2) {
    if (has_res && d.res_f16) lin_load_res<MODE>(p, rres, m0, n0, wm, wn, lane);
  }

#ifdef TFMQ_PHASE_TIMERS
  // where a wave's K-step goes: [0] counted vmcnt wait, [1] barrier, [2] DMA issue, [3] fragment reads until their data is there,
  // [4] MFMA issue (the wave is held while the matrix pipe is busy), [5] steps  (shader cycles, wave 0 of the block; s_memtime costs ~10 %)
  unsigned long long kacc[6] = {0, 0, 0, 0, 0, 0};
#define KT(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = clock64(); kacc[i] += t_ - kt; kt = t_; } while (0)
  unsigned long long kt = clock64();
#else
#define KT(i) do { } while (0)
#endif
  int st_c = 0, st_i = NST2 ? 1 : 2;
  for (int s = 0; s < p.nsteps; ++s) {
    if (!NST2 && s + 1 < p.nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KT(0);
    asm volatile("s_barrier" ::: "memory");
    KT(1);
    if (s == 0) LIN_MARK(1);
    const bool late = p.issue_split && wid >= 2;      // (wave-uniform)
    constexpr int AHEAD = NST2 ? 1 : 2;
    if (!late && s + AHEAD < p.nsteps) issue(s + AHEAD, st_i);
    KT(2);
    const unsigned char* sa = lds + st_c * STAGE;
    const unsigned char* sb = sa + BM * 64;
    // GEGLU modes (round 4): both K halves' fragments are requested up front, the second half's reads travel under the first half's MFMAs
    // (same-box A/B, gpurun_out/r04/lin_prefetch_ab.txt: GEGLU 640 -> 5120 and 1280 -> 10240 -6.5 %; the fp16 / int8-output modes
    // -2 ... +7 %: they keep the read-then-multiply order per half)
    constexpr bool PF = LIN_PREFETCH && lin_is_geglu<MODE>;
    v4i af[2][NI], bf[2][2];
    auto read_frags = [&](int ks) {
#pragma unroll
      for (int i = 0; i < NI; ++i) af[ks][i] = *reinterpret_cast<const v4i*>(sa + ((wm * NI + i) * 32 + (lane & 31)) * 64 + (fsw ^ (ks << 5)));
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[ks][j] = *reinterpret_cast<const v4i*>(sb + (ncol0(j) + brow) * 64 + (bsw ^ (ks << 5)));
    };
    read_frags(0);
    if constexpr (PF) read_frags(1);
#ifdef TFMQ_PHASE_TIMERS
    asm volatile("" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(bf[0][0]), "+v"(bf[0][1]));
    KT(3);
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (!PF) {
        if (ks == 1) read_frags(1);
      }
      // operands swapped: the accumulator tile is (channels x pixels) -- lane = pixel, register quad = 4 consecutive channels
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (F16OP) {
            typedef _Float16 v8h_t __attribute__((ext_vector_type(8)));
            typedef float v16f_t __attribute__((ext_vector_type(16)));
            v16f_t& af32 = *reinterpret_cast<v16f_t*>(&acc[i][j]);
            af32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<v8h_t*>(&bf[ks][j]), *reinterpret_cast<v8h_t*>(&af[ks][i]), af32, 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
          }
        }
      if (ks == 0 && late && s + AHEAD < p.nsteps) issue(s + AHEAD, st_i);
    }
    st_c = st_c == NST - 1 ? 0 : st_c + 1;
    st_i = st_i == NST - 1 ? 0 : st_i + 1;
#ifdef TFMQ_PHASE_TIMERS
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][1]));
    KT(4);
    kacc[5] += 1;
#endif
  }
#ifdef TFMQ_PHASE_TIMERS
  if (p.dbg2 && tid == 0)
    for (int i = 0; i < 6; ++i) p.dbg2[static_cast<size_t>(blockIdx.x) * 8 + i] = kacc[i];
#endif

  LIN_MARK(2);
  // ---- per-column constants -> LDS table {scale, zero-point correction (as float bits of an int), bias}
  float* cs = reinterpret_cast<float*>(lds + CONST_OFF);
  const int za = static_cast<int>(aqp.y);
  if (tid < BN) {
    const float sc = aqp.x * c_ws;
    const int kc = (128 - za) * (c_rs - p.Ktot * c_zp);
    if constexpr (MODE == LIN_GEGLU_FAST) {      // zero-point correction folded into the bias; value columns pre-divided by the output delta
      const float bf = sc * static_cast<float>(kc) + c_bias;
      const bool val = (tid & 64) == 0;
      cs[tid] = val ? sc / oqp.x : sc;
      cs[2 * BN + tid] = val ? bf / oqp.x : bf;
    } else {
      cs[tid] = sc;
      reinterpret_cast<int*>(cs)[BN + tid] = kc;
      cs[2 * BN + tid] = c_bias;
    }
  }
  __syncthreads();

  // statistics partials [BM / 8 eight-row groups][BN] live in the (now idle) DMA stages
  float2* ldsP = (MODE == LIN_F16 && d.stats) ? reinterpret_cast<float2*>(lds) : nullptr;
  // (store staging: 32 rows x 144 bytes per wave behind the 16 (32) KB of statistics partials, all inside the idle DMA stages)
  unsigned char* stg = MODE == LIN_F16 ? lds + (BM / 8) * BN * 8 + wid * (32 * LIN_STG_ROW) : nullptr;
  if constexpr (MODE == LIN_Q8) stg = (d.Cout & 15) == 0 ? lds + wid * (32 * LIN_STG_ROW_Q8) : nullptr;
  lin_epilogue<MODE, 0, F16OP, NI>(p, acc, cs, rres, has_res, m0, n0, wm, wn, lane, oqp, ldsP, stg);
  if constexpr (MODE == LIN_F16) {
    if (d.stats) {
      __syncthreads();
      const int seg = d.stats_seg, nseg = BM / seg, gps = seg / 8;
      for (int o = tid; o < nseg * BN; o += 256) {
        const int sidx = o / BN, col = o - sidx * BN;
        float2 a = make_float2(0.0f, 0.0f);
        for (int q = 0; q < gps; ++q) {            // a segment = its 8-row groups added in row order
          const float2 b = ldsP[(sidx * gps + q) * BN + col];
          a.x += b.x;
          a.y += b.y;
        }
        const int row0 = m0 + sidx * seg, n = n0 + col;
        if (row0 < p.M && n < d.Cout) reinterpret_cast<float2*>(d.stats)[static_cast<size_t>(row0 / seg) * d.Cout + n] = a;
      }
    }
  }
  LIN_MARK(3);
}


// ---------------------------------------------------------------------------------------------------------------------
// (Round 2 carried two persistent variants of this kernel here -- K5q `k_lin_stream`: a producer wave streaming the LDS-DMA of consecutive
// tiles through one ring for four consumer waves; K5r `k_lin_persist`: symmetric waves, the ring running across tile boundaries with exact
// counted waits.  Both were bit-identical to k_lin_direct and 3-60 % slower on every SD shape (the in-order vmcnt
// couples a wave's stores to its DMA, one producer wave cannot feed a CU's L2 -> LDS stream, the compiler waits for "everything" at the
// first use of prefetched constants).  Negative results stay in DESIGN.md; the 420 lines were removed in round 3.)

}  // namespace

bool launch_conv_lin(tfmq_handle h, ConvP& p, hipStream_t st, bool m256) {
  const tfmq_conv_desc& d = p.d;
  if (m256 && (d.residual || p.M < 256 || (d.stats && 256 % d.stats_seg != 0))) m256 = false;      // (falls back to the 128-row form)
  if (d.KH != 1 || d.KW != 1 || d.stride != 1 || d.up2x || d.pad_t != 0 || d.pad_l != 0 || d.Ho != d.H || d.Wo != d.W) return false;
  if (d.Cin % 32 != 0 || p.chunks != (d.Cin + 63) / 64 || static_cast<size_t>(d.B) * d.H * d.W * d.Cin >= (static_cast<size_t>(1) << 31)) return false;
  if (d.rowadd || (d.Cout & 7) != 0) return false;                 // a lane moves whole 8-channel octets
  if (d.stats && (d.out_mode != TFMQ_OUT_F16 || d.yt || 128 % d.stats_seg != 0)) return false;
  if (d.yt && (d.out_mode != TFMQ_OUT_F16 || d.residual)) return false;
  int mode;
  if (d.out_mode == TFMQ_OUT_F16) {
    if (((d.ldy | d.y_coff) & 7) != 0) return false;
    mode = LIN_F16;
  } else if (d.out_mode == TFMQ_OUT_Q8) {
    mode = LIN_Q8;
  } else if (d.out_mode == TFMQ_OUT_GEGLU_Q8 || d.out_mode == TFMQ_OUT_GEGLU_Q8_FAST) {
    if (d.residual || d.Cout % 128 != 0 || (d.Cout >> 1) % 8 != 0) return false;
    mode = d.out_mode == TFMQ_OUT_GEGLU_Q8 ? LIN_GEGLU : LIN_GEGLU_FAST;
  } else {
    return false;
  }
  p.tiles_n = (d.Cout + 127) / 128;
  static const int issue_split_env = getenv("TFMQ_LIN_ISSUE_SPLIT") ? atoi(getenv("TFMQ_LIN_ISSUE_SPLIT")) : 0;
  p.issue_split = issue_split_env;
  // round 6: the consumer-sized GEGLU form on TWO operand stages, four blocks per CU (-5 % on 640 -> 5120, -1 ... -2 % on the others, same bits:
  // profiles/r06_ab_lin_geglu_nst2_attn80.txt); TFMQ_LIN_GEGLU_NST2=0 restores three stages / three blocks
  static const int nst2_env = getenv("TFMQ_LIN_GEGLU_NST2") ? atoi(getenv("TFMQ_LIN_GEGLU_NST2")) : 1;
  // (measured and dropped: the fp16 / int8-output forms held to 128 VGPRs for a fourth block -- 16-24 registers spilled, the residual octets among
  // them: residual layers 10-20 % SLOWER, the others equal; and two stages leave no room for the epilogue's staging: profiles/r06_ab_lin_geglu_nst2_attn80.txt)
  const int tiles_m = m256 ? (p.M + 255) / 256 : (p.M + 127) / 128;
  const int n_tiles = p.tiles_n * tiles_m;
  dim3 grid(static_cast<unsigned>(n_tiles));
#ifdef TFMQ_PHASE_TIMERS
  static unsigned long long* dbuf2 = nullptr;
  if (!dbuf2) (void)hipMalloc(reinterpret_cast<void**>(&dbuf2), sizeof(unsigned long long) * 4 * (1u << 18));
  p.dbg = grid.x <= (1u << 18) ? dbuf2 : nullptr;
  static unsigned long long* dbuf3 = nullptr;
  if (!dbuf3) (void)hipMalloc(reinterpret_cast<void**>(&dbuf3), sizeof(unsigned long long) * 8 * (1u << 18));
  p.dbg2 = grid.x <= (1u << 18) ? dbuf3 : nullptr;
#endif
  if (m256) {
    if (mode == LIN_F16) hipLaunchKernelGGL((k_lin_direct<LIN_F16, false, 4>), grid, dim3(256), 0, st, p);
    else if (mode == LIN_Q8) hipLaunchKernelGGL((k_lin_direct<LIN_Q8, false, 4>), grid, dim3(256), 0, st, p);
    else if (mode == LIN_GEGLU) hipLaunchKernelGGL((k_lin_direct<LIN_GEGLU, false, 4>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_lin_direct<LIN_GEGLU_FAST, false, 4>), grid, dim3(256), 0, st, p);
  } else if (mode == LIN_F16) hipLaunchKernelGGL((k_lin_direct<LIN_F16>), grid, dim3(256), 0, st, p);
  else if (mode == LIN_Q8) hipLaunchKernelGGL((k_lin_direct<LIN_Q8>), grid, dim3(256), 0, st, p);
  else if (mode == LIN_GEGLU && nst2_env) hipLaunchKernelGGL((k_lin_direct<LIN_GEGLU, false, 2, true>), grid, dim3(256), 0, st, p);
  else if (mode == LIN_GEGLU) hipLaunchKernelGGL((k_lin_direct<LIN_GEGLU>), grid, dim3(256), 0, st, p);
  else if (nst2_env) hipLaunchKernelGGL((k_lin_direct<LIN_GEGLU_FAST, false, 2, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_lin_direct<LIN_GEGLU_FAST>), grid, dim3(256), 0, st, p);
#ifdef TFMQ_PHASE_TIMERS
  if (p.dbg && getenv("TFMQ_PHASE_PRINT")) {
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> hb(static_cast<size_t>(grid.x) * 4);
    (void)hipMemcpy(hb.data(), dbuf2, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double a = 0, b = 0, c = 0;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (unsigned i = 0; i < grid.x; ++i) {
      a += double(hb[i * 4 + 1] - hb[i * 4]);
      b += double(hb[i * 4 + 2] - hb[i * 4 + 1]);
      c += double(hb[i * 4 + 3] - hb[i * 4 + 2]);
      t0 = hb[i * 4] < t0 ? hb[i * 4] : t0;
      t1 = hb[i * 4 + 3] > t1 ? hb[i * 4 + 3] : t1;
    }
    {
      std::vector<unsigned long long> kb(static_cast<size_t>(grid.x) * 8);
      (void)hipMemcpy(kb.data(), dbuf3, kb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double ks[6] = {0, 0, 0, 0, 0, 0};
      for (unsigned i = 0; i < grid.x; ++i)
        for (int q = 0; q < 6; ++q) ks[q] += double(kb[i * 8 + q]);
      const double st = ks[5] > 0 ? ks[5] : 1;
      fprintf(stderr, "[lin_direct Cin%d Cout%d mode%d] K-step of wave 0, shader cycles: vmcnt wait %.0f, barrier %.0f, DMA issue %.0f, fragment reads %.0f, MFMA issue %.0f = %.0f per step\n",
              d.Cin, d.Cout, mode, ks[0] / st, ks[1] / st, ks[2] / st, ks[3] / st, ks[4] / st, (ks[0] + ks[1] + ks[2] + ks[3] + ks[4]) / st);
    }
    const double span = double(t1 - t0) / 100.0, blocks_per_cu = double(grid.x) / h->cu_count;
    fprintf(stderr, "[lin_direct Cin%d Cout%d mode%d] blocks %u (%.1f per CU): start->first data %.2f us, K loop %.2f us, epilogue (to last store issued) %.2f us per block; span %.1f us = %.2f us per block slot of 3 per CU\n",
            d.Cin, d.Cout, mode, grid.x, blocks_per_cu, a / grid.x / 100, b / grid.x / 100, c / grid.x / 100, span, span / (blocks_per_cu / 3.0));
  }
#endif
  return true;
}

// Un-quantised pointwise layers (skip-connection 1x1 convs of the UNets' up path: fp16 NHWC input written by the producing
// GroupNorm, fp16 weights): the same register-direct kernel on f16 MFMA.  fp16 output only; false = not taken.
bool launch_conv_lin_f16(tfmq_handle h, ConvP& p, hipStream_t st) {
  const tfmq_conv_desc& d = p.d;
  if (d.KH != 1 || d.KW != 1 || d.stride != 1 || d.up2x || d.pad_t != 0 || d.pad_l != 0 || d.Ho != d.H || d.Wo != d.W) return false;
  if (!d.x_f16 || d.Cin % 32 != 0 || p.cin_pad != d.Cin || static_cast<size_t>(d.B) * d.H * d.W * d.Cin * 2 >= (static_cast<size_t>(1) << 31)) return false;
  if (d.x2 && (d.cin1 <= 0 || d.cin1 >= d.Cin || d.cin1 % 32 != 0)) return false;
  if (d.out_mode != TFMQ_OUT_F16 || d.rowadd || (d.Cout & 7) != 0 || ((d.ldy | d.y_coff) & 7) != 0) return false;
  if (d.yt && (d.residual || d.stats || d.t_col0 % 128 != 0)) return false;       // transposed V^T region: as the w4a8 launcher
  if (d.residual && !d.res_f16) return false;
  if (d.stats && 128 % d.stats_seg != 0) return false;
  (void)h;
  p.issue_split = 0;
  p.tiles_n = (d.Cout + 127) / 128;
  const int tiles_m = (p.M + 127) / 128;
  dim3 grid(static_cast<unsigned>(p.tiles_n) * tiles_m);
  hipLaunchKernelGGL((k_lin_direct<LIN_F16, true>), grid, dim3(256), 0, st, p);
  return true;
}
