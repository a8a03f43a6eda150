// K5/K6: Conv2d / Linear as implicit GEMM on the gfx950 matrix cores.
//
//   w4a8 path : A = int8 NHWC activations (bin-128), B = int8 (q_w - z_w) expanded once from the packed int4
//               weights (tfmq_expand_w4), v_mfma_i32_32x32x32_i8, exact int32 accumulation.
//               y = da*dw[c] * ( sum a'.(q_w - z_w) + (128-za) * sum_k (q_w - z_w) ) + b[c]
//               which equals the reference's F.conv2d on the two fake-quantised operands
//               (quant/quant_layer.py:318-338) up to fp32 rounding of the final scale.
//               Zero padding must be a *real* zero => padded taps carry a' = za-128.
//   f16  path : un-quantised layers (conv_in/conv_out, nin_shortcut, downsample.conv;
//               quant/quant_model.py:57-58,103-120): A = fp32 NHWC converted to f16 while
//               staging, B = f16, v_mfma_f32_32x32x16_f16 with fp32 accumulation.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin ordered (kh,kw,cin) so a
// K-step is 64 contiguous bytes of one input pixel.  256 threads = 4 waves; LDS rows are
// 64 bytes (64 int8 or 32 f16) with a 16-byte-slot XOR swizzle so ds_read_b128 fragment
// reads are bank-conflict free.
//
// Two main loops share one epilogue:
//   k_conv_dma   (w4a8, Cin % 64 == 0, <= 9 taps): both operands travel global -> LDS by LDS-DMA
//                (global_load_lds_dwordx4), three LDS stages, loads of K-step s+2 in flight while step s is
//                multiplied, counted vmcnt + one raw s_barrier per K-step, no staging registers, no ds_write,
//                ~20 VALU instructions per K-step.
//   k_conv_igemm (f16 layers; w4a8 shapes the DMA loop does not take): register-prefetched, double-buffered.
#include "conv_common.hpp"
#include <type_traits>
#include <cstdlib>
#ifdef TFMQ_PHASE_TIMERS
#include <cstdio>
#endif

template <int BM, int BN>
__host__ __device__ constexpr int epi_lds_bytes() {
  constexpr int PR = (BN == 128) ? 64 : BM;
  constexpr int NTR = 256 / (BN / 4), RPT = PR / NTR;
  constexpr int NTRh = RPT == 8 ? 2 * NTR : NTR;      // fp16-stream store pass: 4-row partials when a thread row holds 8 rows
  return PR * (BN + 4) * 4 + NTRh * BN * 8 + BN * 8;
}

// ---- epilogue.  The accumulators (C/D layout of the 32x32 MFMA: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) are dequantised in registers, staged through LDS one
// PR-row pass at a time, and leave the CU as whole rows: every lane moves 16 B (float4) of the
// temb row, the residual and the output, so the fp32 traffic of the epilogue is fully coalesced
// (per-lane 4-byte strided accesses made this phase 2.5x slower than the MFMA loop).  The same
// pass produces the per-channel sum / sum-of-squares of every SEG-row segment for the GroupNorm
// that consumes this tensor, so that GroupNorm never has to re-read it for statistics.
// per-column constants of a wave's output tiles (weight scale, bias, {zero point, row sum} of the packed weights) and the
// consumer quantizer's parameters: plain global loads whose latency a short-K block cannot afford at the start of its
// epilogue -- the LDS-DMA kernels request them right behind their first DMA pieces
template <int WN_TILES>
struct EpiCols {
  float ws[WN_TILES], bias[WN_TILES];
  int zp[WN_TILES], rs[WN_TILES];
  float2 oqp;
};

template <bool INT8, int WAVES_N, int WN_TILES>
__device__ __forceinline__ EpiCols<WN_TILES> load_epi_cols(const ConvP& p, int n0) {
  const tfmq_conv_desc& d = p.d;
  const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) % WAVES_N;
  EpiCols<WN_TILES> ec;
#pragma unroll
  for (int j = 0; j < WN_TILES; ++j) {
    const int n = n0 + (wn * WN_TILES + j) * 32 + (lane & 31);
    const bool nok = n < d.Cout;
    ec.ws[j] = (d.wscale && nok) ? d.wscale[n] : 1.0f;
    ec.bias[j] = (d.bias && nok) ? d.bias[n] : 0.0f;
    ec.zp[j] = ec.rs[j] = 0;
    if constexpr (INT8) {
      if (nok) {
        const int4 wmv = reinterpret_cast<const int4*>(d.wmeta)[n];
        ec.zp[j] = wmv.x;
        ec.rs[j] = wmv.y;
      }
    }
  }
  ec.oqp = make_float2(1.0f, 0.0f);
  if (d.out_mode == TFMQ_OUT_Q8 || d.out_mode == TFMQ_OUT_GEGLU_Q8) ec.oqp = load_qparam(d.oq);
  return ec;
}

template <bool INT8, int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES, bool RES_PRE = false, typename ACC>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, unsigned char* lds, ACC (&acc)[WM_TILES][WN_TILES],
                                              int m0, int n0, float2 aqp, int za, const EpiCols<WN_TILES>& ec) {
  constexpr int BM = WAVES_M * WM_TILES * 32;
  constexpr int BN = WAVES_N * WN_TILES * 32;
  constexpr int PR = (BN == 128) ? 64 : BM;       // rows per pass
  constexpr int LDO = BN + 4;                      // padded LDS row (floats)
  constexpr int TPR = BN / 4;                      // threads per output row (float4 each)
  constexpr int NTR = 256 / TPR;                   // thread-rows
  constexpr int RPT = PR / NTR;                    // consecutive rows per thread
  static_assert(RPT >= 1 && RPT <= 16, "rows per thread");
  const tfmq_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  float* ldsO = reinterpret_cast<float*>(lds);                      // [PR][LDO]
  float2* ldsP = reinterpret_cast<float2*>(lds + PR * LDO * 4);     // [NTR][BN] partial (sum, sumsq)
  float2* ldsG = ldsP + (RPT == 8 ? 2 * NTR : NTR) * BN;           // [BN] running segment sums (behind the fp16 pass's 4-row partials)
  const int hw = d.Ho * d.Wo;
  const float* rowadd = d.rowadd;
  if (rowadd && d.rowadd_step) rowadd += static_cast<size_t>(load_scalar_i32(d.rowadd_step)) * d.rowadd_step_stride;
  const bool vec_ok = ((d.Cout | d.ldy | d.y_coff) & 3) == 0 && (!d.rowadd || (d.rowadd_ld & 3) == 0);
  const int seg = d.stats ? d.stats_seg : 0;
  const int tr = tid / TPR, c4 = (tid % TPR) * 4;
  // tiles of the transposed output region (V^T for the attention kernel) stage with an odd row pitch so the
  // column-wise LDS reads of their store pass are (at most 2-way) conflict free
  const bool q8 = d.out_mode == TFMQ_OUT_Q8;
  const float2 oqp = ec.oqp;
  // every global access of the store pass moves 16 bytes per lane where the shapes allow it (8-byte fp16 / 4-byte int8
  // accesses made these passes instruction-issue bound, not bandwidth bound): 8 channels per item for fp16 outputs,
  // 16 for int8 outputs
  const bool vec8_ok = vec_ok && ((d.Cout | d.ldy | d.y_coff) & 7) == 0;
  const bool vec16_ok = vec_ok && (d.Cout & 15) == 0;
  const bool transposed = d.out_mode == TFMQ_OUT_F16 && d.yt && n0 >= d.t_col0;
  const int ldo = transposed ? BN + 1 : LDO;

  float sc_[WN_TILES], bias_[WN_TILES];
  int corr_[WN_TILES];
#pragma unroll
  for (int j = 0; j < WN_TILES; ++j) {
    const int n = n0 + (wn * WN_TILES + j) * 32 + (lane & 31);
    const bool nok = n < d.Cout;
    sc_[j] = (!INT8 && d.wscale) ? ec.ws[j] : 1.0f;
    bias_[j] = ec.bias[j];
    corr_[j] = 0;
    if constexpr (INT8) {
      if (nok) {
        corr_[j] = (128 - za) * (ec.rs[j] - p.Ktot * ec.zp[j]);
        sc_[j] = aqp.x * ec.ws[j];
      }
    }
  }

  // the residual rows of a pass are requested before its accumulators are staged, so their latency (HBM: these
  // layers run at the fp32-activation roofline) overlaps the staging and the barrier instead of following them
  // (RES_PRE kernels are launched only for vectorisable fp32 / Q8 outputs with a residual; a separate instantiation,
  // because the 32 extra live registers cost the other output modes spills)
  // segment sums of the GroupNorm statistics from the per-thread-row partials in ldsP (one fixed order, see phase 2)
  auto reduce_stats = [&](int pass, auto ppg_tag) {
    __syncthreads();
    constexpr int PPG = decltype(ppg_tag)::value;          // stored partials per 8-row group
    const int rows_seg = seg < PR ? seg : PR;
    const int nseg_pass = PR / rows_seg, g8 = rows_seg / 8;
    for (int o = tid; o < nseg_pass * BN; o += 256) {
      const int sidx = o / BN, col = o % BN;
      // a segment spanning several passes keeps adding its groups to the running sum, in the same order
      float2 a = seg <= PR ? make_float2(0.0f, 0.0f) : ldsG[col];
      for (int q = 0; q < g8; ++q) {
        const int base = (sidx * g8 + q) * PPG;
        float2 b = ldsP[base * BN + col];
        if constexpr (PPG == 2) {
          const float2 b2 = ldsP[(base + 1) * BN + col];
          b.x += b2.x;
          b.y += b2.y;
        }
        a.x += b.x;
        a.y += b.y;
      }
      const int n = n0 + col;
      if (seg <= PR) {
        const int row0 = m0 + pass * PR + sidx * seg;
        if (row0 < p.M && n < d.Cout) reinterpret_cast<float2*>(d.stats)[static_cast<size_t>(row0 / seg) * d.Cout + n] = a;
      } else {  // write after the segment's last pass
        const bool last = ((pass + 1) * PR) % seg == 0;
        if (last) {
          const int srow0 = m0 + (pass + 1) * PR - seg;
          if (srow0 < p.M && n < d.Cout) reinterpret_cast<float2*>(d.stats)[static_cast<size_t>(srow0 / seg) * d.Cout + n] = a;
          a = make_float2(0.0f, 0.0f);
        }
        ldsG[col] = a;
      }
    }
  };
  for (int pass = 0; pass < BM / PR; ++pass) {
    float4 rpre[RES_PRE ? RPT : 1];
    if constexpr (RES_PRE) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int m = m0 + pass * PR + tr * RPT + k, n = n0 + c4;
        rpre[k] = (m < p.M && n < d.Cout) ? load_res4(d, m, n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // fp16-stream store pass (below): its fp16 residual rows are requested here, branch-free, for the same reason
    constexpr int TPR8p = BN / 8, RPThp = RPT == 8 ? 4 : RPT, NTRhp = PR / RPThp;
    const bool act8 = d.out_mode == TFMQ_OUT_F16 && vec8_ok && !transposed && tid < NTRhp * TPR8p;
    uint4 rpre8[RPThp];
    if (act8 && d.residual && d.res_f16) {
      const int nc = (n0 + (tid % TPR8p) * 8) < d.Cout ? (n0 + (tid % TPR8p) * 8) : 0;
#pragma unroll
      for (int k = 0; k < RPThp; ++k) {
        const int mm = m0 + pass * PR + (tid / TPR8p) * RPThp + k;
        const int mc = mm < p.M ? mm : p.M - 1;
        rpre8[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(d.residual) + static_cast<size_t>(mc) * d.Cout + nc);
      }
    }
    __syncthreads();  // previous pass fully consumed (also orders the main loop's LDS reads before the overwrite)
    if (seg && pass == 0) {
      for (int o = tid; o < BN; o += 256) ldsG[o] = make_float2(0.0f, 0.0f);
    }
    // phase 1: registers -> LDS (row pitch a compile-time constant on either path: the 64 store offsets fold
    // into ds_write immediates)
    auto phase1 = [&](auto pitch_tag) {
      constexpr int PITCH = decltype(pitch_tag)::value;
#pragma unroll
      for (int i = 0; i < WM_TILES; ++i) {
        const int tile_row0 = (wm * WM_TILES + i) * 32;
        if (tile_row0 / PR != pass) continue;
#pragma unroll
        for (int j = 0; j < WN_TILES; ++j) {
          const int col = (wn * WN_TILES + j) * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = tile_row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float v;
            if constexpr (INT8) {
              v = sc_[j] * static_cast<float>(acc[i][j][r] + corr_[j]) + bias_[j];
            } else {
              v = d.wscale ? sc_[j] * acc[i][j][r] + bias_[j] : acc[i][j][r] + bias_[j];
            }
            ldsO[(row - pass * PR) * PITCH + col] = v;
          }
        }
      }
    };
    if (transposed) phase1(std::integral_constant<int, BN + 1>{});
    else phase1(std::integral_constant<int, LDO>{});
    __syncthreads();
    if constexpr (INT8 && BN == 128) {
      if (d.out_mode == TFMQ_OUT_GEGLU_Q8) {
        // tile columns [0,64) = value, [64,128) = gate of the same 64 output channels:
        // yq = quant(value * gelu(gate)), gelu exact (erf), the arithmetic of k_geglu
        const int inner = d.Cout >> 1, prow = tid >> 2, g16 = (tid & 3) * 16;      // PR = 64 rows: one row x 16 outputs per thread
        const int m = m0 + pass * PR + prow;
        if (m < p.M) {
          unsigned w[4];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 a = *reinterpret_cast<const float4*>(ldsO + prow * LDO + g16 + 4 * q4);
            const float4 g = *reinterpret_cast<const float4*>(ldsO + prow * LDO + 64 + g16 + 4 * q4);
            const f2 o01 = f2{a.x, a.y} * gelu2(f2{g.x, g.y}), o23 = f2{a.z, a.w} * gelu2(f2{g.z, g.w});
            w[q4] = pack_q4(o01.x, o01.y, o23.x, o23.y, oqp);
          }
          *reinterpret_cast<uint4*>(d.yq + static_cast<size_t>(m) * inner + (n0 >> 1) + g16) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        continue;
      }
    }
    if (transposed) {
      // yt[b][n - t_col0][t]: a thread takes 4 consecutive pixels of one channel -> one 8-byte store; the PR/4
      // threads of a channel write PR*2 contiguous bytes
      constexpr int TPC = PR / 4, CPI = 256 / TPC;
      const int cv = d.Cout - d.t_col0;
      const int rg = tid % TPC;
#pragma unroll
      for (int it = 0; it < BN / CPI; ++it) {
        const int col = it * CPI + tid / TPC;
        const int m = m0 + pass * PR + 4 * rg, n = n0 + col;
        if (m >= p.M || n >= d.Cout) continue;
        const float* src = ldsO + (4 * rg) * ldo + col;
        const __half2 lo = __floats2half2_rn(src[0], src[ldo]), hi = __floats2half2_rn(src[2 * ldo], src[3 * ldo]);
        uint2 u;
        u.x = *reinterpret_cast<const unsigned*>(&lo);
        u.y = *reinterpret_cast<const unsigned*>(&hi);
        const int b = m / hw, t = m - b * hw;
        *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(d.yt) + (static_cast<size_t>(b) * cv + (n - d.t_col0)) * hw + t) = u;
      }
      continue;
    }
    if (q8 && vec16_ok) {
      // int8 output (the consumer quantizer's bins): items of one row x 16 channels, balanced over the 256 threads
      constexpr int TPR16 = BN / 16, NTR16 = 256 / TPR16, RPT16 = PR / NTR16;
      static_assert(RPT16 >= 1, "rows per thread (int8 output)");
      const int tr16 = tid / TPR16, c16 = (tid % TPR16) * 16;
#pragma unroll
      for (int k = 0; k < RPT16; ++k) {
        const int prow = tr16 * RPT16 + k;
        const int m = m0 + pass * PR + prow, n = n0 + c16;
        if (m >= p.M || n >= d.Cout) continue;
        unsigned w[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float4 v = *reinterpret_cast<const float4*>(ldsO + prow * LDO + c16 + 4 * q4);
          if (rowadd) {
            const float4 a = *reinterpret_cast<const float4*>(rowadd + static_cast<size_t>(m / hw) * d.rowadd_ld + n + 4 * q4);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          if (d.residual) {
            const float4 a = load_res4(d, m, n + 4 * q4);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          w[q4] = pack_q4(v.x, v.y, v.z, v.w, oqp);
        }
        *reinterpret_cast<uint4*>(d.yq + static_cast<size_t>(m) * d.Cout + n) = make_uint4(w[0], w[1], w[2], w[3]);
      }
      continue;
    }
    if (d.out_mode == TFMQ_OUT_F16 && vec8_ok) {
      // fp16 activation stream: items of 8 channels (16 bytes out, 16 bytes of fp16 residual in); the thread rows keep
      // the 4-wide mapping's NTR x RPT so that the statistics grouping -- and with it the summation order -- is the
      // same for every output mode and tile shape; threads beyond NTR * BN/8 only join the barriers
      constexpr int TPR8 = BN / 8;
      constexpr int RPTh = RPT == 8 ? 4 : RPT;      // an 8-row thread row splits into its two 4-row halves: all 256 threads work
      constexpr int NTRh = PR / RPTh;
      static_assert(NTRh * TPR8 <= 256, "fp16 store pass mapping");
      if (act8) {
        const int tr8 = tid / TPR8, c8 = (tid % TPR8) * 8;
        float ps[1][8], pss[1][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ps[0][q] = pss[0][q] = 0.0f;
#pragma unroll
        for (int k = 0; k < RPTh; ++k) {
          const int prow = tr8 * RPTh + k;
          const int m = m0 + pass * PR + prow, n = n0 + c8;
          if (m >= p.M || n >= d.Cout) continue;
          const float4 v0 = *reinterpret_cast<const float4*>(ldsO + prow * LDO + c8);
          const float4 v1 = *reinterpret_cast<const float4*>(ldsO + prow * LDO + c8 + 4);
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          if (rowadd) {
            const float4 a0 = *reinterpret_cast<const float4*>(rowadd + static_cast<size_t>(m / hw) * d.rowadd_ld + n);
            const float4 a1 = *reinterpret_cast<const float4*>(rowadd + static_cast<size_t>(m / hw) * d.rowadd_ld + n + 4);
            v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
          }
          if (d.residual) {
            if (d.res_f16) {
              const unsigned uw[4] = {rpre8[k].x, rpre8[k].y, rpre8[k].z, rpre8[k].w};
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&uw[q]));
                v[2 * q] += f.x;
                v[2 * q + 1] += f.y;
              }
            } else {
              const float4 a0 = load_res4(d, m, n), a1 = load_res4(d, m, n + 4);
              v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
            }
          }
          *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(d.y) + static_cast<size_t>(m) * d.ldy + d.y_coff + n) =
              make_uint4(pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7]));
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            ps[0][q] += v[q];
            pss[0][q] += v[q] * v[q];
          }
        }
        if (seg) {      // one partial per 4 rows: a group = (rows 0..3 in order) + (rows 4..7 in order), as everywhere
          float2* pp = ldsP + tr8 * BN + c8;
#pragma unroll
          for (int q = 0; q < 8; ++q) pp[q] = make_float2(ps[0][q], pss[0][q]);
        }
      }
      if (seg) reduce_stats(pass, std::integral_constant<int, 8 / RPTh>{});
      continue;
    }
    if (d.out_mode == TFMQ_OUT_F16 && !vec_ok) {
      __half* yh = reinterpret_cast<__half*>(d.y);
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int prow = tr * RPT + k;
        const int m = m0 + pass * PR + prow;
        const int n = n0 + c4;
        if (m >= p.M || n >= d.Cout) continue;
        const float4 v = *reinterpret_cast<const float4*>(ldsO + prow * LDO + c4);
        __half* dst = yh + static_cast<size_t>(m) * d.ldy + d.y_coff + n;
        if (vec_ok) {
          const __half2 lo = __floats2half2_rn(v.x, v.y), hi = __floats2half2_rn(v.z, v.w);
          uint2 u;
          u.x = *reinterpret_cast<const unsigned*>(&lo);
          u.y = *reinterpret_cast<const unsigned*>(&hi);
          *reinterpret_cast<uint2*>(dst) = u;
        } else {
          const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (n + q < d.Cout) dst[q] = __float2half_rn(e[q]);
        }
      }
      continue;
    }
    // phase 2: whole rows out, + temb row + residual, + statistics
    // Statistics are summed in ONE order whatever the tile shape (the shape is picked per batch size, the result must
    // not depend on it): a segment = its 8-row groups added in row order, an 8-row group = (rows 0..3 added in
    // order) + (rows 4..7 added in order).
    constexpr int G4 = RPT / 4;
    static_assert(RPT == 4 || RPT == 8, "rows per thread: one or two 4-row groups");
    float4 ps[G4], pss[G4];
#pragma unroll
    for (int gi = 0; gi < G4; ++gi) ps[gi] = pss[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int prow = tr * RPT + k;
      const int m = m0 + pass * PR + prow;
      const int n = n0 + c4;
      if (m >= p.M || n >= d.Cout) continue;
      float4 v = *reinterpret_cast<const float4*>(ldsO + prow * LDO + c4);
      if (vec_ok) {
        if (rowadd) {
          const float4 a = *reinterpret_cast<const float4*>(rowadd + static_cast<size_t>(m / hw) * d.rowadd_ld + n);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if (d.residual) {
          float4 a;
          if constexpr (RES_PRE) a = rpre[k];
          else a = load_res4(d, m, n);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if (d.out_mode == TFMQ_OUT_F16) {      // fp16 activation stream: the statistics below still see the fp32 values
          const __half2 lo = __floats2half2_rn(v.x, v.y), hi = __floats2half2_rn(v.z, v.w);
          uint2 u;
          u.x = *reinterpret_cast<const unsigned*>(&lo);
          u.y = *reinterpret_cast<const unsigned*>(&hi);
          *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(d.y) + static_cast<size_t>(m) * d.ldy + d.y_coff + n) = u;
        } else if (q8) {     // the only consumer is the next QuantLayer's activation quantizer: write its bins
          char4 q;
          q = quant_char4(v.x, v.y, v.z, v.w, make_quantp(oqp));
          *reinterpret_cast<char4*>(d.yq + static_cast<size_t>(m) * d.Cout + n) = q;
        } else {
          *reinterpret_cast<float4*>(d.y + static_cast<size_t>(m) * d.ldy + d.y_coff + n) = v;
        }
      } else {
        float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (n + q >= d.Cout) { e[q] = 0.0f; continue; }
          if (rowadd) e[q] += rowadd[static_cast<size_t>(m / hw) * d.rowadd_ld + n + q];
          if (d.residual) e[q] += d.residual[static_cast<size_t>(m) * d.Cout + n + q];
          d.y[static_cast<size_t>(m) * d.ldy + d.y_coff + n + q] = e[q];
        }
        v = make_float4(e[0], e[1], e[2], e[3]);
      }
      ps[k >> 2].x += v.x; ps[k >> 2].y += v.y; ps[k >> 2].z += v.z; ps[k >> 2].w += v.w;
      pss[k >> 2].x += v.x * v.x; pss[k >> 2].y += v.y * v.y; pss[k >> 2].z += v.z * v.z; pss[k >> 2].w += v.w * v.w;
    }
    if (seg) {
      float4 s = ps[0], ss = pss[0];
      if constexpr (G4 == 2) {
        s.x += ps[1].x; s.y += ps[1].y; s.z += ps[1].z; s.w += ps[1].w;
        ss.x += pss[1].x; ss.y += pss[1].y; ss.z += pss[1].z; ss.w += pss[1].w;
      }
      float2* pp = ldsP + tr * BN + c4;     // one partial per thread-row: 8 rows (RPT 8) or 4 rows (RPT 4)
      pp[0] = make_float2(s.x, ss.x);
      pp[1] = make_float2(s.y, ss.y);
      pp[2] = make_float2(s.z, ss.z);
      pp[3] = make_float2(s.w, ss.w);
      reduce_stats(pass, std::integral_constant<int, 8 / RPT>{});
    }
  }
}

// ================================================================================================
// main path: LDS-DMA pipeline.  F16 = false: w4a8 (int8 operands, 64 channels per K-step, i8 MFMA);
// F16 = true: un-quantised layers on fp16 operands (fp16 NHWC activations written by the producing kernel, 32
// channels per K-step, f16 MFMA, fp32 accumulation) -- the same 64-byte rows, swizzle and pipeline.
// ================================================================================================
template <bool F16, int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES, bool RES_PRE = false, int NST = 3>
__global__ __launch_bounds__(256, (WM_TILES * WN_TILES > 4 || NST > 3) ? 2 : 3) void k_conv_dma(ConvP p) {
  constexpr int BM = WAVES_M * WM_TILES * 32;
  constexpr int BN = WAVES_N * WN_TILES * 32;
  constexpr int STAGE = (BM + BN) * 64;          // bytes of one K-step: A tile then B tile, 64-byte rows
  // NST stages: multiply s while s+1 .. s+NST-1 have landed or are landing.  3 for long K loops; 5 ("whole K resident")
  // for pointwise layers of K <= 320, whose five K-steps are all requested before the first one is multiplied -- a short
  // loop at prefetch distance 2 waits out most of a DMA latency at every step
  constexpr int A_CH = BM / 64;                   // 1-KiB (16-row) DMA pieces per wave
  constexpr int B_CH = BN >= 64 ? BN / 64 : 1;    // (BN = 32: two pieces, waves 2/3 repeat them)
  constexpr int NLOAD = A_CH + B_CH;              // DMA instructions per wave per K-step
  constexpr int MAXT = 9;
  constexpr int LDS_MAIN = NST * STAGE;
  constexpr int LDS_EPI = epi_lds_bytes<BM, BN>();
  // 256-row tiles (stride 1, no fused upsample -- the launcher's rule): the offset table shrinks to {offset of the
  // window origin, bit mask of in-image taps} per row, so two workgroups (2 x 80 KiB) still share a CU
  constexpr bool COMPACT = BM > 128;
  constexpr int TAB_BYTES = NST > 3 ? 0 : (COMPACT ? BM * 8 : MAXT * BM * 4);      // (the whole-K variant is pointwise: no table)
  // the offset table is dead once the K loop ends: the epilogue's staging may overlay it
  constexpr int LDS_ALL = (LDS_MAIN + TAB_BYTES) > LDS_EPI ? (LDS_MAIN + TAB_BYTES) : LDS_EPI;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_ALL];
  int* tab = reinterpret_cast<int*>(lds + LDS_MAIN);  // [tap][row] byte offset of the input pixel, -1 = padding
  TFMQ_MARK(0);

  const tfmq_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  // split-K (tfmq_conv_desc.ksplit, w4a8 only): `ksplit` consecutive block ids -- neighbours on one XCD under the tile order --
  // share an output tile, slice z multiplying K-steps [nsteps z / k, nsteps (z + 1) / k)
  const int bid0 = xcd_tile_id();
  const int ksp = F16 ? 1 : p.ksplit;
  const int bid = ksp > 1 ? bid0 / ksp : bid0;
  const int kslice = bid0 - bid * ksp;
  const int s_begin = ksp > 1 ? static_cast<int>(static_cast<long>(p.nsteps) * kslice / ksp) : 0;
  const int n_my = (ksp > 1 ? static_cast<int>(static_cast<long>(p.nsteps) * (kslice + 1) / ksp) : p.nsteps) - s_begin;
  const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // two dependent global loads (step counter -> table row): issued first, consumed after the offset table is built
  float2 aqp = make_float2(1.0f, 0.0f);

  // ---- pixel offset table (any stride / padding / fused 2x upsample).  A Linear / 1x1 stride-1 conv needs none:
  // pixel m reads input pixel m (most launches of a transformer UNet: no table, no barrier in the prologue)
  const bool pointwise = d.KH * d.KW == 1 && d.stride == 1 && !d.up2x && d.pad_t == 0 && d.pad_l == 0;
  // a pointwise layer needs the activation quantizer's parameters only in its epilogue: requested after its first DMA
  // pieces (below); a layer with padded taps needs the zero point for the pad rows first
  if constexpr (!F16) {
    if (!pointwise) aqp = load_qparam(d.aq);
  }
  if (!pointwise && COMPACT) {
    const int taps = d.KH * d.KW, hw = d.Ho * d.Wo;
    for (int row = tid; row < BM; row += 256) {
      const int m = m0 + row;
      int base = 0, mask = 0;
      if (m < p.M) {
        const int b = m / hw, r = m - b * hw;
        const int ho = r / d.Wo, wo = r - ho * d.Wo;
        base = ((b * d.H + ho - d.pad_t) * d.W + wo - d.pad_l) * d.Cin * (F16 ? 2 : 1);
        for (int tap = 0; tap < taps; ++tap) {
          const int hi = ho + tap / d.KW - d.pad_t, wi = wo + tap % d.KW - d.pad_l;
          mask |= (hi >= 0 && hi < d.H && wi >= 0 && wi < d.W) ? (1 << tap) : 0;
        }
      }
      reinterpret_cast<int2*>(tab)[row] = make_int2(base, mask);
    }
  } else if (!pointwise) {
    const int taps = d.KH * d.KW, hw = d.Ho * d.Wo;
    for (int idx = tid; idx < taps * BM; idx += 256) {
      const int tap = idx / BM, row = idx - tap * BM;
      const int m = m0 + row;
      int off = -1;
      if (m < p.M) {
        const int b = m / hw, r = m - b * hw;
        const int ho = r / d.Wo, wo = r - ho * d.Wo;
        const int kh = tap / d.KW, kw = tap - kh * d.KW;
        int hi = ho * d.stride + kh - d.pad_t, wi = wo * d.stride + kw - d.pad_l;
        if (hi >= 0 && hi < p.Hv && wi >= 0 && wi < p.Wv) {
          if (d.up2x) {
            hi >>= 1;
            wi >>= 1;
          }
          off = ((b * d.H + hi) * d.W + wi) * d.Cin * (F16 ? 2 : 1);  // bytes; < 2^31, checked by the launcher
        }
      }
      tab[idx] = off;
    }
  }

  // ---- per-thread DMA sources.  Lane l of a piece lands on LDS row l/4, physical slot l%4, so it must fetch
  // logical slot (l%4) ^ swizzle(row): the swizzle lives on the SOURCE address and on the fragment reads.
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));
  const unsigned char* xb = static_cast<const unsigned char*>(d.x);
  int a_row[A_CH], a_col[A_CH], a_off[A_CH];
  unsigned a_dst[A_CH], b_dst[B_CH];
  const unsigned char* b_ptr[B_CH];
#pragma unroll
  for (int it = 0; it < A_CH; ++it) {
    const int piece = wid * A_CH + it;
    a_row[it] = piece * 16 + (lane >> 2);
    a_col[it] = ((lane & 3) ^ ((a_row[it] >> 2) & 3)) * 16;
    a_dst[it] = __builtin_amdgcn_readfirstlane(piece * 1024);
    a_off[it] = (pointwise && m0 + a_row[it] < p.M) ? (m0 + a_row[it]) * d.Cin * (F16 ? 2 : 1) : -1;
  }
#pragma unroll
  for (int it = 0; it < B_CH; ++it) {
    const int piece = BN >= 64 ? wid * B_CH + it : (wid & 1);
    const int row = piece * 16 + (lane >> 2);
    int n = n0 + row;
    if constexpr (F16) {       // fp16 weights [cout][tap][cin] row-major (tfmq_pack_w_f16; cin % 32 == 0 here)
      n = n < d.Cout ? n : d.Cout - 1;
      b_ptr[it] = static_cast<const unsigned char*>(d.w) + static_cast<size_t>(n) * (d.KH * d.KW) * p.cin_pad * 2 +
                  ((lane & 3) ^ ((row >> 2) & 3)) * 16;
    } else {
      n = n < p.cout_pad ? n : p.cout_pad - 1;
      b_ptr[it] = static_cast<const unsigned char*>(d.w) +
                  (static_cast<size_t>(n / 32) * p.nsteps * 32 + (n % 32)) * 64 + ((lane & 3) ^ ((row >> 2) & 3)) * 16;
    }
    b_dst[it] = __builtin_amdgcn_readfirstlane(BM * 64 + piece * 1024);
  }
  if (!pointwise) __syncthreads();  // tab visible
  // real zero == bin za  ->  stored byte za-128; its 64-byte row in the pad table feeds the padded taps
  // (fp16 operands: row 0 = zeros).  A pointwise layer has no padded taps -- only rows past M, whose results are never
  // stored -- so its first DMA does not wait for the two dependent loads behind aqp
  const unsigned char* padp = p.pad_table;
  if (!pointwise && !F16) padp += (static_cast<unsigned>(static_cast<int>(aqp.y) - 128) & 0xffu) * 64;

  int i_tap = s_begin / p.chunks, i_chunk = s_begin - (s_begin / p.chunks) * p.chunks;
  bool tap_fresh = true;          // (a K slice may start inside a tap)
  auto issue = [&](int s, int stage) {
    if ((i_chunk == 0 || tap_fresh) && !pointwise) {
      tap_fresh = false;
      if constexpr (COMPACT) {
        const int tapoff = ((i_tap / d.KW) * d.W + (i_tap % d.KW)) * d.Cin * (F16 ? 2 : 1);
#pragma unroll
        for (int it = 0; it < A_CH; ++it) {
          const int2 e = reinterpret_cast<const int2*>(tab)[a_row[it]];
          a_off[it] = ((e.y >> i_tap) & 1) ? e.x + tapoff : -1;
        }
      } else {
#pragma unroll
        for (int it = 0; it < A_CH; ++it) a_off[it] = tab[i_tap * BM + a_row[it]];
      }
    }
    const int c0 = i_chunk * 64;
    const unsigned sbase = lds0 + stage * STAGE;
#pragma unroll
    for (int it = 0; it < A_CH; ++it) {
      const unsigned char* src = a_off[it] >= 0 ? xb + static_cast<size_t>(static_cast<unsigned>(a_off[it])) + c0 + a_col[it]
                                                : padp + a_col[it];
      glds16(src, sbase + a_dst[it]);
    }
    const size_t b_off = F16 ? static_cast<size_t>(i_tap * p.cin_pad + i_chunk * 32) * 2 : static_cast<size_t>(s) * 2048;
#pragma unroll
    for (int it = 0; it < B_CH; ++it) glds16(b_ptr[it] + b_off, sbase + b_dst[it]);
    if (++i_chunk == p.chunks) {
      i_chunk = 0;
      ++i_tap;
    }
  };

  using acc_t = typename std::conditional<F16, v16f, v16i>::type;
  acc_t acc[WM_TILES][WN_TILES];
#pragma unroll
  for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
    for (int j = 0; j < WN_TILES; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  auto compute = [&](int stage) {
    const unsigned char* sa = lds + stage * STAGE;
    const unsigned char* sb = sa + BM * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v4i af[WM_TILES], bf[WN_TILES];
      const int kslot = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < WM_TILES; ++i)
        af[i] = *reinterpret_cast<const v4i*>(sa + swz((wm * WM_TILES + i) * 32 + (lane & 31), kslot));
#pragma unroll
      for (int j = 0; j < WN_TILES; ++j)
        bf[j] = *reinterpret_cast<const v4i*>(sb + swz((wn * WN_TILES + j) * 32 + (lane & 31), kslot));
#pragma unroll
      for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
        for (int j = 0; j < WN_TILES; ++j) {
          if constexpr (F16)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<v8h*>(&af[i]), *reinterpret_cast<v8h*>(&bf[j]),
                                                               acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  TFMQ_MARK(1);
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < n_my) issue(s_begin + s0, s0);
  // requested behind the first DMA pieces, consumed in the epilogue (the counted waits of the loop only become more
  // conservative on its first step: these loads are younger than the pieces they must not overtake)
  if constexpr (!F16) {
    if (pointwise) aqp = load_qparam(d.aq);
  }
  const EpiCols<WN_TILES> ec = load_epi_cols<!F16, WAVES_N, WN_TILES>(p, n0);
  int st_c = 0, st_i = NST - 1;
  for (int s = 0; s < n_my; ++s) {
    // this wave's pieces of K-step s have landed (those of the up to NST-2 later steps may still be in flight) ...
    const int ahead = n_my - 1 - s;
    if (ahead >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NLOAD) : "memory");
    else if (NST > 3 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLOAD) : "memory");
    else if (NST > 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and after the barrier every wave's have, and nobody still reads the stage refilled below
    asm volatile("s_barrier" ::: "memory");
    // diagnostics builds (-DTFMQ_DBG_NO_DMA / -DTFMQ_DBG_NO_MFMA, results are garbage): the K loop without its
    // L2 -> LDS traffic, or without its fragment reads and MFMAs -- DESIGN.md section 4 quotes both
#ifdef TFMQ_DBG_NO_DMA
    if (s + NST - 1 < n_my && s < 1) issue(s_begin + s + NST - 1, st_i);
#else
    if (s + NST - 1 < n_my) issue(s_begin + s + NST - 1, st_i);
#endif
#ifndef TFMQ_DBG_NO_MFMA
    compute(st_c);
#endif
    st_c = st_c == NST - 1 ? 0 : st_c + 1;
    st_i = st_i == NST - 1 ? 0 : st_i + 1;
  }

  TFMQ_MARK(2);
  if constexpr (!F16) {
    if (ksp > 1) {
      // Publish this slice's int32 partial sums, take a ticket; the tile's last arriver adds the other slabs to its registers and
      // runs the epilogue (integer sums: any order gives the bits of the unsplit launch).  Hand-off per
      // cdna_hip_programming.md section 5 (in-launch split-K): write-through (sc1) slab stores at the accumulators' natural 4-byte
      // width = relaxed agent-scope atomic stores, every wave drains its stores, workgroup barrier, ONE relaxed agent-scope ticket;
      // the reducer reads the slabs with sc1 (relaxed agent-scope) loads -- correct for any placement of the slices over XCDs / CUs.
      // Slab layout: lane-linear ([wave][mfma tile][register][lane]): every store / load instruction moves 256 contiguous bytes.
      constexpr int TILE_INTS = BM * BN;
      int* slab0 = p.ks_ws + static_cast<size_t>(bid) * ksp * TILE_INTS + (wid * WM_TILES * WN_TILES) * 16 * 64 + lane;
      int* mine = slab0 + static_cast<size_t>(kslice) * TILE_INTS;
#pragma unroll
      for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
        for (int j = 0; j < WN_TILES; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __hip_atomic_store(mine + ((i * WN_TILES + j) * 16 + r) * 64, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                       // every wave's slab stores are out; nobody reads the pipeline stages any more
      int* flag = reinterpret_cast<int*>(lds);
      if (tid == 0) *flag = __hip_atomic_fetch_add(p.ks_cnt + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int ticket = *flag;
      __syncthreads();                       // (the epilogue stages through the same LDS)
      if (ticket != ksp - 1) return;
      if (tid == 0) __hip_atomic_store(p.ks_cnt + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
      for (int z = 0; z < ksp; ++z) {
        if (z == kslice) continue;
        const int* other = slab0 + static_cast<size_t>(z) * TILE_INTS;
#pragma unroll
        for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
          for (int j = 0; j < WN_TILES; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[i][j][r] += __hip_atomic_load(other + ((i * WN_TILES + j) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  conv_epilogue<!F16, WAVES_M, WAVES_N, WM_TILES, WN_TILES, RES_PRE>(p, lds, acc, m0, n0, aqp, static_cast<int>(aqp.y), ec);
  TFMQ_MARK(3);
}

// ================================================================================================
// generic path: f16 layers, and w4a8 shapes outside the DMA loop's domain
// ================================================================================================
template <bool INT8, bool FAST, int CK8, int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES>
__global__ __launch_bounds__(256, 3) void k_conv_igemm(ConvP p) {
  constexpr int BM = WAVES_M * WM_TILES * 32;
  constexpr int BN = WAVES_N * WN_TILES * 32;
  constexpr int CK = INT8 ? CK8 : 32;           // channels per K-step (CK8 = 64 or 32 int8 channels)
  constexpr int SLOTS = INT8 ? CK8 / 16 : 4;    // 16-byte slots used per 64-byte LDS row
  constexpr int KSUB = INT8 ? CK8 / 32 : 2;     // MFMA k-sub-steps per K-step
  constexpr int A_TOTAL = BM * SLOTS;
  constexpr int A_ITEMS = (A_TOTAL + 255) / 256;  // 16-byte LDS items per thread (A)
  constexpr int B_TOTAL = BN * SLOTS;
  constexpr int B_ITEMS = (B_TOTAL + 255) / 256;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");

  // LDS: main loop 2 x (A tile + B tile); the epilogue re-uses the same bytes for its output staging
  // tile + statistics partials.
  constexpr int LDS_MAIN = 2 * (BM + BN) * 64;
  constexpr int LDS_EPI = epi_lds_bytes<BM, BN>();
  constexpr int LDS_BODY = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BODY];
  auto ldsA = [&](int buf) -> unsigned char* { return lds + buf * ((BM + BN) * 64); };
  auto ldsB = [&](int buf) -> unsigned char* { return lds + buf * ((BM + BN) * 64) + BM * 64; };

  const tfmq_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int bid = xcd_tile_id();
  const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  float2 aqp = make_float2(1.0f, 0.0f);
  int za = 0;
  unsigned pad_word = 0;
  if constexpr (INT8) {
    aqp = load_qparam(d.aq);
    za = static_cast<int>(aqp.y);
    const unsigned pb = static_cast<unsigned>(za - 128) & 0xffu;  // real zero == bin za
    pad_word = pb * 0x01010101u;
  }

  // ---- per-thread A rows (fixed for the whole K loop)
  int a_row[A_ITEMS], a_slot[A_ITEMS], a_b[A_ITEMS], a_ho[A_ITEMS], a_wo[A_ITEMS];
  bool a_ok[A_ITEMS], a_in[A_ITEMS];
#pragma unroll
  for (int it = 0; it < A_ITEMS; ++it) {
    const int item = tid + it * 256;
    a_row[it] = item / SLOTS;
    a_slot[it] = item % SLOTS;
    a_in[it] = item < A_TOTAL;
    const int m = m0 + a_row[it];
    a_ok[it] = a_in[it] && m < p.M;
    const int mm = a_ok[it] ? m : 0;
    const int hw = d.Ho * d.Wo;
    a_b[it] = mm / hw;
    const int r = mm - a_b[it] * hw;
    a_ho[it] = r / d.Wo;
    a_wo[it] = r - a_ho[it] * d.Wo;
  }

  // two register sets: the global loads of K-step s+2 are issued while step s is being multiplied
  uint4 a_reg0[A_ITEMS], a_reg1[A_ITEMS];
  uint4 b_reg0[B_ITEMS], b_reg1[B_ITEMS];

  // Fast addressing (stride 1, no fused upsample, whole K-steps): everything that depends on the thread
  // is computed ONCE -- a base pointer per staged item and a bit mask of the taps that fall inside the
  // image -- and a K-step only adds a wave-uniform (scalar) offset.
  const unsigned char* a_base[A_ITEMS];
  unsigned a_mask[A_ITEMS];
  const unsigned char* b_base[B_ITEMS];
  bool b_ok[B_ITEMS];
  if constexpr (FAST) {
    constexpr int ESZ = INT8 ? 1 : 4;  // bytes per input element
#pragma unroll
    for (int it = 0; it < A_ITEMS; ++it) {
      const size_t pix = (static_cast<size_t>(a_b[it]) * d.H + a_ho[it]) * d.W + a_wo[it];
      a_base[it] = static_cast<const unsigned char*>(d.x) + (pix * d.Cin + a_slot[it] * (INT8 ? 16 : 8)) * ESZ;
      unsigned mask = 0;
      for (int t = 0; t < d.KH * d.KW; ++t) {
        const int hi = a_ho[it] + t / d.KW - d.pad_t, wi = a_wo[it] + t % d.KW - d.pad_l;
        if (a_ok[it] && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W) mask |= 1u << t;
      }
      a_mask[it] = mask;
    }
#pragma unroll
    for (int it = 0; it < B_ITEMS; ++it) {
      const int item = tid + it * 256;
      const int n = n0 + item / SLOTS;
      b_ok[it] = item < B_TOTAL && n < d.Cout;
      const int nn = b_ok[it] ? n : 0;
      if constexpr (INT8)
        b_base[it] = static_cast<const unsigned char*>(d.w) +
                     ((static_cast<size_t>(nn / 32) * p.nsteps) * 32 + (nn % 32)) * CK + (item % SLOTS) * 16;
      else
        b_base[it] = static_cast<const unsigned char*>(d.w) +
                     (static_cast<size_t>(nn) * (d.KH * d.KW) * p.cin_pad + (item & 3) * 8) * 2;
    }
  }

  auto load_step = [&](int s, uint4 (&ar)[A_ITEMS], uint4 (&br)[B_ITEMS]) {
    const int tap = s / p.chunks;
    const int c0 = (s - tap * p.chunks) * CK;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    if constexpr (FAST) {
      // wave-uniform byte offsets of this K-step
      const long a_off = (static_cast<long>((kh - d.pad_t) * d.W + (kw - d.pad_l)) * d.Cin + c0) * (INT8 ? 1 : 4);
      const long b_off = static_cast<long>(s) * (INT8 ? 32 * CK : 64);  // int8: 32 rows x CK bytes per K-step (tile-major)
#pragma unroll
      for (int it = 0; it < A_ITEMS; ++it) {
        const bool ok = (a_mask[it] >> tap) & 1u;
        if constexpr (INT8) {
          uint4 v = make_uint4(pad_word, pad_word, pad_word, pad_word);
          if (ok) v = *reinterpret_cast<const uint4*>(a_base[it] + a_off);
          ar[it] = v;
        } else {
          float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
          if (ok) {
            v0 = *reinterpret_cast<const float4*>(a_base[it] + a_off);
            v1 = *reinterpret_cast<const float4*>(a_base[it] + a_off + 16);
          }
          v8h hv = {static_cast<_Float16>(v0.x), static_cast<_Float16>(v0.y), static_cast<_Float16>(v0.z),
                    static_cast<_Float16>(v0.w), static_cast<_Float16>(v1.x), static_cast<_Float16>(v1.y),
                    static_cast<_Float16>(v1.z), static_cast<_Float16>(v1.w)};
          ar[it] = *reinterpret_cast<uint4*>(&hv);
        }
      }
#pragma unroll
      for (int it = 0; it < B_ITEMS; ++it) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (b_ok[it]) v = *reinterpret_cast<const uint4*>(b_base[it] + b_off);
        br[it] = v;
      }
      return;
    }
    // ---- A
#pragma unroll
    for (int it = 0; it < A_ITEMS; ++it) {
      int hi = a_ho[it] * d.stride + kh - d.pad_t;
      int wi = a_wo[it] * d.stride + kw - d.pad_l;
      const bool ok = a_ok[it] && hi >= 0 && hi < p.Hv && wi >= 0 && wi < p.Wv;
      if (d.up2x) {
        hi >>= 1;
        wi >>= 1;
      }
      const size_t pix = (static_cast<size_t>(a_b[it]) * d.H + hi) * d.W + wi;
      if constexpr (INT8) {
        uint4 v = make_uint4(pad_word, pad_word, pad_word, pad_word);
        if (ok) v = *reinterpret_cast<const uint4*>(static_cast<const int8_t*>(d.x) + pix * d.Cin + c0 + a_slot[it] * 16);
        ar[it] = v;
      } else {
        const int c = c0 + a_slot[it] * 8;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 0.0f;
        if (ok) {
          const float* src = static_cast<const float*>(d.x) + pix * d.Cin + c;
          if ((d.Cin & 3) == 0 && c + 8 <= d.Cin) {
            const float4 v0 = *reinterpret_cast<const float4*>(src);
            const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
            f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w;
            f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (c + j < d.Cin) f[j] = src[j];
          }
        }
        v8h hv;
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = static_cast<_Float16>(f[j]);
        ar[it] = *reinterpret_cast<uint4*>(&hv);
      }
    }
    // ---- B
#pragma unroll
    for (int it = 0; it < B_ITEMS; ++it) {
      const int item = tid + it * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (item < B_TOTAL) {
        const int n = n0 + item / SLOTS;
        if (n < d.Cout) {
          if constexpr (INT8)
            v = *reinterpret_cast<const uint4*>(static_cast<const uint8_t*>(d.w) +
                                                ((static_cast<size_t>(n / 32) * p.nsteps + s) * 32 + (n % 32)) * CK +
                                                (item % SLOTS) * 16);
          else
            v = *reinterpret_cast<const uint4*>(static_cast<const __half*>(d.w) +
                                                (static_cast<size_t>(n) * (d.KH * d.KW) + tap) * p.cin_pad + c0 + (item & 3) * 8);
        }
      }
      br[it] = v;
    }
  };

  auto store_step = [&](int buf, uint4 (&ar)[A_ITEMS], uint4 (&br)[B_ITEMS]) {
#pragma unroll
    for (int it = 0; it < A_ITEMS; ++it)
      if (a_in[it]) *reinterpret_cast<uint4*>(ldsA(buf) + swz(a_row[it], a_slot[it])) = ar[it];
#pragma unroll
    for (int it = 0; it < B_ITEMS; ++it) {
      const int item = tid + it * 256;
      if (item < B_TOTAL) *reinterpret_cast<uint4*>(ldsB(buf) + swz(item / SLOTS, item % SLOTS)) = br[it];
    }
  };

  using acc_t = typename std::conditional<INT8, v16i, v16f>::type;
  acc_t acc[WM_TILES][WN_TILES];
#pragma unroll
  for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
    for (int j = 0; j < WN_TILES; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
      uint4 af[WM_TILES], bf[WN_TILES];
      const int kslot = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < WM_TILES; ++i)
        af[i] = *reinterpret_cast<const uint4*>(ldsA(buf) + swz((wm * WM_TILES + i) * 32 + (lane & 31), kslot));
#pragma unroll
      for (int j = 0; j < WN_TILES; ++j)
        bf[j] = *reinterpret_cast<const uint4*>(ldsB(buf) + swz((wn * WN_TILES + j) * 32 + (lane & 31), kslot));
#pragma unroll
      for (int i = 0; i < WM_TILES; ++i)
#pragma unroll
        for (int j = 0; j < WN_TILES; ++j) {
          if constexpr (INT8) {
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<v4i*>(&af[i]),
                                                              *reinterpret_cast<v4i*>(&bf[j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<v8h*>(&af[i]),
                                                               *reinterpret_cast<v8h*>(&bf[j]), acc[i][j], 0, 0, 0);
          }
        }
    }
  };

  load_step(0, a_reg0, b_reg0);
  if (p.nsteps > 1) load_step(1, a_reg1, b_reg1);
  for (int s = 0; s < p.nsteps; s += 2) {
    store_step(0, a_reg0, b_reg0);
    LDS_BARRIER();
    if (s + 2 < p.nsteps) load_step(s + 2, a_reg0, b_reg0);
    compute(0);
    if (s + 1 < p.nsteps) {
      store_step(1, a_reg1, b_reg1);
      LDS_BARRIER();
      if (s + 3 < p.nsteps) load_step(s + 3, a_reg1, b_reg1);
      compute(1);
    }
  }

  const EpiCols<WN_TILES> ec = load_epi_cols<INT8, WAVES_N, WN_TILES>(p, n0);
  conv_epilogue<INT8, WAVES_M, WAVES_N, WM_TILES, WN_TILES>(p, lds, acc, m0, n0, aqp, za, ec);
}

template <bool INT8>
static int launch_conv(tfmq_handle h, const tfmq_conv_desc* dd, void* stream) {
  TFMQ_CHECK_ARG(h, h && dd, "conv: null pointer");
  const tfmq_conv_desc& d = *dd;
  TFMQ_CHECK_ARG(h, d.x && d.w && (d.y || d.out_mode == TFMQ_OUT_GEGLU_Q8 || d.out_mode == TFMQ_OUT_GEGLU_Q8_FAST || d.out_mode == TFMQ_OUT_Q8), "conv: null operand");
  TFMQ_CHECK_ARG(h, d.B > 0 && d.H > 0 && d.W > 0 && d.Cin > 0 && d.Cout > 0 && d.KH > 0 && d.KW > 0 && d.stride > 0,
                 "conv: bad geometry");
  TFMQ_CHECK_ARG(h, d.Ho > 0 && d.Wo > 0 && d.ldy >= d.Cout + d.y_coff, "conv: bad output geometry");
  TFMQ_CHECK_ARG(h, d.out_mode == TFMQ_OUT_F32 || d.out_mode == TFMQ_OUT_Q8 || d.out_mode == TFMQ_OUT_F16 || (!d.rowadd && !d.residual),
                 "conv: rowadd / residual need out_mode F32, F16 or Q8");
  TFMQ_CHECK_ARG(h, d.out_mode == TFMQ_OUT_F32 || d.out_mode == TFMQ_OUT_F16 || !d.stats, "conv: stats need out_mode F32 or F16");
  TFMQ_CHECK_ARG(h, d.out_mode != TFMQ_OUT_Q8 || (d.yq && d.oq.qtable && d.Cout % 4 == 0 && (!d.rowadd || d.rowadd_ld % 4 == 0)),
                 "conv: Q8 output needs yq, oq and Cout % 4 == 0");
  TFMQ_CHECK_ARG(h, (d.out_mode != TFMQ_OUT_GEGLU_Q8 && d.out_mode != TFMQ_OUT_GEGLU_Q8_FAST) ||
                        (INT8 && d.yq && d.oq.qtable && d.KH == 1 && d.KW == 1 && d.Cout % 128 == 0 && d.Cout / 2 % 4 == 0),
                 "conv: GEGLU epilogue needs a w4a8 Linear with Cout % 128 == 0, yq and oq");
  TFMQ_CHECK_ARG(h, d.out_mode >= 0 && d.out_mode <= 4, "conv: bad out_mode");
  TFMQ_CHECK_ARG(h, !d.yt || (d.out_mode == TFMQ_OUT_F16 && d.t_col0 >= 0 && d.t_col0 % 128 == 0 && d.t_col0 < d.Cout &&
                              (d.Ho * d.Wo) % 4 == 0),
                 "conv: transposed region needs out_mode F16, t_col0 % 128 == 0 and Ho*Wo % 4 == 0");
  TFMQ_CHECK_ARG(h, !d.stats || ((d.stats_seg == 16 || d.stats_seg == 32 || d.stats_seg == 64 || d.stats_seg == 128) &&
                                 (d.Ho * d.Wo) % d.stats_seg == 0),
                 "conv: stats_seg must be 16/32/64/128 and divide Ho*Wo");
  ConvP p;
  p.d = d;
  p.issue_split = 0;
  p.M = d.B * d.Ho * d.Wo;
  p.Hv = d.up2x ? 2 * d.H : d.H;
  p.Wv = d.up2x ? 2 * d.W : d.W;
  p.Ktot = d.KH * d.KW * d.Cin;
  p.cout_pad = (d.Cout + 31) / 32 * 32;
  p.pad_table = h->pad_table;
  if (INT8) {
    TFMQ_CHECK_ARG(h, d.Cin % 32 == 0, "conv_w4a8: Cin must be a multiple of 32");
    TFMQ_CHECK_ARG(h, d.wmeta && d.wscale && d.aq.qtable, "conv_w4a8: wmeta/wscale/aq required");
    p.chunks = d.Cin % 64 == 0 ? d.Cin / 64 : d.Cin / 32;
    p.cin_pad = d.Cin;
    // Cin % 64 == 32 with the K-padded operand (tfmq_conv_desc.w64): 64-channel K-steps, the LDS-DMA kernels.  Only when the DMA
    // kernels are sure to take the launch (the register-staged fallback reads the 32-channel-step operand d.w)
    if (d.Cin % 64 != 0 && d.w64 && d.KH * d.KW <= 9 && static_cast<size_t>(d.B) * d.H * d.W * d.Cin < (static_cast<size_t>(1) << 31)) {
      p.chunks = (d.Cin + 63) / 64;
      p.d.w = d.w64;
    }
  } else {
    p.chunks = (d.Cin + 31) / 32;
    p.cin_pad = p.chunks * 32;
  }
  p.nsteps = d.KH * d.KW * p.chunks;
  const bool geglu = d.out_mode == TFMQ_OUT_GEGLU_Q8;  // epilogue pairs columns inside one 128-wide tile
  const bool narrow = d.Cout <= 32;
  // small-M layers (4x4 / 8x8 feature maps): 128x128 tiles leave most of the 256 CUs idle -> 64x64 tiles
  // (a statistics segment must not span tiles, so 128-pixel segments keep the 128-row tile)
  const bool small_ok = !narrow && !geglu && !(d.stats && d.stats_seg > 64);
  // large-M layers: 256 x 128 tiles move a quarter fewer L2 -> LDS bytes per MFMA (the K loop is bound by that path)
  const bool dma8 = INT8 && p.chunks == (d.Cin + 63) / 64 && d.KH * d.KW <= 9 &&
                    static_cast<size_t>(d.B) * d.H * d.W * d.Cin < (static_cast<size_t>(1) << 31);
  // (the fp16-activation DMA kernel shares the pipeline, hence the tile shapes)
  const bool dma16 = !INT8 && d.x_f16 && d.Cin % 32 == 0 && d.KH * d.KW <= 9;
  const bool big_ok = (dma8 || dma16) && !narrow && d.stride == 1 && !d.up2x;
  const long tiles128 = static_cast<long>((p.M + 127) / 128) * ((d.Cout + 127) / 128);
  bool small, big;
  const bool half_n = d.tile == TFMQ_TILE_128x64 && (dma8 || dma16) && small_ok;     // 128 x 64: no column waste for Cout = 64 (2k+1)
  if (half_n) {
    small = big = false;
  } else if (d.tile == TFMQ_TILE_128 || (d.tile == TFMQ_TILE_64 && small_ok) || (d.tile == TFMQ_TILE_256 && big_ok)) {
    // the caller measured the variants for this launch (ops.set_conv_autotune) -- tile quantisation against
    // 256 CUs x 2..5 resident blocks is not something a closed-form rule gets right for every batch size
    small = d.tile == TFMQ_TILE_64;
    big = d.tile == TFMQ_TILE_256;
  } else {
    small = small_ok && tiles128 < 2L * h->cu_count;
    // (measured: pays for the fp16-output token Linears -- q/k/v projections, -16 % -- whose epilogue is light; layers
    // with the fp32 residual epilogue lose more from 2 instead of 3 resident workgroups than the K loop gains)
    big = big_ok && !small && d.out_mode == TFMQ_OUT_F16 && tiles128 >= 4L * h->cu_count;
  }
  if constexpr (!INT8) {
    if ((d.tile == TFMQ_TILE_AUTO || d.tile == TFMQ_TILE_DIRECT || d.x2) && dma16 && launch_conv_lin_f16(h, p, as_stream(stream))) {
      TFMQ_LAUNCH_CHECK(h);
      return TFMQ_OK;
    }
  }
  TFMQ_CHECK_ARG(h, !d.x2, "conv2d: a second input source (x2) is only read by the fp16 pointwise kernel (tfmq_conv_desc.x2)");
  if constexpr (INT8) {
    // token Linears / 1x1 convs writing fp16, int8 or GEGLU-int8: the register-direct-epilogue kernel (conv_lin.hip)
    if ((d.tile == TFMQ_TILE_AUTO || d.tile == TFMQ_TILE_DIRECT || d.tile == TFMQ_TILE_DIRECT256) && d.wmeta && d.wscale && d.aq.qtable &&
        launch_conv_lin(h, p, as_stream(stream), d.tile == TFMQ_TILE_DIRECT256)) {
      TFMQ_LAUNCH_CHECK(h);
      return TFMQ_OK;
    }
    if (d.out_mode == TFMQ_OUT_GEGLU_Q8_FAST) {      // only the register-direct kernel carries the consumer-sized GELU
      h->err = "conv_w4a8: TFMQ_OUT_GEGLU_Q8_FAST needs a launch the register-direct pointwise kernel takes (tile AUTO / DIRECT, Cin % 32 == 0)";
      return TFMQ_ERR_UNSUPPORTED;
    }
  }
  if constexpr (INT8) {
    // 3x3 / stride 1 / pad 1 on a grid that fills the chip: the slab kernel (conv_slab.hip), unless the caller pinned
    // another tile shape
    if ((d.tile == TFMQ_TILE_AUTO || d.tile == TFMQ_TILE_SLAB) && dma8 &&
        launch_conv_slab(h, p, as_stream(stream), d.tile == TFMQ_TILE_SLAB)) {
      TFMQ_LAUNCH_CHECK(h);
      return TFMQ_OK;
    }
    // the 128-pixel form of the same kernel (two blocks per CU): a caller's measured choice, or the rule's fallback when the
    // 256-pixel grid would leave CUs idle (8x8 / 16x16 feature maps at small batches)
    if ((d.tile == TFMQ_TILE_AUTO || d.tile == TFMQ_TILE_SLAB128) && dma8 &&
        launch_conv_slab(h, p, as_stream(stream), d.tile == TFMQ_TILE_SLAB128, false, true)) {
      TFMQ_LAUNCH_CHECK(h);
      return TFMQ_OK;
    }
  } else {
    // the same kernel on fp16 operands (un-quantised / weight-only 3x3 layers, fp16 input).  Its K order is (channel chunk, tap),
    // the tile kernels' (tap, channel chunk): fp32 sums differ in the last bits, so the choice must not depend on the batch size --
    // every launch whose GEOMETRY the slab kernel takes runs on it (whatever the grid size), and the tile kernels only when a
    // caller pins one (tests, A/B runs).  A UNet-batch-12 forward then equals two batch-6 forwards bit for bit as before.
    // (TFMQ_TILE_SLAB128: the 128-pixel form -- the same K order and MFMA sequence per output, hence the same bits)
    if (d.tile == TFMQ_TILE_SLAB128 && dma16 && !d.x2 && launch_conv_slab(h, p, as_stream(stream), true, true, true)) {
      TFMQ_LAUNCH_CHECK(h);
      return TFMQ_OK;
    }
    if ((d.tile == TFMQ_TILE_AUTO || d.tile == TFMQ_TILE_SLAB || d.tile == TFMQ_TILE_SLAB128) && dma16 && !d.x2 &&
        launch_conv_slab(h, p, as_stream(stream), true, true)) {
      TFMQ_LAUNCH_CHECK(h);
      return TFMQ_OK;
    }
  }
  TFMQ_CHECK_ARG(h, (!d.res_f16 && (d.out_mode != TFMQ_OUT_F16 || (!d.rowadd && !d.residual && !d.stats))) ||
                        (((d.Cout | d.ldy | d.y_coff) & 3) == 0 && (!d.rowadd || (d.rowadd_ld & 3) == 0)),
                 "conv: fp16 residual / fp16 output with rowadd, residual or stats needs Cout, ldy, y_coff, rowadd_ld % 4 == 0");
  const int BM = big ? 256 : (small ? 64 : 128), BN = narrow ? 32 : ((small || half_n) ? 64 : 128);
  p.tiles_n = (d.Cout + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  dim3 grid(static_cast<unsigned>(p.tiles_n) * tiles_m);
  hipStream_t st = as_stream(stream);
  p.ksplit = 1;
  bool ks_record = false;
  p.ks_ws = h->ksplit_ws;
  p.ks_cnt = h->ksplit_cnt;
  if constexpr (INT8) {
    const bool dma = p.chunks == (d.Cin + 63) / 64 && d.KH * d.KW <= 9 &&
                     static_cast<size_t>(d.B) * d.H * d.W * d.Cin < (static_cast<size_t>(1) << 31);
    if (d.ksplit > 1) {
      const size_t tiles = static_cast<size_t>(p.tiles_n) * tiles_m;
      TFMQ_CHECK_ARG(h, dma && d.ksplit <= p.nsteps && tiles <= static_cast<size_t>(tfmq_ctx::KSPLIT_MAX_TILES) &&
                            tiles * d.ksplit * BM * BN <= tfmq_ctx::KSPLIT_WS_INTS,
                     "conv_w4a8: ksplit needs the LDS-DMA tile kernel, ksplit <= K-steps and tiles * ksplit * tile elements <= 16 Mi");
      p.ksplit = d.ksplit;
      grid.x *= static_cast<unsigned>(d.ksplit);
      // one workspace / ticket array per handle: order this launch behind the last split-K launch of another stream (common.hpp)
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(st, &cap);
      if (h->ksplit_owner_set && h->ksplit_owner != stream && h->ksplit_ev_valid && cap == hipStreamCaptureStatusNone)
        TFMQ_HIP(h, hipStreamWaitEvent(st, h->ksplit_ev, 0));
      h->ksplit_owner = stream;
      h->ksplit_owner_set = true;
      ks_record = cap == hipStreamCaptureStatusNone;
    }
    if (dma) {
#ifdef TFMQ_PHASE_TIMERS
      static unsigned long long* dbuf = nullptr;
      const size_t dwords = static_cast<size_t>(grid.x) * 4;
      if (!dbuf) (void)hipMalloc(reinterpret_cast<void**>(&dbuf), sizeof(unsigned long long) * 4 * (1u << 20));
      p.dbg = grid.x <= (1u << 20) ? dbuf : nullptr;
#endif
      // residual rows prefetched ahead of the staging (see conv_epilogue): the epilogue's vector path only
      const bool vec4 = ((d.Cout | d.ldy | d.y_coff) & 3) == 0 && (!d.rowadd || (d.rowadd_ld & 3) == 0);
      const bool wide16 = d.out_mode == TFMQ_OUT_Q8 && vec4 && (d.Cout & 15) == 0;               // the epilogue's 16-byte item paths
      const bool wide8 = d.out_mode == TFMQ_OUT_F16 && vec4 && ((d.Cout | d.ldy | d.y_coff) & 7) == 0;   // load their residual themselves
      const bool res_pre = d.residual && vec4 && !wide16 && !wide8 &&
                           (d.out_mode == TFMQ_OUT_F32 || d.out_mode == TFMQ_OUT_Q8 || d.out_mode == TFMQ_OUT_F16);
      // (a "whole K resident" variant -- NST = 5 stages for the five K-steps of a K = 320 token Linear, every step requested up
      // front -- was measured and lost: 80 KiB of LDS leaves two blocks per CU instead of three, 246 -> 282 us on the
      // 320 -> 320 residual Linear at UNet batch 128)
      if (narrow) hipLaunchKernelGGL((k_conv_dma<false, 4, 1, 1, 1>), grid, dim3(256), 0, st, p);
      else if (half_n && res_pre) hipLaunchKernelGGL((k_conv_dma<false, 2, 2, 2, 1, true>), grid, dim3(256), 0, st, p);
      else if (half_n) hipLaunchKernelGGL((k_conv_dma<false, 2, 2, 2, 1>), grid, dim3(256), 0, st, p);
      else if (small && res_pre) hipLaunchKernelGGL((k_conv_dma<false, 2, 2, 1, 1, true>), grid, dim3(256), 0, st, p);
      else if (small) hipLaunchKernelGGL((k_conv_dma<false, 2, 2, 1, 1>), grid, dim3(256), 0, st, p);
      else if (big) hipLaunchKernelGGL((k_conv_dma<false, 2, 2, 4, 2>), grid, dim3(256), 0, st, p);
      else if (res_pre) hipLaunchKernelGGL((k_conv_dma<false, 2, 2, 2, 2, true>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((k_conv_dma<false, 2, 2, 2, 2>), grid, dim3(256), 0, st, p);
#ifdef TFMQ_PHASE_TIMERS
      if (p.dbg && getenv("TFMQ_PHASE_PRINT")) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> hbuf(dwords);
        (void)hipMemcpy(hbuf.data(), dbuf, dwords * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double a = 0, b = 0, c = 0;
        unsigned long long lo = ~0ull, hi = 0;
        for (unsigned i = 0; i < grid.x; ++i) {
          a += double(hbuf[i * 4 + 1] - hbuf[i * 4]);
          b += double(hbuf[i * 4 + 2] - hbuf[i * 4 + 1]);
          c += double(hbuf[i * 4 + 3] - hbuf[i * 4 + 2]);
          lo = hbuf[i * 4] < lo ? hbuf[i * 4] : lo;
          hi = hbuf[i * 4 + 3] > hi ? hbuf[i * 4 + 3] : hi;
        }
        fprintf(stderr, "[conv_dma %dx%dx%d Cin%d Cout%d k%d mode%d] blocks %u nsteps %d: prologue %.0f  loop %.0f  epilogue %.0f  ticks/block; span %llu ticks\n",
                d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.out_mode, grid.x, p.nsteps, a / grid.x, b / grid.x, c / grid.x, hi - lo);
      }
#endif
    } else {
#define TFMQ_LAUNCH(CK, A, B_, C_, D_) hipLaunchKernelGGL((k_conv_igemm<true, false, CK, A, B_, C_, D_>), grid, dim3(256), 0, st, p)
      const bool k32 = d.Cin % 64 != 0;
      if (narrow) { if (k32) TFMQ_LAUNCH(32, 4, 1, 1, 1); else TFMQ_LAUNCH(64, 4, 1, 1, 1); }
      else if (small) { if (k32) TFMQ_LAUNCH(32, 2, 2, 1, 1); else TFMQ_LAUNCH(64, 2, 2, 1, 1); }
      else { if (k32) TFMQ_LAUNCH(32, 2, 2, 2, 2); else TFMQ_LAUNCH(64, 2, 2, 2, 2); }
#undef TFMQ_LAUNCH
    }
  } else if (d.x_f16) {
    // fp16 activations (written as fp16 by the producer): LDS-DMA pipeline, any stride / padding / upsample
    TFMQ_CHECK_ARG(h, d.Cin % 32 == 0 && d.KH * d.KW <= 9 &&
                          static_cast<size_t>(d.B) * d.H * d.W * d.Cin * 2 < (static_cast<size_t>(1) << 31),
                   "conv_f16: fp16 input needs Cin % 32 == 0, <= 9 taps and < 2 GiB of input");
    if (narrow) hipLaunchKernelGGL((k_conv_dma<true, 4, 1, 1, 1>), grid, dim3(256), 0, st, p);
    else if (half_n) hipLaunchKernelGGL((k_conv_dma<true, 2, 2, 2, 1>), grid, dim3(256), 0, st, p);
    else if (small) hipLaunchKernelGGL((k_conv_dma<true, 2, 2, 1, 1>), grid, dim3(256), 0, st, p);
    else if (big) hipLaunchKernelGGL((k_conv_dma<true, 2, 2, 4, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_conv_dma<true, 2, 2, 2, 2>), grid, dim3(256), 0, st, p);
  } else {
    // fast addressing: stride 1, no fused upsample, whole K-steps, <= 32 taps, 16-byte aligned rows
    const bool fast = d.stride == 1 && !d.up2x && d.KH * d.KW <= 32 && d.Cin % 32 == 0;
#define TFMQ_LAUNCH(F, A, B_, C_, D_) hipLaunchKernelGGL((k_conv_igemm<false, F, 64, A, B_, C_, D_>), grid, dim3(256), 0, st, p)
    if (narrow) { if (fast) TFMQ_LAUNCH(true, 4, 1, 1, 1); else TFMQ_LAUNCH(false, 4, 1, 1, 1); }
    else if (small) { if (fast) TFMQ_LAUNCH(true, 2, 2, 1, 1); else TFMQ_LAUNCH(false, 2, 2, 1, 1); }
    else { if (fast) TFMQ_LAUNCH(true, 2, 2, 2, 2); else TFMQ_LAUNCH(false, 2, 2, 2, 2); }
#undef TFMQ_LAUNCH
  }
  TFMQ_LAUNCH_CHECK(h);
  if (ks_record) {
    if (!h->ksplit_ev) TFMQ_HIP(h, hipEventCreateWithFlags(&h->ksplit_ev, hipEventDisableTiming));
    TFMQ_HIP(h, hipEventRecord(h->ksplit_ev, st));
    h->ksplit_ev_valid = true;
  }
  return TFMQ_OK;
}

extern "C" int tfmq_conv2d_w4a8(tfmq_handle h, const tfmq_conv_desc* d, void* stream) {
  return launch_conv<true>(h, d, stream);
}
extern "C" int tfmq_conv2d_f16(tfmq_handle h, const tfmq_conv_desc* d, void* stream) {
  return launch_conv<false>(h, d, stream);
}
