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

namespace {

enum { LIN_F16 = 0, LIN_Q8 = 1, LIN_GEGLU = 2 };

template <int MODE>
__global__ __launch_bounds__(256, 3) void k_lin_direct(ConvP p) {
  constexpr int BM = 128, BN = 128;
  constexpr int STAGE = (BM + BN) * 64;
  constexpr int NST = 3;
  constexpr int NLOAD = 4;                       // DMA pieces per wave per K-step: 2 of A, 2 of B
  constexpr int CONST_OFF = NST * STAGE;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE + 3 * BN * 4];

  const tfmq_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int bid = xcd_tile_id();
  const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- DMA sources (pointwise: pixel m reads input pixel m; rows past M read the zero row of the pad table)
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));
  const unsigned char* xb = static_cast<const unsigned char*>(d.x);
  const int dcol = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
  const unsigned char* a_ptr[2];
  const unsigned char* b_ptr[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int piece = wid * 2 + it;
    const int m = m0 + piece * 16 + (lane >> 2);
    a_ptr[it] = m < p.M ? xb + static_cast<size_t>(m) * d.Cin + dcol : p.pad_table + dcol;
    int n = n0 + piece * 16 + (lane >> 2);
    n = n < p.cout_pad ? n : p.cout_pad - 1;
    b_ptr[it] = static_cast<const unsigned char*>(d.w) + (static_cast<size_t>(n / 32) * p.nsteps * 32 + (n % 32)) * 64 + dcol;
  }
  const bool a_live0 = m0 + (wid * 2) * 16 + (lane >> 2) < p.M, a_live1 = m0 + (wid * 2 + 1) * 16 + (lane >> 2) < p.M;
  auto issue = [&](int s, int stage) {
    const unsigned sbase = lds0 + stage * STAGE;
    glds16(a_ptr[0] + (a_live0 ? s * 64 : 0), sbase + __builtin_amdgcn_readfirstlane((wid * 2) * 1024));
    glds16(a_ptr[1] + (a_live1 ? s * 64 : 0), sbase + __builtin_amdgcn_readfirstlane((wid * 2 + 1) * 1024));
    glds16(b_ptr[0] + static_cast<size_t>(s) * 2048, sbase + __builtin_amdgcn_readfirstlane(BM * 64 + (wid * 2) * 1024));
    glds16(b_ptr[1] + static_cast<size_t>(s) * 2048, sbase + __builtin_amdgcn_readfirstlane(BM * 64 + (wid * 2 + 1) * 1024));
  };

  // fragment rows: pixels (wm * 2 + i) * 32 + lane % 32; channels of N-tile j: plain (wn * 2 + j) * 32, GEGLU j * 64 + wn * 32
  // (tile columns [0, 64) = value, [64, 128) = gate of the same 64 output channels)
  auto ncol0 = [&](int j) { return MODE == LIN_GEGLU ? j * 64 + wn * 32 : (wn * 2 + j) * 32; };
  const int fsw = (h ^ ((lane >> 2) & 3)) << 4;           // physical 16-byte slot of k-slot h in this lane's row

  v16i acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  issue(0, 0);
  if (p.nsteps > 1) issue(1, 1);

  // ---- requested now, consumed after the K loop: per-column constants (threads < BN), the quantizer parameters, and the
  // residual values of this lane's outputs (branch-free: clamped addresses).  They are younger than the first DMA
  // pieces, so the loop's counted waits stay correct (only its first step waits for more than it needs).
  float c_ws = 1.0f, c_bias = 0.0f;
  int c_zp = 0, c_rs = 0;
  if (tid < BN) {
    const int n = n0 + tid;
    if (n < d.Cout) {
      const int4 wmv = reinterpret_cast<const int4*>(d.wmeta)[n];
      c_zp = wmv.x;
      c_rs = wmv.y;
      c_ws = d.wscale[n];
      c_bias = d.bias ? d.bias[n] : 0.0f;
    }
  }
  const float2 aqp = load_qparam(d.aq);
  float2 oqp = make_float2(1.0f, 0.0f);
  if constexpr (MODE != LIN_F16) oqp = load_qparam(d.oq);
  uint2 rres[2][2][4];                     // MODE F16 / Q8 with a residual: 4 channels x fp16 per (i, j, quad)
  float4 rres32[MODE == LIN_GEGLU ? 1 : 1];
  (void)rres32;
  const bool has_res = MODE != LIN_GEGLU && d.residual != nullptr;
  if (has_res && d.res_f16) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + (wm * 2 + i) * 32 + (lane & 31);
      const int mc = m < p.M ? m : p.M - 1;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + ncol0(j) + 8 * q + 4 * h;
          const int nc = n < d.Cout ? n : 0;
          rres[i][j][q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(d.residual) + static_cast<size_t>(mc) * d.Cout + nc);
        }
    }
  }

  int st_c = 0, st_i = 2;
  for (int s = 0; s < p.nsteps; ++s) {
    if (s + 1 < p.nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (s + 2 < p.nsteps) issue(s + 2, st_i);
    const unsigned char* sa = lds + st_c * STAGE;
    const unsigned char* sb = sa + BM * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v4i af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const v4i*>(sa + ((wm * 2 + i) * 32 + (lane & 31)) * 64 + (fsw ^ (ks << 5)));
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const v4i*>(sb + (ncol0(j) + (lane & 31)) * 64 + (fsw ^ (ks << 5)));
      // operands swapped: the accumulator tile is (channels x pixels) -- lane = pixel, register quad = 4 consecutive channels
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    st_c = st_c == NST - 1 ? 0 : st_c + 1;
    st_i = st_i == NST - 1 ? 0 : st_i + 1;
  }

  // ---- per-column constants -> LDS table {scale, zero-point correction (as float bits of an int), bias}
  float* cs = reinterpret_cast<float*>(lds + CONST_OFF);
  const int za = static_cast<int>(aqp.y);
  if (tid < BN) {
    cs[tid] = aqp.x * c_ws;
    reinterpret_cast<int*>(cs)[BN + tid] = (128 - za) * (c_rs - p.Ktot * c_zp);
    cs[2 * BN + tid] = c_bias;
  }
  __syncthreads();

  // ---- epilogue out of the registers.  acc[i][j][4q + c] = channel ncol0(j) + 8q + 4h + c of pixel (wm*2+i)*32 + lane%32.
  // Packed fp32 arithmetic (two outputs per VALU instruction: the GEGLU epilogue -- 2 affine maps, the erf polynomial, exp,
  // rcp and the quantizer per output -- is what bounds these layers, ~45 scalar-lane instructions per output before).
  auto affine2 = [&](int a0, int a1, float sx, float sy, int kx, int ky, float bx, float by) -> f2 {
    return f2{sx, sy} * f2{static_cast<float>(a0 + kx), static_cast<float>(a1 + ky)} + f2{bx, by};
  };
  auto epi = [&](auto exact_div) {
    constexpr bool EX = decltype(exact_div)::value;
    const QuantP qP = make_quantp(oqp);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + (wm * 2 + i) * 32 + (lane & 31);
      const bool mok = m < p.M;
      if constexpr (MODE == LIN_GEGLU) {
        const int inner = d.Cout >> 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cv = ncol0(0) + 8 * q + 4 * h, cg = ncol0(1) + 8 * q + 4 * h;     // value / gate columns inside the tile
          const float4 sv = *reinterpret_cast<const float4*>(cs + cv), sg = *reinterpret_cast<const float4*>(cs + cg);
          const int4 kv = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(cs) + BN + cv);
          const int4 kg = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(cs) + BN + cg);
          const float4 bv = *reinterpret_cast<const float4*>(cs + 2 * BN + cv), bg = *reinterpret_cast<const float4*>(cs + 2 * BN + cg);
          const f2 a01 = affine2(acc[i][0][4 * q + 0], acc[i][0][4 * q + 1], sv.x, sv.y, kv.x, kv.y, bv.x, bv.y);
          const f2 a23 = affine2(acc[i][0][4 * q + 2], acc[i][0][4 * q + 3], sv.z, sv.w, kv.z, kv.w, bv.z, bv.w);
          const f2 g01 = affine2(acc[i][1][4 * q + 0], acc[i][1][4 * q + 1], sg.x, sg.y, kg.x, kg.y, bg.x, bg.y);
          const f2 g23 = affine2(acc[i][1][4 * q + 2], acc[i][1][4 * q + 3], sg.z, sg.w, kg.z, kg.w, bg.z, bg.w);
          const unsigned w = quant_pack4_t<EX>(a01 * gelu2(g01), a23 * gelu2(g23), qP);
          const int oc = (n0 >> 1) + wn * 32 + 8 * q + 4 * h;                           // output channel (of Cout / 2)
          if (mok && oc < inner) *reinterpret_cast<unsigned*>(d.yq + static_cast<size_t>(m) * inner + oc) = w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ct = ncol0(j) + 8 * q + 4 * h, n = n0 + ct;
            const float4 sc = *reinterpret_cast<const float4*>(cs + ct);
            const int4 kc = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(cs) + BN + ct);
            const float4 bb = *reinterpret_cast<const float4*>(cs + 2 * BN + ct);
            f2 v01 = affine2(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], sc.x, sc.y, kc.x, kc.y, bb.x, bb.y);
            f2 v23 = affine2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3], sc.z, sc.w, kc.z, kc.w, bb.z, bb.w);
            if (has_res) {
              if (d.res_f16) {
                const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&rres[i][j][q].x));
                const float2 hi = __half22float2(*reinterpret_cast<const __half2*>(&rres[i][j][q].y));
                v01 += f2{lo.x, lo.y};
                v23 += f2{hi.x, hi.y};
              } else if (mok && n < d.Cout) {
                const float4 a = *reinterpret_cast<const float4*>(d.residual + static_cast<size_t>(m) * d.Cout + n);
                v01 += f2{a.x, a.y};
                v23 += f2{a.z, a.w};
              }
            }
            if (!mok || n >= d.Cout) continue;
            if constexpr (MODE == LIN_F16) {
              *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(d.y) + static_cast<size_t>(m) * d.ldy + d.y_coff + n) =
                  make_uint2(pack_h2(v01.x, v01.y), pack_h2(v23.x, v23.y));
            } else {
              *reinterpret_cast<unsigned*>(d.yq + static_cast<size_t>(m) * d.Cout + n) = quant_pack4_t<EX>(v01, v23, qP);
            }
            __builtin_amdgcn_sched_barrier(0);      // one quad at a time: interleaving them all spilled the accumulators
          }
      }
    }
  };
  if constexpr (MODE == LIN_F16) {
    epi(std::false_type{});
  } else {
    if (__builtin_expect((__float_as_uint(oqp.x) & 0x7fffffu) == 0x7fffffu, 0)) epi(std::true_type{});
    else epi(std::false_type{});
  }
}

}  // namespace

bool launch_conv_lin(tfmq_handle h, ConvP& p, hipStream_t st) {
  const tfmq_conv_desc& d = p.d;
  if (d.KH != 1 || d.KW != 1 || d.stride != 1 || d.up2x || d.pad_t != 0 || d.pad_l != 0 || d.Ho != d.H || d.Wo != d.W) return false;
  if (d.Cin % 64 != 0 || static_cast<size_t>(d.B) * d.H * d.W * d.Cin >= (static_cast<size_t>(1) << 31)) return false;
  if (d.rowadd || d.stats || d.yt || (d.Cout & 3) != 0) return false;
  int mode;
  if (d.out_mode == TFMQ_OUT_F16) {
    if (((d.ldy | d.y_coff) & 3) != 0) return false;
    mode = LIN_F16;
  } else if (d.out_mode == TFMQ_OUT_Q8) {
    mode = LIN_Q8;
  } else if (d.out_mode == TFMQ_OUT_GEGLU_Q8) {
    if (d.residual || d.Cout % 128 != 0) return false;
    mode = LIN_GEGLU;
  } else {
    return false;
  }
  (void)h;
  p.tiles_n = (d.Cout + 127) / 128;
  const int tiles_m = (p.M + 127) / 128;
  dim3 grid(static_cast<unsigned>(p.tiles_n) * tiles_m);
  if (mode == LIN_F16) hipLaunchKernelGGL((k_lin_direct<LIN_F16>), grid, dim3(256), 0, st, p);
  else if (mode == LIN_Q8) hipLaunchKernelGGL((k_lin_direct<LIN_Q8>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_lin_direct<LIN_GEGLU>), grid, dim3(256), 0, st, p);
  return true;
}
