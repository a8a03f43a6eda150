// K5f (round 4): the feed-forward half of a BasicTransformerBlock as ONE launch, a token per lane.
//
//   x -> LayerNorm -> quantise -> ff.net.0.proj (C -> 2 * inner, w4a8) -> value * gelu(gate) -> quantise
//     -> ff.net.2 (inner -> C, w4a8) -> + x -> fp16 stream (or the consumer quantizer's int8 bins)
// (ldm/modules/attention.py:37-64 FeedForward / GEGLU, :152-215 BasicTransformerBlock._forward `x = self.ff(self.norm3(x)) + x`;
//  QuantBasicTransformerBlock, quant/quant_block.py:248-299; the QuantLayers' quantizers, quant/quant_layer.py:306-340).
//
// As three launches (k_layernorm_hs, k_lin_direct<GEGLU>, k_lin_direct<F16 | Q8>) the chain moved 5120 B per token through HBM -- the
// int8 LayerNorm output, the inner-wide int8 GEGLU tensor written and read back, the residual -- and spent 1.48 ms at the 64 x 64 level of
// SD (UNet batch 128): the K = 320 projection is five K-steps long, a block of the pointwise kernel lives mostly in its prologue and its
// VALU-bound epilogue.  Here a workgroup owns 256 tokens for the whole chain and HBM sees the fp16 row once in and once out (1280 B):
//   * a wave owns 32 tokens, LANE = TOKEN.  With the MFMA operands swapped (acc = W_tile . X^T, conv_lin.hip) a lane holds, per 32-channel
//     output tile, the 16 consecutive channels 16h .. 16h + 15 of its token (h = lane / 32) -- and that is exactly the 16-byte K-slice the
//     same lane must supply when the tile is the NEXT GEMM's input k-substep.  So the GEGLU bins never leave the registers: value / gate
//     accumulators -> affine map, gelu, quantise (the TFMQ_OUT_GEGLU_Q8_FAST arithmetic, bit for bit) -> packed int8 -> operand of the
//     ff.net.2 MFMAs.  No LDS round trip, no barrier between the two GEMMs.
//   * the LayerNorm runs on the lane's own row (in-lane sums, ONE v_permlane32_swap per statistic with the lane that holds the row's other
//     half), in k_layernorm_hs's summation order: bins bit-identical to the stand-alone kernel.  Its int8 output is the wave's private
//     10 KB of LDS in the K-step-major swizzled layout the fragment reads of conv_lin.hip use.
//   * only the weights stream: per pair of hidden tiles (64 channels) three phases -- value | gate tiles of the first 32 channels (20 KB),
//     of the second (20 KB), the K-step of ff.net.2 for all C outputs (20 KB) -- through a 3-slot LDS-DMA ring shared by the 8 waves, two
//     phases ahead, one counted s_waitcnt + s_barrier per phase.  1.23 MB of (L2-resident) weights per 256 tokens.
//   * ff.net.2's C outputs of a token stay in the accumulators (C / 32 tiles x 16 registers) across all inner / 64 pairs.
// Epilogue: scale * float(acc + kc) + bias + x (fp16 re-read, L2-warm) -> fp16 rows through a wave-private LDS transpose (whole 128-byte
// row segments), or the consumer's int8 bins (proj_out's quantizer when the tokens feed nothing else).  Same operations as
// k_lin_direct's epilogue: the three-launch chain and this kernel agree bit for bit (tests/test_ff_fused_gpu.py).
#include "conv_common.hpp"
#include <type_traits>

// Timing-only ablations (scratch/r04_ff_abl.sh builds one library per mask with -DFF_ABLATE=<mask>; results are garbage, times are not):
// 1 no GELU / quantise arithmetic, 2 no projection MFMAs, 4 no ff.net.2 MFMAs, 8 no constant reads, 16 no LayerNorm arithmetic, 32 no fragment reads
#ifndef FF_ABLATE
#define FF_ABLATE 0
#endif

namespace {

struct FfP {
  tfmq_ff_desc d;
  const unsigned char* pad_table;
  int prio;            // TFMQ_SETPRIO=1 (A/B runs): waves 4-7 at s_setprio 1
};

template <int N>
__device__ __forceinline__ void ff_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- folded per-channel constants of the GEGLU projection: c1[T][4][32] = {value scale / delta_o, value bias', gate scale, gate bias'}
// of hidden channels 32 T .. 32 T + 31 (k_lin_direct<LIN_GEGLU_FAST>'s table, operation for operation).  One tiny launch per call: the
// activation deltas belong to the current Finite-Set row (device step counter).
__global__ __launch_bounds__(256) void k_ff_fold(tfmq_ff_desc d) {
  const int hc = blockIdx.x * 256 + threadIdx.x;
  if (hc >= d.inner) return;
  const float2 aqp = load_qparam(d.aq0), oqp = load_qparam(d.aq2);
  const int za = static_cast<int>(aqp.y);
  const int T = hc >> 5, j = hc & 31;
  float* out = d.ws + static_cast<size_t>(T) * 128 + j;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int pr = (hc >> 6) * 128 + (hc & 63) + 64 * g;          // packed row (ops.geglu_perm): value rows, then the gate rows
    const int4 wmv = reinterpret_cast<const int4*>(d.wmeta1)[pr];
    const float sc = aqp.x * d.wscale1[pr];
    const int kc = (128 - za) * (wmv.y - d.C * wmv.x);
    const float bf = sc * static_cast<float>(kc) + (d.bias1 ? d.bias1[pr] : 0.0f);
    out[(2 * g) * 32] = g == 0 ? sc / oqp.x : sc;
    out[(2 * g + 1) * 32] = g == 0 ? bf / oqp.x : bf;
  }
}

// the optional Linears in front of / behind the feed-forward (K = C): per phase (64 columns) {scale[64], kc[64] (int bits), bias[64], pad[64]}
// (k_lin_direct's table) behind the GEGLU constants: ws + 4 * inner + (5 * which + phase) * 256
__global__ __launch_bounds__(256) void k_ff_fold_lin(tfmq_ff_desc d) {
  const int which = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (n >= d.C) return;
  const int32_t* wmeta = which ? d.wmeta3 : d.wmeta0;
  const float* wscale = which ? d.wscale3 : d.wscale0;
  const float* bias = which ? d.bias3 : d.bias0;
  if (!wmeta) return;
  const float2 aqp = load_qparam(which ? d.oq : d.aq_pre);
  const int4 wmv = reinterpret_cast<const int4*>(wmeta)[n];
  float* out = d.ws + 4 * static_cast<size_t>(d.inner) + static_cast<size_t>(5 * which + (n >> 6)) * 256 + (n & 63);
  out[0] = aqp.x * wscale[n];
  reinterpret_cast<int*>(out)[64] = (128 - static_cast<int>(aqp.y)) * (wmv.y - d.C * wmv.x);
  out[128] = bias ? bias[n] : 0.0f;
}

constexpr int FF_STG_ROW = 144;         // output staging: 64 fp16 + 16 bytes per token row (conv_lin.hip)
constexpr int FF_STG_ROW_Q8 = 80;

template <int C>
struct FfGeo {
  static constexpr int NCH = C / 64;                 // 64-channel chunks (K-steps) of a token row
  static constexpr int NT = C / 32;                  // 32-channel tiles
  static constexpr int XW = NCH * 2048;              // bytes of a wave's quantised rows
  static constexpr int SLOT = 2 * NCH * 2048 + 1024; // value | gate tiles of one hidden tile (or ff.net.2's K-step: NT * 2 KB) + constants
  static constexpr int RING_OFF = 0;                 // (first: every ring address is a 16-bit immediate away from a fragment base register)
  static constexpr int X_OFF = 3 * SLOT;
  static constexpr int CS2_OFF = X_OFF + 8 * XW;
  static constexpr int LNGB_OFF = CS2_OFF + 3 * C * 4;
  static constexpr int TOTAL = (LNGB_OFF + 2 * C * 4) > (CS2_OFF + 16384) ? (LNGB_OFF + 2 * C * 4) : (CS2_OFF + 16384);   // (POST: statistics partials alias the tables)
  static constexpr int NPIECE = 2 * NCH * 2;         // 1-KiB DMA pieces of a phase (value + gate tiles; = NT * 2 for ff.net.2's K-step)
  static_assert(NT * 2 == NPIECE, "");
  static constexpr int PPW = (NPIECE + 1 + 7) / 8;   // pieces a wave issues per phase (uniform: surplus slots re-load a piece)
};

// PRE: a C -> C Linear (+ bias, + fp16 residual) on int8 input rows in front (attn2.to_out: its output row is stored, normalised in place and
// becomes the residual of the feed-forward).  POST: a C -> C Linear on the feed-forward's output bins behind it (proj_out: + bias, + fp16
// residual, fp16 rows out, GroupNorm statistics of the consumer).  Both on the chain phases of row_chain.hip (two output tiles per phase,
// slots 0 / 1 of the ring).  M % 256 == 0 with either.
template <int C, bool PRE, bool POST>
__global__ __launch_bounds__(512, 2) void k_ff_fused(FfP p) {
  using G = FfGeo<C>;
  constexpr int NCH = G::NCH, NT = G::NT, PPW = G::PPW;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G::TOTAL];
  const tfmq_ff_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, pl = lane & 31;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 256;
  if (p.prio && wid >= 4) __builtin_amdgcn_s_setprio(1);
  const int m = m0 + wid * 32 + pl;
  const int mc = m < d.M ? m : d.M - 1;
  const bool mok = m < d.M;
  const int npairs = d.inner >> 6, nsteps2 = npairs;

  // ---- weight stream: phase ph = 3 q + kind; kind 0 / 1: value | gate tiles of hidden tile 2 q + kind (+ the pair's constants), kind 2:
  // ff.net.2's K-step q for all C / 32 output tiles.  Piece pi of a phase = 16 weight rows x 64 B, lane-linear in LDS, swizzled on the source.
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));
  const unsigned voff = static_cast<unsigned>((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
  const unsigned char* w1 = reinterpret_cast<const unsigned char*>(d.w1);
  const unsigned char* w2 = reinterpret_cast<const unsigned char*>(d.w2);
  auto issue = [&](int kind, int q) {
    const unsigned sbase = lds0 + G::RING_OFF + kind * G::SLOT;
#pragma unroll
    for (int it = 0; it < PPW; ++it) {
      int pi = wid + 8 * it;
      if (pi == G::NPIECE && kind != 2) {           // the pair's folded constants: 1 KiB, lane-linear (no swizzle)
        glds16_sv(reinterpret_cast<const unsigned char*>(d.ws) + static_cast<size_t>(q) * 1024, static_cast<unsigned>(lane * 16),
                  sbase + __builtin_amdgcn_readfirstlane(G::NPIECE * 1024));
        continue;
      }
      if (pi >= G::NPIECE) pi -= 8;                  // surplus slot: the wave's previous piece again (same bytes to the same place)
      const int j = pi & 1;
      const unsigned char* src;
      if (kind != 2) {
        const int gate = pi >= 2 * NCH, s = (pi - gate * 2 * NCH) >> 1;
        const int ntile = 4 * q + 2 * gate + kind;
        src = w1 + ((static_cast<size_t>(ntile) * NCH + s) * 32 + j * 16) * 64;
      } else {
        const int nt = pi >> 1;
        src = w2 + ((static_cast<size_t>(nt) * nsteps2 + q) * 32 + j * 16) * 64;
      }
      glds16_sv(src, voff, sbase + __builtin_amdgcn_readfirstlane(pi * 1024));
    }
  };
  // ---- chain phases of the optional Linears (row_chain.hip): phase ph of `which` (0 = PRE, 1 = POST) = output tiles 2 ph, 2 ph + 1
  auto issue_lin = [&](int which, int ph) {
    const unsigned char* w = reinterpret_cast<const unsigned char*>(which ? d.w3 : d.w0);
    const unsigned sbase = lds0 + G::RING_OFF + (ph & 1) * G::SLOT;
#pragma unroll
    for (int it = 0; it < PPW; ++it) {
      int pi = wid + 8 * it;
      if (pi == G::NPIECE) {
        glds16_sv(reinterpret_cast<const unsigned char*>(d.ws + 4 * static_cast<size_t>(d.inner)) + static_cast<size_t>(5 * which + ph) * 1024,
                  static_cast<unsigned>(lane * 16), sbase + __builtin_amdgcn_readfirstlane(G::NPIECE * 1024));
        continue;
      }
      if (pi >= G::NPIECE) pi -= 8;
      const int tile = pi / (2 * NCH), rem = pi - tile * 2 * NCH, sI = rem >> 1, j = rem & 1;
      const unsigned char* src = w + ((static_cast<size_t>(2 * ph + tile) * NCH + sI) * 32 + j * 16) * 64;
      glds16_sv(src, voff, sbase + __builtin_amdgcn_readfirstlane(pi * 1024));
    }
  };
  unsigned char* Xw = lds + G::X_OFF + wid * G::XW;
  const int fsw = (h ^ ((pl >> 2) & 3)) << 4;
  const int brow = lin_brow(pl);
  const int bsw = (h ^ ((brow >> 2) & 3)) << 4;
  const unsigned char* xfr = Xw + pl * 64;
  const int swz_x = (pl >> 2) & 3;
  auto x_store = [&](int t, unsigned w0, unsigned w1_, unsigned w2_, unsigned w3_) {
    *reinterpret_cast<uint4*>(Xw + (t >> 1) * 2048 + pl * 64 + (((2 * (t & 1) + h) ^ swz_x) << 4)) = make_uint4(w0, w1_, w2_, w3_);
  };
  // one phase: scale * float(acc + kc) + bias + residual -> fp16 (k_lin_direct's operations); rows out through the wave's staging in slot 2;
  // KEEP: the packed row stays in `keep` (PRE); STATS: the consumer GroupNorm's {sum, sum of squares} partials of the fp32 values (POST)
  auto lin_phase = [&](auto ph_tag, auto which_tag, uint4 (&keep)[NT][2], const __half* res, __half* yout, bool first) {
    constexpr int PH = decltype(ph_tag)::value, WHICH = decltype(which_tag)::value;
    if (first) ff_wait_vmcnt<0>();
    else ff_wait_vmcnt<4>();
    asm volatile("s_barrier" ::: "memory");
    if (PH + 1 < NT / 2) issue_lin(WHICH, PH + 1);
    const unsigned char* slot = lds + G::RING_OFF + (PH & 1) * G::SLOT;
    const float* cs = reinterpret_cast<const float*>(slot + G::NPIECE * 1024);
    unsigned char* stgl = lds + G::RING_OFF + 2 * G::SLOT + wid * (32 * 80);
    float2* part = reinterpret_cast<float2*>(lds + G::CS2_OFF);
    uint4 rr[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) rr[j][u] = *reinterpret_cast<const uint4*>(res + static_cast<size_t>(m) * C + 64 * PH + 32 * j + 16 * h + 8 * u);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      v16i acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
      for (int sidx = 0; sidx < NCH; ++sidx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const v4i xf = *reinterpret_cast<const v4i*>(xfr + sidx * 2048 + (fsw ^ (ks << 5)));
          const v4i wf = *reinterpret_cast<const v4i*>(slot + (j * NCH + sidx) * 2048 + brow * 64 + (bsw ^ (ks << 5)));
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, acc, 0, 0, 0);
        }
      unsigned hw[8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ct = 32 * j + 16 * h + 8 * u;
        f2 vv[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float4 sc = *reinterpret_cast<const float4*>(cs + ct + 4 * e);
          const int4 kc = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(cs) + 64 + ct + 4 * e);
          const float4 bb = *reinterpret_cast<const float4*>(cs + 128 + ct + 4 * e);
          vv[2 * e] = f2{sc.x, sc.y} * f2{static_cast<float>(acc[8 * u + 4 * e] + kc.x), static_cast<float>(acc[8 * u + 4 * e + 1] + kc.y)} + f2{bb.x, bb.y};
          vv[2 * e + 1] = f2{sc.z, sc.w} * f2{static_cast<float>(acc[8 * u + 4 * e + 2] + kc.z), static_cast<float>(acc[8 * u + 4 * e + 3] + kc.w)} + f2{bb.z, bb.w};
        }
        const uint4 rw = rr[j][u];
        const float2 r0 = __half22float2(*reinterpret_cast<const __half2*>(&rw.x)), r1 = __half22float2(*reinterpret_cast<const __half2*>(&rw.y));
        const float2 r2 = __half22float2(*reinterpret_cast<const __half2*>(&rw.z)), r3 = __half22float2(*reinterpret_cast<const __half2*>(&rw.w));
        vv[0] += f2{r0.x, r0.y};
        vv[1] += f2{r1.x, r1.y};
        vv[2] += f2{r2.x, r2.y};
        vv[3] += f2{r3.x, r3.y};
        if constexpr (WHICH == 1) {
          if (d.stats) {       // k_lin_direct's statistics: DPP sums over the 8 lanes (token rows) of a group, canonical order (group8_sum)
            const int grp = wid * 4 + (pl >> 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x0 = vv[e].x, x1 = vv[e].y;
              const float s0 = group8_sum(x0), s1 = group8_sum(x1);
              const float q0 = group8_sum(x0 * x0), q1 = group8_sum(x1 * x1);
              if ((lane & 7) == 0) *reinterpret_cast<float4*>(part + grp * 64 + ct + 2 * e) = make_float4(s0, q0, s1, q1);
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) hw[4 * u + e] = pack_h2(vv[e].x, vv[e].y);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (WHICH == 0) {
        keep[2 * PH + j][0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        keep[2 * PH + j][1] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
      }
      *reinterpret_cast<uint4*>(stgl + pl * 80 + 32 * h) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      *reinterpret_cast<uint4*>(stgl + pl * 80 + 32 * h + 16) = make_uint4(hw[4], hw[5], hw[6], hw[7]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 16 + (lane >> 2), pc = lane & 3;
        const uint4 w = *reinterpret_cast<const uint4*>(stgl + row * 80 + pc * 16);
        *reinterpret_cast<uint4*>(yout + static_cast<size_t>(m0 + wid * 32 + row) * C + 64 * PH + 32 * j + pc * 8) = w;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if constexpr (WHICH == 1) {
      if (d.stats) {
        LDS_BARRIER();
        const int seg = d.stats_seg, nseg = 256 / seg, gps = seg / 8;
        for (int o = tid; o < nseg * 64; o += 512) {
          const int sidx = o >> 6, col = o & 63;
          float2 a = make_float2(0.0f, 0.0f);
          for (int q = 0; q < gps; ++q) {            // a segment = its 8-row groups added in row order
            const float2 b = part[(sidx * gps + q) * 64 + col];
            a.x += b.x;
            a.y += b.y;
          }
          reinterpret_cast<float2*>(d.stats)[static_cast<size_t>((m0 + sidx * seg) / seg) * C + 64 * PH + col] = a;
        }
      }
    }
  };
  auto lin_all = [&](auto which_tag, uint4 (&keep)[NT][2], const __half* res, __half* yout) {
    lin_phase(std::integral_constant<int, 0>{}, which_tag, keep, res, yout, true);
    lin_phase(std::integral_constant<int, 1>{}, which_tag, keep, res, yout, false);
    lin_phase(std::integral_constant<int, 2>{}, which_tag, keep, res, yout, false);
    lin_phase(std::integral_constant<int, 3>{}, which_tag, keep, res, yout, false);
    lin_phase(std::integral_constant<int, 4>{}, which_tag, keep, res, yout, false);
  };
  static_assert(NT == 10, "the chain phases are written for C = 320");

  uint4 raw[NT][2];
  if constexpr (PRE) {
    // int8 input rows (the cross attention's output bins): 10 pieces per wave straight into the wave's X region, then the Linear
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(d.xq_pre) + static_cast<size_t>(m0 + wid * 32) * C;
#pragma unroll
    for (int sI = 0; sI < NCH; ++sI)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        glds16_sv(xb + static_cast<size_t>(j * 16) * C + sI * 64, static_cast<unsigned>((lane >> 2) * C + (((lane & 3) ^ ((lane >> 4) & 3)) * 16)),
                  lds0 + G::X_OFF + __builtin_amdgcn_readfirstlane(wid * G::XW + sI * 2048 + j * 1024));
    issue_lin(0, 0);
    lin_all(std::integral_constant<int, 0>{}, raw, reinterpret_cast<const __half*>(d.res_pre), reinterpret_cast<__half*>(d.y_pre));
    asm volatile("s_barrier" ::: "memory");          // slots 0 / 1 and the staging in slot 2 are free
  }
  issue(0, 0);
  issue(1, 0);
  issue(2, 0);

  // ---- per-column constants of ff.net.2 {scale, zero-point correction, bias} and the LayerNorm's gamma | beta -> LDS
  const float2 aqp0 = load_qparam(d.aq0), aqp2 = load_qparam(d.aq2);
  float* cs2 = reinterpret_cast<float*>(lds + G::CS2_OFF);
  float* lngb = reinterpret_cast<float*>(lds + G::LNGB_OFF);
  if (tid < C) {
    const int4 wmv = reinterpret_cast<const int4*>(d.wmeta2)[tid];
    cs2[tid] = aqp2.x * d.wscale2[tid];
    reinterpret_cast<int*>(cs2)[C + tid] = (128 - static_cast<int>(aqp2.y)) * (wmv.y - d.inner * wmv.x);
    cs2[2 * C + tid] = d.bias2 ? d.bias2[tid] : 0.0f;
    lngb[tid] = d.gamma[tid];
    lngb[C + tid] = d.beta[tid];
  }

  // ---- this lane's half of its token's row: channels 32 t + 16 h .. + 15 of every tile t
  const __half* xrow = reinterpret_cast<const __half*>(PRE ? d.y_pre : d.x) + static_cast<size_t>(mc) * C + 16 * h;
  {
    if constexpr (!PRE) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        raw[t][0] = *reinterpret_cast<const uint4*>(xrow + 32 * t);
        raw[t][1] = *reinterpret_cast<const uint4*>(xrow + 32 * t + 8);
      }
    }
    LDS_BARRIER();                    // the tables above
    // LayerNorm in k_layernorm_hs<C / 40>'s order: a row's 8-channel pieces idx = 4 t + 2 h + e belong to sub-lane j = idx % LPR and are
    // added in the order k = idx / LPR; the LPR partial sums go through the DPP tree ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)) ...
    float v[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const __half2* hp = reinterpret_cast<const __half2*>(&raw[t][e]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float2 f = __half22float2(hp[i]);
          v[t][8 * e + 2 * i] = f.x;
          v[t][8 * e + 2 * i + 1] = f.y;
        }
      }
    constexpr int LPR = C / 40;       // 8 (C = 320), 16 (C = 640)
    static_assert(LPR == 8 || LPR == 16, "token widths 320 / 640");
    // this lane's pieces by sub-lane: piece (t, e) -> j = (4 t + 2 h + e) % LPR.  With LPR = 8: t even -> j = 2 h + e, t odd -> 4 + 2 h + e;
    // LPR = 16: t % 4 = r -> j = 4 r + 2 h + e.  NSL = LPR / 4 (sub-lane, e) accumulators per e.
    constexpr int NSL = LPR / 4;
    auto row_total = [&](float (&s)[NSL][2]) -> float {
      // s[r][e] = partial sum of sub-lane 4 r + 2 h + e.  Tree of ln_group_sum<LPR>: pairs (j, j ^ 1), then (.., .. ^ 2), then the
      // half mirror (j, 7 - j) inside groups of 8, then (LPR = 16) the mirror (j, 15 - j).
      float tot8[NSL];
#pragma unroll
      for (int r = 0; r < NSL; ++r) {
        const float a = s[r][0] + s[r][1];                        // (s_{4r+2h} + s_{4r+2h+1})
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
        tot8[r] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);   // (s_{4r} + s_{4r+1}) + (s_{4r+2} + s_{4r+3})
      }
      if constexpr (NSL == 2) return tot8[0] + tot8[1];
      else return (tot8[0] + tot8[1]) + (tot8[2] + tot8[3]);
    };
    float s[NSL][2];
#pragma unroll
    for (int r = 0; r < NSL; ++r) s[r][0] = s[r][1] = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float* q = &v[t][8 * e];
        s[t % NSL][e] += ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
      }
    const float mean = row_total(s) / static_cast<float>(C);
#pragma unroll
    for (int r = 0; r < NSL; ++r) s[r][0] = s[r][1] = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float tt = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float a = v[t][8 * e + i] - mean;
          tt = __builtin_fmaf(a, a, tt);
        }
        s[t % NSL][e] += tt;
      }
    const float rstd = 1.0f / sqrtf(row_total(s) / static_cast<float>(C) + d.eps);
    const QuantP qq = make_quantp(aqp0);
    const int sw = (pl >> 2) & 3;
    auto norm_quant = [&](auto exact_div) {
      constexpr bool EX = decltype(exact_div)::value;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 gm = *reinterpret_cast<const float4*>(lngb + 32 * t + 16 * h + 4 * i);
          const float4 bt = *reinterpret_cast<const float4*>(lngb + C + 32 * t + 16 * h + 4 * i);
          const float y0 = (v[t][4 * i] - mean) * rstd * gm.x + bt.x, y1 = (v[t][4 * i + 1] - mean) * rstd * gm.y + bt.y;
          const float y2 = (v[t][4 * i + 2] - mean) * rstd * gm.z + bt.z, y3 = (v[t][4 * i + 3] - mean) * rstd * gm.w + bt.w;
          w[i] = quant_pack4_t<EX>(f2{y0, y1}, f2{y2, y3}, qq);
        }
        *reinterpret_cast<uint4*>(Xw + (t >> 1) * 2048 + pl * 64 + (((2 * (t & 1) + h) ^ sw) << 4)) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    };
    if (__builtin_expect(qq.bad, 0)) norm_quant(std::true_type{});
    else norm_quant(std::false_type{});
  }

  // ---- fragment addressing (conv_lin.hip): activation rows by token, weight rows in the permuted order that makes register r = channel 16 h + r
  const float zp2 = aqp2.y;

  v16i acc2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[t][r] = 0;
  v4i hb[2];

  v16i av, ag;
  // ---- the five steps of a pair of hidden tiles: A0 (value | gate MFMAs of tile 2q), E0 (their bins), A1, E1, C (ff.net.2's K-step q).
  // Waves 0-3 (one per SIMD) run them at the global intervals 5q .. 5q + 4, waves 4-7 ONE INTERVAL LATER: the two waves of a SIMD are
  // then in complementary steps -- one issues MFMAs while the other does the GELU arithmetic -- in four of five intervals (in lockstep
  // the SIMD alternated between two waves of MFMAs and two waves of VALU work: 1014 us; staggered: see DESIGN.md).  One barrier per
  // interval; the weight stages are re-filled as soon as BOTH groups are through with them (stage kind lives in slot kind):
  //   interval 5q (+0): wait A0(q) | issue A1(q)   (+1): issue C(q)   (+2): wait A1(q)   (+3): issue A0(q+1)   (+4): wait C(q)
  auto sync = [&](auto r_tag, int qi) {
    constexpr int R = decltype(r_tag)::value;
    if constexpr (R == 0) ff_wait_vmcnt<0>();
    if constexpr (R == 2) ff_wait_vmcnt<PPW>();
    if constexpr (R == 4) {
      if (qi + 1 < npairs) ff_wait_vmcnt<PPW>();
      else ff_wait_vmcnt<0>();
    }
    asm volatile("s_barrier" ::: "memory");
    if constexpr (R == 0) {
      if (qi >= 1 && qi < npairs) issue(1, qi);
    }
    if constexpr (R == 1) {
      if (qi >= 1 && qi < npairs) issue(2, qi);
    }
    if constexpr (R == 3) {
      if (qi + 1 < npairs) issue(0, qi + 1);
    }
  };
  auto step_a = [&](auto kind_tag) {
    constexpr int kind = decltype(kind_tag)::value;
    const unsigned char* slot = lds + G::RING_OFF + kind * G::SLOT;
#pragma unroll
    for (int r = 0; r < 16; ++r) av[r] = ag[r] = 0;
#pragma unroll
    for (int sidx = 0; sidx < NCH; ++sidx)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        v4i xf, vf, gf;
        if constexpr (FF_ABLATE & 32) {
          xf = vf = gf = v4i{sidx, ks, lane, 1};
        } else {
          xf = *reinterpret_cast<const v4i*>(xfr + sidx * 2048 + (fsw ^ (ks << 5)));
          vf = *reinterpret_cast<const v4i*>(slot + sidx * 2048 + brow * 64 + (bsw ^ (ks << 5)));
          gf = *reinterpret_cast<const v4i*>(slot + NCH * 2048 + sidx * 2048 + brow * 64 + (bsw ^ (ks << 5)));
        }
        if constexpr (FF_ABLATE & 2) {
          av[ks] += vf[0] + xf[1];
          ag[ks] += gf[0];
        } else {
          av = __builtin_amdgcn_mfma_i32_32x32x32_i8(vf, xf, av, 0, 0, 0);
          ag = __builtin_amdgcn_mfma_i32_32x32x32_i8(gf, xf, ag, 0, 0, 0);
        }
      }
    // (pin the accumulators: the interval ends here)
    asm volatile("" : "+v"(av), "+v"(ag));
  };
  auto step_e = [&](auto kind_tag) {
    constexpr int kind = decltype(kind_tag)::value;
    // value' * gelu(gate) -> bins: TFMQ_OUT_GEGLU_Q8_FAST's arithmetic (conv_lin.hip), constants {sv, bv, sg, bg}[32] of this tile
    const float* cst = reinterpret_cast<const float*>(lds + G::RING_OFF + kind * G::SLOT + G::NPIECE * 1024) + kind * 128 + 16 * h;
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 sv, bv, sg, bg;
      if constexpr (FF_ABLATE & 8) {
        sv = bv = sg = bg = make_float4(zp2, 1.0f, 0.5f, 0.25f);
      } else {
        sv = *reinterpret_cast<const float4*>(cst + 4 * i), bv = *reinterpret_cast<const float4*>(cst + 32 + 4 * i);
        sg = *reinterpret_cast<const float4*>(cst + 64 + 4 * i), bg = *reinterpret_cast<const float4*>(cst + 96 + 4 * i);
      }
      if constexpr (FF_ABLATE & 1) {
        w[i] = static_cast<unsigned>(av[4 * i] + ag[4 * i + 1]) ^ __float_as_uint(sv.x + bg.y);
        continue;
      }
      const f2 a0 = pk_fma(f2{sv.x, sv.y}, f2{static_cast<float>(av[4 * i]), static_cast<float>(av[4 * i + 1])}, f2{bv.x, bv.y});
      const f2 a1 = pk_fma(f2{sv.z, sv.w}, f2{static_cast<float>(av[4 * i + 2]), static_cast<float>(av[4 * i + 3])}, f2{bv.z, bv.w});
      const f2 g0 = pk_fma(f2{sg.x, sg.y}, f2{static_cast<float>(ag[4 * i]), static_cast<float>(ag[4 * i + 1])}, f2{bg.x, bg.y});
      const f2 g1 = pk_fma(f2{sg.z, sg.w}, f2{static_cast<float>(ag[4 * i + 2]), static_cast<float>(ag[4 * i + 3])}, f2{bg.z, bg.w});
      w[i] = geglu_fast_pack4(a0, gelu_fast2(g0), a1, gelu_fast2(g1), zp2);
      __builtin_amdgcn_sched_barrier(0);        // four outputs at a time: interleaving the groups spilled the accumulators
    }
    // (pin the bins here: without it the compiler sinks this step's arithmetic below the next step's MFMAs and keeps both
    // steps' accumulators and constants alive -- 180 spilled registers)
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    hb[kind] = v4i{static_cast<int>(w[0]), static_cast<int>(w[1]), static_cast<int>(w[2]), static_cast<int>(w[3])};
  };
  auto step_c = [&]() {
    const unsigned char* slot = lds + G::RING_OFF + 2 * G::SLOT;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        v4i wf;
        if constexpr (FF_ABLATE & 32) wf = v4i{t, ks, lane, 1};
        else wf = *reinterpret_cast<const v4i*>(slot + t * 2048 + brow * 64 + (bsw ^ (ks << 5)));
        if constexpr (FF_ABLATE & 4) acc2[t][ks] += wf[0] + hb[ks][1];
        else acc2[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, hb[ks], acc2[t], 0, 0, 0);
      }
  };
  auto run = [&](auto g_tag) {
    constexpr int GR = decltype(g_tag)::value;
    if constexpr (GR == 1) sync(std::integral_constant<int, 0>{}, 0);
    for (int q = 0; q < npairs; ++q) {
      sync(std::integral_constant<int, (0 + GR) % 5>{}, q + (0 + GR) / 5);
      step_a(std::integral_constant<int, 0>{});
      sync(std::integral_constant<int, (1 + GR) % 5>{}, q + (1 + GR) / 5);
      step_e(std::integral_constant<int, 0>{});
      sync(std::integral_constant<int, (2 + GR) % 5>{}, q + (2 + GR) / 5);
      step_a(std::integral_constant<int, 1>{});
      sync(std::integral_constant<int, (3 + GR) % 5>{}, q + (3 + GR) / 5);
      step_e(std::integral_constant<int, 1>{});
      sync(std::integral_constant<int, (4 + GR) % 5>{}, q + (4 + GR) / 5);
      step_c();
    }
    if constexpr (GR == 0) sync(std::integral_constant<int, 0>{}, npairs);
  };
  if (wid < 4) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  // ---- epilogue: scale * float(acc + kc) + bias + x -> fp16 rows / int8 bins (k_lin_direct's operations); the ring is idle: staging
  asm volatile("s_barrier" ::: "memory");
  const bool q8 = d.oq.qtable != nullptr;
  float2 oqp = make_float2(1.0f, 0.0f);
  if (q8) oqp = load_qparam(d.oq);
  const QuantP qP = make_quantp(oqp);
  unsigned char* stg = lds + G::RING_OFF + wid * (32 * FF_STG_ROW);
  auto epilogue = [&](auto q8_t, auto exact_div) {
  constexpr bool Q8 = decltype(q8_t)::value, EX = decltype(exact_div)::value;
#pragma unroll
  for (int tp = 0; tp < NT / 2; ++tp) {
    uint4 rr[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) rr[j][u] = *reinterpret_cast<const uint4*>(xrow + 32 * (2 * tp + j) + 8 * u);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = 2 * tp + j;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ct = 32 * t + 16 * h + 8 * u;
        f2 vv[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float4 sc = *reinterpret_cast<const float4*>(cs2 + ct + 4 * e);
          const int4 kc = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(cs2) + C + ct + 4 * e);
          const float4 bb = *reinterpret_cast<const float4*>(cs2 + 2 * C + ct + 4 * e);
          vv[2 * e] = f2{sc.x, sc.y} * f2{static_cast<float>(acc2[t][8 * u + 4 * e] + kc.x), static_cast<float>(acc2[t][8 * u + 4 * e + 1] + kc.y)} + f2{bb.x, bb.y};
          vv[2 * e + 1] = f2{sc.z, sc.w} * f2{static_cast<float>(acc2[t][8 * u + 4 * e + 2] + kc.z), static_cast<float>(acc2[t][8 * u + 4 * e + 3] + kc.w)} + f2{bb.z, bb.w};
        }
        const uint4 rw = rr[j][u];
        const float2 r0 = __half22float2(*reinterpret_cast<const __half2*>(&rw.x)), r1 = __half22float2(*reinterpret_cast<const __half2*>(&rw.y));
        const float2 r2 = __half22float2(*reinterpret_cast<const __half2*>(&rw.z)), r3 = __half22float2(*reinterpret_cast<const __half2*>(&rw.w));
        vv[0] += f2{r0.x, r0.y};
        vv[1] += f2{r1.x, r1.y};
        vv[2] += f2{r2.x, r2.y};
        vv[3] += f2{r3.x, r3.y};
        if constexpr (Q8) {
          const unsigned q0 = quant_pack4_t<EX>(vv[0], vv[1], qP), q1 = quant_pack4_t<EX>(vv[2], vv[3], qP);
          if constexpr (POST) {        // the bins are the next Linear's operand: this lane's 16 channels of tile t, 8 bytes at a time
            *reinterpret_cast<uint2*>(Xw + (t >> 1) * 2048 + pl * 64 + (((2 * (t & 1) + h) ^ swz_x) << 4) + 8 * u) = make_uint2(q0, q1);
          } else {
            *reinterpret_cast<uint2*>(stg + pl * FF_STG_ROW_Q8 + j * 32 + 16 * h + 8 * u) = make_uint2(q0, q1);
          }
        } else {
          *reinterpret_cast<uint4*>(stg + pl * FF_STG_ROW + (j * 32 + 16 * h + 8 * u) * 2) =
              make_uint4(pack_h2(vv[0].x, vv[0].y), pack_h2(vv[1].x, vv[1].y), pack_h2(vv[2].x, vv[2].y), pack_h2(vv[3].x, vv[3].y));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (POST) continue;
    // wave-private transpose: whole row segments out (8 lanes x 16 B per fp16 row of 64 channels; 4 lanes per int8 row)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if constexpr (Q8) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 16 + (lane >> 2), pc = lane & 3;
        const uint4 w = *reinterpret_cast<const uint4*>(stg + row * FF_STG_ROW_Q8 + pc * 16);
        const int m2 = m0 + wid * 32 + row;
        if (m2 < d.M) *reinterpret_cast<uint4*>(d.yq + static_cast<size_t>(m2) * C + 64 * tp + pc * 16) = w;
      }
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), pc = lane & 7;
        const uint4 w = *reinterpret_cast<const uint4*>(stg + row * FF_STG_ROW + pc * 16);
        const int m2 = m0 + wid * 32 + row;
        if (m2 < d.M) *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(d.y) + static_cast<size_t>(m2) * C + 64 * tp + pc * 8) = w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  };
  if constexpr (POST) {
    issue_lin(1, 0);                  // (slots 0 / 1 are idle since the barrier above)
    if (__builtin_expect(qP.bad, 0)) epilogue(std::true_type{}, std::true_type{});
    else epilogue(std::true_type{}, std::false_type{});
    // proj_out on the bins just written: + bias + the SpatialTransformer's input -> fp16 rows + the consumer GroupNorm's statistics
    lin_all(std::integral_constant<int, 1>{}, raw, reinterpret_cast<const __half*>(d.res_post), reinterpret_cast<__half*>(d.y_post));
  } else {
    if (!q8) epilogue(std::false_type{}, std::false_type{});
    else if (__builtin_expect(qP.bad, 0)) epilogue(std::true_type{}, std::true_type{});
    else epilogue(std::true_type{}, std::false_type{});
  }
  (void)mok;
}

}  // namespace

extern "C" int tfmq_ff_fused(tfmq_handle h, const tfmq_ff_desc* dd, void* stream) {
  TFMQ_CHECK_ARG(h, h && dd, "ff_fused: null pointer");
  const tfmq_ff_desc& d = *dd;
  const bool pre = d.w0 != nullptr, post = d.w3 != nullptr;
  TFMQ_CHECK_ARG(h, d.M > 0 && (d.x || pre) && d.gamma && d.beta && d.w1 && d.wmeta1 && d.wscale1 && d.w2 && d.wmeta2 && d.wscale2 && d.ws,
                 "ff_fused: null operand");
  TFMQ_CHECK_ARG(h, d.aq0.qtable && d.aq2.qtable, "ff_fused: both activation quantizers are required");
  TFMQ_CHECK_ARG(h, post || (d.oq.qtable && d.yq) || (!d.oq.qtable && d.y), "ff_fused: fp16 output y, or oq with the int8 output yq");
  TFMQ_CHECK_ARG(h, !pre || (d.xq_pre && d.wmeta0 && d.wscale0 && d.aq_pre.qtable && d.res_pre && d.y_pre), "ff_fused: the Linear in front needs xq_pre, w0 / wmeta0 / wscale0, aq_pre, res_pre and y_pre");
  TFMQ_CHECK_ARG(h, !post || (d.wmeta3 && d.wscale3 && d.oq.qtable && d.res_post && d.y_post &&
                              (!d.stats || ((d.stats_seg == 16 || d.stats_seg == 32 || d.stats_seg == 64 || d.stats_seg == 128)))),
                 "ff_fused: the Linear behind needs w3 / wmeta3 / wscale3, oq (its activation quantizer), res_post, y_post (and stats_seg 16 ... 128 with stats)");
  if ((pre || post) && d.M % 256 != 0) {
    h->err = "ff_fused: the Linears in front of / behind the feed-forward need M % 256 == 0";
    return TFMQ_ERR_UNSUPPORTED;
  }
  if (d.C != 320 || d.inner % 64 != 0 || d.inner <= 0) {
    h->err = "ff_fused: token width 320 and inner % 64 == 0 only";
    return TFMQ_ERR_UNSUPPORTED;
  }
  TFMQ_CHECK_ARG(h, static_cast<size_t>(d.M) * d.C < (static_cast<size_t>(1) << 31), "ff_fused: M * C must stay below 2^31");
  hipStream_t st = as_stream(stream);
  FfP p;
  p.d = d;
  p.pad_table = h->pad_table;
  static const int prio_env = getenv("TFMQ_SETPRIO") ? atoi(getenv("TFMQ_SETPRIO")) : 0;
  p.prio = prio_env;
  hipLaunchKernelGGL(k_ff_fold, dim3((d.inner + 255) / 256), dim3(256), 0, st, d);
  if (pre || post) hipLaunchKernelGGL(k_ff_fold_lin, dim3((d.C + 255) / 256, 2), dim3(256), 0, st, d);
  const dim3 grid((d.M + 255) / 256);
  if (pre && post) hipLaunchKernelGGL((k_ff_fused<320, true, true>), grid, dim3(512), 0, st, p);
  else if (pre) hipLaunchKernelGGL((k_ff_fused<320, true, false>), grid, dim3(512), 0, st, p);
  else if (post) hipLaunchKernelGGL((k_ff_fused<320, false, true>), grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL((k_ff_fused<320, false, false>), grid, dim3(512), 0, st, p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
