// K5c (round 4): chains of token Linears around the attention of a BasicTransformerBlock as ONE launch, a token per lane.
//
//   "pre"   x -> GroupNorm affine -> quantise -> proj_in (+ bias) -> h (fp16, stored) -> LayerNorm -> quantise -> fused to_q | to_k | to_v
//           -> q | k as fp16 rows, v as its fp16 transpose        (SpatialTransformer.forward ldm/modules/attention.py:238-261: self.norm,
//           self.proj_in; BasicTransformerBlock._forward :212: self.attn1(self.norm1(x)); CrossAttention.forward :168-177)
//   "mid"   attention output bins -> to_out (+ bias) + x -> x' (fp16, stored) -> LayerNorm -> quantise -> attn2.to_q -> fp16 rows
//           (CrossAttention.forward :194 to_out; BasicTransformerBlock._forward :212-213: x = attn1(...) + x; attn2(self.norm2(x)))
// every Linear a w4a8 QuantLayer (quant/quant_layer.py:306-340), the quantised blocks of quant/quant_block.py:178-299.
//
// As separate launches these stages are bound by bytes and by the life of short blocks: each writes a tensor the next reads back (the pre
// chain: 4 launches, 7360 B per token through HBM, 0.85 ms at the 64 x 64 level of SD).  Here a workgroup keeps 256 tokens for the
// whole chain (the layout of ff_fused.hip): a wave owns 32 tokens, lane = token, the quantised row lives in the wave's 10 KB of LDS in
// the K-step-major swizzled layout, a GEMM's output tile leaves the accumulators as the lane's 16 consecutive channels -- the fp16 row
// the LayerNorm needs stays packed in 80 registers (and is stored once, for the residual that follows), its bins become the next GEMM's
// operand in place.  Only weights stream: 20 KB phases (two 32-channel output tiles x K = 320) + their folded per-column constants through
// a 2-slot LDS-DMA ring, one counted s_waitcnt + s_barrier per phase.  HBM sees the input row, the stored stream row and the outputs.
// Every stage repeats the operations of the stand-alone kernel it replaces (k_gn_apply_h8, k_lin_direct's epilogue, k_layernorm_hs in
// its summation order): the chain and the launches agree bit for bit (tests/test_row_chain_gpu.py).
#include "conv_common.hpp"
#include <type_traits>

namespace {

// W = C / 320 waves share a group of 32 tokens: W = 1 (C = 320): 8 token groups, a wave computes both output tiles of a phase over
// the whole K; W = 2 (C = 640, the 32 x 32 level): 4 token groups, the two waves of a group take one output tile each, a phase covers
// HALF of K (the accumulators live across the two phases of a tile pair), each wave normalises / quantises / stores its own half of the
// channels and the LayerNorm's row statistics cross the pair through LDS (k_layernorm_hs<16>'s tree: the pair's partial sums meet
// where that kernel's lanes 0-3 | 4-7 and 8-11 | 12-15 meet).
template <int W>
struct RcGeo {
  static constexpr int C = 320 * W, NCH = C / 64, NT = C / 32;
  static constexpr int TG = 8 / W, BT = 32 * TG;               // token groups and tokens per workgroup
  static constexpr int TPW = 2 / W;                            // output tiles a wave computes per phase
  static constexpr int XG = NCH * 2048;                        // a token group's quantised rows
  static constexpr int SLOT = 2 * 5 * 2048 + 1024;             // two output tiles x five K-steps + the tile pair's constants
  static constexpr int RING_OFF = 0;
  static constexpr int X_OFF = 2 * SLOT;
  static constexpr int TAB_OFF = X_OFF + TG * XG;              // GroupNorm A | B of the block's image, then the LayerNorm's gamma | beta
  static constexpr int STG_OFF = TAB_OFF + 2 * C * 4;
  static constexpr int STG_ROW = 80;                           // 32 fp16 + 16 bytes per staged token row
  static constexpr int EXC_OFF = STG_OFF + 8 * 32 * STG_ROW;   // W = 2: the pair's partial row sums (mean pass | variance pass)
  static constexpr int TOTAL = EXC_OFF + (W > 1 ? 2 * TG * W * 2 * 32 * 4 : 0);
  static constexpr int NPIECE = 20, PPW = 3;
  static constexpr int NKEEP = NT / W;                         // tiles of a row a wave keeps for the LayerNorm (10)
  static_assert(TOTAL <= 160 * 1024, "LDS");
};

struct ChainP {
  tfmq_chain_desc d;
  int nphase;
  int ph0[4];          // first phase of GEMM g (ph0[n_gemm] = nphase); a GEMM has (N / 64) * W phases
  int prio;            // TFMQ_SETPRIO=1 (A/B runs): waves 4-7 at s_setprio 1
};

template <int N>
__device__ __forceinline__ void rc_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// per-column constants of every tile pair: ws[phase of its LAST K part][{scale[64], kc[64] (int bits), bias[64], pad[64]}] (k_lin_direct's table)
__global__ __launch_bounds__(256) void k_chain_fold(ChainP p, int W) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  int g = 0, n = col;
  for (; g < p.d.n_gemm; ++g) {
    if (n < p.d.g[g].N) break;
    n -= p.d.g[g].N;
  }
  if (g >= p.d.n_gemm) return;
  const tfmq_chain_gemm& L = p.d.g[g];
  const float2 aqp = load_qparam(L.aq);
  const int4 wmv = reinterpret_cast<const int4*>(L.wmeta)[n];
  float* out = p.d.ws + static_cast<size_t>(p.ph0[g] + (n >> 6) * W + (W - 1)) * 256 + (n & 63);
  out[0] = aqp.x * L.wscale[n];
  reinterpret_cast<int*>(out)[64] = (128 - static_cast<int>(aqp.y)) * (wmv.y - p.d.C * wmv.x);
  out[128] = L.bias ? L.bias[n] : 0.0f;
}

template <int W>
__global__ __launch_bounds__(512, 2) void k_row_chain(ChainP p) {
  using G = RcGeo<W>;
  constexpr int C = G::C, NCH = G::NCH, TPW = G::TPW, NKEEP = G::NKEEP;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G::TOTAL];
  const tfmq_chain_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, pl = lane & 31;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wid / W, part = wid % W;            // token group of this wave, its share of the group's tiles
  const int m0 = blockIdx.x * G::BT;
  if (p.prio && wid >= 4) __builtin_amdgcn_s_setprio(1);
  const int m = m0 + tg * 32 + pl;                   // (M % BT == 0: the launcher's condition -- no ragged rows, fixed store counts per phase)

  // ---- weight stream: phase = two output tiles (64 columns) x five K-steps of the GEMM that owns it
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));
  const unsigned voff = static_cast<unsigned>((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
  auto gemm_of = [&](int ph) { return ph >= p.ph0[2] ? 2 : (ph >= p.ph0[1] ? 1 : 0); };
  auto issue = [&](int ph) {
    const int g = gemm_of(ph);
    const int pr = ph - p.ph0[g], tp = pr / W, kp = pr - tp * W;
    const unsigned char* w = reinterpret_cast<const unsigned char*>(d.g[g].w);
    const unsigned sbase = lds0 + G::RING_OFF + (ph & 1) * G::SLOT;
#pragma unroll
    for (int it = 0; it < G::PPW; ++it) {
      int pi = wid + 8 * it;
      if (pi == G::NPIECE) {                        // the tile pair's constants: 1 KiB, lane-linear (read after the last K part)
        glds16_sv(reinterpret_cast<const unsigned char*>(d.ws) + static_cast<size_t>(ph) * 1024, static_cast<unsigned>(lane * 16),
                  sbase + __builtin_amdgcn_readfirstlane(G::NPIECE * 1024));
        continue;
      }
      if (pi >= G::NPIECE) pi -= 8;                 // surplus slot: the wave's previous piece again
      const int tile = pi / 10, rem = pi - tile * 10, sI = rem >> 1, j = rem & 1;
      const unsigned char* src = w + ((static_cast<size_t>(2 * tp + tile) * NCH + 5 * kp + sI) * 32 + j * 16) * 64;
      glds16_sv(src, voff, sbase + __builtin_amdgcn_readfirstlane(pi * 1024));
    }
  };

  unsigned char* Xg = lds + G::X_OFF + tg * G::XG;
  float* tab = reinterpret_cast<float*>(lds + G::TAB_OFF);
  const int sw = (pl >> 2) & 3;
  auto x_store = [&](int t, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
    *reinterpret_cast<uint4*>(Xg + (t >> 1) * 2048 + pl * 64 + (((2 * (t & 1) + h) ^ sw) << 4)) = make_uint4(w0, w1, w2, w3);
  };
  // the tiles of a token row this wave normalises / quantises: i (W = 1), part + 2 i (W = 2), i = 0 .. 9.  `part` is a run-time value:
  // folded ONCE into the table / row bases (x_store_i, tab_p, below) so that every per-tile offset is an immediate
  const float* tab_p = tab + (W == 1 ? 0 : 32 * part) + 16 * h;
  unsigned char* xs2 = Xg + pl * 64 + (((2 * part + h) ^ sw) << 4);          // W = 2: tile part + 2 i -> chunk i, half `part`
  auto x_store_i = [&](int i, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
    if constexpr (W == 1) x_store(i, w0, w1, w2, w3);
    else *reinterpret_cast<uint4*>(xs2 + i * 2048) = make_uint4(w0, w1, w2, w3);
  };
  constexpr int TSTEP = 32 * W;                    // channels between consecutive tiles of this wave

  // ---- input stage
  if (d.in_mode == 0) {
    // int8 rows (the attention kernel's output bins) straight into the group's X region: 2 NCH pieces, split over the group's waves
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(d.x) + static_cast<size_t>(m0 + tg * 32) * C;
#pragma unroll
    for (int sI = 0; sI < 5; ++sI)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ch = part * 5 + sI;               // chunk (K-step) of the row
        glds16_sv(xb + static_cast<size_t>(j * 16) * C + ch * 64, static_cast<unsigned>((lane >> 2) * C + (((lane & 3) ^ ((lane >> 4) & 3)) * 16)),
                  lds0 + G::X_OFF + __builtin_amdgcn_readfirstlane(tg * G::XG + ch * 2048 + j * 1024));
      }
    issue(0);
  } else {
    issue(0);
    // fp16 rows + the GroupNorm's per-(image, channel) affine y = A x + B (k_gn_finalize), then the quantizer: k_gn_apply's operations
    const int img = m0 / d.T;
    for (int c = tid; c < C; c += 512) {
      tab[c] = d.gn_a[static_cast<size_t>(img) * C + c];
      tab[C + c] = d.gn_b[static_cast<size_t>(img) * C + c];
    }
    const __half* xrow = reinterpret_cast<const __half*>(d.x) + static_cast<size_t>(m) * C + 16 * h + (W == 1 ? 0 : 32 * part);
    uint4 raw[NKEEP][2];
#pragma unroll
    for (int i = 0; i < NKEEP; ++i) {
      raw[i][0] = *reinterpret_cast<const uint4*>(xrow + TSTEP * i);
      raw[i][1] = *reinterpret_cast<const uint4*>(xrow + TSTEP * i + 8);
    }
    LDS_BARRIER();
    const QuantP qq = make_quantp(load_qparam(d.g[0].aq));
    auto gn_quant = [&](auto exact_div) {
      constexpr bool EX = decltype(exact_div)::value;
#pragma unroll
      for (int i = 0; i < NKEEP; ++i) {
        unsigned w[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 a = *reinterpret_cast<const float4*>(tab_p + TSTEP * i + 4 * q4);
          const float4 b = *reinterpret_cast<const float4*>(tab_p + C + TSTEP * i + 4 * q4);
          const __half2* hp = reinterpret_cast<const __half2*>(&raw[i][q4 >> 1]) + 2 * (q4 & 1);
          const float2 f0 = __half22float2(hp[0]), f1 = __half22float2(hp[1]);
          const float y0 = a.x * f0.x + b.x, y1 = a.y * f0.y + b.y, y2 = a.z * f1.x + b.z, y3 = a.w * f1.y + b.w;
          w[q4] = quant_pack4_t<EX>(f2{y0, y1}, f2{y2, y3}, qq);
        }
        x_store_i(i, w[0], w[1], w[2], w[3]);
      }
    };
    if (__builtin_expect(qq.bad, 0)) gn_quant(std::true_type{});
    else gn_quant(std::false_type{});
    LDS_BARRIER();                                   // the table is re-used for the LayerNorm's gamma | beta
  }
  if (d.ln_gamma) {
    for (int c = tid; c < C; c += 512) {
      tab[c] = d.ln_gamma[c];
      tab[C + c] = d.ln_beta[c];
    }
  }

  const int fsw = (h ^ ((pl >> 2) & 3)) << 4;
  const int brow = lin_brow(pl);
  const int bsw = (h ^ ((brow >> 2) & 3)) << 4;
  const unsigned char* xfr = Xg + pl * 64;
  // (W = 2: the wave's tile of a phase -- `part` -- is a run-time value: folded once into the fragment / constant bases, so that every
  // other offset of a phase stays an immediate; computed per phase the compiler kept a register per phase and spilled 39)
  const int part_w = W == 1 ? 0 : part * 5 * 2048, part_c = W == 1 ? 0 : part * 32;
  const unsigned char* wfr = lds + G::RING_OFF + brow * 64 + part_w;
  unsigned char* stg = lds + G::STG_OFF + wid * (32 * G::STG_ROW);
  uint4 kept[NKEEP][2];                               // this wave's tiles of the fp16 row a LayerNorm consumes (N = C)
  v16i acc[TPW];                                      // (W = 2: alive across the K parts of a tile pair)

  int s_prev = -1;                                    // stores of the previous phase (0, 2 / 4 row tiles, 16 / 32 transposed); -1 = first phase
  // one phase: K part kp of tile pair tp of GEMM g.  KEEP >= 0: the pair's tiles belong to a row a LayerNorm will consume (KEEP = tp)
  auto phase = [&](auto keep_tag, int ph, int g, int tp, int kp) {
    constexpr int KEEP = decltype(keep_tag)::value;
    if (s_prev == 4) rc_wait_vmcnt<4>();
    else if (s_prev == 32) rc_wait_vmcnt<32>();
    else if (s_prev == 2) rc_wait_vmcnt<2>();
    else if (s_prev == 16) rc_wait_vmcnt<16>();
    else rc_wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    if (ph + 1 < p.nphase) issue(ph + 1);
    const tfmq_chain_gemm& L = d.g[g];
    const unsigned char* wslot = wfr + (ph & 1) * G::SLOT;
    const float* cs = reinterpret_cast<const float*>(lds + G::RING_OFF + (ph & 1) * G::SLOT + G::NPIECE * 1024) + part_c;
    const unsigned char* xk = xfr + kp * (5 * 2048);
    const bool last = kp == W - 1;
    const bool transposed = L.yt != nullptr && 64 * tp >= L.t_col0;
    const bool has_res = L.residual != nullptr;
    uint4 rr[TPW][2];
    if (has_res && last) {
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          rr[j][u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(L.residual) + static_cast<size_t>(m) * L.N + 64 * tp + 32 * (part * TPW + j) + 16 * h + 8 * u);
    }
    if (kp == 0) {
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    }
    // the token fragment of a K slice is read ONCE for the wave's TPW tiles (with the tile loop outside, the stores of tile 0's epilogue
    // stood between the two reads of the same fragment and the compiler kept both: 2 ds_read_b128 per MFMA instead of 1.5)
#pragma unroll
    for (int sidx = 0; sidx < 5; ++sidx)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const v4i xf = *reinterpret_cast<const v4i*>(xk + sidx * 2048 + (fsw ^ (ks << 5)));
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const v4i wf = *reinterpret_cast<const v4i*>(wslot + ((W == 1 ? j : 0) * 5 + sidx) * 2048 + (bsw ^ (ks << 5)));
          acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, acc[j], 0, 0, 0);
        }
      }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int tl = part * TPW + j;                   // tile of the pair
      if (!last) continue;
      // epilogue of this lane's 16 channels: scale * float(acc + kc) + bias (+ residual), k_lin_direct's operations
      unsigned hw[8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ct = 32 * (W == 1 ? j : 0) + 16 * h + 8 * u;      // (+ 32 part: in cs)
        f2 vv[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float4 sc = *reinterpret_cast<const float4*>(cs + ct + 4 * e);
          const int4 kc = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(cs) + 64 + ct + 4 * e);
          const float4 bb = *reinterpret_cast<const float4*>(cs + 128 + ct + 4 * e);
          vv[2 * e] = f2{sc.x, sc.y} * f2{static_cast<float>(acc[j][8 * u + 4 * e] + kc.x), static_cast<float>(acc[j][8 * u + 4 * e + 1] + kc.y)} + f2{bb.x, bb.y};
          vv[2 * e + 1] = f2{sc.z, sc.w} * f2{static_cast<float>(acc[j][8 * u + 4 * e + 2] + kc.z), static_cast<float>(acc[j][8 * u + 4 * e + 3] + kc.w)} + f2{bb.z, bb.w};
        }
        if (has_res) {
          const uint4 rw = rr[j][u];
          const float2 r0 = __half22float2(*reinterpret_cast<const __half2*>(&rw.x)), r1 = __half22float2(*reinterpret_cast<const __half2*>(&rw.y));
          const float2 r2 = __half22float2(*reinterpret_cast<const __half2*>(&rw.z)), r3 = __half22float2(*reinterpret_cast<const __half2*>(&rw.w));
          vv[0] += f2{r0.x, r0.y};
          vv[1] += f2{r1.x, r1.y};
          vv[2] += f2{r2.x, r2.y};
          vv[3] += f2{r3.x, r3.y};
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) hw[4 * u + e] = pack_h2(vv[e].x, vv[e].y);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int n0 = 64 * tp + 32 * tl;                // first output channel of this tile
      if constexpr (KEEP >= 0) {
        kept[KEEP * TPW + j][0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        kept[KEEP * TPW + j][1] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
      }
      if (transposed) {
        // yt[b][n - t_col0][tok]: the 32 lanes of a half-wave hold 32 consecutive tokens of one channel: 64-byte runs per 2-byte store
        const int b = m / d.T, tok = m - b * d.T;
        __half* dst = reinterpret_cast<__half*>(L.yt) + (static_cast<size_t>(b) * (L.N - L.t_col0) + (n0 + 16 * h - L.t_col0)) * d.T + tok;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const __half2 v2 = *reinterpret_cast<const __half2*>(&hw[e]);
          dst[static_cast<size_t>(2 * e) * d.T] = __low2half(v2);
          dst[static_cast<size_t>(2 * e + 1) * d.T] = __high2half(v2);
        }
      } else {
        // row-major fp16 through the wave-private transpose: 4 lanes x 16 B = the tile's 64 bytes of a token row, 16 rows per store
        *reinterpret_cast<uint4*>(stg + pl * G::STG_ROW + 32 * h) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *reinterpret_cast<uint4*>(stg + pl * G::STG_ROW + 32 * h + 16) = make_uint4(hw[4], hw[5], hw[6], hw[7]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 16 + (lane >> 2), pc = lane & 3;
          const uint4 w = *reinterpret_cast<const uint4*>(stg + row * G::STG_ROW + pc * 16);
          *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(L.y) + static_cast<size_t>(m0 + tg * 32 + row) * L.ldy + n0 + pc * 8) = w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    s_prev = last ? (transposed ? 16 * TPW : 2 * TPW) : 0;
  };

  // the phases of a C-wide GEMM a LayerNorm follows, with compile-time tile indices (the packed row is kept in registers)
  auto keep_phases = [&](auto self, auto i_tag, int ph, int g) -> void {
    constexpr int I = decltype(i_tag)::value;           // phase index inside the GEMM: tile pair I / W, K part I % W
    if constexpr (I < (C / 64) * W) {
      phase(std::integral_constant<int, I / W>{}, ph + I, g, I / W, I % W);
      self(self, std::integral_constant<int, I + 1>{}, ph, g);
    }
  };

  int ph = 0;
  for (int g = 0; g < d.n_gemm; ++g) {
    const tfmq_chain_gemm& L = d.g[g];
    if (!L.next) {
      const int np = (L.N >> 6) * W;
      for (int pr = 0; pr < np; ++pr, ++ph) phase(std::integral_constant<int, -1>{}, ph, g, pr / W, pr % W);
      continue;
    }
    keep_phases(keep_phases, std::integral_constant<int, 0>{}, ph, g);
    ph += (C / 64) * W;
    {
      // LayerNorm of the row (its fp16-rounded values, as the stand-alone kernel reads them back) in k_layernorm_hs<8 W>'s summation order
      // (ff_fused.hip), then the next GEMM's quantizer: the bins replace the group's X rows.  The row stays packed (80 registers); every
      // pass widens a tile's 16 values when it needs them.  Piece idx = 4 t + 2 h + e of tile t belongs to sub-lane idx % (8 W).
      auto widen = [&](int i, float (&v)[16]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const __half2* hp = reinterpret_cast<const __half2*>(&kept[i][e]);
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float2 f = __half22float2(hp[q4]);
            v[8 * e + 2 * q4] = f.x;
            v[8 * e + 2 * q4 + 1] = f.y;
          }
        }
      };
      float* exc = reinterpret_cast<float*>(lds + G::EXC_OFF);
      auto row_total = [&](float (&s)[2][2], int pass) -> float {
        float tot8[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const float a = s[r][0] + s[r][1];
          const auto swp = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
          tot8[r] = __uint_as_float(swp[0]) + __uint_as_float(swp[1]);
        }
        if constexpr (W == 1) {
          return tot8[0] + tot8[1];
        } else {
          // the pair's waves hold the sub-lane groups {0-3, 8-11} | {4-7, 12-15}: (b0 + b4) + (b8 + b12), part 0's value first
          float* mine = exc + ((pass * G::TG + tg) * W + part) * 64;
          const float* other = exc + ((pass * G::TG + tg) * W + (part ^ 1)) * 64;
          if (h == 0) {
            mine[pl] = tot8[0];
            mine[32 + pl] = tot8[1];
          }
          LDS_BARRIER();
          const float o0 = other[pl], o1 = other[32 + pl];
          const float c0 = part == 0 ? tot8[0] + o0 : o0 + tot8[0];
          const float c1 = part == 0 ? tot8[1] + o1 : o1 + tot8[1];
          return c0 + c1;
        }
      };
      float s[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
      for (int i = 0; i < NKEEP; ++i) {
        float v[16];
        widen(i, v);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float* q = &v[8 * e];
          s[i & 1][e] += ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
        }
      }
      const float mean = row_total(s, 0) / static_cast<float>(C);
      s[0][0] = s[0][1] = s[1][0] = s[1][1] = 0.0f;
#pragma unroll
      for (int i = 0; i < NKEEP; ++i) {
        float v[16];
        widen(i, v);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float tt = 0.0f;
#pragma unroll
          for (int q8 = 0; q8 < 8; ++q8) {
            const float a = v[8 * e + q8] - mean;
            tt = __builtin_fmaf(a, a, tt);
          }
          s[i & 1][e] += tt;
        }
      }
      const float rstd = 1.0f / sqrtf(row_total(s, 1) / static_cast<float>(C) + d.ln_eps);
      const QuantP qq = make_quantp(load_qparam(d.g[g + 1].aq));
      auto norm_quant = [&](auto exact_div) {
        constexpr bool EX = decltype(exact_div)::value;
#pragma unroll
        for (int i = 0; i < NKEEP; ++i) {
          float v[16];
          widen(i, v);
          unsigned w[4];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 gm = *reinterpret_cast<const float4*>(tab_p + TSTEP * i + 4 * q4);
            const float4 bt = *reinterpret_cast<const float4*>(tab_p + C + TSTEP * i + 4 * q4);
            const float y0 = (v[4 * q4] - mean) * rstd * gm.x + bt.x, y1 = (v[4 * q4 + 1] - mean) * rstd * gm.y + bt.y;
            const float y2 = (v[4 * q4 + 2] - mean) * rstd * gm.z + bt.z, y3 = (v[4 * q4 + 3] - mean) * rstd * gm.w + bt.w;
            w[q4] = quant_pack4_t<EX>(f2{y0, y1}, f2{y2, y3}, qq);
          }
          x_store_i(i, w[0], w[1], w[2], w[3]);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (__builtin_expect(qq.bad, 0)) norm_quant(std::true_type{});
      else norm_quant(std::false_type{});
    }
  }
}

}  // namespace

extern "C" int tfmq_row_chain(tfmq_handle h, const tfmq_chain_desc* dd, void* stream) {
  TFMQ_CHECK_ARG(h, h && dd, "row_chain: null pointer");
  const tfmq_chain_desc& d = *dd;
  TFMQ_CHECK_ARG(h, d.M > 0 && d.x && d.ws && d.n_gemm >= 1 && d.n_gemm <= 3 && d.T > 0, "row_chain: bad argument");
  const int W = d.C / 320, BT = W ? 256 / W : 256;
  if ((d.C != 320 && d.C != 640) || d.M % BT != 0 || (d.in_mode != 0 && d.T % BT != 0)) {
    h->err = "row_chain: token width 320 / 640, M % (81920 / C) == 0 (and T likewise with the GroupNorm input stage) only";
    return TFMQ_ERR_UNSUPPORTED;
  }
  TFMQ_CHECK_ARG(h, d.in_mode == 0 || (d.in_mode == 2 && d.gn_a && d.gn_b), "row_chain: in_mode 0 (int8 rows) or 2 (fp16 rows + GroupNorm affine gn_a / gn_b)");
  TFMQ_CHECK_ARG(h, static_cast<size_t>(d.M) * 3 * d.C < (static_cast<size_t>(1) << 31), "row_chain: M too large");
  ChainP p;
  p.d = d;
  static const int prio_env = getenv("TFMQ_SETPRIO") ? atoi(getenv("TFMQ_SETPRIO")) : 0;
  p.prio = prio_env;
  int ph = 0, cols = 0, n_ln = 0;
  for (int g = 0; g < 4; ++g) p.ph0[g] = 1 << 30;
  for (int g = 0; g < d.n_gemm; ++g) {
    const tfmq_chain_gemm& L = d.g[g];
    TFMQ_CHECK_ARG(h, L.w && L.wmeta && L.wscale && L.aq.qtable && L.N > 0 && L.N % 64 == 0, "row_chain: a GEMM needs w, wmeta, wscale, aq and N % 64 == 0");
    TFMQ_CHECK_ARG(h, (L.y && L.ldy >= (L.yt ? L.t_col0 : L.N) && L.ldy % 8 == 0) || (L.yt && L.t_col0 == 0), "row_chain: output y / ldy");
    TFMQ_CHECK_ARG(h, !L.yt || (L.t_col0 % 64 == 0 && L.t_col0 >= 0 && L.t_col0 < L.N), "row_chain: transposed region starts at a multiple of 64 columns");
    TFMQ_CHECK_ARG(h, !L.next || (L.N == d.C && g + 1 < d.n_gemm && d.ln_gamma && d.ln_beta), "row_chain: a LayerNorm follows a C-wide GEMM that is not the last");
    n_ln += L.next ? 1 : 0;
    p.ph0[g] = ph;
    ph += (L.N / 64) * W;
    cols += L.N;
  }
  TFMQ_CHECK_ARG(h, n_ln <= 1, "row_chain: at most one LayerNorm per chain");
  p.ph0[d.n_gemm] = ph;
  p.nphase = ph;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_chain_fold, dim3((cols + 255) / 256), dim3(256), 0, st, p, W);
  if (W == 1) hipLaunchKernelGGL((k_row_chain<1>), dim3(d.M / 256), dim3(512), 0, st, p);
  else hipLaunchKernelGGL((k_row_chain<2>), dim3(d.M / 128), dim3(512), 0, st, p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
