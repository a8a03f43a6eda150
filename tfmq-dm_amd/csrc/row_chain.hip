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

constexpr int RC_C = 320, RC_NCH = RC_C / 64, RC_NT = RC_C / 32;
constexpr int RC_XW = RC_NCH * 2048;                    // a wave's quantised rows
constexpr int RC_SLOT = 2 * RC_NCH * 2048 + 1024;       // two output tiles x K-steps + the phase's constants
constexpr int RC_RING_OFF = 0;
constexpr int RC_X_OFF = 2 * RC_SLOT;
constexpr int RC_TAB_OFF = RC_X_OFF + 8 * RC_XW;        // GroupNorm A | B of the block's image, then the LayerNorm's gamma | beta
constexpr int RC_STG_OFF = RC_TAB_OFF + 2 * RC_C * 4;
constexpr int RC_STG_ROW = 80;                          // 32 fp16 + 16 bytes per staged token row
constexpr int RC_TOTAL = RC_STG_OFF + 8 * 32 * RC_STG_ROW;
constexpr int RC_NPIECE = 2 * RC_NCH * 2;               // 20 weight pieces of 1 KiB per phase (+ 1 of constants)
constexpr int RC_PPW = 3;

struct ChainP {
  tfmq_chain_desc d;
  int nphase;
  int ph0[4];          // first phase of GEMM g (ph0[n_gemm] = nphase)
};

template <int N>
__device__ __forceinline__ void rc_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// per-column constants of every phase: ws[phase][{scale[64], kc[64] (int bits), bias[64], pad[64]}] (k_lin_direct's table)
__global__ __launch_bounds__(256) void k_chain_fold(ChainP p) {
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
  float* out = p.d.ws + static_cast<size_t>(p.ph0[g] + (n >> 6)) * 256 + (n & 63);
  out[0] = aqp.x * L.wscale[n];
  reinterpret_cast<int*>(out)[64] = (128 - static_cast<int>(aqp.y)) * (wmv.y - RC_C * wmv.x);
  out[128] = L.bias ? L.bias[n] : 0.0f;
}

__global__ __launch_bounds__(512, 2) void k_row_chain(ChainP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[RC_TOTAL];
  const tfmq_chain_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, pl = lane & 31;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 256;
  const int m = m0 + wid * 32 + pl;                 // (M % 256 == 0: the launcher's condition -- no ragged rows, fixed store counts per phase)

  // ---- weight stream: phase ph = two output tiles (64 columns) of the GEMM that owns it
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));
  const unsigned voff = static_cast<unsigned>((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
  auto gemm_of = [&](int ph) { return ph >= p.ph0[2] ? 2 : (ph >= p.ph0[1] ? 1 : 0); };
  auto issue = [&](int ph) {
    const int g = gemm_of(ph);
    const int pr = ph - p.ph0[g];
    const unsigned char* w = reinterpret_cast<const unsigned char*>(d.g[g].w);
    const unsigned sbase = lds0 + RC_RING_OFF + (ph & 1) * RC_SLOT;
#pragma unroll
    for (int it = 0; it < RC_PPW; ++it) {
      int pi = wid + 8 * it;
      if (pi == RC_NPIECE) {                        // the phase's constants: 1 KiB, lane-linear
        glds16_sv(reinterpret_cast<const unsigned char*>(d.ws) + static_cast<size_t>(ph) * 1024, static_cast<unsigned>(lane * 16),
                  sbase + __builtin_amdgcn_readfirstlane(RC_NPIECE * 1024));
        continue;
      }
      if (pi >= RC_NPIECE) pi -= 8;                 // surplus slot: the wave's previous piece again
      const int tile = pi / (2 * RC_NCH), rem = pi - tile * 2 * RC_NCH, s = rem >> 1, j = rem & 1;
      const unsigned char* src = w + ((static_cast<size_t>(2 * pr + tile) * RC_NCH + s) * 32 + j * 16) * 64;
      glds16_sv(src, voff, sbase + __builtin_amdgcn_readfirstlane(pi * 1024));
    }
  };

  unsigned char* Xw = lds + RC_X_OFF + wid * RC_XW;
  float* tab = reinterpret_cast<float*>(lds + RC_TAB_OFF);
  const int sw = (pl >> 2) & 3;
  auto x_store = [&](int t, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
    *reinterpret_cast<uint4*>(Xw + (t >> 1) * 2048 + pl * 64 + (((2 * (t & 1) + h) ^ sw) << 4)) = make_uint4(w0, w1, w2, w3);
  };

  // ---- input stage
  if (d.in_mode == 0) {
    // int8 rows (the attention kernel's output bins): 10 pieces per wave straight into the wave's X region
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(d.x) + static_cast<size_t>(m0 + wid * 32) * RC_C;
#pragma unroll
    for (int s = 0; s < RC_NCH; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        glds16_sv(xb + static_cast<size_t>(j * 16) * RC_C + s * 64, static_cast<unsigned>((lane >> 2) * RC_C + (((lane & 3) ^ ((lane >> 4) & 3)) * 16)),
                  lds0 + RC_X_OFF + __builtin_amdgcn_readfirstlane(wid * RC_XW + s * 2048 + j * 1024));
    issue(0);
  } else {
    issue(0);
    // fp16 rows + the GroupNorm's per-(image, channel) affine y = A x + B (k_gn_finalize), then the quantizer: k_gn_apply's operations
    const int img = m0 / d.T;
    if (tid < RC_C) {
      tab[tid] = d.gn_a[static_cast<size_t>(img) * RC_C + tid];
      tab[RC_C + tid] = d.gn_b[static_cast<size_t>(img) * RC_C + tid];
    }
    const __half* xrow = reinterpret_cast<const __half*>(d.x) + static_cast<size_t>(m) * RC_C + 16 * h;
    uint4 raw[RC_NT][2];
#pragma unroll
    for (int t = 0; t < RC_NT; ++t) {
      raw[t][0] = *reinterpret_cast<const uint4*>(xrow + 32 * t);
      raw[t][1] = *reinterpret_cast<const uint4*>(xrow + 32 * t + 8);
    }
    LDS_BARRIER();
    const QuantP qq = make_quantp(load_qparam(d.g[0].aq));
    auto gn_quant = [&](auto exact_div) {
      constexpr bool EX = decltype(exact_div)::value;
#pragma unroll
      for (int t = 0; t < RC_NT; ++t) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 a = *reinterpret_cast<const float4*>(tab + 32 * t + 16 * h + 4 * i);
          const float4 b = *reinterpret_cast<const float4*>(tab + RC_C + 32 * t + 16 * h + 4 * i);
          const __half2* hp = reinterpret_cast<const __half2*>(&raw[t][i >> 1]) + 2 * (i & 1);
          const float2 f0 = __half22float2(hp[0]), f1 = __half22float2(hp[1]);
          const float y0 = a.x * f0.x + b.x, y1 = a.y * f0.y + b.y, y2 = a.z * f1.x + b.z, y3 = a.w * f1.y + b.w;
          w[i] = quant_pack4_t<EX>(f2{y0, y1}, f2{y2, y3}, qq);
        }
        x_store(t, w[0], w[1], w[2], w[3]);
      }
    };
    if (__builtin_expect(qq.bad, 0)) gn_quant(std::true_type{});
    else gn_quant(std::false_type{});
    LDS_BARRIER();                                   // the table is re-used for the LayerNorm's gamma | beta
  }
  if (d.ln_gamma && tid < RC_C) {
    tab[tid] = d.ln_gamma[tid];
    tab[RC_C + tid] = d.ln_beta[tid];
  }

  const int fsw = (h ^ ((pl >> 2) & 3)) << 4;
  const int brow = lin_brow(pl);
  const int bsw = (h ^ ((brow >> 2) & 3)) << 4;
  const unsigned char* xfr = Xw + pl * 64;
  unsigned char* stg = lds + RC_STG_OFF + wid * (32 * RC_STG_ROW);
  uint4 kept[RC_NT][2];                               // the fp16 row of a GEMM whose output a LayerNorm consumes (N = C)

  int s_prev = -1;                                    // stores of the previous phase: 4 (row tiles) or 32 (transposed tiles); -1 = first phase
  // one phase: two output tiles (64 columns) of GEMM g.  KEEP >= 0: they are tiles 2 KEEP, 2 KEEP + 1 of a row a LayerNorm will consume
  auto phase = [&](auto keep_tag, int ph, int g, int pr) {
    constexpr int KEEP = decltype(keep_tag)::value;
    if (s_prev == 4) rc_wait_vmcnt<4>();
    else if (s_prev == 32) rc_wait_vmcnt<32>();
    else rc_wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    if (ph + 1 < p.nphase) issue(ph + 1);
    const tfmq_chain_gemm& L = d.g[g];
    const unsigned char* slot = lds + RC_RING_OFF + (ph & 1) * RC_SLOT;
    const float* cs = reinterpret_cast<const float*>(slot + RC_NPIECE * 1024);
    const bool transposed = L.yt != nullptr && 64 * pr >= L.t_col0;
    const bool has_res = L.residual != nullptr;
    uint4 rr[2][2];
    if (has_res) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          rr[j][u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(L.residual) + static_cast<size_t>(m) * L.N + 64 * pr + 32 * j + 16 * h + 8 * u);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      v16i acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
      for (int sidx = 0; sidx < RC_NCH; ++sidx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const v4i xf = *reinterpret_cast<const v4i*>(xfr + sidx * 2048 + (fsw ^ (ks << 5)));
          const v4i wf = *reinterpret_cast<const v4i*>(slot + (j * RC_NCH + sidx) * 2048 + brow * 64 + (bsw ^ (ks << 5)));
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, acc, 0, 0, 0);
        }
      // epilogue of this lane's 16 channels: scale * float(acc + kc) + bias (+ residual), k_lin_direct's operations
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
      const int n0 = 64 * pr + 32 * j;                 // first output channel of this tile
      if constexpr (KEEP >= 0) {                      // (N = C: tile index = n0 / 32 = 2 KEEP + j)
        kept[2 * KEEP + j][0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        kept[2 * KEEP + j][1] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
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
        *reinterpret_cast<uint4*>(stg + pl * RC_STG_ROW + 32 * h) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *reinterpret_cast<uint4*>(stg + pl * RC_STG_ROW + 32 * h + 16) = make_uint4(hw[4], hw[5], hw[6], hw[7]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 16 + (lane >> 2), pc = lane & 3;
          const uint4 w = *reinterpret_cast<const uint4*>(stg + row * RC_STG_ROW + pc * 16);
          *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(L.y) + static_cast<size_t>(m0 + wid * 32 + row) * L.ldy + n0 + pc * 8) = w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    s_prev = transposed ? 32 : 4;
  };

  int ph = 0;
  for (int g = 0; g < d.n_gemm; ++g) {
    const tfmq_chain_gemm& L = d.g[g];
    if (!L.next) {
      for (int pr = 0; pr < (L.N >> 6); ++pr, ++ph) phase(std::integral_constant<int, -1>{}, ph, g, pr);
      continue;
    }
    phase(std::integral_constant<int, 0>{}, ph, g, 0);
    phase(std::integral_constant<int, 1>{}, ph + 1, g, 1);
    phase(std::integral_constant<int, 2>{}, ph + 2, g, 2);
    phase(std::integral_constant<int, 3>{}, ph + 3, g, 3);
    phase(std::integral_constant<int, 4>{}, ph + 4, g, 4);
    ph += 5;
    {
      // LayerNorm of the row this lane just finished (its fp16-rounded values, as the stand-alone kernel reads them back) in
      // k_layernorm_hs<8>'s summation order (ff_fused.hip), then the next GEMM's quantizer: the bins replace the wave's X rows
      // (the row stays packed: 80 registers; every pass widens a tile's 16 values when it needs them)
      auto widen = [&](int t, float (&v)[16]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const __half2* hp = reinterpret_cast<const __half2*>(&kept[t][e]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float2 f = __half22float2(hp[i]);
            v[8 * e + 2 * i] = f.x;
            v[8 * e + 2 * i + 1] = f.y;
          }
        }
      };
      auto row_total = [&](float (&s)[2][2]) -> float {
        float tot8[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const float a = s[r][0] + s[r][1];
          const auto swp = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
          tot8[r] = __uint_as_float(swp[0]) + __uint_as_float(swp[1]);
        }
        return tot8[0] + tot8[1];
      };
      float s[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
      for (int t = 0; t < RC_NT; ++t) {
        float v[16];
        widen(t, v);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float* q = &v[8 * e];
          s[t & 1][e] += ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
        }
      }
      const float mean = row_total(s) / static_cast<float>(RC_C);
      s[0][0] = s[0][1] = s[1][0] = s[1][1] = 0.0f;
#pragma unroll
      for (int t = 0; t < RC_NT; ++t) {
        float v[16];
        widen(t, v);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float tt = 0.0f;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float a = v[8 * e + i] - mean;
            tt = __builtin_fmaf(a, a, tt);
          }
          s[t & 1][e] += tt;
        }
      }
      const float rstd = 1.0f / sqrtf(row_total(s) / static_cast<float>(RC_C) + d.ln_eps);
      const QuantP qq = make_quantp(load_qparam(d.g[g + 1].aq));
      auto norm_quant = [&](auto exact_div) {
        constexpr bool EX = decltype(exact_div)::value;
#pragma unroll
        for (int t = 0; t < RC_NT; ++t) {
          float v[16];
          widen(t, v);
          unsigned w[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 gm = *reinterpret_cast<const float4*>(tab + 32 * t + 16 * h + 4 * i);
            const float4 bt = *reinterpret_cast<const float4*>(tab + RC_C + 32 * t + 16 * h + 4 * i);
            const float y0 = (v[4 * i] - mean) * rstd * gm.x + bt.x, y1 = (v[4 * i + 1] - mean) * rstd * gm.y + bt.y;
            const float y2 = (v[4 * i + 2] - mean) * rstd * gm.z + bt.z, y3 = (v[4 * i + 3] - mean) * rstd * gm.w + bt.w;
            w[i] = quant_pack4_t<EX>(f2{y0, y1}, f2{y2, y3}, qq);
          }
          x_store(t, w[0], w[1], w[2], w[3]);
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
  if (d.C != RC_C || d.M % 256 != 0 || (d.in_mode != 0 && d.T % 256 != 0)) {
    h->err = "row_chain: token width 320, M % 256 == 0 (and T % 256 == 0 with the GroupNorm input stage) only";
    return TFMQ_ERR_UNSUPPORTED;
  }
  TFMQ_CHECK_ARG(h, d.in_mode == 0 || (d.in_mode == 2 && d.gn_a && d.gn_b), "row_chain: in_mode 0 (int8 rows) or 2 (fp16 rows + GroupNorm affine gn_a / gn_b)");
  TFMQ_CHECK_ARG(h, static_cast<size_t>(d.M) * 960 < (static_cast<size_t>(1) << 31), "row_chain: M too large");
  ChainP p;
  p.d = d;
  int ph = 0, cols = 0, n_ln = 0;
  for (int g = 0; g < 4; ++g) p.ph0[g] = 1 << 30;
  for (int g = 0; g < d.n_gemm; ++g) {
    const tfmq_chain_gemm& L = d.g[g];
    TFMQ_CHECK_ARG(h, L.w && L.wmeta && L.wscale && L.aq.qtable && L.N > 0 && L.N % 64 == 0, "row_chain: a GEMM needs w, wmeta, wscale, aq and N % 64 == 0");
    TFMQ_CHECK_ARG(h, (L.y && L.ldy >= (L.yt ? L.t_col0 : L.N) && L.ldy % 8 == 0) || (L.yt && L.t_col0 == 0), "row_chain: output y / ldy");
    TFMQ_CHECK_ARG(h, !L.yt || (L.t_col0 % 64 == 0 && L.t_col0 >= 0 && L.t_col0 < L.N), "row_chain: transposed region starts at a multiple of 64 columns");
    TFMQ_CHECK_ARG(h, !L.next || (L.N == RC_C && g + 1 < d.n_gemm && d.ln_gamma && d.ln_beta), "row_chain: a LayerNorm follows a C-wide GEMM that is not the last");
    n_ln += L.next ? 1 : 0;
    p.ph0[g] = ph;
    ph += L.N / 64;
    cols += L.N;
  }
  TFMQ_CHECK_ARG(h, n_ln <= 1, "row_chain: at most one LayerNorm per chain");
  p.ph0[d.n_gemm] = ph;
  p.nphase = ph;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_chain_fold, dim3((cols + 255) / 256), dim3(256), 0, st, p);
  hipLaunchKernelGGL(k_row_chain, dim3(d.M / 256), dim3(512), 0, st, p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
