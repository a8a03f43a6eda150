// K9: LayerNorm (+8-bit quantise) and GEGLU (+8-bit quantise) of the SpatialTransformer blocks
// (nn.LayerNorm ldm/modules/attention.py:203-205, GEGLU :37-44; each feeds QuantLayer aqtizers of
// to_q/to_k/to_v, ff.net.0.proj, ff.net.2 -- quant/quant_layer.py:318-325).  HBM-bound token-wise
// kernels: one wave per token row, 16-byte loads, the row lives in registers between the
// statistics and the apply step (one read of fp32, one write of int8).
#include "common.hpp"
#include <cstdlib>

// rows of up to 64*4*MAXV floats (MAXV float4 per lane): 5 -> C <= 1280.  A wave works on R rows at once: one row per wave
// was latency-bound at full occupancy (load -> two dependent wave reductions -> gamma / beta -> store, ~4 us per row with
// eight waves per SIMD resident = 2 TB/s at C = 320); with the loads of R rows in flight and their reductions interleaved
// the same chain is paid once per R rows (R = 4 for C <= 512: 266 -> 234 us at 524288 x 320 fp16; wider rows measured
// 3-8 % slower with R = 2 and keep R = 1).  Per-row arithmetic unchanged.
template <int MAXV, bool XH = false, int R = 1>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, long rows, int Cc,
                                                   tfmq_qsel aq, int8_t* __restrict__ yq, float* __restrict__ yf) {
  const long row0 = (static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * R;
  const int lane = threadIdx.x & 63;
  if (row0 >= rows) return;
  const int c4 = Cc / 4;
  float4 v[R][MAXV];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long row = row0 + r < rows ? row0 + r : rows - 1;          // past the end: a valid row, never stored
    const float4* xr = reinterpret_cast<const float4*>(x + row * Cc);
    const uint2* xh = reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(x) + row * Cc);   // XH: fp16 row
    s[r] = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int idx = lane + i * 64;
      if constexpr (XH) {
        v[r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < c4) {
          const uint2 u = xh[idx];
          const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&u.x)), hi = __half22float2(*reinterpret_cast<const __half2*>(&u.y));
          v[r][i] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
      } else {
        v[r][i] = idx < c4 ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s[r] += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) s[r] = wave_reduce_sum(s[r]);
  float mean[R], ss[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    mean[r] = s[r] / static_cast<float>(Cc);
    ss[r] = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (lane + i * 64 < c4) {
        const float a = v[r][i].x - mean[r], b = v[r][i].y - mean[r], c = v[r][i].z - mean[r], d = v[r][i].w - mean[r];
        ss[r] += (a * a + b * b) + (c * c + d * d);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) ss[r] = wave_reduce_sum(ss[r]);
#pragma unroll
  for (int r = 0; r < R; ++r) rstd[r] = 1.0f / sqrtf(ss[r] / static_cast<float>(Cc) + eps);
  const bool quant = aq.qtable != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(aq);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx >= c4) continue;
    const float4 g = reinterpret_cast<const float4*>(gamma)[idx];
    const float4 b = reinterpret_cast<const float4*>(beta)[idx];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long row = row0 + r;
      if (row >= rows) continue;
      float4 y;
      y.x = (v[r][i].x - mean[r]) * rstd[r] * g.x + b.x;
      y.y = (v[r][i].y - mean[r]) * rstd[r] * g.y + b.y;
      y.z = (v[r][i].z - mean[r]) * rstd[r] * g.z + b.z;
      y.w = (v[r][i].w - mean[r]) * rstd[r] * g.w + b.w;
      if (yf) reinterpret_cast<float4*>(yf + row * Cc)[idx] = y;
      if (quant) {
        char4 q;
        q = quant_char4(y.x, y.y, y.z, y.w, make_quantp(qp));
        reinterpret_cast<char4*>(yq + row * Cc)[idx] = q;
      }
    }
  }
}

// fp16 rows, sub-wave layout (the token widths of the LDM / SD transformers, C <= 40 * LPR): LPR lanes share a row, 64 / LPR
// rows per wave and iteration, lane j of a row owns its 16-byte pieces j, j + LPR, ... (one load instruction = whole 128-byte
// lines of 64 / LPR rows; every lane busy at C = 320, where a wave per row left 38 % of the lanes idle in every instruction).
// Persistent waves: gamma / beta of a lane's pieces stay in registers for all its rows, the next row group's pieces are
// requested before this group's arithmetic (two groups ahead measured 0-15 % slower), and the two row reductions are 3-5 DPP / swizzle steps instead of six
// ds_bpermute round trips.  Per-element arithmetic as in k_layernorm ((v - mean) * rstd * g + b, then the quantizer); the
// summation order of the statistics differs (tolerance-based parity, tests/test_hip_kernels.py).
template <int CTRL>
__device__ __forceinline__ float ln_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int LPR>
__device__ __forceinline__ float ln_group_sum(float v) {     // total over the LPR lanes of a row, identical in all of them
  v += ln_dpp<0xB1>(v);                                      // quad_perm [1,0,3,2]
  v += ln_dpp<0x4E>(v);                                      // quad_perm [2,3,0,1]
  v += ln_dpp<0x141>(v);                                     // row_half_mirror: lane i <-> 7 - i
  if constexpr (LPR >= 16) v += ln_dpp<0x140>(v);            // row_mirror: lane i <-> 15 - i
  if constexpr (LPR >= 32) v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));   // lane ^ 16
  if constexpr (LPR >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
template <int LPR, bool XH>
__global__ __launch_bounds__(256, 2) void k_layernorm_hs(const void* __restrict__ xv, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, long rows, int Cc,
                                                         tfmq_qsel aq, int8_t* __restrict__ yq, float* __restrict__ yf) {
  constexpr int NCH = 5, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, j = lane % LPR;
  const int chunks = Cc >> 3;
  const long nwaves = static_cast<long>(gridDim.x) * 4, wave = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const long ngroups = (rows + RPW - 1) / RPW;
  if (wave >= ngroups) return;
  float4 g[NCH][2], bt[NCH][2];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int idx = j + LPR * k;
    const bool ok = idx < chunks;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      g[k][u] = ok ? reinterpret_cast<const float4*>(gamma)[idx * 2 + u] : make_float4(0.f, 0.f, 0.f, 0.f);
      bt[k][u] = ok ? reinterpret_cast<const float4*>(beta)[idx * 2 + u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const bool quant = aq.qtable != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(aq);
  const QuantP qq = make_quantp(qp);
  const float invC = 1.0f / static_cast<float>(Cc);     // (only used for the comparison below; the mean divides)
  (void)invC;
  constexpr int PW = XH ? 1 : 2;                         // 16-byte loads per 8-channel piece (fp16 / fp32 rows)
  uint4 raw[NCH][PW];
  auto fetch = [&](long grp, uint4 (*dst)[PW]) {
    long row = grp * RPW + sub;
    row = row < rows ? row : rows - 1;                   // past the end: a valid row, never stored
    const uint4* xr = reinterpret_cast<const uint4*>(static_cast<const unsigned char*>(xv) + row * Cc * (XH ? 2 : 4));
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int idx = j + LPR * k, pi = idx < chunks ? idx : 0;   // a lane's surplus pieces re-read piece 0 (masked out of the sums)
#pragma unroll
      for (int u = 0; u < PW; ++u) dst[k][u] = xr[pi * PW + u];
    }
  };
  fetch(wave, raw);
  for (long grp = wave; grp < ngroups; grp += nwaves) {
    uint4 nxt[NCH][PW];
    const long ng = grp + nwaves;
    if constexpr (XH) fetch(ng < ngroups ? ng : grp, nxt);   // (the last iteration re-reads its own rows: no branch around loads)
    float v[NCH][8];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const bool ok = j + LPR * k < chunks;
      if constexpr (XH) {
        const __half2* hp = reinterpret_cast<const __half2*>(&raw[k][0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float2 f = __half22float2(hp[e]);
          v[k][2 * e] = ok ? f.x : 0.0f;
          v[k][2 * e + 1] = ok ? f.y : 0.0f;
        }
      } else {
        const float* fp = reinterpret_cast<const float*>(&raw[k][0]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[k][e] = ok ? fp[e] : 0.0f;
      }
      s += ((v[k][0] + v[k][1]) + (v[k][2] + v[k][3])) + ((v[k][4] + v[k][5]) + (v[k][6] + v[k][7]));
    }
    const float mean = ln_group_sum<LPR>(s) / static_cast<float>(Cc);
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      if (j + LPR * k < chunks) {
        float t = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = v[k][e] - mean;
          t = __builtin_fmaf(a, a, t);
        }
        ss += t;
      }
    }
    const float rstd = 1.0f / sqrtf(ln_group_sum<LPR>(ss) / static_cast<float>(Cc) + eps);
    const long row = grp * RPW + sub;
    if (row < rows) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int idx = j + LPR * k;
        if (idx >= chunks) continue;
        float y[8];
        const float* gk = reinterpret_cast<const float*>(&g[k][0]);
        const float* bk = reinterpret_cast<const float*>(&bt[k][0]);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (v[k][e] - mean) * rstd * gk[e] + bk[e];
        if (yf) {
          reinterpret_cast<float4*>(yf + row * Cc)[idx * 2] = make_float4(y[0], y[1], y[2], y[3]);
          reinterpret_cast<float4*>(yf + row * Cc)[idx * 2 + 1] = make_float4(y[4], y[5], y[6], y[7]);
        }
        if (quant) {
          uint2 w;
          w.x = quant_pack4(y[0], y[1], y[2], y[3], qq);
          w.y = quant_pack4(y[4], y[5], y[6], y[7], qq);
          reinterpret_cast<uint2*>(yq + row * Cc)[idx] = w;
        }
      }
    }
    if constexpr (XH) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) raw[k][0] = nxt[k][0];
    } else {                                             // fp32 rows (calibration, fp32 stream): no prefetch, same arithmetic
      if (ng < ngroups) fetch(ng, raw);
    }
  }
}

template <bool XH>
static void launch_layernorm_hs(tfmq_handle h, const void* x, const float* gamma, const float* beta, float eps, long rows, int C,
                                tfmq_qsel aq, int8_t* yq, float* yf, void* stream) {
  auto pgrid = [&](int rpw) {
    const long groups = (rows + rpw - 1) / rpw, blocks = (groups + 3) / 4, cap = static_cast<long>(h->cu_count) * 2;
    return dim3(static_cast<unsigned>(blocks < cap ? blocks : cap));
  };
  if (C <= 40 * 8) hipLaunchKernelGGL((k_layernorm_hs<8, XH>), pgrid(8), dim3(256), 0, as_stream(stream), x, gamma, beta, eps, rows, C, aq, yq, yf);
  else if (C <= 40 * 16) hipLaunchKernelGGL((k_layernorm_hs<16, XH>), pgrid(4), dim3(256), 0, as_stream(stream), x, gamma, beta, eps, rows, C, aq, yq, yf);
  else hipLaunchKernelGGL((k_layernorm_hs<32, XH>), pgrid(2), dim3(256), 0, as_stream(stream), x, gamma, beta, eps, rows, C, aq, yq, yf);
}

extern "C" int tfmq_layernorm(tfmq_handle h, const float* x, const float* gamma, const float* beta, float eps, long rows,
                              int C, tfmq_qsel aq, int8_t* yq, float* yf, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && gamma && beta && rows > 0 && C > 0, "layernorm: bad argument");
  TFMQ_CHECK_ARG(h, (aq.qtable && yq) || yf, "layernorm: no output requested");
  TFMQ_CHECK_ARG(h, C % 4 == 0 && C <= 64 * 4 * 8, "layernorm: C must be a multiple of 4 and <= 2048");
  if (C % 8 == 0 && C <= 40 * 32 && !getenv("TFMQ_LN_WAVE_PER_ROW")) {       // the same layout as the fp16 rows: bit-identical results
    launch_layernorm_hs<false>(h, x, gamma, beta, eps, rows, C, aq, yq, yf, stream);
    TFMQ_LAUNCH_CHECK(h);
    return TFMQ_OK;
  }
  auto grid = [&](int r) { return dim3(static_cast<unsigned>((rows + 4 * r - 1) / (4 * r))); };
  if (C <= 64 * 4 * 2) hipLaunchKernelGGL((k_layernorm<2, false, 4>), grid(4), dim3(256), 0, as_stream(stream), x, gamma, beta, eps, rows, C, aq, yq, yf);
  else if (C <= 64 * 4 * 5) hipLaunchKernelGGL((k_layernorm<5, false, 1>), grid(1), dim3(256), 0, as_stream(stream), x, gamma, beta, eps, rows, C, aq, yq, yf);
  else hipLaunchKernelGGL((k_layernorm<8, false, 1>), grid(1), dim3(256), 0, as_stream(stream), x, gamma, beta, eps, rows, C, aq, yq, yf);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_layernorm_h(tfmq_handle h, const uint16_t* x, const float* gamma, const float* beta, float eps, long rows,
                                int C, tfmq_qsel aq, int8_t* yq, float* yf, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && gamma && beta && rows > 0 && C > 0, "layernorm_h: bad argument");
  TFMQ_CHECK_ARG(h, (aq.qtable && yq) || yf, "layernorm_h: no output requested");
  TFMQ_CHECK_ARG(h, C % 4 == 0 && C <= 64 * 4 * 8, "layernorm_h: C must be a multiple of 4 and <= 2048");
  auto grid = [&](int r) { return dim3(static_cast<unsigned>((rows + 4 * r - 1) / (4 * r))); };
  const float* xf = reinterpret_cast<const float*>(x);
  if (C % 8 == 0 && C <= 40 * 32 && !getenv("TFMQ_LN_WAVE_PER_ROW")) {       // sub-wave rows, persistent waves
    launch_layernorm_hs<true>(h, x, gamma, beta, eps, rows, C, aq, yq, yf, stream);
    TFMQ_LAUNCH_CHECK(h);
    return TFMQ_OK;
  }
  if (C <= 64 * 4 * 2) hipLaunchKernelGGL((k_layernorm<2, true, 4>), grid(4), dim3(256), 0, as_stream(stream), xf, gamma, beta, eps, rows, C, aq, yq, yf);
  else if (C <= 64 * 4 * 5) hipLaunchKernelGGL((k_layernorm<5, true, 1>), grid(1), dim3(256), 0, as_stream(stream), xf, gamma, beta, eps, rows, C, aq, yq, yf);
  else hipLaunchKernelGGL((k_layernorm<8, true, 1>), grid(1), dim3(256), 0, as_stream(stream), xf, gamma, beta, eps, rows, C, aq, yq, yf);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// y[m][i] = h[m][i] * gelu(h[m][I + i]),  gelu(g) = 0.5 g (1 + erf(g / sqrt 2))  (exact, F.gelu default)
__global__ __launch_bounds__(256) void k_geglu(const float* __restrict__ hin, long rows, int I, tfmq_qsel aq,
                                               int8_t* __restrict__ yq, float* __restrict__ yf) {
  const long total = rows * (I / 4);
  const bool quant = aq.qtable != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(aq);
  // (token, channel group) without a 64-bit division per item: 32-bit arithmetic (the launcher checks the item count),
  // advancing by the grid stride = sdiv tokens + smod channel groups with carry
  const unsigned cv = static_cast<unsigned>(I / 4);
  const unsigned stride = gridDim.x * blockDim.x, i0 = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned sdiv = stride / cv, smod = stride - sdiv * cv;
  unsigned mu = i0 / cv, cq = i0 - mu * cv;
  for (unsigned i = i0; i < static_cast<unsigned>(total); i += stride, mu += sdiv, cq += smod) {
    if (cq >= cv) {
      cq -= cv;
      ++mu;
    }
    const long m = mu;
    const int c = static_cast<int>(cq) * 4;
    const float4 a = *reinterpret_cast<const float4*>(hin + m * 2 * I + c);
    const float4 g = *reinterpret_cast<const float4*>(hin + m * 2 * I + I + c);
    float4 y;
    const f2 y01 = f2{a.x, a.y} * gelu2(f2{g.x, g.y}), y23 = f2{a.z, a.w} * gelu2(f2{g.z, g.w});
    y = make_float4(y01.x, y01.y, y23.x, y23.y);
    if (yf) *reinterpret_cast<float4*>(yf + m * I + c) = y;
    if (quant) {
      char4 q;
      q = quant_char4(y.x, y.y, y.z, y.w, make_quantp(qp));
      *reinterpret_cast<char4*>(yq + m * I + c) = q;
    }
    if (i + stride < i) break;      // 32-bit wrap of the item index
  }
}

extern "C" int tfmq_geglu(tfmq_handle h, const float* hin, long rows, int inner, tfmq_qsel aq, int8_t* yq, float* yf,
                          void* stream) {
  TFMQ_CHECK_ARG(h, h && hin && rows > 0 && inner > 0 && inner % 4 == 0, "geglu: bad argument");
  TFMQ_CHECK_ARG(h, (aq.qtable && yq) || yf, "geglu: no output requested");
  const long total = rows * (inner / 4);
  TFMQ_CHECK_ARG(h, total < (1L << 32), "geglu: more than 2^32 items");
  int blocks = ceil_div(total, 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  hipLaunchKernelGGL(k_geglu, dim3(blocks), dim3(256), 0, as_stream(stream), hin, rows, inner, aq, yq, yf);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
