// K9: LayerNorm (+8-bit quantise) and GEGLU (+8-bit quantise) of the SpatialTransformer blocks
// (nn.LayerNorm ldm/modules/attention.py:203-205, GEGLU :37-44; each feeds QuantLayer aqtizers of
// to_q/to_k/to_v, ff.net.0.proj, ff.net.2 -- quant/quant_layer.py:318-325).  HBM-bound token-wise
// kernels: one wave per token row, 16-byte loads, the row lives in registers between the
// statistics and the apply step (one read of fp32, one write of int8).
#include "common.hpp"

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

extern "C" int tfmq_layernorm(tfmq_handle h, const float* x, const float* gamma, const float* beta, float eps, long rows,
                              int C, tfmq_qsel aq, int8_t* yq, float* yf, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && gamma && beta && rows > 0 && C > 0, "layernorm: bad argument");
  TFMQ_CHECK_ARG(h, (aq.qtable && yq) || yf, "layernorm: no output requested");
  TFMQ_CHECK_ARG(h, C % 4 == 0 && C <= 64 * 4 * 8, "layernorm: C must be a multiple of 4 and <= 2048");
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
