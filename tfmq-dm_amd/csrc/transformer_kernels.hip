// K9: LayerNorm (+8-bit quantise) and GEGLU (+8-bit quantise) of the SpatialTransformer blocks
// (nn.LayerNorm ldm/modules/attention.py:203-205, GEGLU :37-44; each feeds QuantLayer aqtizers of
// to_q/to_k/to_v, ff.net.0.proj, ff.net.2 -- quant/quant_layer.py:318-325).  HBM-bound token-wise
// kernels: one wave per token row, 16-byte loads, the row lives in registers between the
// statistics and the apply step (one read of fp32, one write of int8).
#include "common.hpp"

// rows of up to 64*4*MAXV floats (MAXV float4 per lane): 5 -> C <= 1280
template <int MAXV, bool XH = false>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, long rows, int Cc,
                                                   tfmq_qsel aq, int8_t* __restrict__ yq, float* __restrict__ yf) {
  const long row = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int c4 = Cc / 4;
  const float4* xr = reinterpret_cast<const float4*>(x + row * Cc);
  const uint2* xh = reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(x) + row * Cc);   // XH: fp16 row
  float4 v[MAXV];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if constexpr (XH) {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < c4) {
        const uint2 u = xh[idx];
        const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&u.x)), hi = __half22float2(*reinterpret_cast<const __half2*>(&u.y));
        v[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
      }
    } else {
      v[i] = idx < c4 ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  s = wave_reduce_sum(s);
  const float mean = s / static_cast<float>(Cc);
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (lane + i * 64 < c4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss += (a * a + b * b) + (c * c + d * d);
    }
  }
  ss = wave_reduce_sum(ss);
  const float rstd = 1.0f / sqrtf(ss / static_cast<float>(Cc) + eps);
  const bool quant = aq.qtable != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(aq);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx >= c4) continue;
    const float4 g = reinterpret_cast<const float4*>(gamma)[idx];
    const float4 b = reinterpret_cast<const float4*>(beta)[idx];
    float4 y;
    y.x = (v[i].x - mean) * rstd * g.x + b.x;
    y.y = (v[i].y - mean) * rstd * g.y + b.y;
    y.z = (v[i].z - mean) * rstd * g.z + b.z;
    y.w = (v[i].w - mean) * rstd * g.w + b.w;
    if (yf) reinterpret_cast<float4*>(yf + row * Cc)[idx] = y;
    if (quant) {
      char4 q;
      q = quant_char4(y.x, y.y, y.z, y.w, make_quantp(qp));
      reinterpret_cast<char4*>(yq + row * Cc)[idx] = q;
    }
  }
}

extern "C" int tfmq_layernorm(tfmq_handle h, const float* x, const float* gamma, const float* beta, float eps, long rows,
                              int C, tfmq_qsel aq, int8_t* yq, float* yf, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && gamma && beta && rows > 0 && C > 0, "layernorm: bad argument");
  TFMQ_CHECK_ARG(h, (aq.qtable && yq) || yf, "layernorm: no output requested");
  TFMQ_CHECK_ARG(h, C % 4 == 0 && C <= 64 * 4 * 8, "layernorm: C must be a multiple of 4 and <= 2048");
  dim3 grid(static_cast<unsigned>((rows + 3) / 4));
  if (C <= 64 * 4 * 2) hipLaunchKernelGGL(k_layernorm<2>, grid, dim3(256), 0, as_stream(stream), x, gamma, beta, eps, rows, C, aq, yq, yf);
  else if (C <= 64 * 4 * 5) hipLaunchKernelGGL(k_layernorm<5>, grid, dim3(256), 0, as_stream(stream), x, gamma, beta, eps, rows, C, aq, yq, yf);
  else hipLaunchKernelGGL(k_layernorm<8>, grid, dim3(256), 0, as_stream(stream), x, gamma, beta, eps, rows, C, aq, yq, yf);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_layernorm_h(tfmq_handle h, const uint16_t* x, const float* gamma, const float* beta, float eps, long rows,
                                int C, tfmq_qsel aq, int8_t* yq, float* yf, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && gamma && beta && rows > 0 && C > 0, "layernorm_h: bad argument");
  TFMQ_CHECK_ARG(h, (aq.qtable && yq) || yf, "layernorm_h: no output requested");
  TFMQ_CHECK_ARG(h, C % 4 == 0 && C <= 64 * 4 * 8, "layernorm_h: C must be a multiple of 4 and <= 2048");
  dim3 grid(static_cast<unsigned>((rows + 3) / 4));
  const float* xf = reinterpret_cast<const float*>(x);
  if (C <= 64 * 4 * 2) hipLaunchKernelGGL((k_layernorm<2, true>), grid, dim3(256), 0, as_stream(stream), xf, gamma, beta, eps, rows, C, aq, yq, yf);
  else if (C <= 64 * 4 * 5) hipLaunchKernelGGL((k_layernorm<5, true>), grid, dim3(256), 0, as_stream(stream), xf, gamma, beta, eps, rows, C, aq, yq, yf);
  else hipLaunchKernelGGL((k_layernorm<8, true>), grid, dim3(256), 0, as_stream(stream), xf, gamma, beta, eps, rows, C, aq, yq, yf);
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
