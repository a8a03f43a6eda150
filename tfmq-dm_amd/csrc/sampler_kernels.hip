// K7 (temporal-information block GEMVs) and K11 (sampler elementwise, layout transforms).
#include "common.hpp"

// ------------------------------------------------------------------------------ K11
// coef row of the current step: {sqrt(1-a_t), sqrt(a_t), sqrt(a_next), c1, c2, 0, 0, 0}
// (all fp32, computed on the host exactly as ddim/functions/denoising.py:21-22,31-37 does).
//   x0 = (x - eps*sqrt(1-a_t)) / sqrt(a_t);  x_next = sqrt(a_next)*x0 + c1*z + c2*eps
__global__ __launch_bounds__(256) void k_ddim_update(const float* __restrict__ x, const float* __restrict__ eps,
                                                     const float* __restrict__ z, float* __restrict__ xn,
                                                     float* __restrict__ x0o, size_t n, const float* __restrict__ coef,
                                                     const int32_t* __restrict__ step) {
  const float* c = coef + static_cast<size_t>(step ? *step : 0) * 8;
  const float s1m = c[0], sa = c[1], san = c[2], c1 = c[3], c2 = c[4];
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float e = eps[i];
    const float x0 = (x[i] - e * s1m) / sa;
    if (x0o) x0o[i] = x0;
    float r = san * x0;
    r = r + c1 * (z ? z[i] : 0.0f);   // the reference adds c1*randn even when c1 == 0 (eta = 0)
    r = r + c2 * e;
    xn[i] = r;
  }
}

extern "C" int tfmq_ddim_update(tfmq_handle h, const float* x, const float* eps, const float* noise, float* x_next,
                                float* x0, size_t n, const float* coef, const int32_t* step, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && eps && x_next && coef, "ddim_update: null pointer");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_ddim_update, dim3(blocks), dim3(256), 0, as_stream(stream), x, eps, noise, x_next, x0, n, coef,
                     step);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// latent DDIM step with classifier-free guidance (p_sample_ddim, ldm/models/diffusion/ddim.py:181-211):
//   e = e_u + s (e_c - e_u);  x0 = (x - sqrt(1-a_t) e)/sqrt(a_t);  x' = sqrt(a_prev) x0 + sqrt(1-a_prev-sig^2) e + sig z
// (same coef row layout as k_ddim_update; note the reference adds dir_xt before the noise here)
__global__ __launch_bounds__(256) void k_ddim_update_cfg(const float* __restrict__ x, const float* __restrict__ eps_u,
                                                         const float* __restrict__ eps_c, float scale,
                                                         const float* __restrict__ z, float* __restrict__ xn,
                                                         float* __restrict__ x0o, size_t n,
                                                         const float* __restrict__ coef, const int32_t* __restrict__ step) {
  const float* c = coef + static_cast<size_t>(step ? *step : 0) * 8;
  const float s1m = c[0], sa = c[1], san = c[2], c1 = c[3], c2 = c[4];
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float eu = eps_u[i];
    const float e = eu + scale * (eps_c[i] - eu);
    const float x0 = (x[i] - s1m * e) / sa;
    if (x0o) x0o[i] = x0;
    float r = san * x0;
    r = r + c2 * e;
    r = r + c1 * (z ? z[i] : 0.0f);
    xn[i] = r;
  }
}

extern "C" int tfmq_ddim_update_cfg(tfmq_handle h, const float* x, const float* eps_u, const float* eps_c, float scale,
                                    const float* noise, float* x_next, float* x0, size_t n, const float* coef,
                                    const int32_t* step, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && eps_u && eps_c && x_next && coef, "ddim_update_cfg: null pointer");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_ddim_update_cfg, dim3(blocks), dim3(256), 0, as_stream(stream), x, eps_u, eps_c, scale, noise, x_next,
                     x0, n, coef, step);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

__global__ void k_step_advance(int32_t* step, int delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *step += delta;
}
extern "C" int tfmq_step_advance(tfmq_handle h, int32_t* step, int delta, void* stream) {
  TFMQ_CHECK_ARG(h, h && step, "step_advance: null pointer");
  hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(64), 0, as_stream(stream), step, delta);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// [B][C][HW] <-> [B][HW][C] through a padded LDS tile (coalesced on both sides)
template <bool TO_NHWC>
__global__ __launch_bounds__(256) void k_layout(const float* __restrict__ x, float* __restrict__ y, int C, int HW) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xb = x + static_cast<size_t>(b) * C * HW;
  float* yb = y + static_cast<size_t>(b) * C * HW;
  if (TO_NHWC) {
    for (int j = ty; j < 32; j += 8) {
      const int c = c0 + j, px = p0 + tx;
      if (c < C && px < HW) tile[j][tx] = xb[static_cast<size_t>(c) * HW + px];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
      const int px = p0 + j, c = c0 + tx;
      if (c < C && px < HW) yb[static_cast<size_t>(px) * C + c] = tile[tx][j];
    }
  } else {
    for (int j = ty; j < 32; j += 8) {
      const int px = p0 + j, c = c0 + tx;
      if (c < C && px < HW) tile[j][tx] = xb[static_cast<size_t>(px) * C + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
      const int c = c0 + j, px = p0 + tx;
      if (c < C && px < HW) yb[static_cast<size_t>(c) * HW + px] = tile[tx][j];
    }
  }
}

extern "C" int tfmq_nchw_to_nhwc(tfmq_handle h, const float* x, float* y, int B, int C, int HW, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && y && B > 0 && C > 0 && HW > 0 && B < 65536, "nchw_to_nhwc: bad argument");
  hipLaunchKernelGGL(k_layout<true>, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, as_stream(stream), x, y, C, HW);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
extern "C" int tfmq_nhwc_to_nchw(tfmq_handle h, const float* x, float* y, int B, int C, int HW, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && y && B > 0 && C > 0 && HW > 0 && B < 65536, "nhwc_to_nchw: bad argument");
  hipLaunchKernelGGL(k_layout<false>, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, as_stream(stream), x, y, C, HW);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// y = x*sigmoid(x) (nonlinearity, ddim/models/diffusion.py:27-29): only used standalone when the
// activation calibration has to observe the SiLU'd embedding before it is quantised.
__global__ __launch_bounds__(256) void k_silu(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = silu_f(x[i]);
}
extern "C" int tfmq_silu(tfmq_handle h, const float* x, float* y, size_t n, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && y, "silu: null pointer");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_silu, dim3(blocks), dim3(256), 0, as_stream(stream), x, y, n);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------------------ K7
__global__ void k_timestep_embedding(const float* __restrict__ t, int m, int dim, int ldm_order, float* __restrict__ emb) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * half) return;
  const int r = i / half, j = i - r * half;
  // freq_j = exp(j * -(ln 1e4 / denom)) in fp32 like the reference
  // (ddim/models/diffusion.py:16-18: denom = half-1; ldm util.py:161-163: denom = half)
  // The fp32 argument of exp() is formed exactly as the reference does; exp itself is evaluated
  // in double and rounded once (correctly rounded fp32), so it can differ from the CPU libm of
  // the reference by at most 1 ulp of the frequency.
  float f;
  if (ldm_order) {
    const float arg = static_cast<float>(-log(10000.0)) * static_cast<float>(j) / static_cast<float>(half);
    f = static_cast<float>(exp(static_cast<double>(arg)));
  } else {
    const float e = static_cast<float>(log(10000.0) / static_cast<double>(half - 1));
    f = static_cast<float>(exp(static_cast<double>(static_cast<float>(j) * -e)));
  }
  const float a = t[r] * f;
  float* o = emb + static_cast<size_t>(r) * dim;
  if (ldm_order) {
    o[j] = cosf(a);
    o[half + j] = sinf(a);
  } else {
    o[j] = sinf(a);
    o[half + j] = cosf(a);
  }
  if ((dim & 1) && j == 0) o[dim - 1] = 0.0f;
}

extern "C" int tfmq_timestep_embedding(tfmq_handle h, const float* t, int m, int dim, int ldm_order, float* emb,
                                       void* stream) {
  TFMQ_CHECK_ARG(h, h && t && emb && m > 0 && dim >= 4, "timestep_embedding: bad argument");
  hipLaunchKernelGGL(k_timestep_embedding, dim3(ceil_div(static_cast<long>(m) * (dim / 2), 256)), dim3(256), 0,
                     as_stream(stream), t, m, dim, ldm_order, emb);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// y[m][n] = act(x[m][:]) . W[n][:] + b[n]; one wave per output column n, all (<= 8) rows of a
// row-block at once: the weight row is streamed exactly once (GEMV: weight-bandwidth bound).
#define LS_ROWS 8
__global__ __launch_bounds__(256) void k_linear_small_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int m,
                                                          int n, int k, int silu_in) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n) return;
  const int r0 = blockIdx.y * LS_ROWS;
  float acc[LS_ROWS];
#pragma unroll
  for (int r = 0; r < LS_ROWS; ++r) acc[r] = 0.0f;
  const float* wr = w + static_cast<size_t>(wave) * k;
  for (int j = lane; j < k; j += 64) {
    const float wv = wr[j];
#pragma unroll
    for (int r = 0; r < LS_ROWS; ++r) {
      if (r0 + r < m) {
        float xv = x[static_cast<size_t>(r0 + r) * k + j];
        if (silu_in) xv = silu_f(xv);
        acc[r] += xv * wv;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < LS_ROWS; ++r) {
    const float s = wave_reduce_sum(acc[r]);
    if (lane == 0 && r0 + r < m) y[static_cast<size_t>(r0 + r) * n + wave] = s + (bias ? bias[wave] : 0.0f);
  }
}

extern "C" int tfmq_linear_small_f32(tfmq_handle h, const float* x, const float* w, const float* bias, float* y, int m,
                                     int n, int k, int silu_in, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && w && y && m > 0 && n > 0 && k > 0, "linear_small_f32: bad argument");
  hipLaunchKernelGGL(k_linear_small_f32, dim3(ceil_div(n, 4), ceil_div(m, LS_ROWS)), dim3(256), 0, as_stream(stream), x,
                     w, bias, y, m, n, k, silu_in);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// int4-weight variant: y = da*dw[n] * sum (qa - za)(qw - zw) + b  or, without an activation
// quantiser (weight-only layer, quant_model.py:110-120), y = dw[n] * sum x*(qw - zw) + b.
__global__ __launch_bounds__(256) void k_linear_small_w4(const float* __restrict__ x, const uint32_t* __restrict__ wp,
                                                         const int32_t* __restrict__ wmeta,
                                                         const float* __restrict__ wscale, const float* __restrict__ bias,
                                                         tfmq_qsel aq, float* __restrict__ y, int m, int n, int k,
                                                         int silu_in) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n) return;
  const int r0 = blockIdx.y * LS_ROWS;
  const bool quant = aq.qtable != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(aq);
  const int zw = wmeta[wave * 4];
  float facc[LS_ROWS];
  int iacc[LS_ROWS];
#pragma unroll
  for (int r = 0; r < LS_ROWS; ++r) {
    facc[r] = 0.0f;
    iacc[r] = 0;
  }
  const int ck = w4_ck(k);
  for (int g = lane; g < k / 8; g += 64) {
    const uint32_t word = wp[w4_word_index(wave, g, k, ck)];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int qw = static_cast<int>((word >> ((i & 3) * 8 + (i >> 2) * 4)) & 15u) - zw;
#pragma unroll
      for (int r = 0; r < LS_ROWS; ++r) {
        if (r0 + r < m) {
          float xv = x[static_cast<size_t>(r0 + r) * k + g * 8 + i];
          if (silu_in) xv = silu_f(xv);
          if (quant) {
            const int qa = static_cast<int>(quant_index_f(xv, qp.x, qp.y, 255.0f)) - static_cast<int>(qp.y);
            iacc[r] += qa * qw;
          } else {
            facc[r] += xv * static_cast<float>(qw);
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < LS_ROWS; ++r) {
    float s;
    if (quant) {
      int t = iacc[r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      s = (qp.x * wscale[wave]) * static_cast<float>(t);
    } else {
      s = wscale[wave] * wave_reduce_sum(facc[r]);
    }
    if (lane == 0 && r0 + r < m) y[static_cast<size_t>(r0 + r) * n + wave] = s + (bias ? bias[wave] : 0.0f);
  }
}

extern "C" int tfmq_linear_small_w4(tfmq_handle h, const float* x, const uint8_t* wpacked, const int32_t* wmeta,
                                    const float* wscale, const float* bias, tfmq_qsel aq, float* y, int m, int n, int k,
                                    int silu_in, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && wpacked && wmeta && wscale && y && m > 0 && n > 0 && k > 0 && k % 8 == 0,
                 "linear_small_w4: bad argument");
  hipLaunchKernelGGL(k_linear_small_w4, dim3(ceil_div(n, 4), ceil_div(m, LS_ROWS)), dim3(256), 0, as_stream(stream), x,
                     reinterpret_cast<const uint32_t*>(wpacked), wmeta, wscale, bias, aq, y, m, n, k, silu_in);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ PLMS (Adams-Bashforth) pieces
// e = e_u + s (e_c - e_u)   (get_model_output, ldm/models/diffusion/plms.py:186-195; same operation order)
__global__ __launch_bounds__(256) void k_cfg_combine(const float* __restrict__ eu, const float* __restrict__ ec, float s,
                                                     float* __restrict__ out, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = eu[i] + s * (ec[i] - eu[i]);
}

extern "C" int tfmq_cfg_combine(tfmq_handle h, const float* eps_u, const float* eps_c, float scale, float* out, size_t n,
                                void* stream) {
  TFMQ_CHECK_ARG(h, h && eps_u && eps_c && out, "cfg_combine: null pointer");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_cfg_combine, dim3(blocks), dim3(256), 0, as_stream(stream), eps_u, eps_c, scale, out, n);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// e' of p_sample_plms (plms.py:224-240), evaluated in the reference's operation order:
//   order 1: (e0 + e1) / 2                    e1 = model output at t_next (pseudo improved Euler)
//   order 2: (3 e0 - e1) / 2                  e1, e2, e3 = previous outputs, newest first
//   order 3: (23 e0 - 16 e1 + 5 e2) / 12
//   order 4: (55 e0 - 59 e1 + 37 e2 - 9 e3) / 24
__global__ __launch_bounds__(256) void k_plms_combine(int order, const float* __restrict__ e0, const float* __restrict__ e1,
                                                      const float* __restrict__ e2, const float* __restrict__ e3,
                                                      float* __restrict__ out, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v;
    if (order == 1) v = (e0[i] + e1[i]) / 2.0f;
    else if (order == 2) v = (3.0f * e0[i] - e1[i]) / 2.0f;
    else if (order == 3) v = (23.0f * e0[i] - 16.0f * e1[i] + 5.0f * e2[i]) / 12.0f;
    else v = (55.0f * e0[i] - 59.0f * e1[i] + 37.0f * e2[i] - 9.0f * e3[i]) / 24.0f;
    out[i] = v;
  }
}

extern "C" int tfmq_plms_combine(tfmq_handle h, int order, const float* e0, const float* e1, const float* e2,
                                 const float* e3, float* out, size_t n, void* stream) {
  TFMQ_CHECK_ARG(h, h && e0 && out && order >= 1 && order <= 4, "plms_combine: bad argument");
  TFMQ_CHECK_ARG(h, e1 && (order < 3 || e2) && (order < 4 || e3), "plms_combine: missing history");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_plms_combine, dim3(blocks), dim3(256), 0, as_stream(stream), order, e0, e1, e2, e3, out, n);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ fp32 -> fp16 (round to nearest even)
__global__ __launch_bounds__(256) void k_f32_to_f16(const float* __restrict__ x, __half* __restrict__ y, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  const size_t n4 = n / 4;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const __half2 lo = __floats2half2_rn(v.x, v.y), hi = __floats2half2_rn(v.z, v.w);
    uint2 u;
    u.x = *reinterpret_cast<const unsigned*>(&lo);
    u.y = *reinterpret_cast<const unsigned*>(&hi);
    reinterpret_cast<uint2*>(y)[i] = u;
  }
  for (size_t i = n4 * 4 + static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = __float2half_rn(x[i]);
}

// y[b][t][c] = x[b][t][c] + r[b][c], eight channels per thread (tfmq_row_broadcast_add)
template <bool XH>
__global__ __launch_bounds__(256) void k_row_broadcast_add(const void* __restrict__ xv, const float* __restrict__ r, unsigned total8,
                                                           unsigned tc8, unsigned c8, void* __restrict__ yv) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += gridDim.x * blockDim.x) {
    const unsigned b = i / tc8, c = (i % c8) * 8;
    const float4 r0 = *reinterpret_cast<const float4*>(r + static_cast<size_t>(b) * c8 * 8 + c);
    const float4 r1 = *reinterpret_cast<const float4*>(r + static_cast<size_t>(b) * c8 * 8 + c + 4);
    const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    if constexpr (XH) {
      const uint4 u = reinterpret_cast<const uint4*>(xv)[i];
      const __half2* hp = reinterpret_cast<const __half2*>(&u);
      __half2 o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float2 f = __half22float2(hp[e]);
        o[e] = __floats2half2_rn(f.x + rr[2 * e], f.y + rr[2 * e + 1]);
      }
      reinterpret_cast<uint4*>(yv)[i] = *reinterpret_cast<const uint4*>(o);
    } else {
      const float4 a = reinterpret_cast<const float4*>(xv)[2 * static_cast<size_t>(i)], bq = reinterpret_cast<const float4*>(xv)[2 * static_cast<size_t>(i) + 1];
      reinterpret_cast<float4*>(yv)[2 * static_cast<size_t>(i)] = make_float4(a.x + rr[0], a.y + rr[1], a.z + rr[2], a.w + rr[3]);
      reinterpret_cast<float4*>(yv)[2 * static_cast<size_t>(i) + 1] = make_float4(bq.x + rr[4], bq.y + rr[5], bq.z + rr[6], bq.w + rr[7]);
    }
  }
}

extern "C" int tfmq_row_broadcast_add(tfmq_handle h, const void* x, const float* r, int B, long T, int C, int x_f16, void* y, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && r && y && B > 0 && T > 0 && C > 0 && C % 8 == 0, "row_broadcast_add: bad argument");
  const long total8 = static_cast<long>(B) * T * (C / 8);
  TFMQ_CHECK_ARG(h, total8 < (1L << 32), "row_broadcast_add: more than 2^32 items");
  int blocks = ceil_div(total8, 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  const unsigned c8 = static_cast<unsigned>(C / 8), tc8 = static_cast<unsigned>(T * c8);
  if (x_f16) hipLaunchKernelGGL(k_row_broadcast_add<true>, dim3(blocks), dim3(256), 0, as_stream(stream), x, r, static_cast<unsigned>(total8), tc8, c8, y);
  else hipLaunchKernelGGL(k_row_broadcast_add<false>, dim3(blocks), dim3(256), 0, as_stream(stream), x, r, static_cast<unsigned>(total8), tc8, c8, y);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_f32_to_f16(tfmq_handle h, const float* x, uint16_t* y, size_t n, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && y, "f32_to_f16: null pointer");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n / 4 + 1), 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  hipLaunchKernelGGL(k_f32_to_f16, dim3(blocks), dim3(256), 0, as_stream(stream), x, reinterpret_cast<__half*>(y), n);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ DPM-Solver++ (multistep, data prediction) pieces
// x0 = (x - sigma_t * eps) / alpha_t          (DPM_Solver.data_prediction_fn, dpm_solver.py:386-399)
__global__ __launch_bounds__(256) void k_dpm_x0(const float* __restrict__ x, const float* __restrict__ eps, float sigma,
                                                float alpha, float* __restrict__ out, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (x[i] - sigma * eps[i]) / alpha;
}

extern "C" int tfmq_dpm_x0(tfmq_handle h, const float* x, const float* eps, float sigma, float alpha, float* out, size_t n,
                           void* stream) {
  TFMQ_CHECK_ARG(h, h && x && eps && out, "dpm_x0: null pointer");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_dpm_x0, dim3(blocks), dim3(256), 0, as_stream(stream), x, eps, sigma, alpha, out, n);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// order 1: x_t = c_x x - c_m m0                                      (dpm_solver_first_update, :504-549)
// order 2: x_t = c_x x - c_m m0 - c_d (inv_r0 (m0 - m1))            (multistep_dpm_solver_second_update, :755-810)
// with c_x = sigma_t/sigma_s, c_m = alpha_t (e^{-h} - 1), c_d = 0.5 c_m, m0 / m1 = newest / previous data prediction
__global__ __launch_bounds__(256) void k_dpm_update(int order, const float* __restrict__ x, const float* __restrict__ m0,
                                                    const float* __restrict__ m1, float c_x, float c_m, float c_d, float inv_r0,
                                                    float* __restrict__ out, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v = c_x * x[i] - c_m * m0[i];
    if (order == 2) v = v - c_d * (inv_r0 * (m0[i] - m1[i]));
    out[i] = v;
  }
}

extern "C" int tfmq_dpm_update(tfmq_handle h, int order, const float* x, const float* m0, const float* m1_or_null, float c_x,
                               float c_m, float c_d, float inv_r0, float* out, size_t n, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && m0 && out && (order == 1 || (order == 2 && m1_or_null)), "dpm_update: bad argument");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_dpm_update, dim3(blocks), dim3(256), 0, as_stream(stream), order, x, m0, m1_or_null, c_x, c_m, c_d,
                     inv_r0, out, n);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
