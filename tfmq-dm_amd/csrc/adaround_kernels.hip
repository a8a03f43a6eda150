// K12-K14: AdaRound soft rounding forward / backward, rounding regulariser, Adam step and the
// reconstruction loss (quant/adaptive_rounding.py; quant/reconstruction_util.py:50-91;
// quant/quant_layer.py:146-156; torch.optim.Adam defaults, quant/reconstruction.py:42).
// Elementwise over weight-sized tensors, HBM-bound: one fused pass reads w, alpha, m, v, g and
// writes alpha, m, v (32 B/element) instead of the ~20 ATen passes of one reference iteration.
#include "common.hpp"
#include <cstdint>

#define ADA_GAMMA (-0.1f)
#define ADA_ZETA_M_GAMMA (1.2f)  // fp32(1.1 - (-0.1))

__device__ __forceinline__ float sigmoid_f(float a) { return 1.0f / (1.0f + expf(-a)); }

__global__ __launch_bounds__(256) void k_adaround_init(const float* __restrict__ w, const float* __restrict__ delta,
                                                       float* __restrict__ alpha, size_t rows, size_t cols) {
  const size_t n = rows * cols, stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = delta[i / cols];
    const float t = w[i] / d;
    const float rest = t - floorf(t);
    alpha[i] = -logf(ADA_ZETA_M_GAMMA / (rest - ADA_GAMMA) - 1.0f);
  }
}

extern "C" int tfmq_adaround_init(tfmq_handle h, const float* w, const float* delta, float* alpha, size_t rows,
                                  size_t cols, void* stream) {
  TFMQ_CHECK_ARG(h, h && w && delta && alpha && rows > 0 && cols > 0, "adaround_init: bad argument");
  int blocks = ceil_div(static_cast<long>(rows * cols), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_adaround_init, dim3(blocks), dim3(256), 0, as_stream(stream), w, delta, alpha, rows, cols);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

__global__ __launch_bounds__(256) void k_adaround_soft_fwd(const float* __restrict__ w, const float* __restrict__ alpha,
                                                           const float* __restrict__ delta, const float* __restrict__ zp,
                                                           float* __restrict__ w_hat, size_t rows, size_t cols,
                                                           float lmax, int hard) {
  const size_t n = rows * cols, stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t r = i / cols;
    const float d = delta[r], z = zp[r];
    float hsoft = sigmoid_f(alpha[i]) * ADA_ZETA_M_GAMMA + ADA_GAMMA;
    hsoft = fminf(fmaxf(hsoft, 0.0f), 1.0f);
    if (hard) hsoft = alpha[i] >= 0.0f ? 1.0f : 0.0f;  // inference-time rounding (adaptive_rounding.py:63)
    float q = floorf(w[i] / d) + hsoft;
    q = fminf(fmaxf(q + z, 0.0f), lmax);
    w_hat[i] = d * (q - z);
  }
}

// four consecutive columns per thread (see k_adaround_bwd_adam4); per-element arithmetic = the scalar kernel's
__global__ __launch_bounds__(256) void k_adaround_soft_fwd4(const float* __restrict__ w, const float* __restrict__ alpha,
                                                            const float* __restrict__ delta, const float* __restrict__ zp,
                                                            float* __restrict__ w_hat, unsigned rows, unsigned cols4, float lmax, int hard) {
  const unsigned n4 = rows * cols4, stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const unsigned r = i / cols4;
    const float d = delta[r], z = zp[r];
    const float4 a4 = reinterpret_cast<const float4*>(alpha)[i], w4 = reinterpret_cast<const float4*>(w)[i];
    const float av[4] = {a4.x, a4.y, a4.z, a4.w}, wv[4] = {w4.x, w4.y, w4.z, w4.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float hsoft = sigmoid_f(av[e]) * ADA_ZETA_M_GAMMA + ADA_GAMMA;
      hsoft = fminf(fmaxf(hsoft, 0.0f), 1.0f);
      if (hard) hsoft = av[e] >= 0.0f ? 1.0f : 0.0f;
      float q = floorf(wv[e] / d) + hsoft;
      q = fminf(fmaxf(q + z, 0.0f), lmax);
      o[e] = d * (q - z);
    }
    reinterpret_cast<float4*>(w_hat)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int tfmq_adaround_soft_fwd(tfmq_handle h, const float* w, const float* alpha, const float* delta,
                                      const float* zp, float* w_hat, size_t rows, size_t cols, int level, int hard,
                                      void* stream) {
  TFMQ_CHECK_ARG(h, h && w && alpha && delta && zp && w_hat && rows > 0 && cols > 0, "adaround_soft_fwd: bad argument");
  const bool al16 = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(alpha) | reinterpret_cast<uintptr_t>(w_hat)) & 15) == 0;
  if (cols % 4 == 0 && al16 && rows * cols < (1ull << 32)) {
    int blocks4 = ceil_div(static_cast<long>(rows * cols / 4), 256);
    if (blocks4 > h->cu_count * 8) blocks4 = h->cu_count * 8;
    hipLaunchKernelGGL(k_adaround_soft_fwd4, dim3(blocks4), dim3(256), 0, as_stream(stream), w, alpha, delta, zp, w_hat,
                       static_cast<unsigned>(rows), static_cast<unsigned>(cols / 4), static_cast<float>(level - 1), hard);
    TFMQ_LAUNCH_CHECK(h);
    return TFMQ_OK;
  }
  int blocks = ceil_div(static_cast<long>(rows * cols), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_adaround_soft_fwd, dim3(blocks), dim3(256), 0, as_stream(stream), w, alpha, delta, zp, w_hat,
                     rows, cols, static_cast<float>(level - 1), hard);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// backward through w_hat = d*(clamp(floor(w/d) + h(alpha) + z, 0, L-1) - z) and the regulariser
// w_reg*sum(1 - |2h-1|^b), then Adam.  torch.clamp passes the gradient on the closed interval.
__global__ __launch_bounds__(256) void k_adaround_bwd_adam(const float* __restrict__ w, float* __restrict__ alpha,
                                                           const float* __restrict__ delta, const float* __restrict__ zp,
                                                           const float* __restrict__ g_what, float* __restrict__ m,
                                                           float* __restrict__ v, size_t rows, size_t cols, float lmax,
                                                           float w_reg, float b_temp, float step_size, float bc2_sqrt,
                                                           float* __restrict__ round_loss, const float* __restrict__ dyn) {
  const float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
  if (dyn) {      // tfmq_adaround_bwd_adam_dyn: the four per-iteration scalars come from device memory (a captured iteration replays them)
    w_reg = dyn[0]; b_temp = dyn[1]; step_size = dyn[2]; bc2_sqrt = dyn[3];
  }
  const size_t n = rows * cols, stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  float rl = 0.0f;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t r = i / cols;
    const float d = delta[r], z = zp[r];
    const float a = alpha[i];
    const float sg = sigmoid_f(a);
    const float hs = sg * ADA_ZETA_M_GAMMA + ADA_GAMMA;
    const float hsoft = fminf(fmaxf(hs, 0.0f), 1.0f);
    const float qv = floorf(w[i] / d) + hsoft + z;
    float gh = (qv >= 0.0f && qv <= lmax) ? g_what[i] * d : 0.0f;
    if (b_temp > 0.0f) {
      const float c = hsoft - 0.5f;
      const float u = fabsf(c) * 2.0f;
      const float ub1 = powf(u, b_temp - 1.0f);
      rl += 1.0f - ub1 * u;
      const float sgn = c > 0.0f ? 1.0f : (c < 0.0f ? -1.0f : 0.0f);
      gh += -w_reg * b_temp * ub1 * 2.0f * sgn;
    }
    const float g = (hs >= 0.0f && hs <= 1.0f) ? gh * (ADA_ZETA_M_GAMMA * sg * (1.0f - sg)) : 0.0f;
    // Adam (torch single-tensor form)
    const float mi = m[i] + (g - m[i]) * (1.0f - beta1);
    const float vi = v[i] * beta2 + (1.0f - beta2) * g * g;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    alpha[i] = a - step_size * (mi / denom);
  }
  if (round_loss && b_temp > 0.0f) {
    rl = wave_reduce_sum(rl);
    if ((threadIdx.x & 63) == 0) atomicAdd(round_loss, w_reg * rl);
  }
}

// Four consecutive columns per thread (cols % 4 == 0: every conv / Linear of the UNets): 16-byte loads and stores of the six
// streams, one 32-bit row division per item instead of a 64-bit one per element.  Per-element arithmetic = the scalar kernel's.
__global__ __launch_bounds__(256) void k_adaround_bwd_adam4(const float* __restrict__ w, float* __restrict__ alpha,
                                                            const float* __restrict__ delta, const float* __restrict__ zp,
                                                            const float* __restrict__ g_what, float* __restrict__ m,
                                                            float* __restrict__ v, unsigned rows, unsigned cols4, float lmax,
                                                            float w_reg, float b_temp, float step_size, float bc2_sqrt,
                                                            float* __restrict__ round_loss, const float* __restrict__ dyn) {
  const float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
  if (dyn) {
    w_reg = dyn[0]; b_temp = dyn[1]; step_size = dyn[2]; bc2_sqrt = dyn[3];
  }
  const unsigned n4 = rows * cols4, stride = gridDim.x * blockDim.x;
  float rl = 0.0f;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const unsigned r = i / cols4;
    const float d = delta[r], z = zp[r];
    const float4 a4 = reinterpret_cast<const float4*>(alpha)[i], w4 = reinterpret_cast<const float4*>(w)[i];
    const float4 g4 = reinterpret_cast<const float4*>(g_what)[i];
    const float4 m4 = reinterpret_cast<const float4*>(m)[i], v4 = reinterpret_cast<const float4*>(v)[i];
    const float av[4] = {a4.x, a4.y, a4.z, a4.w}, wv[4] = {w4.x, w4.y, w4.z, w4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
    const float mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
    float ao[4], mo[4], vo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = av[e];
      const float sg = sigmoid_f(a);
      const float hs = sg * ADA_ZETA_M_GAMMA + ADA_GAMMA;
      const float hsoft = fminf(fmaxf(hs, 0.0f), 1.0f);
      const float qv = floorf(wv[e] / d) + hsoft + z;
      float gh = (qv >= 0.0f && qv <= lmax) ? gv[e] * d : 0.0f;
      if (b_temp > 0.0f) {
        const float c = hsoft - 0.5f;
        const float u = fabsf(c) * 2.0f;
        const float ub1 = powf(u, b_temp - 1.0f);
        rl += 1.0f - ub1 * u;
        const float sgn = c > 0.0f ? 1.0f : (c < 0.0f ? -1.0f : 0.0f);
        gh += -w_reg * b_temp * ub1 * 2.0f * sgn;
      }
      const float g = (hs >= 0.0f && hs <= 1.0f) ? gh * (ADA_ZETA_M_GAMMA * sg * (1.0f - sg)) : 0.0f;
      const float mi = mv[e] + (g - mv[e]) * (1.0f - beta1);
      const float vi = vv[e] * beta2 + (1.0f - beta2) * g * g;
      mo[e] = mi;
      vo[e] = vi;
      const float denom = sqrtf(vi) / bc2_sqrt + eps;
      ao[e] = a - step_size * (mi / denom);
    }
    reinterpret_cast<float4*>(m)[i] = make_float4(mo[0], mo[1], mo[2], mo[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vo[0], vo[1], vo[2], vo[3]);
    reinterpret_cast<float4*>(alpha)[i] = make_float4(ao[0], ao[1], ao[2], ao[3]);
  }
  if (round_loss && b_temp > 0.0f) {      // one atomic per block: 8192 wave atomics on the one address cost ~100 us per launch
    __shared__ float part[4];
    rl = wave_reduce_sum(rl);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = rl;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(round_loss, w_reg * ((part[0] + part[1]) + (part[2] + part[3])));
  }
}

// {w_reg, b_temp, lr / (1 - 0.9^t), sqrt(1 - 0.999^t)}: the per-iteration scalars of the fused kernel, computed in ONE place for both entry points
static void adam_scalars(float w_reg, float b_temp, float lr, int t, float* out4) {
  const double bc1 = 1.0 - pow(0.9, t), bc2 = 1.0 - pow(0.999, t);
  out4[0] = w_reg;
  out4[1] = b_temp;
  out4[2] = static_cast<float>(lr / bc1);
  out4[3] = static_cast<float>(sqrt(bc2));
}

static int launch_bwd_adam(tfmq_handle h, const float* w, float* alpha, const float* delta, const float* zp, const float* g_what, float* m,
                           float* v, size_t rows, size_t cols, int level, const float* s4, const float* dyn, float* round_loss, void* stream) {
  const bool al16 = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(alpha) | reinterpret_cast<uintptr_t>(g_what) |
                      reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  if (cols % 4 == 0 && al16 && rows * cols < (1ull << 32)) {
    int blocks4 = ceil_div(static_cast<long>(rows * cols / 4), 256);
    if (blocks4 > h->cu_count * 4) blocks4 = h->cu_count * 4;
    hipLaunchKernelGGL(k_adaround_bwd_adam4, dim3(blocks4), dim3(256), 0, as_stream(stream), w, alpha, delta, zp, g_what, m, v,
                       static_cast<unsigned>(rows), static_cast<unsigned>(cols / 4), static_cast<float>(level - 1), s4[0], s4[1], s4[2], s4[3],
                       round_loss, dyn);
    TFMQ_LAUNCH_CHECK(h);
    return TFMQ_OK;
  }
  int blocks = ceil_div(static_cast<long>(rows * cols), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_adaround_bwd_adam, dim3(blocks), dim3(256), 0, as_stream(stream), w, alpha, delta, zp, g_what, m,
                     v, rows, cols, static_cast<float>(level - 1), s4[0], s4[1], s4[2], s4[3], round_loss, dyn);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_adaround_bwd_adam(tfmq_handle h, const float* w, float* alpha, const float* delta, const float* zp,
                                      const float* g_what, float* m, float* v, size_t rows, size_t cols, int level,
                                      float w_reg, float b_temp, float lr, int t, float* round_loss, void* stream) {
  TFMQ_CHECK_ARG(h, h && w && alpha && delta && zp && g_what && m && v && rows > 0 && cols > 0 && t >= 1,
                 "adaround_bwd_adam: bad argument");
  float s4[4];
  adam_scalars(w_reg, b_temp, lr, t, s4);
  return launch_bwd_adam(h, w, alpha, delta, zp, g_what, m, v, rows, cols, level, s4, nullptr, round_loss, stream);
}

// Round 5: the same launch with its four per-iteration scalars read from DEVICE memory, so that a reconstruction iteration can be captured
// once as a hipGraph and replayed: the host writes scalars[4] = tfmq_adaround_scalars(w_reg, b_temp, lr, t) (a 16-byte copy on the stream)
// in front of each replay.  Same kernels, same arithmetic: equal to tfmq_adaround_bwd_adam bit for bit.
extern "C" int tfmq_adaround_scalars(float w_reg, float b_temp, float lr, int t, float* out4) {
  if (!out4 || t < 1) return TFMQ_ERR_ARG;
  adam_scalars(w_reg, b_temp, lr, t, out4);
  return TFMQ_OK;
}

extern "C" int tfmq_adaround_bwd_adam_dyn(tfmq_handle h, const float* w, float* alpha, const float* delta, const float* zp,
                                          const float* g_what, float* m, float* v, size_t rows, size_t cols, int level,
                                          const float* scalars_dev, float* round_loss, void* stream) {
  TFMQ_CHECK_ARG(h, h && w && alpha && delta && zp && g_what && m && v && rows > 0 && cols > 0 && scalars_dev,
                 "adaround_bwd_adam_dyn: bad argument");
  const float s4[4] = {0.0f, 0.0f, 0.0f, 1.0f};
  return launch_bwd_adam(h, w, alpha, delta, zp, g_what, m, v, rows, cols, level, s4, scalars_dev, round_loss, stream);
}

// loss = sum |pred - tgt|^2 / denom  (lp_loss p=2: sum over dim 1, mean over the rest => denom =
// numel / size(1)); g = 2 (pred - tgt) / denom
__global__ __launch_bounds__(256) void k_recon_loss(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                    float* __restrict__ g, size_t n, float inv_denom,
                                                    float* __restrict__ loss) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  double acc = 0.0;
  const bool v4 = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(tgt) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
  if (v4) {                               // 16-byte items; per element the same operations in the same order within a thread
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n / 4; i += stride) {
      const float4 p4 = reinterpret_cast<const float4*>(pred)[i], t4 = reinterpret_cast<const float4*>(tgt)[i];
      const float dl[4] = {p4.x - t4.x, p4.y - t4.y, p4.z - t4.z, p4.w - t4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += static_cast<double>(dl[e]) * dl[e];
      if (g) reinterpret_cast<float4*>(g)[i] = make_float4(2.0f * dl[0] * inv_denom, 2.0f * dl[1] * inv_denom, 2.0f * dl[2] * inv_denom, 2.0f * dl[3] * inv_denom);
    }
  } else {
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
      const float dlt = pred[i] - tgt[i];
      acc += static_cast<double>(dlt) * dlt;
      if (g) g[i] = 2.0f * dlt * inv_denom;
    }
  }
  // one atomic per block (the wave atomics of a full grid on the one address cost more than the pass itself)
  __shared__ double part[4];
  acc = wave_reduce_sum_d(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, static_cast<float>(((part[0] + part[1]) + (part[2] + part[3])) * inv_denom));
}

extern "C" int tfmq_recon_loss(tfmq_handle h, const float* pred, const float* tgt, float* g, size_t n, size_t denom,
                               float* loss, void* stream) {
  TFMQ_CHECK_ARG(h, h && pred && tgt && loss && n > 0 && denom > 0, "recon_loss: bad argument");
  int blocks = ceil_div(static_cast<long>(n), 1024);
  if (blocks > h->cu_count * 4) blocks = h->cu_count * 4;
  hipLaunchKernelGGL(k_recon_loss, dim3(blocks), dim3(256), 0, as_stream(stream), pred, tgt, g, n,
                     static_cast<float>(1.0 / static_cast<double>(denom)), loss);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
