// K1-K4: activation quantiser, min/max statistics, MSE scale search, int4 weight packing.
// HBM-bound scan/reduce kernels: 16-byte loads, wave shuffles for the reductions, one pass
// over the data per statistic (the reference makes 6-8 elementwise passes per call,
// quant/quant_layer.py:20-64,223-244).
#include "common.hpp"

// ------------------------------------------------------------------------------ K1
__global__ __launch_bounds__(256) void k_quantize_act(const float* __restrict__ x, int8_t* __restrict__ q, size_t n,
                                                      tfmq_qsel qs, float lmax) {
  const float2 p = load_qparam(qs);
  const size_t n4 = n >> 2;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  char4* q4 = reinterpret_cast<char4*>(q);
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = x4[i];
    char4 o;
    o.x = static_cast<signed char>(static_cast<int>(quant_index_f(v.x, p.x, p.y, lmax)) - 128);
    o.y = static_cast<signed char>(static_cast<int>(quant_index_f(v.y, p.x, p.y, lmax)) - 128);
    o.z = static_cast<signed char>(static_cast<int>(quant_index_f(v.z, p.x, p.y, lmax)) - 128);
    o.w = static_cast<signed char>(static_cast<int>(quant_index_f(v.w, p.x, p.y, lmax)) - 128);
    q4[i] = o;
  }
  // tail
  if (blockIdx.x == 0) {
    for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x)
      q[i] = static_cast<int8_t>(static_cast<int>(quant_index_f(x[i], p.x, p.y, lmax)) - 128);
  }
}

extern "C" int tfmq_quantize_act(tfmq_handle h, const float* x, int8_t* q, size_t n, tfmq_qsel qs, int level,
                                 void* stream) {
  TFMQ_CHECK_ARG(h, h && x && q && qs.qtable, "quantize_act: null pointer");
  TFMQ_CHECK_ARG(h, level >= 2 && level <= 256, "quantize_act: level must be in [2,256]");
  TFMQ_CHECK_ARG(h, (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(q) & 3) == 0,
                 "quantize_act: x must be 16-byte and q 4-byte aligned");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>((n + 3) / 4), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_quantize_act, dim3(blocks), dim3(256), 0, as_stream(stream), x, q, n, qs,
                     static_cast<float>(level - 1));
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

__global__ __launch_bounds__(256) void k_fake_quant_sel(const float* __restrict__ x, float* __restrict__ y, size_t n, tfmq_qsel qs,
                                                        float lmax, float pre) {
  const float2 p = load_qparam(qs);
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = x[i] * pre;
    y[i] = p.x * (quant_index_f(v, p.x, p.y, lmax) - p.y);
  }
}

extern "C" int tfmq_fake_quant_sel(tfmq_handle h, const float* x, float* y, size_t n, tfmq_qsel qs, int level, float pre, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && y && qs.qtable, "fake_quant_sel: null pointer");
  TFMQ_CHECK_ARG(h, level >= 2 && level <= 65536, "fake_quant_sel: level must be in [2, 65536]");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_fake_quant_sel, dim3(blocks), dim3(256), 0, as_stream(stream), x, y, n, qs, static_cast<float>(level - 1), pre);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// W8A8 layers (include/tfmq_hip.h): activation bins -> their integer grid (b - z_a), exact in fp16 / fp32
template <bool HALF>
__global__ __launch_bounds__(256) void k_bins_to_grid(const int8_t* __restrict__ xq, tfmq_qsel qs, void* __restrict__ out, size_t n) {
  const float zp = load_qparam(qs).y;
  const size_t n4 = n >> 2;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  const char4* q4 = reinterpret_cast<const char4*>(xq);
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const char4 c = q4[i];
    const float a = static_cast<float>(c.x + 128) - zp, b = static_cast<float>(c.y + 128) - zp;
    const float e = static_cast<float>(c.z + 128) - zp, f = static_cast<float>(c.w + 128) - zp;
    if constexpr (HALF) {
      const __half2 lo = __floats2half2_rn(a, b), hi = __floats2half2_rn(e, f);
      reinterpret_cast<uint2*>(out)[i] = make_uint2(*reinterpret_cast<const unsigned*>(&lo), *reinterpret_cast<const unsigned*>(&hi));
    } else {
      reinterpret_cast<float4*>(out)[i] = make_float4(a, b, e, f);
    }
  }
  if (!HALF && blockIdx.x == 0)
    for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) reinterpret_cast<float*>(out)[i] = static_cast<float>(xq[i] + 128) - zp;
}

extern "C" int tfmq_bins_to_grid(tfmq_handle h, const int8_t* xq, tfmq_qsel qs, void* out, int out_f16, size_t n, void* stream) {
  TFMQ_CHECK_ARG(h, h && xq && out && qs.qtable, "bins_to_grid: null pointer");
  TFMQ_CHECK_ARG(h, !out_f16 || n % 4 == 0, "bins_to_grid: fp16 output needs n % 4 == 0");
  TFMQ_CHECK_ARG(h, (reinterpret_cast<uintptr_t>(xq) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "bins_to_grid: alignment");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>((n + 3) / 4), 256);
  if (blocks > 8192) blocks = 8192;
  if (out_f16) hipLaunchKernelGGL(k_bins_to_grid<true>, dim3(blocks), dim3(256), 0, as_stream(stream), xq, qs, out, n);
  else hipLaunchKernelGGL(k_bins_to_grid<false>, dim3(blocks), dim3(256), 0, as_stream(stream), xq, qs, out, n);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

__global__ void k_scale_by_qdelta(const float* __restrict__ ws, tfmq_qsel qs, float* __restrict__ out, int n) {
  const float d = load_qparam(qs).x;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = d * ws[i];
}

extern "C" int tfmq_scale_by_qdelta(tfmq_handle h, const float* ws, tfmq_qsel qs, float* out, int n, void* stream) {
  TFMQ_CHECK_ARG(h, h && ws && out && qs.qtable && n > 0, "scale_by_qdelta: null pointer / empty");
  hipLaunchKernelGGL(k_scale_by_qdelta, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), ws, qs, out, n);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// the same on a tensor of the fp16 activation stream (n % 4 == 0)
__global__ __launch_bounds__(256) void k_quantize_act_h(const __half* __restrict__ x, int8_t* __restrict__ q, size_t n,
                                                        tfmq_qsel qs, float lmax) {
  const float2 p = load_qparam(qs);
  const size_t n4 = n >> 2;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  const uint2* x4 = reinterpret_cast<const uint2*>(x);
  char4* q4 = reinterpret_cast<char4*>(q);
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const uint2 u = x4[i];
    const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&u.x)), hi = __half22float2(*reinterpret_cast<const __half2*>(&u.y));
    char4 o;
    o.x = static_cast<signed char>(static_cast<int>(quant_index_f(lo.x, p.x, p.y, lmax)) - 128);
    o.y = static_cast<signed char>(static_cast<int>(quant_index_f(lo.y, p.x, p.y, lmax)) - 128);
    o.z = static_cast<signed char>(static_cast<int>(quant_index_f(hi.x, p.x, p.y, lmax)) - 128);
    o.w = static_cast<signed char>(static_cast<int>(quant_index_f(hi.y, p.x, p.y, lmax)) - 128);
    q4[i] = o;
  }
}

extern "C" int tfmq_quantize_act_h(tfmq_handle h, const uint16_t* x, int8_t* q, size_t n, tfmq_qsel qs, int level,
                                   void* stream) {
  TFMQ_CHECK_ARG(h, h && x && q && qs.qtable, "quantize_act_h: null pointer");
  TFMQ_CHECK_ARG(h, level >= 2 && level <= 256 && n % 4 == 0, "quantize_act_h: level must be in [2,256], n a multiple of 4");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n / 4), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_quantize_act_h, dim3(blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<const __half*>(x), q, n, qs,
                     static_cast<float>(level - 1));
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// fake-quant with per-row (or per-tensor, rows==1) parameters; optional index output
__global__ __launch_bounds__(256) void k_fake_quant(const float* __restrict__ x, float* __restrict__ y,
                                                    uint8_t* __restrict__ idx, size_t rows, size_t cols,
                                                    const float* __restrict__ delta, const float* __restrict__ zp,
                                                    float lmax) {
  const size_t n = rows * cols;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t r = i / cols;
    const float d = delta[r], z = zp[r];
    const float q = quant_index_f(x[i], d, z, lmax);
    if (y) y[i] = d * (q - z);
    if (idx) idx[i] = static_cast<uint8_t>(static_cast<int>(q));
  }
}

extern "C" int tfmq_fake_quant(tfmq_handle h, const float* x, float* y, uint8_t* idx, size_t rows, size_t cols,
                               const float* delta, const float* zp, int level, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && delta && zp && (y || idx), "fake_quant: null pointer");
  TFMQ_CHECK_ARG(h, level >= 2 && level <= 256, "fake_quant: level must be in [2,256]");
  const size_t n = rows * cols;
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_fake_quant, dim3(blocks), dim3(256), 0, as_stream(stream), x, y, idx, rows, cols, delta, zp,
                     static_cast<float>(level - 1));
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// Backward of the per-tensor fake quantisation through the straight-through round (the delta-learning reconstruction mode,
// reference quant/quant_layer.py:211-227 under autograd, quant/reconstruction.py:135-166):
//   y = delta * (clamp(rint(x / delta) + zp, 0, L-1) - zp);  d round(u)/du := 1
//   dL/dx     = g            where 0 <= rint(x/delta) + zp <= L-1, else 0
//   dL/ddelta = sum g * ((q - zp) - (x / delta) [in range])
// gx may be NULL.  part[gridDim.x]: per-block partial sums of dL/ddelta in double (the caller adds them in order: deterministic).
__global__ __launch_bounds__(256) void k_fake_quant_bwd(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ gx,
                                                        size_t n, const float* __restrict__ delta, const float* __restrict__ zp,
                                                        float lmax, double* __restrict__ part) {
  const float d = delta[0], z = zp[0];
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  double acc = 0.0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float u = x[i] / d;
    const float xi = rintf(u) + z;
    const bool in = xi >= 0.0f && xi <= lmax;
    const float q = fminf(fmaxf(xi, 0.0f), lmax);
    const float gi = g[i];
    if (gx) gx[i] = in ? gi : 0.0f;
    acc += static_cast<double>(gi) * static_cast<double>((q - z) - (in ? u : 0.0f));
  }
  acc = wave_reduce_sum_d(acc);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

extern "C" int tfmq_fake_quant_bwd(tfmq_handle h, const float* x, const float* g, float* gx, size_t n, const float* delta,
                                   const float* zp, int level, double* part, int nparts, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && g && delta && zp && part, "fake_quant_bwd: null pointer");
  TFMQ_CHECK_ARG(h, level >= 2 && level <= 65536 && nparts >= 1 && nparts <= 4096, "fake_quant_bwd: level in [2,65536], 1..4096 partial sums");
  hipLaunchKernelGGL(k_fake_quant_bwd, dim3(nparts), dim3(256), 0, as_stream(stream), x, g, gx, n, delta, zp, static_cast<float>(level - 1), part);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------------------ K2
// grid = (bpr, rows): block (j, r) reduces a contiguous slice of row r.
static inline int blocks_per_row(size_t rows, size_t cols, int cu) {
  long want = static_cast<long>(cu) * 8 / static_cast<long>(rows ? rows : 1);
  if (want < 1) want = 1;
  long maxb = static_cast<long>((cols + 4095) / 4096);  // >= 16 elements per thread
  if (maxb < 1) maxb = 1;
  if (want > maxb) want = maxb;
  if (want > 1024) want = 1024;
  return static_cast<int>(want);
}

__global__ __launch_bounds__(256) void k_minmax_partial(const float* __restrict__ x, size_t cols, int bpr,
                                                        float2* __restrict__ part) {
  const size_t r = blockIdx.y;
  const float* row = x + r * cols;
  const size_t chunk = (cols + bpr - 1) / bpr;
  const size_t beg = static_cast<size_t>(blockIdx.x) * chunk;
  size_t end = beg + chunk;
  if (end > cols) end = cols;
  float lo = INFINITY, hi = -INFINITY;
  // vector body when the slice start is 16-byte aligned
  size_t i = beg + threadIdx.x;
  if (((reinterpret_cast<uintptr_t>(row + beg)) & 15) == 0) {
    const float4* p = reinterpret_cast<const float4*>(row + beg);
    const size_t n4 = (end > beg) ? (end - beg) >> 2 : 0;
    for (size_t j = threadIdx.x; j < n4; j += blockDim.x) {
      float4 v = p[j];
      lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
      hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    i = beg + (n4 << 2) + threadIdx.x;
  }
  for (; i < end; i += blockDim.x) {
    float v = row[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  lo = wave_reduce_min(lo);
  hi = wave_reduce_max(hi);
  __shared__ float2 s[4];
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s[wid] = make_float2(lo, hi);
  __syncthreads();
  if (threadIdx.x == 0) {
    float2 a = s[0];
    for (int w = 1; w < 4; ++w) {
      a.x = fminf(a.x, s[w].x);
      a.y = fmaxf(a.y, s[w].y);
    }
    part[r * bpr + blockIdx.x] = a;
  }
}

__global__ void k_minmax_final(const float2* __restrict__ part, int bpr, float2* __restrict__ out) {
  const size_t r = blockIdx.x;
  float lo = INFINITY, hi = -INFINITY;
  for (int j = threadIdx.x; j < bpr; j += 64) {
    float2 a = part[r * bpr + j];
    lo = fminf(lo, a.x);
    hi = fmaxf(hi, a.y);
  }
  lo = wave_reduce_min(lo);
  hi = wave_reduce_max(hi);
  if (threadIdx.x == 0) out[r] = make_float2(lo, hi);
}

extern "C" size_t tfmq_minmax_ws_bytes(size_t rows, size_t cols) {
  (void)cols;
  return rows * 1024 * sizeof(float2);
}

extern "C" int tfmq_minmax(tfmq_handle h, const float* x, size_t rows, size_t cols, float* out, void* ws,
                           void* stream) {
  TFMQ_CHECK_ARG(h, h && x && out && ws, "minmax: null pointer");
  TFMQ_CHECK_ARG(h, rows > 0 && cols > 0 && rows < 65536, "minmax: rows must be in [1,65535], cols > 0");
  const int bpr = blocks_per_row(rows, cols, h->cu_count);
  hipLaunchKernelGGL(k_minmax_partial, dim3(bpr, rows), dim3(256), 0, as_stream(stream), x, cols, bpr,
                     reinterpret_cast<float2*>(ws));
  TFMQ_LAUNCH_CHECK(h);
  hipLaunchKernelGGL(k_minmax_final, dim3(rows), dim3(64), 0, as_stream(stream),
                     reinterpret_cast<const float2*>(ws), bpr, reinterpret_cast<float2*>(out));
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// MINMAX scaler arithmetic (quant_layer.py:23-35), identical operation order on device:
// double subtraction and division, round to fp32, zp = rint(fp32(-lo) / delta).
__device__ __forceinline__ float2 minmax_qparam(float mn, float mx, int level, int always_zero) {
  const double lo = fmin(static_cast<double>(mn), 0.0), hi = fmax(static_cast<double>(mx), 0.0);
  float delta = static_cast<float>((hi - lo) / static_cast<double>(level - 1));
  if (always_zero) delta = static_cast<float>(hi / static_cast<double>(level - 1));
  if (delta < 1e-8f) delta = 1e-8f;  // the reference crashes here; defined behaviour instead
  const float zp = always_zero ? 0.0f : rintf(static_cast<float>(-lo) / delta);
  return make_float2(delta, zp);
}

__global__ void k_minmax_to_qparam(const float2* __restrict__ mm, size_t rows, int level, int always_zero,
                                   float2* __restrict__ qp) {
  size_t r = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r < rows) qp[r] = minmax_qparam(mm[r].x, mm[r].y, level, always_zero);
}

extern "C" int tfmq_minmax_to_qparam(tfmq_handle h, const float* mm, size_t rows, int level, int always_zero,
                                     float* qparam, void* stream) {
  TFMQ_CHECK_ARG(h, h && mm && qparam && rows > 0, "minmax_to_qparam: bad argument");
  hipLaunchKernelGGL(k_minmax_to_qparam, dim3(ceil_div(rows, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float2*>(mm), rows, level, always_zero, reinterpret_cast<float2*>(qparam));
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

__global__ void k_act_range_update(const float2* __restrict__ mm, float2* __restrict__ state, float2* __restrict__ qp,
                                   float momentum, float om, int level, int init) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float2 s = state[0];
  const float2 b = mm[0];
  if (init) {
    s = b;  // _init_quantization_param leaf_param branch (quant_layer.py:206-207)
  } else {
    // x_min * m + batch_min * (1 - m): fp32 tensor * Python double -> fp32 ops (quant_layer.py:237-238)
    s.x = s.x * momentum + b.x * om;
    s.y = s.y * momentum + b.y * om;
  }
  state[0] = s;
  qp[0] = minmax_qparam(s.x, s.y, level, 0);
}

extern "C" int tfmq_act_range_update(tfmq_handle h, const float* mm, float* state, float* qparam, double momentum,
                                     int level, int init, void* stream) {
  TFMQ_CHECK_ARG(h, h && mm && state && qparam, "act_range_update: null pointer");
  hipLaunchKernelGGL(k_act_range_update, dim3(1), dim3(64), 0, as_stream(stream), reinterpret_cast<const float2*>(mm),
                     reinterpret_cast<float2*>(state), reinterpret_cast<float2*>(qparam), static_cast<float>(momentum),
                     static_cast<float>(1.0 - momentum), level, init);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------------------ K3
#define NCAND 80
// The 80 (delta_i, zp_i) of quant_layer.py:45-55 in the Python's own operation order:
// double shrink factor 1.0 - (i*0.01), double range, fp32 delta, zp = rint(fp32(-new_min)/delta).
__device__ __forceinline__ float2 mse_candidate(float mn, float mx, int i, int level, int always_zero) {
  const double f = 1.0 - (static_cast<double>(i) * 0.01);
  const double nmin = static_cast<double>(mn) * f, nmax = static_cast<double>(mx) * f;
  float nd = static_cast<float>((nmax - nmin) / static_cast<double>(level - 1));
  if (always_zero) nd = static_cast<float>(nmax / static_cast<double>(level - 1));
  const float nz = always_zero ? 0.0f : rintf(static_cast<float>(-nmin) / nd);
  return make_float2(nd, nz);
}

// |d|^2.4 (lp_loss p=2.4, quant_layer.py:59)
__device__ __forceinline__ float pow24(float a) { return a == 0.0f ? 0.0f : exp2f(2.4f * log2f(a)); }

template <int CPT>  // candidates per thread-pass
__global__ __launch_bounds__(256) void k_mse_partial(const float* __restrict__ x, size_t cols, int bpr,
                                                     const float2* __restrict__ mm, int level, int always_zero,
                                                     double* __restrict__ part /*[rows][bpr][80]*/) {
  __shared__ float2 cand[NCAND];
  __shared__ double red[4][NCAND];
  const size_t r = blockIdx.y;
  const float2 m = mm[r];
  if (threadIdx.x < NCAND) cand[threadIdx.x] = mse_candidate(m.x, m.y, threadIdx.x, level, always_zero);
  __syncthreads();
  const float lmax = static_cast<float>(level - 1);
  const float* row = x + r * cols;
  const size_t chunk = (cols + bpr - 1) / bpr;
  const size_t beg = static_cast<size_t>(blockIdx.x) * chunk;
  size_t end = beg + chunk;
  if (end > cols) end = cols;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // candidates are processed CPT at a time so the accumulators stay in registers; the row
  // slice is re-read from L2 for each group (it was just streamed by this same block).
  for (int c0 = 0; c0 < NCAND; c0 += CPT) {
    float acc[CPT];
    float2 cd[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      acc[c] = 0.0f;
      cd[c] = cand[c0 + c];
    }
    for (size_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
      const float v = row[i];
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const float q = quant_index_f(v, cd[c].x, cd[c].y, lmax);
        const float dq = cd[c].x * (q - cd[c].y);
        acc[c] += pow24(fabsf(dq - v));
      }
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      double s = wave_reduce_sum_d(static_cast<double>(acc[c]));
      if (lane == 0) red[wid][c0 + c] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < NCAND)
    part[(r * bpr + blockIdx.x) * NCAND + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void k_mse_final(const double* __restrict__ part, int bpr, size_t cols, const float2* __restrict__ mm,
                            int level, int always_zero, float2* __restrict__ qp, float* __restrict__ losses,
                            int32_t* __restrict__ best_out) {
  __shared__ float mean[NCAND];
  const size_t r = blockIdx.x;
  if (threadIdx.x < NCAND) {
    double s = 0.0;
    for (int j = 0; j < bpr; ++j) s += part[(r * bpr + j) * NCAND + threadIdx.x];
    const float mu = static_cast<float>(s / static_cast<double>(cols));
    mean[threadIdx.x] = mu;
    if (losses) losses[r * NCAND + threadIdx.x] = mu;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // first strict minimum, initial s = 1e10 (quant_layer.py:44,60-62)
    float s = 1e10f;
    int best = -1;
    for (int i = 0; i < NCAND; ++i)
      if (mean[i] < s) {
        s = mean[i];
        best = i;
      }
    if (best_out) best_out[r] = best;
    const float2 m = mm[r];
    qp[r] = best >= 0 ? mse_candidate(m.x, m.y, best, level, always_zero) : make_float2(NAN, NAN);
  }
}

extern "C" size_t tfmq_mse_ws_bytes(size_t rows, size_t cols) {
  (void)cols;
  return rows * 1024 * NCAND * sizeof(double);
}

extern "C" int tfmq_mse_search(tfmq_handle h, const float* x, size_t rows, size_t cols, const float* mm, int level,
                               int always_zero, float* qparam, float* losses, int32_t* best, void* ws, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && mm && qparam && ws, "mse_search: null pointer");
  TFMQ_CHECK_ARG(h, rows > 0 && cols > 0 && rows < 65536, "mse_search: rows must be in [1,65535], cols > 0");
  TFMQ_CHECK_ARG(h, level >= 2 && level <= 65536, "mse_search: bad level");
  int bpr = blocks_per_row(rows, cols, h->cu_count);
  hipLaunchKernelGGL(k_mse_partial<8>, dim3(bpr, rows), dim3(256), 0, as_stream(stream), x, cols, bpr,
                     reinterpret_cast<const float2*>(mm), level, always_zero, reinterpret_cast<double*>(ws));
  TFMQ_LAUNCH_CHECK(h);
  hipLaunchKernelGGL(k_mse_final, dim3(rows), dim3(128), 0, as_stream(stream), reinterpret_cast<const double*>(ws),
                     bpr, cols, reinterpret_cast<const float2*>(mm), level, always_zero,
                     reinterpret_cast<float2*>(qparam), losses, best);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------------------ K4
// one block per output channel; thread handles groups of 8 consecutive k (one packed dword)
__global__ __launch_bounds__(256) void k_pack_w4(const float* __restrict__ w, const float* __restrict__ alpha,
                                                 const float* __restrict__ delta, const float* __restrict__ zp,
                                                 int cin, int kh, int kw, uint32_t* __restrict__ packed,
                                                 int32_t* __restrict__ wmeta) {
  const int co = blockIdx.x;
  const int K = kh * kw * cin;
  const int khw = kh * kw;
  const float d = delta[co], z = zp[co];
  const float* wr = w + static_cast<size_t>(co) * K;
  const float* ar = alpha ? alpha + static_cast<size_t>(co) * K : nullptr;
  int sum = 0;
  for (int g = threadIdx.x; g < K / 8; g += blockDim.x) {
    const int k0 = g * 8;
    const int tap = k0 / cin, ci0 = k0 - tap * cin;  // cin % 8 == 0 => the 8 values share a tap
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t src = static_cast<size_t>(ci0 + i) * khw + tap;  // OIHW: [ci][kh][kw]
      const float v = wr[src];
      float q;
      if (ar) {  // AdaRound hard rounding (adaptive_rounding.py:51,63,67-68)
        q = floorf(v / d) + (ar[src] >= 0.0f ? 1.0f : 0.0f) + z;
      } else {   // nearest (quant_layer.py:225)
        q = rintf(v / d) + z;
      }
      q = fminf(fmaxf(q, 0.0f), 15.0f);
      const uint32_t qi = static_cast<uint32_t>(static_cast<int>(q));
      sum += static_cast<int>(qi);
      word |= qi << ((i & 3) * 8 + (i >> 2) * 4);
    }
    packed[w4_word_index(co, g, K, w4_ck(cin))] = word;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  __shared__ int s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int4 m;
    m.x = static_cast<int>(z);
    m.y = s[0] + s[1] + s[2] + s[3];
    m.z = 0;
    m.w = 0;
    reinterpret_cast<int4*>(wmeta)[co] = m;
  }
}

extern "C" int tfmq_pack_w4(tfmq_handle h, const float* w, const float* alpha, const float* delta, const float* zp,
                            int cout, int cin, int kh, int kw, uint8_t* packed, int32_t* wmeta, void* stream) {
  TFMQ_CHECK_ARG(h, h && w && delta && zp && packed && wmeta, "pack_w4: null pointer");
  TFMQ_CHECK_ARG(h, cout > 0 && cin > 0 && kh > 0 && kw > 0, "pack_w4: bad shape");
  TFMQ_CHECK_ARG(h, cin % 8 == 0, "pack_w4: cin must be a multiple of 8");
  hipLaunchKernelGGL(k_pack_w4, dim3(cout), dim3(256), 0, as_stream(stream), w, alpha, delta, zp, cin, kh, kw,
                     reinterpret_cast<uint32_t*>(packed), wmeta);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

__global__ void k_unpack_w4(const uint32_t* __restrict__ packed, int cout, int cin, int kh, int kw,
                            uint8_t* __restrict__ idx) {
  const int K = kh * kw * cin, khw = kh * kw;
  const size_t total = static_cast<size_t>(cout) * (K / 8);
  for (size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < total;
       g += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int co = g / (K / 8), k0 = (g % (K / 8)) * 8;
    const int tap = k0 / cin, ci0 = k0 - tap * cin;
    const uint32_t word = packed[w4_word_index(co, static_cast<int>(g % (K / 8)), K, w4_ck(cin))];
    for (int i = 0; i < 8; ++i) {
      const uint32_t qi = (word >> ((i & 3) * 8 + (i >> 2) * 4)) & 15u;
      idx[static_cast<size_t>(co) * K + static_cast<size_t>(ci0 + i) * khw + tap] = static_cast<uint8_t>(qi);
    }
  }
}

extern "C" int tfmq_unpack_w4(tfmq_handle h, const uint8_t* packed, int cout, int cin, int kh, int kw, uint8_t* idx,
                              void* stream) {
  TFMQ_CHECK_ARG(h, h && packed && idx && cin % 8 == 0, "unpack_w4: bad argument");
  const size_t total = static_cast<size_t>(cout) * (kh * kw * cin / 8);
  hipLaunchKernelGGL(k_unpack_w4, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const uint32_t*>(packed), cout, cin, kh, kw, idx);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// Packed int4 (at-rest format) -> MFMA-ready int8 operand: byte (n, k) = q_w - z_w in [-15, 15], tile-major
//   byte(n, k) = ((n/32 * nsteps + k/ck) * 32 + n%32) * ck + k%ck,   ck = w4_ck(cin), nsteps = K/ck
// so one K-step of 32 output channels is one contiguous 32*ck-byte block the conv kernel DMAs straight into
// LDS.  Folding z_w into the operand removes the per-row activation sums from the GEMM
// (sum a'(q_w - z_w) needs no -z_w*sum a' correction).  Rows cout..cout_pad32 are zero.
__global__ void k_expand_w4(const uint32_t* __restrict__ packed, const int32_t* __restrict__ wmeta, int cout,
                            int cout_pad, int K, int ck, int8_t* __restrict__ w8) {
  const size_t total = static_cast<size_t>(cout_pad) * (K / 8);
  const int nsteps = K / ck;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int n = static_cast<int>(i / (K / 8)), g = static_cast<int>(i % (K / 8));
    const int k0 = g * 8;
    int8_t o[8];
    if (n < cout) {
      const uint32_t word = packed[w4_word_index(n, g, K, ck)];
      const int z = wmeta[4 * n];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = static_cast<int8_t>(static_cast<int>((word >> ((j & 3) * 8 + (j >> 2) * 4)) & 15u) - z);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0;
    }
    const size_t dst = ((static_cast<size_t>(n / 32) * nsteps + k0 / ck) * 32 + (n % 32)) * ck + (k0 % ck);
    *reinterpret_cast<uint2*>(w8 + dst) = *reinterpret_cast<const uint2*>(o);
  }
}

extern "C" int tfmq_expand_w4(tfmq_handle h, const uint8_t* packed, const int32_t* wmeta, int cout, int cin, int kh,
                              int kw, int8_t* w8, void* stream) {
  TFMQ_CHECK_ARG(h, h && packed && wmeta && w8, "expand_w4: null pointer");
  TFMQ_CHECK_ARG(h, cout > 0 && cin > 0 && kh > 0 && kw > 0 && cin % 32 == 0, "expand_w4: cin must be a multiple of 32");
  const int K = kh * kw * cin, cout_pad = (cout + 31) / 32 * 32;
  const size_t total = static_cast<size_t>(cout_pad) * (K / 8);
  hipLaunchKernelGGL(k_expand_w4, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const uint32_t*>(packed), wmeta, cout, cout_pad, K, w4_ck(cin), w8);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// K-padded form (tfmq_expand_w4_k64): 64-channel K-steps, the channels cin .. 64*chunks of every tap are zero
__global__ void k_expand_w4_k64(const uint32_t* __restrict__ packed, const int32_t* __restrict__ wmeta, int cout, int cout_pad,
                                int cin, int khw, int ck, int8_t* __restrict__ w8p) {
  const int chunks = (cin + 63) / 64, nsteps = khw * chunks, K = khw * cin;
  const size_t total = static_cast<size_t>(cout_pad) * nsteps * 8;          // 8-channel groups
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int n = static_cast<int>(i / (static_cast<size_t>(nsteps) * 8));
    const int r = static_cast<int>(i - static_cast<size_t>(n) * nsteps * 8);
    const int step = r >> 3, c0 = (step % chunks) * 64 + (r & 7) * 8, tap = step / chunks;
    int8_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n < cout && c0 < cin) {
      const uint32_t word = packed[w4_word_index(n, (tap * cin + c0) / 8, K, ck)];
      const int z = wmeta[4 * n];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = static_cast<int8_t>(static_cast<int>((word >> ((j & 3) * 8 + (j >> 2) * 4)) & 15u) - z);
    }
    const size_t dst = ((static_cast<size_t>(n / 32) * nsteps + step) * 32 + (n % 32)) * 64 + (r & 7) * 8;
    *reinterpret_cast<uint2*>(w8p + dst) = *reinterpret_cast<const uint2*>(o);
  }
}

extern "C" int tfmq_expand_w4_k64(tfmq_handle h, const uint8_t* packed, const int32_t* wmeta, int cout, int cin, int kh, int kw,
                                  int8_t* w8p, void* stream) {
  TFMQ_CHECK_ARG(h, h && packed && wmeta && w8p, "expand_w4_k64: null pointer");
  TFMQ_CHECK_ARG(h, cout > 0 && cin > 0 && kh > 0 && kw > 0 && cin % 32 == 0, "expand_w4_k64: cin must be a multiple of 32");
  const int cout_pad = (cout + 31) / 32 * 32, chunks = (cin + 63) / 64;
  const size_t total = static_cast<size_t>(cout_pad) * kh * kw * chunks * 8;
  hipLaunchKernelGGL(k_expand_w4_k64, dim3(ceil_div(static_cast<long>(total), 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const uint32_t*>(packed), wmeta, cout, cout_pad, cin, kh * kw, w4_ck(cin), w8p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

__global__ void k_pack_w_f16(const float* __restrict__ w, const float* __restrict__ alpha,
                             const float* __restrict__ delta, const float* __restrict__ zp, float lmax, int cout,
                             int cin, int cin_pad, int khw, __half* __restrict__ out) {
  const size_t total = static_cast<size_t>(cout) * cin_pad * khw;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    // destination index: [co][tap][ci] with ci padded (zeros) to a multiple of 32
    const int ci = i % cin_pad;
    const int tap = (i / cin_pad) % khw;
    const int co = i / (static_cast<size_t>(cin_pad) * khw);
    float v = 0.0f;
    if (ci < cin) {
      const size_t src = (static_cast<size_t>(co) * cin + ci) * khw + tap;
      v = w[src];
      if (delta) {  // integer grid coordinate of the weight quantizer (exact in f16)
        const float d = delta[co], z = zp[co];
        float q = alpha ? floorf(v / d) + (alpha[src] >= 0.0f ? 1.0f : 0.0f) + z : rintf(v / d) + z;
        q = fminf(fmaxf(q, 0.0f), lmax);
        v = q - z;
      }
    }
    out[i] = __float2half_rn(v);
  }
}

extern "C" int tfmq_pack_w_f16(tfmq_handle h, const float* w, const float* alpha, const float* delta, const float* zp,
                               int level, int cout, int cin, int kh, int kw, uint16_t* out, void* stream) {
  TFMQ_CHECK_ARG(h, h && w && out && cout > 0 && cin > 0, "pack_w_f16: bad argument");
  TFMQ_CHECK_ARG(h, (delta == nullptr) == (zp == nullptr) && (!alpha || delta), "pack_w_f16: delta/zp/alpha mismatch");
  const int cin_pad = (cin + 31) / 32 * 32;
  const size_t total = static_cast<size_t>(cout) * cin_pad * kh * kw;
  hipLaunchKernelGGL(k_pack_w_f16, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w, alpha, delta, zp,
                     static_cast<float>(level - 1), cout, cin, cin_pad, kh * kw, reinterpret_cast<__half*>(out));
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Hardware semantics the epilogues rely on and the ISA text leaves implicit, checked on the device once per handle user
// (tfmq_hw_selftest; the Python layer calls it when it creates a handle and refuses to run otherwise):
//   * v_cvt_pk_u8_f32 SATURATES to [0, 255] (quant_pack4 leaves the clamp of clamp(rint(x / delta) + zp, 0, 255) to it), converts
//     integers exactly and keeps the other three bytes of the destination;
//   * v_cvt_pk_u8_f32 rounds to nearest-even (the consumer-sized GEGLU epilogue converts without a v_rndne_f32);
//   * DPP quad_perm / row_shl source lanes as group8_sum (conv_common.hpp) assumes them.
__global__ void k_hw_selftest(unsigned* out) {
  const int lane = threadIdx.x;
  unsigned fail = 0;
  const float vals[13] = {-1e30f, -300.0f, -1.0f, -0.0f, 0.0f, 1.0f, 127.0f, 128.0f, 254.0f, 255.0f, 256.0f, 300.0f, 1e30f};
  const unsigned want[13] = {0, 0, 0, 0, 0, 1, 127, 128, 254, 255, 255, 255, 255};
#pragma unroll
  for (int i = 0; i < 13; ++i) {
    const unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(vals[i], 2, 0xAABBCCDDu);
    if (((w >> 16) & 0xffu) != want[i] || (w & 0xff00ffffu) != 0xAA00CCDDu) fail |= 1u;
  }
  for (int k = 0; k < 4; ++k) {                       // every integer 0..255
    const unsigned v = lane * 4 + k;
    if ((__builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>(v), 0, 0u) & 0xffu) != v) fail |= 2u;
  }
  for (int k = 0; k < 4; ++k) {                       // round-half-even on its own (geglu_fast_pack4 has no v_rndne_f32 in front of it)
    const float v = static_cast<float>(lane * 4 + k);
    const float probes[5] = {v + 0.5f, v + 0.25f, v + 0.75f, __uint_as_float(__float_as_uint(v + 0.5f) + 1u), __uint_as_float(__float_as_uint(v + 0.5f) - 1u)};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const float r = fminf(__builtin_rintf(probes[i]), 255.0f);
      if ((__builtin_amdgcn_cvt_pk_u8_f32(probes[i], 0, 0u) & 0xffu) != static_cast<unsigned>(r)) fail |= 16u;
    }
  }
  const float x = static_cast<float>(lane);
  const float q1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x55, 0xf, 0xf, true));
  const float q3 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xFF, 0xf, 0xf, true));
  const float s4 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xf, 0xf, true));
  if (q1 != static_cast<float>((lane & ~3) + 1) || q3 != static_cast<float>((lane & ~3) + 3)) fail |= 4u;
  if ((lane & 15) < 12 && s4 != static_cast<float>(lane + 4)) fail |= 8u;
  if (fail) atomicOr(out, fail);
}

extern "C" int tfmq_hw_selftest(tfmq_handle h, uint32_t* report) {
  TFMQ_CHECK_ARG(h, h != nullptr, "hw_selftest: null handle");
  // (allocates, launches on the null stream and copies synchronously: run it before any stream capture on this device -- tfmq_create /
  // the first handle(dev) does; the device is selected explicitly so that a handle made from another current device tests ITS GPU)
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(h->device) != hipSuccess) return TFMQ_ERR_HIP;
  struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
  unsigned* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned)) != hipSuccess) return TFMQ_ERR_HIP;
  (void)hipMemset(d, 0, sizeof(unsigned));
  hipLaunchKernelGGL(k_hw_selftest, dim3(1), dim3(64), 0, nullptr, d);
  unsigned r = 0xffffffffu;
  const hipError_t e = hipMemcpy(&r, d, sizeof(unsigned), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return TFMQ_ERR_HIP;
  if (report) *report = r;
  if (r != 0) {
    h->err = "hardware self-test failed (bit 0/1: v_cvt_pk_u8_f32 saturation / exactness, bit 2/3: DPP lane selection, bit 4: v_cvt_pk_u8_f32 round-half-even): mask " + std::to_string(r);
    return TFMQ_ERR_HIP;
  }
  return TFMQ_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Histogram with numpy's bin assignment (np.histogram on `bins` equal-width bins), for the KL / HIST scalers
// (quant/quant_layer.py:67-133): index = trunc(((v - first) / (last - first)) * bins) in the edges' own precision, the
// value on the last edge goes to the last bin, then the +-1 correction against the edge table (numpy's guard against the
// ~1 ulp inconsistency of the index arithmetic); values outside [first, last] are dropped.  T = float for fp32 data with
// fp32 edges (np.histogram(a_f32)), double for the clipped data (np.clip with float64 bounds promotes to float64).
template <typename T>
__global__ __launch_bounds__(256) void k_np_histogram(const float* __restrict__ x, size_t n, int do_clip, double clip_lo, double clip_hi,
                                                      const T* __restrict__ edges, int bins, unsigned* __restrict__ counts) {
  extern __shared__ unsigned sh[];
  for (int i = threadIdx.x; i < bins; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  const T first = edges[0], last = edges[bins];
  const T denom = last - first;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    T v = static_cast<T>(x[i]);
    if (do_clip) {                       // np.clip(a, lo, hi) = minimum(maximum(a, lo), hi) in float64
      const double c = fmin(fmax(static_cast<double>(x[i]), clip_lo), clip_hi);
      v = static_cast<T>(c);
    }
    if (!(v >= first) || !(v <= last)) continue;
    int idx = static_cast<int>(((v - first) / denom) * static_cast<T>(bins));
    if (idx == bins) idx -= 1;
    if (v < edges[idx]) idx -= 1;
    if (idx != bins - 1 && v >= edges[idx + 1]) idx += 1;
    atomicAdd(&sh[idx], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += blockDim.x)
    if (sh[i]) atomicAdd(&counts[i], sh[i]);
}

// the same for `rows` tensors of n values at once (the per-output-channel loop of the KL / HIST weight scalers, quant_layer.py:193-204):
// row r has its own edge table edges[r][bins + 1] and clip bounds; blockIdx.y = row
template <typename T>
__global__ __launch_bounds__(256) void k_np_histogram_rows(const float* __restrict__ x, size_t n, int do_clip, const double* __restrict__ clip_lo,
                                                           const double* __restrict__ clip_hi, const T* __restrict__ edges_all, int bins,
                                                           unsigned* __restrict__ counts_all) {
  extern __shared__ unsigned sh[];
  const int row = blockIdx.y;
  const float* xr = x + static_cast<size_t>(row) * n;
  const T* edges = edges_all + static_cast<size_t>(row) * (bins + 1);
  unsigned* counts = counts_all + static_cast<size_t>(row) * bins;
  for (int i = threadIdx.x; i < bins; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  const T first = edges[0], last = edges[bins];
  const T denom = last - first;
  const double lo = do_clip ? clip_lo[row] : 0.0, hi = do_clip ? clip_hi[row] : 0.0;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    T v = static_cast<T>(xr[i]);
    if (do_clip) v = static_cast<T>(fmin(fmax(static_cast<double>(xr[i]), lo), hi));
    if (!(v >= first) || !(v <= last)) continue;
    int idx = static_cast<int>(((v - first) / denom) * static_cast<T>(bins));
    if (idx == bins) idx -= 1;
    if (v < edges[idx]) idx -= 1;
    if (idx != bins - 1 && v >= edges[idx + 1]) idx += 1;
    atomicAdd(&sh[idx], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += blockDim.x)
    if (sh[i]) atomicAdd(&counts[i], sh[i]);
}

extern "C" int tfmq_np_histogram_rows(tfmq_handle h, const float* x, size_t rows, size_t n, int f64, int do_clip, const double* clip_lo,
                                      const double* clip_hi, const void* edges, int bins, uint32_t* counts, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && edges && counts && rows > 0 && rows <= 65535 && n > 0 && bins > 0 && bins <= 4096 && (!do_clip || (clip_lo && clip_hi)),
                 "np_histogram_rows: bad argument");
  TFMQ_HIP(h, hipMemsetAsync(counts, 0, sizeof(uint32_t) * bins * rows, as_stream(stream)));
  int blocks = ceil_div(static_cast<long>(n), 256 * 8);
  if (blocks > 64) blocks = 64;
  if (blocks < 1) blocks = 1;
  const size_t shm = sizeof(unsigned) * bins;
  dim3 grid(blocks, static_cast<unsigned>(rows));
  if (f64) hipLaunchKernelGGL(k_np_histogram_rows<double>, grid, dim3(256), shm, as_stream(stream), x, n, do_clip, clip_lo, clip_hi,
                              static_cast<const double*>(edges), bins, counts);
  else hipLaunchKernelGGL(k_np_histogram_rows<float>, grid, dim3(256), shm, as_stream(stream), x, n, do_clip, clip_lo, clip_hi,
                          static_cast<const float*>(edges), bins, counts);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_np_histogram(tfmq_handle h, const float* x, size_t n, int f64, int do_clip, double clip_lo, double clip_hi,
                                 const void* edges, int bins, uint32_t* counts, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && edges && counts && n > 0 && bins > 0 && bins <= 4096, "np_histogram: bad argument");
  TFMQ_HIP(h, hipMemsetAsync(counts, 0, sizeof(uint32_t) * bins, as_stream(stream)));
  int blocks = ceil_div(static_cast<long>(n), 256 * 8);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  if (blocks < 1) blocks = 1;
  const size_t shm = sizeof(unsigned) * bins;
  if (f64) hipLaunchKernelGGL(k_np_histogram<double>, dim3(blocks), dim3(256), shm, as_stream(stream), x, n, do_clip, clip_lo, clip_hi,
                              static_cast<const double*>(edges), bins, counts);
  else hipLaunchKernelGGL(k_np_histogram<float>, dim3(blocks), dim3(256), shm, as_stream(stream), x, n, do_clip, clip_lo, clip_hi,
                          static_cast<const float*>(edges), bins, counts);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
