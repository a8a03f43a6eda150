// K8: GroupNorm (+SiLU) (+8-bit quantise), reading a *virtual* channel concat of two NHWC
// tensors.  Replaces Normalize -> nonlinearity -> aqtizer (ddim/models/diffusion.py:27-33,
// 117-118,123-124; quant/quant_layer.py:223-226): ONE read of fp32 and one write of int8,
// instead of the reference's ~10 elementwise passes.
//
// One block = one image x one chunk of whole groups spanning ~32 channels (128-byte rows of the
// NHWC tensor).  When the chunk fits (pixels/iteration x R iterations), every thread keeps its
// elements in registers between the statistics pass and the apply pass, so HBM sees each input
// element exactly once: algorithmic bytes = 4 B read + 1 B written per element.  Larger groups
// (SD 64x64) fall back to a second read (normally an L2 / Infinity-Cache hit).
// Statistics: fp32 partial sums per thread (<= 128 values), combined in double in a fixed order
// (deterministic run to run).
#include "common.hpp"
#include <cstdlib>

struct GnP {
  tfmq_gn_desc d;
  int C, cpg, gpb, cw;  // channels, channels/group, groups/block, chunk width = gpb*cpg
  int tpp, ppi;         // threads per pixel, pixels per iteration
  int lpg;              // lanes cooperating on one group's reduction (power of two <= 64)
};

template <int V>
__device__ __forceinline__ void gn_load(const GnP& p, size_t pix, int c, float (&v)[V]) {
  const tfmq_gn_desc& d = p.d;
  // V consecutive channels never straddle the concat boundary (C1 % V == 0 is checked on the host)
  if (d.x_f16) {       // fp16 activation stream: x1 / x2 are fp16 buffers
    const __half* sh = c < d.C1 ? reinterpret_cast<const __half*>(d.x1) + pix * d.C1 + c
                                : reinterpret_cast<const __half*>(d.x2) + pix * d.C2 + (c - d.C1);
    if constexpr (V == 4) {
      const uint2 u = *reinterpret_cast<const uint2*>(sh);
      const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&u.x)), hi = __half22float2(*reinterpret_cast<const __half2*>(&u.y));
      v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) v[i] = __half2float(sh[i]);
    }
    return;
  }
  const float* src = c < d.C1 ? d.x1 + pix * d.C1 + c : d.x2 + pix * d.C2 + (c - d.C1);
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(src);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (V == 2) {
    const float2 t = *reinterpret_cast<const float2*>(src);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = src[0];
  }
}

template <int V>
__device__ __forceinline__ void gn_apply_store(const GnP& p, size_t o, const float (&v)[V], const float (&ga)[V],
                                               const float (&gb)[V], bool quant, float2 qp) {
  const tfmq_gn_desc& d = p.d;
  if (d.xcat_or_null) {
    if (d.half_out) {
      __half* xh = reinterpret_cast<__half*>(d.xcat_or_null);
#pragma unroll
      for (int i = 0; i < V; ++i) xh[o + i] = __float2half_rn(v[i]);
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) d.xcat_or_null[o + i] = v[i];
    }
  }
  float y[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    y[i] = ga[i] * v[i] + gb[i];
    if (d.silu) y[i] = silu_f(y[i]);
  }
  if (quant) {
    signed char q[V];
    if constexpr (V != 4) {
#pragma unroll
      for (int i = 0; i < V; ++i)
        q[i] = static_cast<signed char>(static_cast<int>(quant_index_f(y[i], qp.x, qp.y, 255.0f)) - 128);
    }
    if constexpr (V == 4) {
      *reinterpret_cast<char4*>(d.yq + o) = quant_char4(y[0], y[1], y[2], y[3], make_quantp(qp));
    } else if constexpr (V == 2) {
      *reinterpret_cast<char2*>(d.yq + o) = make_char2(q[0], q[1]);
    } else {
      d.yq[o] = q[0];
    }
  }
  if (d.yf) {
    if (d.half_out) {
      __half* yh = reinterpret_cast<__half*>(d.yf);
#pragma unroll
      for (int i = 0; i < V; ++i) yh[o + i] = __float2half_rn(y[i]);
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) d.yf[o + i] = y[i];
    }
  }
}

// R > 0: register-cached single-read path (HW <= R * ppi);  R == 0: re-read path.
template <int V, int R>
__global__ __launch_bounds__(256) void k_groupnorm(GnP p) {
  const tfmq_gn_desc& d = p.d;
  __shared__ double s_part[256][2];
  __shared__ float s_stat[64][2];  // mean, rstd per group of this block
  const int b = blockIdx.y;
  const int c_base = blockIdx.x * p.cw;
  const int tpp = p.tpp, ppi = p.ppi;
  const int t = threadIdx.x;
  const bool active = t < tpp * ppi;
  const int c_off = (t % tpp) * V;    // channel offset inside the chunk
  const int p_off = t / tpp;
  const int g_rel = c_off / p.cpg;    // V divides cpg => all V channels share the group
  const size_t pix0 = static_cast<size_t>(b) * d.HW;
  constexpr int RR = R > 0 ? R : 1;
  float cache[RR][V];

  // ---- pass 1: sum and sum of squares
  float s = 0.0f, ss = 0.0f;
  if (active) {
    if constexpr (R > 0) {
      // issue every load first (clamped address, no predicate) so they are all in flight together
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int px = p_off + r * ppi;
        gn_load<V>(p, pix0 + (px < d.HW ? px : d.HW - 1), c_base + c_off, cache[r]);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float m = (p_off + r * ppi) < d.HW ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          s += m * cache[r][i];
          ss += m * (cache[r][i] * cache[r][i]);
        }
      }
    } else {
      double ds = 0.0, dss = 0.0;
      for (int px0 = p_off; px0 < d.HW; px0 += ppi * 32) {  // fp32 runs of <= 32 pixels, double across runs
        float fs = 0.0f, fss = 0.0f;
        for (int px = px0; px < d.HW && px < px0 + ppi * 32; px += ppi) {
          float v[V];
          gn_load<V>(p, pix0 + px, c_base + c_off, v);
#pragma unroll
          for (int i = 0; i < V; ++i) {
            fs += v[i];
            fss += v[i] * v[i];
          }
        }
        ds += fs;
        dss += fss;
      }
      s_part[t][0] = ds;
      s_part[t][1] = dss;
    }
  }
  if constexpr (R > 0) {
    s_part[t][0] = active ? static_cast<double>(s) : 0.0;
    s_part[t][1] = active ? static_cast<double>(ss) : 0.0;
  } else {
    if (!active) {
      s_part[t][0] = 0.0;
      s_part[t][1] = 0.0;
    }
  }
  __syncthreads();
  // ---- fixed-order reduction: lpg lanes per group, each sums a strided subset, then xor-shuffle
  {
    const int g = t / p.lpg, l = t % p.lpg;
    double a = 0.0, aa = 0.0;
    if (g < p.gpb) {
      const int nact = tpp * ppi;
      for (int j = l; j < nact; j += p.lpg) {
        if (((j % tpp) * V) / p.cpg == g) {
          a += s_part[j][0];
          aa += s_part[j][1];
        }
      }
    }
    for (int o = p.lpg >> 1; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      aa += __shfl_xor(aa, o, 64);
    }
    if (g < p.gpb && l == 0) {
      const double n = static_cast<double>(d.HW) * p.cpg;
      const double mean = a / n;
      double var = aa / n - mean * mean;
      if (var < 0.0) var = 0.0;
      s_stat[g][0] = static_cast<float>(mean);
      s_stat[g][1] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(d.eps)));
    }
  }
  __syncthreads();
  if (!active) return;

  // ---- pass 2: y = a*x + b (a = rstd*gamma, b = beta - a*mean), SiLU, quantise
  const float mean = s_stat[g_rel][0], rstd = s_stat[g_rel][1];
  float ga[V], gb[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = c_base + c_off + i;
    ga[i] = rstd * d.gamma[c];
    gb[i] = d.beta[c] - ga[i] * mean;
  }
  float2 qp = make_float2(1.0f, 0.0f);
  const bool quant = d.aq.qtable != nullptr;
  if (quant) qp = load_qparam(d.aq);
  if constexpr (R > 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int px = p_off + r * ppi;
      if (px < d.HW) gn_apply_store<V>(p, (pix0 + px) * p.C + c_base + c_off, cache[r], ga, gb, quant, qp);
    }
  } else {
    for (int px = p_off; px < d.HW; px += ppi) {
      float v[V];
      gn_load<V>(p, pix0 + px, c_base + c_off, v);
      gn_apply_store<V>(p, (pix0 + px) * p.C + c_base + c_off, v, ga, gb, quant, qp);
    }
  }
}

template <int V>
static void launch_gn(const GnP& p, dim3 grid, hipStream_t st) {
  const int iters = (p.d.HW + p.ppi - 1) / p.ppi;
  if (iters <= 8) hipLaunchKernelGGL((k_groupnorm<V, 8>), grid, dim3(256), 0, st, p);
  else if (iters <= 32) hipLaunchKernelGGL((k_groupnorm<V, 32>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_groupnorm<V, 0>), grid, dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------------
// Split form: the producing conv's epilogue already wrote per-channel {sum, sumsq} of every `seg`
// output pixels (tfmq_conv_desc.stats).  k_gn_finalize folds them into a per-(image, channel) affine
// y = A*x + Bb (A = rstd*gamma, Bb = beta - A*mean); k_gn_apply is then a pure, fully coalesced
// elementwise pass: 4 B read + 1 B written per element, nothing else.
__global__ __launch_bounds__(256) void k_gn_finalize(const float2* __restrict__ st1, int C1, const float2* __restrict__ st2,
                                                     int C2, int HW, int seg, int groups, float eps,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ A, float* __restrict__ Bb) {
  // block = (image, 8 groups); 32 lanes per group stride over its nseg x cpg partial sums, double accumulation,
  // fixed-order butterfly -> deterministic
  const int b = blockIdx.x, Cc = C1 + C2, cpg = Cc / groups, nseg = HW / seg;
  const int g = blockIdx.y * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  if (g >= groups) return;
  double s = 0.0, ss = 0.0;
  const int items = nseg * cpg;
  for (int i = l; i < items; i += 32) {
    const int sgi = i / cpg, c = g * cpg + (i - sgi * cpg);
    const size_t row = static_cast<size_t>(b) * nseg + sgi;
    const float2 v = c < C1 ? st1[row * C1 + c] : st2[row * C2 + (c - C1)];
    s += v.x;
    ss += v.y;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    ss += __shfl_xor(ss, o, 64);
  }
  const double n = static_cast<double>(HW) * cpg;
  const double mean = s / n;
  double var = ss / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float meanf = static_cast<float>(mean);
  const float rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  for (int c = g * cpg + l; c < (g + 1) * cpg; c += 32) {
    const float a = rstd * gamma[c];
    A[static_cast<size_t>(b) * Cc + c] = a;
    Bb[static_cast<size_t>(b) * Cc + c] = beta[c] - a * meanf;
  }
}

template <int V>
__global__ __launch_bounds__(256) void k_gn_apply(tfmq_gn_desc d, const float* __restrict__ A, const float* __restrict__ Bb) {
  const int Cc = d.C1 + d.C2;
  const int cv = Cc / V;
  const size_t total = static_cast<size_t>(d.B) * d.HW * cv;
  const bool quant = d.aq.qtable != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(d.aq);
  // (pixel, channel group) of this thread's items without a 64-bit division per item (two of them were a third of the
  // kernel's instructions): 32-bit index arithmetic (the launcher checks the item count), the item index advances by
  // the grid stride = sdiv pixels + smod channel groups with carry, the image index is a shift when HW is a power of two.
  const unsigned stride = gridDim.x * blockDim.x;
  const unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned ucv = static_cast<unsigned>(cv), uhw = static_cast<unsigned>(d.HW);
  const unsigned sdiv = stride / ucv, smod = stride - sdiv * ucv;
  const int hw_shift = (uhw & (uhw - 1)) == 0 ? __builtin_ctz(uhw) : -1;
  unsigned pix = i0 / ucv;
  unsigned cq = i0 - pix * ucv;
  for (unsigned i = i0; i < static_cast<unsigned>(total); i += stride) {
    const int b = static_cast<int>(hw_shift >= 0 ? pix >> hw_shift : pix / uhw);
    const int c = static_cast<int>(cq) * V;
    const float* src = c < d.C1 ? d.x1 + static_cast<size_t>(pix) * d.C1 + c : d.x2 + static_cast<size_t>(pix) * d.C2 + (c - d.C1);
    float v[V], a[V], bb[V], y[V];
    if constexpr (V == 4) {
      const float4 ta = *reinterpret_cast<const float4*>(A + static_cast<size_t>(b) * Cc + c);
      const float4 tb = *reinterpret_cast<const float4*>(Bb + static_cast<size_t>(b) * Cc + c);
      if (d.x_f16) {
        const __half* sh = c < d.C1 ? reinterpret_cast<const __half*>(d.x1) + static_cast<size_t>(pix) * d.C1 + c
                                    : reinterpret_cast<const __half*>(d.x2) + static_cast<size_t>(pix) * d.C2 + (c - d.C1);
        const uint2 u = *reinterpret_cast<const uint2*>(sh);
        const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&u.x)), hi = __half22float2(*reinterpret_cast<const __half2*>(&u.y));
        v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
      } else {
        const float4 t = *reinterpret_cast<const float4*>(src);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      }
      a[0] = ta.x; a[1] = ta.y; a[2] = ta.z; a[3] = ta.w;
      bb[0] = tb.x; bb[1] = tb.y; bb[2] = tb.z; bb[3] = tb.w;
    } else {
#pragma unroll
      for (int q = 0; q < V; ++q) {
        v[q] = d.x_f16 ? __half2float((c < d.C1 ? reinterpret_cast<const __half*>(d.x1) + static_cast<size_t>(pix) * d.C1 + c
                                                 : reinterpret_cast<const __half*>(d.x2) + static_cast<size_t>(pix) * d.C2 + (c - d.C1))[q])
                       : src[q];
        a[q] = A[static_cast<size_t>(b) * Cc + c + q];
        bb[q] = Bb[static_cast<size_t>(b) * Cc + c + q];
      }
    }
    const size_t o = static_cast<size_t>(pix) * Cc + c;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      y[q] = a[q] * v[q] + bb[q];
      if (d.silu) y[q] = silu_f(y[q]);
    }
    if (d.xcat_or_null) {
      if (d.half_out) {
        __half* xh = reinterpret_cast<__half*>(d.xcat_or_null);
#pragma unroll
        for (int q = 0; q < V; ++q) xh[o + q] = __float2half_rn(v[q]);
      } else if constexpr (V == 4) *reinterpret_cast<float4*>(d.xcat_or_null + o) = make_float4(v[0], v[1], v[2], v[3]);
      else
#pragma unroll
        for (int q = 0; q < V; ++q) d.xcat_or_null[o + q] = v[q];
    }
    if (quant) {
      signed char qq[V];
      if constexpr (V != 4) {
#pragma unroll
        for (int q = 0; q < V; ++q)
          qq[q] = static_cast<signed char>(static_cast<int>(quant_index_f(y[q], qp.x, qp.y, 255.0f)) - 128);
      }
      if constexpr (V == 4) *reinterpret_cast<char4*>(d.yq + o) = quant_char4(y[0], y[1], y[2], y[3], make_quantp(qp));
      else
#pragma unroll
        for (int q = 0; q < V; ++q) d.yq[o + q] = qq[q];
    }
    if (d.yf) {
      if (d.half_out) {
        __half* yh = reinterpret_cast<__half*>(d.yf);
#pragma unroll
        for (int q = 0; q < V; ++q) yh[o + q] = __float2half_rn(y[q]);
      } else if constexpr (V == 4) *reinterpret_cast<float4*>(d.yf + o) = make_float4(y[0], y[1], y[2], y[3]);
      else
#pragma unroll
        for (int q = 0; q < V; ++q) d.yf[o + q] = y[q];
    }
    pix += sdiv;
    cq += smod;
    if (cq >= ucv) {
      cq -= ucv;
      ++pix;
    }
    if (i + stride < i) break;      // 32-bit wrap of the item index (total close to 2^32)
  }
}

// fp16 activation stream: 8 channels per thread -- one 16-byte load, one 8-byte int8 store (and 16-byte fp16 stores for the
// optional normalised / concatenated copies); the 4-channel form moves 8 / 4 bytes per lane and is instruction-issue bound.
__global__ __launch_bounds__(256) void k_gn_apply_h8(tfmq_gn_desc d, const float* __restrict__ A, const float* __restrict__ Bb) {
  const int Cc = d.C1 + d.C2;
  const unsigned ucv = static_cast<unsigned>(Cc / 8), uhw = static_cast<unsigned>(d.HW);
  const size_t total = static_cast<size_t>(d.B) * d.HW * ucv;
  const bool quant = d.aq.qtable != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(d.aq);
  const unsigned stride = gridDim.x * blockDim.x;
  const unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned sdiv = stride / ucv, smod = stride - sdiv * ucv;
  const int hw_shift = (uhw & (uhw - 1)) == 0 ? __builtin_ctz(uhw) : -1;
  unsigned pix = i0 / ucv;
  unsigned cq = i0 - pix * ucv;
  for (unsigned i = i0; i < static_cast<unsigned>(total); i += stride) {
    const int b = static_cast<int>(hw_shift >= 0 ? pix >> hw_shift : pix / uhw);
    const int c = static_cast<int>(cq) * 8;
    const __half* sh = c < d.C1 ? reinterpret_cast<const __half*>(d.x1) + static_cast<size_t>(pix) * d.C1 + c
                                : reinterpret_cast<const __half*>(d.x2) + static_cast<size_t>(pix) * d.C2 + (c - d.C1);
    const uint4 u = *reinterpret_cast<const uint4*>(sh);
    const float4 a0 = *reinterpret_cast<const float4*>(A + static_cast<size_t>(b) * Cc + c);
    const float4 a1 = *reinterpret_cast<const float4*>(A + static_cast<size_t>(b) * Cc + c + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(Bb + static_cast<size_t>(b) * Cc + c);
    const float4 b1 = *reinterpret_cast<const float4*>(Bb + static_cast<size_t>(b) * Cc + c + 4);
    const unsigned uw[4] = {u.x, u.y, u.z, u.w};
    const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float v[8], y[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&uw[q]));
      v[2 * q] = f.x;
      v[2 * q + 1] = f.y;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      y[q] = a[q] * v[q] + bb[q];
      if (d.silu) y[q] = silu_f(y[q]);
    }
    const size_t o = static_cast<size_t>(pix) * Cc + c;
    if (d.xcat_or_null) {
      if (d.half_out) *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(d.xcat_or_null) + o) = u;     // fp16 in, fp16 copy out
      else {
        *reinterpret_cast<float4*>(d.xcat_or_null + o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(d.xcat_or_null + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    if (quant) {
      unsigned w[2];
      const QuantP qq = make_quantp(qp);
#pragma unroll
      for (int h = 0; h < 2; ++h) w[h] = quant_pack4(y[4 * h], y[4 * h + 1], y[4 * h + 2], y[4 * h + 3], qq);
      *reinterpret_cast<uint2*>(d.yq + o) = make_uint2(w[0], w[1]);
    }
    if (d.yf) {
      if (d.half_out) {
        __half2 h2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h2[q] = __floats2half2_rn(y[2 * q], y[2 * q + 1]);
        *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(d.yf) + o) = *reinterpret_cast<const uint4*>(h2);
      } else {
        *reinterpret_cast<float4*>(d.yf + o) = make_float4(y[0], y[1], y[2], y[3]);
        *reinterpret_cast<float4*>(d.yf + o + 4) = make_float4(y[4], y[5], y[6], y[7]);
      }
    }
    pix += sdiv;
    cq += smod;
    if (cq >= ucv) {
      cq -= ucv;
      ++pix;
    }
    if (i + stride < i) break;
  }
}

// The same pass with the layout of k_layernorm_hs (transformer_kernels.hip): LPR lanes share a pixel, lane j owns the 8-channel
// pieces j, j + LPR, ... of every pixel it visits, a wave walks a contiguous range of pixels and keeps the (image, channel)
// scale / shift pairs of its pieces in registers until the image changes -- one 16-byte load per 16 bytes of activation instead
// of five, no per-item index arithmetic, the next pixel group's pieces requested before this group's arithmetic.
// Per-element arithmetic unchanged (bit-identical outputs).  C <= 40 * LPR, HW % (64 / LPR) == 0.
template <int LPR>
__global__ __launch_bounds__(256, 2) void k_gn_apply_hs(tfmq_gn_desc d, const float* __restrict__ A, const float* __restrict__ Bb,
                                                        int wpi, int nslots) {
  constexpr int NCH = 5, RPW = 64 / LPR;
  const int Cc = d.C1 + d.C2, chunks = Cc >> 3;
  const int lane = threadIdx.x & 63, sub = lane / LPR, j = lane % LPR;
  // wpi waves share an image and stride through its pixel groups together (the chip sweeps the tensor front to back; a
  // contiguous range per wave = 2048 separate streams measured 20 % slower); a wave's images are slot, slot + nslots, ...
  const int gpi = d.HW / RPW;                                  // pixel groups per image
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int slot = wave / wpi, wl = wave - slot * wpi;
  if (slot >= nslots || slot >= d.B) return;
  const bool quant = d.aq.qtable != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(d.aq);
  const QuantP qq = make_quantp(qp);
  // per piece: source pointer select (virtual concat) and column offsets, fixed for the whole kernel
  int coff[NCH];
  bool ok[NCH], second[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int idx = j + LPR * k;
    ok[k] = idx < chunks;
    const int c = (ok[k] ? idx : 0) * 8;
    second[k] = c >= d.C1;
    coff[k] = c;
  }
  auto fetch = [&](long grp, uint4* dst) {
    const long pix = grp * RPW + sub;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const __half* sh = second[k] ? reinterpret_cast<const __half*>(d.x2) + pix * d.C2 + (coff[k] - d.C1)
                                   : reinterpret_cast<const __half*>(d.x1) + pix * d.C1 + coff[k];
      dst[k] = *reinterpret_cast<const uint4*>(sh);
    }
  };
  float4 a[NCH][2], bsh[NCH][2];
  int cur_b = -1;
  uint4 raw[NCH];
  int img = slot, g = wl;
  fetch(static_cast<long>(img) * gpi + g, raw);
  for (bool have = true; have;) {
    uint4 nxt[NCH];
    int ng = g + wpi, nimg = img;
    if (ng >= gpi) {
      ng = wl;
      nimg = img + nslots;
    }
    const bool nhave = nimg < d.B;
    const long grp = static_cast<long>(img) * gpi + g;
    fetch(nhave ? static_cast<long>(nimg) * gpi + ng : grp, nxt);   // (the last iteration re-reads its own pixels: no branch around loads)
    const long pix = grp * RPW + sub;
    const int b = img;
    if (b != cur_b) {
      cur_b = b;
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          a[k][u] = *reinterpret_cast<const float4*>(A + static_cast<size_t>(b) * Cc + coff[k] + 4 * u);
          bsh[k][u] = *reinterpret_cast<const float4*>(Bb + static_cast<size_t>(b) * Cc + coff[k] + 4 * u);
        }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      if (!ok[k]) continue;
      const uint4 u = raw[k];
      const unsigned uw[4] = {u.x, u.y, u.z, u.w};
      const float* ak = reinterpret_cast<const float*>(&a[k][0]);
      const float* bk = reinterpret_cast<const float*>(&bsh[k][0]);
      float v[8], y[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&uw[q]));
        v[2 * q] = f.x;
        v[2 * q + 1] = f.y;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        y[q] = ak[q] * v[q] + bk[q];
        if (d.silu) y[q] = silu_f(y[q]);
      }
      const size_t o = static_cast<size_t>(pix) * Cc + coff[k];
      if (d.xcat_or_null) {
        if (d.half_out) *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(d.xcat_or_null) + o) = u;
        else {
          *reinterpret_cast<float4*>(d.xcat_or_null + o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(d.xcat_or_null + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
      if (quant) {
        unsigned w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) w[h] = quant_pack4(y[4 * h], y[4 * h + 1], y[4 * h + 2], y[4 * h + 3], qq);
        *reinterpret_cast<uint2*>(d.yq + o) = make_uint2(w[0], w[1]);
      }
      if (d.yf) {
        if (d.half_out) {
          __half2 h2[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) h2[q] = __floats2half2_rn(y[2 * q], y[2 * q + 1]);
          *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(d.yf) + o) = *reinterpret_cast<const uint4*>(h2);
        } else {
          *reinterpret_cast<float4*>(d.yf + o) = make_float4(y[0], y[1], y[2], y[3]);
          *reinterpret_cast<float4*>(d.yf + o + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) raw[k] = nxt[k];
    img = nimg;
    g = ng;
    have = nhave;
  }
}

template <int LPR>
static void launch_gn_apply_hs(tfmq_handle h, const tfmq_gn_desc& d, const float* A, const float* Bb, hipStream_t st) {
  constexpr int RPW = 64 / LPR;
  const int gpi = d.HW / RPW;
  const int resident = h->cu_count * 2 * 4;                     // two blocks per CU, one round
  int wpi = resident / d.B;
  wpi = wpi < 1 ? 1 : (wpi > gpi ? gpi : wpi);
  int nslots = resident / wpi;
  nslots = nslots > d.B ? d.B : nslots;
  const int blocks = (nslots * wpi + 3) / 4;
  hipLaunchKernelGGL((k_gn_apply_hs<LPR>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, d, A, Bb, wpi, nslots);
}

extern "C" int tfmq_groupnorm_from_stats(tfmq_handle h, const tfmq_gn_desc* dd, const float* stats1, const float* stats2,
                                         int seg, float* ws, void* stream) {
  TFMQ_CHECK_ARG(h, h && dd && stats1 && ws, "groupnorm_from_stats: null pointer");
  const tfmq_gn_desc& d = *dd;
  TFMQ_CHECK_ARG(h, d.x1 && d.gamma && d.beta && (d.C2 == 0 || (d.x2 && stats2)), "groupnorm_from_stats: null operand");
  TFMQ_CHECK_ARG(h, (d.aq.qtable && d.yq) || d.yf, "groupnorm_from_stats: no output requested");
  TFMQ_CHECK_ARG(h, d.B > 0 && d.HW > 0 && d.groups > 0 && d.groups <= 64 && (d.C1 + d.C2) % d.groups == 0,
                 "groupnorm_from_stats: bad shape");
  TFMQ_CHECK_ARG(h, seg > 0 && d.HW % seg == 0, "groupnorm_from_stats: seg must divide HW");
  const int Cc = d.C1 + d.C2;
  float* A = ws;
  float* Bb = ws + static_cast<size_t>(d.B) * Cc;
  hipLaunchKernelGGL(k_gn_finalize, dim3(d.B, (d.groups + 7) / 8), dim3(256), 0, as_stream(stream), reinterpret_cast<const float2*>(stats1), d.C1,
                     reinterpret_cast<const float2*>(stats2), d.C2, d.HW, seg, d.groups, d.eps, d.gamma, d.beta, A, Bb);
  TFMQ_LAUNCH_CHECK(h);
  const bool v4 = (d.C1 % 4 == 0) && (d.C2 % 4 == 0);
  const bool h8 = d.x_f16 && (d.C1 % 8 == 0) && (d.C2 % 8 == 0);
  const size_t total = static_cast<size_t>(d.B) * d.HW * (h8 ? Cc / 8 : (v4 ? Cc / 4 : Cc));
  TFMQ_CHECK_ARG(h, total < (1ull << 32), "groupnorm_from_stats: more than 2^32 items");
  int blocks = ceil_div(static_cast<long>(total), 256 * 4);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  if (blocks < 1) blocks = 1;
  const int chunks8 = Cc / 8;
  const int lpr = chunks8 <= 40 ? 8 : (chunks8 <= 80 ? 16 : (chunks8 <= 160 ? 32 : (chunks8 <= 320 ? 64 : 0)));
  // measured at the SD shapes (scratch/bench_gn.py): the sub-wave kernel is 25-30 % faster where a pass reads two sources and / or
  // writes the fp16 concat copy beside its int8 output (the up path's ResBlock inputs: 4.2-5.1 TB/s against 3.0-3.6), the per-item
  // kernel 0-20 % faster on the plain single-source pass
  const bool hs = (d.C2 > 0 || d.xcat_or_null != nullptr) ? !getenv("TFMQ_GN_APPLY_ITEMS") : getenv("TFMQ_GN_APPLY_ROWS") != nullptr;
  if (h8 && lpr && d.HW % (64 / lpr) == 0 && hs) {
    if (lpr == 8) launch_gn_apply_hs<8>(h, d, A, Bb, as_stream(stream));
    else if (lpr == 16) launch_gn_apply_hs<16>(h, d, A, Bb, as_stream(stream));
    else if (lpr == 32) launch_gn_apply_hs<32>(h, d, A, Bb, as_stream(stream));
    else launch_gn_apply_hs<64>(h, d, A, Bb, as_stream(stream));
  } else if (h8) hipLaunchKernelGGL(k_gn_apply_h8, dim3(blocks), dim3(256), 0, as_stream(stream), d, A, Bb);
  else if (v4) hipLaunchKernelGGL(k_gn_apply<4>, dim3(blocks), dim3(256), 0, as_stream(stream), d, A, Bb);
  else hipLaunchKernelGGL(k_gn_apply<1>, dim3(blocks), dim3(256), 0, as_stream(stream), d, A, Bb);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_gn_finalize(tfmq_handle h, const tfmq_gn_desc* dd, const float* stats1, const float* stats2, int seg, float* a, float* bsh,
                                void* stream) {
  TFMQ_CHECK_ARG(h, h && dd && stats1 && a && bsh, "gn_finalize: null pointer");
  const tfmq_gn_desc& d = *dd;
  TFMQ_CHECK_ARG(h, d.gamma && d.beta && d.B > 0 && d.HW > 0 && d.C1 > 0 && d.C2 >= 0 && (d.C2 == 0 || stats2) && d.groups > 0 && d.groups <= 64 &&
                        (d.C1 + d.C2) % d.groups == 0 && seg > 0 && d.HW % seg == 0, "gn_finalize: bad shape");
  hipLaunchKernelGGL(k_gn_finalize, dim3(d.B, (d.groups + 7) / 8), dim3(256), 0, as_stream(stream), reinterpret_cast<const float2*>(stats1), d.C1,
                     reinterpret_cast<const float2*>(stats2), d.C2, d.HW, seg, d.groups, d.eps, d.gamma, d.beta, a, bsh);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_groupnorm(tfmq_handle h, const tfmq_gn_desc* dd, void* stream) {
  TFMQ_CHECK_ARG(h, h && dd, "groupnorm: null pointer");
  const tfmq_gn_desc& d = *dd;
  TFMQ_CHECK_ARG(h, d.x1 && d.gamma && d.beta && (d.C2 == 0 || d.x2), "groupnorm: null operand");
  TFMQ_CHECK_ARG(h, (d.aq.qtable && d.yq) || d.yf, "groupnorm: no output requested");
  TFMQ_CHECK_ARG(h, d.B > 0 && d.HW > 0 && d.C1 > 0 && d.C2 >= 0 && d.groups > 0 && d.groups <= 64, "groupnorm: bad shape");
  GnP p;
  p.d = d;
  p.C = d.C1 + d.C2;
  TFMQ_CHECK_ARG(h, p.C % d.groups == 0, "groupnorm: channels not divisible by groups");
  p.cpg = p.C / d.groups;
  int gpb = 32 / p.cpg;
  if (gpb < 1) gpb = 1;
  while (gpb > 1 && d.groups % gpb != 0) --gpb;
  p.gpb = gpb;
  p.cw = gpb * p.cpg;
  int V = 1;
  if (p.cpg % 4 == 0 && d.C1 % 4 == 0 && p.cw / 4 <= 256) V = 4;
  else if (p.cpg % 2 == 0 && d.C1 % 2 == 0 && p.cw / 2 <= 256) V = 2;
  TFMQ_CHECK_ARG(h, p.cw / V <= 256, "groupnorm: group too wide (channels per group > 1024)");
  p.tpp = p.cw / V;
  p.ppi = 256 / p.tpp;
  int lpg = 64;
  while (lpg * gpb > 256) lpg >>= 1;
  p.lpg = lpg;
  dim3 grid(d.groups / gpb, d.B);
  if (V == 4) launch_gn<4>(p, grid, as_stream(stream));
  else if (V == 2) launch_gn<2>(p, grid, as_stream(stream));
  else launch_gn<1>(p, grid, as_stream(stream));
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
