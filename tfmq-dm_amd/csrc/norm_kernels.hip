// K8: GroupNorm (+SiLU) (+8-bit quantise), reading a *virtual* channel concat of two NHWC
// tensors.  Replaces Normalize -> nonlinearity -> aqtizer (ddim/models/diffusion.py:27-33,
// 117-118,123-124; quant/quant_layer.py:223-226): 1 statistics read + 1 apply read of fp32,
// 1 write of int8, instead of the reference's ~10 elementwise passes.
//
// One block = one image x one chunk of whole groups spanning ~32 channels (128-byte rows of
// the NHWC tensor).  Statistics are accumulated in double per thread and combined in a fixed
// order (deterministic).  HBM-bound: algorithmic bytes = 4 B read (+4 B re-read, normally an
// L2/MALL hit) + 1 B written per element.
#include "common.hpp"

struct GnP {
  tfmq_gn_desc d;
  int C, cpg, gpb, cw;  // channels, channels/group, groups/block, chunk width = gpb*cpg
};

template <int V>
__device__ __forceinline__ void gn_load(const GnP& p, size_t pix, int c, float (&v)[V]) {
  const tfmq_gn_desc& d = p.d;
  // V consecutive channels never straddle the concat boundary (C1 % V == 0 is checked on the host)
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
__global__ __launch_bounds__(256) void k_groupnorm(GnP p) {
  const tfmq_gn_desc& d = p.d;
  __shared__ double s_part[256][2];
  __shared__ float s_stat[64][2];  // mean, rstd per group of this block
  const int b = blockIdx.y;
  const int c_base = blockIdx.x * p.cw;
  const int tpp = p.cw / V;           // threads per pixel
  const int ppi = 256 / tpp;          // pixels per iteration
  const int t = threadIdx.x;
  const bool active = t < tpp * ppi;
  const int c_off = (t % tpp) * V;    // channel offset inside the chunk
  const int p_off = t / tpp;
  const int g_rel = c_off / p.cpg;    // V divides cpg => all V channels share the group
  const size_t pix0 = static_cast<size_t>(b) * d.HW;

  // ---- pass 1: sum and sum of squares (double)
  double s = 0.0, ss = 0.0;
  if (active) {
    for (int px = p_off; px < d.HW; px += ppi) {
      float v[V];
      gn_load<V>(p, pix0 + px, c_base + c_off, v);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const double x = static_cast<double>(v[i]);
        s += x;
        ss += x * x;
      }
    }
  }
  s_part[t][0] = s;
  s_part[t][1] = ss;
  __syncthreads();
  if (t < p.gpb) {
    double a = 0.0, aa = 0.0;
    const int nact = tpp * ppi;
    for (int j = 0; j < nact; ++j) {
      if (((j % tpp) * V) / p.cpg == t) {
        a += s_part[j][0];
        aa += s_part[j][1];
      }
    }
    const double n = static_cast<double>(d.HW) * p.cpg;
    const double mean = a / n;
    double var = aa / n - mean * mean;
    if (var < 0.0) var = 0.0;
    s_stat[t][0] = static_cast<float>(mean);
    s_stat[t][1] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(d.eps)));
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
  for (int px = p_off; px < d.HW; px += ppi) {
    float v[V];
    gn_load<V>(p, pix0 + px, c_base + c_off, v);
    const size_t o = (pix0 + px) * p.C + c_base + c_off;
    if (d.xcat_or_null) {
#pragma unroll
      for (int i = 0; i < V; ++i) d.xcat_or_null[o + i] = v[i];
    }
    float y[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      y[i] = ga[i] * v[i] + gb[i];
      if (d.silu) y[i] = silu_f(y[i]);
    }
    if (quant) {
      signed char q[V];
#pragma unroll
      for (int i = 0; i < V; ++i) q[i] = static_cast<signed char>(static_cast<int>(quant_index_f(y[i], qp.x, qp.y, 255.0f)) - 128);
      if constexpr (V == 4) {
        *reinterpret_cast<char4*>(d.yq + o) = make_char4(q[0], q[1], q[2], q[3]);
      } else if constexpr (V == 2) {
        *reinterpret_cast<char2*>(d.yq + o) = make_char2(q[0], q[1]);
      } else {
        d.yq[o] = q[0];
      }
    }
    if (d.yf) {
#pragma unroll
      for (int i = 0; i < V; ++i) d.yf[o + i] = y[i];
    }
  }
}

extern "C" int tfmq_groupnorm(tfmq_handle h, const tfmq_gn_desc* dd, void* stream) {
  TFMQ_CHECK_ARG(h, h && dd, "groupnorm: null pointer");
  const tfmq_gn_desc& d = *dd;
  TFMQ_CHECK_ARG(h, d.x1 && d.gamma && d.beta && (d.C2 == 0 || d.x2), "groupnorm: null operand");
  TFMQ_CHECK_ARG(h, (d.aq.qtable && d.yq) || d.yf, "groupnorm: no output requested");
  TFMQ_CHECK_ARG(h, d.B > 0 && d.HW > 0 && d.C1 > 0 && d.C2 >= 0 && d.groups > 0, "groupnorm: bad shape");
  GnP p;
  p.d = d;
  p.C = d.C1 + d.C2;
  TFMQ_CHECK_ARG(h, p.C % d.groups == 0, "groupnorm: channels not divisible by groups");
  p.cpg = p.C / d.groups;
  int gpb = 32 / p.cpg;
  if (gpb < 1) gpb = 1;
  while (gpb > 1 && (d.groups % gpb != 0 || gpb > 64)) --gpb;
  // keep the chunk width <= 256 threads' worth even for very wide groups
  p.gpb = gpb;
  p.cw = gpb * p.cpg;
  int V = 1;
  if (p.cpg % 4 == 0 && d.C1 % 4 == 0 && p.cw / 4 <= 256) V = 4;
  else if (p.cpg % 2 == 0 && d.C1 % 2 == 0 && p.cw / 2 <= 256) V = 2;
  TFMQ_CHECK_ARG(h, p.cw / V <= 256, "groupnorm: group too wide (channels per group > 1024)");
  dim3 grid(d.groups / gpb, d.B);
  if (V == 4) hipLaunchKernelGGL(k_groupnorm<4>, grid, dim3(256), 0, as_stream(stream), p);
  else if (V == 2) hipLaunchKernelGGL(k_groupnorm<2>, grid, dim3(256), 0, as_stream(stream), p);
  else hipLaunchKernelGGL(k_groupnorm<1>, grid, dim3(256), 0, as_stream(stream), p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
