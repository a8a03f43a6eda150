// K10q8 (round 4): the attention of a block whose matmul quantizers are LIVE, on the int8 matrix cores.
//
//   sim  = aqtizer_q(q) aqtizer_k(k)^T * scale          (cross_attn_forward, QuantAttnBlock.forward, QuantQKMatMul with use_aq:
//   attn = softmax(sim)                                   reference quant/quant_block.py:226-243, 318-323, 483-500)
//   out  = aqtizer_w(attn) aqtizer_v(v)                   (aqtizer_w: zero point 0, `always_zero`; QuantSMVMatMul :350-351)
//
// Both products are sums of (bin - zero point) pairs: exact in int32.  v_mfma_i32_32x32x32_i8 takes signed bytes, the operands are
// bins - 128, and the zero points come back as rank-one corrections:
//   sum_d (bq - zq)(bk - zk) = sum_d x y + ck sum_d x + cq sum_d y + d cq ck,          x = bq - 128, y = bk - 128, cq = 128 - zq, ck = 128 - zk
//   sum_k  bw (bv - zv)      = sum_k u y + cv sum_k u + 128 sum_k y + 128 cv K,         u = bw - 128, y = bv - 128, cv = 128 - zv
// The softmax bins need the row's FINAL normaliser (bw = rint(p / delta_w) is not rescalable), hence two passes over the keys: pass 1 the
// running maximum / sum of exp2 in fp32 from the exact integer scores, pass 2 the scores again, p = exp2(s - m) / l, the bins, and the
// second product.  The score accumulator of a lane holds 16 CONSECUTIVE keys (K rows staged in conv_common.hpp's lin_brow order), so its
// 16 softmax bytes ARE the B operand of the P V MFMA (the trick of ff_fused.hip); V arrives transposed ([B][heads d][keys], int8).
// Functional kernel of a diagnostics path (no driver of the reference switches these quantizers on): 128 queries per workgroup, 32-key
// tiles, one staging buffer.  Against ops.attention_quant (fp32 products of the DEquantised values) the integer sums are the exact ones;
// bins agree up to values on a rounding boundary (tests/test_attention_q8_gpu.py, fixture F21).
#include "conv_common.hpp"
#include <type_traits>

namespace {

struct AttnQ8P {
  const int8_t *q, *k, *vt;
  int ldq, ldk;
  int B, heads, Tq, Tk, Tks, d;
  float scale;
  tfmq_qsel aq, ak, av, aw;
  int w_level;
  float* out;
  int ldo;
};

__device__ __forceinline__ int sum_bytes16(const v4i& v) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_sdot4(v[i], 0x01010101, s, false);
  return s;
}

template <int NKS>
__global__ __launch_bounds__(256, 2) void k_attention_q8(AttnQ8P p) {
  constexpr int KROW = NKS * 32 + 16, DPAD = NKS * 32, VROW = 32 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char sK[32 * KROW];
  __shared__ __attribute__((aligned(16))) unsigned char sV[DPAD * VROW];
  __shared__ int sKsum[32];
  __shared__ int sVsum[DPAD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  const int nqb = (p.Tq + 127) / 128;
  const int bh = blockIdx.x / nqb, b = bh / p.heads, hd = bh - b * p.heads;
  const int q0 = (blockIdx.x - bh * nqb) * 128;
  const int d = p.d;
  const float2 qq = load_qparam(p.aq), qk = load_qparam(p.ak), qv = load_qparam(p.av), qw = load_qparam(p.aw);
  const int cq = 128 - static_cast<int>(qq.y), ck = 128 - static_cast<int>(qk.y), cv = 128 - static_cast<int>(qv.y);
  const float c2 = qq.x * qk.x * p.scale * 1.44269504088896340736f;       // exp(sim) = exp2(c2 * integer score)

  for (int i = tid; i < 32 * KROW / 4; i += 256) reinterpret_cast<int*>(sK)[i] = 0;
  for (int i = tid; i < DPAD * VROW / 4; i += 256) reinterpret_cast<int*>(sV)[i] = 0;

  // ---- Q fragments (B operand of S^T = K Q^T): lane (query j, half hh) holds bytes 32 ks + 16 hh .. + 15 of its row, zeros beyond d
  const int qrow = q0 + wid * 32 + j;
  const bool qok = qrow < p.Tq;
  v4i qf[NKS];
  int qsum = 0;
  {
    const int8_t* qp = p.q + (static_cast<size_t>(b) * p.Tq + (qok ? qrow : 0)) * p.ldq + hd * d;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      uint2 lo = make_uint2(0, 0), hi = make_uint2(0, 0);
      const int c = 32 * ks + 16 * hh;
      if (c < d) lo = *reinterpret_cast<const uint2*>(qp + c);
      if (c + 8 < d) hi = *reinterpret_cast<const uint2*>(qp + c + 8);
      qf[ks] = v4i{static_cast<int>(lo.x), static_cast<int>(lo.y), static_cast<int>(hi.x), static_cast<int>(hi.y)};
      qsum += sum_bytes16(qf[ks]);
    }
    qsum += __shfl_xor(qsum, 32, 64);
  }
  const int8_t* kb = p.k + static_cast<size_t>(b) * p.Tk * p.ldk + hd * d;
  const int8_t* vb = p.vt + (static_cast<size_t>(b) * p.heads + hd) * d * p.Tks;
  const int ntiles = (p.Tk + 31) / 32;

  // one 32-key tile into LDS: K rows (8-byte pieces) at lin_brow^-1 ... row R of the buffer feeds MFMA row R, i.e. holds key lin_brow(R);
  // V^T rows (channels) x 32 keys; per-key sums of the K bytes
  auto stage = [&](int kt, bool with_v) {
    const int k0 = kt * 32;
    for (int i = tid; i < 32 * (d / 8); i += 256) {
      const int R = i / (d / 8), pc = i - R * (d / 8);
      const int key = k0 + lin_brow(R);
      uint2 v = make_uint2(0, 0);
      if (key < p.Tk) v = *reinterpret_cast<const uint2*>(kb + static_cast<size_t>(key) * p.ldk + pc * 8);
      *reinterpret_cast<uint2*>(sK + R * KROW + pc * 8) = v;
    }
    if (with_v) {
      for (int i = tid; i < d * 4; i += 256) {
        const int c = i >> 2, pc = i & 3;
        uint2 v = make_uint2(0, 0);
        if (k0 + pc * 8 < p.Tks) v = *reinterpret_cast<const uint2*>(vb + static_cast<size_t>(c) * p.Tks + k0 + pc * 8);
        *reinterpret_cast<uint2*>(sV + c * VROW + pc * 8) = v;
      }
    }
    __syncthreads();
    if (tid < 32) {
      int s = 0;
      for (int i = 0; i < d / 4; ++i) s = __builtin_amdgcn_sdot4(*reinterpret_cast<const int*>(sK + tid * KROW + 4 * i), 0x01010101, s, false);
      sKsum[tid] = s;
    }
    __syncthreads();
  };
  // the exact integer scores of this lane's 16 keys (register r = key k0 + 16 hh + r), as the exp2 argument; padding -> -inf
  auto scores = [&](int kt, float (&sf)[16]) {
    v16i acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const v4i a = *reinterpret_cast<const v4i*>(sK + j * KROW + 32 * ks + 16 * hh);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[ks], acc, 0, 0, 0);
    }
    const int base = ck * qsum + d * cq * ck;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // accumulator register r of lane half hh = MFMA row 8 (r >> 2) + 4 hh + (r & 3) = buffer row R -> key lin_brow(R) = 16 hh + r
      const int R = 8 * (r >> 2) + 4 * hh + (r & 3);
      const int key = kt * 32 + 16 * hh + r;
      const int st = acc[r] + base + cq * sKsum[R];
      sf[r] = key < p.Tk ? static_cast<float>(st) * c2 : -INFINITY;
    }
  };

  // ---- pass 1: the row's maximum and normaliser
  float m_run = -INFINITY, l_run = 0.0f;
  for (int kt = 0; kt < ntiles; ++kt) {
    stage(kt, false);
    float sf[16];
    scores(kt, sf);
    float mx = sf[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sf[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) rs += __builtin_amdgcn_exp2f(sf[r] - m_new);
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + rs;
    m_run = m_new;
    __syncthreads();
  }
  const float inv_l = 1.0f / l_run;

  // ---- pass 2: softmax bins and the second product
  v16i o[NKS];
#pragma unroll
  for (int t = 0; t < NKS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0;
  int usum = 0, vsum = 0;
  const float lw = static_cast<float>(p.w_level - 1);
  for (int kt = 0; kt < ntiles; ++kt) {
    stage(kt, true);
    if (tid < d) {
#pragma unroll
      for (int i = 0; i < 8; ++i) vsum = __builtin_amdgcn_sdot4(*reinterpret_cast<const int*>(sV + tid * VROW + 4 * i), 0x01010101, vsum, false);
    }
    float sf[16];
    scores(kt, sf);
    unsigned pw[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned w = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pr = __builtin_amdgcn_exp2f(sf[4 * g + e] - m_run) * inv_l;
        const int bw = static_cast<int>(quant_index_f(pr, qw.x, qw.y, lw));             // (padding: p = 0 -> bin = zero point = 0)
        usum += bw - 128;
        w |= (static_cast<unsigned>(bw - 128) & 0xffu) << (8 * e);
      }
      pw[g] = w;
    }
    const v4i pb = v4i{static_cast<int>(pw[0]), static_cast<int>(pw[1]), static_cast<int>(pw[2]), static_cast<int>(pw[3])};
#pragma unroll
    for (int t = 0; t < NKS; ++t) {
      const v4i a = *reinterpret_cast<const v4i*>(sV + (t * 32 + j) * VROW + 16 * hh);
      o[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, pb, o[t], 0, 0, 0);
    }
    __syncthreads();
  }
  usum += __shfl_xor(usum, 32, 64);
  if (tid < DPAD) sVsum[tid] = tid < d ? vsum : 0;
  __syncthreads();
  if (!qok) return;
  // ---- out = delta_w delta_v * (sum u y + cv sum u + 128 sum y + 128 cv K): lane (query j, half hh) owns channels 32 t + 8 g + 4 hh + (0 .. 3)
  const float so = qw.x * qv.x;
  const int kall = ntiles * 32;
  float* orow = p.out + (static_cast<size_t>(b) * p.Tq + qrow) * p.ldo + hd * d;
#pragma unroll
  for (int t = 0; t < NKS; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * t + 8 * g + 4 * hh;
      if (c >= d) continue;
      float4 v;
      float* vp = reinterpret_cast<float*>(&v);
#pragma unroll
      for (int e = 0; e < 4; ++e) vp[e] = so * static_cast<float>(o[t][4 * g + e] + cv * usum + 128 * sVsum[c + e] + 128 * cv * kall);
      *reinterpret_cast<float4*>(orow + c) = v;
    }
}

// int8 [B][T][C] -> [B][C][Tp] (Tp >= T, zero beyond T): the V^T operand
__global__ __launch_bounds__(256) void k_transpose_i8(const int8_t* __restrict__ x, int8_t* __restrict__ y, int T, int C, int Tp) {
  __shared__ int8_t tile[64][65];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int tt = i >> 6, cc = i & 63;
    tile[tt][cc] = (t0 + tt < T && c0 + cc < C) ? x[(static_cast<size_t>(b) * T + t0 + tt) * C + c0 + cc] : static_cast<int8_t>(0);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int cc = i >> 6, tt = i & 63;
    if (c0 + cc < C && t0 + tt < Tp) y[(static_cast<size_t>(b) * C + c0 + cc) * Tp + t0 + tt] = tile[tt][cc];
  }
}

}  // namespace

extern "C" int tfmq_transpose_i8(tfmq_handle h, const int8_t* x, int8_t* y, int B, int T, int C, int Tp, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && y && B > 0 && T > 0 && C > 0 && Tp >= T, "transpose_i8: bad argument");
  hipLaunchKernelGGL(k_transpose_i8, dim3((Tp + 63) / 64, (C + 63) / 64, B), dim3(256), 0, as_stream(stream), x, y, T, C, Tp);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_attention_q8(tfmq_handle h, const int8_t* q, const int8_t* k, const int8_t* vt, int ldq, int ldk, tfmq_qsel aq_q, tfmq_qsel aq_k,
                                 tfmq_qsel aq_v, tfmq_qsel aq_w, int w_level, float* out, int ldo, int B, int heads, int Tq, int Tk, int Tk_stride,
                                 int d, float scale, void* stream) {
  TFMQ_CHECK_ARG(h, h && q && k && vt && out, "attention_q8: null pointer");
  TFMQ_CHECK_ARG(h, B > 0 && heads > 0 && Tq > 0 && Tk > 0 && d > 0 && d % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && Tk_stride % 8 == 0 && Tk_stride >= Tk &&
                        ldo % 4 == 0, "attention_q8: head dim, leading dimensions and Tk_stride are multiples of 8, Tk_stride >= Tk");
  TFMQ_CHECK_ARG(h, aq_q.qtable && aq_k.qtable && aq_v.qtable && aq_w.qtable && w_level >= 2 && w_level <= 256,
                 "attention_q8: the four quantizers are required; softmax levels 2 ... 256 (wider softmax quantizers: tfmq_fake_quant_sel path)");
  if (d > 160) {
    h->err = "attention_q8: head dim > 160 not built (ops.attention_quant's functional path)";
    return TFMQ_ERR_UNSUPPORTED;
  }
  AttnQ8P p{q, k, vt, ldq, ldk, B, heads, Tq, Tk, Tk_stride, d, scale, aq_q, aq_k, aq_v, aq_w, w_level, out, ldo};
  const dim3 grid(static_cast<unsigned>((Tq + 127) / 128) * B * heads);
  hipStream_t st = as_stream(stream);
  if (d <= 32) hipLaunchKernelGGL((k_attention_q8<1>), grid, dim3(256), 0, st, p);
  else if (d <= 64) hipLaunchKernelGGL((k_attention_q8<2>), grid, dim3(256), 0, st, p);
  else if (d <= 96) hipLaunchKernelGGL((k_attention_q8<3>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_attention_q8<5>), grid, dim3(256), 0, st, p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
