// K10: fused softmax(Q K^T * scale) V on un-quantised q,k,v (QuantAttnBlock.forward,
// quant/quant_block.py:483-500; the attention-matmul quantizers are never enabled, SURVEY §0
// fact 2).  Flash-style: the [Tq x Tk] score matrix never reaches HBM (the reference
// materialises it in fp32, e.g. 8 x 4096^2 x 4 B = 537 MB per SD sample).
//
// One block = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
//   S^T[key][q]  = mfma_f32_32x32x16_f16(A = K tile, B = Q tile)  -> a lane holds 16 keys of ONE query,
//                  so the row max / row sum are 15 in-lane ops + one exchange with lane^32;
//   O^T[dcol][q] += mfma(A = V^T tile, B = P^T) -> P^T fragments are the lane's own S^T registers
//                  (regs 8s..8s+7 for k-step s) and the running rescale is lane-local.
// K tile row-major f16 (+16 B row pad), V staged transposed (+8 B row pad): all fragment
// reads are conflict-free ds_read_b128 / ds_read_b64.  q,k,v arrive as fp32 (outputs of the
// w4a8 projection GEMMs) and are converted to f16 while staging; accumulation is fp32.
#include "common.hpp"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));

struct AttnP {
  const float *q, *k, *v;
  int ldq, ldk, ldv;
  float* out;
  int ldo;
  int8_t* yq;
  tfmq_qsel aq;
  int B, heads, Tq, Tk, d;
  float scale;
};

template <int DPAD>
__global__ __launch_bounds__(256) void k_attention(AttnP p) {
  constexpr int QROW = DPAD * 2 + 16;   // bytes per Q / K row in LDS
  constexpr int VROW = 32 * 2 + 8;      // bytes per V^T row (32 keys)
  constexpr int NT = DPAD / 32;         // output column tiles
  constexpr int NKS = DPAD / 16;        // k-steps of the score MFMA
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sQ = smem;                       // [128][QROW]
  unsigned char* sK = smem + 128 * QROW;          // [32][QROW]
  unsigned char* sV = sK + 32 * QROW;             // [DPAD][VROW]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y / p.heads, hd = blockIdx.y % p.heads;
  const int q0 = blockIdx.x * 128;
  const int d = p.d;

  // ---- stage Q (128 x DPAD) as f16
  for (int idx = tid; idx < 128 * (DPAD / 4); idx += 256) {
    const int row = idx / (DPAD / 4), c4 = (idx % (DPAD / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + row < p.Tq && c4 < d)
      v = *reinterpret_cast<const float4*>(p.q + (static_cast<size_t>(b) * p.Tq + q0 + row) * p.ldq + hd * d + c4);
    v4h hv = {static_cast<_Float16>(v.x), static_cast<_Float16>(v.y), static_cast<_Float16>(v.z), static_cast<_Float16>(v.w)};
    *reinterpret_cast<v4h*>(sQ + row * QROW + c4 * 2) = hv;
  }

  float m_run = -INFINITY, l_run = 0.0f;
  v16f o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;

  const int ntiles = (p.Tk + 31) / 32;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();  // previous tile fully consumed (also orders the Q staging before the first read)
    for (int idx = tid; idx < 32 * (DPAD / 4); idx += 256) {
      const int key = idx / (DPAD / 4), c4 = (idx % (DPAD / 4)) * 4;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      const int kg = kt * 32 + key;
      if (kg < p.Tk && c4 < d) {
        kv = *reinterpret_cast<const float4*>(p.k + (static_cast<size_t>(b) * p.Tk + kg) * p.ldk + hd * d + c4);
        vv = *reinterpret_cast<const float4*>(p.v + (static_cast<size_t>(b) * p.Tk + kg) * p.ldv + hd * d + c4);
      }
      v4h hk = {static_cast<_Float16>(kv.x), static_cast<_Float16>(kv.y), static_cast<_Float16>(kv.z), static_cast<_Float16>(kv.w)};
      *reinterpret_cast<v4h*>(sK + key * QROW + c4 * 2) = hk;
      *reinterpret_cast<_Float16*>(sV + (c4 + 0) * VROW + key * 2) = static_cast<_Float16>(vv.x);
      *reinterpret_cast<_Float16*>(sV + (c4 + 1) * VROW + key * 2) = static_cast<_Float16>(vv.y);
      *reinterpret_cast<_Float16*>(sV + (c4 + 2) * VROW + key * 2) = static_cast<_Float16>(vv.z);
      *reinterpret_cast<_Float16*>(sV + (c4 + 3) * VROW + key * 2) = static_cast<_Float16>(vv.w);
    }
    __syncthreads();

    // ---- S^T = K Q^T
    v16f s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const v8h a = *reinterpret_cast<const v8h*>(sK + j * QROW + (ks * 16 + hh * 8) * 2);
      const v8h bq = *reinterpret_cast<const v8h*>(sQ + (wid * 32 + j) * QROW + (ks * 16 + hh * 8) * 2);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, s, 0, 0, 0);
    }
    // ---- online softmax over this lane's 16 keys (+ partner lane^32)
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      s[r] = key < p.Tk ? s[r] * p.scale : -INFINITY;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = expf(s[r] - m_new);
      rs += s[r];
    }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      v8h bp;
#pragma unroll
      for (int e = 0; e < 8; ++e) bp[e] = static_cast<_Float16>(s[8 * s2 + e]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const unsigned char* vr = sV + (t * 32 + j) * VROW + (16 * s2 + 4 * hh) * 2;
        const v4h lo = *reinterpret_cast<const v4h*>(vr);
        const v4h hi = *reinterpret_cast<const v4h*>(vr + 16);
        v8h a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bp, o[t], 0, 0, 0);
      }
    }
  }

  // ---- normalise and store: lane (query j, half hh) owns dcols t*32 + (r&3) + 8*(r>>2) + 4*hh
  const int qg = q0 + wid * 32 + j;
  if (qg >= p.Tq) return;
  const float inv = 1.0f / l_run;
  const bool quant = p.yq != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(p.aq);
  const size_t tok = static_cast<size_t>(b) * p.Tq + qg;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dc = t * 32 + 8 * g + 4 * hh;
      if (dc >= d) continue;
      float4 v = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
      if (p.out) *reinterpret_cast<float4*>(p.out + tok * p.ldo + hd * d + dc) = v;
      if (quant) {
        char4 c;
        c = quant_char4(v.x, v.y, v.z, v.w, make_quantp(qp));
        *reinterpret_cast<char4*>(p.yq + tok * (static_cast<size_t>(p.heads) * d) + hd * d + dc) = c;
      }
    }
  }
}

template <int DPAD>
static int launch_attn(tfmq_handle h, const AttnP& p, void* stream) {
  constexpr size_t smem = 160 * (DPAD * 2 + 16) + static_cast<size_t>(DPAD) * (32 * 2 + 8);
  static bool configured = false;
  if (!configured) {
    TFMQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention<DPAD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
    configured = true;
  }
  dim3 grid((p.Tq + 127) / 128, p.B * p.heads);
  hipLaunchKernelGGL(k_attention<DPAD>, grid, dim3(256), smem, as_stream(stream), p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_attention(tfmq_handle h, const float* q, const float* k, const float* v, int ldq, int ldk, int ldv,
                              float* out, int ldo, int8_t* yq, tfmq_qsel aq, int B, int heads, int Tq, int Tk, int d,
                              float scale, void* stream) {
  TFMQ_CHECK_ARG(h, h && q && k && v && (out || yq), "attention: null pointer");
  TFMQ_CHECK_ARG(h, B > 0 && heads > 0 && Tq > 0 && Tk > 0 && d > 0, "attention: bad shape");
  TFMQ_CHECK_ARG(h, d % 4 == 0 && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && (!out || ldo % 4 == 0),
                 "attention: head dim and leading dims must be multiples of 4");
  TFMQ_CHECK_ARG(h, !yq || aq.qtable, "attention: quantised output needs a qparam");
  TFMQ_CHECK_ARG(h, static_cast<long>(B) * heads < 65536, "attention: B*heads must be < 65536");
  AttnP p{q, k, v, ldq, ldk, ldv, out, ldo, yq, aq, B, heads, Tq, Tk, d, scale};
  if (d <= 32) return launch_attn<32>(h, p, stream);
  if (d <= 64) return launch_attn<64>(h, p, stream);
  if (d <= 96) return launch_attn<96>(h, p, stream);
  if (d <= 128) return launch_attn<128>(h, p, stream);
  if (d <= 160) return launch_attn<160>(h, p, stream);
  if (d <= 256) return launch_attn<256>(h, p, stream);
  if (h) h->err = "attention: head dim > 256 not supported yet";
  return TFMQ_ERR_UNSUPPORTED;
}
