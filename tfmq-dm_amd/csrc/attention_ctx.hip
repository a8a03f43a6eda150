// K10c (round 4): cross attention over a SHORT context (the 77 CLIP tokens of Stable Diffusion, stored padded to 80 keys) with the
// int8 output of to_out's activation quantizer -- CrossAttention.forward with `context`, ldm/modules/attention.py:168-194, as
// QuantBasicTransformerBlock runs it (quant/quant_block.py:226-243; the attention matmuls stay un-quantised).
//
// k_attention_h (attention_f16.hip) gives every (batch, head, 128 queries) its own workgroup: with 77 keys that is two key tiles of work
// behind a zero-fill, a staging pass and a prologue, and 4-byte output stores scattered over the token row -- 345 us at the 64 x 64 level of
// SD (UNet batch 128) for 0.5 GB of traffic and 54 GFLOP.  Here a workgroup keeps its 128 queries for ALL heads:
//   * the context's K_h (keys x d) and V_h^T (d x keys) of the next head are loaded while this head is computed (registers -> LDS, one
//     buffer, two barriers per head); K rows sit at swap_bits_2_3(key) so that a lane's score registers hold 8 consecutive keys per 16-key
//     P V step (attention_f16.hip) and V^T is read in its natural layout;
//   * all <= 96 keys are in the accumulators at once: one exact softmax, no running maximum, no rescale;
//   * the quantised output of every head lands in an LDS image of the block's 128 token rows and leaves as whole rows (16-byte pieces).
// Same mathematics and operand precision as k_attention_h (fp16 operands, fp32 accumulation, exp2 with the scale folded in, P rounded to
// fp16 for the P V product); the softmax denominator is summed in fp32 from the unrounded probabilities.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

namespace {

struct AttnCtxP {
  const __half *q, *k, *vt;
  int ldq, ldk;
  int8_t* yq;
  tfmq_qsel aq;
  int B, heads, Tq, Tk, Tks;
  float scale;
  long q_bs, q_hs;       // element strides of q: batch, head (token stride = ldq): [B][T][heads * d] rows
};

template <int D>
struct CtxGeo {
  static constexpr int NKS = (D + 15) / 16, NRT = (D + 31) / 32, DPAD = NRT * 32;
  static constexpr int KROW = NKS * 32 + 16, VROW = 96 * 2 + 16;
  static constexpr int KBYTES = 96 * KROW, VBYTES = DPAD * VROW;
  static constexpr int KPIECES = 96 * (D / 8), VPIECES = D * 12;         // 16-byte pieces staged per head (keys / rows beyond the data: zeros)
  static constexpr int KPT = (KPIECES + 255) / 256, VPT = (VPIECES + 255) / 256;
};

template <int D>
__global__ __launch_bounds__(256, 2) void k_attention_ctx(AttnCtxP p) {
  using G = CtxGeo<D>;
  constexpr int NKS = G::NKS, NRT = G::NRT, KROW = G::KROW, VROW = G::VROW, KPT = G::KPT, VPT = G::VPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = smem + G::KBYTES;
  unsigned char* sO = sV + G::VBYTES;                    // [128 tokens][heads * D] int8
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  const int nqb = p.Tq / 128;
  const int b = blockIdx.x / nqb, q0 = (blockIdx.x - b * nqb) * 128;
  const int Cc = p.heads * D;

  // ---- staging plan (the same for every head): K piece -> (key, d-piece), V^T piece -> (row, key-piece); zeros outside the data
  int k_go[KPT], k_lo[KPT], v_go[VPT], v_lo[VPT];
  bool k_ok[KPT], v_ok[VPT], k_in[KPT], v_in[VPT];
#pragma unroll
  for (int it = 0; it < KPT; ++it) {
    const int idx = tid + it * 256;
    const int key = idx / (D / 8), pc = idx - key * (D / 8);
    const int row = (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1);      // swap bits 2 and 3 (inside each 32-key sub-tile)
    k_in[it] = idx < G::KPIECES;
    k_ok[it] = k_in[it] && key < p.Tks;
    k_go[it] = key * p.ldk + pc * 8;
    k_lo[it] = row * KROW + pc * 16;
  }
#pragma unroll
  for (int it = 0; it < VPT; ++it) {
    const int idx = tid + it * 256;
    const int dc = idx / 12, kp = idx - dc * 12;
    v_in[it] = idx < G::VPIECES;
    v_ok[it] = v_in[it] && kp * 8 < p.Tks;
    v_go[it] = dc * p.Tks + kp * 8;
    v_lo[it] = dc * VROW + kp * 16;
  }
  uint4 kreg[KPT], vreg[VPT];
  auto load_head = [&](int hd) {
    const __half* kb = p.k + static_cast<size_t>(b) * p.Tks * p.ldk + hd * D;
    const __half* vb = p.vt + (static_cast<size_t>(b) * p.heads + hd) * D * p.Tks;
#pragma unroll
    for (int it = 0; it < KPT; ++it) kreg[it] = k_ok[it] ? *reinterpret_cast<const uint4*>(kb + k_go[it]) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int it = 0; it < VPT; ++it) vreg[it] = v_ok[it] ? *reinterpret_cast<const uint4*>(vb + v_go[it]) : make_uint4(0, 0, 0, 0);
  };
  auto store_head = [&]() {
#pragma unroll
    for (int it = 0; it < KPT; ++it)
      if (k_in[it]) *reinterpret_cast<uint4*>(sK + k_lo[it]) = kreg[it];
#pragma unroll
    for (int it = 0; it < VPT; ++it)
      if (v_in[it]) *reinterpret_cast<uint4*>(sV + v_lo[it]) = vreg[it];
  };
  const __half* qrow = p.q + static_cast<size_t>(b) * p.q_bs + static_cast<size_t>(q0 + wid * 32 + j) * p.ldq;
  auto load_q = [&](int hd, v8h (&qf)[NKS]) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const int c = ks * 16 + hh * 8;
      if (c < D) v = *reinterpret_cast<const uint4*>(qrow + static_cast<size_t>(hd) * p.q_hs + c);
      qf[ks] = *reinterpret_cast<v8h*>(&v);
    }
  };

  // zero the padding once (K columns D .. 16 NKS and the row pads, V^T rows D .. DPAD and the key pads): the staging never touches it
  for (int i = tid; i < (G::KBYTES + G::VBYTES) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  load_head(0);
  v8h qf[NKS], qn[NKS];
  load_q(0, qf);
  __syncthreads();
  store_head();
  __syncthreads();

  const float c2 = p.scale * 1.44269504088896340736f;
  const float2 qp = load_qparam(p.aq);
  const QuantP qq = make_quantp(qp);
  for (int hd = 0; hd < p.heads; ++hd) {
    if (hd + 1 < p.heads) {
      load_head(hd + 1);
      load_q(hd + 1, qn);
    }
    // ---- S^T = K Q^T: three 32-key sub-tiles in the accumulators
    v16f s[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const v8h a = *reinterpret_cast<const v8h*>(sK + (kt * 32 + j) * KROW + (ks * 16 + hh * 8) * 2);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], s[kt], 0, 0, 0);
      }
    }
    // register r of sub-tile kt holds key kt * 32 + 16 (r >> 3) + 8 hh + (r & 7); keys >= Tk are padding
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + 16 * (r >> 3) + 8 * hh + (r & 7);
        if (key >= p.Tk) s[kt][r] = -INFINITY;
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * c2;
    float rs = 0.0f;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c2, -mc));
        rs += s[kt][r];
      }
    rs += __shfl_xor(rs, 32, 64);
    // ---- O^T = V^T P^T: six k-steps of 16 keys
    v16f o[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      v8h bp;
#pragma unroll
      for (int e = 0; e < 8; ++e) bp[e] = static_cast<_Float16>(s[u >> 1][8 * (u & 1) + e]);
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
        const v8h a = *reinterpret_cast<const v8h*>(sV + (t * 32 + j) * VROW + (16 * u + 8 * hh) * 2);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bp, o[t], 0, 0, 0);
      }
    }
    // ---- normalise, quantise: lane (query j, half hh) owns channels t * 32 + 8 g + 4 hh + (0 .. 3)
    const float inv = 1.0f / rs;
    unsigned char* orow = sO + (wid * 32 + j) * Cc + hd * D;
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dc = t * 32 + 8 * g + 4 * hh;
        if (dc >= D) continue;
        *reinterpret_cast<unsigned*>(orow + dc) = quant_pack4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv, qq);
      }
    if (hd + 1 < p.heads) {
      __syncthreads();               // every wave is through with this head's K / V^T
      store_head();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) qf[ks] = qn[ks];
      __syncthreads();
    }
  }
  __syncthreads();
  // ---- the block's 128 token rows out, 16 bytes per thread and step
  const int ppr = Cc / 16;
  int8_t* yb = p.yq + (static_cast<size_t>(b) * p.Tq + q0) * Cc;
  for (int i = tid; i < 128 * ppr; i += 256) reinterpret_cast<uint4*>(yb)[i] = reinterpret_cast<const uint4*>(sO)[i];
}

template <int D>
int launch_ctx(tfmq_handle h, const AttnCtxP& p, hipStream_t st) {
  using G = CtxGeo<D>;
  const size_t smem = static_cast<size_t>(G::KBYTES) + G::VBYTES + 128 * static_cast<size_t>(p.heads) * D;
  static bool configured = false;
  if (!configured) {
    TFMQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_ctx<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    configured = true;
  }
  hipLaunchKernelGGL((k_attention_ctx<D>), dim3(static_cast<unsigned>(p.B * (p.Tq / 128))), dim3(256), smem, st, p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

}  // namespace

// Called by tfmq_attention_f16 (attention_f16.hip) for the launches this kernel takes; < 0: not taken.
int launch_attention_ctx(tfmq_handle h, const uint16_t* q, const uint16_t* k, const uint16_t* vt, int ldq, int ldk, int8_t* yq, tfmq_qsel aq, int B,
                         int heads, int Tq, int Tk, int Tks, int d, float scale, void* stream, bool* taken) {
  *taken = false;
  // TFMQ_ATTN_CTX: 0 off, 2 also d = 80 (read per call: tests/test_attention_f16_gpu.py switches it inside one process; a getenv is noise beside
  // a launch).  `tfmq_attention_f16` documents that the kernel -- hence the last fp16 rounding of a bin -- follows from which outputs are asked
  // for: this kernel exists for the int8-only call of the sampling path.
  const char* ev = getenv("TFMQ_ATTN_CTX");
  const bool on = !(ev && atoi(ev) == 0);
  // (d = 80, the 32 x 32 level: measured 190 us against k_attention_h's 178 at UNet batch 128 -- not taken unless TFMQ_ATTN_CTX=2;
  //  d = 40, the 64 x 64 level: 291 against 347)
  const bool all = ev && atoi(ev) == 2;
  if (!on || !yq || Tk > 96 || Tks > 96 || Tq % 128 != 0 || (d != 40 && !(d == 80 && all)) || (heads * d) % 16 != 0) return TFMQ_OK;
  const size_t smem = (d == 40 ? CtxGeo<40>::KBYTES + CtxGeo<40>::VBYTES : CtxGeo<80>::KBYTES + CtxGeo<80>::VBYTES) + 128 * static_cast<size_t>(heads) * d;
  if (smem > 160 * 1024) return TFMQ_OK;
  *taken = true;
  // queries as token rows [B][Tq][ldq], head hd at columns hd * d (the only layout the ABI describes)
  AttnCtxP p{reinterpret_cast<const __half*>(q), reinterpret_cast<const __half*>(k), reinterpret_cast<const __half*>(vt), ldq, ldk, yq, aq,
             B, heads, Tq, Tk, Tks, scale, static_cast<long>(Tq) * ldq, d};
  return d == 40 ? launch_ctx<40>(h, p, as_stream(stream)) : launch_ctx<80>(h, p, as_stream(stream));
}
