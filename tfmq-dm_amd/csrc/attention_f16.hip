// K10 (main path): fused softmax(Q K^T * scale) V with fp16 operands written by the projection GEMM's epilogue
// (TFMQ_OUT_F16: q, k row-major [B][T][ld]; v TRANSPOSED [B][heads*d][Tk], tfmq_conv_desc.yt).  Same mathematics
// and the same MFMA operand roles as k_attention (attention.hip); what changes is how the tiles travel:
//
//   * no conversion and no transposition in the kernel: K rows and V^T rows are copied as 16-byte pieces;
//   * Q fragments live in registers for the whole kernel (no LDS for Q);
//   * 64-key tiles, double-buffered LDS, ONE barrier per tile; the global loads of tile t+1 are issued before
//     the MFMAs of tile t and written to the other buffer after them (register staging split);
//   * key k of a 32-key sub-tile is stored at LDS row swap_bits_2_3(k): the S^T accumulator registers of a lane
//     then hold 8 CONSECUTIVE keys per 16-key MFMA step, so the V^T fragment is one ds_read_b128;
//   * exp2 with the softmax scale folded in, running-max rescale of O skipped when no lane's max moved.
#include "common.hpp"
#include <type_traits>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

struct AttnHP {
  const __half *q, *k, *vt;
  int ldq, ldk;
  float* out;
  int ldo;
  int8_t* yq;
  tfmq_qsel aq;
  int B, heads, Tq, Tk, Tks, d;   // Tks: keys per batch item in memory (K rows, V^T row length), >= Tk, % 8 == 0
  float scale;
  int nsl;                        // output slices of 32*NT channels (1 unless 16*NKS > 32*NT: wide heads, see below)
};

// NKS = k-steps of the score MFMA (16 channels each), NT = 32-column output tiles: compile-time, so that the MFMA
// chains stay straight-line code (run-time trip counts made the compiler shuttle the accumulators between
// AGPRs and VGPRs: ~900 moves per key tile).  d <= 16*NKS, d <= 32*NT; the LDS padding is zero.
// ONES_ROW = d when 32*NT > d (else -1): V^T row d (a padding row of the last output tile) is set to 1.0 once, so the
// PV MFMA accumulates the softmax denominator sum_k P[q][k] in output row d -- rescaled with O for free -- and the
// 32 adds + one cross-lane exchange per key tile disappear from the VALU-bound softmax.
// FOLD (needs a spare score column, 16*NKS > d, and the ones-row): the softmax shift rides in the score MFMA.  K's
// padding column d holds 1.0, Q is pre-multiplied by scale*log2(e) and its column d holds -m (the running shift of
// the query, kept fp16-representable), so the accumulator comes out as the exp2 argument and the 32 fused
// multiply-adds per key tile disappear from the VALU-bound softmax.  The shift may lag the true row maximum by up to
// 2^8 (probabilities up to 256 in fp16 -- relative precision is unchanged, the common factor cancels in O / l);
// a tile whose maximum exceeds that re-bases the query (sub + rescale of O, as rarely as the maximum jumps).
template <int NKS, int NT, int ONES_ROW, bool FOLD = false>
// (the d = 40 self-attention variant is held to 128 VGPRs = four waves per SIMD -- its 37 KB of LDS allow four blocks per CU: 4.21 -> 4.08 ms
// at UNet batch 128, same-box A/B)
__global__ __launch_bounds__(256, (NT <= 2 && FOLD) ? 4 : 1) void k_attention_h(AttnHP p) {
  constexpr int DPAD = NT * 32;
  static_assert(!FOLD || (ONES_ROW >= 0 && 16 * NKS > ONES_ROW && ONES_ROW % 8 == 0), "FOLD: spare score column + ones-row");
  // Wide heads (16 * NKS > 32 * NT; the single 384-channel head of cin256-v2): the scores need the whole head dimension, the
  // output does not -- a block computes the scores over all 16 * NKS channels and the 32 * NT output channels of ITS slice
  // (p.nsl slices, blockIdx fastest); the softmax is recomputed per slice, identically.
  constexpr int KD = NKS * 16 > DPAD ? NKS * 16 : DPAD;   // channels of a staged K row
  static_assert(NKS * 16 <= DPAD || ONES_ROW < 0, "sliced output: no ones-row / fold");
  constexpr int KROW = KD * 2 + 16;      // bytes per K row in LDS (odd number of 16-byte slots: conflict-free)
  constexpr int VROW = 64 * 2 + 16;      // bytes per V^T row (64 keys)
  constexpr int KBUF = 64 * KROW, VBUF = DPAD * VROW;
  constexpr int KPT = (64 * (KD / 8) + 255) / 256;   // 16-byte pieces per thread, K tile (upper bound)
  constexpr int VPT = (DPAD * 8 + 255) / 256;          // ... V^T tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;                 // [2][64][KROW]
  unsigned char* sV = smem + 2 * KBUF;      // [2][DPAD][VROW]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  // XCD-aware order: the dispatcher puts block i on XCD i % 8; give each XCD a contiguous range of (batch, head,
  // query-block) triples so all query blocks of one (batch, head) read its K / V^T through ONE L2 (the row-major
  // order made every XCD fetch every K/V: 8x the HBM traffic, measured 766 MB for 126 MB of operands)
  int bid = blockIdx.x;
  {
    const int nb = gridDim.x, xcd = bid & 7, qn = nb >> 3, r = nb & 7;
    bid = (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + (bid >> 3);
  }
  const int sl = bid % p.nsl;               // output slice (wide heads; nsl = 1 otherwise)
  bid /= p.nsl;
  const int nqb = (p.Tq + 127) / 128;
  const int bh = bid / nqb;
  const int b = bh / p.heads, hd = bh % p.heads;
  const int q0 = (bid - bh * nqb) * 128;
  const int d = p.d, dp8 = d >> 3;          // pieces per K row

  // zero both buffers once: the pad pieces (halves d..16*nks of a K row, V^T rows d..32*nt) are never staged
  for (int i = tid; i < (2 * KBUF + 2 * VBUF) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  // ---- Q fragments (B operand of S^T = K Q^T): lane (query j, half hh) holds d-range ks*16 + hh*8 .. +7
  v8h qf[NKS];
  {
    const int qrow = q0 + wid * 32 + j;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const int c = ks * 16 + hh * 8;
      if (qrow < p.Tq && c < d)
        v = *reinterpret_cast<const uint4*>(p.q + (static_cast<size_t>(b) * p.Tq + qrow) * p.ldq + hd * d + c);
      qf[ks] = *reinterpret_cast<v8h*>(&v);
    }
    if constexpr (FOLD) {   // q * scale*log2(e), rounded once to fp16
      const float c2q = p.scale * 1.44269504088896340736f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[ks][e] = static_cast<_Float16>(static_cast<float>(qf[ks][e]) * c2q);
    }
    // Pin the arrival of the (conditional) Q loads HERE.  Otherwise the compiler's wait-count bookkeeping carries
    // "Q may still be in flight" into the key loop and puts s_waitcnt vmcnt(0) in front of every score MFMA,
    // which also drains the K/V prefetch of the next tile that was issued just before.
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks]));
  }

  const __half* kbase = p.k + static_cast<size_t>(b) * p.Tks * p.ldk + hd * d;
  const __half* vbase = p.vt + ((static_cast<size_t>(b) * p.heads + hd) * d + static_cast<size_t>(sl) * DPAD) * p.Tks;
  const int drows = (d - sl * DPAD) < DPAD ? (d - sl * DPAD) : DPAD;       // V^T rows / output channels of this slice
  // ---- staging plan of this thread (the same for every key tile): global offset, LDS offset, first key
  const int kpieces = 64 * dp8, vpieces = drows * 8;
  int k_go[KPT], k_lo[KPT], k_key[KPT], v_go[VPT], v_lo[VPT], v_key[VPT];
#pragma unroll
  for (int it = 0; it < KPT; ++it) {
    const int idx = tid + it * 256;
    const int key = idx / dp8, pc = idx - key * dp8;
    const int row = (key & 0x33) | ((key & 4) << 1) | ((key & 8) >> 1);   // swap bits 2 and 3
    k_key[it] = idx < kpieces ? key : (1 << 30);    // never valid
    k_go[it] = key * p.ldk + pc * 8;
    k_lo[it] = row * KROW + pc * 16;
  }
#pragma unroll
  for (int it = 0; it < VPT; ++it) {
    const int idx = tid + it * 256;
    const int dc = idx >> 3, kp = idx & 7;
    v_key[it] = idx < vpieces ? kp * 8 : (1 << 30);
    v_go[it] = dc * p.Tks + kp * 8;
    v_lo[it] = dc * VROW + kp * 16;
  }
  uint4 kreg[KPT], vreg[VPT];
  auto load_tile = [&](int kt) {
    const int left = p.Tk - kt * 64;   // keys left from the start of this tile
    const __half* kb = kbase + static_cast<size_t>(kt) * 64 * p.ldk;
    const __half* vb = vbase + kt * 64;
#pragma unroll
    for (int it = 0; it < KPT; ++it) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (k_key[it] < left) v = *reinterpret_cast<const uint4*>(kb + k_go[it]);
      kreg[it] = v;
    }
#pragma unroll
    for (int it = 0; it < VPT; ++it) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (v_key[it] < left) v = *reinterpret_cast<const uint4*>(vb + v_go[it]);
      vreg[it] = v;
    }
  };
  // whole tiles: no bounds to test -- a piece this thread does not own reads the tile's first bytes and is never stored
  // (the guarded form cost ~25 VALU instructions and 8 branches per tile in a loop that is bound by the VALU issue port)
  auto load_tile_full = [&](int kt) {
    const __half* kb = kbase + static_cast<size_t>(kt) * 64 * p.ldk;
    const __half* vb = vbase + kt * 64;
#pragma unroll
    for (int it = 0; it < KPT; ++it) kreg[it] = *reinterpret_cast<const uint4*>(kb + (k_key[it] < 64 ? k_go[it] : 0));
#pragma unroll
    for (int it = 0; it < VPT; ++it) vreg[it] = *reinterpret_cast<const uint4*>(vb + (v_key[it] < 64 ? v_go[it] : 0));
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int it = 0; it < KPT; ++it)
      if (k_key[it] < 64) *reinterpret_cast<uint4*>(sK + buf * KBUF + k_lo[it]) = kreg[it];
#pragma unroll
    for (int it = 0; it < VPT; ++it)
      if (v_key[it] < 64) *reinterpret_cast<uint4*>(sV + buf * VBUF + v_lo[it]) = vreg[it];
  };

  float m_run = FOLD ? 0.0f : -INFINITY, l_run = 0.0f;    // FOLD: m_run = the shift carried in Q's column d
  v16f o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
  const float c2 = p.scale * 1.44269504088896340736f;   // exp(x*scale) = exp2(x*c2)

  // (Round 2, measured and dropped: PMC on the d = 40 self-attention at UNet batch 128 -- MFMA pipe 47 % busy, VALU 62 %, 100
  // VALU instructions + 14 MFMAs per key tile and wave, ~950 cycles per tile where max(MFMA, VALU) would be ~580.  s_setprio(1)
  // around the two MFMA clusters: -1 %.  Issuing the next tile's score MFMAs before this tile's softmax (K staged two tiles
  // ahead, V one, still one barrier per tile; bit-identical): 182 VGPRs = two waves per SIMD instead of three, 4.2 -> 6.0 ms.)
  // one 64-key tile: scores, online softmax, O += P V.  MASK (compile time) = the ragged last tile.
  auto tile = [&](int kt, int buf, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const unsigned char* bK = sK + buf * KBUF;
    const unsigned char* bV = sV + buf * VBUF;
    // ---- S^T = K Q^T for the two 32-key sub-tiles
    v16f s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[sub][r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const v8h a = *reinterpret_cast<const v8h*>(bK + (sub * 32 + j) * KROW + (ks * 16 + hh * 8) * 2);
        s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], s[sub], 0, 0, 0);
      }
    }
    // register r of sub-tile sub holds key kt*64 + sub*32 + 16*(r>>3) + 8*hh + (r&7)
    if constexpr (MASK) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 64 + sub * 32 + 16 * (r >> 3) + 8 * hh + (r & 7);
          if (key >= p.Tk) s[sub][r] = -INFINITY;
        }
    }
    if constexpr (FOLD) {
      // s already holds (score*scale - m_run) * log2(e)
      float mx = s[0][0];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (__builtin_amdgcn_ballot_w64(mx > 8.0f) != 0) {          // re-base the queries whose maximum ran away
        const float m_new = mx > 8.0f ? static_cast<float>(static_cast<_Float16>(m_run + mx)) : m_run;
        const float delta = m_new - m_run;                          // exact: both fp16-representable
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[sub][r] -= delta;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        m_run = m_new;
        constexpr int ksb = ONES_ROW / 16, eb = ONES_ROW % 16;      // Q column d: k-step, lane half, element
        if (hh == eb / 8) qf[ksb][eb % 8] = static_cast<_Float16>(-m_run);
      }
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[sub][r] = __builtin_amdgcn_exp2f(s[sub][r]);
    } else {
    // ---- online softmax over this lane's 32 keys (+ partner lane^32)
    float mx = s[0][0];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float mc = m_new * c2;
    float rs = 0.0f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[sub][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sub][r], c2, -mc));
        if constexpr (ONES_ROW < 0) rs += s[sub][r];
      }
    if constexpr (ONES_ROW < 0) rs += __shfl_xor(rs, 32, 64);
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {   // some query's running max moved: rescale
      const float alpha = __builtin_amdgcn_exp2f(m_run * c2 - mc);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m_run = m_new;
    }
    l_run += rs;
    }
    // ---- O^T += V^T P^T : 4 MFMA k-steps of 16 keys
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v8h bp;
#pragma unroll
      for (int e = 0; e < 8; ++e) bp[e] = static_cast<_Float16>(s[u >> 1][8 * (u & 1) + e]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const v8h a = *reinterpret_cast<const v8h*>(bV + (t * 32 + j) * VROW + (16 * u + 8 * hh) * 2);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bp, o[t], 0, 0, 0);
      }
    }
  };

  const int ntiles = (p.Tk + 63) / 64;
  const int nfull = p.Tk / 64;       // tiles without a ragged tail
  load_tile(0);
  __syncthreads();   // zero fill done
  if constexpr (FOLD) {     // K's padding column d = 1.0 in both buffers (the staging never touches padding pieces)
    if (tid < 128)
      *reinterpret_cast<unsigned short*>(sK + (tid >> 6) * KBUF + (tid & 63) * KROW + ONES_ROW * 2) = 0x3C00u;
  }
  if constexpr (ONES_ROW >= 0) {
    if (tid < 16) {
      const uint4 ones = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);   // 8 x fp16 1.0
      *reinterpret_cast<uint4*>(sV + (tid >> 3) * VBUF + ONES_ROW * VROW + (tid & 7) * 16) = ones;
    }
  }
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nfull; ++kt) {
    if constexpr (NT <= 2 && FOLD) {           // (the register-capped d = 40 variant spills with the unguarded loader: +3 %)
      if (kt + 1 < ntiles) load_tile(kt + 1);
    } else {
      if (kt + 1 < nfull) load_tile_full(kt + 1);       // d = 80: 547 -> 524 us at UNet batch 128
      else if (kt + 1 < ntiles) load_tile(kt + 1);
    }
    tile(kt, kt & 1, std::false_type{});
    if (kt + 1 < ntiles) store_tile((kt & 1) ^ 1);
    __syncthreads();
  }
  if (nfull < ntiles) tile(nfull, nfull & 1, std::true_type{});

  // ---- normalise and store: lane (query j, half hh) owns dcols t*32 + (r&3) + 8*(r>>2) + 4*hh
  const int qg = q0 + wid * 32 + j;
  if (qg >= p.Tq) return;
  if constexpr (ONES_ROW >= 0) {     // the denominator sits in output row ONES_ROW: lanes of half hh_one, register r_one
    constexpr int lr = ONES_ROW % 32, hh_one = (lr >> 2) & 1, r_one = (lr & 3) + 4 * (lr >> 3);
    const float mine = o[ONES_ROW / 32][r_one];
    const float other = __shfl_xor(mine, 32, 64);
    l_run = hh == hh_one ? mine : other;
  }
  const float inv = 1.0f / l_run;
  const bool quant = p.yq != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(p.aq);
  const size_t tok = static_cast<size_t>(b) * p.Tq + qg;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dc = sl * DPAD + t * 32 + 8 * g + 4 * hh;
      if (dc >= d || t * 32 + 8 * g + 4 * hh >= drows) continue;
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

template <int NKS, int NT, int ONES_ROW = -1, bool FOLD = false>
static int launch_attn_h(tfmq_handle h, const AttnHP& p, void* stream) {
  constexpr int DPAD = NT * 32;
  constexpr int KD = NKS * 16 > DPAD ? NKS * 16 : DPAD;
  constexpr size_t smem = 2 * (64 * (KD * 2 + 16) + static_cast<size_t>(DPAD) * (64 * 2 + 16));
  static bool configured = false;
  if (!configured) {
    TFMQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_h<NKS, NT, ONES_ROW, FOLD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
    configured = true;
  }
  AttnHP q = p;
  q.nsl = NKS * 16 > DPAD ? (p.d + DPAD - 1) / DPAD : 1;
  dim3 grid(static_cast<unsigned>((p.Tq + 127) / 128) * p.B * p.heads * q.nsl);
  hipLaunchKernelGGL((k_attention_h<NKS, NT, ONES_ROW, FOLD>), grid, dim3(256), smem, as_stream(stream), q);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_attention_f16(tfmq_handle h, const uint16_t* q, const uint16_t* k, const uint16_t* vt, int ldq, int ldk,
                                  float* out, int ldo, int8_t* yq, tfmq_qsel aq, int B, int heads, int Tq, int Tk,
                                  int Tk_stride, int d, float scale, void* stream) {
  TFMQ_CHECK_ARG(h, h && q && k && vt && (out || yq), "attention_f16: null pointer");
  TFMQ_CHECK_ARG(h, B > 0 && heads > 0 && Tq > 0 && Tk > 0 && d > 0, "attention_f16: bad shape");
  TFMQ_CHECK_ARG(h, d % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && Tk_stride % 8 == 0 && Tk_stride >= Tk && (!out || ldo % 4 == 0),
                 "attention_f16: head dim, leading dims and Tk_stride must be multiples of 8, Tk_stride >= Tk");
  TFMQ_CHECK_ARG(h, !yq || aq.qtable, "attention_f16: quantised output needs a qparam");
  AttnHP p{reinterpret_cast<const __half*>(q), reinterpret_cast<const __half*>(k), reinterpret_cast<const __half*>(vt),
           ldq, ldk, out, ldo, yq, aq, B, heads, Tq, Tk, Tk_stride, d, scale, 1};
  if (d <= 32) return launch_attn_h<2, 1>(h, p, stream);
  if (d == 40) return launch_attn_h<3, 2, 40, true>(h, p, stream);  // SD v1 at 64x64
  if (d <= 48) return launch_attn_h<3, 2>(h, p, stream);
  if (d <= 64) return launch_attn_h<4, 2>(h, p, stream);
  if (d == 80) return launch_attn_h<5, 3, 80>(h, p, stream);  // SD v1 at 32x32
  if (d <= 80) return launch_attn_h<5, 3>(h, p, stream);
  if (d <= 96) return launch_attn_h<6, 3>(h, p, stream);
  if (d <= 128) return launch_attn_h<8, 4>(h, p, stream);
  if (d <= 160) return launch_attn_h<10, 5>(h, p, stream);   // SD v1 at 16x16 / 8x8
  if (d <= 256) return launch_attn_h<16, 8>(h, p, stream);   // DDPM UNet's single 256-channel head (one wave per SIMD)
  if (d <= 384) return launch_attn_h<24, 4>(h, p, stream);   // cin256-v2's single head at 32x32: scores over 384 channels, 3 output slices of 128
  if (h) h->err = "attention_f16: head dim > 384 not supported (use tfmq_attention)";
  return TFMQ_ERR_UNSUPPORTED;
}
