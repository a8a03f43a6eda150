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
#include <cstdlib>
#include <type_traits>
#ifdef TFMQ_PHASE_TIMERS
#include <cstdio>
#include <vector>
#endif

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

int launch_attention_ctx(tfmq_handle h, const uint16_t* q, const uint16_t* k, const uint16_t* vt, int ldq, int ldk, int8_t* yq, tfmq_qsel aq, int B,
                         int heads, int Tq, int Tk, int Tks, int d, float scale, void* stream, bool* taken);

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
#ifdef TFMQ_PHASE_TIMERS
  unsigned long long* dbg;        // diagnostics build: [blocks][2 waves][8] shader cycles per segment of k_attention_d40_pp
#endif
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
// (round 6, measured and dropped: the d = 80 form <5, 3, 80> held to 168 VGPRs by __launch_bounds__(256, 3) for a third block per CU -- 3 x 54 272 B
// of LDS fit -- spills 12 registers and is 5 % SLOWER, 523 / 512 -> 549 / 555 us at UNet batch 128: profiles/r06_ab_lin_geglu_nst2_attn80.txt)
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


// ---------------------------------------------------------------------------------------------------------------------
// K10p: the d = 40 self-attention (SD v1 at 64x64: T = 4096, 8 heads -- the largest single kernel of the sampling step)
// as a SOFTWARE-PIPELINED loop.  Same operand roles, fold of the softmax shift into the score MFMA and ones-row
// denominator as k_attention_h<3, 2, 40, true>; what changes is the schedule.  In k_attention_h a wave runs
// scores -> softmax -> P V of one key tile as ONE dependence chain (every PV MFMA waits for the ds_read issued right
// before it, the softmax waits for the score MFMAs, a 17-deep max chain, an LDS round trip for the lane exchange), and
// measured MFMA time + VALU time add up: 4.06 ms at UNet batch 128.  scratch/ubench/mfma_valu_overlap.hip: on one SIMD a
// v_mfma_f32_32x32x16_f16 (15 ns) hides ~5 plain VALU instructions or ~2.5 v_exp_f32 of the waves resident there, but only
// when they sit next to it in the instruction streams.  Here iteration t of a wave holds three INDEPENDENT pieces of work,
//
//       P V of tile t-1        (8 MFMAs, operands: the packed P of the previous iteration, V^T(t-1) fragments read then)
//       softmax of tile t      (VALU: max tree, 32 exp2, 16 cvt_pk; input: the scores computed in iteration t-1)
//       scores of tile t+1     (6 MFMAs, K(t+1) fragments)
//
// so every MFMA has VALU work of the same wave beside it that does not wait for it.  K and V^T tiles arrive by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write) into rings of four buffers, K three tiles ahead of its
// use and V^T two, with COUNTED waits (only the batch issued one iteration earlier must have landed); the V^T fragments
// of the next iteration are read before the barrier, so the first MFMAs behind it wait for nothing.  One barrier per
// tile; 2 waves per SIMD (two 4-wave blocks per CU).
//
// LDS image (lane-linear DMA destinations, so the layouts are chosen on the SOURCE side):
//   K tile   64 rows x 80 B (5 pieces, no padding); row rho holds key swap_bits_2_3(rho) (P^T fragments = 8 consecutive
//            keys).  80-byte rows are conflict-free for ds_read_b128 (5 rho mod 16 is a bijection on each lane group).
//            The third k-step (channels 32..47) reads channels 32..39 in the lower lane half; the upper half -- the
//            padding columns 40..47, of which column 40 carries the folded shift -- reads a constant piece {1, 0 x 7}.
//   V^T tile 64 rows x 128 B (rows 0..39 by DMA, row 40 = ones, rows 41..63 zero); slot s of row r holds the piece
//            s ^ ((r >> 1) & 7): conflict-free for the 16-lane groups of ds_read_b128.
// The rescale decision of tile t (some query's score above the shift by more than 2^8) sits between the P V MFMAs and the
// score MFMAs: O *= alpha after P V(t-1) is complete, the exponentials of tile t already taken against the old shift are
// taken again, the others get the new shift, and the scores of tile t+1 are computed with the new Q column.
// v_max3_f32 without the canonicalising v_max_f32 x, x the compiler puts in front of fmaxf on MFMA outputs (6 per tile)
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// (volatile: keeps its place behind the volatile s_nop padding that separates it from the MFMAs whose results it reads)
__device__ __forceinline__ float vmax3v(float a, float b, float c) {
  float r;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  // lane l: 16 bytes from sbase + voff(l) -> LDS byte lds_dst + 16 l
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

// (Measured and dropped: the P V accumulator in AccVGPRs -- inline-asm MFMAs with "+a" operands; in scratch/ubench/mfma_valu_overlap.hip six
// plain VALU instructions hide beside such an MFMA against four beside the arch-VGPR form.  In this kernel it is 4 % SLOWER (3.89 vs 3.72 ms):
// the compiler schedules nothing around opaque asm statements, and with AccVGPRs in use it splits a 256-register budget 128 : 128.)
// DBG (timing-only ablations, results are garbage): 1 no exp2, 2 no PV MFMAs, 4 no score MFMAs, 8 no fragment reads, 16 no max tree,
// 32 no fp16 packing, 64 no DMA, 128 no sched_group_barrier, 256 no barrier / waits
// NW = waves per block (32 queries each): 4, or 8 -- half the DMA instructions and L2 -> LDS bytes per query, one barrier for both waves of a SIMD
// ACC (round 5): the P V accumulators live in AccVGPRs, o[0] = a[0:15], o[1] = a[16:31] BY NAME (physical-register constraints on
// non-volatile inline-asm MFMAs, so the scheduler still moves them and no copy is ever made: with "+a" the register allocator kept the
// loop-carried value in arch VGPRs and moved 32 registers in and out per key tile; in round 3 it split the budget 128 : 128 and spilled).
// The score accumulators -- which the softmax reads with VALU instructions -- stay in arch VGPRs.  scratch/ubench/mfma_valu_overlap.hip:
// six plain VALU instructions hide beside an MFMA whose accumulator is an AccVGPR tuple, four beside the arch-VGPR form.
template <bool ACC, int TT>
__device__ __forceinline__ void pv_mfma(v16f& o, const v8h& a, const v8h& b) {
  if constexpr (!ACC) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, o, 0, 0, 0);
  else if constexpr (TT == 0) asm("v_mfma_f32_32x32x16_f16 a[0:15], %1, %2, a[0:15]" : "+{a[0:15]}"(o) : "v"(a), "v"(b));
  else asm("v_mfma_f32_32x32x16_f16 a[16:31], %1, %2, a[16:31]" : "+{a[16:31]}"(o) : "v"(a), "v"(b));
}
// O *= alpha on the named AccVGPRs (the rare rescale branch); wait states around it by hand: the compiler does not know the producers are MFMAs
#define TFMQ_ACC_SCALE1(R) "v_accvgpr_read_b32 %2, a" #R "\n\tv_mul_f32 %2, %2, %3\n\tv_accvgpr_write_b32 a" #R ", %2\n\t"
#define TFMQ_ACC_SCALE4(A, B, C, D) TFMQ_ACC_SCALE1(A) TFMQ_ACC_SCALE1(B) TFMQ_ACC_SCALE1(C) TFMQ_ACC_SCALE1(D)
__device__ __forceinline__ void acc_scale32(v16f& o0, v16f& o1, float alpha) {
  float tmp;
  asm volatile("s_nop 15\n\ts_nop 3\n\t"
               TFMQ_ACC_SCALE4(0, 1, 2, 3) TFMQ_ACC_SCALE4(4, 5, 6, 7) TFMQ_ACC_SCALE4(8, 9, 10, 11) TFMQ_ACC_SCALE4(12, 13, 14, 15)
               TFMQ_ACC_SCALE4(16, 17, 18, 19) TFMQ_ACC_SCALE4(20, 21, 22, 23) TFMQ_ACC_SCALE4(24, 25, 26, 27) TFMQ_ACC_SCALE4(28, 29, 30, 31)
               "s_nop 7"
               : "+{a[0:15]}"(o0), "+{a[16:31]}"(o1), "=&v"(tmp)
               : "v"(alpha));
}
// COMPACT (round 6): the two constant pieces live in zero rows 41 and 61 of V^T buffer 0 (rows the DMA never writes; as V^T rows they
// only feed output channels 41 and 61, which are not stored) instead of a 3 KiB region of their own: 53 248 bytes of LDS, so that THREE
// 4-wave blocks share a CU (168 VGPRs allow three waves per SIMD; 56 320 bytes allowed two blocks).
template <int NW, int DBG = 0, int OCC = 2, bool ACC = false, bool COMPACT = false>
__global__ __launch_bounds__(64 * NW, OCC) void k_attention_d40(AttnHP p) {
  constexpr int NTH = 64 * NW, QB = 32 * NW;
  constexpr int NSLOT = (10 + NW - 1) / NW, NFULL = 10 % NW;      // DMA wave-instructions per wave: NSLOT for waves < NFULL, else NSLOT - 1
  constexpr int KROW = 80, KSUB = 32 * KROW, KBUF = 64 * KROW;   // 2560, 5120
  constexpr int VBASE = COMPACT ? 4 * KBUF : 23552, VROW = 128, VBUF = 64 * VROW;    // 8192
  constexpr int ONES = COMPACT ? VBASE + 41 * VROW : 4 * KBUF;  // constant pieces at ONES, ONES + KSUB (COMPACT: rows 41 and 61 of V^T buffer 0)
  constexpr int LDS_BYTES = VBASE + 4 * VBUF;                   // 56320 (COMPACT: 53248)
  constexpr int NE1 = 20;                                       // exponentials taken before the rescale decision
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  int bid = blockIdx.x;
  {
    const int nb = gridDim.x, xcd = bid & 7, qn = nb >> 3, r = nb & 7;
    bid = (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + (bid >> 3);
  }
  const int nqb = p.Tq / QB;
  const int bh = bid / nqb;
  const int b = bh / p.heads, hd = bh % p.heads;
  const int q0 = (bid - bh * nqb) * QB;
  constexpr int d = 40;

  for (int i = tid; i < LDS_BYTES / 16; i += NTH) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  // ---- Q fragments, pre-multiplied by scale * log2(e) (rounded once to fp16); column 40 will carry -shift
  v8h qf[3];
  {
    const int qrow = q0 + wid * 32 + j;
    const float c2q = p.scale * 1.44269504088896340736f;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const int c = ks * 16 + hh * 8;
      if (c < d) v = *reinterpret_cast<const uint4*>(p.q + (static_cast<size_t>(b) * p.Tq + qrow) * p.ldq + hd * d + c);
      qf[ks] = *reinterpret_cast<v8h*>(&v);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[ks][e] = static_cast<_Float16>(static_cast<float>(qf[ks][e]) * c2q);
    }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) asm volatile("" : "+v"(qf[ks]));
  }

  // ---- DMA plan: wave-instructions 0..4 = a K tile (320 pieces), 5..9 = a V^T tile; wave w issues w, w + NW, ...
  const unsigned char* kptr = reinterpret_cast<const unsigned char*>(p.k + static_cast<size_t>(b) * p.Tks * p.ldk + hd * d);
  const unsigned char* vptr = reinterpret_cast<const unsigned char*>(p.vt + (static_cast<size_t>(b) * p.heads + hd) * d * p.Tks);
  const size_t kstep = static_cast<size_t>(64) * p.ldk * 2;     // bytes per K tile (a V^T tile: 128)
  unsigned dma_off[NSLOT], slot_dst[NSLOT];
  bool slot_k[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int id = wid + NW * s;
    slot_k[s] = id < 5;
    if (id < 5) {
      const int n = id * 64 + lane, rho = n / 5, c = n - rho * 5;
      const int key = (rho & 0x33) | ((rho & 4) << 1) | ((rho & 8) >> 1);
      dma_off[s] = static_cast<unsigned>(key * p.ldk * 2 + c * 16);
      slot_dst[s] = id * 1024;
    } else {
      const int n = (id - 5) * 64 + lane, r = n >> 3, sl = n & 7;
      dma_off[s] = static_cast<unsigned>(r * p.Tks * 2 + ((sl ^ ((r >> 1) & 7)) << 4));
      slot_dst[s] = VBASE + (id - 5) * 1024;
    }
  }
  const bool slot2 = wid < NFULL;  // the last slot exists only for the first waves
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));   // LDS byte address of the image
  unsigned slot_lds[NSLOT];        // wave-uniform LDS destinations, in SGPRs (M0 is written from them)
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) slot_lds[s] = __builtin_amdgcn_readfirstlane(lds0 + slot_dst[s]);
  auto dma = [&](const unsigned char* kp, const unsigned char* vp, int kb, int vb, bool do_k, bool do_v) {
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      if (s == NSLOT - 1 && !slot2) continue;
      const bool isk = slot_k[s];          // wave-uniform
      if (isk ? do_k : do_v) glds16_s(isk ? kp : vp, dma_off[s], slot_lds[s] + (isk ? kb * KBUF : vb * VBUF));
    }
  };

  // ---- fragment addresses (bytes; + buffer offset as an immediate)
  const unsigned ka = j * KROW + hh * 16;                         // k-steps 0, 1 at +0, +32
  unsigned ka2[4];                                                // k-step 2: the upper lane half reads the constant piece
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) ka2[kb] = hh ? static_cast<unsigned>(ONES - kb * KBUF) : static_cast<unsigned>(j * KROW + 64);
  unsigned va[4];
  {
    const int g = (j >> 1) & 7;
#pragma unroll
    for (int u = 0; u < 4; ++u) va[u] = VBASE + j * VROW + (((2 * u + hh) ^ g) << 4);      // row j + 32 tt: + 4096 tt, same g
  }

  float m_run = 0.0f;
  v16f o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
  unsigned pp[16];                 // packed fp16 P of the previous tile: pp[4 u + e] = keys 16 u + 8 hh + 2 e, +1
#pragma unroll
  for (int i = 0; i < 16; ++i) pp[i] = 0u;
  v8h vf[4][2];                    // V^T fragments of the tile whose P is in pp
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) vf[u][tt] = qf[0];   // finite; multiplied by P = 0 in the first iteration

  __syncthreads();                 // zero fill done
  if (tid < 2) *reinterpret_cast<uint4*>(smem + ONES + tid * KSUB) = make_uint4(0x00003C00u, 0, 0, 0);
  if (tid >= 64 && tid < 96) {
    const uint4 ones = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);
    *reinterpret_cast<uint4*>(smem + VBASE + ((tid - 64) >> 3) * VBUF + 40 * VROW + ((tid - 64) & 7) * 16) = ones;
  }
  const int nt = p.Tk >> 6;        // a multiple of 4
  dma(kptr, vptr, 0, 0, true, true);                          // K(0), V(0)
  dma(kptr + kstep, vptr + 128, 1, 1, true, true);            // K(1), V(1)
  dma(kptr + 2 * kstep, vptr, 2, 0, true, false);             // K(2)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  auto row_max = [&](const v16f (&s)[2]) {
    auto vmax3 = [](float a, float b, float c) { return ACC ? vmax3v(a, b, c) : ::vmax3(a, b, c); };
    float m0 = vmax3(s[0][0], s[0][1], s[0][2]), m1 = vmax3(s[0][3], s[0][4], s[0][5]);
    float m2 = vmax3(s[1][0], s[1][1], s[1][2]), m3 = vmax3(s[1][3], s[1][4], s[1][5]);
    m0 = vmax3(m0, s[0][6], s[0][7]);   m1 = vmax3(m1, s[0][8], s[0][9]);
    m2 = vmax3(m2, s[1][6], s[1][7]);   m3 = vmax3(m3, s[1][8], s[1][9]);
    m0 = vmax3(m0, s[0][10], s[0][11]); m1 = vmax3(m1, s[0][12], s[0][13]);
    m2 = vmax3(m2, s[1][10], s[1][11]); m3 = vmax3(m3, s[1][12], s[1][13]);
    m0 = vmax3(m0, s[0][14], s[0][15]); m2 = vmax3(m2, s[1][14], s[1][15]);
    m0 = vmax3(m0, m1, m2);
    m0 = vmax3(m0, m3, m3);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);   // lane ^ 32: the query's other 32 keys
    return vmax3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
  };
  constexpr int ksb = 2, qe = 0;       // Q column 40 = k-step 2, upper lane half, element 0

  v16f sA[2], sB[2];
  {   // scores of tile 0; its row maximum is the first shift (fp16-representable), in both directions
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sA[sub][r] = 0.0f;
      const v8h k0 = *reinterpret_cast<const v8h*>(smem + ka + sub * KSUB), k1 = *reinterpret_cast<const v8h*>(smem + ka + sub * KSUB + 32);
      const v8h k2 = *reinterpret_cast<const v8h*>(smem + ka2[0] + sub * KSUB);
      sA[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[0], sA[sub], 0, 0, 0);
      sA[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[1], sA[sub], 0, 0, 0);
      sA[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k2, qf[2], sA[sub], 0, 0, 0);
    }
    asm volatile("s_nop 15" : "+v"(sA[0]), "+v"(sA[1]));     // MFMA results -> inline-asm VALU readers: the wait states the compiler cannot count
    const float mx = row_max(sA);
    m_run = static_cast<float>(static_cast<_Float16>(mx));
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) sA[sub][r] -= m_run;
    if (hh) qf[ksb][qe] = static_cast<_Float16>(-m_run);
  }

  // Iteration t (B = t & 3): sc = scores of tile t (in), sn = scores of tile t+1 (out).  DMA of K(t+3) -> K buffer (t+3)&3
  // and V^T(t+2) -> V buffer (t+2)&3; K(t+1) fragments from K buffer (t+1)&3; vf / pp hold V^T(t-1) / P(t-1) on entry and
  // V^T(t) (from V buffer t&3) / P(t) on exit.  COUNTED: only the batch of the previous iteration must have landed.
  auto iter = [&](v16f (&sc)[2], v16f (&sn)[2], auto b_tag, auto kdma_tag, auto vdma_tag, auto last_tag, auto counted_tag,
                  const unsigned char* kp, const unsigned char* vp) {
    constexpr int B = decltype(b_tag)::value, KB = (B + 1) & 3;
    constexpr bool KDMA = decltype(kdma_tag)::value, VDMA = decltype(vdma_tag)::value, LAST = decltype(last_tag)::value;
    constexpr bool COUNTED = decltype(counted_tag)::value && !(DBG & 2048);
    if constexpr (DBG & 4096) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (!(DBG & 64)) dma(kp, vp, (B + 3) & 3, (B + 2) & 3, KDMA, VDMA);
    v8h kf[2][3];
    if constexpr (!LAST) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if constexpr (DBG & 8) {
          kf[sub][0] = qf[0]; kf[sub][1] = qf[1]; kf[sub][2] = qf[2];
        } else {
          kf[sub][0] = *reinterpret_cast<const v8h*>(smem + KB * KBUF + ka + sub * KSUB);
          kf[sub][1] = *reinterpret_cast<const v8h*>(smem + KB * KBUF + ka + sub * KSUB + 32);
          kf[sub][2] = *reinterpret_cast<const v8h*>(smem + KB * KBUF + ka2[KB] + sub * KSUB);
        }
      }
    }
    // ---- block 1: P V of tile t-1  ||  row maximum of tile t, its first NE1 exponentials
    float mx;
    float pe[32];                          // P in fp32
    if constexpr (ACC) {
      // The inline-asm MFMAs are opaque to the scheduler (it clumps them): the interleave is written out, one MFMA per group with ~7-8
      // issue slots of VALU work behind it (an exponential takes two), groups fenced by sched_barrier.  The first exponentials are ordinary
      // instructions -- the compiler counts their distance to the score MFMAs of the previous iteration itself -- and the asm maximum tree
      // sits behind them, so it needs no s_nop padding of its own.
      auto bfrag = [&](int u) {
        v8h bp;
        unsigned* bw = reinterpret_cast<unsigned*>(&bp);
#pragma unroll
        for (int e = 0; e < 4; ++e) bw[e] = pp[4 * u + e];
        return bp;
      };
      auto ex = [&](int i) { pe[i] = __builtin_amdgcn_exp2f(sc[i >> 4][i & 15]); };
      float m0, m1, m2, m3;
      pv_mfma<true, 0>(o[0], vf[0][0], bfrag(0));
      __builtin_amdgcn_sched_barrier(0);
      ex(0); ex(1);
      __builtin_amdgcn_sched_barrier(0);
      m0 = vmax3v(sc[0][0], sc[0][1], sc[0][2]); m1 = vmax3v(sc[0][3], sc[0][4], sc[0][5]);
      m2 = vmax3v(sc[1][0], sc[1][1], sc[1][2]); m3 = vmax3v(sc[1][3], sc[1][4], sc[1][5]);
      __builtin_amdgcn_sched_barrier(0);
      pv_mfma<true, 1>(o[1], vf[0][1], bfrag(0));
      __builtin_amdgcn_sched_barrier(0);
      m0 = vmax3v(m0, sc[0][6], sc[0][7]); m1 = vmax3v(m1, sc[0][8], sc[0][9]);
      m2 = vmax3v(m2, sc[1][6], sc[1][7]); m3 = vmax3v(m3, sc[1][8], sc[1][9]);
      ex(2);
      __builtin_amdgcn_sched_barrier(0);
      pv_mfma<true, 0>(o[0], vf[1][0], bfrag(1));
      __builtin_amdgcn_sched_barrier(0);
      m0 = vmax3v(m0, sc[0][10], sc[0][11]); m1 = vmax3v(m1, sc[0][12], sc[0][13]);
      m2 = vmax3v(m2, sc[1][10], sc[1][11]); m3 = vmax3v(m3, sc[1][12], sc[1][13]);
      ex(3); ex(4);
      __builtin_amdgcn_sched_barrier(0);
      pv_mfma<true, 1>(o[1], vf[1][1], bfrag(1));
      __builtin_amdgcn_sched_barrier(0);
      m0 = vmax3v(m0, sc[0][14], sc[0][15]); m2 = vmax3v(m2, sc[1][14], sc[1][15]);
      m0 = vmax3v(m0, m1, m2);
      ex(5); ex(6);
      __builtin_amdgcn_sched_barrier(0);
      pv_mfma<true, 0>(o[0], vf[2][0], bfrag(2));
      __builtin_amdgcn_sched_barrier(0);
      m0 = vmax3v(m0, m3, m3);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);
        mx = vmax3v(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
      }
      ex(7); ex(8);
      __builtin_amdgcn_sched_barrier(0);
      pv_mfma<true, 1>(o[1], vf[2][1], bfrag(2));
      __builtin_amdgcn_sched_barrier(0);
      ex(9); ex(10); ex(11); ex(12);
      __builtin_amdgcn_sched_barrier(0);
      pv_mfma<true, 0>(o[0], vf[3][0], bfrag(3));
      __builtin_amdgcn_sched_barrier(0);
      ex(13); ex(14); ex(15); ex(16);
      __builtin_amdgcn_sched_barrier(0);
      pv_mfma<true, 1>(o[1], vf[3][1], bfrag(3));
      __builtin_amdgcn_sched_barrier(0);
      ex(17); ex(18); ex(19);
      __builtin_amdgcn_sched_barrier(0);
      static_assert(NE1 == 20, "the written-out interleave takes 20 exponentials in front of the rescale decision");
    } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v8h bp;
      unsigned* bw = reinterpret_cast<unsigned*>(&bp);
#pragma unroll
      for (int e = 0; e < 4; ++e) bw[e] = pp[4 * u + e];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if constexpr (DBG & 2) { o[tt][u] += static_cast<float>(vf[u][tt][0]) + __uint_as_float(bw[tt]); }
        else o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[u][tt], bp, o[tt], 0, 0, 0);
      }
    }
    // The maximum tree is inline asm (v_max3_f32 without the compiler's canonicalising v_max): the compiler's hazard recogniser does not
    // know that these statements READ registers an MFMA wrote (11 wait states after an 8-pass MFMA).  The barrier and the DMA issue lie
    // in between, except in the last iterations, which issue no DMA: pad.
    asm volatile("s_nop 7");
    if constexpr (DBG & 16) mx = sc[0][3] + sc[1][5];
    else mx = row_max(sc);
#pragma unroll
    for (int i = 0; i < NE1; ++i) {
      if constexpr (DBG & 1) pe[i] = sc[i >> 4][i & 15];
      else pe[i] = __builtin_amdgcn_exp2f(sc[i >> 4][i & 15]);
    }
    }
    if constexpr (DBG & 128) {
      // interleave: the K fragment reads first, then per MFMA two exponentials and three plain VALU instructions of the maximum tree
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#define TFMQ_SGB1(NT_) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x400, NT_, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      TFMQ_SGB1(3) TFMQ_SGB1(3) TFMQ_SGB1(3) TFMQ_SGB1(3) TFMQ_SGB1(2) TFMQ_SGB1(2) TFMQ_SGB1(2) TFMQ_SGB1(2)
    }
    // (the exponentials are pinned in front of the branch: the compiler would otherwise sink them into both of its arms, behind the MFMAs)
#pragma unroll
    for (int i = 0; i < NE1; ++i) asm volatile("" : "+v"(pe[i]));
    // ---- the rescale decision (rare)
    if (__builtin_amdgcn_ballot_w64(mx > 8.0f) != 0) {
      const float m_new = mx > 8.0f ? static_cast<float>(static_cast<_Float16>(m_run + mx)) : m_run;
      const float delta = m_new - m_run;                 // exact: both fp16-representable
      const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
      for (int i = 0; i < 32; ++i) sc[i >> 4][i & 15] -= delta;
#pragma unroll
      for (int i = 0; i < NE1; ++i) pe[i] = __builtin_amdgcn_exp2f(sc[i >> 4][i & 15]);
      if constexpr (ACC) acc_scale32(o[0], o[1], alpha);
      else {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[tt][r] *= alpha;
      }
      m_run = m_new;
      if (hh) qf[ksb][qe] = static_cast<_Float16>(-m_run);
    }
    // ---- block 2: scores of tile t+1  ||  the other exponentials, packing of P; V^T(t) fragments for the next iteration
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if constexpr (!(DBG & 8)) vf[u][tt] = *reinterpret_cast<const v8h*>(smem + B * VBUF + va[u] + tt * 32 * VROW);
      }
    if constexpr (!LAST) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sn[sub][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          if constexpr (DBG & 4) sn[sub][ks] += static_cast<float>(kf[sub][ks][0]);
          else sn[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[sub][ks], qf[ks], sn[sub], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = NE1; i < 32; ++i) {
      if constexpr (DBG & 1) pe[i] = sc[i >> 4][i & 15];
      else pe[i] = __builtin_amdgcn_exp2f(sc[i >> 4][i & 15]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i0 = 16 * (u >> 1) + 8 * (u & 1) + 2 * e;
        if constexpr (DBG & 32) pp[4 * u + e] = __float_as_uint(pe[i0]) ^ __float_as_uint(pe[i0 + 1]);
        else {
          const __half2 h2 = __floats2half2_rn(pe[i0], pe[i0 + 1]);
          pp[4 * u + e] = *reinterpret_cast<const unsigned*>(&h2);
        }
      }
    if constexpr (DBG & 128) {
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      TFMQ_SGB1(2) TFMQ_SGB1(2) TFMQ_SGB1(2) TFMQ_SGB1(2) TFMQ_SGB1(2) TFMQ_SGB1(2)
    }
    // The packed P and the fragments are operands of an (empty) volatile statement in front of the barrier: otherwise the compiler
    // sinks the exponentials, conversions and reads behind the barrier, next to their use in the next iteration.
    asm volatile(""
                 : "+v"(pp[0]), "+v"(pp[1]), "+v"(pp[2]), "+v"(pp[3]), "+v"(pp[4]), "+v"(pp[5]), "+v"(pp[6]), "+v"(pp[7]),
                   "+v"(pp[8]), "+v"(pp[9]), "+v"(pp[10]), "+v"(pp[11]), "+v"(pp[12]), "+v"(pp[13]), "+v"(pp[14]), "+v"(pp[15]),
                   "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[2][0]), "+v"(vf[2][1]), "+v"(vf[3][0]), "+v"(vf[3][1]));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG & 256) {
    } else if constexpr (COUNTED) {
      if (slot2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSLOT) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSLOT - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using Y = std::true_type;
  using N = std::false_type;
  const unsigned char* kp = kptr + 3 * kstep;     // K(t + 3)
  const unsigned char* vp = vptr + 256;           // V^T(t + 2)
  for (int t = 0; t + 4 < nt; t += 4) {
    iter(sA, sB, I0{}, Y{}, Y{}, N{}, Y{}, kp, vp);
    iter(sB, sA, I1{}, Y{}, Y{}, N{}, Y{}, kp + kstep, vp + 128);
    iter(sA, sB, I2{}, Y{}, Y{}, N{}, Y{}, kp + 2 * kstep, vp + 256);
    iter(sB, sA, I3{}, Y{}, Y{}, N{}, Y{}, kp + 3 * kstep, vp + 384);
    kp += 4 * kstep;
    vp += 512;
  }
  iter(sA, sB, I0{}, Y{}, Y{}, N{}, N{}, kp, vp);                 // t = nt-4: K(nt-1), V(nt-2)
  iter(sB, sA, I1{}, N{}, Y{}, N{}, N{}, kp, vp + 128);           // t = nt-3: V(nt-1)
  iter(sA, sB, I2{}, N{}, N{}, N{}, N{}, kp, vp);                 // t = nt-2
  iter(sB, sA, I3{}, N{}, N{}, Y{}, N{}, kp, vp);                 // t = nt-1: no scores left to compute
  // ---- P V of the last tile
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    v8h bp;
    unsigned* bw = reinterpret_cast<unsigned*>(&bp);
#pragma unroll
    for (int e = 0; e < 4; ++e) bw[e] = pp[4 * u + e];
    pv_mfma<ACC, 0>(o[0], vf[u][0], bp);
    pv_mfma<ACC, 1>(o[1], vf[u][1], bp);
  }
  if constexpr (ACC) asm volatile("s_nop 15\n\ts_nop 3" : "+{a[0:15]}"(o[0]), "+{a[16:31]}"(o[1]));

  // ---- normalise and store (as k_attention_h): lane (query j, half hh) owns channels t*32 + (r&3) + 8*(r>>2) + 4*hh;
  // the denominator sits in output row 40 = tile 1, lanes of the lower half, register 4
  const int qg = q0 + wid * 32 + j;
  float l_run;
  {
    constexpr int lr = 40 % 32, hh_one = (lr >> 2) & 1, r_one = (lr & 3) + 4 * (lr >> 3);
    const float mine = o[1][r_one];
    const float other = __shfl_xor(mine, 32, 64);
    l_run = hh == hh_one ? mine : other;
  }
  const float inv = 1.0f / l_run;
  const bool quant = p.yq != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(p.aq);
  const size_t tok = static_cast<size_t>(b) * p.Tq + qg;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dc = tt * 32 + 8 * g + 4 * hh;
      if (dc >= d) continue;
      float4 v = make_float4(o[tt][4 * g] * inv, o[tt][4 * g + 1] * inv, o[tt][4 * g + 2] * inv, o[tt][4 * g + 3] * inv);
      if (p.out) *reinterpret_cast<float4*>(p.out + tok * p.ldo + hd * d + dc) = v;
      if (quant) {
        char4 c = quant_char4(v.x, v.y, v.z, v.w, make_quantp(qp));
        *reinterpret_cast<char4*>(p.yq + tok * (static_cast<size_t>(p.heads) * d) + hd * d + dc) = c;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// K10pp (round 6): the d = 40 self-attention in PING-PONG form.  k_attention_d40 mixes the MFMAs and the softmax VALU work of three
// key tiles inside every wave and relies on the two pipes overlapping within one instruction stream; measured (PMC, profiles/
// r04_pmc_attention_d40.json) the matrix pipe is 58 % busy, the VALU 64 %, and only a fifth of the VALU time runs beside an MFMA:
// an in-order wave issues one instruction at a time, and the two waves of a SIMD -- from two different 4-wave blocks -- drift freely.
// Here a block has EIGHT waves (256 queries), waves w and w + 4 share a SIMD, and the two halves run the same key tile HALF A TILE
// APART, two workgroup barriers per tile (the "compute segment | load segment" pairing of MI355X_MICROARCH.md, two waves per SIMD):
//       V segment   softmax of tile t (row maximum, the rare re-base, 32 exp2, 16 cvt_pk -> P(t)), the fragment reads of
//                   K(t+1) and V^T(t), this wave's LDS-DMA pieces of K(t+4) / V^T(t+3): VALU + LDS + VMEM, no MFMA
//       M segment   S(t+1) = K(t+1) Q^T (6 MFMAs) and O += V^T(t) P(t) (8 MFMAs): nothing else
// so on every SIMD one wave's 14 MFMAs (448 matrix-pipe cycles) lie beside the other wave's ~75 VALU instructions.  A wave's own
// chain V(t) -> M(t) -> V(t+1) is serial -- no software pipelining inside a wave, one score buffer, no second P buffer, the re-base
// happens before any exponential is taken (none is taken twice).  Same operand roles, LDS image, folded shift and ones-row
// denominator as k_attention_d40; rings of four K and four V^T buffers, K four tiles ahead of its fragment read and V^T three,
// counted waits at the END of a V segment (everything but the pieces of this and the previous V segment has landed: >= 4 segments
// of latency) in front of the barrier that publishes them.
template <int DBG = 0>
__global__ __launch_bounds__(512, 2) void k_attention_d40_pp(AttnHP p) {
  constexpr int NW = 8, NTH = 64 * NW, QB = 32 * NW;
  constexpr int NSLOT = 2, NFULL = 2;                            // DMA wave-instructions per wave and tile: 2 for waves 0, 1, else 1
  constexpr int KROW = 80, KSUB = 32 * KROW, KBUF = 64 * KROW;   // 2560, 5120
  constexpr int ONES = 4 * KBUF;                                // constant pieces at ONES, ONES + KSUB
  constexpr int VBASE = 23552, VROW = 128, VBUF = 64 * VROW;    // 8192
  constexpr int LDS_BYTES = VBASE + 4 * VBUF;                   // 56320
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;                                      // waves w and w + 4 share a SIMD: the late half runs half a tile behind
  const int j = lane & 31, hh = lane >> 5;
  int bid = blockIdx.x;
  {
    const int nb = gridDim.x, xcd = bid & 7, qn = nb >> 3, r = nb & 7;
    bid = (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + (bid >> 3);
  }
  const int nqb = p.Tq / QB;
  const int bh = bid / nqb;
  const int b = bh / p.heads, hd = bh % p.heads;
  const int q0 = (bid - bh * nqb) * QB;
  constexpr int d = 40;

  for (int i = tid; i < LDS_BYTES / 16; i += NTH) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  // ---- Q fragments, pre-multiplied by scale * log2(e) (rounded once to fp16); column 40 carries -shift
  v8h qf[3];
  {
    const int qrow = q0 + wid * 32 + j;
    const float c2q = p.scale * 1.44269504088896340736f;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const int c = ks * 16 + hh * 8;
      if (c < d) v = *reinterpret_cast<const uint4*>(p.q + (static_cast<size_t>(b) * p.Tq + qrow) * p.ldq + hd * d + c);
      qf[ks] = *reinterpret_cast<v8h*>(&v);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[ks][e] = static_cast<_Float16>(static_cast<float>(qf[ks][e]) * c2q);
    }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) asm volatile("" : "+v"(qf[ks]));
  }

  // ---- DMA plan (as k_attention_d40<8>): wave-instructions 0..4 = a K tile, 5..9 = a V^T tile; wave w issues w and w + 8
  const unsigned char* kptr = reinterpret_cast<const unsigned char*>(p.k + static_cast<size_t>(b) * p.Tks * p.ldk + hd * d);
  const unsigned char* vptr = reinterpret_cast<const unsigned char*>(p.vt + (static_cast<size_t>(b) * p.heads + hd) * d * p.Tks);
  const size_t kstep = static_cast<size_t>(64) * p.ldk * 2;     // bytes per K tile (a V^T tile: 128)
  unsigned dma_off[NSLOT], slot_dst[NSLOT];
  bool slot_k[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int id = wid + NW * s;
    slot_k[s] = id < 5;
    if (id < 5) {
      const int n = id * 64 + lane, rho = n / 5, c = n - rho * 5;
      const int key = (rho & 0x33) | ((rho & 4) << 1) | ((rho & 8) >> 1);
      dma_off[s] = static_cast<unsigned>(key * p.ldk * 2 + c * 16);
      slot_dst[s] = id * 1024;
    } else {
      const int n = (id - 5) * 64 + lane, r = n >> 3, sl = n & 7;
      dma_off[s] = static_cast<unsigned>(r * p.Tks * 2 + ((sl ^ ((r >> 1) & 7)) << 4));
      slot_dst[s] = VBASE + (id - 5) * 1024;
    }
  }
  const bool slot2 = wid < NFULL;  // the second slot exists only for waves 0, 1
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
  unsigned slot_lds[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) slot_lds[s] = __builtin_amdgcn_readfirstlane(lds0 + slot_dst[s]);
  auto dma = [&](const unsigned char* kp, const unsigned char* vp, int kb, int vb, bool do_k, bool do_v) {
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      if (s == NSLOT - 1 && !slot2) continue;
      const bool isk = slot_k[s];          // wave-uniform
      if (isk ? do_k : do_v) glds16_s(isk ? kp : vp, dma_off[s], slot_lds[s] + (isk ? kb * KBUF : vb * VBUF));
    }
  };

  // ---- fragment addresses (bytes; + buffer offset as an immediate)
  const unsigned ka = j * KROW + hh * 16;                         // k-steps 0, 1 at +0, +32
  unsigned ka2[4];                                                // k-step 2: the upper lane half reads the constant piece
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) ka2[kb] = hh ? static_cast<unsigned>(ONES - kb * KBUF) : static_cast<unsigned>(j * KROW + 64);
  unsigned va[4];
  {
    const int g = (j >> 1) & 7;
#pragma unroll
    for (int u = 0; u < 4; ++u) va[u] = VBASE + j * VROW + (((2 * u + hh) ^ g) << 4);      // row j + 32 tt: + 4096 tt, same g
  }

  float m_run = 0.0f;
  v16f o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;

  __syncthreads();                 // zero fill done
  if (tid < 2) *reinterpret_cast<uint4*>(smem + ONES + tid * KSUB) = make_uint4(0x00003C00u, 0, 0, 0);
  if (tid >= 64 && tid < 96) {
    const uint4 ones = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);
    *reinterpret_cast<uint4*>(smem + VBASE + ((tid - 64) >> 3) * VBUF + 40 * VROW + ((tid - 64) & 7) * 16) = ones;
  }
  const int nt = p.Tk >> 6;        // a multiple of 4
  dma(kptr, vptr, 0, 0, true, true);                          // K(0), V(0)
  dma(kptr + kstep, vptr + 128, 1, 1, true, true);            // K(1), V(1)
  dma(kptr + 2 * kstep, vptr, 2, 0, true, false);             // K(2)
  dma(kptr + 3 * kstep, vptr + 256, 3, 2, true, true);        // K(3), V(2): the batch that may still fly behind the first barrier
  if (slot2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSLOT) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSLOT - 1) : "memory");

  // (volatile: the tree keeps its place behind the volatile s_nop that separates it from the MFMAs whose results it reads)
  auto row_max = [&](const v16f (&s)[2]) {
    float m0 = vmax3v(s[0][0], s[0][1], s[0][2]), m1 = vmax3v(s[0][3], s[0][4], s[0][5]);
    float m2 = vmax3v(s[1][0], s[1][1], s[1][2]), m3 = vmax3v(s[1][3], s[1][4], s[1][5]);
    m0 = vmax3v(m0, s[0][6], s[0][7]);   m1 = vmax3v(m1, s[0][8], s[0][9]);
    m2 = vmax3v(m2, s[1][6], s[1][7]);   m3 = vmax3v(m3, s[1][8], s[1][9]);
    m0 = vmax3v(m0, s[0][10], s[0][11]); m1 = vmax3v(m1, s[0][12], s[0][13]);
    m2 = vmax3v(m2, s[1][10], s[1][11]); m3 = vmax3v(m3, s[1][12], s[1][13]);
    m0 = vmax3v(m0, s[0][14], s[0][15]); m2 = vmax3v(m2, s[1][14], s[1][15]);
    m0 = vmax3v(m0, m1, m2);
    m0 = vmax3v(m0, m3, m3);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);   // lane ^ 32: the query's other 32 keys
    return vmax3v(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
  };
  constexpr int ksb = 2, qe = 0;       // Q column 40 = k-step 2, upper lane half, element 0

  v16f sc[2];                      // the scores of the tile whose softmax comes next
  {   // scores of tile 0 (every wave, before the halves part); its row maximum is the first shift (fp16-representable)
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[sub][r] = 0.0f;
      const v8h k0 = *reinterpret_cast<const v8h*>(smem + ka + sub * KSUB), k1 = *reinterpret_cast<const v8h*>(smem + ka + sub * KSUB + 32);
      const v8h k2 = *reinterpret_cast<const v8h*>(smem + ka2[0] + sub * KSUB);
      sc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[0], sc[sub], 0, 0, 0);
      sc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[1], sc[sub], 0, 0, 0);
      sc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k2, qf[2], sc[sub], 0, 0, 0);
    }
    asm volatile("s_nop 15" : "+v"(sc[0]), "+v"(sc[1]));     // MFMA results -> inline-asm VALU readers: the wait states the compiler cannot count
    const float mx = row_max(sc);
    m_run = static_cast<float>(static_cast<_Float16>(mx));
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[sub][r] -= m_run;
    if (hh) qf[ksb][qe] = static_cast<_Float16>(-m_run);
  }

#ifdef TFMQ_PHASE_TIMERS
  // [0] V segment up to its wait, [1] the counted wait, [2] barrier behind the V segment, [3] M segment (MFMA issue), [4] barrier behind it, [5] tiles
  unsigned long long kacc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long kt = clock64();
#define AKT(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = clock64(); kacc[i] += t_ - kt; kt = t_; } while (0)
#else
#define AKT(i) do { } while (0)
#endif
  unsigned pp[16];                 // packed fp16 P of the tile: pp[4 u + e] = keys 16 u + 8 hh + 2 e, +1
  v8h vf[4][2];                    // V^T fragments of the same tile
  v8h kf[2][3];                    // K fragments of the next tile

  // ---- V segment of tile t (B = t & 3): in sc the scores of tile t; out pp = P(t), vf = V^T(t) fragments, kf = K(t+1) fragments
  auto vseg = [&](auto b_tag, auto kdma_tag, auto vdma_tag, auto last_tag, auto counted_tag, const unsigned char* kp, const unsigned char* vp) {
    constexpr int B = decltype(b_tag)::value, KB = (B + 1) & 3;
    constexpr bool KDMA = decltype(kdma_tag)::value, VDMA = decltype(vdma_tag)::value, LAST = decltype(last_tag)::value;
    constexpr bool COUNTED = decltype(counted_tag)::value;
    if constexpr (!(DBG & 64)) dma(kp, vp, B, (B + 3) & 3, KDMA, VDMA);          // K(t+4) -> buffer t & 3, V^T(t+3) -> buffer (t+3) & 3
    if constexpr (!LAST) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        kf[sub][0] = *reinterpret_cast<const v8h*>(smem + KB * KBUF + ka + sub * KSUB);
        kf[sub][1] = *reinterpret_cast<const v8h*>(smem + KB * KBUF + ka + sub * KSUB + 32);
        kf[sub][2] = *reinterpret_cast<const v8h*>(smem + KB * KBUF + ka2[KB] + sub * KSUB);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) vf[u][tt] = *reinterpret_cast<const v8h*>(smem + B * VBUF + va[u] + tt * 32 * VROW);
    // (the maximum tree is inline asm reading MFMA results: the barrier and the DMA issue lie in between; pad for the iterations without DMA)
    asm volatile("s_nop 7");
    float mx;
    if constexpr (DBG & 16) mx = sc[0][3] + sc[1][5];
    else mx = row_max(sc);
    // ---- the re-base (rare): before any exponential of this tile is taken
    if (__builtin_amdgcn_ballot_w64(mx > 8.0f) != 0) {
      const float m_new = mx > 8.0f ? static_cast<float>(static_cast<_Float16>(m_run + mx)) : m_run;
      const float delta = m_new - m_run;                 // exact: both fp16-representable
      const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
      for (int i = 0; i < 32; ++i) sc[i >> 4][i & 15] -= delta;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[tt][r] *= alpha;
      m_run = m_new;
      if (hh) qf[ksb][qe] = static_cast<_Float16>(-m_run);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i0 = 16 * (u >> 1) + 8 * (u & 1) + 2 * e;
        float e0, e1;
        if constexpr (DBG & 1) { e0 = sc[i0 >> 4][i0 & 15]; e1 = sc[(i0 + 1) >> 4][(i0 + 1) & 15]; }
        else { e0 = __builtin_amdgcn_exp2f(sc[i0 >> 4][i0 & 15]); e1 = __builtin_amdgcn_exp2f(sc[(i0 + 1) >> 4][(i0 + 1) & 15]); }
        const __half2 h2 = __floats2half2_rn(e0, e1);
        pp[4 * u + e] = *reinterpret_cast<const unsigned*>(&h2);
      }
    // The packed P and the fragments are operands of an (empty) volatile statement in front of the barrier: otherwise the instruction
    // selector orders the exponentials, conversions and reads behind the barrier, into the M segment.
    asm volatile(""
                 : "+v"(pp[0]), "+v"(pp[1]), "+v"(pp[2]), "+v"(pp[3]), "+v"(pp[4]), "+v"(pp[5]), "+v"(pp[6]), "+v"(pp[7]),
                   "+v"(pp[8]), "+v"(pp[9]), "+v"(pp[10]), "+v"(pp[11]), "+v"(pp[12]), "+v"(pp[13]), "+v"(pp[14]), "+v"(pp[15]),
                   "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[2][0]), "+v"(vf[2][1]), "+v"(vf[3][0]), "+v"(vf[3][1]));
    if constexpr (!LAST)
      asm volatile("" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[0][2]), "+v"(kf[1][0]), "+v"(kf[1][1]), "+v"(kf[1][2]));
    __builtin_amdgcn_sched_barrier(0);
#ifdef TFMQ_PHASE_TIMERS
    AKT(0);
    if constexpr (COUNTED) {
      if (slot2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NSLOT) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NSLOT - 1)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    AKT(1);
    asm volatile("s_barrier" ::: "memory");
    AKT(2);
#else
    if constexpr (DBG & 256) {
    } else if constexpr (COUNTED) {
      if (slot2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NSLOT) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (NSLOT - 1)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- M segment of tile t: S(t+1) = K(t+1) Q^T into sc (its old content is spent), O += V^T(t) P(t)
  auto mseg = [&](auto last_tag, bool tail_barrier) {
    constexpr bool LAST = decltype(last_tag)::value;
    if constexpr (!LAST) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[sub][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          if constexpr (DBG & 4) sc[sub][ks] += static_cast<float>(kf[sub][ks][0]);
          else sc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[sub][ks], qf[ks], sc[sub], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v8h bp;
      unsigned* bw = reinterpret_cast<unsigned*>(&bp);
#pragma unroll
      for (int e = 0; e < 4; ++e) bw[e] = pp[4 * u + e];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if constexpr (DBG & 2) o[tt][u] += static_cast<float>(vf[u][tt][0]) + __uint_as_float(bw[tt]);
        else o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[u][tt], bp, o[tt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef TFMQ_PHASE_TIMERS
    asm volatile("" : "+v"(o[0]), "+v"(o[1]));
    AKT(3);
#endif
    if constexpr (!(DBG & 256)) {
      if (tail_barrier) asm volatile("s_barrier" ::: "memory");
    }
#ifdef TFMQ_PHASE_TIMERS
    AKT(4);
    kacc[5] += 1;
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using Y = std::true_type;
  using N = std::false_type;
  if constexpr (!(DBG & 256)) {
    if (grp == 1) asm volatile("s_barrier" ::: "memory");       // the late half: half a tile behind
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned char* kp = kptr + 4 * kstep;     // K(t + 4)
  const unsigned char* vp = vptr + 384;           // V^T(t + 3)
  for (int t = 0; t + 4 < nt; t += 4) {
    vseg(I0{}, Y{}, Y{}, N{}, Y{}, kp, vp);                       mseg(N{}, true);
    vseg(I1{}, Y{}, Y{}, N{}, Y{}, kp + kstep, vp + 128);         mseg(N{}, true);
    vseg(I2{}, Y{}, Y{}, N{}, Y{}, kp + 2 * kstep, vp + 256);     mseg(N{}, true);
    vseg(I3{}, Y{}, Y{}, N{}, Y{}, kp + 3 * kstep, vp + 384);     mseg(N{}, true);
    kp += 4 * kstep;
    vp += 512;
  }
  vseg(I0{}, N{}, Y{}, N{}, N{}, kp, vp);     mseg(N{}, true);     // t = nt-4: V^T(nt-1); the last tiles drain the queue
  vseg(I1{}, N{}, N{}, N{}, N{}, kp, vp);     mseg(N{}, true);     // t = nt-3
  vseg(I2{}, N{}, N{}, N{}, N{}, kp, vp);     mseg(N{}, true);     // t = nt-2
  vseg(I3{}, N{}, N{}, Y{}, N{}, kp, vp);     mseg(Y{}, grp == 0); // t = nt-1: no scores left to compute; the late half's last M segment has no partner

#ifdef TFMQ_PHASE_TIMERS
  if (p.dbg && (tid == 0 || tid == 256))
    for (int i = 0; i < 6; ++i) p.dbg[(static_cast<size_t>(blockIdx.x) * 2 + (tid >> 8)) * 8 + i] = kacc[i];
#endif
  // ---- normalise and store (as k_attention_d40)
  const int qg = q0 + wid * 32 + j;
  float l_run;
  {
    constexpr int lr = 40 % 32, hh_one = (lr >> 2) & 1, r_one = (lr & 3) + 4 * (lr >> 3);
    const float mine = o[1][r_one];
    const float other = __shfl_xor(mine, 32, 64);
    l_run = hh == hh_one ? mine : other;
  }
  const float inv = 1.0f / l_run;
  const bool quant = p.yq != nullptr;
  float2 qp = make_float2(1.0f, 0.0f);
  if (quant) qp = load_qparam(p.aq);
  const size_t tok = static_cast<size_t>(b) * p.Tq + qg;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dc = tt * 32 + 8 * g + 4 * hh;
      if (dc >= d) continue;
      float4 v = make_float4(o[tt][4 * g] * inv, o[tt][4 * g + 1] * inv, o[tt][4 * g + 2] * inv, o[tt][4 * g + 3] * inv);
      if (p.out) *reinterpret_cast<float4*>(p.out + tok * p.ldo + hd * d + dc) = v;
      if (quant) {
        char4 c = quant_char4(v.x, v.y, v.z, v.w, make_quantp(qp));
        *reinterpret_cast<char4*>(p.yq + tok * (static_cast<size_t>(p.heads) * d) + hd * d + dc) = c;
      }
    }
  }
}

static int launch_attn_d40(tfmq_handle h, const AttnHP& p, void* stream) {
  static const int nw = getenv("TFMQ_ATTN_PIPE_NW") ? atoi(getenv("TFMQ_ATTN_PIPE_NW")) : 4;
  static const int pp = getenv("TFMQ_ATTN_PP") ? atoi(getenv("TFMQ_ATTN_PP")) : 0;        // ping-pong form (round 6; measured 8 % SLOWER: opt-in)
  if (pp && p.Tq % 256 == 0) {
    dim3 gridp(static_cast<unsigned>(p.Tq / 256) * p.B * p.heads);
#ifdef TFMQ_ATTN_ABLATE
    static const int dbgp = getenv("TFMQ_ATTN_DBG") ? atoi(getenv("TFMQ_ATTN_DBG")) : 0;
#define TFMQ_ABLP(D) if (dbgp == D) { hipLaunchKernelGGL((k_attention_d40_pp<D>), gridp, dim3(512), 0, as_stream(stream), p); TFMQ_LAUNCH_CHECK(h); return TFMQ_OK; }
    TFMQ_ABLP(1) TFMQ_ABLP(2) TFMQ_ABLP(4) TFMQ_ABLP(6) TFMQ_ABLP(16) TFMQ_ABLP(17) TFMQ_ABLP(64) TFMQ_ABLP(256)
#endif
#ifdef TFMQ_PHASE_TIMERS
    {
      static unsigned long long* dbuf = nullptr;
      if (!dbuf) (void)hipMalloc(reinterpret_cast<void**>(&dbuf), sizeof(unsigned long long) * 16 * (1u << 16));
      AttnHP pd = p;
      pd.dbg = gridp.x <= (1u << 16) ? dbuf : nullptr;
      hipLaunchKernelGGL((k_attention_d40_pp<0>), gridp, dim3(512), 0, as_stream(stream), pd);
      if (pd.dbg && getenv("TFMQ_PHASE_PRINT")) {
        (void)hipStreamSynchronize(as_stream(stream));
        std::vector<unsigned long long> hb(static_cast<size_t>(gridp.x) * 16);
        (void)hipMemcpy(hb.data(), dbuf, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double a[2][6] = {{0}};
        for (unsigned i = 0; i < gridp.x; ++i)
          for (int g = 0; g < 2; ++g)
            for (int q = 0; q < 6; ++q) a[g][q] += double(hb[(static_cast<size_t>(i) * 2 + g) * 8 + q]);
        const double n0 = a[0][5] > 0 ? a[0][5] : 1, n1 = a[1][5] > 0 ? a[1][5] : 1;
        fprintf(stderr, "[attention_d40_pp B%d h%d T%d] per key tile, shader cycles, wave 0 | wave 4: V segment %.0f | %.0f, vmcnt wait %.0f | %.0f, barrier (V) %.0f | %.0f, M segment %.0f | %.0f, barrier (M) %.0f | %.0f = %.0f | %.0f\n",
                p.B, p.heads, p.Tq, a[0][0] / n0, a[1][0] / n1, a[0][1] / n0, a[1][1] / n1, a[0][2] / n0, a[1][2] / n1, a[0][3] / n0, a[1][3] / n1, a[0][4] / n0, a[1][4] / n1,
                (a[0][0] + a[0][1] + a[0][2] + a[0][3] + a[0][4]) / n0, (a[1][0] + a[1][1] + a[1][2] + a[1][3] + a[1][4]) / n1);
      }
      TFMQ_LAUNCH_CHECK(h);
      return TFMQ_OK;
    }
#endif
    hipLaunchKernelGGL((k_attention_d40_pp<0>), gridp, dim3(512), 0, as_stream(stream), p);
    TFMQ_LAUNCH_CHECK(h);
    return TFMQ_OK;
  }
  if (nw == 8 && p.Tq % 256 == 0) {
    dim3 grid8(static_cast<unsigned>(p.Tq / 256) * p.B * p.heads);
#ifdef TFMQ_ATTN_ABLATE
    static const int dbg8 = getenv("TFMQ_ATTN_DBG") ? atoi(getenv("TFMQ_ATTN_DBG")) : 0;
#define TFMQ_ABL8(D) if (dbg8 == D) { hipLaunchKernelGGL((k_attention_d40<8, D>), grid8, dim3(512), 0, as_stream(stream), p); TFMQ_LAUNCH_CHECK(h); return TFMQ_OK; }
    TFMQ_ABL8(1) TFMQ_ABL8(6) TFMQ_ABL8(8) TFMQ_ABL8(64) TFMQ_ABL8(49) TFMQ_ABL8(14) TFMQ_ABL8(78) TFMQ_ABL8(128)
#endif
    hipLaunchKernelGGL((k_attention_d40<8>), grid8, dim3(512), 0, as_stream(stream), p);
    TFMQ_LAUNCH_CHECK(h);
    return TFMQ_OK;
  }
  dim3 grid(static_cast<unsigned>(p.Tq / 128) * p.B * p.heads);
#ifdef TFMQ_ATTN_ABLATE
  static const int dbg = getenv("TFMQ_ATTN_DBG") ? atoi(getenv("TFMQ_ATTN_DBG")) : 0;
#define TFMQ_ABL(D) if (dbg == D) { hipLaunchKernelGGL((k_attention_d40<4, D>), grid, dim3(256), 0, as_stream(stream), p); TFMQ_LAUNCH_CHECK(h); return TFMQ_OK; }
  TFMQ_ABL(1) TFMQ_ABL(2) TFMQ_ABL(4) TFMQ_ABL(6) TFMQ_ABL(8) TFMQ_ABL(16) TFMQ_ABL(32) TFMQ_ABL(64) TFMQ_ABL(49) TFMQ_ABL(14) TFMQ_ABL(78) TFMQ_ABL(128) TFMQ_ABL(328) TFMQ_ABL(456) TFMQ_ABL(320) TFMQ_ABL(256) TFMQ_ABL(72) TFMQ_ABL(2048) TFMQ_ABL(4096) TFMQ_ABL(6144)
  if (dbg == 1000) { hipLaunchKernelGGL((k_attention_d40<4, 0, 3>), grid, dim3(256), 0, as_stream(stream), p); TFMQ_LAUNCH_CHECK(h); return TFMQ_OK; }
#endif
  static const int acc = getenv("TFMQ_ATTN_ACC") ? atoi(getenv("TFMQ_ATTN_ACC")) : 0;
  static const int lds3 = getenv("TFMQ_ATTN_LDS3") ? atoi(getenv("TFMQ_ATTN_LDS3")) : 1;      // three blocks per CU (round 6)
  if (acc) hipLaunchKernelGGL((k_attention_d40<4, 0, 1, true>), grid, dim3(256), 0, as_stream(stream), p);
  else if (lds3) hipLaunchKernelGGL((k_attention_d40<4, 0, 2, false, true>), grid, dim3(256), 0, as_stream(stream), p);
  else hipLaunchKernelGGL((k_attention_d40<4>), grid, dim3(256), 0, as_stream(stream), p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
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
  if (!out && yq) {       // a short context (cross attention over the 77 CLIP tokens) with int8 output: all heads per workgroup (attention_ctx.hip)
    bool taken = false;
    const int rc = launch_attention_ctx(h, q, k, vt, ldq, ldk, yq, aq, B, heads, Tq, Tk, Tk_stride, d, scale, stream, &taken);
    if (taken) return rc;
  }
  if (d <= 32) return launch_attn_h<2, 1>(h, p, stream);
  if (d == 40) {   // SD v1 at 64x64
    static const bool pipe = !(getenv("TFMQ_ATTN_PIPE") && atoi(getenv("TFMQ_ATTN_PIPE")) == 0);
    if (pipe && Tq % 128 == 0 && Tk % 256 == 0) return launch_attn_d40(h, p, stream);
    return launch_attn_h<3, 2, 40, true>(h, p, stream);
  }
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
