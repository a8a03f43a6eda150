// K15 (reconstruction): exact-fp32 fused attention, forward and backward, on the fp32 matrix cores
// (v_mfma_f32_32x32x2f32: fp32 products, fp32 accumulation -- the arithmetic of the strided-GEMM path it replaces).
// The block reconstruction of a BasicTransformerBlock (quant/reconstruction.py:86-209 on quant_block.py:248-299)
// runs softmax(Q K^T / sqrt(d)) V and its backward 20 000 times per unit; through GEMMs the 4096-token self attention
// of the SD 64x64 level materialises S, P, dP and dS (4 x 4.3 GB at mini-batch 8), which is most of the iteration.
// Here nothing of size T x T touches memory: the forward keeps the row log-sum-exp, the backward recomputes the
// probabilities tile by tile (two kernels, no atomics: one owns key blocks -> dK, dV; one owns query blocks -> dQ).
//
// Layouts: q [B][Tq][ldq], k / v [B][Tk][ldk] with head h at channels h*d .. h*d+d-1 (the packed [B,T,heads*d] tensors
// of the unit), d in {32, 40, 64, 80}, Tq a multiple of 32, any Tk (a ragged last key tile is masked).  Probabilities live in the exp2 domain:
// p = exp2(c2 * s - lse), c2 = scale * log2(e), lse[b][h][q] = m + log2(sum exp2(c2 s - m)).
//
// MFMA operand map (32x32x2): A lane (row = lane&31, k = lane>>5), B lane (col = lane&31, k = lane>>5); accumulator
// lane (col = lane&31) holds rows (r&3) + 8*(r>>2) + 4*(lane>>5).  A product whose contraction index is the row index
// of an accumulator tile takes that tile straight from registers as its B operand: step s contracts rows
// {rowmap(s, 0), rowmap(s, 1)}, rowmap(s, half) = (s&3) + 8*(s>>2) + 4*half.
#include "common.hpp"

struct AttnF32P {
  const float *q, *k, *v;
  int ldq, ldk;
  float* out;            // fwd: O; bwd: unused
  int ldo;
  float* lse;            // [B][heads][Tq]
  const float *dout;     // bwd: dO [B][Tq][ldo]
  const float *dsum;     // bwd: D[b][h][q] = sum_j dO[q][j] O[q][j]
  float *dq, *dk, *dv;   // bwd outputs (dq uses ldq, dk / dv use ldk)
  int B, heads, Tq, Tk, d;
  float scale;
  int qsplit = 1;        // k_attn_bx3_bwd_kv: query-range slices per key block (short key sequences: cross attention on 77 tokens)
  float* part = nullptr; // [2][qsplit][B][Tk][heads * d] partial dV | dK (unscaled) of the slices, summed in slice order afterwards
};

// floats per LDS row: odd -> column-wise fragment reads are conflict free (65 for d <= 64, 97 for d <= 96)
#define PITCH_OF(NT) ((NT) * 32 + 1)

__device__ __forceinline__ int rowmap(int s, int half) { return (s & 3) + 8 * (s >> 2) + 4 * half; }

// copy a [32][d] tile (row stride ld floats; rows >= nrows are zero) into LDS [32][PITCH]; d % 4 == 0
template <int PITCH>
__device__ __forceinline__ void stage32(float* lds, const float* src, long ld, int d, int tid, int nrows = 32) {
  const int per = d >> 2;                  // float4 per row
  for (int i = tid; i < 32 * per; i += 256) {
    const int r = i / per, c = (i - r * per) * 4;
    const float4 v = r < nrows ? *reinterpret_cast<const float4*>(src + r * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* p = lds + r * PITCH + c;
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
  }
}

// The same tile copy split in two: buffer loads into registers at the top of a loop iteration (rows past `nrows` --
// a ragged or non-existent next tile -- get an offset beyond the descriptor's extent and read zeros: no branch, so the
// compiler does not wait for the loads where they are issued), LDS stores after the iteration's MFMAs.  With the
// one-piece copy every wave sat out the full global latency of the next tile before it started multiplying.
typedef int v4i32 __attribute__((ext_vector_type(4)));
template <int KD, int PITCH>
struct TileStage {
  static constexpr int PER = KD / 2, N = 32 * PER, ITEMS = (N + 255) / 256;   // float4 per row / per tile / per thread
  float4 r[ITEMS];
  __amdgpu_buffer_rsrc_t rsrc;
  long ld;
  __device__ __forceinline__ void init(const float* base, long ld_, long extent_elems) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, static_cast<int>(extent_elems * 4), 0x00020000);
    ld = ld_;
  }
  __device__ __forceinline__ void load(int row0, int nrows) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int i = threadIdx.x + it * 256;
      const int rr = i / PER, c = (i - rr * PER) * 4;
      const bool ok = i < N && rr < nrows;
      const unsigned off = ok ? static_cast<unsigned>(((row0 + rr) * ld + c) * 4) : 0x80000000u;
      r[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
  }
  __device__ __forceinline__ void store(float* lds) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int i = threadIdx.x + it * 256;
      if (i >= N) continue;
      const int rr = i / PER, c = (i - rr * PER) * 4;
      float* q = lds + rr * PITCH + c;
      q[0] = r[it].x; q[1] = r[it].y; q[2] = r[it].z; q[3] = r[it].w;
    }
  }
  // bf16x3 kernels: the tile leaves for LDS already split, hi = bf16(x), lo = bf16(x - hi), once per block instead of once per wave
  // and fragment.  rw_* : row-wise [32][pr bytes] (element (row, c) at row * pr + 2 c) for the products that contract over the
  // channels; tr_* : transposed [channel][80 bytes] with the tile row r at slot_of_row(r) for the products that contract over the
  // tile's rows (the order in which the score registers hold them).  Null pointers skip a layout.
  __device__ __forceinline__ void store_split(unsigned char* rw_hi, unsigned char* rw_lo, int pr, unsigned char* tr_hi, unsigned char* tr_lo);
};

// ---------------------------------------------------------------------------------------------- bf16x3 operand form
// (tfmq_set_gemm_precision(1), the default of the reconstruction iterations.)  Every fp32 operand value a is split as hi = bf16(a),
// lo = bf16(a - hi) and a b ~ hi hi' + hi lo' + lo hi' on v_mfma_f32_32x32x16_bf16 (fp32 accumulation; 2^-16 relative per product, as in
// k_gemm_f32_mfma): 9 + 12 MFMAs of 32 cycles per 32-key tile at d = 40 instead of 20 + 32 fp32 MFMAs of 64.  Same decomposition, same
// masks, same row terms as the fp32 kernels below.  Operand map (32x32x16): A lane (row = lane & 31) holds k = 8 (lane >> 5) .. + 7,
// B lane (col = lane & 31) likewise; the accumulator layout is that of 32x32x2.  A product that contracts over the rows of a score
// tile takes the tile from the accumulator registers as B: step z of two holds rows rowmap(8 z + i, half), i < 8 -- the transposed LDS
// images store tile row r at slot_of_row(r) so that the A fragment of that step is 16 contiguous bytes.
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v16f_ __attribute__((ext_vector_type(16)));
struct HL8 { v8bf hi, lo; };
__device__ __forceinline__ int slot_of_row(int r) {
  const int x = (r & 3) + 4 * (r >> 3);
  return 16 * (x >> 3) + 8 * ((r >> 2) & 1) + (x & 7);
}
__device__ __forceinline__ HL8 split8(const float* x) {
  HL8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hv = static_cast<__bf16>(x[e]);
    o.hi[e] = hv;
    o.lo[e] = static_cast<__bf16>(x[e] - static_cast<float>(hv));
  }
  return o;
}
__device__ __forceinline__ v16f_ mma3(const v8bf ah, const v8bf al, const HL8& b, v16f_ acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.lo, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.hi, acc, 0, 0, 0);
}
template <int KD, int PITCH>
__device__ __forceinline__ void TileStage<KD, PITCH>::store_split(unsigned char* rw_hi, unsigned char* rw_lo, int pr, unsigned char* tr_hi,
                                                                   unsigned char* tr_lo) {
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int i = threadIdx.x + it * 256;
    if (i >= N) continue;
    const int rr = i / PER, c = (i - rr * PER) * 4;
    const float x[4] = {r[it].x, r[it].y, r[it].z, r[it].w};
    __bf16 hv[4], lv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hv[e] = static_cast<__bf16>(x[e]);
      lv[e] = static_cast<__bf16>(x[e] - static_cast<float>(hv[e]));
    }
    if (rw_hi) {
      typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
      *reinterpret_cast<v4bf*>(rw_hi + rr * pr + 2 * c) = v4bf{hv[0], hv[1], hv[2], hv[3]};
      *reinterpret_cast<v4bf*>(rw_lo + rr * pr + 2 * c) = v4bf{lv[0], lv[1], lv[2], lv[3]};
    }
    if (tr_hi) {
      const int sl = 2 * slot_of_row(rr);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *reinterpret_cast<__bf16*>(tr_hi + (c + e) * 80 + sl) = hv[e];
        *reinterpret_cast<__bf16*>(tr_lo + (c + e) * 80 + sl) = lv[e];
      }
    }
  }
}
// 8 consecutive channels 16 st + 8 half .. of a per-lane row vector (scaled), zero beyond d
__device__ __forceinline__ HL8 row_frag(const float* row, int st, int half, int d, float scale, bool ok) {
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 16 * st + 8 * half + e;
    x[e] = (ok && c < d) ? row[c] * scale : 0.0f;
  }
  return split8(x);
}
#define BX3_RW(base, row, st, half, PR) (*reinterpret_cast<const v8bf*>((base) + (row) * (PR) + 32 * (st) + 16 * (half)))
#define BX3_TR(base, ch, z, half) (*reinterpret_cast<const v8bf*>((base) + (ch) * 80 + 32 * (z) + 16 * (half)))

// forward: block = 128 queries of one (batch, head), 32-key tiles; K row-wise, V transposed
template <int KD, int NS, int NT>
__global__ __launch_bounds__(256) void k_attn_bx3_fwd(AttnF32P p) {
  constexpr int PR = 32 * NS + 16, RWB = 32 * PR, TRB = 32 * NT * 80;
  __shared__ __attribute__((aligned(16))) unsigned char sK[2][2][RWB];      // [buffer][hi | lo]
  __shared__ __attribute__((aligned(16))) unsigned char sV[2][2][TRB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 31, half = lane >> 5;
  const int nqb = p.Tq / 128 + (p.Tq % 128 ? 1 : 0);
  const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int d = p.d;
  const int q_row = qb * 128 + wid * 32 + j;
  const bool q_ok = q_row < p.Tq;
  const float c2 = p.scale * 1.44269504088896340736f;
  for (int i = tid * 4; i < static_cast<int>(sizeof(sK)); i += 1024) *reinterpret_cast<unsigned*>(&sK[0][0][0] + i) = 0u;     // channel padding
  for (int i = tid * 4; i < static_cast<int>(sizeof(sV)); i += 1024) *reinterpret_cast<unsigned*>(&sV[0][0][0] + i) = 0u;
  HL8 qs[NS];
  {
    const float* qp = p.q + (static_cast<long>(b) * p.Tq + (q_ok ? q_row : 0)) * p.ldq + h * d;
#pragma unroll
    for (int st = 0; st < NS; ++st) qs[st] = row_frag(qp, st, half, d, c2, q_ok);
  }
  v16f_ o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  const float* kb = p.k + static_cast<long>(b) * p.Tk * p.ldk + h * d;
  const float* vb = p.v + static_cast<long>(b) * p.Tk * p.ldk + h * d;
  const int ntile = (p.Tk + 31) / 32;
  TileStage<KD, 1> tk, tv;
  tk.init(kb, p.ldk, static_cast<long>(p.Tk - 1) * p.ldk + d);
  tv.init(vb, p.ldk, static_cast<long>(p.Tk - 1) * p.ldk + d);
  tk.load(0, p.Tk);
  tv.load(0, p.Tk);
  __syncthreads();                                   // (zero fill done)
  tk.store_split(sK[0][0], sK[0][1], PR, nullptr, nullptr);
  tv.store_split(nullptr, nullptr, 0, sV[0][0], sV[0][1]);
  __syncthreads();
  for (int kt = 0; kt < ntile; ++kt) {
    const int buf = kt & 1;
    tk.load((kt + 1) * 32, p.Tk - (kt + 1) * 32);
    tv.load((kt + 1) * 32, p.Tk - (kt + 1) * 32);
    __builtin_amdgcn_sched_barrier(0);
    v16f_ s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
    for (int st = 0; st < NS; ++st) s = mma3(BX3_RW(sK[buf][0], j, st, half, PR), BX3_RW(sK[buf][1], j, st, half, PR), qs[st], s);
    if ((kt + 1) * 32 > p.Tk) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * 32 + rowmap(r, half) >= p.Tk) s[r] = -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    float rs = 0.0f, pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      rs += pv[r];
    }
    rs += __shfl_xor(rs, 32, 64);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
    for (int z = 0; z < 2; ++z) {
      const HL8 ps = split8(pv + 8 * z);
#pragma unroll
      for (int t = 0; t < NT; ++t) o[t] = mma3(BX3_TR(sV[buf][0], t * 32 + j, z, half), BX3_TR(sV[buf][1], t * 32 + j, z, half), ps, o[t]);
    }
    tk.store_split(sK[buf ^ 1][0], sK[buf ^ 1][1], PR, nullptr, nullptr);
    tv.store_split(nullptr, nullptr, 0, sV[buf ^ 1][0], sV[buf ^ 1][1]);
    __syncthreads();
  }
  if (!q_ok) return;
  const float inv = 1.0f / l_run;
  float* op = p.out + (static_cast<long>(b) * p.Tq + q_row) * p.ldo + h * d;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dd = t * 32 + 8 * g + 4 * half;
      if (dd < d) *reinterpret_cast<float4*>(op + dd) = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
    }
  if (half == 0) p.lse[(static_cast<long>(b) * p.heads + h) * p.Tq + q_row] = m_run + __builtin_amdgcn_logf(l_run);
}

// ---------------------------------------------------------------------------------------------- forward
// block = 128 queries (4 waves x 32) of one (batch, head); loops over 32-key tiles (double-buffered K, V in LDS)
template <int KD, int NT>   // d / 2 score MFMA steps; 32-row tiles covering d
__global__ __launch_bounds__(256) void k_attn_f32_fwd(AttnF32P p) {
  constexpr int PITCH = PITCH_OF(NT);
  __shared__ float sK[2][32 * PITCH];
  __shared__ float sV[2][32 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 31, half = lane >> 5;
  const int nqb = p.Tq / 128 + (p.Tq % 128 ? 1 : 0);
  const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int d = p.d;
  const int q_row = qb * 128 + wid * 32 + j;
  const bool q_ok = q_row < p.Tq;
  const float c2 = p.scale * 1.44269504088896340736f;
  float qf[KD];
  {
    const float* qp = p.q + (static_cast<long>(b) * p.Tq + (q_ok ? q_row : 0)) * p.ldq + h * d;
#pragma unroll
    for (int s = 0; s < KD; ++s) qf[s] = q_ok ? qp[2 * s + half] * c2 : 0.0f;
  }
  typedef float v16f __attribute__((ext_vector_type(16)));
  v16f o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  const float* kb = p.k + static_cast<long>(b) * p.Tk * p.ldk + h * d;
  const float* vb = p.v + static_cast<long>(b) * p.Tk * p.ldk + h * d;
  const int ntile = (p.Tk + 31) / 32;      // the last tile may be ragged: its missing keys are staged as zeros and masked
  TileStage<KD, PITCH> tk, tv;
  tk.init(kb, p.ldk, static_cast<long>(p.Tk - 1) * p.ldk + d);
  tv.init(vb, p.ldk, static_cast<long>(p.Tk - 1) * p.ldk + d);
  tk.load(0, p.Tk);
  tv.load(0, p.Tk);
  tk.store(sK[0]);
  tv.store(sV[0]);
  __syncthreads();
  for (int kt = 0; kt < ntile; ++kt) {
    const int buf = kt & 1;
    tk.load((kt + 1) * 32, p.Tk - (kt + 1) * 32);
    tv.load((kt + 1) * 32, p.Tk - (kt + 1) * 32);
    __builtin_amdgcn_sched_barrier(0);
    // S^T = K Q^T (already times c2)
    v16f s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
    for (int st = 0; st < KD; ++st)
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[buf][j * PITCH + 2 * st + half], qf[st], s, 0, 0, 0);
    if ((kt + 1) * 32 > p.Tk) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * 32 + rowmap(r, half) >= p.Tk) s[r] = -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      rs += s[r];
    }
    rs += __shfl_xor(rs, 32, 64);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // exp2(-inf) = 0 on the first tile
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // O^T += V^T P^T : contraction over the 32 keys, two per step, straight from the score registers
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int key = rowmap(st, half);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int dd = t * 32 + j;
        const float a = dd < d ? sV[buf][key * PITCH + dd] : 0.0f;
        o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[st], o[t], 0, 0, 0);
      }
    }
    tk.store(sK[buf ^ 1]);      // (past the last tile: zeros, into the buffer nobody reads again)
    tv.store(sV[buf ^ 1]);
    __syncthreads();
  }
  if (!q_ok) return;
  const float inv = 1.0f / l_run;
  float* op = p.out + (static_cast<long>(b) * p.Tq + q_row) * p.ldo + h * d;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dd = t * 32 + 8 * g + 4 * half;
      if (dd < d) *reinterpret_cast<float4*>(op + dd) = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
    }
  if (half == 0) p.lse[(static_cast<long>(b) * p.heads + h) * p.Tq + q_row] = m_run + __builtin_amdgcn_logf(l_run);   // v_log_f32 = log2
}

extern "C" int tfmq_attention_f32_fwd(tfmq_handle h, const float* q, const float* k, const float* v, int ldq, int ldk,
                                      float* out, int ldo, float* lse, int B, int heads, int Tq, int Tk, int d, float scale,
                                      void* stream) {
  TFMQ_CHECK_ARG(h, h && q && k && v && out && lse, "attention_f32_fwd: null pointer");
  TFMQ_CHECK_ARG(h, B > 0 && heads > 0 && Tq > 0 && Tk > 0 && Tq % 32 == 0, "attention_f32_fwd: Tq must be a multiple of 32");
  TFMQ_CHECK_ARG(h, d == 40 || d == 64 || d == 32 || d == 80, "attention_f32_fwd: head dim 32, 40, 64 or 80");
  TFMQ_CHECK_ARG(h, ldq % 4 == 0 && ldk % 4 == 0 && ldo % 4 == 0, "attention_f32_fwd: leading dims must be multiples of 4");
  TFMQ_CHECK_ARG(h, static_cast<long>(Tk) * ldk < (1L << 28) && static_cast<long>(Tq) * ldq < (1L << 28),
                 "attention_f32_fwd: one batch item's q / k / v must stay below 1 GiB (32-bit buffer offsets)");
  AttnF32P p{q, k, v, ldq, ldk, out, ldo, lse, nullptr, nullptr, nullptr, nullptr, nullptr, B, heads, Tq, Tk, d, scale};
  dim3 grid(static_cast<unsigned>((Tq + 127) / 128) * B * heads);
  hipStream_t st = as_stream(stream);
  if (h->gemm_prec == 1 && !getenv("TFMQ_ATTN_F32_EXACT")) {       // the reconstruction iterations' bf16x3 operand mode
    if (d == 40) hipLaunchKernelGGL((k_attn_bx3_fwd<20, 3, 2>), grid, dim3(256), 0, st, p);
    else if (d == 32) hipLaunchKernelGGL((k_attn_bx3_fwd<16, 2, 1>), grid, dim3(256), 0, st, p);
    else if (d == 64) hipLaunchKernelGGL((k_attn_bx3_fwd<32, 4, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_attn_bx3_fwd<40, 5, 3>), grid, dim3(256), 0, st, p);
    TFMQ_LAUNCH_CHECK(h);
    return TFMQ_OK;
  }
  if (d == 40) hipLaunchKernelGGL((k_attn_f32_fwd<20, 2>), grid, dim3(256), 0, st, p);
  else if (d == 32) hipLaunchKernelGGL((k_attn_f32_fwd<16, 1>), grid, dim3(256), 0, st, p);
  else if (d == 64) hipLaunchKernelGGL((k_attn_f32_fwd<32, 2>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_attn_f32_fwd<40, 3>), grid, dim3(256), 0, st, p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ---------------------------------------------------------------------------------------------- backward
// D[b][h][q] = sum_j dO[b][q][h*d+j] * O[b][q][h*d+j]  (= sum_k P_qk dP_qk, the softmax-backward row term)
__global__ void k_attn_f32_rowdot(const float* __restrict__ o, const float* __restrict__ dout, int ldo, float* __restrict__ dsum,
                                  int B, int heads, int Tq, int d) {
  const long total = static_cast<long>(B) * heads * Tq;
  for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * blockDim.x) {
    const int qi = static_cast<int>(i % Tq);
    const long bh = i / Tq;
    const int h = static_cast<int>(bh % heads), b = static_cast<int>(bh / heads);
    const long off = (static_cast<long>(b) * Tq + qi) * ldo + h * d;
    float a = 0.0f;
    for (int jx = 0; jx < d; jx += 4) {
      const float4 x = *reinterpret_cast<const float4*>(o + off + jx), y = *reinterpret_cast<const float4*>(dout + off + jx);
      a += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
    dsum[i] = a;
  }
}

// dK, dV: block = 128 keys (4 waves x 32) of one (batch, head); loops over 32-query tiles (Q, dO double-buffered in LDS).
// Score tile S = Q K^T with col = key, rows = queries, so P and dS feed the two accumulating products as B operands:
//   dV^T += dO^T P,   dP = dO V^T,   dS = P o (dP - D),   dK^T += Q^T dS   (dK scaled by `scale` at the end)
template <int KD, int NT>
__global__ __launch_bounds__(256) void k_attn_f32_bwd_kv(AttnF32P p) {
  constexpr int PITCH = PITCH_OF(NT);
  __shared__ float sQ[2][32 * PITCH];
  __shared__ float sO[2][32 * PITCH];
  __shared__ __attribute__((aligned(16))) float sL[2][32];
  __shared__ __attribute__((aligned(16))) float sD[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 31, half = lane >> 5;
  const int nkb = (p.Tk + 127) / 128;
  const int bh = blockIdx.x / nkb, kbk = blockIdx.x - bh * nkb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int d = p.d;
  const int key = kbk * 128 + wid * 32 + j;
  const bool k_ok = key < p.Tk;
  const float c2 = p.scale * 1.44269504088896340736f;
  float kf[KD], vf[KD];
  {
    const long off = (static_cast<long>(b) * p.Tk + (k_ok ? key : 0)) * p.ldk + h * d;
#pragma unroll
    for (int s = 0; s < KD; ++s) {
      kf[s] = k_ok ? p.k[off + 2 * s + half] * c2 : 0.0f;
      vf[s] = k_ok ? p.v[off + 2 * s + half] : 0.0f;
    }
  }
  typedef float v16f __attribute__((ext_vector_type(16)));
  v16f dv[NT], dk[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[t][r] = dk[t][r] = 0.0f;
  const float* qb = p.q + static_cast<long>(b) * p.Tq * p.ldq + h * d;
  const float* ob = p.dout + static_cast<long>(b) * p.Tq * p.ldo + h * d;
  const float* lb = p.lse + (static_cast<long>(b) * p.heads + h) * p.Tq;
  const float* db = p.dsum + (static_cast<long>(b) * p.heads + h) * p.Tq;
  const int ntile = p.Tq / 32;
  TileStage<KD, PITCH> tq, to;
  tq.init(qb, p.ldq, static_cast<long>(p.Tq - 1) * p.ldq + d);
  to.init(ob, p.ldo, static_cast<long>(p.Tq - 1) * p.ldo + d);
  // row terms of the tile: threads 0-31 carry lse, 32-63 carry D (everybody else reads zeros past the extent)
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lb), 0, p.Tq * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(db), 0, p.Tq * 4, 0x00020000);
  float l_reg, d_reg;
  auto load = [&](int qt) {
    tq.load(qt * 32, p.Tq - qt * 32);
    to.load(qt * 32, p.Tq - qt * 32);
    const unsigned off = static_cast<unsigned>((qt * 32 + (tid & 31)) * 4);      // qt == ntile: past the extent
    l_reg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, tid < 32 ? off : 0x80000000u, 0, 0));
    d_reg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, (tid >= 32 && tid < 64) ? off : 0x80000000u, 0, 0));
  };
  auto store = [&](int buf) {
    tq.store(sQ[buf]);
    to.store(sO[buf]);
    if (tid < 32) sL[buf][tid] = l_reg;
    else if (tid < 64) sD[buf][tid - 32] = d_reg;
  };
  load(0);
  store(0);
  __syncthreads();
  for (int qt = 0; qt < ntile; ++qt) {
    const int buf = qt & 1;
    load(qt + 1);
    __builtin_amdgcn_sched_barrier(0);
    v16f s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.0f;
#pragma unroll
    for (int st = 0; st < KD; ++st) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(sQ[buf][j * PITCH + 2 * st + half], kf[st], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sO[buf][j * PITCH + 2 * st + half], vf[st], dp, 0, 0, 0);
    }
    // rows of the tile = queries rowmap(r, half): four consecutive per r>>2
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 l4 = *reinterpret_cast<const float4*>(&sL[buf][8 * g + 4 * half]);
      const float4 d4 = *reinterpret_cast<const float4*>(&sD[buf][8 * g + 4 * half]);
      const float le[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pr = __builtin_amdgcn_exp2f(s[4 * g + e] - le[e]);
        s[4 * g + e] = pr;                          // P
        dp[4 * g + e] = pr * (dp[4 * g + e] - de[e]);   // dS (without the scale factor)
      }
    }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int qr = rowmap(st, half);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int dd = t * 32 + j;
        const float ao = dd < d ? sO[buf][qr * PITCH + dd] : 0.0f;
        const float aq = dd < d ? sQ[buf][qr * PITCH + dd] : 0.0f;
        dv[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, s[st], dv[t], 0, 0, 0);
        dk[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq, dp[st], dk[t], 0, 0, 0);
      }
    }
    store(buf ^ 1);
    __syncthreads();
  }
  if (!k_ok) return;
  const long off = (static_cast<long>(b) * p.Tk + key) * p.ldk + h * d;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dd = t * 32 + 8 * g + 4 * half;
      if (dd >= d) continue;
      *reinterpret_cast<float4*>(p.dv + off + dd) = make_float4(dv[t][4 * g], dv[t][4 * g + 1], dv[t][4 * g + 2], dv[t][4 * g + 3]);
      *reinterpret_cast<float4*>(p.dk + off + dd) = make_float4(dk[t][4 * g] * p.scale, dk[t][4 * g + 1] * p.scale, dk[t][4 * g + 2] * p.scale,
                                                                dk[t][4 * g + 3] * p.scale);
    }
}

// dQ: block = 128 queries of one (batch, head); loops over 32-key tiles (K, V double-buffered in LDS).
//   S^T = K Q^T (col = query),  dP^T = V dO^T,  dS^T = P^T o (dP^T - D_q),  dQ^T += K^T dS^T   (scaled at the end)
template <int KD, int NT>
__global__ __launch_bounds__(256) void k_attn_f32_bwd_q(AttnF32P p) {
  constexpr int PITCH = PITCH_OF(NT);
  __shared__ float sK[2][32 * PITCH];
  __shared__ float sV[2][32 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 31, half = lane >> 5;
  const int nqb = (p.Tq + 127) / 128;
  const int bh = blockIdx.x / nqb, qbk = blockIdx.x - bh * nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int d = p.d;
  const int q_row = qbk * 128 + wid * 32 + j;
  const bool q_ok = q_row < p.Tq;
  const float c2 = p.scale * 1.44269504088896340736f;
  float qf[KD], dof[KD];
  {
    const long oq = (static_cast<long>(b) * p.Tq + (q_ok ? q_row : 0)) * p.ldq + h * d;
    const long oo = (static_cast<long>(b) * p.Tq + (q_ok ? q_row : 0)) * p.ldo + h * d;
#pragma unroll
    for (int s = 0; s < KD; ++s) {
      qf[s] = q_ok ? p.q[oq + 2 * s + half] * c2 : 0.0f;
      dof[s] = q_ok ? p.dout[oo + 2 * s + half] : 0.0f;
    }
  }
  const long li = (static_cast<long>(b) * p.heads + h) * p.Tq + (q_ok ? q_row : 0);
  const float lse_q = p.lse[li], d_q = p.dsum[li];
  typedef float v16f __attribute__((ext_vector_type(16)));
  v16f dq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.0f;
  const float* kb = p.k + static_cast<long>(b) * p.Tk * p.ldk + h * d;
  const float* vb = p.v + static_cast<long>(b) * p.Tk * p.ldk + h * d;
  const int ntile = (p.Tk + 31) / 32;
  TileStage<KD, PITCH> tk, tv;
  tk.init(kb, p.ldk, static_cast<long>(p.Tk - 1) * p.ldk + d);
  tv.init(vb, p.ldk, static_cast<long>(p.Tk - 1) * p.ldk + d);
  tk.load(0, p.Tk);
  tv.load(0, p.Tk);
  tk.store(sK[0]);
  tv.store(sV[0]);
  __syncthreads();
  for (int kt = 0; kt < ntile; ++kt) {
    const int buf = kt & 1;
    tk.load((kt + 1) * 32, p.Tk - (kt + 1) * 32);
    tv.load((kt + 1) * 32, p.Tk - (kt + 1) * 32);
    __builtin_amdgcn_sched_barrier(0);
    v16f s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.0f;
#pragma unroll
    for (int st = 0; st < KD; ++st) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[buf][j * PITCH + 2 * st + half], qf[st], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[buf][j * PITCH + 2 * st + half], dof[st], dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = __builtin_amdgcn_exp2f(s[r] - lse_q) * (dp[r] - d_q);
    if ((kt + 1) * 32 > p.Tk) {        // keys past the end: no gradient through them
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * 32 + rowmap(r, half) >= p.Tk) dp[r] = 0.0f;
    }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int kr = rowmap(st, half);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int dd = t * 32 + j;
        const float a = dd < d ? sK[buf][kr * PITCH + dd] : 0.0f;
        dq[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dp[st], dq[t], 0, 0, 0);
      }
    }
    tk.store(sK[buf ^ 1]);
    tv.store(sV[buf ^ 1]);
    __syncthreads();
  }
  if (!q_ok) return;
  float* op = p.dq + (static_cast<long>(b) * p.Tq + q_row) * p.ldq + h * d;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dd = t * 32 + 8 * g + 4 * half;
      if (dd < d)
        *reinterpret_cast<float4*>(op + dd) = make_float4(dq[t][4 * g] * p.scale, dq[t][4 * g + 1] * p.scale, dq[t][4 * g + 2] * p.scale,
                                                          dq[t][4 * g + 3] * p.scale);
    }
}

// ---- bf16x3 operand form of the two backward kernels (see k_attn_bx3_fwd): Q and dO row-wise AND transposed for the key-block kernel,
// K both ways and V row-wise for the query-block kernel
template <int KD, int NS, int NT>
__global__ __launch_bounds__(256) void k_attn_bx3_bwd_kv(AttnF32P p) {
  constexpr int PR = 32 * NS + 16, RWB = 32 * PR, TRB = 32 * NT * 80;
  __shared__ __attribute__((aligned(16))) unsigned char sQr[2][2][RWB], sOr[2][2][RWB];      // [buffer][hi | lo]
  __shared__ __attribute__((aligned(16))) unsigned char sQt[2][2][TRB], sOt[2][2][TRB];
  __shared__ __attribute__((aligned(16))) float sL[2][32];
  __shared__ __attribute__((aligned(16))) float sD[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 31, half = lane >> 5;
  const int nkb = (p.Tk + 127) / 128;
  const int nz = p.qsplit, zq = blockIdx.x % nz, bx = blockIdx.x / nz;
  const int bh = bx / nkb, kbk = bx - bh * nkb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int d = p.d;
  const int key = kbk * 128 + wid * 32 + j;
  const bool k_ok = key < p.Tk;
  const float c2 = p.scale * 1.44269504088896340736f;
  for (int i = tid * 4; i < static_cast<int>(sizeof(sQr)); i += 1024) {
    *reinterpret_cast<unsigned*>(&sQr[0][0][0] + i) = 0u;
    *reinterpret_cast<unsigned*>(&sOr[0][0][0] + i) = 0u;
  }
  for (int i = tid * 4; i < static_cast<int>(sizeof(sQt)); i += 1024) {
    *reinterpret_cast<unsigned*>(&sQt[0][0][0] + i) = 0u;
    *reinterpret_cast<unsigned*>(&sOt[0][0][0] + i) = 0u;
  }
  HL8 ks[NS], vs[NS];
  {
    const long off = (static_cast<long>(b) * p.Tk + (k_ok ? key : 0)) * p.ldk + h * d;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      ks[st] = row_frag(p.k + off, st, half, d, c2, k_ok);
      vs[st] = row_frag(p.v + off, st, half, d, 1.0f, k_ok);
    }
  }
  v16f_ dv[NT], dk[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[t][r] = dk[t][r] = 0.0f;
  const float* qb = p.q + static_cast<long>(b) * p.Tq * p.ldq + h * d;
  const float* ob = p.dout + static_cast<long>(b) * p.Tq * p.ldo + h * d;
  const float* lb = p.lse + (static_cast<long>(b) * p.heads + h) * p.Tq;
  const float* db = p.dsum + (static_cast<long>(b) * p.heads + h) * p.Tq;
  const int ntile_all = p.Tq / 32;
  const int qt0 = static_cast<int>(static_cast<long>(ntile_all) * zq / nz), ntile = static_cast<int>(static_cast<long>(ntile_all) * (zq + 1) / nz);
  TileStage<KD, 1> tq, to;
  tq.init(qb, p.ldq, static_cast<long>(p.Tq - 1) * p.ldq + d);
  to.init(ob, p.ldo, static_cast<long>(p.Tq - 1) * p.ldo + d);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lb), 0, p.Tq * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(db), 0, p.Tq * 4, 0x00020000);
  float l_reg, d_reg;
  auto load = [&](int qt) {
    tq.load(qt * 32, p.Tq - qt * 32);
    to.load(qt * 32, p.Tq - qt * 32);
    const unsigned off = static_cast<unsigned>((qt * 32 + (tid & 31)) * 4);
    l_reg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, tid < 32 ? off : 0x80000000u, 0, 0));
    d_reg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, (tid >= 32 && tid < 64) ? off : 0x80000000u, 0, 0));
  };
  auto store = [&](int buf) {
    tq.store_split(sQr[buf][0], sQr[buf][1], PR, sQt[buf][0], sQt[buf][1]);
    to.store_split(sOr[buf][0], sOr[buf][1], PR, sOt[buf][0], sOt[buf][1]);
    if (tid < 32) sL[buf][tid] = l_reg;
    else if (tid < 64) sD[buf][tid - 32] = d_reg;
  };
  load(qt0);
  __syncthreads();
  store(0);
  __syncthreads();
  for (int qt = qt0; qt < ntile; ++qt) {
    const int buf = (qt - qt0) & 1;
    if (qt + 1 < ntile) load(qt + 1);
    else load(ntile_all);                    // past the extent: zeros
    __builtin_amdgcn_sched_barrier(0);
    v16f_ s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.0f;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      s = mma3(BX3_RW(sQr[buf][0], j, st, half, PR), BX3_RW(sQr[buf][1], j, st, half, PR), ks[st], s);
      dp = mma3(BX3_RW(sOr[buf][0], j, st, half, PR), BX3_RW(sOr[buf][1], j, st, half, PR), vs[st], dp);
    }
    float pv[16], dsv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 l4 = *reinterpret_cast<const float4*>(&sL[buf][8 * g + 4 * half]);
      const float4 d4 = *reinterpret_cast<const float4*>(&sD[buf][8 * g + 4 * half]);
      const float le[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pr = __builtin_amdgcn_exp2f(s[4 * g + e] - le[e]);
        pv[4 * g + e] = pr;
        dsv[4 * g + e] = pr * (dp[4 * g + e] - de[e]);
      }
    }
#pragma unroll
    for (int z = 0; z < 2; ++z) {
      const HL8 ps = split8(pv + 8 * z), dss = split8(dsv + 8 * z);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        dv[t] = mma3(BX3_TR(sOt[buf][0], t * 32 + j, z, half), BX3_TR(sOt[buf][1], t * 32 + j, z, half), ps, dv[t]);
        dk[t] = mma3(BX3_TR(sQt[buf][0], t * 32 + j, z, half), BX3_TR(sQt[buf][1], t * 32 + j, z, half), dss, dk[t]);
      }
    }
    store(buf ^ 1);
    __syncthreads();
  }
  if (!k_ok) return;
  if (nz > 1) {        // partial sums of this query slice: [dV | dK][slice][b][key][heads * d], reduced in slice order by k_attn_qsplit_reduce
    const long cc = static_cast<long>(p.heads) * d, plane = static_cast<long>(p.B) * p.Tk * cc;
    float* pv_ = p.part + zq * plane + (static_cast<long>(b) * p.Tk + key) * cc + h * d;
    float* pk_ = pv_ + nz * plane;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = t * 32 + 8 * g + 4 * half;
        if (dd >= d) continue;
        *reinterpret_cast<float4*>(pv_ + dd) = make_float4(dv[t][4 * g], dv[t][4 * g + 1], dv[t][4 * g + 2], dv[t][4 * g + 3]);
        *reinterpret_cast<float4*>(pk_ + dd) = make_float4(dk[t][4 * g], dk[t][4 * g + 1], dk[t][4 * g + 2], dk[t][4 * g + 3]);
      }
    return;
  }
  const long off = (static_cast<long>(b) * p.Tk + key) * p.ldk + h * d;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dd = t * 32 + 8 * g + 4 * half;
      if (dd >= d) continue;
      *reinterpret_cast<float4*>(p.dv + off + dd) = make_float4(dv[t][4 * g], dv[t][4 * g + 1], dv[t][4 * g + 2], dv[t][4 * g + 3]);
      *reinterpret_cast<float4*>(p.dk + off + dd) = make_float4(dk[t][4 * g] * p.scale, dk[t][4 * g + 1] * p.scale, dk[t][4 * g + 2] * p.scale,
                                                                dk[t][4 * g + 3] * p.scale);
    }
}

// dV, dK = sums of the query slices' partials in slice order (deterministic), dK times `scale`
__global__ void k_attn_qsplit_reduce(AttnF32P p) {
  const long cc = static_cast<long>(p.heads) * p.d, plane = static_cast<long>(p.B) * p.Tk * cc;
  for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < plane; i += static_cast<long>(gridDim.x) * blockDim.x) {
    float a = 0.0f, c = 0.0f;
    for (int z = 0; z < p.qsplit; ++z) {
      a += p.part[z * plane + i];
      c += p.part[(p.qsplit + z) * plane + i];
    }
    const long row = i / cc, col = i - row * cc;
    p.dv[row * p.ldk + col] = a;
    p.dk[row * p.ldk + col] = c * p.scale;
  }
}

template <int KD, int NS, int NT>
__global__ __launch_bounds__(256) void k_attn_bx3_bwd_q(AttnF32P p) {
  constexpr int PR = 32 * NS + 16, RWB = 32 * PR, TRB = 32 * NT * 80;
  __shared__ __attribute__((aligned(16))) unsigned char sKr[2][2][RWB], sVr[2][2][RWB];
  __shared__ __attribute__((aligned(16))) unsigned char sKt[2][2][TRB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 31, half = lane >> 5;
  const int nqb = (p.Tq + 127) / 128;
  const int bh = blockIdx.x / nqb, qbk = blockIdx.x - bh * nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int d = p.d;
  const int q_row = qbk * 128 + wid * 32 + j;
  const bool q_ok = q_row < p.Tq;
  const float c2 = p.scale * 1.44269504088896340736f;
  for (int i = tid * 4; i < static_cast<int>(sizeof(sKr)); i += 1024) {
    *reinterpret_cast<unsigned*>(&sKr[0][0][0] + i) = 0u;
    *reinterpret_cast<unsigned*>(&sVr[0][0][0] + i) = 0u;
  }
  for (int i = tid * 4; i < static_cast<int>(sizeof(sKt)); i += 1024) *reinterpret_cast<unsigned*>(&sKt[0][0][0] + i) = 0u;
  HL8 qs[NS], dos[NS];
  {
    const long oq = (static_cast<long>(b) * p.Tq + (q_ok ? q_row : 0)) * p.ldq + h * d;
    const long oo = (static_cast<long>(b) * p.Tq + (q_ok ? q_row : 0)) * p.ldo + h * d;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      qs[st] = row_frag(p.q + oq, st, half, d, c2, q_ok);
      dos[st] = row_frag(p.dout + oo, st, half, d, 1.0f, q_ok);
    }
  }
  const long li = (static_cast<long>(b) * p.heads + h) * p.Tq + (q_ok ? q_row : 0);
  const float lse_q = p.lse[li], d_q = p.dsum[li];
  v16f_ dq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.0f;
  const float* kb = p.k + static_cast<long>(b) * p.Tk * p.ldk + h * d;
  const float* vb = p.v + static_cast<long>(b) * p.Tk * p.ldk + h * d;
  const int ntile = (p.Tk + 31) / 32;
  TileStage<KD, 1> tk, tv;
  tk.init(kb, p.ldk, static_cast<long>(p.Tk - 1) * p.ldk + d);
  tv.init(vb, p.ldk, static_cast<long>(p.Tk - 1) * p.ldk + d);
  tk.load(0, p.Tk);
  tv.load(0, p.Tk);
  __syncthreads();
  tk.store_split(sKr[0][0], sKr[0][1], PR, sKt[0][0], sKt[0][1]);
  tv.store_split(sVr[0][0], sVr[0][1], PR, nullptr, nullptr);
  __syncthreads();
  for (int kt = 0; kt < ntile; ++kt) {
    const int buf = kt & 1;
    tk.load((kt + 1) * 32, p.Tk - (kt + 1) * 32);
    tv.load((kt + 1) * 32, p.Tk - (kt + 1) * 32);
    __builtin_amdgcn_sched_barrier(0);
    v16f_ s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.0f;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      s = mma3(BX3_RW(sKr[buf][0], j, st, half, PR), BX3_RW(sKr[buf][1], j, st, half, PR), qs[st], s);
      dp = mma3(BX3_RW(sVr[buf][0], j, st, half, PR), BX3_RW(sVr[buf][1], j, st, half, PR), dos[st], dp);
    }
    float dsv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dsv[r] = __builtin_amdgcn_exp2f(s[r] - lse_q) * (dp[r] - d_q);
    if ((kt + 1) * 32 > p.Tk) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * 32 + rowmap(r, half) >= p.Tk) dsv[r] = 0.0f;
    }
#pragma unroll
    for (int z = 0; z < 2; ++z) {
      const HL8 dss = split8(dsv + 8 * z);
#pragma unroll
      for (int t = 0; t < NT; ++t) dq[t] = mma3(BX3_TR(sKt[buf][0], t * 32 + j, z, half), BX3_TR(sKt[buf][1], t * 32 + j, z, half), dss, dq[t]);
    }
    tk.store_split(sKr[buf ^ 1][0], sKr[buf ^ 1][1], PR, sKt[buf ^ 1][0], sKt[buf ^ 1][1]);
    tv.store_split(sVr[buf ^ 1][0], sVr[buf ^ 1][1], PR, nullptr, nullptr);
    __syncthreads();
  }
  if (!q_ok) return;
  float* op = p.dq + (static_cast<long>(b) * p.Tq + q_row) * p.ldq + h * d;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dd = t * 32 + 8 * g + 4 * half;
      if (dd < d)
        *reinterpret_cast<float4*>(op + dd) = make_float4(dq[t][4 * g] * p.scale, dq[t][4 * g + 1] * p.scale, dq[t][4 * g + 2] * p.scale,
                                                          dq[t][4 * g + 3] * p.scale);
    }
}

extern "C" int tfmq_attention_f32_bwd(tfmq_handle h, const float* q, const float* k, const float* v, int ldq, int ldk,
                                      const float* out, const float* dout, int ldo, const float* lse, float* dsum_ws, float* dq,
                                      float* dk, float* dv, int B, int heads, int Tq, int Tk, int d, float scale, void* stream) {
  TFMQ_CHECK_ARG(h, h && q && k && v && out && dout && lse && dsum_ws && dq && dk && dv, "attention_f32_bwd: null pointer");
  TFMQ_CHECK_ARG(h, B > 0 && heads > 0 && Tq > 0 && Tk > 0 && Tq % 32 == 0, "attention_f32_bwd: Tq must be a multiple of 32");
  TFMQ_CHECK_ARG(h, d == 40 || d == 64 || d == 32 || d == 80, "attention_f32_bwd: head dim 32, 40, 64 or 80");
  TFMQ_CHECK_ARG(h, ldq % 4 == 0 && ldk % 4 == 0 && ldo % 4 == 0, "attention_f32_bwd: leading dims must be multiples of 4");
  TFMQ_CHECK_ARG(h, static_cast<long>(Tk) * ldk < (1L << 28) && static_cast<long>(Tq) * ldq < (1L << 28) &&
                        static_cast<long>(Tq) * ldo < (1L << 28),
                 "attention_f32_bwd: one batch item's q / k / v / dO must stay below 1 GiB (32-bit buffer offsets)");
  AttnF32P p{q, k, v, ldq, ldk, nullptr, ldo, const_cast<float*>(lse), dout, dsum_ws, dq, dk, dv, B, heads, Tq, Tk, d, scale};
  hipStream_t st = as_stream(stream);
  const long rows = static_cast<long>(B) * heads * Tq;
  int blocks = ceil_div(rows, 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  hipLaunchKernelGGL(k_attn_f32_rowdot, dim3(blocks), dim3(256), 0, st, out, dout, ldo, dsum_ws, B, heads, Tq, d);
  dim3 gkv(static_cast<unsigned>((Tk + 127) / 128) * B * heads), gq(static_cast<unsigned>((Tq + 127) / 128) * B * heads);
#define TFMQ_BWD3(KD_, NS_, NT_)                                                       \
  do {                                                                                 \
    hipLaunchKernelGGL((k_attn_bx3_bwd_kv<KD_, NS_, NT_>), gkv, dim3(256), 0, st, p);   \
    hipLaunchKernelGGL((k_attn_bx3_bwd_q<KD_, NS_, NT_>), gq, dim3(256), 0, st, p);     \
  } while (0)
  if (h->gemm_prec == 1 && !getenv("TFMQ_ATTN_F32_EXACT")) {
    // short key sequences (cross attention: 77 keys = ONE key block per (batch, head)) leave most CUs idle in the key-block kernel:
    // slice its query loop over several workgroups, partial dK / dV summed in slice order afterwards
    const long kv_blocks = static_cast<long>(gkv.x);
    int nz = 1;
    if (kv_blocks < h->cu_count && Tq / 32 >= 8 && !getenv("TFMQ_ATTN_NO_QSPLIT")) {
      nz = static_cast<int>((2L * h->cu_count + kv_blocks - 1) / kv_blocks);
      if (nz > Tq / 32 / 4) nz = Tq / 32 / 4;
      if (nz > 16) nz = 16;
      if (nz < 1) nz = 1;
    }
    if (nz > 1) {
      const size_t need = static_cast<size_t>(2) * nz * B * Tk * heads * d * sizeof(float);
      if (need > h->gemm_ws_bytes) {
        if (h->gemm_ws) (void)hipFree(h->gemm_ws);
        h->gemm_ws = nullptr;
        h->gemm_ws_bytes = 0;
        if (hipMalloc(reinterpret_cast<void**>(&h->gemm_ws), need) != hipSuccess) {
          h->err = "attention_f32_bwd: workspace allocation failed";
          return TFMQ_ERR_HIP;
        }
        h->gemm_ws_bytes = need;
      }
      p.qsplit = nz;
      p.part = h->gemm_ws;
      gkv.x *= static_cast<unsigned>(nz);
    }
    if (d == 40) TFMQ_BWD3(20, 3, 2);
    else if (d == 32) TFMQ_BWD3(16, 2, 1);
    else if (d == 64) TFMQ_BWD3(32, 4, 2);
    else TFMQ_BWD3(40, 5, 3);
    if (nz > 1) {
      const long plane = static_cast<long>(B) * Tk * heads * d;
      int rb = ceil_div(plane, 256);
      if (rb > h->cu_count * 8) rb = h->cu_count * 8;
      hipLaunchKernelGGL(k_attn_qsplit_reduce, dim3(rb), dim3(256), 0, st, p);
    }
    TFMQ_LAUNCH_CHECK(h);
    return TFMQ_OK;
  }
#undef TFMQ_BWD3
#define TFMQ_BWD(KD_, NT_)                                                             \
  do {                                                                                 \
    hipLaunchKernelGGL((k_attn_f32_bwd_kv<KD_, NT_>), gkv, dim3(256), 0, st, p);        \
    hipLaunchKernelGGL((k_attn_f32_bwd_q<KD_, NT_>), gq, dim3(256), 0, st, p);          \
  } while (0)
  if (d == 40) TFMQ_BWD(20, 2);
  else if (d == 32) TFMQ_BWD(16, 1);
  else if (d == 64) TFMQ_BWD(32, 2);
  else TFMQ_BWD(40, 3);
#undef TFMQ_BWD
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
