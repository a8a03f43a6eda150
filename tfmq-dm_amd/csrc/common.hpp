// Internal helpers shared by the gfx950 kernels of libtfmq_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/tfmq_hip.h"

struct tfmq_ctx {
  int device = 0;
  int cu_count = 256;
  int clock_khz = 2400000;
  size_t hbm_bytes = 0;
  std::string err;
  std::vector<hipGraphExec_t> graphs;
  std::vector<hipEvent_t> events;
  // 256 rows x 64 bytes, row v filled with byte v: source of "real zero" activations (bin za-128) for the
  // padded taps of the LDS-DMA convolution, whose loads cannot substitute a value in registers.
  unsigned char* pad_table = nullptr;
  // grow-only scratch of the split-K reconstruction GEMM (partial sums); used in stream order by one stream at a time
  float* gemm_ws = nullptr;
  size_t gemm_ws_bytes = 0;
  // RCCL communicator of the sharded calibration (comm.hip); opaque here so that no kernel file needs rccl.h
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 0;
  // split-K of the w4a8 tile kernel (tfmq_conv_desc.ksplit): int32 partial slabs [tile][slice][BM * BN] and one arrival ticket per
  // tile (zero between launches: the last arriver resets it).  Allocated with the handle; split-K launches of one handle are
  // stream-ordered (one stream at a time), like gemm_ws.
  int* ksplit_ws = nullptr;
  int* ksplit_cnt = nullptr;
  // the stream that issued the last split-K launch and an event behind that launch (recorded outside stream capture): a split-K launch on
  // ANOTHER stream first waits for it, so that launches of one handle that are ordered on the host are ordered on the device as well.  Two
  // streams replaying captured split-K launches of one handle CONCURRENTLY remain the caller's to order (use one handle per such stream).
  void* ksplit_owner = nullptr;
  bool ksplit_owner_set = false, ksplit_ev_valid = false;
  hipEvent_t ksplit_ev = nullptr;
  static constexpr size_t KSPLIT_WS_INTS = static_cast<size_t>(16) << 20;      // 64 MiB: 1024 slabs of 128 x 128
  static constexpr int KSPLIT_MAX_TILES = 65536;
  // operand precision of tfmq_gemm_f32's matrix-core path (tfmq_set_gemm_precision): 0 exact fp32, 1 bf16x3 split, 2 fp16
  int gemm_prec = 0;
};

// strided fp32 GEMM of the reconstruction units (recon_kernels.hip, gemm_f32_mfma.hip)
struct GemmP {
  const float* A; const float* B; float* C;
  int M, N, K;
  long sam, sak, sbk, sbn, scm;      // element strides: A(m,k)=A[m*sam+k*sak], B(k,n)=B[k*sbk+n*sbn], C(m,n)=C[m*scm+n]
  long bsa, bsb, bsc;                // batch strides
  float alpha;
  const float* bias;                 // [N] or null
  const float* rowadd;               // rowadd[(m / rows_per_img) * rowadd_ld + n] or null
  int rows_per_img, rowadd_ld;
  const float* residual;             // same layout as C, or null
  int accumulate;                    // C += result
  int ksplit, kchunk;                // split-K: blockIdx.z = batch * ksplit + split, split covers K range [split*kchunk, +kchunk)
  float* partial;                    // [ksplit][batch][M][N] raw partial sums (ksplit > 1)
  int tiles_n;                       // column tiles (the MFMA kernel's 1-D tile grid)
  // two-level batch (multi-head attention on the packed [B,T,heads*d] layout): item z = (z / nb2, z % nb2), offsets
  // (z / nb2) * bs? + (z % nb2) * bs?2.  nb2 = 1 for a plain batch.
  int nb2;
  long bsa2, bsb2, bsc2;
  __host__ __device__ long off_a(int z) const { return (z / nb2) * bsa + (z % nb2) * bsa2; }
  __host__ __device__ long off_b(int z) const { return (z / nb2) * bsb + (z % nb2) * bsb2; }
  __host__ __device__ long off_c(int z) const { return (z / nb2) * bsc + (z % nb2) * bsc2; }
};

#define TFMQ_CHECK_ARG(h, cond, msg)          \
  do {                                        \
    if (!(cond)) {                            \
      if (h) (h)->err = std::string("bad argument: ") + (msg); \
      return TFMQ_ERR_ARG;                    \
    }                                         \
  } while (0)

#define TFMQ_HIP(h, expr)                                                            \
  do {                                                                               \
    hipError_t e__ = (expr);                                                         \
    if (e__ != hipSuccess) {                                                         \
      if (h) (h)->err = std::string(#expr) + ": " + hipGetErrorString(e__);          \
      return TFMQ_ERR_HIP;                                                           \
    }                                                                                \
  } while (0)

#define TFMQ_LAUNCH_CHECK(h)                                                         \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      if (h) (h)->err = std::string("kernel launch: ") + hipGetErrorString(e__);     \
      return TFMQ_ERR_HIP;                                                           \
    }                                                                                \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------- device helpers
// A device scalar (the Finite-Set step counter, written by an earlier launch) through the SCALAR cache.  As a plain `*p` of a
// pointer out of a by-value struct the compiler emits global_load_dword + s_waitcnt vmcnt(0) + v_readfirstlane -- and vmcnt(0)
// also waits for every LDS-DMA piece / prefetch load the kernel issued before it (the compiler does not see the asm-issued ones):
// round 6 found k_lin_direct's blocks spending 2.6 ... 3 us there, in front of their first K-step (profiles/r06_phase_lin_direct.txt).
// s_load counts on lgkmcnt only; the scalar cache is invalidated at every kernel boundary, so a value an earlier launch wrote is seen.
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {      // the (wave-uniform) pointer in an SGPR pair, wherever the compiler keeps it
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a >> 32));
  return reinterpret_cast<const T*>((static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ int load_scalar_i32(const int32_t* p) {
  int v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(uniform_ptr(p)) : "memory");
  return v;
}
__device__ __forceinline__ float2 load_qparam(const tfmq_qsel& qs) {
  // {delta, zero_point} of the current FSC group
  const int k = qs.step ? load_scalar_i32(qs.step) : 0;
  const float* p = qs.qtable + (static_cast<size_t>(k) * qs.q_stride + qs.qid) * 2;
  unsigned long long v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(uniform_ptr(p)) : "memory");
  return make_float2(__uint_as_float(static_cast<unsigned>(v)), __uint_as_float(static_cast<unsigned>(v >> 32)));
}

// q = clamp(rint(x/delta)+zp, 0, L-1): true IEEE division, round-half-even
// (quant/quant_layer.py:225).  Compiled without fast-math.
// The quotient is the correctly rounded x / delta, computed without the division sequence: delta is one value per
// kernel, so its reciprocal is loop invariant (the compiler hoists the one real division), and two residual
// corrections q <- q + (x - q delta) r give RN(x / delta) [Markstein: r = RN(1/delta), q faithful after the first
// correction, exact residual by FMA; the one excluded divisor -- a significand of all ones -- takes the division].
// 5 FMA-class instructions per element instead of 12; tests/test_quant_division_gpu.py compares the two bit for bit.
__device__ __forceinline__ float div_rn_f(float x, float d) {
  if (__builtin_expect((__float_as_uint(d) & 0x7fffffu) == 0x7fffffu, 0)) return x / d;
  const float r = 1.0f / d;
  float q = x * r;
  q = __builtin_fmaf(__builtin_fmaf(-q, d, x), r, q);
  q = __builtin_fmaf(__builtin_fmaf(-q, d, x), r, q);
  return q;
}
__device__ __forceinline__ float quant_index_f(float x, float delta, float zp, float lmax) {
  float q = rintf(div_rn_f(x, delta)) + zp;
  return fminf(fmaxf(q, 0.0f), lmax);
}

// ---- the 8-bit activation quantizer on packed fp32 (v_pk_mul / v_pk_fma / v_pk_add: two lanes of arithmetic per
// instruction) with the byte packing done by v_cvt_pk_u8_f32.  Same arithmetic, operation for operation, as
// quant_index_f (every packed operation is the IEEE operation on each half), so the bins are bit-identical; what changes
// is the instruction count: 4 bins = 10 packed + 4 rndne + 4 cvt_pk (saturating) + 1 xor instead of ~60 scalar-lane ones.
// Epilogues that apply a transcendental and the quantizer to every output (GEGLU) are VALU-bound, not MFMA-bound.
typedef float f2 __attribute__((ext_vector_type(2)));
struct QuantP {
  float d, r, zp;   // delta, RN(1 / delta) (one true division per kernel), zero point
  bool bad;         // delta's significand is all ones: the one divisor the reciprocal-refinement does not cover
};
__device__ __forceinline__ QuantP make_quantp(float2 qp) {
  QuantP q;
  q.d = qp.x;
  q.zp = qp.y;
  q.r = 1.0f / qp.x;
  q.bad = (__float_as_uint(qp.x) & 0x7fffffu) == 0x7fffffu;
  return q;
}
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// RN(x / delta) for both halves (div_rn_f's sequence)
template <bool EXACT_DIV>
__device__ __forceinline__ f2 quant_quot2(f2 x, const QuantP& p) {
  if constexpr (EXACT_DIV) return f2{x.x / p.d, x.y / p.d};
  const f2 r = {p.r, p.r}, d = {p.d, p.d};
  f2 q = x * r;
  q = pk_fma(pk_fma(-q, d, x), r, q);
  q = pk_fma(pk_fma(-q, d, x), r, q);
  return q;
}
// four values -> their bins - 128 as the four bytes of a word (bins clamp to [0, 255])
template <bool EXACT_DIV>
__device__ __forceinline__ unsigned quant_pack4_t(f2 lo, f2 hi, const QuantP& p) {
  const f2 z = {p.zp, p.zp};
  f2 a = quant_quot2<EXACT_DIV>(lo, p), b = quant_quot2<EXACT_DIV>(hi, p);
  a = f2{__builtin_rintf(a.x), __builtin_rintf(a.y)} + z;
  b = f2{__builtin_rintf(b.x), __builtin_rintf(b.y)} + z;
  // the clamp to [0, 255] is the conversion's own saturation (tfmq_hw_selftest pins it on the device)
  unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(a.x, 0, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(a.y, 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(b.x, 2, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(b.y, 3, w);
  return w ^ 0x80808080u;
}
__device__ __forceinline__ unsigned quant_pack4(float a, float b, float c, float e, const QuantP& p) {
  if (__builtin_expect(p.bad, 0)) return quant_pack4_t<true>(f2{a, b}, f2{c, e}, p);
  return quant_pack4_t<false>(f2{a, b}, f2{c, e}, p);
}
__device__ __forceinline__ char4 quant_char4(float a, float b, float c, float e, const QuantP& p) {
  const unsigned w = quant_pack4(a, b, c, e, p);
  return *reinterpret_cast<const char4*>(&w);
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_reduce_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// x * sigmoid(x) on the hardware exp2 / reciprocal (each within 1 ulp): the library expf and the IEEE division made
// the GroupNorm-apply pass instruction-bound (72 VALU instructions per element, 2.7 TB/s); the result differs from the
// exactly rounded form by a few 1e-7 relative -- 5e-5 of an 8-bit activation bin at most.
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}

// erf to |error| <= 1.5e-7 absolute (Abramowitz & Stegun 7.1.26) plus a few fp32 roundings: 5 fma, one v_rcp, one
// v_exp.  The library erff (two divergent branches, ~140 issued instructions per call) made the GEGLU arithmetic --
// 84 M evaluations for one SD feed-forward -- cost more than the GEMM that feeds it.  GEGLU's product
// a * 0.5 g (1 + erf(g/sqrt 2)) is reproduced to ~1e-7 of its range: far below one 8-bit activation bin.
__device__ __forceinline__ float erf_fast_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  float pl = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  pl = __builtin_fmaf(pl, t, 1.421413741f);
  pl = __builtin_fmaf(pl, t, -0.284496736f);
  pl = __builtin_fmaf(pl, t, 0.254829592f);
  pl *= t;
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * ax * ax);
  return copysignf(__builtin_fmaf(-pl, e, 1.0f), x);
}
// gelu(g) = g Phi(g), Phi(g) = 0.5 (1 + erf(g / sqrt 2))   (F.gelu default, "none" approximation), the same 7.1.26
// polynomial arranged for the fewest instructions: h = Phi(-|g|) = 0.5 poly(t) exp(-g^2 / 2) with the halves folded into
// the coefficients and 1/sqrt 2 into the constants, and g Phi(g) = relu(g) - |g| h.  |error| <= 5e-7 absolute over
// [-12, 12] (the arrangement above it: 4.7e-7).  gelu2 is the same operation sequence on packed fp32 -- bit-identical
// halves -- for the GEMM epilogues.
__device__ __forceinline__ float gelu_f(float g) {
  const float ag = fabsf(g);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.23164189f, ag, 1.0f));
  float pl = __builtin_fmaf(0.5307027145f, t, -0.7265760135f);
  pl = __builtin_fmaf(pl, t, 0.7107068705f);
  pl = __builtin_fmaf(pl, t, -0.142248368f);
  pl = __builtin_fmaf(pl, t, 0.127414796f);
  pl *= t;
  const float e = __builtin_amdgcn_exp2f((g * g) * -0.72134752044448170368f);
  return fmaxf(g, 0.0f) - ag * (pl * e);          // g Phi(g) = relu(g) - |g| Phi(-|g|)
}
__device__ __forceinline__ f2 gelu2(f2 g) {
  const f2 one = {1.0f, 1.0f};
  const f2 ag = {fabsf(g.x), fabsf(g.y)};
  const f2 dn = pk_fma(f2{0.23164189f, 0.23164189f}, ag, one);
  const f2 t = {__builtin_amdgcn_rcpf(dn.x), __builtin_amdgcn_rcpf(dn.y)};
  f2 pl = pk_fma(f2{0.5307027145f, 0.5307027145f}, t, f2{-0.7265760135f, -0.7265760135f});
  pl = pk_fma(pl, t, f2{0.7107068705f, 0.7107068705f});
  pl = pk_fma(pl, t, f2{-0.142248368f, -0.142248368f});
  pl = pk_fma(pl, t, f2{0.127414796f, 0.127414796f});
  pl = pl * t;
  const f2 m = (g * g) * f2{-0.72134752044448170368f, -0.72134752044448170368f};
  const f2 e = {__builtin_amdgcn_exp2f(m.x), __builtin_amdgcn_exp2f(m.y)};
  return f2{fmaxf(g.x, 0.0f), fmaxf(g.y, 0.0f)} - ag * (pl * e);
}

// ---- GELU sized for its consumer (round 4).  In the fused GEGLU epilogues value * gelu(gate) is rounded to one of 256 activation bins by
// the next instruction, and the epilogue is VALU-bound (the 7.1.26 form above: ~11.5 issue slots per value counting v_rcp / v_exp twice).
// Phi(g) ~= 1 / (1 + exp2(g (c1 + c3 g^2 + c5 g^4))) with the minimax coefficients below (fitted against 0.5 (1 + erf(g / sqrt 2)) in
// float64, -log2(e) folded in): |g Phi(g) - gelu(g)| <= 2.8e-5 absolute over the whole line -- 8 issue slots, no abs / max / select.  The
// argument of the polynomial is clamped to [-9, 9] (c5 < 0: the quartic turns over at |g| ~ 11; beyond 9 the logistic is 0 or 1 to 1e-11),
// the final product uses the unclamped g.  Used by TFMQ_OUT_GEGLU_Q8_FAST only; the calibration path (k_geglu) and TFMQ_OUT_GEGLU_Q8 keep
// the 5e-7 form.  tests/test_geglu_fast_gpu.py: bins within 1 of the exact epilogue's, < 2e-3 of them moved.
__device__ __forceinline__ f2 gelu_fast2(f2 g) {
  const f2 gc = {__builtin_amdgcn_fmed3f(g.x, -9.0f, 9.0f), __builtin_amdgcn_fmed3f(g.y, -9.0f, 9.0f)};
  const f2 x2 = gc * gc;
  f2 pl = pk_fma(f2{1.02381220e-3f, 1.02381220e-3f}, x2, f2{-1.06834617e-1f, -1.06834617e-1f});
  pl = pk_fma(pl, x2, f2{-2.30105644f, -2.30105644f});
  const f2 u = pl * gc;
  const f2 dn = f2{__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y)} + f2{1.0f, 1.0f};
  return g * f2{__builtin_amdgcn_rcpf(dn.x), __builtin_amdgcn_rcpf(dn.y)};
}
__device__ __forceinline__ float gelu_fast_f(float g) {
  const f2 r = gelu_fast2(f2{g, g});
  return r.x;
}
// four (value / delta) and gelu(gate) pairs -> bins - 128: q = value' * G + zp in one packed FMA, round-half-even, saturating byte pack
__device__ __forceinline__ unsigned geglu_fast_pack4(f2 a0, f2 G0, f2 a1, f2 G1, float zp) {
  const f2 z = {zp, zp};
  const f2 q0 = pk_fma(a0, G0, z), q1 = pk_fma(a1, G1, z);
  // v_cvt_pk_u8_f32 rounds to nearest-even itself (scratch/ubench/cvt_pk_round.hip: equal to rint + convert on every half-integer;
  // tfmq_hw_selftest bit 4 pins it on the device): no v_rndne_f32 in front of it
  unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(q0.x, 0, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(q0.y, 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(q1.x, 2, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(q1.y, 3, w);
  return w ^ 0x80808080u;
}

static inline int ceil_div(long a, long b) { return static_cast<int>((a + b - 1) / b); }

// ---------------------------------------------------------------- packed int4 weight layout
// "Tile-major": output channels are grouped in blocks of 32 rows; inside a block the bytes of one
// K-step (ck input channels of one tap = ck/8 32-bit words per row) of all 32 rows are contiguous,
// so the implicit-GEMM loader reads whole 128-byte lines and uses every byte of them
// (row-major [cout][K/2] made every K-step touch a quarter of 128 different lines).
//   word(n, g) = ((n/32 * nsteps + g/wps) * 32 + n%32) * wps + g%wps,   wps = ck/8, nsteps = K/ck
// ck = 64 if cin % 64 == 0, else 32 if cin % 32 == 0, else 8 (plain row-major; GEMV / tests only).
__host__ __device__ inline int w4_ck(int cin) { return (cin % 64 == 0) ? 64 : ((cin % 32 == 0) ? 32 : 8); }
__host__ __device__ inline size_t w4_word_index(int n, int g, int K, int ck) {
  const int wps = ck / 8, nsteps = K / ck;
  return ((static_cast<size_t>(n / 32) * nsteps + g / wps) * 32 + (n % 32)) * wps + (g % wps);
}
