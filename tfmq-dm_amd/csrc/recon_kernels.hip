// K15: building blocks of the block-reconstruction forward/backward (quant/reconstruction.py:
// `out_quant = block(*cur_inputs); err.backward()`), hand-written instead of autograd.
// The AdaRound soft targets make the weights non-integer, so this path is floating point by
// construction (SURVEY §7 hard part 1); it runs in exact fp32 (FMA GEMM) so the loss curve can be
// compared with the reference's fp32 training.  Batch 32 units: the convolution is lowered to
// im2col + GEMM (the 9x activation blow-up is irrelevant next to 288 GB of HBM), which gives
// forward, weight gradient and input gradient from ONE strided-GEMM kernel:
//   y    = col(x) W^T (+bias +temb row +residual)       gemm(A=col,  B=W^T)
//   dW   = dY^T col(x)                                   gemm(A=dY^T, B=col)
//   dcol = dY W ; dx = col2im(dcol)                      gemm(A=dY,   B=W) + gather
#include "common.hpp"

// ------------------------------------------------------------------ strided batched GEMM (fp32)

int tfmq_gemm_f32_mfma_launch(tfmq_handle h, GemmP& p, int batch, hipStream_t st);   // gemm_f32_mfma.hip

#define GT 64
#define GK 16
__global__ __launch_bounds__(256) void k_gemm_f32(GemmP p) {
  __shared__ float As[GK][GT + 4];
  __shared__ float Bs[GK][GT + 4];
  const int bz = blockIdx.z;
  const float* A = p.A + p.off_a(bz);
  const float* B = p.B + p.off_b(bz);
  float* C = p.C + p.off_c(bz);
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;  // 16 x 16 threads, 4x4 outputs each
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
  const bool a_kfast = p.sak == 1, b_nfast = p.sbn == 1;
  for (int k0 = 0; k0 < p.K; k0 += GK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      int m, k;
      if (a_kfast) { k = e & 15; m = e >> 4; } else { m = e & 63; k = e >> 6; }
      const int gm = m0 + m, gk = k0 + k;
      As[k][m] = (gm < p.M && gk < p.K) ? A[gm * p.sam + gk * p.sak] : 0.0f;
      int n, kb;
      if (b_nfast) { n = e & 63; kb = e >> 6; } else { kb = e & 15; n = e >> 4; }
      const int gn = n0 + n, gkb = k0 + kb;
      Bs[kb][n] = (gn < p.N && gkb < p.K) ? B[gkb * p.sbk + gn * p.sbn] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= p.N) continue;
      float v = p.alpha * acc[i][j];
      if (p.bias) v += p.bias[n];
      if (p.rowadd) v += p.rowadd[static_cast<long>(m / p.rows_per_img) * p.rowadd_ld + n];
      if (p.residual) v += p.residual[p.off_c(bz) + m * p.scm + n];
      float* c = C + m * p.scm + n;
      *c = p.accumulate ? *c + v : v;
    }
  }
}

static int gemm_f32_impl(tfmq_handle h, const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak,
                         long sbk, long sbn, long scm, int batch, long bsa, long bsb, long bsc, int heads, long hsa, long hsb,
                         long hsc, float alpha, const float* bias, const float* rowadd, int rows_per_img, int rowadd_ld,
                         const float* residual, int accumulate, void* stream) {
  TFMQ_CHECK_ARG(h, h && A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && heads > 0 &&
                        static_cast<long>(batch) * heads < 65536, "gemm_f32: bad argument");
  TFMQ_CHECK_ARG(h, !rowadd || rows_per_img > 0, "gemm_f32: rowadd needs rows_per_img");
  GemmP p{A, B, C, M, N, K, sam, sak, sbk, sbn, scm, bsa, bsb, bsc, alpha, bias, rowadd, rows_per_img, rowadd_ld, residual,
          accumulate, 1, 0, nullptr, 0, heads, hsa, hsb, hsc};
  batch *= heads;
  // fp32 matrix cores (gemm_f32_mfma.hip); small problems keep the FMA tile.  Skinny outputs with a long reduction
  // (the context-side gradients of cross attention: 77 x 40..160, K = 256..4096) also go there: split-K fills the chip
  // ... and the mini-batch-row GEMMs of the TIB unit (8 x 1280 x 1280: 96 us on the FMA tile, 20 blocks walking K; 24 us here)
  if ((M >= 96 && N >= 24 && K >= 8) || (M >= 32 && N >= 24 && K >= 64) || (N >= 64 && K >= 256)) {
    const int rc = tfmq_gemm_f32_mfma_launch(h, p, batch, as_stream(stream));
    if (rc != TFMQ_OK) return rc;
    TFMQ_LAUNCH_CHECK(h);
    return TFMQ_OK;
  }
  dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, batch);
  hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, as_stream(stream), p);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_set_gemm_precision(tfmq_handle h, int mode) {
  TFMQ_CHECK_ARG(h, h && mode >= 0 && mode <= 2, "set_gemm_precision: mode 0 (exact fp32), 1 (bf16x3) or 2 (fp16)");
  h->gemm_prec = mode;
  return TFMQ_OK;
}

extern "C" int tfmq_gemm_f32(tfmq_handle h, const float* A, const float* B, float* C, int M, int N, int K, long sam,
                             long sak, long sbk, long sbn, long scm, int batch, long bsa, long bsb, long bsc, float alpha,
                             const float* bias, const float* rowadd, int rows_per_img, int rowadd_ld,
                             const float* residual, int accumulate, void* stream) {
  return gemm_f32_impl(h, A, B, C, M, N, K, sam, sak, sbk, sbn, scm, batch, bsa, bsb, bsc, 1, 0, 0, 0, alpha, bias, rowadd,
                       rows_per_img, rowadd_ld, residual, accumulate, stream);
}

// Two-level batch: item (b, hd) at offsets b * bs? + hd * hs?.  One launch for all heads of a multi-head attention
// product on the packed [B, T, heads * d] layout (per-head launches of 64 blocks left three quarters of the chip idle).
extern "C" int tfmq_gemm_f32_heads(tfmq_handle h, const float* A, const float* B, float* C, int M, int N, int K, long sam,
                                   long sak, long sbk, long sbn, long scm, int batch, long bsa, long bsb, long bsc, int heads,
                                   long hsa, long hsb, long hsc, float alpha, int accumulate, void* stream) {
  return gemm_f32_impl(h, A, B, C, M, N, K, sam, sak, sbk, sbn, scm, batch, bsa, bsb, bsc, heads, hsa, hsb, hsc, alpha, nullptr,
                       nullptr, 1, 0, nullptr, accumulate, stream);
}

// ------------------------------------------------------------------ im2col / col2im (NHWC, zero pad)
// col[(b,ho,wo)][(kh,kw,c)] = x[b][ho*s+kh-pt][wo*s+kw-pl][c]   (0 outside)
__global__ __launch_bounds__(256) void k_im2col(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W,
                                                int Cc, int KH, int KW, int stride, int pt, int pl, int Ho, int Wo) {
  const long total = static_cast<long>(B) * Ho * Wo * KH * KW * Cc;
  for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long>(gridDim.x) * blockDim.x) {
    const int c = i % Cc;
    long r = i / Cc;
    const int kw = r % KW; r /= KW;
    const int kh = r % KH; r /= KH;
    const int wo = r % Wo; r /= Wo;
    const int ho = r % Ho;
    const int b = r / Ho;
    const int hi = ho * stride + kh - pt, wi = wo * stride + kw - pl;
    col[i] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? x[((static_cast<long>(b) * H + hi) * W + wi) * Cc + c] : 0.0f;
  }
}

// The same copy with 32-bit index arithmetic and V channels per item (16-byte loads / stores when C % 4 == 0): the
// element-wise form above spends five 64-bit divisions per float (2 TB/s: instruction-bound); this one five 32-bit ones
// per V floats.  Used whenever the item count fits 32 bits.
template <int V>
__global__ __launch_bounds__(256) void k_im2col_v(const float* __restrict__ x, float* __restrict__ col, unsigned total, int H, int W,
                                                  int Cc, int KH, int KW, int stride, int pt, int pl, int Ho, int Wo) {
  const unsigned cv = Cc / V, taps = KH * KW, gs = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    const unsigned t1 = i / cv, cq = i - t1 * cv;
    const unsigned pix = t1 / taps, tap = t1 - pix * taps;
    const unsigned kh = tap / KW, kw = tap - kh * KW;
    const unsigned t2 = pix / Wo, wo = pix - t2 * Wo;
    const unsigned b = t2 / Ho, ho = t2 - b * Ho;
    const int hi = static_cast<int>(ho) * stride + static_cast<int>(kh) - pt, wi = static_cast<int>(wo) * stride + static_cast<int>(kw) - pl;
    const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
    const float* src = x + ((static_cast<size_t>(b) * H + (ok ? hi : 0)) * W + (ok ? wi : 0)) * Cc + cq * V;
    float* dst = col + static_cast<size_t>(i) * V;
    if constexpr (V == 4) {
      const float4 v = *reinterpret_cast<const float4*>(src);
      *reinterpret_cast<float4*>(dst) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float v = *src;
      *dst = ok ? v : 0.0f;
    }
    if (i + gs < i) break;      // 32-bit wrap of the item index
  }
}

extern "C" int tfmq_im2col(tfmq_handle h, const float* x, float* col, int B, int H, int W, int C, int KH, int KW,
                           int stride, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && col && B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && stride > 0 && Ho > 0 && Wo > 0,
                 "im2col: bad argument");
  const long total = static_cast<long>(B) * Ho * Wo * KH * KW * C;
  const bool v4 = C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(col) & 15) == 0;
  const long items = v4 ? total / 4 : total;
  int blocks = ceil_div(items, 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  if (items < (1L << 32)) {
    if (v4) hipLaunchKernelGGL(k_im2col_v<4>, dim3(blocks), dim3(256), 0, as_stream(stream), x, col, static_cast<unsigned>(items), H,
                               W, C, KH, KW, stride, pad_t, pad_l, Ho, Wo);
    else hipLaunchKernelGGL(k_im2col_v<1>, dim3(blocks), dim3(256), 0, as_stream(stream), x, col, static_cast<unsigned>(items), H, W,
                            C, KH, KW, stride, pad_t, pad_l, Ho, Wo);
  } else
    hipLaunchKernelGGL(k_im2col, dim3(blocks), dim3(256), 0, as_stream(stream), x, col, B, H, W, C, KH, KW, stride, pad_t,
                       pad_l, Ho, Wo);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// dx[b][hi][wi][c] = sum over taps of dcol at the output pixel that read (hi,wi) through that tap (gather: deterministic)
__global__ __launch_bounds__(256) void k_col2im(const float* __restrict__ dcol, float* __restrict__ dx, int B, int H, int W,
                                                int Cc, int KH, int KW, int stride, int pt, int pl, int Ho, int Wo) {
  const long total = static_cast<long>(B) * H * W * Cc;
  for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long>(gridDim.x) * blockDim.x) {
    const int c = i % Cc;
    long r = i / Cc;
    const int wi = r % W; r /= W;
    const int hi = r % H;
    const int b = r / H;
    float s = 0.0f;
    for (int kh = 0; kh < KH; ++kh) {
      const int hn = hi + pt - kh;
      if (hn < 0 || hn % stride) continue;
      const int ho = hn / stride;
      if (ho >= Ho) continue;
      for (int kw = 0; kw < KW; ++kw) {
        const int wn = wi + pl - kw;
        if (wn < 0 || wn % stride) continue;
        const int wo = wn / stride;
        if (wo >= Wo) continue;
        s += dcol[(((static_cast<long>(b) * Ho + ho) * Wo + wo) * KH * KW + kh * KW + kw) * Cc + c];
      }
    }
    dx[i] = s;
  }
}

// fp16 im2col rows of a narrow-input conv (see tfmq_im2col_f16): one 16-byte piece (8 k-values) per thread
__global__ __launch_bounds__(256) void k_im2col_h(const float* __restrict__ x, __half* __restrict__ col, unsigned total, int H, int W,
                                                  int Cc, int KW, int KK, int pad_t, int pad_l, int pieces) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned m = i / pieces, pc = i - m * pieces;
    const int xx = m % W, yy = (m / W) % H;
    const unsigned b = m / (static_cast<unsigned>(W) * H);
    __half v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = pc * 8 + e;
      const int tap = k / Cc, ci = k - tap * Cc;
      const int sy = yy + tap / KW - pad_t, sx = xx + tap % KW - pad_l;
      float f = 0.0f;
      if (tap < KK && sy >= 0 && sy < H && sx >= 0 && sx < W) f = x[((static_cast<size_t>(b) * H + sy) * W + sx) * Cc + ci];
      v[e] = __float2half_rn(f);
    }
    *reinterpret_cast<uint4*>(col + static_cast<size_t>(i) * 8) = *reinterpret_cast<const uint4*>(v);
  }
}

extern "C" int tfmq_im2col_f16(tfmq_handle h, const float* x, uint16_t* col, int B, int H, int W, int C, int KH, int KW, int pad_t,
                               int pad_l, int kp, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && col && B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0, "im2col_f16: bad argument");
  TFMQ_CHECK_ARG(h, kp % 8 == 0 && kp >= KH * KW * C, "im2col_f16: kp must be a multiple of 8 and hold kh*kw*C values");
  const long total = static_cast<long>(B) * H * W * (kp / 8);
  TFMQ_CHECK_ARG(h, total < (1L << 32), "im2col_f16: more than 2^32 pieces");
  int blocks = ceil_div(total, 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  hipLaunchKernelGGL(k_im2col_h, dim3(blocks), dim3(256), 0, as_stream(stream), x, reinterpret_cast<__half*>(col),
                     static_cast<unsigned>(total), H, W, C, KW, KH * KW, pad_t, pad_l, kp / 8);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// per-tap partial sums -> the few output channels of a narrow-output conv (see tfmq_tap_gather_sum); one pixel per thread
template <int CO>
__global__ __launch_bounds__(256) void k_tap_gather(const float* __restrict__ y9, unsigned total, int H, int W, int KH, int KW, int ld,
                                                    int pad_t, int pad_l, const float* __restrict__ bias, float* __restrict__ out) {
  for (unsigned m = blockIdx.x * blockDim.x + threadIdx.x; m < total; m += gridDim.x * blockDim.x) {
    const int xx = m % W, yy = (m / W) % H;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = bias ? bias[c] : 0.0f;
    for (int tap = 0; tap < KH * KW; ++tap) {
      const int dy = tap / KW - pad_t, dx = tap % KW - pad_l;
      const int sy = yy + dy, sx = xx + dx;
      if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
      const float* src = y9 + (static_cast<size_t>(m) + static_cast<long>(dy) * W + dx) * ld + tap * CO;
      if constexpr (CO == 4) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      } else {
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] += src[c];
      }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) out[static_cast<size_t>(m) * CO + c] = acc[c];
  }
}

extern "C" int tfmq_tap_gather_sum(tfmq_handle h, const float* y9, int B, int H, int W, int KH, int KW, int cout, int ld, int pad_t,
                                   int pad_l, const float* bias, float* out, void* stream) {
  TFMQ_CHECK_ARG(h, h && y9 && out && B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0, "tap_gather_sum: bad argument");
  TFMQ_CHECK_ARG(h, cout >= 1 && cout <= 4 && ld >= KH * KW * cout && (cout != 4 || ld % 4 == 0), "tap_gather_sum: 1..4 output channels, ld >= kh*kw*cout");
  const long total = static_cast<long>(B) * H * W;
  TFMQ_CHECK_ARG(h, total < (1L << 31), "tap_gather_sum: too many pixels");
  int blocks = ceil_div(total, 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  const unsigned t = static_cast<unsigned>(total);
  if (cout == 4) hipLaunchKernelGGL(k_tap_gather<4>, dim3(blocks), dim3(256), 0, as_stream(stream), y9, t, H, W, KH, KW, ld, pad_t, pad_l, bias, out);
  else if (cout == 3) hipLaunchKernelGGL(k_tap_gather<3>, dim3(blocks), dim3(256), 0, as_stream(stream), y9, t, H, W, KH, KW, ld, pad_t, pad_l, bias, out);
  else if (cout == 2) hipLaunchKernelGGL(k_tap_gather<2>, dim3(blocks), dim3(256), 0, as_stream(stream), y9, t, H, W, KH, KW, ld, pad_t, pad_l, bias, out);
  else hipLaunchKernelGGL(k_tap_gather<1>, dim3(blocks), dim3(256), 0, as_stream(stream), y9, t, H, W, KH, KW, ld, pad_t, pad_l, bias, out);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_col2im(tfmq_handle h, const float* dcol, float* dx, int B, int H, int W, int C, int KH, int KW,
                           int stride, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  TFMQ_CHECK_ARG(h, h && dcol && dx && B > 0 && H > 0 && W > 0 && C > 0, "col2im: bad argument");
  const long total = static_cast<long>(B) * H * W * C;
  int blocks = ceil_div(total, 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  hipLaunchKernelGGL(k_col2im, dim3(blocks), dim3(256), 0, as_stream(stream), dcol, dx, B, H, W, C, KH, KW, stride, pad_t,
                     pad_l, Ho, Wo);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// OIHW [co][ci][kh][kw] <-> GEMM layout [co][(kh,kw,ci)]  (dir=0: to GEMM layout, dir=1: back)
// One thread per (co, ci): its khw taps are contiguous in OIHW (adjacent threads = adjacent 4*khw-byte runs: coalesced over the
// wave) and khw strided, ci-contiguous elements of the GEMM layout (coalesced per tap).  32-bit indices (the launcher checks).
// The element-per-thread form read / wrote OIHW with a stride of khw floats and paid three 64-bit divisions per element.
__global__ __launch_bounds__(256) void k_w_relayout(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int khw, int dir) {
  const unsigned pairs = static_cast<unsigned>(cout) * cin;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += gridDim.x * blockDim.x) {
    const unsigned co = i / cin, ci = i - co * cin;
    const unsigned o0 = i * khw;                                  // OIHW offset of tap 0
    const unsigned g0 = co * (static_cast<unsigned>(cin) * khw) + ci;    // GEMM-layout offset of tap 0
    if (dir == 0) {
      for (int t = 0; t < khw; ++t) dst[g0 + t * cin] = src[o0 + t];
    } else {
      for (int t = 0; t < khw; ++t) dst[o0 + t] = src[g0 + t * cin];
    }
  }
}

extern "C" int tfmq_w_relayout(tfmq_handle h, const float* src, float* dst, int cout, int cin, int kh, int kw, int dir,
                               void* stream) {
  TFMQ_CHECK_ARG(h, h && src && dst && cout > 0 && cin > 0 && kh > 0 && kw > 0, "w_relayout: bad argument");
  const long total = static_cast<long>(cout) * cin * kh * kw;
  TFMQ_CHECK_ARG(h, total < (1L << 31), "w_relayout: more than 2^31 elements");
  int blocks = ceil_div(static_cast<long>(cout) * cin, 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  hipLaunchKernelGGL(k_w_relayout, dim3(blocks), dim3(256), 0, as_stream(stream), src, dst, cout, cin, kh * kw, dir);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ SiLU backward: gx = gy * d/dx (x sigmoid(x))
__global__ __launch_bounds__(256) void k_silu_bwd(const float* __restrict__ x, const float* __restrict__ gy,
                                                  float* __restrict__ gx, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    const float s = 1.0f / (1.0f + expf(-v));
    gx[i] = gy[i] * (s * (1.0f + v * (1.0f - s)));
  }
}

extern "C" int tfmq_silu_bwd(tfmq_handle h, const float* x, const float* gy, float* gx, size_t n, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && gy && gx, "silu_bwd: null pointer");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_silu_bwd, dim3(blocks), dim3(256), 0, as_stream(stream), x, gy, gx, n);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ GroupNorm(+SiLU) backward w.r.t. the input
// y = silu?(gamma*xhat + beta), xhat = (x-mean)*rstd per (image, group).  Given gy = dL/dy:
//   gh = gy * silu'(h)   (or gy);   gxh = gh*gamma;   gx = rstd*(gxh - mean_g(gxh) - xhat*mean_g(gxh*xhat))
// one block per (image, group); three reads of the (small, batch-32) group.
__global__ __launch_bounds__(256) void k_groupnorm_bwd(const float* __restrict__ x, const float* __restrict__ gy,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ gx, int HW, int Cc, int cpg, float eps,
                                                       int silu) {
  __shared__ double red[4][2];
  const int b = blockIdx.y, g = blockIdx.x;
  const long base = static_cast<long>(b) * HW * Cc + g * cpg;
  const int n = HW * cpg;
  auto block_sum2 = [&](double a, double c, double& oa, double& oc) {
    a = wave_reduce_sum_d(a);
    c = wave_reduce_sum_d(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = c; }
    __syncthreads();
    oa = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    oc = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  };
  double s = 0.0, ss = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = x[base + static_cast<long>(i / cpg) * Cc + (i % cpg)];
    s += v; ss += static_cast<double>(v) * v;
  }
  double S, SS;
  block_sum2(s, ss, S, SS);
  const double mean_d = S / n;
  double var = SS / n - mean_d * mean_d;
  if (var < 0.0) var = 0.0;
  const float mean = static_cast<float>(mean_d), rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  double a1 = 0.0, a2 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i % cpg;
    const long o = base + static_cast<long>(i / cpg) * Cc + c;
    const float xh = (x[o] - mean) * rstd;
    float gh = gy[o];
    if (silu) {
      const float hval = gamma[g * cpg + c] * xh + beta[g * cpg + c];
      const float sg = 1.0f / (1.0f + expf(-hval));
      gh *= sg * (1.0f + hval * (1.0f - sg));
    }
    const float gxh = gh * gamma[g * cpg + c];
    a1 += gxh; a2 += static_cast<double>(gxh) * xh;
  }
  double A1, A2;
  block_sum2(a1, a2, A1, A2);
  const float m1 = static_cast<float>(A1 / n), m2 = static_cast<float>(A2 / n);
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i % cpg;
    const long o = base + static_cast<long>(i / cpg) * Cc + c;
    const float xh = (x[o] - mean) * rstd;
    float gh = gy[o];
    if (silu) {
      const float hval = gamma[g * cpg + c] * xh + beta[g * cpg + c];
      const float sg = 1.0f / (1.0f + expf(-hval));
      gh *= sg * (1.0f + hval * (1.0f - sg));
    }
    const float gxh = gh * gamma[g * cpg + c];
    gx[o] = rstd * (gxh - m1 - xh * m2);
  }
}

extern "C" int tfmq_groupnorm_bwd(tfmq_handle h, const float* x, const float* gy, const float* gamma, const float* beta,
                                  float* gx, int B, int HW, int C, int groups, float eps, int silu, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && gy && gamma && beta && gx && B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && B < 65536,
                 "groupnorm_bwd: bad argument");
  hipLaunchKernelGGL(k_groupnorm_bwd, dim3(groups, B), dim3(256), 0, as_stream(stream), x, gy, gamma, beta, gx, HW, C,
                     C / groups, eps, silu);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ row softmax forward / backward (materialised attention)
// P[r][:] = softmax(scale * S[r][:]);  dS = scale * P * (dP - sum(dP*P))
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ S, float* __restrict__ P, int cols, float scale) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  const float* s = S + r * cols;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < cols; i += 256) mx = fmaxf(mx, s[i] * scale);
  mx = wave_reduce_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.0f;
  for (int i = threadIdx.x; i < cols; i += 256) sum += expf(s[i] * scale - mx);
  sum = wave_reduce_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  sum = red[0] + red[1] + red[2] + red[3];
  for (int i = threadIdx.x; i < cols; i += 256) P[r * cols + i] = expf(s[i] * scale - mx) / sum;
}

__global__ __launch_bounds__(256) void k_softmax_bwd_rows(const float* __restrict__ P, const float* __restrict__ dP,
                                                          float* __restrict__ dS, int cols, float scale) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  float dot = 0.0f;
  for (int i = threadIdx.x; i < cols; i += 256) dot += P[r * cols + i] * dP[r * cols + i];
  dot = wave_reduce_sum(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  dot = red[0] + red[1] + red[2] + red[3];
  for (int i = threadIdx.x; i < cols; i += 256) dS[r * cols + i] = scale * P[r * cols + i] * (dP[r * cols + i] - dot);
}

extern "C" int tfmq_softmax_rows(tfmq_handle h, const float* S, float* P, long rows, int cols, float scale, void* stream) {
  TFMQ_CHECK_ARG(h, h && S && P && rows > 0 && cols > 0 && rows < 2147483647L, "softmax_rows: bad argument");
  hipLaunchKernelGGL(k_softmax_rows, dim3(static_cast<unsigned>(rows)), dim3(256), 0, as_stream(stream), S, P, cols, scale);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

extern "C" int tfmq_softmax_bwd_rows(tfmq_handle h, const float* P, const float* dP, float* dS, long rows, int cols,
                                     float scale, void* stream) {
  TFMQ_CHECK_ARG(h, h && P && dP && dS && rows > 0 && cols > 0 && rows < 2147483647L, "softmax_bwd_rows: bad argument");
  hipLaunchKernelGGL(k_softmax_bwd_rows, dim3(static_cast<unsigned>(rows)), dim3(256), 0, as_stream(stream), P, dP, dS, cols,
                     scale);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ nearest 2x upsample (materialised only for the
// reconstruction cache of `upsample.conv` units; sampling fuses it into the conv's addressing)
__global__ __launch_bounds__(256) void k_upsample2x(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W,
                                                    int Cc) {
  const long total = static_cast<long>(B) * 2 * H * 2 * W * Cc;
  for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long>(gridDim.x) * blockDim.x) {
    const int c = i % Cc;
    long r = i / Cc;
    const int wo = r % (2 * W); r /= 2 * W;
    const int ho = r % (2 * H);
    const int b = r / (2 * H);
    y[i] = x[((static_cast<long>(b) * H + (ho >> 1)) * W + (wo >> 1)) * Cc + c];
  }
}
extern "C" int tfmq_upsample2x(tfmq_handle h, const float* x, float* y, int B, int H, int W, int C, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && y && B > 0 && H > 0 && W > 0 && C > 0, "upsample2x: bad argument");
  const long total = static_cast<long>(B) * 4 * H * W * C;
  int blocks = ceil_div(total, 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  hipLaunchKernelGGL(k_upsample2x, dim3(blocks), dim3(256), 0, as_stream(stream), x, y, B, H, W, C);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ Fisher-weighted reconstruction
// backward of the nearest 2x upsample: gx[b][h][w][c] = sum of the four g[b][2h+i][2w+j][c] (fixed order)
__global__ __launch_bounds__(256) void k_upsample2x_bwd(const float* __restrict__ g, float* __restrict__ gx, int B, int H, int W,
                                                        int Cc) {
  const long total = static_cast<long>(B) * H * W * Cc;
  for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long>(gridDim.x) * blockDim.x) {
    const int c = i % Cc;
    long r = i / Cc;
    const int w = r % W; r /= W;
    const int hh = r % H;
    const int b = r / H;
    const long row0 = ((static_cast<long>(b) * 2 * H + 2 * hh) * 2 * W + 2 * w) * Cc + c;
    const long row1 = row0 + static_cast<long>(2 * W) * Cc;
    gx[i] = (g[row0] + g[row0 + Cc]) + (g[row1] + g[row1 + Cc]);
  }
}
extern "C" int tfmq_upsample2x_bwd(tfmq_handle h, const float* g, float* gx, int B, int H, int W, int C, void* stream) {
  TFMQ_CHECK_ARG(h, h && g && gx && B > 0 && H > 0 && W > 0 && C > 0, "upsample2x_bwd: bad argument");
  const long total = static_cast<long>(B) * H * W * C;
  int blocks = ceil_div(total, 256);
  if (blocks > h->cu_count * 16) blocks = h->cu_count * 16;
  hipLaunchKernelGGL(k_upsample2x_bwd, dim3(blocks), dim3(256), 0, as_stream(stream), g, gx, B, H, W, C);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// GetLayerGrad's loss (reference quant/data_utill.py:246-247): F.kl_div(log_softmax(out_q, 1), softmax(out_fp, 1), 'batchmean') over the
// channel dimension of NHWC rows [n][C]; loss (optional) accumulates sum p_fp (log p_fp - log p_q) / batch.  One thread per pixel, C <= 64.
//   wrt_target = 0: g = dL/d out_q  = (softmax(out_q) - softmax(out_fp)) / batch
//   wrt_target = 1: g = dL/d out_fp = p_fp (l - sum_c p_fp l) / batch, l = log p_fp - log p_q   (the target branch: the reference does
//                   not detach softmax(out_fp), and its backward hook keeps the gradient of the pass autograd reaches LAST -- the FP one)
__global__ __launch_bounds__(256) void k_kl_softmax_grad(const float* __restrict__ q, const float* __restrict__ f, float* __restrict__ g,
                                                         long n, int Cc, float inv_batch, int wrt_target, float* __restrict__ loss) {
  double acc = 0.0;
  for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<long>(gridDim.x) * blockDim.x) {
    const float* qr = q + i * Cc;
    const float* fr = f + i * Cc;
    float mq = qr[0], mf = fr[0];
    for (int c = 1; c < Cc; ++c) { mq = fmaxf(mq, qr[c]); mf = fmaxf(mf, fr[c]); }
    float sq = 0.0f, sf = 0.0f;
    for (int c = 0; c < Cc; ++c) { sq += expf(qr[c] - mq); sf += expf(fr[c] - mf); }
    const float lq = logf(sq), lf = logf(sf);
    float kl = 0.0f;
    for (int c = 0; c < Cc; ++c) {
      const float lpq = qr[c] - mq - lq, lpf = fr[c] - mf - lf;
      kl += expf(lpf) * (lpf - lpq);
    }
    for (int c = 0; c < Cc; ++c) {
      const float lpq = qr[c] - mq - lq, lpf = fr[c] - mf - lf;
      const float pq = expf(lpq), pf = expf(lpf);
      g[i * Cc + c] = (wrt_target ? pf * ((lpf - lpq) - kl) : (pq - pf)) * inv_batch;
    }
    acc += static_cast<double>(kl);
  }
  if (loss) {
    __shared__ double part[4];
    acc = wave_reduce_sum_d(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, static_cast<float>(((part[0] + part[1]) + (part[2] + part[3])) * inv_batch));
  }
}
extern "C" int tfmq_kl_softmax_grad(tfmq_handle h, const float* out_q, const float* out_fp, float* g, long n_rows, int C, int batch,
                                    int wrt_target, float* loss_or_null, void* stream) {
  TFMQ_CHECK_ARG(h, h && out_q && out_fp && g && n_rows > 0 && C > 0 && C <= 64 && batch > 0, "kl_softmax_grad: bad argument (C <= 64)");
  int blocks = ceil_div(n_rows, 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_kl_softmax_grad, dim3(blocks), dim3(256), 0, as_stream(stream), out_q, out_fp, g, n_rows, C,
                     1.0f / static_cast<float>(batch), wrt_target, loss_or_null);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// LossFunc's Fisher modes (reference quant/reconstruction_util.py:53-59); fg = the cached |dL/d out| + 1 of save_grad.
//   DIAG: rec = ((pred - tgt)^2 * fg^2).sum(1).mean()            = sum d^2 fg^2 / denom,           g = 2 d fg^2 / denom
//   FULL: a = |pred - tgt|, w = |fg|, s_b = sum_{chw} a w;  rec = (s_b * a * w).mean() / 100 = sum_b s_b^2 / (100 n),
//         g = 2 s_b w sign(d) / (100 n)   (autograd differentiates both occurrences of a; sign(0) = 0 like torch's abs)
// k_fisher_dot: per-sample s_b in double (one block per sample, fixed order).  k_fisher_loss: loss + gradient.
__global__ __launch_bounds__(256) void k_fisher_dot(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ fg,
                                                    size_t per_sample, double* __restrict__ dot) {
  const size_t base = static_cast<size_t>(blockIdx.x) * per_sample;
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < per_sample; i += blockDim.x)
    acc += static_cast<double>(fabsf(pred[base + i] - tgt[base + i])) * fabsf(fg[base + i]);
  __shared__ double part[4];
  acc = wave_reduce_sum_d(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dot[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256) void k_fisher_loss(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ fg,
                                                     float* __restrict__ g, size_t n, size_t per_sample, int mode, float inv_denom,
                                                     const double* __restrict__ dot, float* __restrict__ loss) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  double acc = 0.0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = pred[i] - tgt[i], w = fg[i];
    if (mode == 1) {
      const float dw = d * d * (w * w);
      acc += static_cast<double>(dw);
      if (g) g[i] = 2.0f * d * (w * w) * inv_denom;
    } else {
      const float sb = static_cast<float>(dot[i / per_sample]);
      const float aw = fabsf(d) * fabsf(w);
      acc += static_cast<double>(sb) * aw;
      if (g) g[i] = 2.0f * sb * fabsf(w) * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * inv_denom;
    }
  }
  __shared__ double part[4];
  acc = wave_reduce_sum_d(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, static_cast<float>(((part[0] + part[1]) + (part[2] + part[3])) * inv_denom));
}
extern "C" int tfmq_fisher_loss(tfmq_handle h, const float* pred, const float* tgt, const float* fgrad, float* g, size_t n_samples,
                                size_t per_sample, int mode, size_t denom, double* dot_scratch, float* loss, void* stream) {
  TFMQ_CHECK_ARG(h, h && pred && tgt && fgrad && loss && n_samples > 0 && per_sample > 0 && (mode == 1 || mode == 2) && denom > 0,
                 "fisher_loss: bad argument (mode 1 = FISHER_DIAG, 2 = FISHER_FULL)");
  TFMQ_CHECK_ARG(h, mode == 1 || (dot_scratch && n_samples < 2147483647UL), "fisher_loss: FISHER_FULL needs n_samples doubles of scratch");
  const size_t n = n_samples * per_sample;
  // FULL: mean over all n elements, / 100
  const double inv = mode == 1 ? 1.0 / static_cast<double>(denom) : 1.0 / (100.0 * static_cast<double>(n));
  if (mode == 2) {
    hipLaunchKernelGGL(k_fisher_dot, dim3(static_cast<unsigned>(n_samples)), dim3(256), 0, as_stream(stream), pred, tgt, fgrad, per_sample,
                       dot_scratch);
    TFMQ_LAUNCH_CHECK(h);
  }
  int blocks = ceil_div(static_cast<long>(n), 1024);
  if (blocks > h->cu_count * 4) blocks = h->cu_count * 4;
  hipLaunchKernelGGL(k_fisher_loss, dim3(blocks), dim3(256), 0, as_stream(stream), pred, tgt, fgrad, g, n, per_sample, mode,
                     static_cast<float>(inv), dot_scratch, loss);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ flat gradient buffer helpers (K16: one all-reduce per iteration)
__global__ void k_scale_add(float* __restrict__ y, const float* __restrict__ x, float a, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) y[i] += a * x[i];
}
extern "C" int tfmq_axpy(tfmq_handle h, float* y, const float* x, float a, size_t n, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && y, "axpy: null pointer");
  if (n == 0) return TFMQ_OK;
  int blocks = ceil_div(static_cast<long>(n), 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_scale_add, dim3(blocks), dim3(256), 0, as_stream(stream), y, x, a, n);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ LayerNorm backward w.r.t. the input
// y = gamma*xhat + beta, xhat = (x-mean)*rstd over the C channels of a token.  Given gy:
//   gxh = gy*gamma;  gx = rstd*(gxh - mean(gxh) - xhat*mean(gxh*xhat)).  One block per token, double sums.
__global__ __launch_bounds__(256) void k_layernorm_bwd(const float* __restrict__ x, const float* __restrict__ gy,
                                                       const float* __restrict__ gamma, float* __restrict__ gx, int Cc,
                                                       float eps) {
  __shared__ double red[4][2];
  const long base = static_cast<long>(blockIdx.x) * Cc;
  auto block_sum2 = [&](double a, double c, double& oa, double& oc) {
    a = wave_reduce_sum_d(a);
    c = wave_reduce_sum_d(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = c; }
    __syncthreads();
    oa = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    oc = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  };
  double s = 0.0, ss = 0.0;
  for (int c = threadIdx.x; c < Cc; c += 256) {
    const double v = x[base + c];
    s += v;
    ss += v * v;
  }
  double S, SS;
  block_sum2(s, ss, S, SS);
  const double mean = S / Cc;
  double var = SS / Cc - mean * mean;
  if (var < 0.0) var = 0.0;
  const float meanf = static_cast<float>(mean), rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  double a = 0.0, b = 0.0;
  for (int c = threadIdx.x; c < Cc; c += 256) {
    const float xh = (x[base + c] - meanf) * rstd;
    const float g = gy[base + c] * gamma[c];
    a += g;
    b += static_cast<double>(g) * xh;
  }
  double A, Bq;
  block_sum2(a, b, A, Bq);
  const float ma = static_cast<float>(A / Cc), mb = static_cast<float>(Bq / Cc);
  for (int c = threadIdx.x; c < Cc; c += 256) {
    const float xh = (x[base + c] - meanf) * rstd;
    const float g = gy[base + c] * gamma[c];
    gx[base + c] = rstd * (g - ma - xh * mb);
  }
}

extern "C" int tfmq_layernorm_bwd(tfmq_handle h, const float* x, const float* gy, const float* gamma, float eps, long rows,
                                  int C, float* gx, void* stream) {
  TFMQ_CHECK_ARG(h, h && x && gy && gamma && gx && rows > 0 && C > 0, "layernorm_bwd: bad argument");
  hipLaunchKernelGGL(k_layernorm_bwd, dim3(static_cast<unsigned>(rows)), dim3(256), 0, as_stream(stream), x, gy, gamma, gx, C, eps);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}

// ------------------------------------------------------------------ GEGLU backward
// y[m][i] = a*gelu(g), a = h[m][i], g = h[m][I+i]:  dh[m][i] = dy*gelu(g);  dh[m][I+i] = dy*a*gelu'(g),
// gelu'(g) = Phi(g) + g*phi(g).
__global__ __launch_bounds__(256) void k_geglu_bwd(const float* __restrict__ hin, const float* __restrict__ dy,
                                                   float* __restrict__ dh, long rows, int I) {
  const long total = rows * I;
  for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long>(gridDim.x) * blockDim.x) {
    const long m = i / I;
    const int c = static_cast<int>(i - m * I);
    const float a = hin[m * 2 * I + c], g = hin[m * 2 * I + I + c], d = dy[i];
    const float Phi = 0.5f * (1.0f + erf_fast_f(g * 0.70710678118654752440f));
    const float phi = 0.39894228040143267794f * expf(-0.5f * g * g);
    dh[m * 2 * I + c] = d * (g * Phi);
    dh[m * 2 * I + I + c] = d * a * (Phi + g * phi);
  }
}

extern "C" int tfmq_geglu_bwd(tfmq_handle h, const float* hin, const float* dy, long rows, int inner, float* dh,
                              void* stream) {
  TFMQ_CHECK_ARG(h, h && hin && dy && dh && rows > 0 && inner > 0, "geglu_bwd: bad argument");
  int blocks = ceil_div(rows * inner, 256);
  if (blocks > h->cu_count * 8) blocks = h->cu_count * 8;
  hipLaunchKernelGGL(k_geglu_bwd, dim3(blocks), dim3(256), 0, as_stream(stream), hin, dy, dh, rows, inner);
  TFMQ_LAUNCH_CHECK(h);
  return TFMQ_OK;
}
