/* CPU oracle, plain-C restatement of the integer / byte-level pieces of the TFMQ hot path.
 * TEST INFRASTRUCTURE ONLY: used by tests/ (and smoke()) as an independent checker of the HIP
 * kernels' bit-exact claims.  Never linked into or called by the product (tfmq-dm_amd/).
 * Pinned to the reference through tests/test_oracle_c.py (golden vectors F1/F3/F4).
 *
 * Each function cites the reference lines it restates (paths relative to the reference root).
 * Compile: gcc -O2 -fPIC -shared -ffp-contract=off -o oracle/_build/liboracle_c.so oracle/tfmq_oracle_c.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* UniformAffineQuantizer.forward bin index (quant/quant_layer.py:223-225):
 * q = clamp(rint(x/delta) + zp, 0, level-1); IEEE division, round-half-even. */
void oc_quant_index(const float* x, size_t n, float delta, float zp, int level, uint8_t* q) {
  const float lmax = (float)(level - 1);
  for (size_t i = 0; i < n; ++i) {
    float v = rintf(x[i] / delta) + zp;
    v = v < 0.0f ? 0.0f : (v > lmax ? lmax : v);
    q[i] = (uint8_t)(int)v;
  }
}

/* minmax scaler (quant/quant_layer.py:20-35, symmetric=False) */
void oc_minmax_qparam(float mn, float mx, int level, int always_zero, float* delta, float* zp) {
  double lo = mn < 0.0f ? (double)mn : 0.0, hi = mx > 0.0f ? (double)mx : 0.0;
  float d = (float)((hi - lo) / (double)(level - 1));
  if (always_zero) d = (float)(hi / (double)(level - 1));
  if (d < 1e-8f) d = 1e-8f;
  *delta = d;
  *zp = always_zero ? 0.0f : rintf((float)(-lo) / d);
}

/* the 80 candidates of the mse scaler (quant/quant_layer.py:45-55) */
void oc_mse_candidates(float mn, float mx, int level, int always_zero, float* deltas, float* zps) {
  for (int i = 0; i < 80; ++i) {
    double f = 1.0 - ((double)i * 0.01);
    double nmin = (double)mn * f, nmax = (double)mx * f;
    float nd = (float)((nmax - nmin) / (double)(level - 1));
    if (always_zero) nd = (float)(nmax / (double)(level - 1));
    deltas[i] = nd;
    zps[i] = always_zero ? 0.0f : rintf((float)(-nmin) / nd);
  }
}

/* weight bin index, nearest (quant_layer.py:225) or AdaRound hard (adaptive_rounding.py:51,63,67-68);
 * w: [cout][k], delta/zp per row, alpha NULL => nearest */
void oc_weight_index(const float* w, const float* alpha, const float* delta, const float* zp, int cout, int k,
                     int level, uint8_t* q) {
  const float lmax = (float)(level - 1);
  for (int c = 0; c < cout; ++c)
    for (int j = 0; j < k; ++j) {
      size_t i = (size_t)c * k + j;
      float v = alpha ? floorf(w[i] / delta[c]) + (alpha[i] >= 0.0f ? 1.0f : 0.0f) + zp[c] : rintf(w[i] / delta[c]) + zp[c];
      v = v < 0.0f ? 0.0f : (v > lmax ? lmax : v);
      q[i] = (uint8_t)(int)v;
    }
}

/* Integer-accumulate w4a8 conv (NHWC in, OIHW weight indices) == F.conv2d on the two fake-quantised
 * operands (quant/quant_layer.py:318-338) with exact int32 accumulation:
 *   y = da*dw[c] * sum (qa - za)(qw - zw[c]) + b[c];   zero padding contributes (za - za) = 0. */
void oc_conv_w4a8(const uint8_t* qa, int B, int H, int W, int Cin, const uint8_t* qw, int Cout, int KH, int KW,
                  int stride, int pad_t, int pad_l, int Ho, int Wo, float da, int za, const float* dw,
                  const int* zw, const float* bias, float* y) {
  for (int b = 0; b < B; ++b)
    for (int ho = 0; ho < Ho; ++ho)
      for (int wo = 0; wo < Wo; ++wo)
        for (int co = 0; co < Cout; ++co) {
          int32_t acc = 0;
          for (int kh = 0; kh < KH; ++kh)
            for (int kw = 0; kw < KW; ++kw) {
              int hi = ho * stride + kh - pad_t, wi = wo * stride + kw - pad_l;
              if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
              const uint8_t* a = qa + (((size_t)b * H + hi) * W + wi) * Cin;
              for (int ci = 0; ci < Cin; ++ci) {
                int qwv = qw[(((size_t)co * Cin + ci) * KH + kh) * KW + kw];
                acc += ((int)a[ci] - za) * (qwv - zw[co]);
              }
            }
          y[(((size_t)b * Ho + ho) * Wo + wo) * Cout + co] = (da * dw[co]) * (float)acc + (bias ? bias[co] : 0.0f);
        }
}
