/*
 * libtfmq_hip.so -- C ABI of the MI355X (gfx950) kernels behind the TFMQ-DM hot path.
 *
 * The reference (ModelTC/TFMQ-DM) has no FFI/plugin boundary: its hot path is plain
 * PyTorch.  This ABI is therefore the *new* boundary that the Python
 * mirror of the reference's `quant/` surface binds with ctypes.
 * Each entry point names the reference code it replaces (paths relative to the
 * reference root).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; tfmq_last_error() gives text.
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller owns
 *     every buffer (the library allocates nothing after tfmq_create, so every call is
 *     legal inside a HIP stream capture).
 *   - `stream` is a hipStream_t passed as void*; NULL = the legacy default stream.
 *   - activations are NHWC ("pixel-major"): [B][H][W][C]; token tensors [B][T][C] are the
 *     same layout.  fp32 unless stated.  Quantised activations are stored as
 *     int8 = bin_index - 128 (bin_index in [0,255], quant/quant_layer.py:225).
 *   - a "qparam" is a device float2 {delta, zero_point} of one activation quantizer; the
 *     kernels read it from qtable[(*step_ptr) * q_stride + qid] so that one captured
 *     hipGraph serves every Finite-Set-Calibration step (`act_k`).
 *   - a handle is bound to one device and is not thread-safe.
 */
#ifndef TFMQ_HIP_H
#define TFMQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tfmq_ctx* tfmq_handle;

#define TFMQ_OK 0
#define TFMQ_ERR_ARG (-1)
#define TFMQ_ERR_HIP (-2)
#define TFMQ_ERR_UNSUPPORTED (-3)

/* ---- handle ---------------------------------------------------------------------- */
int tfmq_create(int device, tfmq_handle* out);
int tfmq_destroy(tfmq_handle h);
const char* tfmq_last_error(tfmq_handle h);
/* cu_count, max clock (kHz), total HBM bytes of the bound device */
int tfmq_device_info(tfmq_handle h, int* cu_count, int* clock_khz, size_t* hbm_bytes);
int tfmq_abi_version(void);

/* Activation-quantizer parameter selector (FSC table lookup, replaces the per-step
 * `model.load_state_dict(act_k)` of ddim/functions/denoising.py:26-29 and
 * ldm/models/diffusion/ddpm.py:1403-1405). */
typedef struct tfmq_qsel {
  const float* qtable;   /* [n_steps][q_stride][2] = {delta, zero_point}; NULL = no act-quant */
  const int32_t* step;   /* device scalar: current FSC group k; NULL = row 0 */
  int32_t q_stride;      /* quantizers per step row */
  int32_t qid;           /* which quantizer */
} tfmq_qsel;

/* ---- K1: activation quantizer (UniformAffineQuantizer.forward, quant_layer.py:223-226) */
/* q[i] = clamp(rint(x[i]/delta)+zp, 0, level-1) - 128   (level <= 256) */
int tfmq_quantize_act(tfmq_handle h, const float* x, int8_t* q, size_t n, tfmq_qsel qs, int level, void* stream);
/* the same on an fp16 tensor (the fp16 activation stream); n % 4 == 0 */
int tfmq_quantize_act_h(tfmq_handle h, const uint16_t* x, int8_t* q, size_t n, tfmq_qsel qs, int level, void* stream);
/* y[i] = delta * (clamp(rint(x[i] * pre / delta) + zp, 0, level - 1) - zp) with {delta, zp} of the current Finite-Set group (device
 * step counter): a fake-quantised activation for the enable-able attention-matmul quantizers (aqtizer_q / _k / _v / _w of
 * QuantAttnBlock, cross_attn_forward, QuantQKMatMul / QuantSMVMatMul: quant_block.py:226-243,318-323,350-351,487-498; level up to 2^16
 * for the softmax quantizer, pre = the d^-1/4 scale QuantQKMatMul applies in front of its quantizers, else 1). */
int tfmq_fake_quant_sel(tfmq_handle h, const float* x, float* y, size_t n, tfmq_qsel qs, int level, float pre, void* stream);
/* ---- W8A8 (the README's --wq 8 recipes, README.md:86-125; QuantLayer.forward quant_layer.py:306-340 with 8-bit weights):
 * 8-bit weights with a per-channel zero point need 9 bits as (q_w - z_w), which does not fit the int8 MFMA operand; both integer
 * grids are exact in fp16, so such a layer runs tfmq_conv2d_f16 on (b - z_a) x (q_w - z_w) with fp32 accumulation (at least as
 * exact as the reference's fp32 conv of the dequantised values) and the per-channel output scale delta_a * delta_w[c].
 * grid[i] = (xq[i] + 128) - zp(step) as fp16 (out_f16 != 0; n % 4 == 0) or fp32: the activation bins of tfmq_quantize_act on
 * their integer grid. */
int tfmq_bins_to_grid(tfmq_handle h, const int8_t* xq, tfmq_qsel qs, void* out, int out_f16, size_t n, void* stream);
/* out[c] = delta(step) * ws[c], c < n: the output scale of a W8A8 layer under the current Finite-Set group */
int tfmq_scale_by_qdelta(tfmq_handle h, const float* ws, tfmq_qsel qs, float* out, int n, void* stream);
/* y[i] = delta * (clamp(rint(x/delta)+zp,0,level-1) - zp); delta/zp per tensor (rows=1)
 * or per row ([rows] arrays, the per-output-channel weight quantizer, quant_layer.py:193-204) */
int tfmq_fake_quant(tfmq_handle h, const float* x, float* y, uint8_t* idx_or_null, size_t rows, size_t cols,
                    const float* delta, const float* zp, int level, void* stream);
/* Backward of the per-tensor fake quantisation through the straight-through round -- the delta-learning reconstruction mode
 * (reference quant/quant_layer.py:211-227 under autograd; quant/reconstruction.py:135-166): gx (may be NULL) = g where the bin is
 * inside [0, level-1], else 0; part[nparts] (double) = per-block partial sums of dL/ddelta = sum g * ((q - zp) - (x/delta)[in range]),
 * to be added in index order by the caller (deterministic).  delta, zp: device scalars. */
int tfmq_fake_quant_bwd(tfmq_handle h, const float* x, const float* g, float* gx, size_t n, const float* delta, const float* zp,
                        int level, double* part, int nparts, void* stream);

/* ---- K2: min/max statistics (minmax, quant_layer.py:20-35; act_momentum_update :229-244) */
/* out[r] = {min, max} of row r ([rows][cols] fp32); rows=1 => whole tensor.
 * ws: caller workspace of at least tfmq_minmax_ws_bytes(rows, cols) bytes. */
size_t tfmq_minmax_ws_bytes(size_t rows, size_t cols);
int tfmq_minmax(tfmq_handle h, const float* x, size_t rows, size_t cols, float* out, void* ws, void* stream);
/* MINMAX scaler from {min,max}: qparam[r] = {delta, zp} exactly as quant_layer.py:23-35
 * (double subtraction, fp32 delta, zp = rint(fp32(-lo)/delta)); always_zero per :29-30,34. */
int tfmq_minmax_to_qparam(tfmq_handle h, const float* mm, size_t rows, int level, int always_zero, float* qparam, void* stream);
/* running-stat EMA (quant_layer.py:233-244): state = {x_min, x_max}; mm = batch {min,max};
 * state <- m*state + (1-m)*mm, qparam <- MINMAX(state).  init!=0: state <- mm first (:206-207). */
int tfmq_act_range_update(tfmq_handle h, const float* mm, float* state, float* qparam, double momentum, int level,
                          int init, void* stream);

/* ---- K3: MSE scale search (mse, quant_layer.py:38-64) ------------------------------- */
/* For every row: 80 shrink candidates (computed on device in double exactly as the Python),
 * L2.4 loss of each in ONE pass over the data, first strict minimum.  Outputs per row:
 * qparam[r] = {delta, zp}; losses_or_null [rows][80]; best_or_null [rows].
 * mm: [rows][2] from tfmq_minmax.  ws: tfmq_mse_ws_bytes(rows, cols) bytes. */
size_t tfmq_mse_ws_bytes(size_t rows, size_t cols);
int tfmq_mse_search(tfmq_handle h, const float* x, size_t rows, size_t cols, const float* mm, int level,
                    int always_zero, float* qparam, float* losses_or_null, int32_t* best_or_null, void* ws,
                    void* stream);

/* ---- K4: int4 weight packing (weight fake-quant recomputed every forward in the reference:
 * UniformAffineQuantizer.forward quant_layer.py:223-226 / AdaRoundQuantizer.forward
 * adaptive_rounding.py:51-69) ------------------------------------------------------------ */
/* w: [cout][cin][kh][kw] fp32 (PyTorch OIHW; Linear = kh=kw=1).  alpha NULL => nearest
 * rounding, else hard AdaRound (floor + [alpha>=0]).  delta/zp: [cout].
 * packed: ceil32(cout) * K/2 bytes, K = kh*kw*cin ordered (kh,kw,cin) to match NHWC activations;
 *   each 32-bit word holds 8 consecutive k: byte i = q[k+i] | q[k+4+i] << 4  (i=0..3);
 *   words are stored tile-major: 32-row blocks of output channels, inside a block one K-step
 *   (ck = 64 or 32 input channels) of all 32 rows is contiguous, so the GEMM loader reads whole
 *   128-byte lines: word(n, g) = ((n/32 * K/ck + g/(ck/8)) * 32 + n%32) * (ck/8) + g%(ck/8).
 * wmeta: [cout][4] int32 = {zp, rowsum_q = sum_k q, 0, 0}.  cin must be a multiple of 8. */
int tfmq_pack_w4(tfmq_handle h, const float* w, const float* alpha_or_null, const float* delta, const float* zp,
                 int cout, int cin, int kh, int kw, uint8_t* packed, int32_t* wmeta, void* stream);
/* inverse (tests): idx[cout][cin][kh][kw] u8 */
int tfmq_unpack_w4(tfmq_handle h, const uint8_t* packed, int cout, int cin, int kh, int kw, uint8_t* idx, void* stream);
/* packed int4 -> the conv kernels' int8 operand: byte (n,k) = q_w - z_w, tile-major
 * [cout_pad32/32][K/ck][32][ck] (ck = 64 if cin % 64 == 0 else 32; cin % 32 == 0 required); w8 holds
 * cout_pad32 * kh*kw*cin bytes.  This is what tfmq_conv_desc.w points at for tfmq_conv2d_w4a8. */
int tfmq_expand_w4(tfmq_handle h, const uint8_t* packed, const int32_t* wmeta, int cout, int cin, int kh, int kw,
                   int8_t* w8, void* stream);
/* The same operand with every tap's channels padded to a multiple of 64 (zeros): byte(n, tap, c) =
 * ((n/32 * nsteps + tap*chunks + c/64) * 32 + n%32) * 64 + c%64, chunks = ceil(cin/64), nsteps = kh*kw*chunks; for cin % 64 == 32
 * (tfmq_conv_desc.w64).  w8p: cout_pad32 * nsteps * 64 bytes. */
int tfmq_expand_w4_k64(tfmq_handle h, const uint8_t* packed, const int32_t* wmeta, int cout, int cin, int kh, int kw, int8_t* w8p,
                       void* stream);
/* fp16 weights for the un-quantised convs, reordered to [cout][kh][kw][cin_pad],
 * cin_pad = cin rounded up to a multiple of 32 (zero filled).  With delta/zp (and optional
 * alpha) non-NULL the stored value is the integer grid coordinate q - zp (exact in f16) of the
 * 4-bit weight quantizer, for weight-only layers (disable_aq, quant_model.py:110-120). */
int tfmq_pack_w_f16(tfmq_handle h, const float* w, const float* alpha_or_null, const float* delta_or_null,
                    const float* zp_or_null, int level, int cout, int cin, int kh, int kw, uint16_t* out, void* stream);

/* ---- K5/K6: conv / linear as implicit GEMM on MFMA (QuantLayer.forward, quant_layer.py:306-340) */
typedef struct tfmq_conv_desc {
  /* geometry */
  int32_t B, H, W, Cin;          /* input  [B][H][W][Cin] */
  int32_t Cout, KH, KW, stride;  /* 1x1 stride 1 == Linear over B*H*W tokens */
  int32_t pad_t, pad_l;          /* zero padding before the first row/col */
  int32_t Ho, Wo;                /* output spatial size */
  int32_t up2x;                  /* !=0: the conv reads a virtual nearest-2x upsample of the input
                                    (Upsample.forward, ddim/models/diffusion.py:47-52): H,W are the
                                    stored input size, the conv sees 2H x 2W */
  /* operands */
  const void* x;                 /* int8 (w4a8) or fp32 (f16 path) NHWC */
  const void* w;                 /* w4a8: int8 (q_w - z_w) tiles (tfmq_expand_w4); f16 path: fp16 (tfmq_pack_w_f16) */
  const int32_t* wmeta;          /* [Cout][4] from tfmq_pack_w4 (w4a8 only) */
  const float* wscale;           /* [Cout] delta_w.  w4a8: required.  f16 path: optional per-channel output scale
                                    (weight-only layers store the integer grid q-z exactly in f16) */
  const float* bias;             /* [Cout] or NULL */
  tfmq_qsel aq;                  /* activation quantizer of x (w4a8 only) */
  /* fused epilogue: y = conv + bias (+ rowadd[b][c]) (+ residual[b][ho][wo][c]) */
  const float* rowadd;           /* per-image channel bias (temb projection, quant_block.py:429) or NULL:
                                    value = rowadd[(*rowadd_step)*rowadd_step_stride + b*rowadd_ld + c] */
  const int32_t* rowadd_step;    /* device scalar selecting a precomputed TIB row (NULL = 0) */
  int32_t rowadd_ld;             /* floats between images (0 = one row broadcast to the whole batch) */
  int32_t rowadd_step_stride;    /* floats between steps */
  const float* residual;         /* fp32 NHWC [B][Ho][Wo][Cout] or NULL */
  float* y;                      /* fp32 NHWC [B][Ho][Wo][ldy]  (written at channel offset y_coff) */
  int32_t ldy, y_coff;           /* output row stride in floats (>= Cout) and channel offset: lets q/k/v or a
                                    concat target share one buffer */
  float* stats;                  /* optional [ceil(M/stats_seg)][Cout][2]: per-channel {sum, sum of squares} of every
                                    stats_seg consecutive output pixels, for the GroupNorm that consumes y (K8) */
  int32_t stats_seg;             /* 16, 32, 64 or 128; must divide Ho*Wo */
  int32_t out_mode;              /* TFMQ_OUT_F32 (0): y as above.
                                    TFMQ_OUT_F16: y is an fp16 buffer (ldy / y_coff in halves): operands of the
                                      f16 attention kernel (q/k/v projections), rounded once exactly as that kernel
                                      would on load.
                                    TFMQ_OUT_GEGLU_Q8 (w4a8, Linear only): the layer is GEGLU's projection
                                      (ldm/modules/attention.py:52-59) with its 2*inner output channels interleaved
                                      per 128-column tile as [64 value | 64 gate] (packed row p holds original row
                                      (p/128)*64 + p%64 (+ inner when p%128 >= 64)); the epilogue
                                      writes yq[m][c] = quant_oq(value * gelu(gate)) - 128, int8 [M][Cout/2],
                                      i.e. the next QuantLayer's input (quant_layer.py:312-313), and y is unused.
                                    TFMQ_OUT_Q8: yq[m][c] = quant_oq(conv + bias (+ rowadd) (+ residual)) - 128, int8
                                      [M][Cout]: for an output whose ONLY consumer is the next QuantLayer's
                                      activation quantizer (quant_layer.py:312-313); y is unused.
                                    TFMQ_OUT_GEGLU_Q8_FAST (4; round 4): TFMQ_OUT_GEGLU_Q8 with the epilogue arithmetic sized for its
                                      consumer -- the value is rounded to one of 256 bins next instruction: Phi(g) as a logistic of an
                                      odd quintic (|g Phi(g) - gelu(g)| <= 2.8e-5), the output delta folded into the value's scale, the
                                      zero-point corrections into the biases: 13.75 instead of 22.75 VALU issue slots per output.
                                      Bins within 1 of TFMQ_OUT_GEGLU_Q8's, < 2e-3 of them moved (tests/test_geglu_fast_gpu.py).
                                      Only the register-direct pointwise kernel (tile AUTO / DIRECT) takes it; otherwise
                                      TFMQ_ERR_UNSUPPORTED.  The sampling path's default (TFMQ_GELU_EXACT=1 restores mode 2).
                                    rowadd / residual are defined for TFMQ_OUT_F32 and TFMQ_OUT_Q8, stats for F32. */
  tfmq_qsel oq;                  /* TFMQ_OUT_GEGLU_Q8 / TFMQ_OUT_Q8: activation quantizer of the consumer */
  int8_t* yq;                    /* TFMQ_OUT_GEGLU_Q8 / TFMQ_OUT_Q8: int8 output */
  uint16_t* yt;                  /* TFMQ_OUT_F16, optional: output channels n >= t_col0 are written TRANSPOSED,
                                    fp16 yt[b][n - t_col0][t], t = ho*Wo + wo (instead of into y): the V^T operand of
                                    tfmq_attention_f16 straight from the fused q|k|v projection */
  int32_t t_col0;                /* multiple of 128; Ho*Wo % 4 == 0 required */
  int32_t x_f16;                 /* tfmq_conv2d_f16 only: x is fp16 NHWC (written as fp16 by its producer, e.g.
                                    tfmq_gn_desc.half_out): operands go global -> LDS by DMA like the w4a8 path.
                                    Needs Cin % 32 == 0 and KH*KW <= 9; bit-identical to the fp32-input path, which
                                    rounds x to fp16 while staging */
  int32_t tile;                  /* TFMQ_TILE_AUTO (0): the library's rule picks the tile shape.  TFMQ_TILE_128 / _64 / _256 / _128x64:
                                    128x128, 64x64, 256x128 or 128x64 output tiles -- the result does not depend on the choice
                                    (int32 sums are exact; the f16 path accumulates each output in the same K order);
                                    a shape the launch is not eligible for falls back to the rule.  Lets a host time
                                    the variants per layer shape once and pin the fastest (ops.set_conv_autotune).
                                    TFMQ_TILE_SLAB: the 3x3 / stride-1 / pad-1 w4a8 kernel that stages the activation slab of a
                                    256-pixel tile once per channel chunk for all nine taps (256 x 320 / 256 / 128 tiles, 8 waves).
                                    TFMQ_TILE_DIRECT: pointwise w4a8 layers with fp16 / int8 / GEGLU-int8 output on the kernel whose
                                    epilogue runs out of the accumulator registers (swapped MFMA operands, no LDS staging).
                                    TFMQ_TILE_SLAB128 (7): the slab kernel on 128-pixel tiles and four waves, two blocks per CU (one block's
                                    epilogue runs under the other's MFMAs; twice the blocks at the 8x8 / 16x16 levels) -- same bits.
                                    (8 was a persistent variant of TFMQ_TILE_DIRECT in round 2 -- measured slower on every shape and
                                    removed in round 3; the value falls back to the rule)
                                    TFMQ_TILE_DIRECT256 (9; round 6): TFMQ_TILE_DIRECT on 256 x 128 tiles, two blocks per CU (a quarter fewer
                                    LDS-DMA pieces per MFMA); layers without a residual, else the 128-row form runs -- same bits. */
  int32_t res_f16;               /* != 0: `residual` is an fp16 buffer [B][Ho][Wo][Cout] (a tensor of the fp16 activation stream:
                                    the TFMQ_OUT_F16 output of an earlier launch) */
  const void* x2;                /* tfmq_conv2d_f16, pointwise, x_f16 only; NULL = one source.  Input channels [cin1, Cin) are read
                                    from x2, an fp16 tensor [B][H][W][Cin - cin1]: the layer sees the channel concat cat(x, x2)
                                    without its copy (the skip_connection / nin_shortcut of an up-path ResBlock reading
                                    th.cat([h, hs.pop()], dim=1), openaimodel.py:771 / ddim/models/diffusion.py:337).  x is then
                                    [B][H][W][cin1].  Needs cin1 % 32 == 0, (Cin - cin1) % 32 == 0 and a launch the register-direct
                                    pointwise kernel takes (fp16 output, no rowadd); anything else is TFMQ_ERR_ARG */
  int32_t cin1;
  const void* w64;               /* tfmq_conv2d_w4a8, Cin % 64 == 32 only (optional): the weight operand of tfmq_expand_w4_k64 -- 64-channel
                                    K-steps with the last one of every tap zero-padded.  The LDS-DMA kernels (slab, register-direct
                                    pointwise, DMA tile kernels) then take such layers too: their last K-step of a pixel reads 32 bytes
                                    past its channel row (the next pixel's, times zero weights), so x must be followed by >= 32 readable
                                    bytes.  NULL: those layers run on the register-staged kernel with 32-channel K-steps */
  int32_t ksplit;                /* tfmq_conv2d_w4a8 on the LDS-DMA tile kernels (tile = TFMQ_TILE_128 / _64 / _128x64 / _256 or the rule's
                                    choice of one of them), 0 / 1 = off.  k > 1: every output tile is computed by k workgroups, each over
                                    a contiguous 1/k of the K-steps; they publish their int32 partial sums to a slab of the handle's
                                    workspace and take a ticket, and the last arriver adds the slabs and runs the epilogue.  Integer sums:
                                    the result is bit-identical to the unsplit launch.  For launches whose M x Cout grid has fewer tiles
                                    than the chip has CUs and whose K is long -- the 1280-channel 3x3 convs at 8x8 / 16x16 of a
                                    UNet(2) forward (1 image under guidance) stream 14.7 MB of weights through 10 ... 40 CUs otherwise.
                                    Needs tiles * k * tile elements <= 16 Mi (the handle's 64 MiB of slabs) and k <= K-steps, else
                                    TFMQ_ERR_ARG; split-K launches of one handle must be stream-ordered.  Ignored by the slab /
                                    register-direct kernels (pin a tile shape together with ksplit). */
} tfmq_conv_desc;
enum { TFMQ_OUT_F32 = 0, TFMQ_OUT_F16 = 1, TFMQ_OUT_GEGLU_Q8 = 2, TFMQ_OUT_Q8 = 3, TFMQ_OUT_GEGLU_Q8_FAST = 4 };
enum { TFMQ_TILE_AUTO = 0, TFMQ_TILE_128 = 1, TFMQ_TILE_64 = 2, TFMQ_TILE_256 = 3, TFMQ_TILE_128x64 = 4, TFMQ_TILE_SLAB = 5, TFMQ_TILE_DIRECT = 6,
       TFMQ_TILE_SLAB128 = 7, TFMQ_TILE_DIRECT256 = 9 };
int tfmq_conv2d_w4a8(tfmq_handle h, const tfmq_conv_desc* d, void* stream);
int tfmq_conv2d_f16(tfmq_handle h, const tfmq_conv_desc* d, void* stream);

/* ---- K5f (round 4): the feed-forward half of a BasicTransformerBlock as one launch:
 *   y = ff.net.2( quant_aq2( value * gelu(gate) ) ) + x,   value | gate = ff.net.0.proj( quant_aq0( LayerNorm(x) ) )
 * (`x = self.ff(self.norm3(x)) + x`, ldm/modules/attention.py:37-64,152-215 under QuantBasicTransformerBlock, quant/quant_block.py:248-299;
 * both Linears are w4a8 QuantLayers, quant/quant_layer.py:306-340).  Replaces tfmq_layernorm_h + tfmq_conv2d_w4a8(TFMQ_OUT_GEGLU_Q8_FAST) +
 * tfmq_conv2d_w4a8(TFMQ_OUT_F16 | TFMQ_OUT_Q8, residual) and agrees with that chain bit for bit; HBM sees the fp16 row once in, once out.
 * A workgroup owns 256 tokens, a lane one token; the GEGLU bins go from the accumulators straight into the second GEMM's MFMA operand. */
typedef struct tfmq_ff_desc {
  int32_t M, C, inner;           /* tokens, token width (320), hidden width of the GEGLU (ff.net.0.proj has 2 * inner outputs; % 64 == 0) */
  const uint16_t* x;             /* fp16 [M][C]: the LayerNorm's input and the residual */
  const float* gamma;            /* LayerNorm weight / bias [C], eps */
  const float* beta;
  float eps;
  tfmq_qsel aq0;                 /* activation quantizer of ff.net.0.proj (consumes the LayerNorm output) */
  const int8_t* w1;              /* ff.net.0.proj: tfmq_expand_w4 of the rows in ops.geglu_perm order ([64 value | 64 gate] per 128 rows) */
  const int32_t* wmeta1;         /* [2 * inner][4], wscale1 / bias1 [2 * inner], in the same row order */
  const float* wscale1;
  const float* bias1;            /* or NULL */
  tfmq_qsel aq2;                 /* activation quantizer of ff.net.2 (consumes value * gelu(gate)) */
  const int8_t* w2;              /* ff.net.2: tfmq_expand_w4, [C][inner] */
  const int32_t* wmeta2;
  const float* wscale2;
  const float* bias2;            /* or NULL */
  uint16_t* y;                   /* fp16 [M][C] output (oq.qtable == NULL) */
  tfmq_qsel oq;                  /* qtable != NULL: the output's only consumer is this activation quantizer: int8 bins to yq */
  int8_t* yq;
  float* ws;                     /* scratch, 4 * inner + 2560 floats: folded per-channel constants (written per call) */
  /* optional (ABI 8, round 4): a C -> C w4a8 Linear IN FRONT of the feed-forward, w0 != NULL -- attn2.to_out of the block
   * (`x = self.attn2(self.norm2(x), context=context) + x`, ldm/modules/attention.py:213, to_out :194): the rows x_pre = to_out(xq_pre) + bias0 +
   * res_pre are written to y_pre (fp16 [M][C]) and take the place of `x` (LayerNorm input and residual).  xq_pre: int8 bins [M][C] of
   * aq_pre (the cross attention kernel's output).  M % 256 == 0. */
  const int8_t* xq_pre;
  const int8_t* w0;
  const int32_t* wmeta0;
  const float* wscale0;
  const float* bias0;
  tfmq_qsel aq_pre;
  const uint16_t* res_pre;
  uint16_t* y_pre;
  /* optional: a C -> C w4a8 Linear BEHIND it, w3 != NULL -- the SpatialTransformer's proj_out (+ x_in, ldm/modules/attention.py:259-261): the
   * feed-forward's output is quantised with oq (proj_out's activation quantizer; yq / y unused) and y_post = proj_out(bins) + bias3 + res_post
   * is written as fp16 [M][C]; stats (optional): the consumer GroupNorm's per-segment {sum, sum of squares}, [M / stats_seg][C][2], as
   * tfmq_conv_desc.stats.  M % 256 == 0. */
  const int8_t* w3;
  const int32_t* wmeta3;
  const float* wscale3;
  const float* bias3;
  const uint16_t* res_post;
  uint16_t* y_post;
  float* stats;
  int32_t stats_seg;
} tfmq_ff_desc;
/* TFMQ_ERR_UNSUPPORTED for a token width other than 320 (callers keep the three-launch chain). */
int tfmq_ff_fused(tfmq_handle h, const tfmq_ff_desc* d, void* stream);

/* ---- K5c (round 4): chains of w4a8 token Linears (K = C = 320 or 640) around the attention of a BasicTransformerBlock as one launch, a token
 * per lane (the layout of tfmq_ff_fused): an input stage, up to three GEMMs over the resident token tile, optionally ONE LayerNorm +
 * quantise between a C-wide GEMM and its successor.  Replaces, bit for bit,
 *   in_mode 2: tfmq_groupnorm_from_stats (apply pass) -> tfmq_conv2d_w4a8 (proj_in, fp16 out) -> tfmq_layernorm_h -> tfmq_conv2d_w4a8
 *              (fused to_q | to_k | to_v: fp16 rows + transposed V)       SpatialTransformer.forward / BasicTransformerBlock._forward /
 *              CrossAttention.forward, ldm/modules/attention.py:238-261, :212, :168-177
 *   in_mode 0: tfmq_conv2d_w4a8 (to_out + residual, on the attention kernel's int8 output) -> tfmq_layernorm_h -> tfmq_conv2d_w4a8 (to_q)
 *              ldm/modules/attention.py:194, :212-213
 * under the quantised blocks of quant/quant_block.py:178-299 (QuantLayers: quant/quant_layer.py:306-340). */
typedef struct tfmq_chain_gemm {
  const int8_t* w;               /* tfmq_expand_w4 operand of the layer, [N][C] */
  const int32_t* wmeta;          /* [N][4] */
  const float* wscale;           /* [N] */
  const float* bias;             /* [N] or NULL */
  int32_t N;                     /* outputs, multiple of 64 */
  tfmq_qsel aq;                  /* activation quantizer of the layer's input (the bins resident in LDS) */
  const uint16_t* residual;      /* fp16 [M][N] added before the rounding to fp16, or NULL */
  uint16_t* y;                   /* fp16 rows [M][ldy]: output columns < t_col0 (all of them without yt) */
  int32_t ldy;
  uint16_t* yt;                  /* optional: columns >= t_col0 transposed, fp16 yt[b][n - t_col0][t] (b = token / T): the V^T operand of
                                    tfmq_attention_f16 */
  int32_t t_col0;                /* multiple of 64 */
  int32_t next;                  /* != 0 (N == C, not the last GEMM): LayerNorm(ln_gamma, ln_beta, ln_eps) of the output row + the next
                                    GEMM's quantizer produce the next GEMM's input */
} tfmq_chain_gemm;
typedef struct tfmq_chain_desc {
  int32_t M, C, T;               /* tokens (multiple of 81920 / C), token width (320 or 640), tokens per image */
  int32_t in_mode;               /* 0: x = int8 bins [M][C] of g[0].aq.  2: x = fp16 [M][C]; y = gn_a[b][c] * x + gn_b[b][c] (the GroupNorm's
                                    per-(image, channel) affine, e.g. from tfmq_gn_finalize), then g[0]'s quantizer; T % (81920 / C) == 0 */
  const void* x;
  const float* gn_a;             /* [M / T][C] */
  const float* gn_b;
  const float* ln_gamma;         /* the chain's LayerNorm (at most one) */
  const float* ln_beta;
  float ln_eps;
  int32_t n_gemm;                /* 1 .. 3 */
  tfmq_chain_gemm g[3];
  float* ws;                     /* scratch: 4 * (sum of N) * (C / 320) floats (per-column constants, rebuilt per call from the current Finite-Set row) */
} tfmq_chain_desc;
int tfmq_row_chain(tfmq_handle h, const tfmq_chain_desc* d, void* stream);

/* ---- K7: temporal-information block GEMVs (QuantTemporalInformationBlockDDIM.forward,
 * quant_block.py:52-64; ddim/models/diffusion.py:6-24,310-313) -------------------------- */
/* emb[m][dim] = [sin(t f_i), cos(t f_i)] (DDIM order, denominator half-1) or
 * [cos, sin] with denominator half (LDM order, ldm util.py:161-166) when ldm_order != 0 */
int tfmq_timestep_embedding(tfmq_handle h, const float* t, int m, int dim, int ldm_order, float* emb, void* stream);
/* y[m][n] = act_in(x[m][k]) @ W^T + b.  silu_in: apply x*sigmoid(x) first.  W fp32 [n][k]. */
int tfmq_linear_small_f32(tfmq_handle h, const float* x, const float* w, const float* bias, float* y, int m, int n,
                          int k, int silu_in, void* stream);
/* same with packed int4 weights (+ optional 8-bit quantisation of the SiLU'd input) */
int tfmq_linear_small_w4(tfmq_handle h, const float* x, const uint8_t* wpacked, const int32_t* wmeta,
                         const float* wscale, const float* bias, tfmq_qsel aq, float* y, int m, int n, int k,
                         int silu_in, void* stream);

/* ---- K8: GroupNorm (+SiLU) (+quantise) (Normalize+nonlinearity, ddim/models/diffusion.py:27-33,
 * 117-118,123-124; feeding QuantLayer's aqtizer, quant_layer.py:318-325) ------------------ */
typedef struct tfmq_gn_desc {
  int32_t B, HW, C1, C2;   /* input = channel concat of x1 [B][HW][C1] and x2 [B][HW][C2] (C2 may be 0):
                              torch.cat([h, hs.pop()], 1) of ddim/models/diffusion.py:341 never materialises */
  const float* x1;
  const float* x2;
  const float* gamma;      /* [C1+C2] */
  const float* beta;
  float eps;
  int32_t groups;          /* 32 */
  int32_t silu;            /* apply x*sigmoid(x) after the affine */
  tfmq_qsel aq;            /* qtable!=NULL: write int8 (bin-128) to yq; else write fp32 to yf */
  int8_t* yq;
  float* yf;
  float* xcat_or_null;     /* optional: also materialise the concat (input of the FP nin_shortcut / skip_connection) */
  int32_t half_out;        /* !=0: yf and xcat_or_null are fp16 buffers (consumers that round to fp16 anyway:
                              tfmq_conv2d_f16 with x_f16) */
  int32_t x_f16;           /* !=0: x1 / x2 are fp16 buffers (the fp16 activation stream: TFMQ_OUT_F16 outputs of the producing
                              convs, whose epilogues still emit the statistics from the fp32 values) */
} tfmq_gn_desc;
int tfmq_groupnorm(tfmq_handle h, const tfmq_gn_desc* d, void* stream);
/* same result when the producing conv(s) already emitted the statistics (tfmq_conv_desc.stats, segment size
 * `seg`): a tiny finalize kernel + ONE coalesced elementwise pass (4 B read, 1 B written per element).
 * stats2 pairs with d->x2 (channel concat).  ws: 2 * B * (C1+C2) floats. */
int tfmq_groupnorm_from_stats(tfmq_handle h, const tfmq_gn_desc* d, const float* stats1, const float* stats2, int seg,
                              float* ws, void* stream);
/* The GroupNorm statistics pass of tfmq_groupnorm_from_stats on its own: a[b][c] = rstd * gamma[c], bsh[b][c] = beta[c] - a * mean from the
 * producing conv's per-segment {sum, sum of squares} (tfmq_conv_desc.stats); a, bsh: [B][C1 + C2] floats. */
int tfmq_gn_finalize(tfmq_handle h, const tfmq_gn_desc* d, const float* stats1, const float* stats2, int seg, float* a, float* bsh, void* stream);

/* ---- K9: LayerNorm (+quantise) and GEGLU (+quantise) of the SpatialTransformer blocks
 * (nn.LayerNorm ldm/modules/attention.py:203-205 eps 1e-5; GEGLU :37-44 exact erf GELU) ------------ */
/* x: [rows][C] fp32; writes int8 (bin-128) to yq when aq.qtable != NULL and/or fp32 to yf */
int tfmq_layernorm(tfmq_handle h, const float* x, const float* gamma, const float* beta, float eps, long rows, int C,
                   tfmq_qsel aq, int8_t* yq, float* yf, void* stream);
/* the same on an fp16 input row (the fp16 activation stream), statistics and arithmetic in fp32 */
int tfmq_layernorm_h(tfmq_handle h, const uint16_t* x, const float* gamma, const float* beta, float eps, long rows, int C,
                     tfmq_qsel aq, int8_t* yq, float* yf, void* stream);
/* hin: [rows][2*inner] (output of ff.net.0.proj): y = hin[:, :inner] * gelu(hin[:, inner:]) */
int tfmq_geglu(tfmq_handle h, const float* hin, long rows, int inner, tfmq_qsel aq, int8_t* yq, float* yf, void* stream);

/* ---- K10: attention core on un-quantised q,k,v (QuantAttnBlock.forward quant_block.py:483-500:
 * bmm, *c^-1/2, softmax, bmm; attention quantizers are never enabled) ---- */
/* q,k,v: fp32, token t of batch b head hd at ptr[(b*T + t)*ld + hd*d ...]; out likewise (ldo).
 * If yq != NULL the result is also quantised (proj_out's aqtizer) to int8. */
int tfmq_attention(tfmq_handle h, const float* q, const float* k, const float* v, int ldq, int ldk, int ldv,
                   float* out, int ldo, int8_t* yq, tfmq_qsel aq, int B, int heads, int Tq, int Tk, int d,
                   float scale, void* stream);
/* Same operation on fp16 operands written by the projection GEMM (TFMQ_OUT_F16): q,k as above (ld in halves),
 * vt = V transposed, fp16 [B][heads*d][Tk_stride] (tfmq_conv_desc.yt).  Tk_stride >= Tk keys per batch item are
 * present in memory (k: [B][Tk_stride][ldk]); keys >= Tk are masked (a 77-token context is stored padded to 80).
 * d % 8 == 0, d <= 384 (above 256 a block computes the scores over the whole head and the 128 output channels of its slice;
 * larger heads: TFMQ_ERR_UNSUPPORTED, use tfmq_attention), Tk_stride % 8 == 0.
 * Which kernel serves a call follows from the shape AND from the outputs asked for: with yq only (out == NULL), d = 40, Tq % 128 == 0 and
 * Tk_stride <= 96 the all-heads context kernel (attention_ctx.hip) takes the launch; it sums the softmax denominator in another order than the
 * tiled kernel, so the int8 bins of the two agree within 1 on < 1 % of the outputs, not bit for bit (tests/test_attention_f16_gpu.py).  A
 * caller that needs the same bins with and without `out` sets TFMQ_ATTN_CTX=0. */
int tfmq_attention_f16(tfmq_handle h, const uint16_t* q, const uint16_t* k, const uint16_t* vt, int ldq, int ldk,
                       float* out, int ldo, int8_t* yq, tfmq_qsel aq, int B, int heads, int Tq, int Tk, int Tk_stride,
                       int d, float scale, void* stream);

/* ---- K10q8 (section 8f-3): the attention of a block whose matmul quantizers are LIVE (aqtizer_q / _k / _v / _w with use_aq:
 * cross_attn_forward quant_block.py:226-243, QuantAttnBlock.forward :483-500, QuantQKMatMul / QuantSMVMatMul :318-323, :350-351), both
 * products on the int8 matrix cores over the quantizers' BINS.  q / k: int8 bins - 128 of tfmq_quantize_act under aq_q / aq_k
 * ([B][Tq][ldq], [B][Tk][ldk], head h at channels h*d ..); vt: the bins - 128 of v under aq_v TRANSPOSED, [B][heads*d][Tk_stride]
 * (tfmq_transpose_i8; keys Tk .. Tk_stride zero).  sum (b_q - z_q)(b_k - z_k) and sum b_w (b_v - z_v) are exact int32 (zero points as
 * rank-one corrections); the softmax in fp32 over delta_q delta_k scale * (integer score), two passes over the keys (the bins need the
 * row's final normaliser); b_w = clamp(rint(p / delta_w), 0, w_level - 1) (the always-zero quantizer: zero point 0), w_level <= 256;
 * out = delta_w delta_v * (integer sum), fp32 [B][Tq][ldo].  d % 8 == 0, d <= 160, ldq / ldk / Tk_stride % 8 == 0, ldo % 4 == 0.
 * Against the fp32 products of the dequantised values (ops.attention_quant) only values on a rounding boundary of aq_w differ. */
int tfmq_attention_q8(tfmq_handle h, const int8_t* q, const int8_t* k, const int8_t* vt, int ldq, int ldk, tfmq_qsel aq_q, tfmq_qsel aq_k,
                      tfmq_qsel aq_v, tfmq_qsel aq_w, int w_level, float* out, int ldo, int B, int heads, int Tq, int Tk, int Tk_stride,
                      int d, float scale, void* stream);
/* y[b][c][t] = x[b][t][c] (int8), t < T; zero for T <= t < Tp: the V^T operand of tfmq_attention_q8 */
int tfmq_transpose_i8(tfmq_handle h, const int8_t* x, int8_t* y, int B, int T, int C, int Tp, void* stream);

/* ---- K15: exact-fp32 fused attention for the reconstruction of BasicTransformerBlock units (quant/reconstruction.py:
 * 86-209 on quant_block.py:248-299; the reference runs einsum / softmax / einsum under autograd).  fp32 operands on
 * the fp32 matrix cores, nothing of size Tq x Tk is written: the forward returns O and the per-row log-sum-exp in the
 * exp2 domain (lse[b][h][q] = m + log2(sum_k exp2(scale*log2(e)*s_qk - m))), the backward recomputes the probabilities.
 * q [B][Tq][ldq], k / v [B][Tk][ldk], head h at channels h*d..; d in {32, 40, 64, 80}; Tq a multiple of 32, any Tk.
 * Under tfmq_set_gemm_precision(h, 1) (bf16x3: the AdaRound iterations' default) forward and backward run the same decomposition with
 * every fp32 operand split hi + lo in bf16 and three v_mfma_f32_32x32x16_bf16 per product (fp32 accumulation, 2^-16 per product:
 * 1e-5 of float64 instead of 1e-6; 2.0 -> 0.83 ms forward, 6.4 -> 3.2 ms backward at T = 4096, d = 40, 64 (batch, head) pairs). */
int tfmq_attention_f32_fwd(tfmq_handle h, const float* q, const float* k, const float* v, int ldq, int ldk, float* out,
                           int ldo, float* lse, int B, int heads, int Tq, int Tk, int d, float scale, void* stream);
/* backward of the above: dq [B][Tq][ldq], dk / dv [B][Tk][ldk] from dout [B][Tq][ldo], the forward's out and lse.
 * dsum_ws: B*heads*Tq floats of scratch (row sums of dout o out).  Two kernels without atomics (key blocks -> dk, dv;
 * query blocks -> dq), deterministic. */
int tfmq_attention_f32_bwd(tfmq_handle h, const float* q, const float* k, const float* v, int ldq, int ldk, const float* out,
                           const float* dout, int ldo, const float* lse, float* dsum_ws, float* dq, float* dk, float* dv,
                           int B, int heads, int Tq, int Tk, int d, float scale, void* stream);

/* ---- K11: sampler elementwise (generalized_steps, ddim/functions/denoising.py:31-37) ----- */
/* coef: device [n_steps][4] = {sqrt(1-a_t), 1/sqrt(a_t)... see DESIGN.md}; step: device scalar.
 * x_next = sqrt(a_next)*x0 + c1*z + c2*eps with x0 = (x - eps*sqrt(1-a_t))/sqrt(a_t). */
int tfmq_ddim_update(tfmq_handle h, const float* x, const float* eps, const float* noise_or_null, float* x_next,
                     float* x0_or_null, size_t n, const float* coef, const int32_t* step, void* stream);
/* latent DDIM step with classifier-free guidance (p_sample_ddim, ldm/models/diffusion/ddim.py:181-211):
 * e = eps_u + scale*(eps_c - eps_u), then the DDIM update with the same coef row layout */
int tfmq_ddim_update_cfg(tfmq_handle h, const float* x, const float* eps_u, const float* eps_c, float scale,
                         const float* noise_or_null, float* x_next, float* x0_or_null, size_t n, const float* coef,
                         const int32_t* step, void* stream);
/* PLMS sampler pieces (ldm/models/diffusion/plms.py:179-242), in the reference's operation order:
 * cfg_combine: out = eps_u + scale*(eps_c - eps_u);
 * plms_combine: order 1 (e0+e1)/2 [e1 = eps at t_next]; 2 (3e0-e1)/2; 3 (23e0-16e1+5e2)/12; 4 (55e0-59e1+37e2-9e3)/24
 * with e1..e3 the previous model outputs, newest first; the result feeds tfmq_ddim_update. */
int tfmq_cfg_combine(tfmq_handle h, const float* eps_u, const float* eps_c, float scale, float* out, size_t n,
                     void* stream);
int tfmq_plms_combine(tfmq_handle h, int order, const float* e0, const float* e1, const float* e2_or_null,
                      const float* e3_or_null, float* out, size_t n, void* stream);
/* DPM-Solver++ multistep (order <= 2, data prediction; ldm/models/diffusion/dpm_solver/dpm_solver.py:386-399,504-549,
 * 755-810): x0 = (x - sigma*eps)/alpha;  x_t = c_x x - c_m m0 [- c_d (inv_r0 (m0 - m1))] */
int tfmq_dpm_x0(tfmq_handle h, const float* x, const float* eps, float sigma, float alpha, float* out, size_t n, void* stream);
int tfmq_dpm_update(tfmq_handle h, int order, const float* x, const float* m0, const float* m1_or_null, float c_x, float c_m,
                    float c_d, float inv_r0, float* out, size_t n, void* stream);
int tfmq_step_advance(tfmq_handle h, int32_t* step, int delta, void* stream);
/* np.histogram(a, bins) bin counts of x (optionally clipped in float64 to [clip_lo, clip_hi] first) on `bins` equal-width bins
 * given by the edge table edges[bins + 1] (device; float when f64 == 0, double otherwise): numpy's index arithmetic and edge
 * correction, values outside [edges[0], edges[bins]] dropped.  For Scaler.KL / Scaler.HIST (quant/quant_layer.py:67-133). */
int tfmq_np_histogram(tfmq_handle h, const float* x, size_t n, int f64, int do_clip, double clip_lo, double clip_hi,
                      const void* edges, int bins, uint32_t* counts, void* stream);
/* The same for `rows` tensors of n values at once (x [rows][n]; edges [rows][bins + 1]; clip_lo / clip_hi [rows] device doubles; counts
 * [rows][bins]): the per-output-channel loop of the KL / HIST weight scalers (quant/quant_layer.py:193-204) as 51 launches in all. */
int tfmq_np_histogram_rows(tfmq_handle h, const float* x, size_t rows, size_t n, int f64, int do_clip, const double* clip_lo,
                           const double* clip_hi, const void* edges, int bins, uint32_t* counts, void* stream);
/* Device self-test of instruction semantics the kernels rely on (v_cvt_pk_u8_f32 saturation to [0, 255], DPP lane
 * selection of the statistics sums).  0 = all hold; otherwise an error with *report = failing-check bit mask.  Synchronous,
 * allocates 4 bytes for the duration of the call; not for use under stream capture. */
int tfmq_hw_selftest(tfmq_handle h, uint32_t* report);
/* fp32 -> fp16 copy (round to nearest even): operand of tfmq_conv2d_f16 with x_f16 when the producer writes fp32 */
int tfmq_f32_to_f16(tfmq_handle h, const float* x, uint16_t* y, size_t n, void* stream);
/* y[b][t][c] = x[b][t][c] + r[b][c] (fp32 arithmetic; x / y fp16 when x_f16 != 0, else fp32; C % 8 == 0; y may alias x).  The residual
 * add of a cross attention whose context is ONE token (class-conditional LDM: softmax over one key is exactly 1, so the attention
 * output of every query is v of that token and to_out's result one row per batch item -- ldm/modules/attention.py:168-194). */
int tfmq_row_broadcast_add(tfmq_handle h, const void* x, const float* r, int B, long T, int C, int x_f16, void* y, void* stream);
/* y = x*sigmoid(x)  (nonlinearity, ddim/models/diffusion.py:27-29) */
int tfmq_silu(tfmq_handle h, const float* x, float* y, size_t n, void* stream);
int tfmq_nchw_to_nhwc(tfmq_handle h, const float* x, float* y, int B, int C, int HW, void* stream);
int tfmq_nhwc_to_nchw(tfmq_handle h, const float* x, float* y, int B, int C, int HW, void* stream);

/* ---- K12-K14: AdaRound (adaptive_rounding.py; reconstruction_util.py:50-91; Adam) --------- */
/* alpha = -log(1.2/(w/delta - floor(w/delta) + 0.1) - 1)   (init_alpha :31-38); delta per row */
int tfmq_adaround_init(tfmq_handle h, const float* w, const float* delta, float* alpha, size_t rows, size_t cols,
                       void* stream);
/* soft forward (:40-41,51,59-60,67-69): w_hat = delta*(clamp(floor(w/delta)+h(alpha)+zp,0,L-1)-zp);
 * hard != 0 uses the inference-time rounding h = [alpha >= 0] (:63) instead of the soft target */
int tfmq_adaround_soft_fwd(tfmq_handle h, const float* w, const float* alpha, const float* delta, const float* zp,
                           float* w_hat, size_t rows, size_t cols, int level, int hard, void* stream);
/* backward of the soft forward + rounding regulariser, fused with one Adam step
 * (torch.optim.Adam defaults lr=1e-3, betas .9/.999, eps 1e-8; reconstruction.py:42):
 *   g = g_what * delta * [0<floor+h+zp<L-1] * h'(alpha) + w_reg * d/dalpha (1-|2h-1|^b)
 * g_what: dL/dw_hat from the block backward; b_temp<=0 disables the regulariser (warm-up).
 * round_loss_or_null accumulates w_reg * sum(1-|2h-1|^b) (one float, atomically). */
int tfmq_adaround_bwd_adam(tfmq_handle h, const float* w, float* alpha, const float* delta, const float* zp,
                           const float* g_what, float* m, float* v, size_t rows, size_t cols, int level, float w_reg,
                           float b_temp, float lr, int t, float* round_loss_or_null, void* stream);
/* Round 5 (ABI 9): the same launch with its four per-iteration scalars in DEVICE memory -- scalars_dev[4] = what tfmq_adaround_scalars
 * computes for (w_reg, b_temp, lr, t) -- so that a whole reconstruction iteration (quant/reconstruction.py:63-78,182-198: forward, loss,
 * backward, optimizer step) can be captured once as a hipGraph and replayed with a 16-byte copy in front of every replay.  Bit-identical to
 * tfmq_adaround_bwd_adam.  tfmq_adaround_scalars is a host function (no handle, no launch). */
int tfmq_adaround_scalars(float w_reg, float b_temp, float lr, int t, float* out4);
int tfmq_adaround_bwd_adam_dyn(tfmq_handle h, const float* w, float* alpha, const float* delta, const float* zp,
                               const float* g_what, float* m, float* v, size_t rows, size_t cols, int level,
                               const float* scalars_dev, float* round_loss_or_null, void* stream);
/* rec = mean over all-but-dim1 of sum_dim1 |pred-tgt|^2 for NHWC tensors == sum(|d|^2)/(n/C)
 * (lp_loss, quant_layer.py:152-153); also writes g = dL/dpred.  loss: one device float. */
int tfmq_recon_loss(tfmq_handle h, const float* pred, const float* tgt, float* g_or_null, size_t n, size_t denom,
                    float* loss, void* stream);

/* ---- K15: block-reconstruction forward/backward pieces (replace autograd through `block(*cur_inputs)`,
 * quant/reconstruction.py:69-71,190-192,295-297).  Exact fp32: the soft AdaRound targets are
 * non-integer, so this path is floating point by construction. ------------------------------ */
/* Operand precision of the matrix-core path of tfmq_gemm_f32 / _heads on this handle, until changed: 0 = exact fp32 products
 * (v_mfma_f32_32x32x2f32; default, what every exactness-critical caller relies on), 1 = "bf16x3" (each fp32 operand value split into
 * bf16 hi + lo, three bf16 MFMAs per product: relative error 2^-16 per product, fp32 accumulation), 2 = fp16 operands (2^-11).
 * Meant for the AdaRound reconstruction iterations only (reference quant/reconstruction.py:63-78,182-198 runs them in fp32 autograd;
 * SURVEY section 7-1); small problems on the FMA tile stay exact.
 * Intended use: a property of the handle, set ONCE after tfmq_create -- a host that wants both precisions keeps one handle per precision
 * (handles are independent: own workspaces, own error state) and picks the handle per launch, as the Python host does
 * (_lib.handle(device, precision), ops.gemm_precision).  Nothing in the library changes it behind the caller's back. */
int tfmq_set_gemm_precision(tfmq_handle h, int mode);
/* batched strided GEMM: C[z] (M x N, row stride scm) = alpha * A[z] B[z] (+bias[n]) (+rowadd) (+residual), or C += ...
 * A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; rowadd[(m / rows_per_img)*rowadd_ld + n] */
int tfmq_gemm_f32(tfmq_handle h, const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak,
                  long sbk, long sbn, long scm, int batch, long bsa, long bsb, long bsc, float alpha, const float* bias,
                  const float* rowadd, int rows_per_img, int rowadd_ld, const float* residual, int accumulate,
                  void* stream);
/* the same product over a two-level batch: item (b, hd), b < batch, hd < heads, at offsets b * bs? + hd * hs? -- all heads
 * of a multi-head attention product (quant_block.py:285-299 on packed [B,T,heads*d] tensors) in one launch */
int tfmq_gemm_f32_heads(tfmq_handle h, const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak,
                        long sbk, long sbn, long scm, int batch, long bsa, long bsb, long bsc, int heads, long hsa, long hsb,
                        long hsc, float alpha, int accumulate, void* stream);
/* LayerNorm / GEGLU backward w.r.t. their inputs (BasicTransformerBlock units, ldm/modules/attention.py:196-215):
 * gx = d LN(x; gamma)/dx applied to gy, rows tokens of C channels;  dh = d(h[:, :I] * gelu(h[:, I:]))/dh applied to dy */
int tfmq_layernorm_bwd(tfmq_handle h, const float* x, const float* gy, const float* gamma, float eps, long rows, int C,
                       float* gx, void* stream);
int tfmq_geglu_bwd(tfmq_handle h, const float* hin, const float* dy, long rows, int inner, float* dh, void* stream);
/* col[(b,ho,wo)][(kh,kw,c)] <- x NHWC (zero padded); col2im is its adjoint (gather form, deterministic) */
int tfmq_im2col(tfmq_handle h, const float* x, float* col, int B, int H, int W, int C, int KH, int KW, int stride,
                int pad_t, int pad_l, int Ho, int Wo, void* stream);
int tfmq_col2im(tfmq_handle h, const float* dcol, float* dx, int B, int H, int W, int C, int KH, int KW, int stride,
                int pad_t, int pad_l, int Ho, int Wo, void* stream);
/* fp16 im2col rows for a narrow-input 3x3 (or kh x kw) stride-1 convolution run as a pointwise GEMM: col[m][k] =
 * fp16(x[b][y + k/C/kw - pad_t][x + (k/C)%kw - pad_l][k%C]) for k < kh*kw*C (zero outside the image), 0 for k in
 * [kh*kw*C, kp); kp % 8 == 0, kp >= kh*kw*C.  The first conv of the UNets (`conv_in` ddim/models/diffusion.py:310,
 * `input_blocks.0.0` openaimodel.py:502-506: 3 or 4 input channels) then takes the register-direct fp16 pointwise kernel with
 * K = kp instead of nine K-steps of 4 live channels out of 32. */
int tfmq_im2col_f16(tfmq_handle h, const float* x, uint16_t* col, int B, int H, int W, int C, int KH, int KW, int pad_t,
                    int pad_l, int kp, void* stream);
/* The other end: a conv with a handful of output channels (`conv_out` ddim/models/diffusion.py:353, `out.2` openaimodel.py:700-704:
 * 320 -> 4) computed as ONE pointwise GEMM to kh*kw*cout per-tap partial sums, y9[m][tap*cout + co] = sum_ci x[m][ci] w[co][tap][ci]
 * (fp32), and this gather: out[b][y][x][co] = bias[co] + sum_tap y9[b][y + tap/kw - pad_t][x + tap%kw - pad_l][tap*cout + co], taps
 * in ascending order, positions outside the image skipped.  The input is read once instead of once per tap. */
int tfmq_tap_gather_sum(tfmq_handle h, const float* y9, int B, int H, int W, int KH, int KW, int cout, int ld, int pad_t, int pad_l,
                        const float* bias, float* out, void* stream);
/* weights OIHW <-> [cout][(kh,kw,cin)] (dir 0: to the GEMM layout, 1: back) */
int tfmq_w_relayout(tfmq_handle h, const float* src, float* dst, int cout, int cin, int kh, int kw, int dir, void* stream);
int tfmq_silu_bwd(tfmq_handle h, const float* x, const float* gy, float* gx, size_t n, void* stream);
/* backward of GroupNorm(+SiLU) w.r.t. its input (x: the un-normalised NHWC input, gy: dL/d output) */
int tfmq_groupnorm_bwd(tfmq_handle h, const float* x, const float* gy, const float* gamma, const float* beta, float* gx,
                       int B, int HW, int C, int groups, float eps, int silu, void* stream);
/* P = softmax(scale*S) per row; dS = scale * P * (dP - sum(dP*P)) */
int tfmq_softmax_rows(tfmq_handle h, const float* S, float* P, long rows, int cols, float scale, void* stream);
int tfmq_softmax_bwd_rows(tfmq_handle h, const float* P, const float* dP, float* dS, long rows, int cols, float scale,
                          void* stream);
/* nearest-neighbour x2 (F.interpolate(scale_factor=2, mode="nearest"), ddim/models/diffusion.py:47-48) */
int tfmq_upsample2x(tfmq_handle h, const float* x, float* y, int B, int H, int W, int C, void* stream);
/* backward of tfmq_upsample2x: gx[b][h][w][c] = sum of the four g[b][2h+i][2w+j][c]  (autograd through Upsample.forward's
 * F.interpolate in the FP tail of GetLayerGrad, quant/data_utill.py:191-256) */
int tfmq_upsample2x_bwd(tfmq_handle h, const float* g, float* gx, int B, int H, int W, int C, void* stream);
/* GetLayerGrad's loss (reference quant/data_utill.py:246-247): F.kl_div(F.log_softmax(out_q, dim=1), F.softmax(out_fp, dim=1),
 * reduction='batchmean') on NHWC rows [n_rows][C] (channel softmax, C <= 64).
 *   wrt_target = 0: g = dL/d out_q  = (softmax(out_q) - softmax(out_fp)) / batch
 *   wrt_target = 1: g = dL/d out_fp = p (l - sum_c p l) / batch with p = softmax(out_fp), l = log p - log_softmax(out_q): the
 *                   reference does not detach the target, and GradSaverHook (:170-188) keeps the gradient of the pass the backward
 *                   reaches last, which is the FP pass -- so THIS is what save_grad caches (tests/test_fisher_gpu.py).
 * loss_or_null accumulates the loss value (one device float, atomically). */
int tfmq_kl_softmax_grad(tfmq_handle h, const float* out_q, const float* out_fp, float* g, long n_rows, int C, int batch,
                         int wrt_target, float* loss_or_null, void* stream);
/* LossFunc's Fisher-weighted reconstruction terms (reference quant/reconstruction_util.py:53-59), fgrad = save_grad's cached
 * |dL/d out| + 1 of the mini-batch (quant/data_utill.py:54-73), tensors [n_samples][per_sample]:
 *   mode 1 (RLOSS.FISHER_DIAG): rec = ((pred - tgt)^2 * fgrad^2).sum(1).mean() = sum(...) / denom, denom = numel / size(1)
 *   mode 2 (RLOSS.FISHER_FULL): a = |pred - tgt|, w = |fgrad|, s_b = sum_sample(a w); rec = (s_b a w).mean() / 100 (denom unused)
 * g_or_null = d rec / d pred; loss: one device float, accumulated; dot_scratch: n_samples doubles (mode 2). */
int tfmq_fisher_loss(tfmq_handle h, const float* pred, const float* tgt, const float* fgrad, float* g_or_null, size_t n_samples,
                     size_t per_sample, int mode, size_t denom, double* dot_scratch, float* loss, void* stream);
/* y += a*x */
int tfmq_axpy(tfmq_handle h, float* y, const float* x, float a, size_t n, void* stream);

/* ---- K16: the exchange step of the sharded calibration: RCCL over xGMI, one rank per GPU.
 * Replaces linklink.allreduce / dist_helper.allaverage (linklink/__init__.py:6-13, linklink/dist_helper.py:33-36) at
 * their call sites: the SUM of a reconstruction unit's gradients every Adam iteration (quant/reconstruction.py:72-75,
 * 193-195,298-300) and the all-average of the activation deltas (quant/quant_model.py:127-132). --------------------- */
#define TFMQ_COMM_ID_BYTES 128
/* rank 0 draws the rendezvous id (host bytes) and hands it to every rank by any side channel (the Python mirror
 * broadcasts it over the torch.distributed store the reference's init_process_group already creates) */
int tfmq_comm_unique_id(uint8_t* id_host /* [TFMQ_COMM_ID_BYTES] */);
/* collective: every rank calls it with the same id; binds a communicator of `world` ranks to the handle's device.
 * TFMQ_ERR_UNSUPPORTED when librccl cannot be loaded. */
int tfmq_comm_init(tfmq_handle h, const uint8_t* id_host, int rank, int world);
/* rank / world of the handle's communicator (world = 0: none) */
int tfmq_comm_info(tfmq_handle h, int* rank, int* world);
/* in-place SUM over all ranks of n fp32 values at device pointer buf, enqueued on `stream` (ordered with the kernels
 * launched on it before and after; no host synchronisation) */
int tfmq_allreduce_sum_f32(tfmq_handle h, float* buf, size_t n, void* stream);
int tfmq_comm_destroy(tfmq_handle h);

/* ---- stream-capture helpers: a sampler step is captured once into a hipGraph and replayed */
int tfmq_graph_begin(tfmq_handle h, void* stream);
int tfmq_graph_end(tfmq_handle h, void* stream, int* graph_id);
int tfmq_graph_launch(tfmq_handle h, int graph_id, void* stream);
int tfmq_graph_destroy(tfmq_handle h, int graph_id);
/* HIP-event timing on a given stream (bench.py roofline leg) */
int tfmq_event_create(tfmq_handle h, int* event_id);
int tfmq_event_record(tfmq_handle h, int event_id, void* stream);
int tfmq_event_elapsed_ms(tfmq_handle h, int start_id, int stop_id, float* ms);
int tfmq_stream_sync(tfmq_handle h, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TFMQ_HIP_H */
