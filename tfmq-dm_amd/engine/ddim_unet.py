"""DDPM/DDIM pixel-space UNet lowered to the HIP kernels.

Mirrors the dataflow of the reference's `Model.forward` (ddim/models/diffusion.py:306-354)
with its quantised blocks (`QuantResnetBlock` quant/quant_block.py:415-444, `QuantAttnBlock`
:474-505) but never runs a torch op on the data path: every step is a launch through the C ABI
(include/tfmq_hip.h).  Fusions:
  GroupNorm -> SiLU -> 8-bit quantise   one kernel, reading the *virtual* concat [h, skip]
  conv + bias + temb-projection row + residual      implicit-GEMM epilogue
  q/k/v 1x1 convs                                   one concatenated-N GEMM (sibling quantizers equal)
  softmax(QK^T c^-1/2)V -> quantise                 one flash-attention kernel
  nearest-2x upsample                               address arithmetic of the consumer conv
  TIB (time-embedding MLP + 22 projections)         depends only on t: a per-step table

Layer modes (QuantLayer.forward, quant/quant_layer.py:306-340):
  'fp'    original fp32 weights        -> f16 MFMA, fp32 accumulate
  'w4'    4-bit weights, fp activations (disable_aq layers / asymmetric reconstruction inputs)
          -> f16 MFMA on the exact integer grid (q - z) with a per-channel output scale
  'w4a8'  4-bit weights, 8-bit activations -> int8 MFMA, exact int32 accumulation
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence

import torch

from .. import ops
from .._lib import TfmqError


class LayerQ:
    """Quantisation state of one QuantLayer handed to the engine."""

    def __init__(self, delta: torch.Tensor, zp: torch.Tensor, alpha: Optional[torch.Tensor] = None,
                 qid: Optional[int] = None, level: int = 16, act_level: int = 256):
        self.delta, self.zp, self.alpha, self.qid = delta, zp, alpha, qid
        self.level, self.act_level = int(level), int(act_level)      # 2 ** bits of the weight / activation quantizer


TAPE = None       # engine/fisher.py: GradTape of the running forward (Fisher-weighted reconstruction), else None


def _tape():
    return TAPE if (TAPE is not None and TAPE.recording) else None


class _Layer:
    def __init__(self, kind, packed, aq, w32=None, wide=False):
        self.kind, self.p, self.aq = kind, packed, aq
        # wide: a quantised-weight x 8-bit-activation layer whose weights have MORE than 16 levels (the --wq 8 recipes).  (q_w - z_w)
        # then needs 9 bits and does not fit the int8 MFMA operand; the layer keeps kind "w4a8" (= consumes activation bins) but its
        # packed weights are the fp16 integer grid and it runs the fp16-operand kernels on exact integers (see _run_wide).
        self.wide = wide
        # TFMQ_EXACT_FP=1 (parity diagnostics): fp32 weights [cout, kh*kw*cin] in the im2col column order.  An un-quantised /
        # weight-only layer then runs as im2col + the exact-fp32 MFMA GEMM (the reference's fp32 F.conv2d / F.linear up to the
        # summation order) instead of the fp16-operand kernels.
        self.w32 = w32

    def _run_exact(self, x, stride=1, pad=(0, 0, 0, 0), up2x=False, rowadd=None, residual=None, out=None, y_coff=0,
                   rowadd_ld=None, rowadd_step=None, rowadd_step_stride=0, x2=None, **_):
        if out is not None or x2 is not None or x.dtype != torch.float32:
            raise TfmqError("exact-fp32 layer: fp32 NHWC input, own output buffer, no virtual concat")
        x_in = x
        if up2x:
            x = ops.upsample2x(x)
        B, H, W, cin = x.shape
        kh, kw, cout = self.p.kh, self.p.kw, self.p.cout
        Ho, Wo = ops.out_hw(H, W, kh, kw, stride, pad[0], pad[1], pad[2], pad[3])
        if kh == 1 and kw == 1 and stride == 1 and tuple(pad) == (0, 0, 0, 0):
            col = x.reshape(B * H * W, cin)
        else:
            col = ops.im2col(x, kh, kw, stride, pad)
        ra = None
        if rowadd is not None:
            if rowadd_step is not None:      # per-step TIB table row (host read of the step counter: diagnostics mode only)
                k = int(rowadd_step.item())
                # `rowadd` is the view table[0, o:] of the [steps][sum Cout] TIB table: row k of this layer starts k * stride elements further
                # in the STORAGE (slicing the one-row view past its end gave an empty tensor for every k >= 1 -- found by fixture F27, round 5)
                ra = torch.as_strided(rowadd, (1, cout), (cout, 1), rowadd.storage_offset() + k * rowadd_step_stride).expand(B, cout).contiguous()
            else:
                ra = rowadd.reshape(-1, cout).contiguous()
                if ra.shape[0] == 1 and B > 1:
                    ra = ra.expand(B, cout).contiguous()
        res = None if residual is None else residual.reshape(B * Ho * Wo, cout)
        if res is not None and res.dtype != torch.float32:
            raise TfmqError("exact-fp32 layer: fp32 residual expected (fp32 stream)")
        y = ops.gemm(col, self.w32, trans_b=True, bias=self.p.bias, rowadd=ra, rows_per_img=Ho * Wo, residual=res)
        y = y.reshape(B, Ho, Wo, cout)
        tp = _tape()
        if tp is not None and tp.depends(x_in, residual):
            w32, pointwise = self.w32, col.data_ptr() == x.data_ptr()

            def bwd(gouts):
                g = gouts[0].reshape(B * Ho * Wo, cout)
                gx = None
                if tp.depends(x_in):
                    dcol = ops.gemm(g, w32)                                  # dL/d(im2col rows) = g W
                    gx = dcol.reshape(B, H, W, cin) if pointwise else ops.col2im(dcol, (B, H, W, cin), kh, kw, stride, pad)
                    if up2x:
                        gx = ops.upsample2x_bwd(gx)
                return [gx, gouts[0] if residual is not None else None]
            tp.rec([x_in, residual], [y], bwd)
        return y

    def _run_wide(self, xq, **kw):
        """W8A8: (b - z_a) and (q_w - z_w) are integers of magnitude <= 255, exact in fp16; their products are exact in the fp32
        accumulator of the fp16 MFMA and the sums round like any fp32 accumulation -- at least as exact as the reference, which runs an
        fp32 conv on the DEquantised values.  Output scale delta_a(step) * delta_w[c]."""
        for k in ("out_q8", "geglu_oq", "t_col0"):
            if kw.get(k) is not None:
                raise TfmqError(f"W8A8 layer: the fused int8-path epilogue '{k}' is not available")
        half = ops.f16_dma_ok(self.p.cin, self.p.kh, self.p.kw)
        xg = ops.bins_to_grid(xq, self.aq, half=half)
        pf = ops.PackedF16(self.p.w16, self.p.bias, self.p.cout, self.p.cin, self.p.kh, self.p.kw, wscale=ops.scale_by_qdelta(self.p.wscale, self.aq))
        return ops.conv2d_f16(xg, pf, **kw)

    def run(self, x, **kw):
        kw.setdefault("want_stats", True)   # conv-epilogue GroupNorm statistics (K8 split form)
        if self.kind == "w4a8":
            if self.wide:
                return self._run_wide(x, **kw)
            return ops.conv2d_w4a8(x, self.p, self.aq, **kw)
        if self.w32 is not None:
            return self._run_exact(x, **kw)
        if (x.dtype == torch.float32 and kw.get("out_f16") and self.p.kh * self.p.kw > 1 and self.p.kh * self.p.kw * self.p.cin <= 64
                and kw.get("pad") == (self.p.kh // 2, self.p.kw // 2, self.p.kh // 2, self.p.kw // 2) and kw.get("stride", 1) == 1
                and not kw.get("up2x") and kw.get("rowadd") is None and kw.get("out") is None
                and os.environ.get("TFMQ_NARROW_CONV_GEMM", "1") != "0"):
            # the UNet's first conv (3 / 4 input channels) in the fp16 stream: fp16 im2col rows (kh*kw*cin values padded to 32 / 64)
            # + the register-direct pointwise kernel, instead of nine K-steps with 4 live channels of 32 on the tile kernel
            g = getattr(self, "_gemm", None)
            if g is None:
                g = self._gemm = ops.narrow_conv_as_gemm(self.p) or False
            if g:
                col = ops.im2col_f16(x, self.p.kh, self.p.kw, self.p.kh // 2, self.p.kw // 2, g.cin)
                kw2 = {k: v for k, v in kw.items() if k not in ("pad", "stride", "up2x")}
                return ops.conv2d_f16(col, g, **kw2)
        if (x.dtype == torch.float16 and self.p.cout <= 4 and self.p.kh * self.p.kw > 1 and not kw.get("out_f16")
                and kw.get("pad") == (self.p.kh // 2, self.p.kw // 2, self.p.kh // 2, self.p.kw // 2) and kw.get("stride", 1) == 1
                and not kw.get("up2x") and kw.get("rowadd") is None and kw.get("residual") is None and kw.get("out") is None
                and os.environ.get("TFMQ_NARROW_CONV_GEMM", "1") != "0"):
            # the UNet's last conv (a few output channels): one pointwise GEMM to the per-tap partial sums + a gather over the taps;
            # the tile kernel re-reads the input per tap (PMC: 4x its bytes from HBM)
            g = getattr(self, "_gemm_out", None)
            if g is None:
                g = self._gemm_out = ops.narrow_out_conv_as_gemm(self.p) or False
            if g:
                y9 = ops.conv2d_f16(x, g, want_stats=False)
                return ops.tap_gather_sum(y9, self.p.kh, self.p.kw, self.p.cout, self.p.kh // 2, self.p.kw // 2, self.p.bias)
        if x.dtype == torch.float32 and ops.f16_dma_ok(self.p.cin, self.p.kh, self.p.kw):
            # un-quantised / weight-only layers round their input to fp16 while staging anyway: one conversion pass and
            # the LDS-DMA pipeline beat the register-staged fp32-input kernel ~3x (FP / weight-only state: the
            # calibration data passes and the FP sampling of the calibration set); bit-identical
            x = ops.to_half(x)
        return ops.conv2d_f16(x, self.p, **kw)


def _check_w4a8_levels(name: str, q: LayerQ) -> None:
    """The int8-MFMA path stores weights as nibbles (<= 16 levels) and activations as 256 bins; weights with up to 2048 levels (the
    reference's --wq 8 recipes: 256) run the fp16 integer-grid path (_Layer._run_wide).  Anything else must not be run silently on
    either: the deltas would have been searched for that width."""
    if q.act_level != 256 or not 2 <= q.level <= 2048:
        raise TfmqError(f"{name}: the HIP engine's quantised layers are 2..2048-level weights x 8-bit activations; got "
                        f"{q.level} weight levels / {q.act_level} activation levels")


def ddim_resblock_names(cfg) -> List[str]:
    nres, nlev = cfg["num_res_blocks"], len(cfg["ch_mult"])
    names = []
    for i in range(nlev):
        names += [f"down.{i}.block.{j}" for j in range(nres)]
    names += ["mid.block_1", "mid.block_2"]
    for i in range(nlev):
        names += [f"up.{i}.block.{j}" for j in range(nres + 1)]
    return names


class UnitReached(Exception):
    """Raised by a StopAt tap dictionary once the requested unit's tensors are recorded."""


class StopAt(dict):
    """taps={} argument of forward() that ends the forward at unit `name` (the reference's DataSaverHook +
    StopForwardException, quant/data_utill.py:76-111): what lies downstream of the unit is never launched."""

    def __init__(self, name: str):
        super().__init__()
        self.stop = name

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        if k == self.stop:
            raise UnitReached(k)


class DdimUNetEngine:
    def __init__(self, sd: Dict[str, torch.Tensor], cfg: dict, device="cuda:0"):
        self.cfg = dict(cfg)
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise TfmqError("DdimUNetEngine needs an MI355X device: the HIP kernels are the only implementation")
        self.sd = {k: v.detach().to(self.dev, torch.float32).contiguous() for k, v in sd.items()}
        self.res_names = ddim_resblock_names(cfg) if "ch_mult" in self.cfg else []
        self.calib_mask = None     # None = observe every live quantizer, else a set of qids
        self.layers: Dict[str, _Layer] = {}
        self.lin: Dict[str, tuple] = {}
        self.qtable = None
        self.step = None
        self.tib_table = None
        self.tib_off: Dict[str, int] = {}
        self.prepared = False
        # activation calibration (Finite-Set Calibration, quant/calibration.py:108-152)
        self.calib = None          # None | ("init", k) | ("running", k)
        # fp16 activation stream: outside calibration / tap capture the tensors that travel between
        # blocks (conv outputs feeding a GroupNorm / LayerNorm / residual add) are stored as fp16 -- their GroupNorm
        # statistics still come from the fp32 values inside the producing epilogue.  TFMQ_STREAM_F32=1 keeps fp32.
        # TFMQ_EXACT_FP=1: parity diagnostics -- every un-quantised / weight-only conv and Linear as an exact-fp32 GEMM, every
        # attention as exact-fp32 matmuls + row softmax, fp32 activation stream.  What remains against the reference's fp32
        # path is summation order and the hardware exp2 / rcp of SiLU / GELU (a few 1e-7): tests/test_exact_fp_mode_gpu.py
        # shows the activation-bin flips of the fast mode collapse in this one, i.e. they are fp16 operand rounding.
        self.exact_fp = os.environ.get("TFMQ_EXACT_FP", "0") not in ("", "0")
        self.stream_f16 = os.environ.get("TFMQ_STREAM_F32") is None and not self.exact_fp
        self._h16 = False
        self.act_state = None      # [n_q, 2] EMA {x_min, x_max} (quant_layer.py:229-244)
        self._qp_scratch = None

    # ------------------------------------------------------------------ weights
    def _w32(self, w: torch.Tensor, pf):
        """fp32 GEMM operand [cout, kh*kw*cin] of an un-quantised (pf None) or weight-only layer for the exact mode; None otherwise.
        Weight-only: (q - z) * delta in fp32 -- the integer grid the fp16 pack holds exactly, times the channel scale, which is
        what the reference's fake-quantised fp32 weight is (quant_layer.py:225-227)."""
        if not self.exact_fp:
            return None
        cout, cin, kh, kw = w.shape
        if pf is None:
            return ops.w_relayout(w.contiguous(), cout, cin, kh, kw, to_gemm=True)
        grid = pf.w16[:, :, :cin].float() * pf.wscale.reshape(cout, 1, 1)          # [cout][tap][cin]
        return grid.reshape(cout, kh * kw * cin).contiguous()

    def _conv_names(self):
        return [k[:-7] for k in self.sd if k.endswith(".weight") and self.sd[k].dim() == 4]

    def prepare(self, wq: Optional[Dict[str, LayerQ]] = None, qtable: Optional[torch.Tensor] = None,
                step: Optional[torch.Tensor] = None, attn_q: Optional[Dict[str, dict]] = None):
        """wq: name -> LayerQ for every weight-quantised layer (absent => FP layer).  A layer with
        LayerQ.qid != None and a qtable runs w4a8; qtable: fp32 [n_steps, n_q, 2] on the device."""
        wq = wq or {}
        self.qtable = None if qtable is None else qtable.to(self.dev, torch.float32).contiguous()
        self.step = step
        # attn_q (off unless a block's `use_aq` was switched on by hand): attention path -> {"q", "k", "v", "w": qid in
        # the activation table, "w_level": levels of the softmax quantizer}; that attention then runs ops.attention_quant
        self.attn_q = dict(attn_q or {})
        if self.attn_q and self.qtable is None:
            raise TfmqError("prepare: attention quantizers need the activation table")
        self.layers.clear()
        self.lin.clear()
        sd = self.sd

        def aq_of(q: Optional[LayerQ]):
            if q is None or q.qid is None or self.qtable is None:
                return None
            return ops.qsel(self.qtable, q.qid, self.step)

        for n in self._conv_names():
            w, b = sd[n + ".weight"], sd.get(n + ".bias")
            q = wq.get(n)
            aq = aq_of(q)
            if q is None:
                self.layers[n] = _Layer("fp", ops.pack_w_f16(w, b), None, self._w32(w, None))
            elif aq is None:
                # weight-only layers hold the exact integer grid q - z in f16: any bit width up to 11 bits
                a = None if q.alpha is None else q.alpha.to(self.dev).contiguous()
                pf = ops.pack_w_f16(w, b, q.delta.to(self.dev), q.zp.to(self.dev), a, level=q.level)
                self.layers[n] = _Layer("w4", pf, None, self._w32(w, pf))
            else:
                _check_w4a8_levels(n, q)
                a = None if q.alpha is None else q.alpha.to(self.dev).contiguous()
                if q.level > 16:
                    self.layers[n] = _Layer("w4a8", ops.pack_w_f16(w, b, q.delta.to(self.dev), q.zp.to(self.dev), a, level=q.level), aq, wide=True)
                else:
                    self.layers[n] = _Layer("w4a8", ops.pack_w4(w, q.delta.to(self.dev), q.zp.to(self.dev), a, b), aq)
        # linears of the temporal-information block
        for n in [k[:-7] for k in sd if k.endswith(".weight") and sd[k].dim() == 2]:
            w, b = sd[n + ".weight"], sd.get(n + ".bias")
            q = wq.get(n)
            if q is None:
                self.lin[n] = ("fp", w, b, None)
            else:
                _check_w4a8_levels(n, q)
                a = None if q.alpha is None else q.alpha.to(self.dev).contiguous()
                if q.level > 16:       # wide weights: dequantised fp32 weights (q - z) delta, fp32 GEMV; activations fake-quantised first
                    pf = ops.pack_w_f16(w.reshape(w.shape[0], w.shape[1], 1, 1).contiguous(), b, q.delta.to(self.dev), q.zp.to(self.dev), a, level=q.level)
                    w32 = (pf.w16[:, 0, :w.shape[1]].float() * pf.wscale.reshape(-1, 1)).contiguous()
                    self.lin[n] = ("w8", w32, b, aq_of(q))
                else:
                    self.lin[n] = ("w4", ops.pack_w4(w, q.delta.to(self.dev), q.zp.to(self.dev), a, b), None, aq_of(q))
        # fused q/k/v GEMM where the three sibling quantizers agree at every step
        self.fused_qkv: Dict[str, _Layer] = {}
        for n in list(self.layers):
            if not n.endswith(".q"):
                continue
            p = n[:-2]
            ls = [self.layers[p + s] for s in (".q", ".k", ".v")]
            if all(l.kind == "w4a8" and not l.wide for l in ls):
                ids = [wq[p + s].qid for s in (".q", ".k", ".v")]
                same = all(bool(torch.equal(self.qtable[:, ids[0]], self.qtable[:, i])) for i in ids[1:])
                if same:
                    pk = ops.PackedW4(torch.cat([l.p.packed for l in ls]), torch.cat([l.p.wmeta for l in ls]),
                                      torch.cat([l.p.wscale for l in ls]), torch.cat([l.p.bias for l in ls]),
                                      3 * ls[0].p.cout, ls[0].p.cin, 1, 1)
                    self.fused_qkv[p] = _Layer("w4a8", pk, ls[0].aq)
                    self.fused_qkv[p].sibling_qids = tuple(ids[1:])
            elif all(l.kind in ("fp", "w4") for l in ls) and len({l.kind for l in ls}) == 1 and not any(l.wide for l in ls):
                pf = ls[0].p
                ws = None if pf.wscale is None else torch.cat([l.p.wscale for l in ls])
                pk = ops.PackedF16(torch.cat([l.p.w16 for l in ls]), torch.cat([l.p.bias for l in ls]), 3 * pf.cout,
                                   pf.cin, 1, 1, ws)
                self.fused_qkv[p] = _Layer(ls[0].kind, pk, None, torch.cat([l.w32 for l in ls]) if self.exact_fp else None)
        self.tib_table = None
        self.prepared = True

    # ------------------------------------------------------------------ activation calibration
    def set_calibration(self, mode: Optional[str], k: int = 0):
        """mode 'init': every live activation quantizer is (re)initialised on the tensor it sees with
        the MSE scaler, in execution order, upstream layers already quantised (lazy init of
        UniformAffineQuantizer.forward, quant_layer.py:211-221, as driven by calibration.py:113-127).
        mode 'running': EMA min/max update then MINMAX (act_momentum_update, quant_layer.py:229-244).
        Results land in qtable[k].  mode 'record': nothing is updated; the fp32 tensor every live quantizer sees under table
        row k is kept in self.observed[qid] (bin-flip-rate tests)."""
        if mode is None:
            self.calib = None
            return
        if self.qtable is None:
            raise TfmqError("set_calibration: prepare() was not given a qtable")
        if self.act_state is None:
            self.act_state = torch.zeros(self.qtable.shape[1], 2, dtype=torch.float32, device=self.dev)
            self._qp_scratch = torch.zeros(1, 2, dtype=torch.float32, device=self.dev)
        self.calib = (mode, int(k))
        if mode == "record":
            self.observed = {}
        if self.step is not None:
            self.step.fill_(int(k))

    def _observe(self, aq, x, siblings=(), level: int = 256, always_zero: bool = False):
        mode, k = self.calib
        qid = aq.qid
        if self.calib_mask is not None and qid not in self.calib_mask:
            return
        if mode == "record":       # parity instrumentation: keep the tensor every live quantizer sees (tests compare its bins with the oracle's)
            self.observed[qid] = x.detach().clone()
            for s in siblings:
                self.observed[s] = self.observed[qid]
            return
        x = x.contiguous()
        if mode == "init":
            qp = ops.mse_search(x, 1, level, always_zero=always_zero)
            self.qtable[k, qid].copy_(qp[0])
            ops.act_range_update(ops.minmax(x), self.act_state[qid:qid + 1], self._qp_scratch, 0.95, level, init=True)
        elif mode == "init_minmax":  # Scaler.MINMAX init (aq_params of the non-calibrating drivers)
            ops.act_range_update(ops.minmax(x), self.act_state[qid:qid + 1], self.qtable[k, qid], 0.95, level, init=True)
        else:
            ops.act_range_update(ops.minmax(x), self.act_state[qid:qid + 1], self.qtable[k, qid], 0.95, level, init=False)
        if always_zero and mode != "init":      # the softmax quantizer: delta = x_max / (level - 1), zero point 0 (quant_layer.py:29-30,34)
            self.qtable[k, qid].copy_(ops.minmax_to_qparam(self.act_state[qid:qid + 1].contiguous(), level, always_zero=True)[0])
        for s in siblings:  # sibling quantizers see the identical tensor
            self.qtable[k, s].copy_(self.qtable[k, qid])
            self.act_state[s].copy_(self.act_state[qid])

    # ------------------------------------------------------------------ temporal information block
    def _linear(self, name, x, silu_in):
        ent = self.lin[name]
        if ent[0] == "fp":
            return ops.linear_small_f32(x, ent[1], ent[2], silu_in=silu_in)
        aq = ent[3]
        if ent[0] == "w8":
            xs = ops.silu(x) if silu_in else x
            if aq is not None:
                if self.calib is not None:
                    self._observe(aq, xs)
                k = 0 if self.step is None else int(self.step.item())
                qp = self.qtable[k, aq.qid]
                xs = ops.fake_quant(xs.contiguous(), qp[0:1].contiguous(), qp[1:2].contiguous(), 256)
            return ops.linear_small_f32(xs, ent[1], ent[2], silu_in=False)
        if aq is not None and self.calib is not None:
            xs = ops.silu(x) if silu_in else x
            self._observe(aq, xs)
            return ops.linear_small_w4(xs, ent[1], aq, silu_in=False)
        return ops.linear_small_w4(x, ent[1], aq if aq is not None else ops.qsel(None), silu_in=silu_in)

    def tib(self, t: torch.Tensor) -> List[torch.Tensor]:
        """QuantTemporalInformationBlockDDIM.forward (quant/quant_block.py:52-64): t [m] fp32 ->
        list of the per-ResnetBlock projections [m, Cout_i]."""
        d0, d1, projs, dim, ldm = self.tib_layout()
        emb = ops.timestep_embedding(t, dim, ldm_order=ldm)
        h = self._linear(d0, emb, False)
        temb = self._linear(d1, h, True)
        self._last_temb = temb
        return [self._linear(n, temb, True) for n in projs]

    def tib_layout(self):
        """(dense0, dense1, [projection layer names], embedding dim, ldm sin/cos order)"""
        return "temb.dense.0", "temb.dense.1", [r + ".temb_proj" for r in self.res_names], self.cfg["ch"], False

    def tib_widths(self):
        return [self.sd[n + ".weight"].shape[0] for n in self.tib_layout()[2]]

    def build_tib_table(self, t_values: Sequence[float]):
        """The TIB depends only on the step: evaluate it once per sampling step (with that
        step's activation parameters) into a [n_steps, sum Cout] table; the conv epilogues then
        index it with the device-side step counter (K7)."""
        if self.step is None and self.qtable is not None and self.qtable.shape[0] > 1:
            raise TfmqError("build_tib_table: a multi-step qtable needs a device step counter")
        # Set-up work on the CALLER's current stream that drives the shared device step counter through every step: nothing on another
        # stream may touch the counter meanwhile, and no stream may read the table before it is complete.  Samplers run on their own
        # (non-blocking) streams, which do not order themselves against this one -- a sampler that started right behind its constructor
        # used to race the loop below (its step.zero_() between a fill_ and the TIB kernels: rows computed under the wrong Finite-Set
        # group; found in round 4 through a bench leg that left the default stream busy).  Device-wide synchronisation on both sides.
        torch.cuda.synchronize(self.dev)
        rows = []
        for s, tv in enumerate(t_values):
            if self.step is not None:
                self.step.fill_(s)
            t = torch.full((1,), float(tv), dtype=torch.float32, device=self.dev)
            rows.append(torch.cat(self.tib(t), dim=1))
        self.tib_table = torch.cat(rows, dim=0).contiguous()
        off = 0
        for r, wdt in zip(self.res_names, self.tib_widths()):
            self.tib_off[r] = off
            off += wdt
        if self.step is not None:
            self.step.zero_()
        torch.cuda.synchronize(self.dev)
        return self.tib_table

    # ------------------------------------------------------------------ blocks
    def _gn(self, name, x1, x2, silu, layer: Optional[_Layer], want_cat=False, eps=1e-6, half=False, half_main=False):
        """half: the fp outputs of this GroupNorm (the concat copy for an un-quantised shortcut conv, or the normalised
        tensor itself when its consumer is an un-quantised conv) may be written as fp16 -- those convs round their input
        to fp16 anyway, so the result is bit-identical and the conv runs on the LDS-DMA path at half the bytes."""
        aq = layer.aq if (layer is not None and layer.kind == "w4a8") else None
        if aq is not None and self.calib is not None:
            _, yf, xcat = ops.groupnorm(x1, self.sd[name + ".weight"], self.sd[name + ".bias"], eps, silu, None, x2=x2,
                                        want_f32=True, want_cat=want_cat)
            self._observe(aq, yf, getattr(layer, "sibling_qids", ()))
            return ops.quantize_act(yf, aq), xcat
        # half: fp16 for the concat copy (and for the main output when there is no consumer layer, e.g. conv_out's
        # input); half_main: the main output feeds an un-quantised / weight-only conv directly, which takes fp16 too.
        # One flag covers both outputs of the kernel, so they must agree.
        if aq is None and layer is not None:
            main_ok = half_main and self._fp_conv_half_ok(layer)
            half = main_ok and (half or not want_cat)
        tp = _tape()
        if tp is not None and tp.depends(x1, x2):
            if aq is not None or half or x1.dtype != torch.float32:
                raise TfmqError("GradTape: the differentiated tail must be un-quantised fp32 (exact mode)")
            want_cat = want_cat or x2 is not None           # the backward needs the concatenated input
        yq, yf, xcat = ops.groupnorm(x1, self.sd[name + ".weight"], self.sd[name + ".bias"], eps, silu, aq, x2=x2,
                                     want_cat=want_cat, half_out=half)
        if tp is not None and tp.depends(x1, x2):
            gamma, beta, c1 = self.sd[name + ".weight"], self.sd[name + ".bias"], x1.shape[-1]
            xin = xcat if x2 is not None else x1

            def bwd(gouts):
                gy, gcat = gouts[0], (gouts[1] if len(gouts) > 1 else None)
                gx = ops.groupnorm_bwd(xin, gy.contiguous(), gamma, beta, eps, silu) if gy is not None else None
                if gcat is not None:                     # the shortcut conv read the concat copy
                    gx = gcat.contiguous().clone() if gx is None else ops.axpy(gx, gcat.contiguous(), 1.0)
                if x2 is None:
                    return [gx, None]
                return [gx[..., :c1].contiguous(), gx[..., c1:].contiguous()]
            tp.rec([x1, x2], [yf] + ([xcat] if xcat is not None else []), bwd)
        return (yq if aq is not None else yf), xcat

    def _o16(self) -> dict:
        return {"out_f16": True} if self._h16 else {}

    def _stream_f16_possible(self) -> bool:
        """Every un-quantised / weight-only conv that consumes a stream tensor must take fp16 input (LDS-DMA path)."""
        for n, l in self.layers.items():
            if l.kind != "w4a8" and not ops.f16_dma_ok(l.p.cin, l.p.kh, l.p.kw) and l.p.cin > 4:
                return False
        return True

    def _fp_conv_half_ok(self, layer: _Layer) -> bool:
        return (not self.exact_fp) and layer.kind != "w4a8" and ops.f16_dma_ok(layer.p.cin, layer.p.kh, layer.p.kw)

    def _attention_quantised(self, key: str, q, k, v, heads: int, scale: float, pre: float = 1.0):
        """The attention of a block whose quantizers were switched on (self.attn_q[key]): fp32 out."""
        cfg = self.attn_q[key]
        sel = {w: ops.qsel(self.qtable, cfg[w], self.step) for w in ("q", "k", "v", "w")}
        obs = None
        if self.calib is not None:
            def obs(which, t):
                self._observe(sel[which], t, level=cfg["w_level"] if which == "w" else 256, always_zero=which == "w")
        # Both products on the int8 matrix cores over the quantizers' bins (tfmq_attention_q8) whenever the quantizers are live and nothing
        # needs the intermediate tensors (no observer, no gradient tape): exact integer sums, pinned against the oracle's restatement
        # (tests/test_attention_q8_gpu.py).  The exact-fp32 engine keeps ops.attention_quant -- fp32 products of the dequantised values in the
        # reference's order, what F21 was pinned with -- unless TFMQ_ATTN_Q8=1 forces the kernel; TFMQ_ATTN_Q8=0 switches it off everywhere.
        env = os.environ.get("TFMQ_ATTN_Q8", "")
        want_q8 = env == "1" or (env != "0" and not self.exact_fp)
        if obs is None and want_q8 and ops.attention_q8_ok(q.shape[-1] // heads, cfg["w_level"]) and _tape() is None:
            return ops.attention_q8(q, k, v, heads, scale, sel["q"], sel["k"], sel["v"], sel["w"], cfg["w_level"], pre)
        return ops.attention_quant(q, k, v, heads, scale, sel["q"], sel["k"], sel["v"], sel["w"], cfg["w_level"], pre, obs)

    def _attention_exact(self, q, k, v, heads: int, scale: float, aq):
        """softmax(q k^T scale) v as exact-fp32 matmuls and a row softmax (three launches per head); (fp32 out, int8 bins | None)"""
        tp = _tape()
        if tp is not None and tp.depends(q, k, v):
            from .fisher import attention_taped
            out = attention_taped(tp, q, k, v, heads, scale)
        else:
            out, _ = ops._attention_wide(q, k, v, heads, scale, None, True)
        return out, (ops.quantize_act(out, aq) if aq is not None else None)

    def _virtual_cat_ok(self, layer: _Layer, x1, x2) -> bool:
        """The shortcut conv of an up-path block can read cat(x1, x2) from its two fp16 sources (tfmq_conv_desc.x2): the
        GroupNorm then does not write the concat copy (2 of its 5 bytes per element).  fp16 stream only."""
        return (x2 is not None and self._h16 and os.environ.get("TFMQ_VIRTUAL_CAT", "1") != "0" and layer.kind != "w4a8" and x1.dtype == torch.float16 and x2.dtype == torch.float16
                and layer.p.cout % 8 == 0 and ops.f16_cat_ok(x1.shape[-1], x2.shape[-1], layer.p.kh, layer.p.kw))

    def _resblock(self, p, x1, x2, rowadd_kw):
        L = self.layers
        has_sc = (p + ".nin_shortcut") in L
        if x2 is not None and not has_sc:
            raise TfmqError(f"{p}: concatenated input without nin_shortcut is not a DDPM-UNet block")
        half = has_sc and self._fp_conv_half_ok(L[p + ".nin_shortcut"])
        virt = has_sc and self._virtual_cat_ok(L[p + ".nin_shortcut"], x1, x2)
        h, xcat = self._gn(p + ".norm1", x1, x2, True, L[p + ".conv1"],
                           want_cat=has_sc and not virt and (x2 is not None or (half and x1.dtype != torch.float16)), half=half, half_main=True)
        h = L[p + ".conv1"].run(h, pad=(1, 1, 1, 1), **rowadd_kw, **self._o16())
        h, _ = self._gn(p + ".norm2", h, None, True, L[p + ".conv2"], half_main=True)
        if virt:
            sc = L[p + ".nin_shortcut"].run(x1, x2=x2, **self._o16())
        elif has_sc:
            sc = L[p + ".nin_shortcut"].run(xcat if xcat is not None else x1, **self._o16())
        else:
            sc = x1
        return L[p + ".conv2"].run(h, pad=(1, 1, 1, 1), residual=sc, **self._o16())

    def _attnblock(self, p, x):
        L = self.layers
        B, H, W, Cc = x.shape
        po = L[p + ".proj_out"]
        f = self.fused_qkv.get(p)
        if (f is not None and f.kind == "w4a8" and self.calib is None and not self.exact_fp and p not in self.attn_q and ops.attention_f16_ok(Cc, H * W)
                and (2 * Cc) % 128 == 0 and (H * W) % 4 == 0):
            # the fused q|k|v GEMM writes q, k as fp16 rows and v as fp16 V^T; the flash kernel reads them tile by tile
            # (what the fp32-operand kernel rounds to on load -- same products, no fp32 round trip)
            h, _ = self._gn(p + ".norm", x, None, False, f)
            y16, vt = ops.conv2d_w4a8(h, f.p, f.aq, out_f16=True, t_col0=2 * Cc)
            y16 = y16.reshape(B, H * W, 3 * Cc)
            aq = po.aq if po.kind == "w4a8" else None
            out, oq = ops.attention_f16(y16[..., :Cc], y16[..., Cc:2 * Cc], vt, 1, float(int(Cc) ** (-0.5)), aq,
                                        want_f32=aq is None)
            return po.run((oq if aq is not None else out).reshape(B, H, W, Cc), residual=x, **self._o16())
        if p in self.fused_qkv:
            f = self.fused_qkv[p]
            h, _ = self._gn(p + ".norm", x, None, False, f)
            qkv = f.run(h)
        else:
            qkv = ops._alloc(B, H, W, 3 * Cc, dtype=torch.float32, device=x.device)
            cache = {}
            for i, s in enumerate((".q", ".k", ".v")):
                l = L[p + s]
                key = (l.kind == "w4a8", l.aq.qid if l.aq is not None else -1)
                if key not in cache:
                    cache[key], _ = self._gn(p + ".norm", x, None, False, l)
                l.run(cache[key], out=qkv, y_coff=i * Cc, want_stats=False)
        qkv = qkv.reshape(B, H * W, 3 * Cc)
        aq = po.aq if po.kind == "w4a8" else None
        attn = self._attention_exact if self.exact_fp else (lambda q, k, v, h, sc, a=None: ops.attention(q, k, v, h, sc, a, want_f32=a is None))
        if p in self.attn_q:      # QuantAttnBlock.use_aq (quant_block.py:487-498): quantised q, k, softmax, v
            def attn(q, k, v, h, sc, a=None):
                o = self._attention_quantised(p, q, k, v, h, sc)
                return o, (ops.quantize_act(o, a) if a is not None else None)
        if aq is not None and self.calib is not None:
            out, _ = attn(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], 1, float(int(Cc) ** (-0.5)), None)
            self._observe(aq, out)
            return po.run(ops.quantize_act(out, aq).reshape(B, H, W, Cc), residual=x)
        out, oq = attn(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], 1, float(int(Cc) ** (-0.5)), aq)
        a = (oq if aq is not None else out).reshape(B, H, W, Cc)
        return po.run(a, residual=x, **self._o16())

    # ------------------------------------------------------------------ forward
    def forward(self, *a, **k):
        """See _forward.  Outside activation calibration the conv tile shapes are measured once per shape
        (ops.autotuned) and reused -- the output does not depend on them."""
        if not hasattr(self, "tiles"):
            self.tiles = {}
        try:
            with ops.autotuned(self.tiles if self.calib is None else None):
                return self._forward(*a, **k)
        except UnitReached:
            return None

    def _forward(self, x: torch.Tensor, t: Optional[torch.Tensor] = None, taps: Optional[dict] = None) -> torch.Tensor:
        """x: fp32 NHWC [B,H,W,C].  t: [B] fp32 timesteps, or None to use the per-step TIB table
        (build_tib_table) indexed by the device step counter."""
        if not self.prepared:
            raise TfmqError("DdimUNetEngine.forward before prepare()")
        cfg, L = self.cfg, self.layers
        nlev, nres = len(cfg["ch_mult"]), cfg["num_res_blocks"]
        self._h16 = self.stream_f16 and self.calib is None and taps is None and self._stream_f16_possible()
        if t is not None:
            projs = dict(zip(self.res_names, self.tib(t)))
            if taps is not None:
                taps["__temb__"] = self._last_temb

            def rowadd(p):
                return dict(rowadd=projs[p])
        else:
            if self.tib_table is None:
                raise TfmqError("forward(t=None) needs build_tib_table() first")

            def rowadd(p):
                o = self.tib_off[p]
                return dict(rowadd=self.tib_table[0, o:], rowadd_ld=0, rowadd_step=self.step,
                            rowadd_step_stride=self.tib_table.shape[1])

        def tap(name, a, b):
            if taps is not None:
                taps[name] = (a, b)

        hs = [L["conv_in"].run(x, pad=(1, 1, 1, 1), **self._o16())]
        res = cfg["resolution"]
        for i in range(nlev):
            for j in range(nres):
                p = f"down.{i}.block.{j}"
                h = self._resblock(p, hs[-1], None, rowadd(p))
                tap(p, hs[-1], h)
                if res in cfg["attn_resolutions"]:
                    hin = h
                    h = self._attnblock(f"down.{i}.attn.{j}", h)
                    tap(f"down.{i}.attn.{j}", hin, h)
                hs.append(h)
            if i != nlev - 1:
                # Downsample (ddim/models/diffusion.py:65-72): pad (0,1,0,1), 3x3 stride 2, un-quantised
                dl = L[f"down.{i}.downsample.conv"]
                hin_d = hs[-1]
                if hin_d.dtype == torch.float32 and self._fp_conv_half_ok(dl):
                    hin_d = ops.to_half(hin_d)
                hs.append(dl.run(hin_d, stride=2, pad=(0, 0, 1, 1), **self._o16()))
                res //= 2
        h = hs[-1]
        hin = h
        h = self._resblock("mid.block_1", h, None, rowadd("mid.block_1"))
        tap("mid.block_1", hin, h)
        hin = h
        h = self._attnblock("mid.attn_1", h)
        tap("mid.attn_1", hin, h)
        hin = h
        h = self._resblock("mid.block_2", h, None, rowadd("mid.block_2"))
        tap("mid.block_2", hin, h)
        for i in reversed(range(nlev)):
            for j in range(nres + 1):
                p = f"up.{i}.block.{j}"
                skip = hs.pop()
                hin = h
                h = self._resblock(p, h, skip, rowadd(p))
                tap(p, (hin, skip), h)
                if res in cfg["attn_resolutions"]:
                    hin = h
                    h = self._attnblock(f"up.{i}.attn.{j}", h)
                    tap(f"up.{i}.attn.{j}", hin, h)
            if i != 0:
                up = L[f"up.{i}.upsample.conv"]
                if up.kind == "w4a8" and self.calib is not None:
                    self._observe(up.aq, h)
                hq = ops.quantize_act(h, up.aq) if up.kind == "w4a8" else h
                hlow = h
                h = up.run(hq, pad=(1, 1, 1, 1), up2x=True, **self._o16())
                if taps is not None:  # layer unit: its input is the up-sampled tensor (Upsample.forward)
                    taps[f"up.{i}.upsample.conv"] = (ops.upsample2x(hlow), h)
                res *= 2
        h, _ = self._gn("norm_out", h, None, True, None, half=self._fp_conv_half_ok(L["conv_out"]))
        return L["conv_out"].run(h, pad=(1, 1, 1, 1))
