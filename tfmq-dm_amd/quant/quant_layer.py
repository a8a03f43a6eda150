"""Quantisation primitives of the reference's quant/quant_layer.py on the HIP kernels.

Same public names and call signatures (`minmax`, `mse`, `Scaler`, `lp_loss`,
`UniformAffineQuantizer`, `QuantLayer`, `QMODE`, `StraightThrough`); every tensor operation is a
launch through the C ABI (K1-K6).  CPU tensors are refused: there is no CPU fallback.
"""
from __future__ import annotations

from enum import Enum
from typing import List, Optional, Union

import os

import torch
import torch.nn as nn

from tfmq_dm_amd import ops
from tfmq_dm_amd._lib import TfmqError


class StraightThrough(nn.Module):
    def forward(self, x):
        return x


def _scalar(qp: torch.Tensor, i: int) -> torch.Tensor:
    return qp[0, i].clone()


def minmax(x: torch.Tensor, symmetric: bool = False, level: int = 256, always_zero: bool = False):
    """MINMAX scaler (reference quant_layer.py:20-35) -> (delta, zero_point) 0-dim device tensors."""
    if symmetric:
        raise NotImplementedError("symmetric quantisation is never selected by the TFMQ drivers")
    qp = ops.minmax_to_qparam(ops.minmax(x.detach().contiguous().float(), 1), level, always_zero)
    return _scalar(qp, 0), _scalar(qp, 1)


def mse(x: torch.Tensor, symmetric: bool = False, level: int = 256, always_zero: bool = False):
    """MSE scaler: 80 shrink candidates, L2.4 loss, first strict minimum (reference :38-64)."""
    if symmetric:
        raise NotImplementedError("symmetric quantisation is never selected by the TFMQ drivers")
    qp = ops.mse_search(x.detach().contiguous().float(), 1, level, always_zero)
    return _scalar(qp, 0), _scalar(qp, 1)


def _np_density(counts, edges):
    """np.histogram(..., density=True) from the bin counts."""
    import numpy as np
    return counts / np.diff(edges) / counts.sum()


def _clipped_qparam(x: torch.Tensor, xmin, xmax, lo, hi, level: int, always_zero: bool):
    """MINMAX of x after `torch.where(x < lo, lo, x)`, `torch.where(x > hi, hi, x)` (reference :110-112): the extremes of the
    clipped tensor are fp32(lo) / fp32(hi) where something was clipped and the data's own extremes otherwise."""
    import numpy as np
    cmin = np.float32(lo) if float(xmin) < lo else np.float32(xmin)
    cmax = np.float32(hi) if float(xmax) > hi else np.float32(xmax)
    if cmin > np.float32(hi):        # everything above hi: the first pass keeps x, the second lowers all of it to hi
        cmin = np.float32(hi)
    mm = torch.tensor([[float(cmin), float(cmax)]], dtype=torch.float32, device=x.device)
    qp = ops.minmax_to_qparam(mm, level, always_zero)
    return _scalar(qp, 0), _scalar(qp, 1)


def kl(x: torch.Tensor, symmetric: bool = False, level: int = 256, always_zero: bool = False):
    """KL scaler (reference quant_layer.py:67-113): clip ratio in linspace(0.5, 1, 50) whose clipped-data histogram, re-binned on
    the raw histogram's grid, has the smallest KL divergence from the raw histogram (level density bins, 1e-5 smoothing, first
    strict minimum); then MINMAX of the clipped tensor.  The 51 histograms are device passes (tfmq_np_histogram: numpy's bin
    arithmetic, fp32 for the raw data, float64 for the clipped data as np.clip's float64 bounds make it); the 256-bin
    bookkeeping is host arithmetic on the counts."""
    import numpy as np
    if symmetric:
        raise NotImplementedError("symmetric quantisation is never selected by the TFMQ drivers")
    xf = x.detach().contiguous().float()
    mm = ops.minmax(xf, 1).cpu().numpy()
    xmin, xmax = np.float32(mm[0, 0]), np.float32(mm[0, 1])
    ref_edges = np.linspace(xmin, xmax, level + 1, endpoint=True, dtype=np.float32)
    ref_hist = _np_density(ops.np_histogram(xf, ref_edges), ref_edges)
    width = np.sum(np.diff(ref_edges))
    p = (ref_hist + 1e-5) / (1.0 + width * 1e-5)
    best, best_r = 1e5, 1.0
    for r in np.linspace(0.5, 1.0, 50):
        lo, hi = xmin * r, xmax * r                                   # float64 (np.float32 * np.float64)
        first, last = min(max(float(xmin), lo), hi), min(max(float(xmax), lo), hi)     # extremes of the clipped data
        q_edges = np.linspace(first, last, level + 1, endpoint=True, dtype=np.float64)
        q_hist = _np_density(ops.np_histogram(xf, q_edges, clip=(lo, hi)), q_edges)
        # walk the raw grid's left edges, stepping ONE clipped bin forward whenever its right edge has been reached (:73-90)
        out = np.zeros(level, dtype=q_hist.dtype)
        v, j, edge = 0.0, 0, q_edges[0]
        for i in range(level):
            left = ref_edges[i]
            if edge <= left:
                if j < level:
                    v = q_hist[j]
                    j += 1
                    edge = q_edges[j]
                else:
                    v = 0.0
                    edge = left + 1.0
            out[i] = v
        q = (out + 1e-5) / (1.0 + width * 1e-5)
        dkl = np.sum(p * np.log(p / q))
        if dkl < best:
            best, best_r = dkl, r
    return _clipped_qparam(xf, xmin, xmax, xmin * best_r, xmax * best_r, level, always_zero)


def hist(x: torch.Tensor, symmetric: bool = False, level: int = 256, always_zero: bool = False):
    """HIST scaler (reference quant_layer.py:116-132): the smallest clip value (i + 0.5) max|x| / level at which the fp32 running
    sum of the (0, max|x|) histogram reaches 99.96 % of its mass; clip to [max(-c, min x), min(c, max x)], then MINMAX."""
    import numpy as np
    if symmetric:
        raise NotImplementedError("symmetric quantisation is never selected by the TFMQ drivers")
    xf = x.detach().contiguous().float()
    mm = ops.minmax(xf, 1).cpu().numpy()
    xmin, xmax = np.float32(mm[0, 0]), np.float32(mm[0, 1])
    amax = max(-xmin, xmax)
    edges = np.linspace(0, amax, level + 1, endpoint=True, dtype=np.float32)
    h = _np_density(ops.np_histogram(xf, edges), edges)
    # float32-rounded densities divided in float64: what `hist.astype(np.float32) / hist.sum()` evaluates to under NumPy >= 2 (NEP 50: a
    # float64 SCALAR is not value-cast any more), the NumPy the fixtures were generated with; explicit so that NumPy 1.x gives the same bits
    h = h.astype(np.float32).astype(np.float64) / np.float64(h.sum())
    acc, lo, hi = 0, None, None
    for i in range(level):
        acc += h[i]
        if acc >= 0.9996:
            c = (i + 0.5) * (amax / level)
            lo, hi = max(-c, xmin), min(c, xmax)
            break
    return _clipped_qparam(xf, xmin, xmax, float(lo), float(hi), level, always_zero)


def _clipped_qparam_rows(x2: torch.Tensor, xmin, xmax, lo, hi, level: int, always_zero: bool):
    """_clipped_qparam for every row (numpy arrays xmin / xmax float32, lo / hi float64) -> device [rows, 2] {delta, zero_point}"""
    import numpy as np
    lo32, hi32 = lo.astype(np.float32), hi.astype(np.float32)
    cmin = np.where(xmin.astype(np.float64) < lo, lo32, xmin)
    cmax = np.where(xmax.astype(np.float64) > hi, hi32, xmax)
    cmin = np.where(cmin > hi32, hi32, cmin)
    mm = torch.from_numpy(np.stack([cmin, cmax], axis=1).astype(np.float32)).to(x2.device)
    return ops.minmax_to_qparam(mm.contiguous(), level, always_zero)


def kl_rows(x2: torch.Tensor, level: int = 256, always_zero: bool = False) -> torch.Tensor:
    """`kl` for every row of x2 [rows, n] (the per-output-channel loop of the reference, quant_layer.py:193-204) with the 51 histograms of
    ALL rows in 51 launches and the 256-bin bookkeeping vectorised over the rows -- the same numpy operations in the same precisions per
    row as `kl`, so delta / zero point are those of the loop bit for bit (tests/test_hip_kernels.py) instead of ~51 launches and host
    synchronisations per channel.  -> device [rows, 2]."""
    import numpy as np
    xf = x2.detach().contiguous().float()
    R = xf.shape[0]
    mm = ops.minmax(xf, R).cpu().numpy()
    xmin, xmax = mm[:, 0].astype(np.float32), mm[:, 1].astype(np.float32)
    ref_edges = np.linspace(xmin, xmax, level + 1, endpoint=True, dtype=np.float32, axis=-1)
    ref_cnt = ops.np_histogram_rows(xf, ref_edges)
    ref_hist = ref_cnt / np.diff(ref_edges, axis=-1) / ref_cnt.sum(axis=-1, keepdims=True)
    width = np.sum(np.diff(ref_edges, axis=-1), axis=-1)
    p = (ref_hist + 1e-5) / (1.0 + width * 1e-5)[:, None]
    best, best_r = np.full(R, 1e5), np.full(R, 1.0)
    xmin64, xmax64 = xmin.astype(np.float64), xmax.astype(np.float64)
    rows = np.arange(R)
    for r in np.linspace(0.5, 1.0, 50):
        lo, hi = xmin * r, xmax * r                                   # float64 (float32 array * np.float64)
        first, last = np.minimum(np.maximum(xmin64, lo), hi), np.minimum(np.maximum(xmax64, lo), hi)
        q_edges = np.linspace(first, last, level + 1, endpoint=True, dtype=np.float64, axis=-1)
        q_cnt = ops.np_histogram_rows(xf, q_edges, clip=(lo, hi))
        q_hist = q_cnt / np.diff(q_edges, axis=-1) / q_cnt.sum(axis=-1, keepdims=True)
        out = np.zeros((R, level), dtype=q_hist.dtype)
        v = np.zeros(R, dtype=np.float64)
        j = np.zeros(R, dtype=np.int64)
        edge = q_edges[:, 0].copy()
        for i in range(level):
            left = ref_edges[:, i]
            step = edge <= left
            take = step & (j < level)
            over = step & ~(j < level)
            jt = np.minimum(j, level - 1)
            v = np.where(take, q_hist[rows, jt], np.where(over, 0.0, v))
            j = np.where(take, j + 1, j)
            edge = np.where(take, q_edges[rows, np.minimum(j, level)], np.where(over, (left + np.float32(1.0)).astype(np.float64), edge))
            out[:, i] = v
        q = (out + 1e-5) / (1.0 + width * 1e-5)[:, None]
        dkl = np.sum(p * np.log(p / q), axis=-1)
        better = dkl < best
        best, best_r = np.where(better, dkl, best), np.where(better, r, best_r)
    return _clipped_qparam_rows(xf, xmin, xmax, xmin * best_r, xmax * best_r, level, always_zero)


def hist_rows(x2: torch.Tensor, level: int = 256, always_zero: bool = False) -> torch.Tensor:
    """`hist` for every row of x2 [rows, n] with one histogram launch for all rows (see kl_rows).  -> device [rows, 2]."""
    import numpy as np
    xf = x2.detach().contiguous().float()
    R = xf.shape[0]
    mm = ops.minmax(xf, R).cpu().numpy()
    xmin, xmax = mm[:, 0].astype(np.float32), mm[:, 1].astype(np.float32)
    amax = np.maximum(-xmin, xmax)
    edges = np.linspace(np.zeros(R, dtype=np.float64), amax, level + 1, endpoint=True, dtype=np.float32, axis=-1)     # (`hist` starts at the int 0: float64 arithmetic)
    cnt = ops.np_histogram_rows(xf, edges)
    h = cnt / np.diff(edges, axis=-1) / cnt.sum(axis=-1, keepdims=True)
    h = h.astype(np.float32).astype(np.float64) / h.sum(axis=-1, keepdims=True).astype(np.float64)      # (as in `hist`: explicit float64)
    lo, hi = np.zeros(R, dtype=np.float64), np.zeros(R, dtype=np.float64)
    for rr in range(R):          # a 256-step running sum per row: host arithmetic in `hist`'s own order
        acc = 0
        for i in range(level):
            acc += h[rr, i]
            if acc >= 0.9996:
                c = (i + 0.5) * (amax[rr] / level)
                lo[rr], hi[rr] = max(-c, xmin[rr]), min(c, xmax[rr])
                break
    return _clipped_qparam_rows(xf, xmin, xmax, lo, hi, level, always_zero)


class Scaler:
    """Namespace of scaler functions (the reference's Enum of plain functions is just that)."""
    MINMAX = staticmethod(minmax)
    MSE = staticmethod(mse)
    KL = staticmethod(kl)
    HIST = staticmethod(hist)


REDUCTION = Enum("REDUCTION", ("NONE", "ALL"))
QMODE = Enum("QMODE", ("QDIFF", "NORMAL", "PTQD"))


def lp_loss(pred: torch.Tensor, tgt: torch.Tensor, p: float = 2.0, reduction: REDUCTION = REDUCTION.NONE) -> torch.Tensor:
    """|pred-tgt|^p summed over dim 1, mean over the rest (reference :146-156); p = 2 on device."""
    if p != 2.0 or reduction != REDUCTION.NONE:
        raise NotImplementedError("lp_loss: the reconstruction path uses p=2, REDUCTION.NONE")
    denom = pred.numel() // pred.shape[1]
    loss, _ = ops.recon_loss(pred.contiguous(), tgt.contiguous(), denom, want_grad=False)
    return loss[0]


def _scaler_kind(fn) -> str:
    name = getattr(fn, "__name__", str(fn))
    if name not in ("mse", "minmax", "kl", "hist"):
        raise NotImplementedError(f"unknown scaler {name}")
    return name


class UniformAffineQuantizer(nn.Module):
    """Asymmetric uniform quantizer with lazy initialisation (reference :163-253)."""

    def __init__(self, bits: int = 8, symmetric: bool = False, channel_wise: bool = False, scaler=Scaler.MINMAX,
                 leaf_param: bool = False, always_zero: bool = False, quant_emb: bool = False) -> None:
        super().__init__()
        if symmetric:
            raise NotImplementedError("symmetric quantisation is never selected by the TFMQ drivers")
        self.level = 2 ** bits
        self.symmetric = symmetric
        self.channel_wise = channel_wise
        self.scaler = scaler
        self.leaf_param = leaf_param
        if leaf_param:
            self.x_min, self.x_max = None, None
        self.running_stat = False
        self.always_zero = always_zero
        self.delta = None
        self.zero_point = None
        self.init = False
        self.quant_emb = quant_emb

    def _init_quantization_param(self, x: torch.Tensor, channel_wise: bool = False):
        kind = _scaler_kind(self.scaler)
        x = x.detach().contiguous().float()
        rows = x.shape[0] if channel_wise else 1
        if kind in ("kl", "hist"):
            # the histogram scalers work on one tensor at a time (reference :193-204 loops the channels through the scaler)
            fn = kl if kind == "kl" else hist
            if channel_wise:
                shape = (-1,) + (1,) * (x.dim() - 1)
                if os.environ.get("TFMQ_SCALER_ROW_LOOP") is None:       # all channels per launch (kl_rows / hist_rows); the loop is kept for the test
                    qp = (kl_rows if kind == "kl" else hist_rows)(x.reshape(rows, -1), self.level, self.always_zero)
                    return qp[:, 0].clone().view(shape), qp[:, 1].clone().view(shape)
                pairs = [fn(x[c], False, self.level, self.always_zero) for c in range(rows)]
                return torch.stack([p[0] for p in pairs]).view(shape), torch.stack([p[1] for p in pairs]).view(shape)
            if self.leaf_param:
                mm = ops.minmax(x, 1)
                self.x_min, self.x_max = mm[0, 0].clone(), mm[0, 1].clone()
            return fn(x, False, self.level, self.always_zero)
        if kind == "mse":
            qp = ops.mse_search(x, rows, self.level, self.always_zero)
        else:
            qp = ops.minmax_to_qparam(ops.minmax(x, rows), self.level, self.always_zero)
        if channel_wise:
            shape = (-1,) + (1,) * (x.dim() - 1)
            return qp[:, 0].clone().view(shape), qp[:, 1].clone().view(shape)
        if self.leaf_param:
            mm = ops.minmax(x, 1)
            self.x_min, self.x_max = mm[0, 0].clone(), mm[0, 1].clone()
        return _scalar(qp, 0), _scalar(qp, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.init:
            self.delta, self.zero_point = self._init_quantization_param(x, self.channel_wise)
            if self.leaf_param:
                self.delta = nn.Parameter(self.delta)
            self.init = True
        if self.running_stat:
            self.act_momentum_update(x)
        zp = self.zero_point if torch.is_tensor(self.zero_point) else torch.tensor(float(self.zero_point), device=x.device)
        return ops.fake_quant(x.detach().contiguous().float(), self.delta.detach(), zp.detach(), self.level)

    def act_momentum_update(self, x: torch.Tensor, act_range_momentum: float = 0.95) -> None:
        assert self.init and self.leaf_param
        state = torch.stack([self.x_min, self.x_max]).reshape(1, 2).contiguous()
        qp = torch.empty(1, 2, device=x.device)
        ops.act_range_update(ops.minmax(x.detach().contiguous().float(), 1), state, qp, act_range_momentum, self.level, init=False)
        self.x_min, self.x_max = state[0, 0].clone(), state[0, 1].clone()
        self.zero_point = qp[0, 1].clone()
        self.delta = nn.Parameter(qp[0, 0].clone())

    def bitwidth_refactor(self, bits: int = 8) -> None:
        self.level = 2 ** bits

    def extra_repr(self) -> str:
        return (f"level={self.level}, symmetric={self.symmetric}, channel_wise={self.channel_wise}, "
                f"scaler={getattr(self.scaler, '__name__', self.scaler)}, leaf_param={self.leaf_param}")


class QuantLayer(nn.Module):
    """Conv2d / Linear wrapper (reference :259-355).  `forward` is the eager per-layer device path
    (NCHW / [..., C] in and out, like the reference); whole-model execution goes through the fused
    engine plan that QuantModel builds from these objects."""

    QMAP = {nn.Conv2d: "conv2d", nn.Linear: "linear"}

    def __init__(self, layer: Union[nn.Conv2d, nn.Linear], wq_params: dict = {}, aq_params: dict = {},
                 disable_aq: bool = False, aq_mode: List[int] = [QMODE.QDIFF.value], quant_emb: bool = False) -> None:
        super().__init__()
        if type(layer) not in self.QMAP:
            raise TfmqError(f"QuantLayer: unsupported layer type {type(layer).__name__}")
        self.wq_params, self.aq_params = wq_params, aq_params
        self.kind = self.QMAP[type(layer)]
        self.fwd_kwargs = {}
        if isinstance(layer, nn.Conv2d):
            if layer.groups != 1 or tuple(layer.dilation) != (1, 1):
                raise TfmqError("QuantLayer: grouped / dilated convolutions are not on the TFMQ hot path")
            self.fwd_kwargs = dict(stride=layer.stride, padding=layer.padding, dilation=layer.dilation, groups=layer.groups)
        self.w = layer.weight
        self.original_w = self.w.data.clone()
        self.b = None
        self.original_b = None
        if layer.bias is not None:
            self.b = layer.bias
            self.original_b = self.b.data.clone()
        self.use_wq = False
        self.use_aq = False
        self.disable_aq = disable_aq
        self.aq_mode = aq_mode
        self.quant_emb = quant_emb
        self.wq_params["quant_emb"] = quant_emb
        self.wqtizer = UniformAffineQuantizer(**self.wq_params)
        self.aqtizer = UniformAffineQuantizer(**self.aq_params)
        self.split = 0
        self.act_func = StraightThrough()
        self.ignore_recon = False
        self.extra_repr = layer.extra_repr
        self._packed = None  # (key, packed weights)

    # -- quantizer state as the engine sees it ------------------------------------------------
    def weight_quant_state(self):
        """(delta, zero_point, alpha|None) of the weight quantizer, initialising it lazily exactly
        like `self.wqtizer(self.w)` would (reference :330-333)."""
        if self.split != 0 and QMODE.QDIFF.value in self.aq_mode:
            raise TfmqError("QuantLayer: a QDIFF-split layer (two quantizer pairs, reference :310-329) runs through its own forward; the "
                            "fused engine plan and the block reconstruction units take un-split layers only (the tree rewrite never "
                            "produces a split one: quant_model.py:57-58 keeps skip / shortcut convs un-quantised)")
        return self._wq_state(self.wqtizer, self.w.data)

    @staticmethod
    def _wq_state(q, w):
        if isinstance(q, UniformAffineQuantizer):
            if not q.init:
                q.delta, q.zero_point = q._init_quantization_param(w, q.channel_wise)
                q.init = True
            return q.delta.detach(), q.zero_point.detach(), None
        return q.delta.detach(), q.zero_point.detach(), q.alpha.detach()  # AdaRoundQuantizer (hard rounding)

    def _pack(self, mode: str, part: int = 0):
        """part 0: the whole layer; 1 / 2: the input-channel halves [:split] / [split:] of a QDIFF-split layer with their own weight
        quantizers (reference :325-329); the bias rides with part 1."""
        wq = self.wqtizer if part < 2 else self.wqtizer1
        aq = self.aqtizer if part < 2 else self.aqtizer1
        key = (mode, part, self.split, self.w.data_ptr(), id(wq), getattr(wq, "_version_", 0))
        hit = self._packed.get(part) if isinstance(self._packed, dict) else None
        if hit is not None and hit[0] == key:
            return hit[1]
        if not isinstance(self._packed, dict):
            self._packed = {}
        dev = self.w.device
        sl = slice(None) if part == 0 else (slice(0, self.split) if part == 1 else slice(self.split, None))
        b = None if (self.b is None or part == 2) else (self.b if mode != "fp" else self.original_b.to(dev)).detach().float().contiguous()
        if mode == "fp":
            pk = ops.pack_w_f16(self.original_w.to(dev)[:, sl].float().contiguous(), b)
        else:
            w = self.w.detach()[:, sl].float().contiguous()
            d, z, a = self._wq_state(wq, w)
            a = None if a is None else a.float().contiguous()
            if mode == "w4a8" and (not 2 <= wq.level <= 2048 or aq.level != 256):
                raise TfmqError(f"QuantLayer: the device path is 2..2048-level weights x 8-bit activations; got {wq.level} weight "
                                f"levels / {aq.level} activation levels")
            # more than 16 weight levels (--wq 8): the fp16 integer grid, run on the fp16-operand kernels (engine _Layer._run_wide)
            int8_path = mode == "w4a8" and wq.level <= 16
            pk = ops.pack_w4(w, d, z, a, b) if int8_path else ops.pack_w_f16(w, b, d, z, a, wq.level)
        self._packed[part] = (key, pk)
        return pk

    def _run(self, xn, mode: str, part: int, stride: int, pad, residual=None):
        """One launch on NHWC input `xn` (fp32): the layer, or one input-channel half of a split layer (its output accumulates onto
        `residual`, the other half's)."""
        aqt = self.aqtizer if part < 2 else self.aqtizer1
        pk = self._pack(mode, part)
        if mode == "w4a8":
            if not aqt.init:
                aqt(xn)  # lazy init on this tensor (mse / minmax), reference :211-221
            elif aqt.running_stat:
                aqt.act_momentum_update(xn)
            zp = aqt.zero_point
            zp = zp.detach() if torch.is_tensor(zp) else torch.tensor(float(zp), device=xn.device)
            qt = torch.stack([aqt.delta.detach().reshape(()), zp.reshape(())]).reshape(1, 1, 2).contiguous()
            sel = ops.qsel(qt)
            xq = ops.quantize_act(xn, sel)
            if isinstance(pk, ops.PackedW4):
                return ops.conv2d_w4a8(xq, pk, sel, stride=stride, pad=pad, residual=residual)
            # W8A8: exact integer grids on the fp16-operand kernel, output scale delta_a * delta_w[c]
            pf = ops.PackedF16(pk.w16, pk.bias, pk.cout, pk.cin, pk.kh, pk.kw, wscale=ops.scale_by_qdelta(pk.wscale, sel))
            return ops.conv2d_f16(ops.bins_to_grid(xq, sel, half=ops.f16_dma_ok(pk.cin, pk.kh, pk.kw)), pf, stride=stride, pad=pad, residual=residual)
        return ops.conv2d_f16(xn, pk, stride=stride, pad=pad, residual=residual)

    def forward(self, x: torch.Tensor, split: int = 0) -> torch.Tensor:
        if split != 0 and self.split == 0:
            # reference :310-316: the first call with a split records it and, under QMODE.QDIFF, creates the second quantizer pair
            # (input channels [split:] of the concatenated input of an up-path shortcut get quantizers of their own).  In the released
            # tree the tree rewrite never wraps skip / shortcut convs (quant_model.py:57-58), so only a hand-built QuantLayer gets here.
            self.split = split
            if QMODE.QDIFF.value in self.aq_mode:
                self.aqtizer1 = UniformAffineQuantizer(**self.aq_params)
                self.wqtizer1 = UniformAffineQuantizer(**self.wq_params)
        if not x.is_cuda:
            raise TfmqError("QuantLayer.forward: CPU tensor (the HIP kernels are the only implementation)")
        quant_act = self.use_aq and not self.disable_aq
        mode = "fp" if not self.use_wq else ("w4a8" if quant_act else "w4")
        qdiff = self.split != 0 and QMODE.QDIFF.value in self.aq_mode
        x = x.float()
        if quant_act and not self.use_wq:
            # act-only fake quant then FP weights (never used by the drivers)
            x = torch.cat([self.aqtizer(x[:, :self.split].contiguous()), self.aqtizer1(x[:, self.split:].contiguous())], dim=1) if qdiff else self.aqtizer(x)
        if self.kind == "conv2d":
            xn = ops.nchw_to_nhwc(x.contiguous())
            ph, pw_ = self.fwd_kwargs["padding"]
            stride = self.fwd_kwargs["stride"][0]
            pad = (ph, pw_, ph, pw_)
            shape_out = None
        else:
            shape_out = x.shape[:-1]
            xn = x.reshape(-1, 1, 1, x.shape[-1]).contiguous()
            stride, pad = 1, (0, 0, 0, 0)
        if qdiff and mode != "fp":
            # y = conv(cat(q(x1), q1(x2)), cat(Q(w1), Q1(w2))) + b = [conv(q(x1), Q(w1)) + b] + conv(q1(x2), Q1(w2)): two launches, the
            # second accumulating onto the first (integer sums per half, one fp32 affine map each)
            y = self._run(xn[..., :self.split].contiguous(), mode, 1, stride, pad)
            y = self._run(xn[..., self.split:].contiguous(), mode, 2, stride, pad, residual=y)
        else:
            y = self._run(xn, mode, 0, stride, pad)
        y = ops.nhwc_to_nchw(y) if self.kind == "conv2d" else y.reshape(tuple(shape_out) + (y.shape[-1],))
        return self.act_func(y)

    def set_quant_state(self, use_wq: bool = False, use_aq: bool = False) -> None:
        self.use_wq = use_wq if not self.ignore_recon else False
        self.use_aq = use_aq if not self.ignore_recon else False

    def set_running_stat(self, running_stat: bool) -> None:
        self.aqtizer.running_stat = running_stat
        if self.split != 0 and QMODE.QDIFF.value in self.aq_mode:
            self.aqtizer1.running_stat = running_stat
