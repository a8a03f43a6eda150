"""Thin typed wrappers: torch CUDA tensors (device memory only) -> C ABI calls.

Nothing here computes anything on the host or in torch; each function validates shapes,
allocates outputs with torch (allocator plumbing) and launches the HIP kernels on torch's
current stream.  There is no CPU path: a CPU tensor raises TfmqError.
"""
from __future__ import annotations

import ctypes as C
import contextvars
import os
from typing import Optional, Tuple

import torch

from ._lib import ChainDesc, ConvDesc, FfDesc, GnDesc, QSel, TfmqError
from ._lib import handle as _lib_handle

NULL = None

# operand precision of the fp32 GEMMs launched from this context (ops.gemm_precision): selects WHICH handle of the device a launch goes through
_gemm_prec = contextvars.ContextVar("tfmq_gemm_precision", default=0)


def handle(device: int = 0):
    """The tfmq_handle this context launches through on `device` (_lib.handle(device, current ops.gemm_precision))."""
    return _lib_handle(device, _gemm_prec.get())


class Arena:
    """Replayable allocation log with liveness-based reuse.  The first pass through a plan records, for every
    allocation, a block of device memory and the view handed out; later passes hand out the same views in the same
    order, so (a) the sequence of launches is legal inside a HIP stream capture (no hipMalloc) and (b) the pointers
    baked into the captured hipGraph stay owned by the plan.

    Reuse: a block whose storage is referenced by nobody but the arena (torch's storage use count back at its
    base-only value: every view of it, and every view of those views, has been dropped by the plan) is dead in
    program order and is handed out again for a later allocation of that size -- everything runs on one stream, so
    the later writer is ordered after the earlier readers.  An SD UNet forward at batch 40 then needs a few GiB
    instead of one block per intermediate (24.7 GiB)."""

    def __init__(self, reuse: bool = True):
        self.blocks = []          # uint8 base tensors
        self.log = []             # (block index, shape, dtype)
        self.cursor = 0
        self.frozen = False
        # liveness needs torch's storage use count (the hook CUDA-graph trees use); without it every allocation keeps
        # its own block, which is correct and merely larger
        self.reuse = reuse and hasattr(torch._C, "_storage_Use_Count") and os.environ.get("TFMQ_ARENA_REUSE", "1") != "0"      # (=0: diagnostics)
        self._by_size = {}        # nbytes -> [block indices]

    def rewind(self):
        self.cursor = 0
        self.frozen = len(self.log) > 0

    @staticmethod
    def _uses(base) -> int:
        return torch._C._storage_Use_Count(base.untyped_storage()._cdata)

    def _view(self, bi, shape, dtype):
        n = 1
        for s_ in shape:
            n *= s_
        nb = n * torch.empty(0, dtype=dtype).element_size()
        return self.blocks[bi][:nb].view(dtype).view(shape)

    def take(self, shape, dtype, device):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        if self.frozen:
            if self.cursor >= len(self.log):
                raise TfmqError("Arena: replay allocates more tensors than the recorded pass")
            bi, shp, dt = self.log[self.cursor]
            if shp != shape or dt != dtype:
                raise TfmqError(f"Arena: replay mismatch at #{self.cursor}: {shp}/{dt} vs {shape}/{dtype}")
        else:
            n = 1
            for s_ in shape:
                n *= s_
            need = max(256, (n * torch.empty(0, dtype=dtype).element_size() + (64 if dtype == torch.int8 else 0) + 255) // 256 * 256)
            bi = None
            if self.reuse:
                for cand in self._by_size.get(need, ()):
                    if self._uses(self.blocks[cand]) == self._base_uses:
                        bi = cand
                        break
            if bi is None:
                base = torch.empty(need, dtype=torch.uint8, device=device)
                if not self.blocks:
                    self._base_uses = self._uses(base)      # use count of a block nobody but the arena refers to
                self.blocks.append(base)
                bi = len(self.blocks) - 1
                self._by_size.setdefault(need, []).append(bi)
            self.log.append((bi, shape, dtype))
        self.cursor += 1
        return self._view(bi, shape, dtype)

    def nbytes(self):
        return sum(b.numel() for b in self.blocks)


_arena: Optional[Arena] = None


class use_arena:
    def __init__(self, arena: Optional[Arena]):
        self.arena = arena

    def __enter__(self):
        global _arena
        self.prev = _arena
        _arena = self.arena
        if self.arena is not None:
            self.arena.rewind()
        return self.arena

    def __exit__(self, *exc):
        global _arena
        _arena = self.prev
        return False


def _alloc(*shape, dtype=torch.float32, device=None):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    if _arena is not None:
        return _arena.take(shape, dtype, device)
    if dtype == torch.int8:       # the K-padded int8 operand path reads up to 32 bytes past the last pixel row (tfmq_conv_desc.w64)
        n = 1
        for s_ in shape:
            n *= int(s_)
        return torch.empty(n + 64, dtype=dtype, device=device)[:n].view(shape)
    return torch.empty(shape, dtype=dtype, device=device)


def _alloc_like(t: torch.Tensor):
    return _alloc(tuple(t.shape), dtype=t.dtype, device=t.device)


def _dev(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise TfmqError("TFMQ hot path got a CPU tensor: the HIP kernels are the only implementation "
                        "(no CPU fallback); move the tensor to an MI355X device")
    return t.device.index or 0


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(dev: int):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_contiguous():
        raise TfmqError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")


def qsel(qtable: Optional[torch.Tensor], qid: int = 0, step: Optional[torch.Tensor] = None) -> QSel:
    """qtable: float32 [n_steps, n_q, 2] (or [n_q, 2] / [2]) device tensor."""
    if qtable is None:
        return QSel(None, None, 0, 0)
    _chk(qtable, torch.float32, "qtable")
    stride = qtable.shape[-2] if qtable.dim() >= 2 else 1
    sel = QSel(qtable.data_ptr(), None if step is None else step.data_ptr(), int(stride), int(qid))
    sel._keep = (qtable, step)  # the struct only holds raw pointers: keep the tensors alive with it
    return sel


# ------------------------------------------------------------------------------ K1 / K2 / K3
def quantize_act(x: torch.Tensor, qs: QSel, level: int = 256, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    d = _dev(x)
    if x.dtype == torch.float16:       # a tensor of the fp16 activation stream
        _chk(x, torch.float16, "x")
        q = out if out is not None else _alloc(x.shape, dtype=torch.int8, device=x.device)
        handle(d).call("quantize_act_h", _p(x), _p(q), x.numel(), qs, level, _stream(d))
        return q
    _chk(x, torch.float32, "x")
    q = out if out is not None else _alloc(x.shape, dtype=torch.int8, device=x.device)
    handle(d).call("quantize_act", _p(x), _p(q), x.numel(), qs, level, _stream(d))
    return q


def fake_quant_sel(x: torch.Tensor, qs: QSel, level: int = 256, pre: float = 1.0) -> torch.Tensor:
    """delta * (clamp(rint(x * pre / delta) + zp, 0, level - 1) - zp) under the current Finite-Set group: the fake-quantised operand
    of an attention matmul whose quantizers are enabled (tfmq_fake_quant_sel)."""
    d = _dev(x)
    x = x.contiguous()
    _chk(x, torch.float32, "x")
    y = _alloc_like(x)
    handle(d).call("fake_quant_sel", _p(x), _p(y), x.numel(), qs, int(level), float(pre), _stream(d))
    return y


def attention_quant(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float, sel_q: QSel, sel_k: QSel, sel_v: QSel,
                    sel_w: QSel, w_level: int = 256, pre: float = 1.0, observe=None) -> torch.Tensor:
    """softmax(q^ k^T scale) quantised to w^, times v^ -- the attention of a block whose `use_aq` is switched on (quant_block.py:226-243,
    318-323,350-351,487-498): x^ = fake-quantised x (8 bit), w^ = the always-zero softmax quantizer (softmax_a_bit).  pre: the d^-1/4
    factor QuantQKMatMul applies to q and k BEFORE their quantizers (then scale = 1).  Exact fp32 products of the dequantised values
    (strided MFMA GEMMs) and an fp32 row softmax: the reference's arithmetic up to summation order.  observe(which, tensor): the
    calibration hook, called with the tensor each quantizer is about to see ('q', 'k', 'v', 'w')."""
    B, Tq, Cq = q.shape
    Tk, d_ = k.shape[1], Cq // heads

    def fq(which, x, sel, level, pr=1.0):
        x = x.contiguous()
        if pr != 1.0:        # the quantizer (and its calibration) sees the scaled tensor
            xs = _alloc_like(x)
            xs.zero_()
            axpy(xs, x, pr)
            x, pr = xs, 1.0
        if observe is not None:
            observe(which, x)
        return fake_quant_sel(x, sel, level, pr)
    qh, kh, vh = fq("q", q, sel_q, 256, pre), fq("k", k, sel_k, 256, pre), fq("v", v, sel_v, 256)
    S = _alloc(B, heads, Tq, Tk, dtype=torch.float32, device=q.device)
    for h in range(heads):
        gemm_strided(qh, h * d_, Cq, 1, Tq * Cq, kh, h * d_, 1, Cq, Tk * Cq, S, h * Tq * Tk, Tk, heads * Tq * Tk, Tq, Tk, d_, B)
    P = softmax_rows(S, float(scale))
    Ph = fq("w", P, sel_w, w_level)
    out = _alloc(B, Tq, Cq, dtype=torch.float32, device=q.device)
    for h in range(heads):
        gemm_strided(Ph, h * Tq * Tk, Tk, 1, heads * Tq * Tk, vh, h * d_, Cq, 1, Tk * Cq, out, h * d_, Cq, Tq * Cq, Tq, d_, Tk, B)
    return out


def attention_q8_ok(d: int, w_level: int) -> bool:
    """tfmq_attention_q8's envelope (include/tfmq_hip.h): head dim a multiple of 8 up to 160, a softmax quantizer of at most 8 bits."""
    return d % 8 == 0 and d <= 160 and 2 <= w_level <= 256


def attention_q8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float, sel_q: QSel, sel_k: QSel, sel_v: QSel,
                 sel_w: QSel, w_level: int = 256, pre: float = 1.0) -> torch.Tensor:
    """attention_quant on the int8 matrix cores (tfmq_attention_q8, csrc/attention_q8.hip): q, k, v -> their quantizers' bins
    (tfmq_quantize_act), v's transposed, both products as exact int32 sums of (bin - zero point) pairs, softmax bins from an fp32 two-pass
    softmax over the integer scores.  Same quantizers, same reference lines as attention_quant (quant_block.py:226-243, 318-323, 350-351,
    487-498); results differ from it only where a softmax value sits on a rounding boundary (tests/test_attention_q8_gpu.py)."""
    B, Tq, C = q.shape
    Tk, d_ = k.shape[1], C // heads
    dv = _dev(q)
    if not attention_q8_ok(d_, w_level):
        raise TfmqError(f"attention_q8: head dim {d_} / softmax levels {w_level} outside the kernel's envelope (ops.attention_quant takes them)")

    def bins(x, sel):
        x = x.contiguous()
        if pre != 1.0 and sel is not sel_v:
            xs = _alloc_like(x)
            xs.zero_()
            axpy(xs, x, pre)
            x = xs
        return quantize_act(x, sel)
    qb, kb, vb = bins(q, sel_q), bins(k, sel_k), bins(v, sel_v)
    Tks = (Tk + 7) // 8 * 8
    vt = _alloc(B, C, Tks, dtype=torch.int8, device=q.device)
    handle(dv).call("transpose_i8", _p(vb), _p(vt), B, Tk, C, Tks, _stream(dv))
    out = _alloc(B, Tq, C, dtype=torch.float32, device=q.device)
    handle(dv).call("attention_q8", _p(qb), _p(kb), _p(vt), C, C, sel_q, sel_k, sel_v, sel_w, int(w_level), _p(out), C, B, heads, Tq, Tk, Tks, d_,
                    float(scale), _stream(dv))
    return out


def bins_to_grid(xq: torch.Tensor, qs: QSel, half: bool = True) -> torch.Tensor:
    """int8 activation bins (quantize_act) -> (b - z_a) on their integer grid, fp16 (exact: |b - z_a| <= 255) or fp32: the
    activation operand of a W8A8 layer on the fp16-operand kernels (include/tfmq_hip.h: tfmq_bins_to_grid)."""
    d = _dev(xq)
    _chk(xq, torch.int8, "xq")
    half = half and xq.numel() % 4 == 0
    out = _alloc(*xq.shape, dtype=torch.float16 if half else torch.float32, device=xq.device)
    handle(d).call("bins_to_grid", _p(xq), qs, _p(out), int(half), xq.numel(), _stream(d))
    return out


def scale_by_qdelta(ws: torch.Tensor, qs: QSel) -> torch.Tensor:
    """[n] fp32 -> delta_a(current Finite-Set group) * ws: the per-channel output scale of a W8A8 layer."""
    d = _dev(ws)
    _chk(ws, torch.float32, "ws")
    out = _alloc(ws.numel(), dtype=torch.float32, device=ws.device)
    handle(d).call("scale_by_qdelta", _p(ws), qs, _p(out), ws.numel(), _stream(d))
    return out


def fake_quant(x: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor, level: int, want_idx: bool = False):
    """Per-tensor (delta/zp numel 1) or per-row (first dim) fake quantisation."""
    d = _dev(x)
    _chk(x, torch.float32, "x")
    rows = delta.numel()
    cols = x.numel() // rows
    dl = delta.reshape(-1).contiguous().float()
    z = zp.reshape(-1).contiguous().float()
    y = _alloc_like(x)
    idx = _alloc(x.shape, dtype=torch.uint8, device=x.device) if want_idx else None
    handle(d).call("fake_quant", _p(x), _p(y), _p(idx), rows, cols, _p(dl), _p(z), level, _stream(d))
    return (y, idx) if want_idx else y


def fake_quant_bwd(x: torch.Tensor, g: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor, level: int, want_gx: bool = True):
    """Backward of per-tensor fake_quant through the straight-through round (tfmq_fake_quant_bwd): -> (dL/dx | None, dL/ddelta [1] fp32)."""
    d = _dev(x)
    _chk(x, torch.float32, "x")
    _chk(g, torch.float32, "g")
    if g.numel() != x.numel() or delta.numel() != 1 or zp.numel() != 1:
        raise TfmqError("fake_quant_bwd: g like x, scalar delta / zero point")
    gx = _alloc_like(x) if want_gx else None
    nparts = max(1, min(1024, (x.numel() + 4095) // 4096))
    part = _alloc(nparts, dtype=torch.float64, device=x.device)
    handle(d).call("fake_quant_bwd", _p(x), _p(g), _p(gx), x.numel(), _p(delta.reshape(1).float().contiguous()), _p(zp.reshape(1).float().contiguous()),
                   int(level), _p(part), nparts, _stream(d))
    return gx, part.sum().float().reshape(1)


def minmax(x: torch.Tensor, rows: int = 1) -> torch.Tensor:
    """-> float32 [rows, 2] = {min, max} per row of x viewed as [rows, -1]."""
    d = _dev(x)
    _chk(x, torch.float32, "x")
    cols = x.numel() // rows
    h = handle(d)
    ws = _alloc(h.lib.tfmq_minmax_ws_bytes(rows, cols), dtype=torch.uint8, device=x.device)
    out = _alloc(rows, 2, dtype=torch.float32, device=x.device)
    h.call("minmax", _p(x), rows, cols, _p(out), _p(ws), _stream(d))
    return out


def minmax_to_qparam(mm: torch.Tensor, level: int, always_zero: bool = False) -> torch.Tensor:
    d = _dev(mm)
    rows = mm.shape[0]
    qp = _alloc(rows, 2, dtype=torch.float32, device=mm.device)
    handle(d).call("minmax_to_qparam", _p(mm), rows, level, int(always_zero), _p(qp), _stream(d))
    return qp


def np_histogram(x: torch.Tensor, edges, clip=None):
    """Bin counts of np.histogram(x, bins) for the equal-width edge table `edges` (numpy float32 or float64, bins + 1 entries):
    numpy's own index arithmetic in the edges' precision; clip = (lo, hi) applies np.clip in float64 first.  -> numpy int64."""
    import numpy as np
    d = _dev(x)
    _chk(x, torch.float32, "x")
    edges = np.ascontiguousarray(edges)
    if edges.dtype not in (np.float32, np.float64):
        raise TfmqError("np_histogram: edges must be float32 or float64")
    bins = edges.shape[0] - 1
    ed = torch.from_numpy(edges).to(x.device)
    counts = _alloc(bins, dtype=torch.int32, device=x.device)
    lo, hi = (0.0, 0.0) if clip is None else (float(clip[0]), float(clip[1]))
    handle(d).call("np_histogram", _p(x), x.numel(), int(edges.dtype == np.float64), int(clip is not None), C.c_double(lo), C.c_double(hi),
                   _p(ed), bins, _p(counts), _stream(d))
    return counts.cpu().numpy().astype(np.int64)


def np_histogram_rows(x: torch.Tensor, edges, clip=None):
    """np_histogram for every row of x [rows, n] at once: edges numpy [rows, bins + 1] (float32 or float64), clip = (lo [rows], hi [rows])
    numpy float64 or None.  -> numpy int64 [rows, bins]."""
    import numpy as np
    d = _dev(x)
    _chk(x, torch.float32, "x")
    rows, n = x.shape
    edges = np.ascontiguousarray(edges)
    if edges.dtype not in (np.float32, np.float64) or edges.shape[0] != rows:
        raise TfmqError("np_histogram_rows: edges must be float32 or float64 [rows, bins + 1]")
    bins = edges.shape[1] - 1
    ed = torch.from_numpy(edges).to(x.device)
    counts = _alloc(rows, bins, dtype=torch.int32, device=x.device)
    lo = hi = None
    if clip is not None:
        lo = torch.from_numpy(np.ascontiguousarray(clip[0], dtype=np.float64)).to(x.device)
        hi = torch.from_numpy(np.ascontiguousarray(clip[1], dtype=np.float64)).to(x.device)
    handle(d).call("np_histogram_rows", _p(x), rows, n, int(edges.dtype == np.float64), int(clip is not None), _p(lo), _p(hi), _p(ed), bins,
                   _p(counts), _stream(d))
    return counts.cpu().numpy().astype(np.int64)


def act_range_update(mm: torch.Tensor, state: torch.Tensor, qparam: torch.Tensor, momentum: float, level: int, init: bool):
    d = _dev(mm)
    handle(d).call("act_range_update", _p(mm), _p(state), _p(qparam), float(momentum), level, int(init), _stream(d))


def mse_search(x: torch.Tensor, rows: int, level: int, always_zero: bool = False, want_losses: bool = False):
    """80-candidate L2.4 search per row of x viewed [rows, -1] -> qparam [rows, 2] (+losses, best)."""
    d = _dev(x)
    _chk(x, torch.float32, "x")
    cols = x.numel() // rows
    h = handle(d)
    mm = minmax(x, rows)
    ws = _alloc(h.lib.tfmq_mse_ws_bytes(rows, cols), dtype=torch.uint8, device=x.device)
    qp = _alloc(rows, 2, dtype=torch.float32, device=x.device)
    losses = _alloc(rows, 80, dtype=torch.float32, device=x.device) if want_losses else None
    best = _alloc(rows, dtype=torch.int32, device=x.device) if want_losses else None
    h.call("mse_search", _p(x), rows, cols, _p(mm), level, int(always_zero), _p(qp), _p(losses), _p(best), _p(ws), _stream(d))
    return (qp, losses, best) if want_losses else qp


# ------------------------------------------------------------------------------ K4
class PackedW4:
    """int4 weights of one QuantLayer, resident on the device.

    packed : at-rest format, two weights per byte (tfmq_pack_w4)
    w8     : the conv/linear kernels' operand, int8 (q_w - z_w) tiles expanded once from `packed`
             (tfmq_expand_w4); None when cin % 32 != 0 (only the small GEMV reads such layers)."""

    def __init__(self, packed, wmeta, wscale, bias, cout, cin, kh, kw):
        self.packed, self.wmeta, self.wscale, self.bias = packed, wmeta, wscale, bias
        self.cout, self.cin, self.kh, self.kw = cout, cin, kh, kw
        self.w8 = None
        if cin % 32 == 0:
            d = _dev(packed)
            self.w8 = torch.empty((cout + 31) // 32 * 32 * kh * kw * cin, dtype=torch.int8, device=packed.device)
            handle(d).call("expand_w4", _p(packed), _p(wmeta), cout, cin, kh, kw, _p(self.w8), _stream(d))
        # Cin % 64 == 32 (the 224-channel multiples of the CelebA UNet): the K-padded operand of the LDS-DMA kernels
        self.w8p = None
        if cin % 64 == 32 and kh * kw <= 9 and os.environ.get("TFMQ_W4_KPAD", "1") != "0":
            self.w8p = torch.empty((cout + 31) // 32 * 32 * kh * kw * ((cin + 63) // 64) * 64, dtype=torch.int8, device=packed.device)
            handle(_dev(packed)).call("expand_w4_k64", _p(packed), _p(wmeta), cout, cin, kh, kw, _p(self.w8p), _stream(_dev(packed)))


def pack_w4(w: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor, alpha: Optional[torch.Tensor] = None,
            bias: Optional[torch.Tensor] = None) -> PackedW4:
    d = _dev(w)
    _chk(w, torch.float32, "w")
    if w.dim() == 2:
        cout, cin, kh, kw = w.shape[0], w.shape[1], 1, 1
    else:
        cout, cin, kh, kw = w.shape
    dl = delta.reshape(-1).contiguous().float()
    z = zp.reshape(-1).contiguous().float()
    if dl.numel() != cout or z.numel() != cout:
        raise TfmqError("pack_w4: delta/zp must have one entry per output channel")
    if float(z.max()) > 127 or float(z.min()) < -112:
        raise TfmqError("pack_w4: weight zero-point outside [-112, 127] (q - z must fit int8)")
    if alpha is not None:
        _chk(alpha, torch.float32, "alpha")
    # tile-major layout (csrc/common.hpp w4_word_index): rows padded to a multiple of 32 output channels
    packed = torch.zeros((cout + 31) // 32 * 32, kh * kw * cin // 2, dtype=torch.uint8, device=w.device)
    wmeta = _alloc(cout, 4, dtype=torch.int32, device=w.device)
    handle(d).call("pack_w4", _p(w), _p(alpha), _p(dl), _p(z), cout, cin, kh, kw, _p(packed), _p(wmeta), _stream(d))
    return PackedW4(packed, wmeta, dl, None if bias is None else bias.contiguous().float(), cout, cin, kh, kw)


def unpack_w4(pw: PackedW4) -> torch.Tensor:
    d = _dev(pw.packed)
    idx = _alloc(pw.cout, pw.cin, pw.kh, pw.kw, dtype=torch.uint8, device=pw.packed.device)
    handle(d).call("unpack_w4", _p(pw.packed), pw.cout, pw.cin, pw.kh, pw.kw, _p(idx), _stream(d))
    return idx


class PackedF16:
    """f16 weights [cout][tap][cin_pad]; wscale != None => integer grid of a weight-only quantised layer."""

    def __init__(self, w16, bias, cout, cin, kh, kw, wscale=None):
        self.w16, self.bias, self.wscale = w16, bias, wscale
        self.cout, self.cin, self.kh, self.kw = cout, cin, kh, kw


def pack_w_f16(w: torch.Tensor, bias: Optional[torch.Tensor] = None, delta: Optional[torch.Tensor] = None,
               zp: Optional[torch.Tensor] = None, alpha: Optional[torch.Tensor] = None, level: int = 16) -> PackedF16:
    d = _dev(w)
    _chk(w, torch.float32, "w")
    if w.dim() == 2:
        cout, cin, kh, kw = w.shape[0], w.shape[1], 1, 1
    else:
        cout, cin, kh, kw = w.shape
    cin_pad = (cin + 31) // 32 * 32
    out = _alloc(cout, kh * kw, cin_pad, dtype=torch.float16, device=w.device)
    dl = None if delta is None else delta.reshape(-1).contiguous().float()
    z = None if zp is None else zp.reshape(-1).contiguous().float()
    handle(d).call("pack_w_f16", _p(w), _p(alpha), _p(dl), _p(z), level, cout, cin, kh, kw, _p(out), _stream(d))
    return PackedF16(out, None if bias is None else bias.contiguous().float(), cout, cin, kh, kw, wscale=dl)


# ------------------------------------------------------------------------------ K5 / K6
_conv_prof = None  # list collecting (start_event, stop_event, algorithmic_ops, kind, algorithmic_bytes, family) per conv launch


def set_conv_profile(rec):
    """bench.py roofline leg: bracket every conv launch with HIP events on its launch stream."""
    global _conv_prof
    _conv_prof = rec


def event_elapsed_ms(e0: int, e1: int, device: int = None) -> float:
    """ms between two events of the profiling hooks (set_conv_profile / set_gemm_profile): they live on the base handle of the device the
    launches ran on (default: the current device), whatever ops.gemm_precision context created or reads them."""
    ms = C.c_float()
    _lib_handle(torch.cuda.current_device() if device is None else device).call("event_elapsed_ms", e0, e1, C.byref(ms))
    return float(ms.value)


def _event_pair(d: int):
    """Two fresh events on the base handle of device d -> (handle, id0, id1)"""
    hb = _lib_handle(d)
    e0, e1 = C.c_int(), C.c_int()
    hb.call("event_create", C.byref(e0))
    hb.call("event_create", C.byref(e1))
    return hb, e0.value, e1.value


def stats_segment(hw: int) -> int:
    """Segment size of the conv-epilogue GroupNorm statistics for an image of `hw` pixels (0 = unsupported)."""
    for s in (128, 64, 32, 16):
        if hw % s == 0:
            return s
    return 0


def _attach_stats(dsc, y: torch.Tensor, B: int, hw: int, cout: int, want_stats: bool):
    """Allocate the statistics buffer of a conv output and remember it on the tensor object."""
    if not want_stats:
        return
    seg = stats_segment(hw)
    if seg == 0 or y.shape[-1] != cout:
        return
    st = _alloc(B * hw // seg, cout, 2, dtype=torch.float32, device=y.device)
    dsc.stats, dsc.stats_seg = st.data_ptr(), seg
    y._tfmq_stats = (st, seg)


# ---- per-shape tile selection by measurement.  Which of the 128x128 / 64x64 / 256x128 tile kernels is fastest for a
# launch depends on how its tile count divides over 256 CUs x 2..5 resident blocks, i.e. on the batch size; with
# autotuning on, the first launch of every distinct shape times the eligible variants (3 launches each, HIP events on
# the launch stream; a conv is idempotent, so re-running it is harmless) and later launches -- in particular the ones
# captured into the sampler's hipGraph -- use the winner.  The result does not depend on the tile shape.
_AUTOTUNE = None      # None = off, else {shape key: tile id}
_TILE_NAMES = {1: "128x128", 2: "64x64", 3: "256x128", 4: "128x64", 5: "slab 256xN (3x3)", 6: "direct 128x128 (pointwise)",
               7: "slab 128xN (3x3, two blocks per CU)", 9: "direct 256x128 (pointwise, two blocks per CU)"}


def set_conv_autotune(cache) -> None:
    """cache: a dict to fill / reuse (shape key -> tile id), or None to switch the selection back to the library's rule."""
    global _AUTOTUNE
    _AUTOTUNE = None if os.environ.get("TFMQ_CONV_AUTOTUNE", "1") == "0" else cache   # env switch: A/B runs


class autotuned:
    """Context manager used by the engines' forward(): measure / reuse tile shapes in `cache` unless a caller (a graph
    sampler's capture) already installed its own cache."""

    def __init__(self, cache):
        self.cache, self.mine = cache, False

    def __enter__(self):
        if _AUTOTUNE is None and self.cache is not None:
            set_conv_autotune(self.cache)
            self.mine = _AUTOTUNE is not None
        return self

    def __exit__(self, *exc):
        if self.mine:
            set_conv_autotune(None)
        return False


def tile_name(v: int) -> str:
    return _TILE_NAMES.get(v & 0xff, "rule") + (f", split-K {v >> 8}" if v >> 8 else "")


def conv_autotune_report():
    return {} if _AUTOTUNE is None else {k: tile_name(v) for k, v in _AUTOTUNE.items()}


def slab_ok(dsc, bm: int = 256) -> bool:
    """Launch geometry the 3x3 slab kernel (csrc/conv_slab.hip, TFMQ_TILE_SLAB / _SLAB128) takes: 3x3 / stride 1 / pad 1, Cin % 64 == 0,
    256- (128-) pixel tiles made of whole image rows (or whole images), a slab of at most 512 (320) pixel rows."""
    hv, wv = (2 * dsc.H, 2 * dsc.W) if dsc.up2x else (dsc.H, dsc.W)      # the fused nearest-2x upsample stages upsampled rows
    f16 = bool(dsc.x_f16)              # the fp16-operand form: 32 channels per 64-byte slab row, no int8 output
    if not (dsc.KH == 3 and dsc.KW == 3 and dsc.stride == 1 and dsc.pad_t == 1 and dsc.pad_l == 1
            and (dsc.Cin % 64 == 0 or (dsc.Cin % 32 == 0 and (f16 or bool(dsc.w64)))) and dsc.Ho == hv and dsc.Wo == wv and not dsc.yt
            and dsc.out_mode in ((0, 1) if f16 else (0, 1, 3)) and dsc.Cout % 8 == 0):
        return False
    hw = hv * wv
    if hw % bm == 0 and bm % wv == 0:
        rows = (bm // wv + 2) * (wv + 2)
    elif bm % hw == 0:
        rows = (bm // hw) * (hv + 2) * (wv + 2)
    else:
        return False
    return rows <= (512 if bm == 256 else 320) and not (dsc.stats and bm % dsc.stats_seg != 0)


def _tune_conv(h, name, kind, d, dsc):
    key = (kind, dsc.B, dsc.H, dsc.W, dsc.Cin, dsc.Cout, dsc.KH, dsc.stride, dsc.up2x, dsc.out_mode, bool(dsc.residual),
           bool(dsc.stats), dsc.stats_seg, dsc.x_f16, bool(dsc.yt))
    t = _AUTOTUNE.get(key)
    if t is not None:
        return t
    if torch.cuda.is_current_stream_capturing() or dsc.Cout <= 32 or (dsc.residual and dsc.residual in (dsc.y, dsc.yq)):
        return 0                      # cannot time inside a capture / nothing to choose / not idempotent
    if kind == "f16" and dsc.x2:
        return 6                      # two sources: only the register-direct pointwise kernel reads them
    if dsc.out_mode == 4:             # TFMQ_OUT_GEGLU_Q8_FAST: only the register-direct pointwise kernel carries that epilogue -- its two tile heights
        cands = [6] + ([9] if dsc.B * dsc.Ho * dsc.Wo >= 256 * 256 and os.environ.get("TFMQ_LIN_M256", "0") == "1" else [])
        if len(cands) == 1:
            return 6
    elif kind == "f16" and dsc.x_f16 and slab_ok(dsc):
        # fp16 3x3: the slab kernel's K order differs from the tile kernels' -- one rule for every batch size; its 256- and 128-pixel
        # forms accumulate every output in the same order (bit-identical), so THAT choice may be measured
        cands = [5] + ([7] if slab_ok(dsc, 128) else [])
        if len(cands) == 1:
            return 5
    else:
        cands = None
    if cands is None:
        cands = _tile_candidates(kind, dsc)
    cands = list(cands) + _ksplit_candidates(kind, dsc)
    best, best_ms = 0, None
    e0, e1 = C.c_int(), C.c_int()
    h.call("event_create", C.byref(e0))
    h.call("event_create", C.byref(e1))
    for t in cands:
        dsc.tile, dsc.ksplit = t & 0xff, t >> 8
        h.call(name, C.byref(dsc), _stream(d))          # warm (instruction cache, clocks)
        h.call("event_record", e0.value, _stream(d))
        for _ in range(3):
            h.call(name, C.byref(dsc), _stream(d))
        h.call("event_record", e1.value, _stream(d))
        ms_ = C.c_float()
        h.call("event_elapsed_ms", e0.value, e1.value, C.byref(ms_))        # synchronises on e1; the events live on THIS device's handle
        ms = ms_.value
        if best_ms is None or ms < best_ms:
            best, best_ms = t, ms
    _AUTOTUNE[key] = best
    return best


def _ksplit_candidates(kind, dsc):
    """Split-K forms of the w4a8 tile kernels (tfmq_conv_desc.ksplit) for launches whose output grid leaves CUs idle while K is long
    (the 8x8 / 16x16 / 32x32 levels of a small-batch forward): (tile | ksplit << 8) codes.  Integer partial sums: same bits."""
    if os.environ.get("TFMQ_KSPLIT", "1") == "0" or kind != "w4a8" or dsc.out_mode == 2:
        return []
    if not (dsc.Cin % 64 == 0 or (dsc.Cin % 32 == 0 and bool(dsc.w64))) or dsc.KH * dsc.KW > 9:
        return []
    M, N = dsc.B * dsc.Ho * dsc.Wo, dsc.Cout
    nsteps = dsc.KH * dsc.KW * ((dsc.Cin + 63) // 64)
    out = []
    for tile, bm, bn in ((1, 128, 128), (4, 128, 64), (2, 64, 64)):
        if tile == 2 and dsc.stats and dsc.stats_seg > 64:
            continue
        nb = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        if nb >= 384:
            continue
        for ks in (2, 3, 4, 6, 8, 12, 16):
            if nsteps // ks >= 3 and 160 <= nb * ks <= 1280 and nb * ks * bm * bn <= (16 << 20):
                out.append(tile | (ks << 8))
    return out


def _tile_candidates(kind, dsc):
    cands = [1, 2]
    k64 = dsc.Cin % 64 == 0 or (dsc.Cin % 32 == 0 and bool(dsc.w64))      # int8 layers the LDS-DMA kernels take
    if (kind == "w4a8" and k64) or (kind == "f16" and dsc.x_f16):
        cands.append(4)
        if dsc.stride == 1 and not dsc.up2x:
            cands.append(3)
    if kind == "w4a8" and slab_ok(dsc):
        cands.append(5)
    if kind == "w4a8" and slab_ok(dsc, 128):
        cands.append(7)
    if (kind == "w4a8" and dsc.KH == 1 and dsc.KW == 1 and dsc.stride == 1 and not dsc.up2x and k64 and dsc.Cout % 4 == 0
            and dsc.out_mode in (1, 2, 3) and not dsc.rowadd and not (dsc.stats and dsc.out_mode != 1) and not (dsc.yt and dsc.residual)):
        cands.append(6)
        if (not dsc.residual and dsc.B * dsc.Ho * dsc.Wo >= 256 * 256 and not (dsc.stats and 256 % dsc.stats_seg != 0)
                and os.environ.get("TFMQ_LIN_M256", "0") == "1"):
            cands.append(9)         # the same kernel on 256 x 128 tiles (round 6; opt-in: measured within +-5 % of the 128-row form on the SD shapes,
                                    # profiles/r06_ab_lin_m256.txt -- not worth a candidate whose 3-launch timing is as noisy as its gain)
    if (kind == "f16" and dsc.x_f16 and dsc.KH == 1 and dsc.KW == 1 and dsc.stride == 1 and not dsc.up2x and dsc.Cin % 32 == 0
            and dsc.Cout % 8 == 0 and dsc.out_mode == 1 and not dsc.rowadd and not (dsc.yt and (dsc.residual or dsc.stats))):
        cands.append(6)             # the same register-direct kernel on fp16 operands (skip-connection 1x1 convs, un-quantised q|k|v)
    return cands


def _profiled_conv(name, kind, d, dsc, nops, nbytes=0.0):
    h = handle(d)
    if _AUTOTUNE is not None:
        t = _tune_conv(h, name, kind, d, dsc)
        dsc.tile, dsc.ksplit = t & 0xff, t >> 8
    if _conv_prof is None:
        h.call(name, C.byref(dsc), _stream(d))
        return
    hb, e0, e1 = _event_pair(d)
    hb.call("event_record", e0, _stream(d))
    h.call(name, C.byref(dsc), _stream(d))
    hb.call("event_record", e1, _stream(d))
    # family of the launch (bench.py's per-family roofline table): which roof binds differs between them
    if kind == "w4a8":
        fam = ("w4a8 3x3 conv" if dsc.KH == 3 else ("w4a8 GEGLU projection" if dsc.out_mode in (3, 4) else "w4a8 pointwise")) if dsc.KH in (1, 3) else "w4a8 other conv"
    else:
        fam = "fp16 3x3 conv" if dsc.KH == 3 else ("fp16 pointwise" if dsc.KH == 1 else "fp16 other conv")
    _conv_prof.append((e0, e1, nops, kind, nbytes, fam))


def _conv_desc(x, B, H, W, cin, cout, kh, kw, stride, pad_t, pad_l, Ho, Wo, up2x, y, ldy, y_coff, rowadd, residual,
               rowadd_ld=None, rowadd_step=None, rowadd_step_stride=0):
    dsc = ConvDesc()
    dsc.B, dsc.H, dsc.W, dsc.Cin = B, H, W, cin
    dsc.Cout, dsc.KH, dsc.KW, dsc.stride = cout, kh, kw, stride
    dsc.pad_t, dsc.pad_l, dsc.Ho, dsc.Wo, dsc.up2x = pad_t, pad_l, Ho, Wo, int(up2x)
    dsc.x = x.data_ptr()
    dsc.rowadd = None if rowadd is None else rowadd.data_ptr()
    dsc.rowadd_ld = cout if rowadd_ld is None else int(rowadd_ld)
    dsc.rowadd_step = None if rowadd_step is None else rowadd_step.data_ptr()
    dsc.rowadd_step_stride = int(rowadd_step_stride)
    dsc.residual = None if residual is None else residual.data_ptr()
    dsc.y = y.data_ptr()
    dsc.ldy, dsc.y_coff = ldy, y_coff
    return dsc


def out_hw(H, W, kh, kw, stride, pad_t, pad_l, pad_b, pad_r, up2x=False):
    if up2x:
        H, W = 2 * H, 2 * W
    return (H + pad_t + pad_b - kh) // stride + 1, (W + pad_l + pad_r - kw) // stride + 1


def geglu_perm(inner: int, device=None) -> torch.Tensor:
    """Row order of a GEGLU projection [2*inner, K] for the fused epilogue (TFMQ_OUT_GEGLU_Q8): every 128-row
    group holds 64 value rows followed by the 64 gate rows of the same output channels."""
    if inner % 64:
        raise TfmqError("geglu_perm: inner must be a multiple of 64")
    p = torch.arange(2 * inner, device=device)
    return (p // 128) * 64 + p % 64 + (p % 128 >= 64) * inner


def conv2d_w4a8(xq: torch.Tensor, pw: PackedW4, aq: QSel, stride: int = 1, pad: Tuple[int, int, int, int] = (0, 0, 0, 0),
                up2x: bool = False, rowadd: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, y_coff: int = 0, rowadd_ld=None, rowadd_step=None,
                rowadd_step_stride: int = 0, want_stats: bool = False, out_f16: bool = False,
                geglu_oq: Optional[QSel] = None, t_col0: Optional[int] = None, out_q8: Optional[QSel] = None,
                geglu_exact: bool = False):
    """xq: int8 NHWC [B,H,W,Cin] (bin-128).  pad = (top, left, bottom, right).  -> fp32 NHWC.
    out_f16: fp16 output (operands of the attention kernel).  geglu_oq: `pw` is a geglu_perm-ordered GEGLU projection;
    returns int8 [B,Ho,Wo,Cout/2] = quant_geglu_oq(value * gelu(gate)) - 128.
    t_col0 (with out_f16): channels >= t_col0 go TRANSPOSED into a second fp16 tensor [B, Cout - t_col0, Ho*Wo]
    (the V^T operand of attention_f16); returns (y, yt), y's channels >= t_col0 are left unwritten."""
    d = _dev(xq)
    _chk(xq, torch.int8, "xq")
    B, H, W, cin = xq.shape
    if cin != pw.cin:
        raise TfmqError(f"conv2d_w4a8: Cin mismatch {cin} vs {pw.cin}")
    Ho, Wo = out_hw(H, W, pw.kh, pw.kw, stride, pad[0], pad[1], pad[2], pad[3], up2x)
    if geglu_oq is not None:
        if out is not None or rowadd is not None or residual is not None or want_stats or out_f16 or out_q8 is not None:
            raise TfmqError("conv2d_w4a8: the GEGLU epilogue takes no other epilogue option")
        y = _alloc(B, Ho, Wo, pw.cout // 2, dtype=torch.int8, device=xq.device)
        ldy = pw.cout
    elif out_q8 is not None:
        # out_q8: the consumer's activation quantizer; returns its int8 input (bias / temb row / residual applied first)
        if out is not None or want_stats or out_f16 or y_coff:
            raise TfmqError("conv2d_w4a8: the int8 output mode takes no out / stats / fp16 / channel-offset option")
        y = _alloc(B, Ho, Wo, pw.cout, dtype=torch.int8, device=xq.device)
        ldy = pw.cout
    else:
        y = out if out is not None else _alloc(B, Ho, Wo, pw.cout, dtype=torch.float16 if out_f16 else torch.float32,
                                               device=xq.device)
        _chk(y, torch.float16 if out_f16 else torch.float32, "out")
        ldy = y.shape[-1]
    dsc = _conv_desc(xq, B, H, W, cin, pw.cout, pw.kh, pw.kw, stride, pad[0], pad[1], Ho, Wo, up2x, y, ldy, y_coff,
                     rowadd, residual, rowadd_ld, rowadd_step, rowadd_step_stride)
    if pw.w8 is None:
        raise TfmqError("conv2d_w4a8: Cin must be a multiple of 32")
    dsc.w, dsc.wmeta, dsc.wscale = pw.w8.data_ptr(), pw.wmeta.data_ptr(), pw.wscale.data_ptr()
    if getattr(pw, "w8p", None) is not None:
        # the K-padded operand makes the last K-step of a pixel read 32 bytes of the NEXT pixel (times zero weights): the buffer
        # must own those bytes.  Tensors from _alloc / the arena carry the slack; a caller-made int8 tensor that ends at its
        # storage's end does not -- those launches keep the register-staged 32-channel-step kernel.
        st = xq.untyped_storage()
        if st.nbytes() - (xq.storage_offset() + xq.numel()) * xq.element_size() >= 32:
            dsc.w64 = pw.w8p.data_ptr()
    dsc.bias = None if pw.bias is None else pw.bias.data_ptr()
    dsc.aq = aq
    osz = 4.0
    if geglu_oq is not None:
        # consumer-sized GELU (TFMQ_OUT_GEGLU_Q8_FAST, round 4) where the register-direct kernel takes the launch; TFMQ_GELU_EXACT=1 or
        # geglu_exact=True keep the 5e-7 form (TFMQ_OUT_GEGLU_Q8, bit-identical to the un-fused geglu + quantise)
        # (the predicate mirrors launch_conv_lin's acceptance test, csrc/conv_lin.hip: out_mode 4 exists on the register-direct kernel only,
        # a launch it declines must keep out_mode 2, which the tile kernels complete -- ADVICE r4)
        fast = ((not geglu_exact) and os.environ.get("TFMQ_GELU_EXACT", "0") != "1" and cin % 64 == 0 and pw.cout % 128 == 0
                and pw.kh == 1 and pw.kw == 1 and stride == 1 and not up2x and tuple(pad) == (0, 0, 0, 0) and B * H * W * cin < 2 ** 31)
        dsc.out_mode, dsc.oq, dsc.yq, dsc.y = (4 if fast else 2), geglu_oq, y.data_ptr(), None
        osz = 0.5
    elif out_q8 is not None:
        dsc.out_mode, dsc.oq, dsc.yq, dsc.y = 3, out_q8, y.data_ptr(), None
        osz = 1.0
    elif out_f16:
        dsc.out_mode = 1
        osz = 2.0
    yt = None
    if t_col0 is not None:
        if not out_f16 or t_col0 % 128 or not 0 <= t_col0 < pw.cout or (Ho * Wo) % 4:
            raise TfmqError("conv2d_w4a8: t_col0 needs out_f16, t_col0 % 128 == 0 and Ho*Wo % 4 == 0")
        yt = _alloc(B, pw.cout - t_col0, Ho * Wo, dtype=torch.float16, device=xq.device)
        dsc.yt, dsc.t_col0 = yt.data_ptr(), int(t_col0)
    _attach_stats(dsc, y, B, Ho * Wo, pw.cout, want_stats and y_coff == 0 and dsc.out_mode in (0, 1) and yt is None)
    rsz = 0.0
    if residual is not None:
        if residual.dtype not in (torch.float32, torch.float16) or not residual.is_contiguous():
            raise TfmqError("conv2d_w4a8: residual must be contiguous fp32 or fp16")
        dsc.res_f16 = int(residual.dtype == torch.float16)
        rsz = 2.0 if dsc.res_f16 else 4.0
    # algorithmic HBM bytes: int8 input once + int8 weight operand + output (+ residual)
    nbytes = B * H * W * cin + pw.cout * pw.kh * pw.kw * cin + B * Ho * Wo * pw.cout * (osz + rsz)
    _profiled_conv("conv2d_w4a8", "w4a8", d, dsc, 2.0 * B * Ho * Wo * pw.cout * pw.kh * pw.kw * cin, nbytes)
    return y if yt is None else (y, yt)


def conv2d_f16(x: torch.Tensor, pf: PackedF16, stride: int = 1, pad: Tuple[int, int, int, int] = (0, 0, 0, 0),
               up2x: bool = False, rowadd: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None, y_coff: int = 0, rowadd_ld=None, rowadd_step=None,
               rowadd_step_stride: int = 0, want_stats: bool = False, out_f16: bool = False,
               t_col0: Optional[int] = None, x2: Optional[torch.Tensor] = None):
    """x2: second fp16 source of a virtual channel concat -- the layer reads cat(x, x2) without the copy (tfmq_conv_desc.x2;
    `f16_cat_ok` says which launches take it).
    x: fp32 NHWC, or fp16 NHWC written as fp16 by its producer (groupnorm(half_out=True), to_half): the fp16 input
    takes the LDS-DMA pipeline (Cin % 32 == 0, <= 9 taps).  Un-quantised layers (f16 MFMA, fp32 accumulate).
    out_f16 / t_col0: as conv2d_w4a8 (fp16 rows, channels >= t_col0 transposed into a second tensor: the operands of
    attention_f16 from a fused q|k|v projection); returns (y, yt) when t_col0 is given."""
    d = _dev(x)
    if x.dtype not in (torch.float32, torch.float16):
        raise TfmqError("conv2d_f16: x must be fp32 or fp16")
    _chk(x, x.dtype, "x")
    B, H, W, cin = x.shape
    cin1 = cin
    if x2 is not None:
        _chk(x2, torch.float16, "x2")
        if x.dtype != torch.float16 or tuple(x2.shape[:3]) != (B, H, W) or not f16_cat_ok(cin, x2.shape[-1], pf.kh, pf.kw, stride, up2x, pad) \
                or not out_f16 or rowadd is not None or t_col0 is not None:
            raise TfmqError("conv2d_f16: x2 needs two fp16 sources of the same pixels, a pointwise layer, channel counts % 32 == 0 and fp16 output")
        cin = cin1 + x2.shape[-1]
    if cin != pf.cin:
        raise TfmqError(f"conv2d_f16: Cin mismatch {cin} vs {pf.cin}")
    Ho, Wo = out_hw(H, W, pf.kh, pf.kw, stride, pad[0], pad[1], pad[2], pad[3], up2x)
    y = out if out is not None else _alloc(B, Ho, Wo, pf.cout, dtype=torch.float16 if out_f16 else torch.float32, device=x.device)
    _chk(y, torch.float16 if out_f16 else torch.float32, "out")
    dsc = _conv_desc(x, B, H, W, cin, pf.cout, pf.kh, pf.kw, stride, pad[0], pad[1], Ho, Wo, up2x, y, y.shape[-1], y_coff,
                     rowadd, residual, rowadd_ld, rowadd_step, rowadd_step_stride)
    yt = None
    if out_f16:
        dsc.out_mode = 1
        if t_col0 is not None:
            if t_col0 % 128 or not 0 <= t_col0 < pf.cout or (Ho * Wo) % 4:
                raise TfmqError("conv2d_f16: t_col0 needs t_col0 % 128 == 0 and Ho*Wo % 4 == 0")
            yt = _alloc(B, pf.cout - t_col0, Ho * Wo, dtype=torch.float16, device=x.device)
            dsc.yt, dsc.t_col0 = yt.data_ptr(), int(t_col0)
    elif t_col0 is not None:
        raise TfmqError("conv2d_f16: t_col0 needs out_f16")
    dsc.w = pf.w16.data_ptr()
    dsc.wscale = None if pf.wscale is None else pf.wscale.data_ptr()
    dsc.bias = None if pf.bias is None else pf.bias.data_ptr()
    dsc.aq = QSel(None, None, 0, 0)
    dsc.x_f16 = int(x.dtype == torch.float16)
    if x2 is not None:
        dsc.x2, dsc.cin1 = x2.data_ptr(), int(cin1)
        dsc._keep_x2 = x2
    _attach_stats(dsc, y, B, Ho * Wo, pf.cout, want_stats and y_coff == 0 and yt is None)
    rsz = 0.0
    if residual is not None:
        if residual.dtype not in (torch.float32, torch.float16) or not residual.is_contiguous():
            raise TfmqError("conv2d_f16: residual must be contiguous fp32 or fp16")
        dsc.res_f16 = int(residual.dtype == torch.float16)
        rsz = 2.0 if dsc.res_f16 else 4.0
    nbytes = ((2.0 if x.dtype == torch.float16 else 4.0) * B * H * W * cin + 2.0 * pf.cout * pf.kh * pf.kw * cin
              + B * Ho * Wo * pf.cout * ((2.0 if out_f16 else 4.0) + rsz))
    _profiled_conv("conv2d_f16", "f16", d, dsc, 2.0 * B * Ho * Wo * pf.cout * pf.kh * pf.kw * cin, nbytes)
    return y if yt is None else (y, yt)


def slice_f16_rows(pf: PackedF16, r0: int, r1: int) -> PackedF16:
    """Output channels [r0, r1) of an fp16 layer as a layer of its own (same weight values, bias, scales)."""
    return PackedF16(pf.w16[r0:r1].contiguous(), None if pf.bias is None else pf.bias[r0:r1].contiguous(), r1 - r0, pf.cin, pf.kh, pf.kw,
                     wscale=None if pf.wscale is None else pf.wscale[r0:r1].contiguous())


def f16_cat_ok(c1: int, c2: int, kh: int = 1, kw: int = 1, stride: int = 1, up2x: bool = False, pad=(0, 0, 0, 0)) -> bool:
    """conv2d_f16(x, ..., x2=...) reads the channel concat of two fp16 tensors without its copy for these launches."""
    return (kh == 1 and kw == 1 and stride == 1 and not up2x and tuple(pad) == (0, 0, 0, 0) and c1 % 32 == 0 and c2 % 32 == 0
            and c1 > 0 and c2 > 0 and (c1 + c2) % 8 == 0)


def f16_dma_ok(cin: int, kh: int, kw: int) -> bool:
    """conv2d_f16 takes fp16 activations (LDS-DMA path) for these layer shapes."""
    return cin % 32 == 0 and kh * kw <= 9


def row_broadcast_add(x: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """x [B, T, C] (fp16 or fp32) + r [B, C] (fp32) broadcast over the tokens, fp32 arithmetic, x's dtype (tfmq_row_broadcast_add)."""
    d = _dev(x)
    if x.dtype not in (torch.float16, torch.float32) or not x.is_contiguous():
        raise TfmqError("row_broadcast_add: x must be contiguous fp16 or fp32")
    _chk(r, torch.float32, "r")
    B, T, Cc = x.shape[0], x.numel() // (x.shape[0] * x.shape[-1]), x.shape[-1]
    if tuple(r.shape) != (B, Cc) or Cc % 8:
        raise TfmqError("row_broadcast_add: r must be [B, C], C % 8 == 0")
    y = _alloc_like(x)
    handle(d).call("row_broadcast_add", _p(x), _p(r), B, T, Cc, int(x.dtype == torch.float16), _p(y), _stream(d))
    return y


def to_half(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> fp16 copy (round to nearest even) for an un-quantised conv whose producer cannot write fp16 itself."""
    d = _dev(x)
    _chk(x, torch.float32, "x")
    y = _alloc(x.shape, dtype=torch.float16, device=x.device)
    handle(d).call("f32_to_f16", _p(x), _p(y), x.numel(), _stream(d))
    return y


# ------------------------------------------------------------------------------ K7
def timestep_embedding(t: torch.Tensor, dim: int, ldm_order: bool = False) -> torch.Tensor:
    d = _dev(t)
    _chk(t, torch.float32, "t")
    emb = _alloc(t.numel(), dim, dtype=torch.float32, device=t.device)
    handle(d).call("timestep_embedding", _p(t), t.numel(), dim, int(ldm_order), _p(emb), _stream(d))
    return emb


def linear_small_f32(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], silu_in: bool = False) -> torch.Tensor:
    d = _dev(x)
    m, k = x.shape
    n = w.shape[0]
    y = _alloc(m, n, dtype=torch.float32, device=x.device)
    handle(d).call("linear_small_f32", _p(x), _p(w), _p(bias), _p(y), m, n, k, int(silu_in), _stream(d))
    return y


def linear_small_w4(x: torch.Tensor, pw: PackedW4, aq: QSel, silu_in: bool = False) -> torch.Tensor:
    d = _dev(x)
    m, k = x.shape
    y = _alloc(m, pw.cout, dtype=torch.float32, device=x.device)
    handle(d).call("linear_small_w4", _p(x), _p(pw.packed), _p(pw.wmeta), _p(pw.wscale), _p(pw.bias), aq, _p(y), m, pw.cout,
                   k, int(silu_in), _stream(d))
    return y


# ------------------------------------------------------------------------------ K8
def groupnorm(x1: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, silu: bool, aq: Optional[QSel] = None,
              x2: Optional[torch.Tensor] = None, groups: int = 32, want_f32: bool = False, want_cat: bool = False,
              half_out: bool = False):
    """x1 (and optional x2, concatenated on channels): fp32 NHWC.  Returns (yq int8 | None, yf | None, xcat | None).
    half_out: yf / xcat are written as fp16 (operands of conv2d_f16's fp16-input path, which rounds to fp16 anyway)."""
    d = _dev(x1)
    if x1.dtype not in (torch.float32, torch.float16) or not x1.is_contiguous():
        raise TfmqError("groupnorm: x1 must be contiguous fp32 or fp16")
    if x2 is not None and (x2.dtype != x1.dtype or not x2.is_contiguous()):
        raise TfmqError("groupnorm: x2 must have x1's dtype (one flag covers both halves of the virtual concat)")
    B = x1.shape[0]
    C1 = x1.shape[-1]
    HW = x1.numel() // (B * C1)
    C2 = 0 if x2 is None else x2.shape[-1]
    shape = tuple(x1.shape[:-1]) + (C1 + C2,)
    g = GnDesc()
    g.B, g.HW, g.C1, g.C2 = B, HW, C1, C2
    g.x1, g.x2 = x1.data_ptr(), (None if x2 is None else x2.data_ptr())
    g.gamma, g.beta, g.eps, g.groups, g.silu = gamma.data_ptr(), beta.data_ptr(), float(eps), groups, int(silu)
    g.x_f16 = int(x1.dtype == torch.float16)
    yq = yf = xcat = None
    if aq is not None and aq.qtable:
        g.aq = aq
        yq = _alloc(shape, dtype=torch.int8, device=x1.device)
        g.yq = yq.data_ptr()
    else:
        g.aq = QSel(None, None, 0, 0)
    odt = torch.float16 if half_out else torch.float32
    g.half_out = int(half_out)
    if want_f32 or yq is None:
        yf = _alloc(shape, dtype=odt, device=x1.device)
        g.yf = yf.data_ptr()
    if want_cat:
        xcat = _alloc(shape, dtype=odt, device=x1.device)
        g.xcat = xcat.data_ptr()
    st1 = getattr(x1, "_tfmq_stats", None)
    st2 = getattr(x2, "_tfmq_stats", None) if x2 is not None else None
    if st1 is not None and (x2 is None or (st2 is not None and st2[1] == st1[1])) and HW % st1[1] == 0:
        ws = _alloc(2 * B * (C1 + C2), dtype=torch.float32, device=x1.device)
        handle(d).call("groupnorm_from_stats", C.byref(g), _p(st1[0]), None if st2 is None else _p(st2[0]), st1[1], _p(ws),
                       _stream(d))
    else:
        handle(d).call("groupnorm", C.byref(g), _stream(d))
    return yq, yf, xcat


# ------------------------------------------------------------------------------ K9
def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, aq: Optional[QSel] = None,
              want_f32: bool = False):
    """x: fp32 [..., C] tokens.  Returns (yq int8 | None, yf fp32 | None)."""
    d = _dev(x)
    xh = x.dtype == torch.float16          # a tensor of the fp16 activation stream
    _chk(x, torch.float16 if xh else torch.float32, "x")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    quant = aq is not None and bool(aq.qtable)
    yq = _alloc(x.shape, dtype=torch.int8, device=x.device) if quant else None
    yf = _alloc(x.shape, dtype=torch.float32, device=x.device) if (want_f32 or not quant) else None
    handle(d).call("layernorm_h" if xh else "layernorm", _p(x), _p(gamma), _p(beta), float(eps), rows, Cc,
                   aq if quant else QSel(None, None, 0, 0), _p(yq), _p(yf), _stream(d))
    return yq, yf


def gn_affine_from_stats(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, groups: int = 32):
    """The statistics half of groupnorm() on its own (tfmq_gn_finalize): per-(image, channel) a, b with GroupNorm(x) = a * x + b, from the
    {sum, sum of squares} segments the producing conv's epilogue attached to x (x._tfmq_stats); None when x carries none."""
    st = getattr(x, "_tfmq_stats", None)
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    if st is None or HW % st[1] != 0:
        return None
    d = _dev(x)
    g = GnDesc()
    g.B, g.HW, g.C1, g.C2 = B, HW, Cc, 0
    g.gamma, g.beta, g.eps, g.groups = gamma.data_ptr(), beta.data_ptr(), float(eps), groups
    ab = _alloc(2, B, Cc, dtype=torch.float32, device=x.device)
    handle(d).call("gn_finalize", C.byref(g), _p(st[0]), None, st[1], _p(ab[0]), _p(ab[1]), _stream(d))
    return ab[0], ab[1]


def row_chain_ok(C_: int, M: int, T: int, gn_in: bool) -> bool:
    """Launches tfmq_row_chain takes (TFMQ_ROW_CHAIN=0: the separate launches, A/B runs)."""
    # C = 640 (two waves per 32-token group, half of K per phase): bit-identical and measured NOT faster than the launches at the 32 x 32
    # level of SD (pre chain 575 vs 572 us, mid chain 368 vs 298 us at UNet batch 128: 1.6 MB of weights streamed per 128 tokens, ten-MFMA
    # phases between barriers) -- off unless TFMQ_ROW_CHAIN_640=1
    if os.environ.get("TFMQ_ROW_CHAIN", "1") == "0" or (C_ == 640 and os.environ.get("TFMQ_ROW_CHAIN_640", "0") != "1"):
        return False
    return row_chain_supported(C_, M, T, gn_in)


def chain_tokens_ok(M: int) -> bool:
    """The engine's size policy for the token-per-lane launches (tfmq_row_chain, tfmq_ff_fused): a workgroup owns 256 tokens and walks the
    whole chain over them, so below ~half a chip's worth of workgroups the separate launches (128 x 128 tiles, split-K) spread the same
    work over more CUs -- SD at 1 image / GPU (8192 tokens, 32 workgroups): 6.66 ms per forward with the chains against 5.9 without.
    TFMQ_CHAIN_MIN_TOKENS overrides the threshold (default 32768 = 128 workgroups)."""
    return M >= int(os.environ.get("TFMQ_CHAIN_MIN_TOKENS", "32768"))


def row_chain_supported(C_: int, M: int, T: int, gn_in: bool) -> bool:
    """Shapes tfmq_row_chain takes (the policy -- which of them the engine uses -- is row_chain_ok)."""
    if C_ not in (320, 640):
        return False
    bt = 81920 // C_             # tokens per workgroup: 256 (C = 320), 128 (C = 640: two waves per 32-token group)
    return M % bt == 0 and (not gn_in or T % bt == 0)


def row_chain(x: torch.Tensor, T: int, gemms, gn=None, ln=None):
    """Token Linears (K = C = 320 or 640) chained over resident token tiles in ONE launch (tfmq_row_chain), a token per lane.
    x: [M, C] int8 bins of gemms[0]'s quantizer, or -- with gn = (a, b) per-(image, channel) GroupNorm affine [M / T, C] -- fp16 rows.
    gemms: up to 3 dicts {pw: PackedW4, aq: QSel, residual: fp16 [M, N] | None, t_col0: int | None, ln: bool}; `ln` (one at most,
    on a C-wide GEMM that is not the last) puts LayerNorm(ln = (gamma, beta, eps)) + the next GEMM's quantizer between the two.
    Returns [(y fp16 [M, N], yt fp16 [M / T, N - t_col0, T] | None), ...]; with t_col0, y's columns >= t_col0 stay unwritten."""
    d = _dev(x)
    M, Cc = x.shape
    _chk(x, torch.float16 if gn is not None else torch.int8, "x")
    if not row_chain_supported(Cc, M, T, gn is not None) or not 1 <= len(gemms) <= 3:
        raise TfmqError("row_chain: unsupported shape (token width 320 / 640, M % (81920 / C) == 0, 1-3 GEMMs)")
    dsc = ChainDesc()
    dsc.M, dsc.C, dsc.T, dsc.in_mode = M, Cc, int(T), (2 if gn is not None else 0)
    dsc.x = x.data_ptr()
    if gn is not None:
        dsc.gn_a, dsc.gn_b = gn[0].data_ptr(), gn[1].data_ptr()
    if ln is not None:
        dsc.ln_gamma, dsc.ln_beta, dsc.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
    dsc.n_gemm = len(gemms)
    outs, keep, ncol, nops, nbytes = [], [x, gn, ln], 0, 0.0, float(x.numel() * x.element_size())
    for i, gm in enumerate(gemms):
        pw = gm["pw"]
        if pw.w8 is None or pw.cin != Cc or pw.kh != 1 or pw.kw != 1 or pw.cout % 64:
            raise TfmqError("row_chain: every GEMM is a w4a8 Linear with Cin = C and Cout % 64 == 0")
        L = dsc.g[i]
        L.w, L.wmeta, L.wscale = pw.w8.data_ptr(), pw.wmeta.data_ptr(), pw.wscale.data_ptr()
        L.bias = None if pw.bias is None else pw.bias.data_ptr()
        L.N, L.aq = pw.cout, gm["aq"]
        res = gm.get("residual")
        if res is not None:
            _chk(res, torch.float16, "residual")
            L.residual = res.data_ptr()
            nbytes += 2.0 * M * pw.cout
        y = _alloc(M, pw.cout, dtype=torch.float16, device=x.device)
        L.y, L.ldy = y.data_ptr(), pw.cout
        yt = None
        if gm.get("t_col0") is not None:
            yt = _alloc(M // T, pw.cout - gm["t_col0"], T, dtype=torch.float16, device=x.device)
            L.yt, L.t_col0 = yt.data_ptr(), int(gm["t_col0"])
        L.next = int(bool(gm.get("ln")))
        outs.append((y, yt))
        keep += [pw, res, gm["aq"]]
        ncol += pw.cout
        nops += 2.0 * M * pw.cout * Cc
        nbytes += 2.0 * M * pw.cout + pw.cout * Cc
    ws = _alloc(4 * ncol * (Cc // 320), dtype=torch.float32, device=x.device)
    dsc.ws = ws.data_ptr()
    dsc._keep = keep
    h = handle(d)
    if _conv_prof is None:
        h.call("row_chain", C.byref(dsc), _stream(d))
        return outs
    hb, e0, e1 = _event_pair(d)
    hb.call("event_record", e0, _stream(d))
    h.call("row_chain", C.byref(dsc), _stream(d))
    hb.call("event_record", e1, _stream(d))
    _conv_prof.append((e0, e1, nops, "w4a8", nbytes, "w4a8 row chain (token per lane)"))
    return outs


def ff_fused_ok(C_: int, inner: int, pw1, pw2) -> bool:
    """Launches tfmq_ff_fused takes: token width 320 (the 64 x 64 level of SD v1), inner % 64 == 0, both Linears with their
    int8-expanded operands (tfmq_expand_w4).  TFMQ_FF_FUSED=0 keeps the three-launch chain (A/B runs)."""
    return (os.environ.get("TFMQ_FF_FUSED", "1") != "0" and os.environ.get("TFMQ_GELU_EXACT", "0") != "1" and C_ == 320 and inner % 64 == 0
            and pw1.w8 is not None and pw2.w8 is not None and pw1.cin == C_ and pw1.cout == 2 * inner and pw2.cin == inner and pw2.cout == C_
            and pw1.kh == pw1.kw == pw2.kh == pw2.kw == 1)


def ff_fused(x: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, eps: float, aq0: QSel, pw1: "PackedW4", aq2: QSel, pw2: "PackedW4",
             out_q8: Optional[QSel] = None, pre: Optional[dict] = None, post: Optional[dict] = None):
    """x: fp16 [..., C] tokens of the fp16 activation stream.  ff.net.2(quant(value * gelu(gate))) + x with value | gate =
    ff.net.0.proj(quant(LayerNorm(x))) in one launch (tfmq_ff_fused; `x = self.ff(self.norm3(x)) + x`, ldm/modules/attention.py:37-64,
    152-215).  pw1: the GEGLU projection packed in ops.geglu_perm row order; pw2: ff.net.2.  Returns fp16 [..., C], or with out_q8 the
    consumer quantizer's int8 bins.  Bit-identical to layernorm -> conv2d_w4a8(geglu_oq) -> conv2d_w4a8(residual=x).
    pre = {xq: int8 [..., C] bins, pw, aq, residual: fp16}: a C -> C Linear in front (attn2.to_out + residual): its fp16 output takes x's place
    (x = None) and is returned as well.  post = {pw, residual: fp16, stats: bool}: a C -> C Linear behind (proj_out + residual) on the
    feed-forward's bins under out_q8 (= that Linear's quantizer): returns its fp16 output (with `_tfmq_stats` when stats).  Both need
    M % 256 == 0.  Returns y, or (y_pre, y) with pre."""
    src = x if pre is None else pre["xq"]
    d = _dev(src)
    _chk(src, torch.float16 if pre is None else torch.int8, "x")
    Cc = src.shape[-1]
    inner = pw1.cout // 2
    if not ff_fused_ok(Cc, inner, pw1, pw2):
        raise TfmqError("ff_fused: unsupported shape (token width 320, inner % 64 == 0, w4a8 Linears)")
    M = src.numel() // Cc
    if (pre is not None or post is not None) and M % 256:
        raise TfmqError("ff_fused: pre / post need M % 256 == 0")
    if post is not None and out_q8 is None:
        raise TfmqError("ff_fused: post needs out_q8 (the Linear's activation quantizer)")
    y = _alloc(src.shape, dtype=torch.int8 if (out_q8 is not None and post is None) else torch.float16, device=src.device)
    ws = _alloc(4 * inner + 2560, dtype=torch.float32, device=src.device)
    dsc = FfDesc()
    dsc.M, dsc.C, dsc.inner = M, Cc, inner
    dsc.x = None if x is None else x.data_ptr()
    dsc.gamma, dsc.beta, dsc.eps = gamma.data_ptr(), beta.data_ptr(), float(eps)
    y_pre = None
    if pre is not None:
        pw0 = pre["pw"]
        if pw0.w8 is None or pw0.cin != Cc or pw0.cout != Cc or pw0.kh != 1 or pw0.kw != 1:
            raise TfmqError("ff_fused: pre is a C -> C w4a8 Linear")
        _chk(pre["residual"], torch.float16, "pre residual")
        y_pre = _alloc(src.shape, dtype=torch.float16, device=src.device)
        dsc.xq_pre, dsc.w0, dsc.wmeta0, dsc.wscale0 = src.data_ptr(), pw0.w8.data_ptr(), pw0.wmeta.data_ptr(), pw0.wscale.data_ptr()
        dsc.bias0 = None if pw0.bias is None else pw0.bias.data_ptr()
        dsc.aq_pre, dsc.res_pre, dsc.y_pre = pre["aq"], pre["residual"].data_ptr(), y_pre.data_ptr()
    if post is not None:
        pw3 = post["pw"]
        if pw3.w8 is None or pw3.cin != Cc or pw3.cout != Cc or pw3.kh != 1 or pw3.kw != 1:
            raise TfmqError("ff_fused: post is a C -> C w4a8 Linear")
        _chk(post["residual"], torch.float16, "post residual")
        dsc.w3, dsc.wmeta3, dsc.wscale3 = pw3.w8.data_ptr(), pw3.wmeta.data_ptr(), pw3.wscale.data_ptr()
        dsc.bias3 = None if pw3.bias is None else pw3.bias.data_ptr()
        dsc.res_post, dsc.y_post = post["residual"].data_ptr(), y.data_ptr()
        if post.get("stats"):
            hw = int(post["hw"])
            seg = stats_segment(hw)
            if seg and 256 % seg == 0:
                st = _alloc(M // seg, Cc, 2, dtype=torch.float32, device=src.device)
                dsc.stats, dsc.stats_seg = st.data_ptr(), seg
                y._tfmq_stats = (st, seg)
    dsc.aq0, dsc.aq2 = aq0, aq2
    dsc.w1, dsc.wmeta1, dsc.wscale1 = pw1.w8.data_ptr(), pw1.wmeta.data_ptr(), pw1.wscale.data_ptr()
    dsc.bias1 = None if pw1.bias is None else pw1.bias.data_ptr()
    dsc.w2, dsc.wmeta2, dsc.wscale2 = pw2.w8.data_ptr(), pw2.wmeta.data_ptr(), pw2.wscale.data_ptr()
    dsc.bias2 = None if pw2.bias is None else pw2.bias.data_ptr()
    if post is not None:
        dsc.oq, dsc.yq, dsc.y = out_q8, None, None
    elif out_q8 is not None:
        dsc.oq, dsc.yq, dsc.y = out_q8, y.data_ptr(), None
    else:
        dsc.oq, dsc.yq, dsc.y = QSel(None, None, 0, 0), None, y.data_ptr()
    dsc.ws = ws.data_ptr()
    h = handle(d)
    ret = y if pre is None else (y_pre, y)
    if _conv_prof is None:
        h.call("ff_fused", C.byref(dsc), _stream(d))
        return ret
    hb, e0, e1 = _event_pair(d)
    hb.call("event_record", e0, _stream(d))
    h.call("ff_fused", C.byref(dsc), _stream(d))
    hb.call("event_record", e1, _stream(d))
    # algorithmic bytes: the fp16 row in and out (or int8 out), both weight operands
    nl = (pre is not None) + (post is not None)
    _conv_prof.append((e0, e1, 2.0 * M * (2 * inner * Cc + inner * Cc + nl * Cc * Cc), "w4a8",
                       M * Cc * (2.0 + (1.0 if (out_q8 is not None and post is None) else 2.0) + 3.0 * (pre is not None) + 2.0 * (post is not None))
                       + 3.0 * inner * Cc + nl * Cc * Cc, "w4a8 fused feed-forward (token per lane)"))
    return ret


def geglu(hin: torch.Tensor, aq: Optional[QSel] = None, want_f32: bool = False):
    """hin: fp32 [..., 2*inner] -> x * gelu(gate) as (int8 | None, fp32 | None) of shape [..., inner]."""
    d = _dev(hin)
    _chk(hin, torch.float32, "hin")
    inner = hin.shape[-1] // 2
    rows = hin.numel() // hin.shape[-1]
    shape = tuple(hin.shape[:-1]) + (inner,)
    quant = aq is not None and bool(aq.qtable)
    yq = _alloc(shape, dtype=torch.int8, device=hin.device) if quant else None
    yf = _alloc(shape, dtype=torch.float32, device=hin.device) if (want_f32 or not quant) else None
    handle(d).call("geglu", _p(hin), rows, inner, aq if quant else QSel(None, None, 0, 0), _p(yq), _p(yf), _stream(d))
    return yq, yf


# ------------------------------------------------------------------------------ K10
def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float, aq: Optional[QSel] = None,
              want_f32: bool = True):
    """q: [B,Tq,*] view with last-dim stride 1 (may be a column slice of a fused qkv buffer); same for k, v.
    Returns (out fp32 [B,Tq,heads*d] | None, yq int8 | None)."""
    d_ = _dev(q)
    B, Tq, Cq = q.shape
    Tk = k.shape[1]
    dh = Cq // heads
    for t in (q, k, v):
        if t.dtype != torch.float32 or t.stride(-1) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
            raise TfmqError("attention: q/k/v must be fp32 [B,T,C] with unit channel stride and dense batch/token strides")
    if dh > 256:
        return _attention_wide(q, k, v, heads, scale, aq, want_f32)
    out = _alloc(B, Tq, Cq, dtype=torch.float32, device=q.device) if want_f32 else None
    yq = None
    sel = QSel(None, None, 0, 0)
    if aq is not None and aq.qtable:
        yq = _alloc(B, Tq, Cq, dtype=torch.int8, device=q.device)
        sel = aq
    handle(d_).call("attention", _p(q), _p(k), _p(v), q.stride(1), k.stride(1), v.stride(1), _p(out), Cq, _p(yq), sel, B,
                    heads, Tq, Tk, dh, float(scale), _stream(d_))
    return out, yq


def _attention_wide(q, k, v, heads: int, scale: float, aq: Optional[QSel], want_f32: bool):
    """Head dims beyond the flash kernels' register budget (cin256: ONE head of 384 / 576 / 960 channels over <= 1024
    tokens): scores, softmax and P.V as three exact-fp32 launches (strided MFMA GEMM, row softmax) -- the score matrix of
    these shapes is a few MB."""
    B, Tq, Cq = q.shape
    Tk, d = k.shape[1], Cq // heads
    lq, lk, lv = q.stride(1), k.stride(1), v.stride(1)
    S = _alloc(B, heads, Tq, Tk, dtype=torch.float32, device=q.device)
    for h in range(heads):
        gemm_strided(q, h * d, lq, 1, Tq * lq, k, h * d, 1, lk, Tk * lk,
                     S, h * Tq * Tk, Tk, heads * Tq * Tk, Tq, Tk, d, B)
    P = softmax_rows(S, float(scale))
    out = _alloc(B, Tq, Cq, dtype=torch.float32, device=q.device)
    for h in range(heads):
        gemm_strided(P, h * Tq * Tk, Tk, 1, heads * Tq * Tk, v, h * d, lv, 1, Tk * lv,
                     out, h * d, Cq, Tq * Cq, Tq, d, Tk, B)
    yq = None
    if aq is not None and aq.qtable:
        yq = quantize_act(out, aq)
    return (out if want_f32 else None), yq


def attention_f16_ok(d: int, Tk: int) -> bool:
    """Tk = keys per batch item as stored (a multiple of 8; fewer may be valid, see attention_f16(n_keys=...))."""
    dmax = 256 if os.environ.get("TFMQ_ATTN_WIDE", "1") == "0" else 384      # (256, 384]: scores over the whole head, output in 128-channel slices
    return d % 8 == 0 and d <= dmax and Tk % 8 == 0


def attention_f16(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, scale: float,
                  aq: Optional[QSel] = None, want_f32: bool = True, n_keys: Optional[int] = None):
    """q, k: fp16 [B,T,*] views with unit channel stride (column slices of the fused projection's fp16 output);
    vt: fp16 [B, heads*d, Tk] (conv2d_w4a8(..., out_f16=True, t_col0=...)).  n_keys: only the first n_keys of the Tk
    stored keys are real (context padded to a multiple of 8).  Returns (out fp32 | None, yq int8 | None)."""
    d_ = _dev(q)
    B, Tq, Cq = q.shape
    Tks = k.shape[1]
    Tk = Tks if n_keys is None else int(n_keys)
    dh = Cq // heads
    for t in (q, k):
        if t.dtype != torch.float16 or t.stride(-1) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
            raise TfmqError("attention_f16: q/k must be fp16 [B,T,C] with unit channel stride and dense batch/token strides")
    if vt.dtype != torch.float16 or not vt.is_contiguous() or tuple(vt.shape) != (B, Cq, Tks) or not 0 < Tk <= Tks:
        raise TfmqError("attention_f16: vt must be contiguous fp16 [B, heads*d, Tk]")
    out = _alloc(B, Tq, Cq, dtype=torch.float32, device=q.device) if want_f32 else None
    yq = None
    sel = QSel(None, None, 0, 0)
    if aq is not None and aq.qtable:
        yq = _alloc(B, Tq, Cq, dtype=torch.int8, device=q.device)
        sel = aq
    handle(d_).call("attention_f16", _p(q), _p(k), _p(vt), q.stride(1), k.stride(1), _p(out), Cq, _p(yq), sel, B,
                    heads, Tq, Tk, Tks, dh, float(scale), _stream(d_))
    return out, yq


# ------------------------------------------------------------------------------ K11
def ddim_update(x, eps, coef, step=None, noise=None, want_x0=False, out=None):
    """out=x is allowed (the update is elementwise)."""
    d = _dev(x)
    xn = out if out is not None else _alloc_like(x)
    x0 = _alloc_like(x) if want_x0 else None
    handle(d).call("ddim_update", _p(x), _p(eps), _p(noise), _p(xn), _p(x0), x.numel(), _p(coef), _p(step), _stream(d))
    return (xn, x0) if want_x0 else xn


def dpm_x0(x: torch.Tensor, eps: torch.Tensor, sigma: float, alpha: float) -> torch.Tensor:
    d = _dev(x)
    o = _alloc_like(x)
    handle(d).call("dpm_x0", _p(x), _p(eps), float(sigma), float(alpha), _p(o), x.numel(), _stream(d))
    return o


def dpm_update(order: int, x, m0, m1, c_x: float, c_m: float, c_d: float = 0.0, inv_r0: float = 0.0) -> torch.Tensor:
    d = _dev(x)
    o = _alloc_like(x)
    handle(d).call("dpm_update", int(order), _p(x), _p(m0), _p(m1), float(c_x), float(c_m), float(c_d), float(inv_r0), _p(o),
                   x.numel(), _stream(d))
    return o


def cfg_combine(eps_u: torch.Tensor, eps_c: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """e_u + scale * (e_c - e_u), the classifier-free-guidance combine in the reference's operation order."""
    d = _dev(eps_u)
    o = out if out is not None else _alloc_like(eps_u)
    handle(d).call("cfg_combine", _p(eps_u), _p(eps_c), float(scale), _p(o), eps_u.numel(), _stream(d))
    return o


def plms_combine(order: int, e0, e1, e2=None, e3=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Adams-Bashforth eps combination of p_sample_plms (order 1 = pseudo improved Euler with e1 = eps at t_next)."""
    d = _dev(e0)
    o = out if out is not None else _alloc_like(e0)
    handle(d).call("plms_combine", int(order), _p(e0), _p(e1), _p(e2), _p(e3), _p(o), e0.numel(), _stream(d))
    return o


def ddim_update_cfg(x, eps_u, eps_c, scale: float, coef, step=None, noise=None, want_x0=False, out=None):
    """Classifier-free-guidance combine + DDIM update in one pass (out=x allowed)."""
    d = _dev(x)
    xn = out if out is not None else _alloc_like(x)
    x0 = _alloc_like(x) if want_x0 else None
    handle(d).call("ddim_update_cfg", _p(x), _p(eps_u), _p(eps_c), float(scale), _p(noise), _p(xn), _p(x0), x.numel(), _p(coef),
                   _p(step), _stream(d))
    return (xn, x0) if want_x0 else xn


def silu(x: torch.Tensor) -> torch.Tensor:
    d = _dev(x)
    _chk(x, torch.float32, "x")
    y = _alloc_like(x)
    handle(d).call("silu", _p(x), _p(y), x.numel(), _stream(d))
    return y


def step_advance(step: torch.Tensor, delta: int = 1):
    d = _dev(step)
    handle(d).call("step_advance", _p(step), delta, _stream(d))


def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    d = _dev(x)
    _chk(x, torch.float32, "x")
    B, Cc, H, W = x.shape
    y = _alloc(B, H, W, Cc, dtype=torch.float32, device=x.device)
    handle(d).call("nchw_to_nhwc", _p(x), _p(y), B, Cc, H * W, _stream(d))
    return y


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    d = _dev(x)
    _chk(x, torch.float32, "x")
    B, H, W, Cc = x.shape
    y = _alloc(B, Cc, H, W, dtype=torch.float32, device=x.device)
    handle(d).call("nhwc_to_nchw", _p(x), _p(y), B, Cc, H * W, _stream(d))
    return y


# ------------------------------------------------------------------------------ K12-K14
def adaround_init(w: torch.Tensor, delta: torch.Tensor) -> torch.Tensor:
    d = _dev(w)
    rows = delta.numel()
    alpha = _alloc_like(w)
    handle(d).call("adaround_init", _p(w), _p(delta.reshape(-1).contiguous()), _p(alpha), rows, w.numel() // rows, _stream(d))
    return alpha


def adaround_soft_fwd(w, alpha, delta, zp, level: int, hard: bool = False) -> torch.Tensor:
    d = _dev(w)
    rows = delta.numel()
    w_hat = _alloc_like(w)
    handle(d).call("adaround_soft_fwd", _p(w), _p(alpha), _p(delta.reshape(-1).contiguous()), _p(zp.reshape(-1).contiguous()),
                   _p(w_hat), rows, w.numel() // rows, level, int(hard), _stream(d))
    return w_hat


def adaround_bwd_adam(w, alpha, delta, zp, g_what, m, v, level: int, w_reg: float, b_temp: float, lr: float, t: int,
                      round_loss: Optional[torch.Tensor] = None):
    d = _dev(w)
    rows = delta.numel()
    handle(d).call("adaround_bwd_adam", _p(w), _p(alpha), _p(delta.reshape(-1).contiguous()), _p(zp.reshape(-1).contiguous()),
                   _p(g_what), _p(m), _p(v), rows, w.numel() // rows, level, float(w_reg), float(b_temp), float(lr), int(t),
                   _p(round_loss), _stream(d))


def adaround_scalars(w_reg: float, b_temp: float, lr: float, t: int):
    """The four per-iteration scalars of the fused AdaRound-backward + Adam kernel as the library computes them for tfmq_adaround_bwd_adam:
    [w_reg, b_temp, lr / (1 - 0.9^t), sqrt(1 - 0.999^t)] (fp32).  Host function; feeds adaround_bwd_adam(..., dyn=...)."""
    from ._lib import load
    out = (C.c_float * 4)()
    rc = load().tfmq_adaround_scalars(float(w_reg), float(b_temp), float(lr), int(t), out)
    if rc != 0:
        raise TfmqError(f"tfmq_adaround_scalars failed ({rc})")
    return [out[0], out[1], out[2], out[3]]


def adaround_bwd_adam_dyn(w, alpha, delta, zp, g_what, m, v, level: int, scalars: torch.Tensor, round_loss: Optional[torch.Tensor] = None):
    """adaround_bwd_adam with its per-iteration scalars read from the device tensor `scalars` [4] (adaround_scalars): capturable."""
    d = _dev(w)
    _chk(scalars, torch.float32, "scalars")
    rows = delta.numel()
    handle(d).call("adaround_bwd_adam_dyn", _p(w), _p(alpha), _p(delta.reshape(-1).contiguous()), _p(zp.reshape(-1).contiguous()),
                   _p(g_what), _p(m), _p(v), rows, w.numel() // rows, level, _p(scalars), _p(round_loss), _stream(d))


# ------------------------------------------------------------------------------ K15 (reconstruction fwd/bwd pieces)
GEMM_PRECISIONS = {"f32": 0, "bf16x3": 1, "f16": 2}


class gemm_precision:
    """with ops.gemm_precision("bf16x3"): the launches inside go through the device's bf16x3 HANDLE (_lib.handle(device, 1): its precision was
    set once when it was created), so the fp32 GEMMs run their matrix-core path on split-bf16 operands; outside, and on every other thread /
    context, launches keep the exact-fp32 handle -- no handle's state is toggled.  `device` is accepted for the callers that name it; the
    selection itself is per context (a contextvars.ContextVar), the handle is looked up per launch from the tensor's device.  Used by the
    AdaRound reconstruction iterations (TFMQ_RECON_GEMM)."""

    def __init__(self, mode: str, device: int = None):
        if mode not in GEMM_PRECISIONS:
            raise TfmqError(f"gemm_precision: {mode!r} is not one of {sorted(GEMM_PRECISIONS)}")
        self.mode, self.dev = GEMM_PRECISIONS[mode], device

    def __enter__(self):
        if self.mode and torch.cuda.is_available():
            _lib_handle(torch.cuda.current_device() if self.dev is None else self.dev, self.mode)    # created outside any stream capture
        self._tok = _gemm_prec.set(self.mode)
        return self

    def __exit__(self, *exc):
        _gemm_prec.reset(self._tok)
        return False


_gemm_prof = None  # {"rec": [(start_event, stop_event, flops, (M, N, K, batch))], "every", "cap", "n"}: bench.py's calibration roofline


def set_gemm_profile(rec, every: int = 1, cap: int = 2048):
    """bench.py `--workload cali`: bracket every `every`-th tfmq_gemm_f32 launch (at most `cap` of them: the handle keeps its events) with HIP
    events on the launch stream.  rec = None switches it off."""
    global _gemm_prof
    _gemm_prof = None if rec is None else {"rec": rec, "every": max(1, int(every)), "cap": int(cap), "n": 0}


def _gemm_call(d, name, flops, shape, *args):
    h = handle(d)
    P = _gemm_prof
    if P is None or torch.cuda.is_current_stream_capturing():      # (an event recorded inside a capture is a graph node, not a timestamp)
        h.call(name, *args)
        return
    P["n"] += 1
    if P["n"] % P["every"] or len(P["rec"]) >= P["cap"]:
        h.call(name, *args)
        return
    hb, e0, e1 = _event_pair(d)   # the events belong to the device's base handle (event_elapsed_ms reads them there), whatever handle launches
    hb.call("event_record", e0, args[-1])
    h.call(name, *args)
    hb.call("event_record", e1, args[-1])
    P["rec"].append((e0, e1, flops, shape))


def gemm(A: torch.Tensor, B: torch.Tensor, trans_a: bool = False, trans_b: bool = False, alpha: float = 1.0,
         bias=None, rowadd=None, rows_per_img: int = 1, residual=None, out=None, accumulate: bool = False) -> torch.Tensor:
    """C = alpha * op(A) @ op(B) (+bias[n]) (+rowadd[m // rows_per_img]) (+residual); A,B: contiguous fp32
    2-D or batched 3-D; exact fp32 FMA accumulation."""
    d = _dev(A)
    _chk(A, torch.float32, "A")
    _chk(B, torch.float32, "B")
    batched = A.dim() == 3
    a2, b2 = (A.shape[-2], A.shape[-1]), (B.shape[-2], B.shape[-1])
    M, K = (a2[1], a2[0]) if trans_a else a2
    Kb, N = (b2[1], b2[0]) if trans_b else b2
    if K != Kb:
        raise TfmqError(f"gemm: inner dims differ ({K} vs {Kb})")
    nb = A.shape[0] if batched else 1
    sam, sak = (1, a2[1]) if trans_a else (a2[1], 1)
    sbk, sbn = (1, b2[1]) if trans_b else (b2[1], 1)
    shape = (nb, M, N) if batched else (M, N)
    Cm = out if out is not None else _alloc(shape, dtype=torch.float32, device=A.device)
    _gemm_call(d, "gemm_f32", 2.0 * M * N * K * nb, (M, N, K, nb), _p(A), _p(B), _p(Cm), M, N, K, sam, sak, sbk, sbn, N, nb, a2[0] * a2[1] if batched else 0,
               (b2[0] * b2[1] if B.dim() == 3 else 0), M * N if batched else 0, float(alpha), _p(bias), _p(rowadd),
               int(rows_per_img), 0 if rowadd is None else rowadd.shape[-1], _p(residual), int(accumulate), _stream(d))
    return Cm


def gemm_strided(A: torch.Tensor, a_off: int, sam: int, sak: int, bsa: int, B: torch.Tensor, b_off: int, sbk: int, sbn: int,
                 bsb: int, Cm: torch.Tensor, c_off: int, scm: int, bsc: int, M: int, N: int, K: int, batch: int = 1,
                 alpha: float = 1.0, accumulate: bool = False, heads: int = 1, hsa: int = 0, hsb: int = 0,
                 hsc: int = 0) -> torch.Tensor:
    """C[z](m,n) (+)= alpha * sum_k A[z](m,k) B[z](k,n) on views of fp32 tensors described by element offsets and strides:
    A(z,m,k) = A.flat[a_off + z*bsa + m*sam + k*sak], likewise B(k,n) and C(m,n) (row stride scm, unit column stride).
    Lets the multi-head attention of a reconstruction unit run on the [B,T,heads*d] layout without permutes.
    heads > 1: a second batch level, item (z, hd) at the additional offsets hd*hsa / hd*hsb / hd*hsc (all heads of one
    attention product in one launch)."""
    d = _dev(A)
    for t in (A, B, Cm):       # views are welcome: the strides are given explicitly
        if not t.is_cuda or t.dtype != torch.float32:
            raise TfmqError("gemm_strided: operands must be fp32 device tensors")
    if heads > 1:
        _gemm_call(d, "gemm_f32_heads", 2.0 * M * N * K * batch * heads, (M, N, K, batch * heads), A.data_ptr() + 4 * a_off, B.data_ptr() + 4 * b_off,
                   Cm.data_ptr() + 4 * c_off, M, N, K, sam, sak, sbk, sbn, scm, batch, bsa, bsb, bsc, heads, hsa, hsb, hsc, float(alpha),
                   int(accumulate), _stream(d))
        return Cm
    _gemm_call(d, "gemm_f32", 2.0 * M * N * K * batch, (M, N, K, batch), A.data_ptr() + 4 * a_off, B.data_ptr() + 4 * b_off, Cm.data_ptr() + 4 * c_off,
               M, N, K, sam, sak, sbk, sbn, scm, batch, bsa, bsb, bsc, float(alpha), None, None, 1, 0, None, int(accumulate),
               _stream(d))
    return Cm


def attention_f32_ok(T: int, L: int, d: int) -> bool:
    return d in (32, 40, 64, 80) and T % 32 == 0 and L >= 1


def attention_f32_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float):
    """Exact-fp32 fused attention (reconstruction units).  q [B,T,C], k / v [B,L,C] contiguous -> (o [B,T,C],
    lse [B,heads,T] in the exp2 domain)."""
    d_ = _dev(q)
    for t in (q, k, v):
        _chk(t, torch.float32, "q/k/v")
    B, T, Cc = q.shape
    L = k.shape[1]
    o = _alloc_like(q)
    lse = _alloc(B, heads, T, dtype=torch.float32, device=q.device)
    handle(d_).call("attention_f32_fwd", _p(q), _p(k), _p(v), Cc, k.shape[2], _p(o), Cc, _p(lse), B, heads, T, L, Cc // heads,
                    float(scale), _stream(d_))
    return o, lse


def attention_f32_bwd(q, k, v, o, lse, g_o, heads: int, scale: float):
    """-> (dQ, dK, dV) of attention_f32_fwd for the upstream gradient g_o [B,T,C]."""
    d_ = _dev(q)
    B, T, Cc = q.shape
    L = k.shape[1]
    dq, dk, dv = _alloc_like(q), _alloc_like(k), _alloc_like(v)
    ws = _alloc(B, heads, T, dtype=torch.float32, device=q.device)
    _chk(g_o, torch.float32, "g_o")
    handle(d_).call("attention_f32_bwd", _p(q), _p(k), _p(v), Cc, k.shape[2], _p(o), _p(g_o), Cc, _p(lse), _p(ws), _p(dq), _p(dk),
                    _p(dv), B, heads, T, L, Cc // heads, float(scale), _stream(d_))
    return dq, dk, dv


def layernorm_bwd(x: torch.Tensor, gy: torch.Tensor, gamma: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    d = _dev(x)
    _chk(x, torch.float32, "x")
    _chk(gy, torch.float32, "gy")
    Cc = x.shape[-1]
    gx = _alloc_like(x)
    handle(d).call("layernorm_bwd", _p(x), _p(gy), _p(gamma), float(eps), x.numel() // Cc, Cc, _p(gx), _stream(d))
    return gx


def geglu_bwd(hin: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """hin [..., 2*inner], dy [..., inner] -> d hin."""
    d = _dev(hin)
    _chk(hin, torch.float32, "hin")
    _chk(dy, torch.float32, "dy")
    inner = hin.shape[-1] // 2
    dh = _alloc_like(hin)
    handle(d).call("geglu_bwd", _p(hin), _p(dy), hin.numel() // (2 * inner), inner, _p(dh), _stream(d))
    return dh


def im2col(x: torch.Tensor, kh: int, kw: int, stride: int = 1, pad=(0, 0, 0, 0)) -> torch.Tensor:
    d = _dev(x)
    _chk(x, torch.float32, "x")
    B, H, W, Cc = x.shape
    Ho, Wo = out_hw(H, W, kh, kw, stride, pad[0], pad[1], pad[2], pad[3])
    col = _alloc(B * Ho * Wo, kh * kw * Cc, dtype=torch.float32, device=x.device)
    handle(d).call("im2col", _p(x), _p(col), B, H, W, Cc, kh, kw, stride, pad[0], pad[1], Ho, Wo, _stream(d))
    return col


def im2col_f16(x: torch.Tensor, kh: int, kw: int, pad_t: int, pad_l: int, kp: int) -> torch.Tensor:
    """fp32 NHWC -> fp16 [B, H, W, kp] rows of (tap, channel) values, zero beyond kh*kw*C and outside the image (stride 1,
    output size = input size): the A operand of a narrow-input conv run as a pointwise GEMM (tfmq_im2col_f16)."""
    d = _dev(x)
    _chk(x, torch.float32, "x")
    B, H, W, Cc = x.shape
    col = _alloc(B, H, W, kp, dtype=torch.float16, device=x.device)
    handle(d).call("im2col_f16", _p(x), _p(col), B, H, W, Cc, kh, kw, pad_t, pad_l, kp, _stream(d))
    return col


def narrow_conv_as_gemm(pf: PackedF16) -> Optional[PackedF16]:
    """The pointwise layer that computes a narrow-input conv from im2col_f16 rows: weights [cout][(tap, ci)] zero-padded to a
    multiple of 32 values -- the same fp16 weight values, the same fp32 accumulation, another K order.  None if the layer is not
    a narrow-input conv (kh*kw*cin <= 64, more than one tap)."""
    kk = pf.kh * pf.kw
    if kk == 1 or kk * pf.cin > 64 or pf.cout % 8:
        return None
    kp = (kk * pf.cin + 31) // 32 * 32
    w = torch.zeros(pf.cout, 1, kp, dtype=torch.float16, device=pf.w16.device)
    w[:, 0, :kk * pf.cin] = pf.w16[:, :, :pf.cin].reshape(pf.cout, kk * pf.cin)
    return PackedF16(w.contiguous(), pf.bias, pf.cout, kp, 1, 1, wscale=pf.wscale)


def narrow_out_conv_as_gemm(pf: PackedF16) -> Optional[PackedF16]:
    """The pointwise layer whose kh*kw*cout outputs are the per-tap partial sums of a conv with a handful of output channels
    (conv_out, out.2: 320 -> 4): row tap*cout + co holds w[co][tap][:].  None if the layer is not one (cout <= 4, more than one tap,
    fp16-input shape).  No bias (tap_gather_sum adds it)."""
    kk = pf.kh * pf.kw
    if kk == 1 or pf.cout > 4 or not f16_dma_ok(pf.cin, 1, 1):
        return None
    n9 = kk * pf.cout
    n9p = (n9 + 7) // 8 * 8
    cin_pad = pf.w16.shape[-1]
    w = torch.zeros(n9p, 1, cin_pad, dtype=torch.float16, device=pf.w16.device)
    w[:n9, 0] = pf.w16.permute(1, 0, 2).reshape(n9, cin_pad)
    ws = None
    if pf.wscale is not None:
        ws = torch.ones(n9p, dtype=torch.float32, device=pf.w16.device)
        ws[:n9] = pf.wscale.reshape(1, pf.cout).repeat(kk, 1).reshape(-1)
    return PackedF16(w.contiguous(), None, n9p, pf.cin, 1, 1, wscale=ws)


def tap_gather_sum(y9: torch.Tensor, kh: int, kw: int, cout: int, pad_t: int, pad_l: int, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """y9 fp32 [B, H, W, ld] per-tap partial sums -> fp32 [B, H, W, cout] (tfmq_tap_gather_sum)."""
    d = _dev(y9)
    _chk(y9, torch.float32, "y9")
    B, H, W, ld = y9.shape
    out = _alloc(B, H, W, cout, dtype=torch.float32, device=y9.device)
    handle(d).call("tap_gather_sum", _p(y9), B, H, W, kh, kw, cout, ld, pad_t, pad_l, _p(bias), _p(out), _stream(d))
    return out


def col2im(dcol: torch.Tensor, shape, kh: int, kw: int, stride: int = 1, pad=(0, 0, 0, 0)) -> torch.Tensor:
    d = _dev(dcol)
    B, H, W, Cc = shape
    Ho, Wo = out_hw(H, W, kh, kw, stride, pad[0], pad[1], pad[2], pad[3])
    dx = _alloc(B, H, W, Cc, dtype=torch.float32, device=dcol.device)
    handle(d).call("col2im", _p(dcol), _p(dx), B, H, W, Cc, kh, kw, stride, pad[0], pad[1], Ho, Wo, _stream(d))
    return dx


def w_relayout(w: torch.Tensor, cout: int, cin: int, kh: int, kw: int, to_gemm: bool) -> torch.Tensor:
    """OIHW <-> [cout, kh*kw*cin] (the im2col column order)."""
    d = _dev(w)
    out = _alloc((cout, kh * kw * cin) if to_gemm else (cout, cin, kh, kw), dtype=torch.float32, device=w.device)
    handle(d).call("w_relayout", _p(w), _p(out), cout, cin, kh, kw, 0 if to_gemm else 1, _stream(d))
    return out


def silu_bwd(x: torch.Tensor, gy: torch.Tensor) -> torch.Tensor:
    d = _dev(x)
    gx = _alloc_like(x)
    handle(d).call("silu_bwd", _p(x), _p(gy), _p(gx), x.numel(), _stream(d))
    return gx


def groupnorm_bwd(x: torch.Tensor, gy: torch.Tensor, gamma, beta, eps: float, silu: bool, groups: int = 32) -> torch.Tensor:
    d = _dev(x)
    B, Cc = x.shape[0], x.shape[-1]
    gx = _alloc_like(x)
    handle(d).call("groupnorm_bwd", _p(x), _p(gy), _p(gamma), _p(beta), _p(gx), B, x.numel() // (B * Cc), Cc, groups, float(eps),
                   int(silu), _stream(d))
    return gx


def softmax_rows(S: torch.Tensor, scale: float) -> torch.Tensor:
    d = _dev(S)
    P = _alloc_like(S)
    handle(d).call("softmax_rows", _p(S), _p(P), S.numel() // S.shape[-1], S.shape[-1], float(scale), _stream(d))
    return P


def softmax_bwd_rows(P: torch.Tensor, dP: torch.Tensor, scale: float) -> torch.Tensor:
    d = _dev(P)
    dS = _alloc_like(P)
    handle(d).call("softmax_bwd_rows", _p(P), _p(dP), _p(dS), P.numel() // P.shape[-1], P.shape[-1], float(scale), _stream(d))
    return dS


def upsample2x(x: torch.Tensor) -> torch.Tensor:
    d = _dev(x)
    _chk(x, torch.float32, "x")
    B, H, W, Cc = x.shape
    y = _alloc(B, 2 * H, 2 * W, Cc, dtype=torch.float32, device=x.device)
    handle(d).call("upsample2x", _p(x), _p(y), B, H, W, Cc, _stream(d))
    return y


def axpy(y: torch.Tensor, x: torch.Tensor, a: float = 1.0):
    d = _dev(y)
    handle(d).call("axpy", _p(y), _p(x), float(a), y.numel(), _stream(d))
    return y


def recon_loss(pred, tgt, denom: int, want_grad: bool = True):
    d = _dev(pred)
    loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
    g = _alloc_like(pred) if want_grad else None
    handle(d).call("recon_loss", _p(pred), _p(tgt), _p(g), pred.numel(), int(denom), _p(loss), _stream(d))
    return loss, g


# ------------------------------------------------------------------------------ Fisher-weighted reconstruction
def upsample2x_bwd(g: torch.Tensor) -> torch.Tensor:
    """Backward of upsample2x: g [B,2H,2W,C] -> [B,H,W,C]."""
    d = _dev(g)
    _chk(g, torch.float32, "g")
    B, H2, W2, Cc = g.shape
    gx = _alloc(B, H2 // 2, W2 // 2, Cc, dtype=torch.float32, device=g.device)
    handle(d).call("upsample2x_bwd", _p(g), _p(gx), B, H2 // 2, W2 // 2, Cc, _stream(d))
    return gx


def kl_softmax_grad(out_q: torch.Tensor, out_fp: torch.Tensor, want_loss: bool = False, wrt_target: bool = False):
    """GetLayerGrad's loss (reference quant/data_utill.py:246-247) on NHWC model outputs [B,H,W,C]: kl_div(log_softmax(out_q, C),
    softmax(out_fp, C), 'batchmean').  -> (dL/d out_q -- or, wrt_target, dL/d out_fp through the un-detached target --, loss [1] | None)."""
    d = _dev(out_q)
    _chk(out_q, torch.float32, "out_q")
    _chk(out_fp, torch.float32, "out_fp")
    if out_q.shape != out_fp.shape:
        raise TfmqError("kl_softmax_grad: shape mismatch")
    g = _alloc_like(out_q)
    loss = torch.zeros(1, dtype=torch.float32, device=out_q.device) if want_loss else None
    Cc = out_q.shape[-1]
    handle(d).call("kl_softmax_grad", _p(out_q), _p(out_fp), _p(g), out_q.numel() // Cc, Cc, out_q.shape[0], int(wrt_target), _p(loss), _stream(d))
    return g, loss


FISHER_DIAG, FISHER_FULL = 1, 2


def fisher_loss(pred: torch.Tensor, tgt: torch.Tensor, fgrad: torch.Tensor, mode: int, denom: int, want_grad: bool = True):
    """LossFunc's RLOSS.FISHER_DIAG / FISHER_FULL (reference quant/reconstruction_util.py:53-59) on [N, ...] tensors with the cached
    Fisher weights `fgrad` (|dL/d out| + 1).  -> (loss [1], d loss / d pred | None)."""
    d = _dev(pred)
    for t, n in ((pred, "pred"), (tgt, "tgt"), (fgrad, "fgrad")):
        _chk(t, torch.float32, n)
    if pred.shape != tgt.shape or pred.shape != fgrad.shape:
        raise TfmqError("fisher_loss: pred / tgt / fgrad shapes differ")
    N = pred.shape[0]
    loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
    g = _alloc_like(pred) if want_grad else None
    dot = torch.empty(N, dtype=torch.float64, device=pred.device) if mode == FISHER_FULL else None
    handle(d).call("fisher_loss", _p(pred), _p(tgt), _p(fgrad), _p(g), N, pred.numel() // N, int(mode), int(denom), _p(dot), _p(loss), _stream(d))
    return loss, g
