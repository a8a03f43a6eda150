"""dL/d(unit output) of the whole model for the Fisher-weighted reconstruction losses (reference quant/data_utill.py:54-73,191-256:
`save_grad` / `GetLayerGrad`; quant/reconstruction_util.py:53-59: `RLOSS.FISHER_DIAG` / `FISHER_FULL`).

The reference runs the FP model and the model "quantised till" the unit, takes
    loss = F.kl_div(F.log_softmax(out_q, dim=1), F.softmax(out_fp, dim=1), reduction='batchmean')
and lets autograd carry dL/d out_q back to the unit's output (a backward hook on the unit).  Everything downstream of the unit is
un-quantised in that state, so the backward is the backward of the FP tail of the UNet.

Here the engine runs that forward in its exact-fp32 mode (every conv / Linear an im2col + fp32 MFMA GEMM, GroupNorm / LayerNorm /
GEGLU / attention in fp32) with a TAPE: from the moment the unit's output exists, every launch whose input depends on it records
its hand-written backward (the kernels of the reconstruction units: GEMM with the transposed operand, col2im, groupnorm_bwd,
layernorm_bwd, geglu_bwd, the attention backward, upsample2x_bwd).  `GradTape.backward` replays the records in reverse and
accumulates gradients per tensor (skip connections and residual adds are tensors with several consumers)."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

from .. import ops
from .._lib import TfmqError


def _key(t: torch.Tensor):
    return (t.data_ptr(), t.numel())


def _root(t: torch.Tensor):
    """(the tensor a column slice was taken from, first column, columns of the root's rows).  Contiguous tensors (and reshapes of
    them) are their own root."""
    b = t._base if t._base is not None else t
    if t.is_contiguous() and t.numel() == b.numel():
        return t, 0, t.shape[-1]
    if not b.is_contiguous() or t.stride(-1) != 1:
        raise TfmqError("GradTape: only column slices of contiguous tensors are differentiable views")
    if t.is_contiguous() and t.shape[-1] == b.shape[-1]:
        # (is_contiguous() ignores size-1 dimensions: kv[..., :C] of a [1, 1, 2C] fused k|v row is "contiguous and smaller than its base"
        # too -- that one is a column slice and falls through; what is refused is a slice that keeps the base's row width)
        # a contiguous view that is SMALLER than its base: a slice of leading (batch) rows, base[:B] or base[B:].  Its gradient has the
        # slice's shape, not the base's -- recording it under the base's key would mis-accumulate when the producer is replayed.
        raise TfmqError("GradTape: a batch (leading-dimension) slice of a taped tensor is not a differentiable view here; tape the un-sliced "
                        "tensor (the guidance-pair prefix is switched off while taps are recorded)")
    rowlen = t.stride(-2) if (t.dim() >= 2 and t.shape[-2] > 1) else b.shape[-1]      # (a single row's stride is arbitrary: take the base's width)
    off = (t.data_ptr() - b.data_ptr()) // t.element_size()
    if off >= rowlen:
        raise TfmqError("GradTape: unsupported view")
    return b, off, rowlen


class GradTape(dict):
    """Passed as the `taps` dictionary of an engine forward: when the engine records the unit `name`, its output becomes the leaf
    and recording starts.  tape.backward(out, g_out) -> dL/d leaf."""

    def __init__(self, name: str):
        super().__init__()
        self.stop = None                 # (StopAt protocol: never ends the forward)
        self.name = name
        self.leaf: Optional[torch.Tensor] = None
        self.entries: List = []
        self.live = set()

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        if k == self.name and self.leaf is None:
            self.leaf = v[1]
            self.live.add(_key(self.leaf))

    @property
    def recording(self) -> bool:
        return self.leaf is not None

    def depends(self, *ts) -> bool:
        return self.leaf is not None and any(t is not None and _key(_root(t)[0]) in self.live for t in ts)

    def rec(self, inputs: Sequence[Optional[torch.Tensor]], outputs: Sequence[torch.Tensor], bwd: Callable):
        """bwd(list of dL/d output | None) -> list of dL/d input | None, one per input (same shapes as the inputs)."""
        if not self.depends(*inputs):
            return
        for o in outputs:
            self.live.add(_key(o))
        self.entries.append((list(inputs), list(outputs), bwd))

    def backward(self, out: torch.Tensor, g_out: torch.Tensor) -> torch.Tensor:
        if self.leaf is None:
            raise TfmqError(f"GradTape: the forward never reached unit '{self.name}'")
        grads = {_key(out): g_out}
        for inputs, outputs, bwd in reversed(self.entries):
            gouts = [grads.pop(_key(o), None) for o in outputs]
            if all(g is None for g in gouts):
                continue
            gins = bwd(gouts)
            for t, g in zip(inputs, gins):
                if t is None or g is None:
                    continue
                root, off, rowlen = _root(t)
                k = _key(root)
                if k not in self.live:
                    continue
                if off != 0 or rowlen != t.shape[-1]:        # gradient of a column slice: zero elsewhere in the root's rows
                    full = torch.zeros(root.shape, dtype=torch.float32, device=root.device)
                    full.reshape(-1, rowlen)[:, off:off + t.shape[-1]].copy_(g.reshape(-1, t.shape[-1]))
                    g = full
                if k in grads:
                    ops.axpy(grads[k], g.contiguous(), 1.0)
                else:
                    # own the buffer: a pass-through gradient (residual add) is shared with another consumer
                    grads[k] = g.contiguous().clone() if any(g is go for go in gouts if go is not None) else g.contiguous()
        g = grads.get(_key(self.leaf))
        if g is None:
            raise TfmqError(f"GradTape: no gradient reached unit '{self.name}' (is it upstream of the model output?)")
        return g.reshape(self.leaf.shape)


class _Heads:
    use_flash = True

    def __init__(self, heads):
        self.heads = heads


def attention_taped(tape: GradTape, q, k, v, heads: int, scale: float) -> torch.Tensor:
    """softmax(q k^T scale) v on packed heads with its backward on the tape (the reconstruction units' attention: exact fp32,
    fused where its shape rules hold, otherwise strided GEMMs + row softmax)."""
    from .recon import TransformerUnit
    Cc = q.shape[-1]
    d = Cc // heads
    if abs(scale - float(d ** -0.5)) > 1e-12 * abs(scale):
        raise TfmqError("attention_taped: scale must be head_dim ** -0.5")
    H = _Heads(heads)
    qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
    o, saved = TransformerUnit._attn_fwd(H, qc, kc, vc)

    def bwd(gouts):
        dq, dk, dv = TransformerUnit._attn_bwd(H, gouts[0].reshape(qc.shape).contiguous(), qc, kc, vc, saved)
        return [dq, dk, dv]
    tape.rec([q, k, v], [o], bwd)
    return o
