"""Reconstruction units: hand-written forward / backward of one quantised unit w.r.t. the AdaRound
`alpha` of its layers (K12-K15), replacing autograd through `block(*cur_inputs)` + torch.optim.Adam
(reference quant/reconstruction.py:63-78,182-198,290-303).

Every unit exposes
    iterate(idx, count) -> (total_loss, rec_loss, round_loss)
running one Adam iteration on the mini-batch `idx` of its cached inputs/targets:
    soft weights (K12) -> unit forward (exact fp32 GEMMs over im2col) -> lp_loss (K13) -> unit
    backward to dL/dW_hat -> [all-reduce SUM across ranks, K16] -> alpha gradient + rounding
    regulariser + Adam (K12/K14 fused).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch

from .. import ops
from .._lib import TfmqError


def temp_decay(count: int, t_max: int, warmup: float, start_b: float = 20.0, end_b: float = 2.0) -> float:
    """LinearTempDecay (reference quant/reconstruction_util.py:176-198; linear, not cosine)."""
    start = warmup * t_max
    if count < start:
        return start_b
    rel = (count - start) / (t_max - start)
    return end_b + (start_b - end_b) * max(0.0, 1.0 - rel)


class AdaLayer:
    """One layer whose rounding is being learned: w (OIHW / [out,in]), frozen delta/zp, alpha + Adam state."""

    def __init__(self, w: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor, bias: Optional[torch.Tensor], level: int = 16,
                 alpha: Optional[torch.Tensor] = None):
        self.w = w.detach().float().contiguous()
        self.cout = w.shape[0]
        self.cin = w.shape[1]
        self.kh, self.kw = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
        self.delta = delta.detach().reshape(-1).float().contiguous()
        self.zp = zp.detach().reshape(-1).float().contiguous()
        self.bias = None if bias is None else bias.detach().float().contiguous()
        self.level = level
        self.alpha = ops.adaround_init(self.w, self.delta) if alpha is None else alpha.detach().float().contiguous().clone()
        self.m = torch.zeros_like(self.alpha)
        self.v = torch.zeros_like(self.alpha)

    def soft_weight_gemm(self) -> torch.Tensor:
        """w_hat in the GEMM layout [cout, kh*kw*cin] (adaptive_rounding.py soft branch)."""
        w_hat = ops.adaround_soft_fwd(self.w, self.alpha, self.delta, self.zp, self.level)
        if self.kh * self.kw == 1:
            return w_hat.reshape(self.cout, self.cin)
        return ops.w_relayout(w_hat, self.cout, self.cin, self.kh, self.kw, to_gemm=True)

    def grad_to_oihw(self, g_gemm: torch.Tensor) -> torch.Tensor:
        if self.kh * self.kw == 1:
            return g_gemm.reshape(self.w.shape)
        return ops.w_relayout(g_gemm, self.cout, self.cin, self.kh, self.kw, to_gemm=False)

    def step(self, g_what: torch.Tensor, w_reg: float, b_temp: float, lr: float, t: int, round_loss: torch.Tensor):
        ops.adaround_bwd_adam(self.w, self.alpha, self.delta, self.zp, g_what.contiguous(), self.m, self.v, self.level,
                              w_reg, b_temp, lr, t, round_loss)

    def step_dyn(self, g_what: torch.Tensor, scalars: torch.Tensor, round_loss: torch.Tensor):
        """The same step with (w_reg, b_temp, lr / bc1, sqrt bc2) read from the device tensor `scalars` (a captured iteration)."""
        ops.adaround_bwd_adam_dyn(self.w, self.alpha, self.delta, self.zp, g_what.contiguous(), self.m, self.v, self.level, scalars, round_loss)


def _chunk_cuts(sizes: Sequence[int], n_chunks: int):
    """Cut a list of layer gradient sizes into <= n_chunks contiguous pieces of roughly equal bytes, at layer boundaries.
    -> [(first layer, end layer, first element, end element)]"""
    total, n = sum(sizes), len(sizes)
    n_chunks = max(1, min(n_chunks, n))
    cuts, l0, e0, acc, k = [], 0, 0, 0, 1
    for i, sz in enumerate(sizes):
        acc += sz
        last = i == n - 1
        if last or (k < n_chunks and acc >= total * k / n_chunks and n - 1 - i >= n_chunks - k):
            cuts.append((l0, i + 1, e0, acc))
            l0, e0, k = i + 1, acc, k + 1
    return cuts



def _rec_loss(unit, out, y, denom, idx):
    """The unit's reconstruction term and its gradient: lp_loss (p = 2), or -- `unit.fisher = (mode, cached |dL/d out| + 1 of the whole
    calibration set)` -- LossFunc's FISHER_DIAG / FISHER_FULL (reference quant/reconstruction_util.py:49-59)."""
    fisher = getattr(unit, "fisher", None)
    if fisher is None:
        return ops.recon_loss(out, y, denom=denom)
    mode, fg = fisher
    if mode == ops.FISHER_FULL and out.dim() != 4:
        raise TfmqError("RLOSS.FISHER_FULL sums over dims (1, 2, 3): 4-D unit outputs only (the reference raises on token tensors too)")
    return ops.fisher_loss(out.contiguous(), y.contiguous(), fg.index_select(0, idx).contiguous(), mode, denom)


class _Unit:
    """Shared iteration driver.  Sub-classes implement _forward_backward(idx) -> (rec_loss_tensor,
    [g_what per AdaLayer, OIHW])."""

    trace = None      # tests: a list -> the first iteration of every unit appends (local flat gradient, reduced flat gradient)

    def __init__(self, layers: Sequence[AdaLayer], iters: int, w: float = 0.01, warmup: float = 0.2, lr: float = 1e-3,
                 world_size: int = 1, allreduce=None, b_range=(20, 2)):
        self.layers = list(layers)
        self.iters, self.w_reg, self.warmup, self.lr = iters, w, warmup, lr
        self.b_range = (float(b_range[0]), float(b_range[1]))      # LossFunc's temperature range (reconstruction.py b_range)
        self.world_size, self.allreduce = world_size, allreduce
        self.count = 0
        dev = self.layers[0].w.device
        self._rl = torch.zeros(1, dtype=torch.float32, device=dev)
        self._flat = None
        self._comm_stream = None
        self.gemm_mode = os.environ.get("TFMQ_RECON_GEMM", "bf16x3")
        self.exchange_chunks = max(1, int(os.environ.get("TFMQ_EXCHANGE_CHUNKS", "2")))
        # Round 5 (measured, OFF by default): TFMQ_RECON_GRAPH=1 captures the iteration (forward, loss, backward, AdaRound backward + Adam of
        # every layer: 60-130 launches) ONCE as a hipGraph through torch.cuda.graph after two eager iterations and replays it; what changes
        # per iteration travels in device memory -- the mini-batch indices (a static index tensor) and the four scalars of the optimizer
        # kernel (tfmq_adaround_bwd_adam_dyn).  Same kernels in the same order: alphas and Adam moments bit-identical
        # (tests/test_recon_graph_gpu.py).  Same-box A/B on the whole SD job (74 units x 1000 iterations): 149.3 s replayed against 147.2 s
        # eager (profiles/r05_ab_recon_graph.txt) -- the eager host already runs ahead of the GPU at every level; the job is bound by its
        # kernels, not by launch rate.  Single-GPU units only (the multi-GPU exchange runs on a side stream between the kernels).
        self.graph_on = os.environ.get("TFMQ_RECON_GRAPH", "0") == "1" and dev.type == "cuda" and not (world_size > 1 and allreduce is not None)
        self._graph = None

    GRAPH_AFTER = 2      # eager iterations in front of the capture (workspaces of the GEMM's split-K path are allocated by then)

    SC_CHUNK = 256       # iterations whose optimizer scalars are computed on the host and uploaded together

    def _graph_iterate(self, idx):
        if self._graph is None:
            self._idx_static = idx.clone()
            self._sc_dev = torch.zeros(4, dtype=torch.float32, device=idx.device)
            self._sc_chunk, self._sc_base = None, 0
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if self.gemm_mode != "f32":
                    with ops.gemm_precision(self.gemm_mode, self._rl.device.index):
                        rec, grads = self._forward_backward(self._idx_static)
                else:
                    rec, grads = self._forward_backward(self._idx_static)
                self._rl.zero_()
                for layer, gg in zip(self.layers, grads):
                    layer.step_dyn(gg, self._sc_dev, self._rl)
            self._graph, self._rec_static = g, rec
        if self._sc_chunk is None or self.count - self._sc_base >= self._sc_chunk.shape[0]:
            # {w_reg, b (0 during the warm-up), lr / (1 - 0.9^t), sqrt(1 - 0.999^t)} of the next SC_CHUNK iterations, computed by the library's
            # own host function (the values tfmq_adaround_bwd_adam would compute for each t); one upload, then device-to-device rows
            rows = []
            # (a unit iterated past `iters` keeps getting rows, as the eager path accepts any count: ADVICE r5)
            for c in range(self.count, max(self.count + 1, min(self.count + self.SC_CHUNK, self.iters + 1))):
                b = temp_decay(c, self.iters, self.warmup, self.b_range[0], self.b_range[1])
                rows.append(ops.adaround_scalars(self.w_reg, b if c >= self.iters * self.warmup else 0.0, self.lr, c))
            self._sc_chunk, self._sc_base = torch.tensor(rows, dtype=torch.float32).to(idx.device), self.count
        self._idx_static.copy_(idx)
        self._sc_dev.copy_(self._sc_chunk[self.count - self._sc_base])
        self._graph.replay()
        return self._rec_static, self._rl

    def iterate(self, idx: torch.Tensor):
        self.count += 1
        if self.graph_on and self.count > self.GRAPH_AFTER and self.fisher is None and _Unit.trace is None:
            return self._graph_iterate(idx)
        # TFMQ_RECON_GEMM = bf16x3 (default) | f32 | f16: operand precision of the unit's forward / backward GEMMs on the matrix cores
        # (ops.gemm_precision; csrc/gemm_f32_mfma.hip).  bf16x3 = each fp32 operand split hi + lo in bf16, three MFMAs per product,
        # fp32 accumulation: 2^-16 relative error per product (measured 4.5e-6 max-normalised on SD shapes against 5e-7..1.6e-6 of the
        # fp32 MFMA's own summation order) at 1.6x the fp32 MFMA's rate; the reference's 400-iteration loss curve and final masks (F8b)
        # are reproduced as closely as with exact fp32 products (tests/test_recon_precision_gpu.py).  f32 restores exact products.  The
        # soft-rounded weights, the loss, the AdaRound / Adam kernels and the fused attention stay fp32 in every mode.
        if self.gemm_mode != "f32":
            with ops.gemm_precision(self.gemm_mode, self._rl.device.index):
                rec, grads = self._forward_backward(idx)
        else:
            rec, grads = self._forward_backward(idx)
        b = temp_decay(self.count, self.iters, self.warmup, self.b_range[0], self.b_range[1])
        reg_on = self.count >= self.iters * self.warmup
        self._rl.zero_()
        dist_on = self.world_size > 1 and self.allreduce is not None
        w_eff = self.w_reg * (self.world_size if dist_on else 1)
        if not dist_on:
            for layer, g in zip(self.layers, grads):
                layer.step(g, w_eff, b if reg_on else 0.0, self.lr, self.count, self._rl)
            return rec, self._rl
        # One flattened fp32 buffer per iteration instead of one all-reduce per tensor (reference reconstruction.py:193-195); the
        # rounding regulariser is computed identically on every rank and is therefore summed world_size times by the reference's
        # grad all-reduce (w_eff).  The buffer is exchanged in `exchange_chunks` contiguous pieces cut at layer boundaries, on a SIDE
        # stream: while piece k travels over xGMI, the fused AdaRound-backward + Adam kernels of piece k-1 run on the unit's stream
        # (events both ways, no host synchronisation).  One piece = the plain in-stream exchange.
        if self._flat is None:
            self._flat = torch.empty(sum(g.numel() for g in grads), dtype=torch.float32, device=grads[0].device)
            self._cuts = _chunk_cuts([g.numel() for g in grads], self.exchange_chunks)
        off, views = 0, []
        for g in grads:
            v = self._flat[off:off + g.numel()]
            v.copy_(g.reshape(-1))
            views.append(v.view_as(g))
            off += g.numel()
        local = self._flat.clone() if (_Unit.trace is not None and self.count == 1) else None
        side = self._flat.is_cuda and len(self._cuts) > 1
        if not side:
            self.allreduce(self._flat)
            if local is not None:
                _Unit.trace.append((type(self).__name__, local.cpu(), self._flat.cpu().clone()))
            for layer, g in zip(self.layers, views):
                layer.step(g, w_eff, b if reg_on else 0.0, self.lr, self.count, self._rl)
            return rec, self._rl
        main = torch.cuda.current_stream(self._flat.device)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(self._flat.device)
            self._ev_ready = torch.cuda.Event()
            self._ev_done = [torch.cuda.Event() for _ in self._cuts]
            self._ev_adam = torch.cuda.Event()
        self._ev_ready.record(main)                         # gradients are in the flat buffer
        self._comm_stream.wait_event(self._ev_ready)
        with torch.cuda.stream(self._comm_stream):
            for k, (l0, l1, e0, e1) in enumerate(self._cuts):
                self.allreduce(self._flat[e0:e1])
                self._ev_done[k].record(self._comm_stream)
        for k, (l0, l1, e0, e1) in enumerate(self._cuts):
            main.wait_event(self._ev_done[k])
            for layer, g in zip(self.layers[l0:l1], views[l0:l1]):
                layer.step(g, w_eff, b if reg_on else 0.0, self.lr, self.count, self._rl)
        self._ev_adam.record(main)                          # the next iteration's copies into the flat buffer are stream-ordered behind this
        self._comm_stream.wait_event(self._ev_adam)
        if local is not None:
            _Unit.trace.append((type(self).__name__, local.cpu(), self._flat.cpu().clone()))
        return rec, self._rl

    def losses(self, rec, rl):
        r, q = float(rec), float(rl) / (self.world_size if (self.world_size > 1 and self.allreduce is not None) else 1)
        return r + q, r, q

    fisher = None       # (ops.FISHER_DIAG | FISHER_FULL, cached Fisher weights [N, ...]) or None = MSE
    _loss = _rec_loss


def _conv_fwd(x, layer: AdaLayer, w_gemm, pad, rowadd=None, residual=None):
    """x NHWC -> (y NHWC, col) via im2col + exact fp32 GEMM."""
    B, H, W, _ = x.shape
    if layer.kh * layer.kw == 1:
        col = x.reshape(B * H * W, layer.cin)
    else:
        col = ops.im2col(x, layer.kh, layer.kw, 1, pad)
    y = ops.gemm(col, w_gemm, trans_b=True, bias=layer.bias, rowadd=rowadd, rows_per_img=H * W,
                 residual=None if residual is None else residual.reshape(B * H * W, -1))
    return y.reshape(B, H, W, layer.cout), col


class LayerUnit(_Unit):
    """Single QuantLayer (layer_reconstruction, reference :13-82); x: the layer's (already up-sampled) input."""

    def __init__(self, layer: AdaLayer, x: torch.Tensor, y: torch.Tensor, pad=(1, 1, 1, 1), **kw):
        super().__init__([layer], **kw)
        self.x, self.y, self.pad = x, y, pad

    def _forward_backward(self, idx):
        L = self.layers[0]
        x, y = self.x.index_select(0, idx), self.y.index_select(0, idx)
        wg = L.soft_weight_gemm()
        out, col = _conv_fwd(x, L, wg, self.pad)
        loss, g = self._loss(out, y, out.numel() // out.shape[-1], idx)
        gw = ops.gemm(g.reshape(-1, L.cout), col, trans_a=True)
        return loss, [L.grad_to_oihw(gw)]


class ResnetUnit(_Unit):
    """QuantResnetBlock (reference quant/quant_block.py:415-444): AdaRound on conv1, conv2; temb_proj is
    `quant_emb` (owned by the TIB unit) and nin_shortcut is FP, both constant here."""

    def __init__(self, conv1: AdaLayer, conv2: AdaLayer, gn1, gn2, shortcut, x, proj, y, eps: float = 1e-6, **kw):
        """eps: 1e-6 for the DDPM UNet's Normalize, 1e-5 for the LDM ResBlock's GroupNorm32 (QuantResBlock, reference
        quant/quant_block.py:131-206: same dataflow with emb_layers in place of temb_proj, skip_connection in place of
        nin_shortcut)."""
        super().__init__([conv1, conv2], **kw)
        self.eps = eps
        self.gn1, self.gn2 = gn1, gn2      # (gamma, beta)
        self.shortcut = shortcut           # None or (w [cout, cin] fp32, bias)
        self.x, self.proj, self.y = x, proj, y

    def _forward_backward(self, idx):
        c1l, c2l = self.layers
        x, proj, y = self.x.index_select(0, idx), self.proj.index_select(0, idx), self.y.index_select(0, idx)
        B, H, W, cin = x.shape
        w1, w2 = c1l.soft_weight_gemm(), c2l.soft_weight_gemm()
        _, a1, _ = ops.groupnorm(x, self.gn1[0], self.gn1[1], self.eps, True, want_f32=True)
        c1, col1 = _conv_fwd(a1, c1l, w1, (1, 1, 1, 1), rowadd=proj)
        _, a2, _ = ops.groupnorm(c1, self.gn2[0], self.gn2[1], self.eps, True, want_f32=True)
        if self.shortcut is not None:
            sc = ops.gemm(x.reshape(B * H * W, cin), self.shortcut[0], trans_b=True, bias=self.shortcut[1]).reshape(B, H, W, -1)
        else:
            sc = x
        out, col2 = _conv_fwd(a2, c2l, w2, (1, 1, 1, 1), residual=sc)
        loss, g = self._loss(out, y, B * H * W, idx)
        g2 = g.reshape(B * H * W, c2l.cout)
        gw2 = ops.gemm(g2, col2, trans_a=True)
        dcol2 = ops.gemm(g2, w2)
        g_a2 = ops.col2im(dcol2, (B, H, W, c2l.cin), 3, 3, 1, (1, 1, 1, 1))
        g_c1 = ops.groupnorm_bwd(c1, g_a2, self.gn2[0], self.gn2[1], self.eps, True)
        gw1 = ops.gemm(g_c1.reshape(B * H * W, c1l.cout), col1, trans_a=True)
        return loss, [c1l.grad_to_oihw(gw1), c2l.grad_to_oihw(gw2)]


class AttnUnit(_Unit):
    """QuantAttnBlock (reference quant/quant_block.py:474-505), single head, un-quantised matmuls."""

    def __init__(self, q: AdaLayer, k: AdaLayer, v: AdaLayer, po: AdaLayer, gn, x, y, **kw):
        super().__init__([q, k, v, po], **kw)
        self.gn, self.x, self.y = gn, x, y

    def _forward_backward(self, idx):
        ql, kl, vl, pl = self.layers
        x, y = self.x.index_select(0, idx), self.y.index_select(0, idx)
        B, H, W, Cc = x.shape
        T = H * W
        scale = float(int(Cc) ** (-0.5))
        wq, wk, wv, wp = (l.soft_weight_gemm() for l in self.layers)
        _, hn, _ = ops.groupnorm(x, self.gn[0], self.gn[1], 1e-6, False, want_f32=True)
        hf = hn.reshape(B * T, Cc)
        q = ops.gemm(hf, wq, trans_b=True, bias=ql.bias).reshape(B, T, Cc)
        k = ops.gemm(hf, wk, trans_b=True, bias=kl.bias).reshape(B, T, Cc)
        v = ops.gemm(hf, wv, trans_b=True, bias=vl.bias).reshape(B, T, Cc)
        S = ops.gemm(q, k, trans_b=True)                        # [B,T,T]
        P = ops.softmax_rows(S, scale)
        o = ops.gemm(P, v)                                      # [B,T,C]
        out = ops.gemm(o.reshape(B * T, Cc), wp, trans_b=True, bias=pl.bias, residual=x.reshape(B * T, Cc)).reshape(B, H, W, Cc)
        loss, g = self._loss(out, y, B * T, idx)
        g2 = g.reshape(B * T, Cc)
        gwp = ops.gemm(g2, o.reshape(B * T, Cc), trans_a=True)
        g_o = ops.gemm(g2, wp).reshape(B, T, Cc)
        dV = ops.gemm(P, g_o, trans_a=True)
        dP = ops.gemm(g_o, v, trans_b=True)
        dS = ops.softmax_bwd_rows(P, dP, scale)
        dQ = ops.gemm(dS, k)
        dK = ops.gemm(dS, q, trans_a=True)
        gwq = ops.gemm(dQ.reshape(B * T, Cc), hf, trans_a=True)
        gwk = ops.gemm(dK.reshape(B * T, Cc), hf, trans_a=True)
        gwv = ops.gemm(dV.reshape(B * T, Cc), hf, trans_a=True)
        return loss, [g.reshape(l.w.shape) for g, l in zip((gwq, gwk, gwv, gwp), self.layers)]


class TransformerUnit(_Unit):
    """QuantBasicTransformerBlock (reference quant/quant_block.py:254-299 with cross_attn_forward :208-251): self-attention,
    cross-attention on `context`, GEGLU feed-forward, pre-LayerNorm residuals; multi-head, un-quantised matmuls.
    Layers in module order: attn1.{to_q,to_k,to_v,to_out.0}, ff.net.0.proj, ff.net.2, attn2.{to_q,to_k,to_v,to_out.0}.
    The heads stay packed in the channel dimension ([B,T,heads*d]); per-head GEMMs address them by stride."""

    def __init__(self, layers: Sequence[AdaLayer], norms, heads: int, x: torch.Tensor, ctx: torch.Tensor, y: torch.Tensor, **kw):
        super().__init__(layers, **kw)
        assert len(layers) == 10
        self.norms, self.heads = norms, heads      # norms: 3 x (gamma, beta)
        self.x, self.ctx, self.y = x, ctx, y

    # ---- multi-head attention on packed heads
    use_flash = True      # exact-fp32 fused attention where its shape rules hold (tests switch it off to compare)

    def _attn_fwd(self, q, k, v):
        """q [B,T,C], k/v [B,L,C] -> (o [B,T,C], saved): saved = P [B,heads,T,L] on the GEMM path, or the row
        log-sum-exp and o for the fused kernels (nothing of size T x L is materialised there)."""
        B, T, Cc = q.shape
        L, H = k.shape[1], self.heads
        d = Cc // H
        if self.use_flash and ops.attention_f32_ok(T, L, d):
            o, lse = ops.attention_f32_fwd(q.contiguous(), k.contiguous(), v.contiguous(), H, float(d ** -0.5))
            return o, ("flash", lse, o)
        S = torch.empty(B, H, T, L, dtype=torch.float32, device=q.device)
        # all heads of a product in ONE launch (two-level batch: item (b, h) at b * batch stride + h * head stride)
        hk = dict(heads=H)
        ops.gemm_strided(q, 0, Cc, 1, T * Cc, k, 0, 1, Cc, L * Cc, S, 0, L, H * T * L, T, L, d, B, hsa=d, hsb=d, hsc=T * L, **hk)
        P = ops.softmax_rows(S, float(d ** -0.5))
        o = torch.empty(B, T, Cc, dtype=torch.float32, device=q.device)
        ops.gemm_strided(P, 0, L, 1, H * T * L, v, 0, Cc, 1, L * Cc, o, 0, Cc, T * Cc, T, d, L, B, hsa=T * L, hsb=d, hsc=d, **hk)
        return o, P

    def _attn_bwd(self, g_o, q, k, v, P):
        """-> dQ [B,T,C], dK [B,L,C], dV [B,L,C]"""
        B, T, Cc = q.shape
        L, H = k.shape[1], self.heads
        d = Cc // H
        if isinstance(P, tuple):
            _, lse, o = P
            return ops.attention_f32_bwd(q.contiguous(), k.contiguous(), v.contiguous(), o, lse, g_o.contiguous(), H, float(d ** -0.5))
        dV, dK, dQ = torch.empty_like(v), torch.empty_like(k), torch.empty_like(q)
        dP = torch.empty_like(P)
        hk = dict(heads=H)
        # dV_h = P_h^T g_o_h ; dP_h = g_o_h V_h^T
        ops.gemm_strided(P, 0, 1, L, H * T * L, g_o, 0, Cc, 1, T * Cc, dV, 0, Cc, L * Cc, L, d, T, B, hsa=T * L, hsb=d, hsc=d, **hk)
        ops.gemm_strided(g_o, 0, Cc, 1, T * Cc, v, 0, 1, Cc, L * Cc, dP, 0, L, H * T * L, T, L, d, B, hsa=d, hsb=d, hsc=T * L, **hk)
        dS = ops.softmax_bwd_rows(P, dP, float(d ** -0.5))
        # dQ_h = dS_h K_h ; dK_h = dS_h^T Q_h
        ops.gemm_strided(dS, 0, L, 1, H * T * L, k, 0, Cc, 1, L * Cc, dQ, 0, Cc, T * Cc, T, d, L, B, hsa=T * L, hsb=d, hsc=d, **hk)
        ops.gemm_strided(dS, 0, 1, L, H * T * L, q, 0, Cc, 1, T * Cc, dK, 0, Cc, L * Cc, L, d, T, B, hsa=T * L, hsb=d, hsc=d, **hk)
        return dQ, dK, dV

    def _forward_backward(self, idx):
        (q1l, k1l, v1l, o1l, f0l, f2l, q2l, k2l, v2l, o2l) = self.layers
        x, ctx, y = self.x.index_select(0, idx), self.ctx.index_select(0, idx), self.y.index_select(0, idx)
        B, T, Cc = x.shape
        L, Dc = ctx.shape[1], ctx.shape[2]
        W = [l.soft_weight_gemm() for l in self.layers]
        (wq1, wk1, wv1, wo1, wf0, wf2, wq2, wk2, wv2, wo2) = W
        (g1, b1), (g2, b2), (g3, b3) = self.norms
        x2d = x.reshape(B * T, Cc)
        c2d = ctx.reshape(B * L, Dc)
        # ---- forward
        n1 = ops.layernorm(x, g1, b1, 1e-5, None)[1].reshape(B * T, Cc)
        q1 = ops.gemm(n1, wq1, trans_b=True, bias=q1l.bias).reshape(B, T, Cc)
        k1 = ops.gemm(n1, wk1, trans_b=True, bias=k1l.bias).reshape(B, T, Cc)
        v1 = ops.gemm(n1, wv1, trans_b=True, bias=v1l.bias).reshape(B, T, Cc)
        o1, P1 = self._attn_fwd(q1, k1, v1)
        x1 = ops.gemm(o1.reshape(B * T, Cc), wo1, trans_b=True, bias=o1l.bias, residual=x2d)          # [B*T, C]
        n2 = ops.layernorm(x1.reshape(B, T, Cc), g2, b2, 1e-5, None)[1].reshape(B * T, Cc)
        q2 = ops.gemm(n2, wq2, trans_b=True, bias=q2l.bias).reshape(B, T, Cc)
        k2 = ops.gemm(c2d, wk2, trans_b=True, bias=k2l.bias).reshape(B, L, Cc)
        v2 = ops.gemm(c2d, wv2, trans_b=True, bias=v2l.bias).reshape(B, L, Cc)
        o2, P2 = self._attn_fwd(q2, k2, v2)
        x2 = ops.gemm(o2.reshape(B * T, Cc), wo2, trans_b=True, bias=o2l.bias, residual=x1)
        n3 = ops.layernorm(x2.reshape(B, T, Cc), g3, b3, 1e-5, None)[1].reshape(B * T, Cc)
        hcat = ops.gemm(n3, wf0, trans_b=True, bias=f0l.bias)                                        # [B*T, 2I]
        gg = ops.geglu(hcat, None)[1]                                                                  # [B*T, I]
        out = ops.gemm(gg, wf2, trans_b=True, bias=f2l.bias, residual=x2)
        # lp_loss on a [B, T, C] tensor sums over dim 1 = the tokens and averages over B * C (quant_layer.py:152-153)
        loss, g = self._loss(out.reshape(B, T, Cc), y, B * Cc, idx)
        # ---- backward (weight gradients only; d/dx of the block input is not needed)
        g_out = g.reshape(B * T, Cc)
        gwf2 = ops.gemm(g_out, gg, trans_a=True)
        d_gg = ops.gemm(g_out, wf2)
        d_hcat = ops.geglu_bwd(hcat, d_gg)
        gwf0 = ops.gemm(d_hcat, n3, trans_a=True)
        d_n3 = ops.gemm(d_hcat, wf0)
        d_x2 = ops.layernorm_bwd(x2.reshape(B, T, Cc), d_n3.reshape(B, T, Cc), g3, 1e-5).reshape(B * T, Cc)
        ops.axpy(d_x2, g_out, 1.0)                                   # residual path of the feed-forward
        # cross attention
        gwo2 = ops.gemm(d_x2, o2.reshape(B * T, Cc), trans_a=True)
        g_o2 = ops.gemm(d_x2, wo2).reshape(B, T, Cc)
        dQ2, dK2, dV2 = self._attn_bwd(g_o2, q2, k2, v2, P2)
        gwq2 = ops.gemm(dQ2.reshape(B * T, Cc), n2, trans_a=True)
        gwk2 = ops.gemm(dK2.reshape(B * L, Cc), c2d, trans_a=True)
        gwv2 = ops.gemm(dV2.reshape(B * L, Cc), c2d, trans_a=True)
        d_n2 = ops.gemm(dQ2.reshape(B * T, Cc), wq2)
        d_x1 = ops.layernorm_bwd(x1.reshape(B, T, Cc), d_n2.reshape(B, T, Cc), g2, 1e-5).reshape(B * T, Cc)
        ops.axpy(d_x1, d_x2, 1.0)                                    # residual path of the cross attention
        # self attention
        gwo1 = ops.gemm(d_x1, o1.reshape(B * T, Cc), trans_a=True)
        g_o1 = ops.gemm(d_x1, wo1).reshape(B, T, Cc)
        dQ1, dK1, dV1 = self._attn_bwd(g_o1, q1, k1, v1, P1)
        gwq1 = ops.gemm(dQ1.reshape(B * T, Cc), n1, trans_a=True)
        gwk1 = ops.gemm(dK1.reshape(B * T, Cc), n1, trans_a=True)
        gwv1 = ops.gemm(dV1.reshape(B * T, Cc), n1, trans_a=True)
        grads = (gwq1, gwk1, gwv1, gwo1, gwf0, gwf2, gwq2, gwk2, gwv2, gwo2)
        return loss, [gw.reshape(l.w.shape) for gw, l in zip(grads, self.layers)]


class TibUnit(_Unit):
    """Temporal-information block (TIAR, reference quant/reconstruction.py:212-318 with
    LossFuncTimeEmbedding :94-173): dense.1 and every temb projection learn their rounding against
    the FP projections; dense.0 is FP (ignore_recon).  s0 = silu(dense0(emb(t))) is constant."""

    def __init__(self, dense1: AdaLayer, projs: Sequence[AdaLayer], s0: torch.Tensor, targets: Sequence[torch.Tensor], **kw):
        super().__init__([dense1] + list(projs), **kw)
        self.s0, self.targets = s0, list(targets)

    def _forward_backward(self, idx):
        d1, projs = self.layers[0], self.layers[1:]
        s0 = self.s0.index_select(0, idx)
        m = s0.shape[0]
        w1 = d1.soft_weight_gemm()
        temb = ops.gemm(s0, w1, trans_b=True, bias=d1.bias)
        s1 = ops.silu(temb)
        total = torch.zeros(1, dtype=torch.float32, device=s0.device)
        g_s1 = None
        grads = [None]
        for pl, tgt in zip(projs, self.targets):
            wp = pl.soft_weight_gemm()
            out = ops.gemm(s1, wp, trans_b=True, bias=pl.bias)
            loss, g = ops.recon_loss(out, tgt.index_select(0, idx), denom=m)
            ops.axpy(total, loss, 1.0)
            grads.append(ops.gemm(g, s1, trans_a=True).reshape(pl.w.shape))
            g_s1 = ops.gemm(g, wp, out=g_s1, accumulate=g_s1 is not None)
        g_temb = ops.silu_bwd(temb, g_s1)
        grads[0] = ops.gemm(g_temb, s0, trans_a=True).reshape(d1.w.shape)
        return total, grads


# ============================================================================================ delta learning (use_aq = True)
class FixedLayer:
    """A layer of a delta-learning unit: fixed (hard-rounded / fake-quantised) weights in the GEMM layout, bias, shape, and the index
    of its activation quantizer in the unit's delta vector (None: the layer's input is not quantised -- `disable_aq`)."""

    def __init__(self, w_hat: torch.Tensor, bias: Optional[torch.Tensor], qi: Optional[int]):
        w_hat = w_hat.detach().float().contiguous()
        self.cout, self.cin = w_hat.shape[0], w_hat.shape[1]
        self.kh, self.kw = (w_hat.shape[2], w_hat.shape[3]) if w_hat.dim() == 4 else (1, 1)
        self.wg = w_hat.reshape(self.cout, self.cin) if self.kh * self.kw == 1 else ops.w_relayout(w_hat, self.cout, self.cin, self.kh, self.kw, to_gemm=True)
        self.bias = None if bias is None else bias.detach().float().contiguous()
        self.qi = qi


class _DeltaUnit:
    """Reconstruction with `use_aq=True` (reference quant/reconstruction.py:36-48,135-166): the unit's weights are FIXED, the trainable
    parameters are the scalar deltas of its activation quantizers; the quantizer is differentiated through the straight-through
    round (quant_layer.py:152-160,211-227: d round(u)/du = 1, clamp passes gradients inside [0, L-1]); Adam(lr) with
    CosineAnnealingLR(T_max=iters, eta_min=0); the loss is the reconstruction term alone (round_loss NONE).  Multi-GPU: the
    reference SUMs `param.grad` over the ranks (not an average), so does this.

    Sub-classes implement _forward_backward(idx) -> (rec loss tensor, [dL/ddelta_i (fp32 [1]) or None per quantizer])."""

    def __init__(self, deltas: Sequence[torch.Tensor], zps: Sequence[torch.Tensor], levels: Sequence[int], iters: int, lr: float = 4e-5,
                 world_size: int = 1, allreduce=None):
        dev = deltas[0].device
        self.delta = torch.stack([d.detach().reshape(()).float() for d in deltas]).to(dev).contiguous()      # [n], trained in place
        self.zp = torch.stack([torch.as_tensor(z).detach().reshape(()).float().to(dev) for z in zps]).contiguous()
        self.levels = [int(l) for l in levels]
        self.iters, self.lr0 = int(iters), float(lr)
        self.world_size, self.allreduce = world_size, allreduce
        self.m, self.v = torch.zeros_like(self.delta), torch.zeros_like(self.delta)
        self.count = 0
        self._zero = torch.zeros(1, dtype=torch.float32, device=dev)
        self.gemm_mode = os.environ.get("TFMQ_RECON_GEMM", "bf16x3")

    # quantizer i on a tensor (forward) / its backward
    def q(self, i: int, x: torch.Tensor) -> torch.Tensor:
        return ops.fake_quant(x.contiguous(), self.delta[i:i + 1], self.zp[i:i + 1], self.levels[i])

    def q_bwd(self, i: int, x: torch.Tensor, g: torch.Tensor, want_gx: bool = True):
        return ops.fake_quant_bwd(x.contiguous(), g.contiguous(), self.delta[i:i + 1], self.zp[i:i + 1], self.levels[i], want_gx)

    # ---- attention with LIVE matmul quantizers (aqtizer_q / _k / _v / _w of QuantAttnBlock, cross_attn_forward,
    # reference quant/quant_block.py:226-243,483-500; trained through the `A` lists of reconstruction.py:145-163).  aq = indices of
    # (q, k, v, w) in the unit's delta vector (w None: a 16-bit softmax quantizer is left out, as the reference does).  Functional path:
    # fake-quantised operands, per-head strided fp32 GEMMs, row softmax, straight-through backward for every quantizer.
    def _attn_fwd_quant(self, q, k, v, aq, heads: int):
        B, T, Cc = q.shape
        L, H = k.shape[1], heads
        d = Cc // H
        qh, kh, vh = self.q(aq[0], q).reshape(B, T, Cc), self.q(aq[1], k).reshape(B, L, Cc), self.q(aq[2], v).reshape(B, L, Cc)
        S = torch.empty(B, H, T, L, dtype=torch.float32, device=q.device)
        ops.gemm_strided(qh, 0, Cc, 1, T * Cc, kh, 0, 1, Cc, L * Cc, S, 0, L, H * T * L, T, L, d, B, hsa=d, hsb=d, hsc=T * L, heads=H)
        P = ops.softmax_rows(S, float(d ** -0.5))
        Ph = self.q(aq[3], P).reshape(B, H, T, L) if aq[3] is not None else P
        o = torch.empty(B, T, Cc, dtype=torch.float32, device=q.device)
        ops.gemm_strided(Ph, 0, L, 1, H * T * L, vh, 0, Cc, 1, L * Cc, o, 0, Cc, T * Cc, T, d, L, B, hsa=T * L, hsb=d, hsc=d, heads=H)
        return o, ("quant", P, Ph, qh, kh, vh)

    def _attn_bwd_quant(self, g_o, q, k, v, saved, aq, heads: int, grads):
        """-> dQ, dK, dV w.r.t. the UN-quantised q, k, v; the four deltas' gradients are written into `grads`."""
        _, P, Ph, qh, kh, vh = saved
        B, T, Cc = q.shape
        L, H = k.shape[1], heads
        d = Cc // H
        g_o = g_o.contiguous()
        dVh, dKh, dQh = torch.empty_like(vh), torch.empty_like(kh), torch.empty_like(qh)
        dPh = torch.empty_like(P)
        ops.gemm_strided(Ph, 0, 1, L, H * T * L, g_o, 0, Cc, 1, T * Cc, dVh, 0, Cc, L * Cc, L, d, T, B, hsa=T * L, hsb=d, hsc=d, heads=H)
        ops.gemm_strided(g_o, 0, Cc, 1, T * Cc, vh, 0, 1, Cc, L * Cc, dPh, 0, L, H * T * L, T, L, d, B, hsa=d, hsb=d, hsc=T * L, heads=H)
        if aq[3] is not None:
            dP, grads[aq[3]] = self.q_bwd(aq[3], P, dPh)
            dP = dP.reshape(P.shape)
        else:
            dP = dPh
        dS = ops.softmax_bwd_rows(P, dP.contiguous(), float(d ** -0.5))
        ops.gemm_strided(dS, 0, L, 1, H * T * L, kh, 0, Cc, 1, L * Cc, dQh, 0, Cc, T * Cc, T, d, L, B, hsa=T * L, hsb=d, hsc=d, heads=H)
        ops.gemm_strided(dS, 0, 1, L, H * T * L, qh, 0, Cc, 1, T * Cc, dKh, 0, Cc, L * Cc, L, d, T, B, hsa=T * L, hsb=d, hsc=d, heads=H)
        dQ, grads[aq[0]] = self.q_bwd(aq[0], q, dQh)
        dK, grads[aq[1]] = self.q_bwd(aq[1], k, dKh)
        dV, grads[aq[2]] = self.q_bwd(aq[2], v, dVh)
        return dQ.reshape(q.shape), dK.reshape(k.shape), dV.reshape(v.shape)

    def iterate(self, idx: torch.Tensor):
        import math
        self.count += 1
        if self.gemm_mode != "f32":
            with ops.gemm_precision(self.gemm_mode, self.delta.device.index):
                rec, grads = self._forward_backward(idx)
        else:
            rec, grads = self._forward_backward(idx)
        g = torch.cat([self._zero if gi is None else gi.reshape(1) for gi in grads]).contiguous()
        if self.world_size > 1 and self.allreduce is not None:
            self.allreduce(g)
        t = self.count
        lr_t = self.lr0 * 0.5 * (1.0 + math.cos(math.pi * (t - 1) / self.iters))       # CosineAnnealingLR, eta_min = 0, after t-1 steps
        b1, b2, eps = 0.9, 0.999, 1e-8
        self.m.mul_(b1).add_(g, alpha=1.0 - b1)
        self.v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (self.v.sqrt() / math.sqrt(1.0 - b2 ** t)).add_(eps)
        self.delta.addcdiv_(self.m, denom, value=-lr_t / (1.0 - b1 ** t))
        return rec, self._zero

    def losses(self, rec, rl):
        r = float(rec)
        return r, r, 0.0

    fisher = None
    _loss = _rec_loss


def _fixed_conv_fwd(x, L: FixedLayer, pad, rowadd=None, residual=None):
    B, H, W, _ = x.shape
    col = x.reshape(B * H * W, L.cin) if L.kh * L.kw == 1 else ops.im2col(x, L.kh, L.kw, 1, pad)
    y = ops.gemm(col, L.wg, trans_b=True, bias=L.bias, rowadd=rowadd, rows_per_img=H * W,
                 residual=None if residual is None else residual.reshape(B * H * W, -1))
    return y.reshape(B, H, W, L.cout)


def _fixed_conv_bwd_input(g, L: FixedLayer, shape, pad):
    """dL/d(input) of a fixed conv / linear: g [B,H,W,cout] -> [B,H,W,cin]"""
    B, H, W, _ = shape
    dcol = ops.gemm(g.reshape(B * H * W, L.cout), L.wg)
    return dcol.reshape(B, H, W, L.cin) if L.kh * L.kw == 1 else ops.col2im(dcol, (B, H, W, L.cin), L.kh, L.kw, 1, pad)


class DeltaLayerUnit(_DeltaUnit):
    """layer_reconstruction(use_aq=True) (reference :36-48): one delta."""

    def __init__(self, layer: FixedLayer, x, y, pad=(1, 1, 1, 1), **kw):
        super().__init__(**kw)
        self.layer, self.x, self.y, self.pad = layer, x, y, pad

    def _forward_backward(self, idx):
        L = self.layer
        x, y = self.x.index_select(0, idx), self.y.index_select(0, idx)
        xq = self.q(L.qi, x) if L.qi is not None else x
        out = _fixed_conv_fwd(xq, L, self.pad)
        loss, g = self._loss(out, y, out.numel() // out.shape[-1], idx)
        if L.qi is None:
            return loss, []
        g_xq = _fixed_conv_bwd_input(g, L, x.shape, self.pad)
        _, gd = self.q_bwd(L.qi, x, g_xq, want_gx=False)
        return loss, [gd]


class DeltaResnetUnit(_DeltaUnit):
    """QuantResnetBlock / QuantResBlock under use_aq=True: the deltas of conv1 and conv2 (temb_proj / emb_layers.1 is `quant_emb`: its
    input IS quantised in the forward -- the caller passes the projection computed that way -- but its delta is not trained)."""

    def __init__(self, conv1: FixedLayer, conv2: FixedLayer, gn1, gn2, shortcut, x, proj, y, eps: float = 1e-6, **kw):
        super().__init__(**kw)
        self.c1, self.c2, self.gn1, self.gn2, self.shortcut, self.eps = conv1, conv2, gn1, gn2, shortcut, eps
        self.x, self.proj, self.y = x, proj, y

    def _forward_backward(self, idx):
        c1l, c2l = self.c1, self.c2
        x, proj, y = self.x.index_select(0, idx), self.proj.index_select(0, idx), self.y.index_select(0, idx)
        B, H, W, cin = x.shape
        _, a1, _ = ops.groupnorm(x, self.gn1[0], self.gn1[1], self.eps, True, want_f32=True)
        a1q = self.q(c1l.qi, a1) if c1l.qi is not None else a1
        c1 = _fixed_conv_fwd(a1q, c1l, (1, 1, 1, 1), rowadd=proj)
        _, a2, _ = ops.groupnorm(c1, self.gn2[0], self.gn2[1], self.eps, True, want_f32=True)
        a2q = self.q(c2l.qi, a2) if c2l.qi is not None else a2
        if self.shortcut is not None:
            sc = ops.gemm(x.reshape(B * H * W, cin), self.shortcut[0], trans_b=True, bias=self.shortcut[1]).reshape(B, H, W, -1)
        else:
            sc = x
        out = _fixed_conv_fwd(a2q, c2l, (1, 1, 1, 1), residual=sc)
        loss, g = self._loss(out, y, B * H * W, idx)
        grads = [None] * self.delta.numel()
        g_a2q = _fixed_conv_bwd_input(g, c2l, a2.shape, (1, 1, 1, 1))
        if c2l.qi is not None:
            g_a2, grads[c2l.qi] = self.q_bwd(c2l.qi, a2, g_a2q)
        else:
            g_a2 = g_a2q
        if c1l.qi is not None:
            g_c1 = ops.groupnorm_bwd(c1, g_a2, self.gn2[0], self.gn2[1], self.eps, True)
            g_a1q = _fixed_conv_bwd_input(g_c1, c1l, a1.shape, (1, 1, 1, 1))
            _, grads[c1l.qi] = self.q_bwd(c1l.qi, a1, g_a1q, want_gx=False)
        return loss, grads


class DeltaAttnUnit(_DeltaUnit):
    """QuantAttnBlock under use_aq=True with the attention-matmul quantizers off (the state every driver leaves them in): the deltas
    of q, k, v (three quantizers on the same normalised input) and proj_out."""

    def __init__(self, q: FixedLayer, k: FixedLayer, v: FixedLayer, po: FixedLayer, gn, x, y, attn_q=None, **kw):
        super().__init__(**kw)
        self.ls, self.gn, self.x, self.y = (q, k, v, po), gn, x, y
        self.attn_q = attn_q          # (iq, ik, iv, iw | None): the block's own matmul quantizers are live (use_aq set by hand)

    def _forward_backward(self, idx):
        if self.attn_q is not None:
            return self._forward_backward_quant(idx)
        ql, kl, vl, pl = self.ls
        x, y = self.x.index_select(0, idx), self.y.index_select(0, idx)
        B, H, W, Cc = x.shape
        T = H * W
        scale = float(int(Cc) ** (-0.5))
        _, hn, _ = ops.groupnorm(x, self.gn[0], self.gn[1], 1e-6, False, want_f32=True)
        hf = hn.reshape(B * T, Cc)

        def lin(L, inp):
            return ops.gemm(self.q(L.qi, inp) if L.qi is not None else inp, L.wg, trans_b=True, bias=L.bias)
        q, k, v = (lin(L, hf).reshape(B, T, Cc) for L in (ql, kl, vl))
        S = ops.gemm(q, k, trans_b=True)
        P = ops.softmax_rows(S, scale)
        o = ops.gemm(P, v).reshape(B * T, Cc)
        oq = self.q(pl.qi, o) if pl.qi is not None else o
        out = ops.gemm(oq, pl.wg, trans_b=True, bias=pl.bias, residual=x.reshape(B * T, Cc)).reshape(B, H, W, Cc)
        loss, g = self._loss(out, y, B * T, idx)
        grads = [None] * self.delta.numel()
        g_oq = ops.gemm(g.reshape(B * T, Cc), pl.wg)
        if pl.qi is not None:
            g_o, grads[pl.qi] = self.q_bwd(pl.qi, o, g_oq)
        else:
            g_o = g_oq
        g_o = g_o.reshape(B, T, Cc)
        dV = ops.gemm(P, g_o, trans_a=True)
        dP = ops.gemm(g_o, v, trans_b=True)
        dS = ops.softmax_bwd_rows(P, dP, scale)
        dQ = ops.gemm(dS, k)
        dK = ops.gemm(dS, q, trans_a=True)
        for L, dd in ((ql, dQ), (kl, dK), (vl, dV)):
            if L.qi is not None:
                g_in = ops.gemm(dd.reshape(B * T, Cc), L.wg)           # dL/d(quantised input of this projection)
                _, grads[L.qi] = self.q_bwd(L.qi, hf, g_in, want_gx=False)
        return loss, grads


    def _forward_backward_quant(self, idx):
        """QuantAttnBlock.forward with use_aq (quant_block.py:483-500): q^, k^ quantised before the score product, v^ and the softmax
        (zero point 0) before the second; one head of C channels, scale C^-1/2."""
        ql, kl, vl, pl = self.ls
        x, y = self.x.index_select(0, idx), self.y.index_select(0, idx)
        B, H, W, Cc = x.shape
        T = H * W
        _, hn, _ = ops.groupnorm(x, self.gn[0], self.gn[1], 1e-6, False, want_f32=True)
        hf = hn.reshape(B * T, Cc)

        def lin(L, inp):
            return ops.gemm(self.q(L.qi, inp) if L.qi is not None else inp, L.wg, trans_b=True, bias=L.bias)
        q, k, v = (lin(L, hf).reshape(B, T, Cc) for L in (ql, kl, vl))
        o, saved = self._attn_fwd_quant(q, k, v, self.attn_q, 1)
        o = o.reshape(B * T, Cc)
        oq = self.q(pl.qi, o) if pl.qi is not None else o
        out = ops.gemm(oq, pl.wg, trans_b=True, bias=pl.bias, residual=x.reshape(B * T, Cc)).reshape(B, H, W, Cc)
        loss, g = self._loss(out, y, B * T, idx)
        grads = [None] * self.delta.numel()
        g_oq = ops.gemm(g.reshape(B * T, Cc), pl.wg)
        if pl.qi is not None:
            g_o, grads[pl.qi] = self.q_bwd(pl.qi, o, g_oq)
        else:
            g_o = g_oq
        dQ, dK, dV = self._attn_bwd_quant(g_o.reshape(B, T, Cc), q, k, v, saved, self.attn_q, 1, grads)
        for L, dd in ((ql, dQ), (kl, dK), (vl, dV)):
            if L.qi is not None:
                g_in = ops.gemm(dd.reshape(B * T, Cc), L.wg)
                _, grads[L.qi] = self.q_bwd(L.qi, hf, g_in, want_gx=False)
        return loss, grads


class DeltaQKUnit(_DeltaUnit):
    """block_reconstruction(use_aq=True) on a QuantQKMatMul (reference quant/reconstruction.py:155-156 on quant_block.py:303-328):
    weight = aqtizer_q(q s) aqtizer_k(k s)^T with s = d^-1/4; the two deltas are the only parameters.  q, k [N, T, heads d] (head-major
    channels), target [N, heads, T, T].  The reference's loss sums over dim 1 of [(b h), T, T] and averages the rest: sum / (N heads T)."""

    def __init__(self, q, k, y, heads: int, pre: Optional[float] = None, **kw):
        super().__init__(**kw)
        d = q.shape[-1] // heads
        pre = float(d ** -0.25) if pre is None else float(pre)
        self.heads = heads

        def scaled(x):
            xs = torch.zeros_like(x, dtype=torch.float32)
            ops.axpy(xs, x.float().contiguous(), pre)
            return xs
        self.qs, self.ks, self.y = scaled(q), scaled(k), y        # the quantizers see the SCALED tensors (quant_block.py:318-319)

    def _forward_backward(self, idx):
        qs, ks, y = self.qs.index_select(0, idx), self.ks.index_select(0, idx), self.y.index_select(0, idx)
        B, T, Cc = qs.shape
        H, L = self.heads, ks.shape[1]
        d = Cc // H
        qh, kh = self.q(0, qs).reshape(B, T, Cc), self.q(1, ks).reshape(B, L, Cc)
        S = torch.empty(B, H, T, L, dtype=torch.float32, device=qs.device)
        ops.gemm_strided(qh, 0, Cc, 1, T * Cc, kh, 0, 1, Cc, L * Cc, S, 0, L, H * T * L, T, L, d, B, hsa=d, hsb=d, hsc=T * L, heads=H)
        loss, g = self._loss(S, y, B * H * L, idx)
        g = g.contiguous()
        dQh, dKh = torch.empty_like(qh), torch.empty_like(kh)
        ops.gemm_strided(g, 0, L, 1, H * T * L, kh, 0, Cc, 1, L * Cc, dQh, 0, Cc, T * Cc, T, d, L, B, hsa=T * L, hsb=d, hsc=d, heads=H)
        ops.gemm_strided(g, 0, 1, L, H * T * L, qh, 0, Cc, 1, T * Cc, dKh, 0, Cc, L * Cc, L, d, T, B, hsa=T * L, hsb=d, hsc=d, heads=H)
        grads = [None, None]
        _, grads[0] = self.q_bwd(0, qs, dQh, want_gx=False)
        _, grads[1] = self.q_bwd(1, ks, dKh, want_gx=False)
        return loss, grads


class DeltaSMVUnit(_DeltaUnit):
    """block_reconstruction(use_aq=True) on a QuantSMVMatMul (reference quant/reconstruction.py:157-160 on quant_block.py:331-354):
    a = aqtizer_w(weight) aqtizer_v(v); deltas [v, w] (w left out when it is a 16-bit quantizer).  weight [N, heads, T, L] (softmax rows),
    v [N, L, heads d], target [N, T, heads d]; loss = sum / (N heads T) (dim 1 of the reference's [(b h), ch, T] is the channel)."""

    def __init__(self, w, v, y, heads: int, has_w: bool, **kw):
        super().__init__(**kw)
        self.wt, self.val, self.y, self.heads, self.has_w = w, v, y, heads, has_w        # (self.v / self.m are Adam's moments)

    def _forward_backward(self, idx):
        w, v, y = self.wt.index_select(0, idx), self.val.index_select(0, idx).contiguous(), self.y.index_select(0, idx)
        B, H, T, L = w.shape
        Cc = v.shape[-1]
        d = Cc // H
        vh = self.q(0, v).reshape(B, L, Cc)
        wh = self.q(1, w).reshape(B, H, T, L) if self.has_w else w.contiguous()
        o = torch.empty(B, T, Cc, dtype=torch.float32, device=v.device)
        ops.gemm_strided(wh, 0, L, 1, H * T * L, vh, 0, Cc, 1, L * Cc, o, 0, Cc, T * Cc, T, d, L, B, hsa=T * L, hsb=d, hsc=d, heads=H)
        loss, g = self._loss(o, y, B * H * T, idx)
        g = g.contiguous()
        dVh = torch.empty_like(vh)
        ops.gemm_strided(wh, 0, 1, L, H * T * L, g, 0, Cc, 1, T * Cc, dVh, 0, Cc, L * Cc, L, d, T, B, hsa=T * L, hsb=d, hsc=d, heads=H)
        grads = [None] * (2 if self.has_w else 1)
        _, grads[0] = self.q_bwd(0, v, dVh, want_gx=False)
        if self.has_w:
            dWh = torch.empty_like(wh)
            ops.gemm_strided(g, 0, Cc, 1, T * Cc, vh, 0, 1, Cc, L * Cc, dWh, 0, L, H * T * L, T, L, d, B, hsa=d, hsb=d, hsc=T * L, heads=H)
            _, grads[1] = self.q_bwd(1, w.contiguous(), dWh, want_gx=False)
        return loss, grads


class DeltaTransformerUnit(_DeltaUnit):
    """QuantBasicTransformerBlock under use_aq=True with the attention-matmul quantizers off: the deltas of its ten QuantLayers
    (module order: attn1.{to_q,to_k,to_v,to_out.0}, ff.net.0.proj, ff.net.2, attn2.{to_q,to_k,to_v,to_out.0}); same dataflow and
    fused exact-fp32 attention as TransformerUnit, gradients taken w.r.t. the layers' (quantised) inputs instead of their weights."""

    _attn_fwd = TransformerUnit._attn_fwd
    _attn_bwd = TransformerUnit._attn_bwd
    use_flash = True

    def __init__(self, layers: Sequence[FixedLayer], norms, heads: int, x, ctx, y, attn_q1=None, attn_q2=None, **kw):
        super().__init__(**kw)
        assert len(layers) == 10
        self.ls, self.norms, self.heads = list(layers), norms, heads
        self.x, self.ctx, self.y = x, ctx, y
        self.attn_q1, self.attn_q2 = attn_q1, attn_q2      # (iq, ik, iv, iw | None) of attn1 / attn2 when their use_aq was set by hand

    def _forward_backward(self, idx):
        (q1l, k1l, v1l, o1l, f0l, f2l, q2l, k2l, v2l, o2l) = self.ls
        x, ctx, y = self.x.index_select(0, idx), self.ctx.index_select(0, idx), self.y.index_select(0, idx)
        B, T, Cc = x.shape
        L, Dc = ctx.shape[1], ctx.shape[2]
        (g1, b1), (g2, b2), (g3, b3) = self.norms
        grads = [None] * self.delta.numel()

        def lin(Ly, inp, residual=None):
            return ops.gemm(self.q(Ly.qi, inp) if Ly.qi is not None else inp, Ly.wg, trans_b=True, bias=Ly.bias, residual=residual)

        def back(Ly, inp, g, want_gx=True):
            """g = dL/d(layer output) -> dL/d(layer input before its quantizer); records the layer's delta gradient"""
            g_in = ops.gemm(g, Ly.wg)
            if Ly.qi is None:
                return g_in
            gx, grads[Ly.qi] = self.q_bwd(Ly.qi, inp, g_in, want_gx)
            return gx
        x2d, c2d = x.reshape(B * T, Cc), ctx.reshape(B * L, Dc).contiguous()
        # ---- forward
        n1 = ops.layernorm(x, g1, b1, 1e-5, None)[1].reshape(B * T, Cc)
        q1, k1, v1 = (lin(Ly, n1).reshape(B, T, Cc) for Ly in (q1l, k1l, v1l))
        o1, P1 = self._attn_fwd(q1, k1, v1) if self.attn_q1 is None else self._attn_fwd_quant(q1, k1, v1, self.attn_q1, self.heads)
        o1 = o1.reshape(B * T, Cc)
        x1 = lin(o1l, o1, residual=x2d)
        n2 = ops.layernorm(x1.reshape(B, T, Cc), g2, b2, 1e-5, None)[1].reshape(B * T, Cc)
        q2 = lin(q2l, n2).reshape(B, T, Cc)
        k2, v2 = lin(k2l, c2d).reshape(B, L, Cc), lin(v2l, c2d).reshape(B, L, Cc)
        o2, P2 = self._attn_fwd(q2, k2, v2) if self.attn_q2 is None else self._attn_fwd_quant(q2, k2, v2, self.attn_q2, self.heads)
        o2 = o2.reshape(B * T, Cc)
        x2 = lin(o2l, o2, residual=x1)
        n3 = ops.layernorm(x2.reshape(B, T, Cc), g3, b3, 1e-5, None)[1].reshape(B * T, Cc)
        hcat = lin(f0l, n3)
        gg = ops.geglu(hcat, None)[1]
        out = lin(f2l, gg, residual=x2)
        loss, g = self._loss(out.reshape(B, T, Cc), y, B * Cc, idx)
        # ---- backward to the quantizer inputs
        g_out = g.reshape(B * T, Cc)
        d_gg = back(f2l, gg, g_out)
        d_hcat = ops.geglu_bwd(hcat, d_gg)
        d_n3 = back(f0l, n3, d_hcat)
        d_x2 = ops.layernorm_bwd(x2.reshape(B, T, Cc), d_n3.reshape(B, T, Cc), g3, 1e-5).reshape(B * T, Cc)
        ops.axpy(d_x2, g_out, 1.0)
        g_o2 = back(o2l, o2, d_x2).reshape(B, T, Cc)
        dQ2, dK2, dV2 = (self._attn_bwd(g_o2, q2, k2, v2, P2) if self.attn_q2 is None
                         else self._attn_bwd_quant(g_o2, q2, k2, v2, P2, self.attn_q2, self.heads, grads))
        d_n2 = back(q2l, n2, dQ2.reshape(B * T, Cc))
        back(k2l, c2d, dK2.reshape(B * L, Cc), want_gx=False)
        back(v2l, c2d, dV2.reshape(B * L, Cc), want_gx=False)
        d_x1 = ops.layernorm_bwd(x1.reshape(B, T, Cc), d_n2.reshape(B, T, Cc), g2, 1e-5).reshape(B * T, Cc)
        ops.axpy(d_x1, d_x2, 1.0)
        g_o1 = back(o1l, o1, d_x1).reshape(B, T, Cc)
        dQ1, dK1, dV1 = (self._attn_bwd(g_o1, q1, k1, v1, P1) if self.attn_q1 is None
                         else self._attn_bwd_quant(g_o1, q1, k1, v1, P1, self.attn_q1, self.heads, grads))
        for Ly, dd in ((q1l, dQ1), (k1l, dK1), (v1l, dV1)):
            back(Ly, n1, dd.reshape(B * T, Cc), want_gx=False)
        return loss, grads
