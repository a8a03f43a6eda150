"""layer_ / block_ / tib_reconstruction with the reference's signatures
(quant/reconstruction.py:13-29,86-102,212-226), executed by engine.recon units on the device.

Per unit: wrap the unit's weight quantizers in AdaRoundQuantizer (soft targets), cache the unit's
inputs / FP targets for the calibration set (save_inout), run `iters` Adam iterations on random
mini-batches -- same RNG calls as the reference (`torch.randperm(n)[:batch_size]` per iteration) -- with
one SUM all-reduce of the flattened gradient buffer per iteration when `multi_gpu`, then switch the
quantizers to hard rounding.
"""
from __future__ import annotations

import logging
from typing import Optional, Tuple

import torch

from tfmq_dm_amd import ops
from tfmq_dm_amd._lib import TfmqError
from tfmq_dm_amd.engine import recon as R
from .adaptive_rounding import AdaRoundQuantizer, RMODE
from .data_utill import save_inout, save_grad
from .quant_block import (BaseQuantBlock, QuantAttnBlock, QuantBasicTransformerBlock, QuantQKMatMul, QuantResBlock, QuantResnetBlock,
                          QuantSMVMatMul, QuantTemporalInformationBlock, QuantTemporalInformationBlockDDIM)
from .quant_layer import QuantLayer, StraightThrough
from .reconstruction_util import RLOSS, LossFunc, LossFuncTimeEmbedding, fisher_mode

logger = logging.getLogger(__name__)


def _dist_kw(multi_gpu: bool):
    if not multi_gpu:
        return dict(world_size=1, allreduce=None)
    from tfmq_dm_amd import linklink as link
    return dict(world_size=link.get_world_size(), allreduce=link.allreduce)


def _to_adaround(layer: QuantLayer) -> AdaRoundQuantizer:
    """module.wqtizer = AdaRoundQuantizer(uaqtizer=..., w=original_w) with soft targets on."""
    d, z, _ = layer.weight_quant_state()
    # the reference wraps unconditionally (reconstruction.py:49-52,113-128): a layer that already carries a learned AdaRoundQuantizer
    # (a checkpoint loaded with load_cali_model, then reconstructed again) starts over from the soft initialisation of original_w
    layer.wqtizer = AdaRoundQuantizer(uaqtizer=layer.wqtizer, rmode=RMODE.LEARNED_HARD_SIGMOID,
                                      w=layer.original_w.data.to(layer.w.device))
    layer.wqtizer.soft_tgt = True
    return layer.wqtizer


def _ada_layer(layer: QuantLayer) -> R.AdaLayer:
    q = _to_adaround(layer)
    return R.AdaLayer(layer.w.data, q.delta, q.zero_point, None if layer.b is None else layer.b.data, q.level, alpha=q.alpha.data)


def _commit(layer: QuantLayer, ada: R.AdaLayer):
    """Write the learned alpha back into the module (state-dict key `...wqtizer.alpha`) and harden."""
    q = layer.wqtizer
    q.alpha.data.copy_(ada.alpha)
    q.soft_tgt = False
    q._version_ += 1


def _hard_weight(layer: QuantLayer) -> torch.Tensor:
    """The layer's weight as its CURRENT weight quantizer produces it (hard AdaRound or nearest), fp32, the module's layout."""
    d, z, a = layer.weight_quant_state()
    w = layer.w.data.float().contiguous()
    if a is not None:
        return ops.adaround_soft_fwd(w, a.float().contiguous(), d, z, layer.wqtizer.level, hard=True)
    return ops.fake_quant(w, d.reshape(-1), z.reshape(-1), layer.wqtizer.level)


class _DeltaSet:
    """The trainable activation deltas of a unit under use_aq=True (reference reconstruction.py:36-48,135-166): every QuantLayer that
    is not `quant_emb`, quantises its input (use_aq and not disable_aq) and has an initialised, non-zero delta."""

    def __init__(self):
        self.layers, self.deltas, self.zps, self.levels = [], [], [], []
        self.extra = []          # (index, attention-matmul quantizer)

    def fixed(self, layer: QuantLayer) -> R.FixedLayer:
        q, qi = layer.aqtizer, None
        if layer.use_aq and not layer.disable_aq:
            if q.delta is None:
                raise TfmqError("delta-learning reconstruction: run a forward with use_aq=True first (an activation quantizer of the unit is uninitialised)")
            if not layer.quant_emb and bool(q.delta != 0):
                qi = len(self.layers)
                self.layers.append(layer)
                self.deltas.append(q.delta.data)
                zp = q.zero_point
                self.zps.append(zp.detach() if torch.is_tensor(zp) else torch.tensor(float(zp), device=q.delta.device))
                self.levels.append(q.level)
        return R.FixedLayer(_hard_weight(layer), None if layer.b is None else layer.b.data, qi)

    def attn(self, owner) -> Optional[tuple]:
        """The live attention-matmul quantizers of `owner` (a QuantAttnBlock, or attn1 / attn2 of a QuantBasicTransformerBlock whose
        `use_aq` was set by hand) as trainable deltas -- the reference's `A` lists (reconstruction.py:145-163): aqtizer_q, _k, _v, and
        aqtizer_w unless it is a 16-bit quantizer.  -> (iq, ik, iv, iw | None) indices into the unit's delta vector, or None."""
        if not getattr(owner, "use_aq", False):
            return None
        idx = []
        for r in "qkvw":
            q = getattr(owner, f"aqtizer_{r}")
            if r == "w" and q.level == 2 ** 16:
                idx.append(None)
                continue
            if q.delta is None or not bool(q.delta != 0):
                raise TfmqError("delta-learning reconstruction: an attention-matmul quantizer is live but uninitialised (run a forward with its use_aq set first)")
            idx.append(len(self.deltas))
            self.layers.append(None)
            self.extra.append((len(self.deltas), q))
            self.deltas.append(q.delta.data)
            zp = q.zero_point
            self.zps.append(zp.detach() if torch.is_tensor(zp) else torch.tensor(float(zp), device=q.delta.device))
            self.levels.append(q.level)
        return tuple(idx)

    def quantizers(self, owner, roles: str) -> int:
        """aqtizer_<r> of a stand-alone matmul module (QuantQKMatMul: "qk", QuantSMVMatMul: "vw") as trainable deltas, in the reference's
        order (reconstruction.py:155-160); a 16-bit aqtizer_w is left out.  -> how many were registered."""
        n = 0
        for r in roles:
            q = getattr(owner, f"aqtizer_{r}")
            if r == "w" and q.level == 2 ** 16:
                continue
            if q.delta is None or not bool(q.delta != 0):
                raise TfmqError("delta-learning reconstruction: a matmul quantizer of the unit is uninitialised (run a forward with its use_aq set first)")
            self.layers.append(None)
            self.extra.append((len(self.deltas), q))
            self.deltas.append(q.delta.data)
            zp = q.zero_point
            self.zps.append(zp.detach() if torch.is_tensor(zp) else torch.tensor(float(zp), device=q.delta.device))
            self.levels.append(q.level)
            n += 1
        return n

    def kw(self, iters, lr, multi_gpu):
        return dict(deltas=self.deltas, zps=self.zps, levels=self.levels, iters=iters, lr=lr, **_dist_kw(multi_gpu))

    def commit(self, unit: R._DeltaUnit):
        for i, layer in enumerate(self.layers):
            if layer is not None:
                layer.aqtizer.delta.data.copy_(unit.delta[i].reshape(layer.aqtizer.delta.shape))
        for i, q in self.extra:
            q.delta.data.copy_(unit.delta[i].reshape(q.delta.shape))


def _quant_emb_projection(layer: QuantLayer, emb: torch.Tensor) -> torch.Tensor:
    """temb_proj / emb_layers.1 of a ResBlock under the block's quant state, input silu(emb) (its own activation quantizer applied when
    live -- the layer is `quant_emb`, so its delta is used but never trained here)."""
    s_ = ops.silu(emb.float().contiguous())
    q = layer.aqtizer
    if layer.use_aq and not layer.disable_aq and q.delta is not None:
        zp = q.zero_point
        zp = zp.detach() if torch.is_tensor(zp) else torch.tensor(float(zp), device=emb.device)
        s_ = ops.fake_quant(s_, q.delta.data.reshape(1), zp.reshape(1), q.level)
    w = _hard_weight(layer).reshape(layer.w.shape[0], -1).contiguous()
    return ops.gemm(s_, w, trans_b=True, bias=None if layer.b is None else layer.b.data.float().contiguous())


def _attach_fisher(unit, model, layer, cali_data, opt_mode, asym, use_aq, batch_size, keep_gpu):
    """opt_mode != MSE (reference :58-61,177-180): cache |dL/d(unit output)| + 1 of the whole calibration set (save_grad) and let the
    unit's loss kernel weight the reconstruction error with it."""
    if opt_mode == RLOSS.MSE:
        return
    unit.fisher = (fisher_mode(opt_mode), save_grad(model, layer, cali_data, asym, use_aq, batch_size, keep_gpu))


LOSS_TRACE = None     # tests: {"counts": (...), "rows": [], "unit": 0} -> rows of (unit index, count, rec, round) at those counts


IDX_CHUNK = 256      # mini-batch index vectors drawn and uploaded per host -> device copy


def _run(unit: R._Unit, n: int, batch_size: int, iters: int, loss_func: LossFunc, device, rank0=True):
    idx_dev = None
    for it in range(iters):
        # torch.randperm(n)[:batch_size] per iteration, in iteration order: the same host RNG stream as the reference (:70, :188).  Round 5:
        # the draws of IDX_CHUNK iterations are made together and travel in ONE copy -- a pageable host -> device copy per iteration is a
        # stream-ordered blocking call, i.e. a host / GPU rendezvous in every iteration (nothing else in an iteration draws host random numbers)
        if it % IDX_CHUNK == 0:
            m = min(IDX_CHUNK, iters - it)
            idx_host = torch.stack([torch.randperm(n)[:batch_size] for _ in range(m)])
            idx_dev = idx_host.to(device)
        idx = idx_dev[it % IDX_CHUNK]
        idx._host = idx_host[it % IDX_CHUNK]      # (caches in pinned host memory -- data_utill.HostRows -- select their rows on the host)
        b, active = loss_func.tick()
        rec, rl = unit.iterate(idx)
        if LOSS_TRACE is not None and loss_func.count in LOSS_TRACE["counts"]:
            tot, r, q = unit.losses(rec, rl)
            LOSS_TRACE["rows"].append((LOSS_TRACE["unit"], loss_func.count, r, q))
        if loss_func.count % 2000 == 0:
            tot, r, q = unit.losses(rec, rl)
            loss_func.log(tot, r, q, b, rank0)
    if LOSS_TRACE is not None:
        LOSS_TRACE["unit"] += 1


def layer_reconstruction(model, layer: QuantLayer, cali_data: Tuple[torch.Tensor], batch_size: int = 128,
                         iters: int = 20000, w: float = 0.001, opt_mode: RLOSS = RLOSS.MSE, asym: bool = False,
                         include_act_func: bool = True, b_range: tuple = (20, 2), warmup: float = 0.0,
                         use_aq: bool = False, lr: float = 4e-5, p: float = 2.0, multi_gpu: bool = False,
                         keep_gpu=True) -> None:
    model.set_quant_state(use_wq=False, use_aq=False)
    layer.set_quant_state(use_wq=True, use_aq=use_aq)
    if use_aq:
        # delta learning (reference :36-48): the layer's activation delta under Adam(lr) + cosine annealing, weights fixed
        ds = _DeltaSet()
        fl = ds.fixed(layer)
        loss_func = LossFunc(o=layer, round_loss=RLOSS.NONE, w=w, max_count=iters, rec_loss=opt_mode, b_range=b_range,
                             decay_start=0.0, warmup=warmup, p=p)
        cached_inputs, cached_outputs = save_inout(model, layer, cali_data, asym, use_aq, batch_size, keep_gpu)
        if not ds.layers:
            return                      # `disable_aq` layer: the reference's optimiser has an un-used parameter and nothing moves
        ph, pw = layer.fwd_kwargs.get("padding", (0, 0))
        unit = R.DeltaLayerUnit(fl, cached_inputs[0], cached_outputs, pad=(ph, pw, ph, pw), **ds.kw(iters, lr, multi_gpu))
        _attach_fisher(unit, model, layer, cali_data, opt_mode, asym, use_aq, batch_size, keep_gpu)
        _run(unit, cached_inputs[0].size(0), batch_size, iters, loss_func, cached_outputs.device)
        ds.commit(unit)
        model.invalidate()
        return
    ada = _ada_layer(layer)
    loss_func = LossFunc(o=layer, round_loss=RLOSS.RELAXATION, w=w, max_count=iters, rec_loss=opt_mode, b_range=b_range,
                         decay_start=0.0, warmup=warmup, p=p)
    cached_inputs, cached_outputs = save_inout(model, layer, cali_data, asym, use_aq, batch_size, keep_gpu)
    ph, pw = layer.fwd_kwargs.get("padding", (0, 0))
    unit = R.LayerUnit(ada, cached_inputs[0], cached_outputs, pad=(ph, pw, ph, pw), iters=iters, w=w, warmup=warmup, b_range=b_range,
                       **_dist_kw(multi_gpu))
    _attach_fisher(unit, model, layer, cali_data, opt_mode, asym, use_aq, batch_size, keep_gpu)
    _run(unit, cached_inputs[0].size(0), batch_size, iters, loss_func, cached_outputs.device)
    _commit(layer, ada)
    model.invalidate()


def block_reconstruction(model, block: BaseQuantBlock, cali_data: torch.Tensor, batch_size: int = 32,
                         iters: int = 20000, w: float = 0.01, opt_mode: RLOSS = RLOSS.MSE, asym: bool = False,
                         include_act_func: bool = True, b_range: tuple = (20, 2), warmup: float = 0.0,
                         use_aq: bool = False, lr: float = 4e-5, p: float = 2.0, multi_gpu: bool = True,
                         keep_gpu=True) -> None:
    model.set_quant_state(use_wq=False, use_aq=False)
    block.set_quant_state(use_wq=True, use_aq=use_aq)
    if use_aq:
        return _block_delta_learning(model, block, cali_data, batch_size, iters, w, opt_mode, asym, b_range, warmup, lr, p, multi_gpu, keep_gpu)
    if not any(isinstance(m, QuantLayer) and not m.quant_emb for m in block.modules()):
        return      # QuantAttentionBlock / QuantQKMatMul / QuantSMVMatMul: nothing to optimise (reference :130-131)
    loss_func = LossFunc(o=block, round_loss=RLOSS.RELAXATION, w=w, max_count=iters, rec_loss=opt_mode, b_range=b_range,
                         decay_start=0.0, warmup=warmup, p=p)
    dev = next(block.parameters()).device
    kw = dict(iters=iters, w=w, warmup=warmup, b_range=b_range, **_dist_kw(multi_gpu))
    if isinstance(block, QuantResnetBlock):
        adas = {"conv1": _ada_layer(block.conv1), "conv2": _ada_layer(block.conv2)}   # temb_proj is quant_emb: excluded
        cached_inputs, cached_outputs = save_inout(model, block, cali_data, asym, use_aq, batch_size, keep_gpu)
        x, temb = cached_inputs
        # frozen temb projection under the block's quant state (hard-rounded by the TIB unit)
        tp = block.temb_proj
        d, z, a = tp.weight_quant_state()
        pk = ops.pack_w4(tp.w.data.float().contiguous(), d, z, None if a is None else a.contiguous(), tp.b.data)
        proj = ops.linear_small_w4(temb.contiguous(), pk, ops.qsel(None), silu_in=True)
        sc = None
        if hasattr(block, "nin_shortcut"):
            ns = block.nin_shortcut
            sc = (ns.weight.data.reshape(ns.weight.shape[0], -1).float().contiguous(), ns.bias.data.float().contiguous())
        unit = R.ResnetUnit(adas["conv1"], adas["conv2"],
                            (block.norm1.weight.data.float(), block.norm1.bias.data.float()),
                            (block.norm2.weight.data.float(), block.norm2.bias.data.float()), sc, x, proj, cached_outputs, **kw)
        layers = [(block.conv1, adas["conv1"]), (block.conv2, adas["conv2"])]
    elif isinstance(block, QuantAttnBlock):
        names = ("q", "k", "v", "proj_out")
        adas = {n: _ada_layer(getattr(block, n)) for n in names}
        cached_inputs, cached_outputs = save_inout(model, block, cali_data, asym, use_aq, batch_size, keep_gpu)
        unit = R.AttnUnit(adas["q"], adas["k"], adas["v"], adas["proj_out"],
                          (block.norm.weight.data.float(), block.norm.bias.data.float()), cached_inputs[0], cached_outputs, **kw)
        layers = [(getattr(block, n), adas[n]) for n in names]
    elif isinstance(block, QuantResBlock):
        conv1, conv2 = block.in_layers[2], block.out_layers[3]
        adas = {"c1": _ada_layer(conv1), "c2": _ada_layer(conv2)}      # emb_layers.1 is quant_emb: excluded (TIB unit)
        cached_inputs, cached_outputs = save_inout(model, block, cali_data, asym, use_aq, batch_size, keep_gpu)
        x, emb = cached_inputs
        ep = block.emb_layers[1]                                       # frozen under the block's quant state
        d, z, a = ep.weight_quant_state()
        pk = ops.pack_w4(ep.w.data.float().contiguous(), d, z, None if a is None else a.contiguous(), ep.b.data)
        proj = ops.linear_small_w4(emb.contiguous(), pk, ops.qsel(None), silu_in=True)
        sc = None
        if isinstance(block.skip_connection, torch.nn.Conv2d):
            ns = block.skip_connection
            sc = (ns.weight.data.reshape(ns.weight.shape[0], -1).float().contiguous(), ns.bias.data.float().contiguous())
        n1, n2 = block.in_layers[0], block.out_layers[0]
        unit = R.ResnetUnit(adas["c1"], adas["c2"], (n1.weight.data.float(), n1.bias.data.float()),
                            (n2.weight.data.float(), n2.bias.data.float()), sc, x, proj, cached_outputs, eps=n1.eps, **kw)
        layers = [(conv1, adas["c1"]), (conv2, adas["c2"])]
    elif isinstance(block, QuantBasicTransformerBlock):
        mods = [block.attn1.to_q, block.attn1.to_k, block.attn1.to_v, block.attn1.to_out[0], block.ff.net[0].proj,
                block.ff.net[2], block.attn2.to_q, block.attn2.to_k, block.attn2.to_v, block.attn2.to_out[0]]
        adal = [_ada_layer(m) for m in mods]
        cached_inputs, cached_outputs = save_inout(model, block, cali_data, asym, use_aq, batch_size, keep_gpu)
        x, ctx = cached_inputs
        norms = [(n.weight.data.float().contiguous(), n.bias.data.float().contiguous()) for n in (block.norm1, block.norm2, block.norm3)]
        unit = R.TransformerUnit(adal, norms, block.attn1.heads, x, ctx, cached_outputs, **kw)
        layers = list(zip(mods, adal))
    else:
        raise TfmqError(f"block_reconstruction: no reconstruction unit for {type(block).__name__} yet")
    _attach_fisher(unit, model, block, cali_data, opt_mode, asym, use_aq, batch_size, keep_gpu)
    _run(unit, cached_inputs[0].size(0), batch_size, iters, loss_func, dev)
    for layer, ada in layers:
        _commit(layer, ada)
    model.invalidate()


def _block_delta_learning(model, block, cali_data, batch_size, iters, w, opt_mode, asym, b_range, warmup, lr, p, multi_gpu, keep_gpu):
    """block_reconstruction(use_aq=True) (reference :135-166): Adam(lr) + CosineAnnealingLR on the activation deltas of the block's
    QuantLayers, weights fixed.  Built for QuantResnetBlock, QuantResBlock, QuantAttnBlock, QuantBasicTransformerBlock; round 4: with the
    attention-matmul quantizers live (use_aq of the attention set by hand -- no driver does) their deltas are trained too, the
    reference's `A` lists (:145-163; fixture F25)."""
    loss_func = LossFunc(o=block, round_loss=RLOSS.NONE, w=w, max_count=iters, rec_loss=opt_mode, b_range=b_range,
                         decay_start=0.0, warmup=warmup, p=p)
    dev = next(block.parameters()).device
    ds = _DeltaSet()
    if isinstance(block, (QuantResnetBlock, QuantResBlock)):
        ddpm = isinstance(block, QuantResnetBlock)
        conv1, conv2 = (block.conv1, block.conv2) if ddpm else (block.in_layers[2], block.out_layers[3])
        f1, f2 = ds.fixed(conv1), ds.fixed(conv2)
        cached_inputs, cached_outputs = save_inout(model, block, cali_data, asym, True, batch_size, keep_gpu)
        x, emb = cached_inputs
        proj = _quant_emb_projection(block.temb_proj if ddpm else block.emb_layers[1], emb)
        ns = getattr(block, "nin_shortcut", None) if ddpm else (block.skip_connection if isinstance(block.skip_connection, torch.nn.Conv2d) else None)
        sc = None if ns is None else (ns.weight.data.reshape(ns.weight.shape[0], -1).float().contiguous(), ns.bias.data.float().contiguous())
        n1, n2 = (block.norm1, block.norm2) if ddpm else (block.in_layers[0], block.out_layers[0])
        if not ds.layers:
            return
        unit = R.DeltaResnetUnit(f1, f2, (n1.weight.data.float(), n1.bias.data.float()), (n2.weight.data.float(), n2.bias.data.float()),
                                 sc, x, proj, cached_outputs, eps=n1.eps, **ds.kw(iters, lr, multi_gpu))
    elif isinstance(block, QuantAttnBlock):
        fl = [ds.fixed(getattr(block, n)) for n in ("q", "k", "v", "proj_out")]
        cached_inputs, cached_outputs = save_inout(model, block, cali_data, asym, True, batch_size, keep_gpu)
        aq_live = ds.attn(block)            # round 4: the block's own matmul quantizers (use_aq set by hand) join the trained deltas
        if not ds.layers:
            return
        unit = R.DeltaAttnUnit(fl[0], fl[1], fl[2], fl[3], (block.norm.weight.data.float(), block.norm.bias.data.float()),
                               cached_inputs[0], cached_outputs, attn_q=aq_live, **ds.kw(iters, lr, multi_gpu))
    elif isinstance(block, QuantBasicTransformerBlock):
        mods = [block.attn1.to_q, block.attn1.to_k, block.attn1.to_v, block.attn1.to_out[0], block.ff.net[0].proj,
                block.ff.net[2], block.attn2.to_q, block.attn2.to_k, block.attn2.to_v, block.attn2.to_out[0]]
        fl = [ds.fixed(m) for m in mods]
        cached_inputs, cached_outputs = save_inout(model, block, cali_data, asym, True, batch_size, keep_gpu)
        a1, a2 = ds.attn(block.attn1), ds.attn(block.attn2)
        if not ds.layers:
            return
        x, ctx = cached_inputs
        norms = [(n.weight.data.float().contiguous(), n.bias.data.float().contiguous()) for n in (block.norm1, block.norm2, block.norm3)]
        unit = R.DeltaTransformerUnit(fl, norms, block.attn1.heads, x, ctx, cached_outputs, attn_q1=a1, attn_q2=a2, **ds.kw(iters, lr, multi_gpu))
    elif isinstance(block, (QuantQKMatMul, QuantSMVMatMul)):
        # the matmul seams of the LDM AttentionBlock as units of their own (reference :155-160; reachable by a direct call only): the
        # deltas of their quantizers are the only parameters.  Fixture F26.
        qk = isinstance(block, QuantQKMatMul)
        n = ds.quantizers(block, "qk" if qk else "vw")
        cached_inputs, cached_outputs = save_inout(model, block, cali_data, asym, True, batch_size, keep_gpu)
        # The reference's tensors at these seams are [(b h), ...]: its mini-batches draw (sample, head) ROWS, `batch_size` of N * heads
        # (reconstruction.py:185-189).  The engine's taps are [N, T, heads d] / [N, heads, T, T]: split the heads into rows, in the
        # reference's row order b * heads + h, and run the units with one head per row.
        def rows(x, heads):         # [N, T, heads d] -> [N heads, T, d]
            N_, T_, C_ = x.shape
            return x.reshape(N_, T_, heads, C_ // heads).permute(0, 2, 1, 3).reshape(N_ * heads, T_, C_ // heads).contiguous()
        if qk:
            Hh = cached_outputs.shape[1]
            S = cached_outputs.reshape(-1, 1, cached_outputs.shape[2], cached_outputs.shape[3]).contiguous()
            cached_inputs = (rows(cached_inputs[0], Hh), rows(cached_inputs[1], Hh))
            cached_outputs = S
            d_head = cached_inputs[0].shape[-1]
            unit = R.DeltaQKUnit(cached_inputs[0], cached_inputs[1], cached_outputs, 1, pre=float(d_head ** -0.25), **ds.kw(iters, lr, multi_gpu))
        else:
            Hh = cached_inputs[0].shape[1]
            Wt = cached_inputs[0].reshape(-1, 1, cached_inputs[0].shape[2], cached_inputs[0].shape[3]).contiguous()
            cached_inputs = (Wt, rows(cached_inputs[1], Hh))
            cached_outputs = rows(cached_outputs, Hh)
            unit = R.DeltaSMVUnit(cached_inputs[0], cached_inputs[1], cached_outputs, 1, n == 2, **ds.kw(iters, lr, multi_gpu))
    else:
        raise NotImplementedError(f"delta-learning reconstruction of {type(block).__name__} is not built (DESIGN.md section 7)")
    _attach_fisher(unit, model, block, cali_data, opt_mode, asym, True, batch_size, keep_gpu)
    _run(unit, cached_inputs[0].size(0), batch_size, iters, loss_func, dev)
    ds.commit(unit)
    model.invalidate()


def tib_reconstruction(block: BaseQuantBlock, cali_data: torch.Tensor, batch_size: int = 32, iters: int = 20000,
                       w: float = 0.01, opt_mode: RLOSS = RLOSS.MSE, asym: bool = False, include_act_func: bool = True,
                       b_range: tuple = (20, 2), warmup: float = 0.0, use_aq: bool = False, lr: float = 4e-5,
                       p: float = 2.0, multi_gpu: bool = True, keep_gpu=True) -> None:
    """Temporal-information-aware reconstruction (TIAR).  Gradients of FP-kept layers (temb.dense.0:
    alpha.grad is None in the reference, which then crashes in its multi-GPU loop) are
    simply absent from the all-reduced buffer."""
    if use_aq:
        raise NotImplementedError("delta-learning reconstruction (use_aq=True) is never requested by the drivers")
    ldm = isinstance(block, QuantTemporalInformationBlock)
    if not (ldm or isinstance(block, QuantTemporalInformationBlockDDIM)):
        raise TfmqError(f"tib_reconstruction: not a temporal-information block: {type(block).__name__}")
    assert opt_mode == RLOSS.MSE
    from tfmq_dm_amd.engine.tib import tib_forward_ddim, tib_forward_ldm
    dev = next(block.parameters()).device
    ts = cali_data[1].to(dev).float().contiguous()
    fwd = tib_forward_ldm if ldm else tib_forward_ddim
    # FP targets (save_inout(block, block, ...): the TIB is its own model there)
    block.set_quant_state(False, False)
    targets = list(fwd(block, ts))
    block.set_quant_state(use_wq=True, use_aq=use_aq)
    if ldm:      # time_embed = Linear, SiLU, Linear; emb_layers = SiLU, Linear (openaimodel.py:466-470,193-199)
        d0, d1 = block.t_emb[0], block.t_emb[2]
        projs = [seq[1] for seq in block.emb_layers]
        emb = ops.timestep_embedding(ts, block.model_channels, ldm_order=True)
    else:
        d0, d1 = block.temb.dense[0], block.temb.dense[1]
        projs = list(block.temb_projs)
        emb = ops.timestep_embedding(ts, block.ch)
    # every QuantLayer of the TIB is wrapped (state-dict parity), the first Linear stays FP (ignore_recon)
    _to_adaround(d0)
    ada1 = _ada_layer(d1)
    adap = [_ada_layer(pj) for pj in projs]
    h0 = ops.linear_small_f32(emb, d0.original_w.to(dev).float().contiguous(),
                              None if d0.original_b is None else d0.original_b.to(dev).float().contiguous())
    s0 = ops.silu(h0)
    loss_func = LossFuncTimeEmbedding(o=block, round_loss=RLOSS.RELAXATION, w=w, max_count=iters, rec_loss=opt_mode,
                                      b_range=b_range, decay_start=0.0, warmup=warmup, p=p)
    unit = R.TibUnit(ada1, adap, s0, targets, iters=iters, w=w, warmup=warmup, b_range=b_range, **_dist_kw(multi_gpu))
    _run(unit, ts.size(0), batch_size, iters, loss_func, dev)
    d0.wqtizer.soft_tgt = False
    _commit(d1, ada1)
    for pj, a in zip(projs, adap):
        _commit(pj, a)
