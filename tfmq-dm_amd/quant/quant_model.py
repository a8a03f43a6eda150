"""QuantModel (reference quant/quant_model.py): module-tree rewrite + lowering to the HIP engine.

Same constructor, state toggles and state-dict layout as the reference.  `forward` does not walk
the module tree: it lowers the tree once into a `DdimUNetEngine` plan (re-lowered only when the
quantisation state changes) and runs that on the device.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from tfmq_dm_amd import ops
from tfmq_dm_amd._lib import TfmqError
from .adaptive_rounding import AdaRoundQuantizer
from .quant_block import (BaseQuantBlock, QuantAttentionBlock, QuantAttnBlock, QuantBasicTransformerBlock, QuantQKMatMul,
                          QuantResBlock, QuantResnetBlock, QuantSMVMatMul, QuantTemporalInformationBlock,
                          QuantTemporalInformationBlockDDIM, b2qb)
from .quant_layer import QMODE, QuantLayer, StraightThrough, UniformAffineQuantizer


class QuantModel(nn.Module):

    def __init__(self, model: nn.Module, wq_params: dict = {}, aq_params: dict = {}, cali: bool = True, **kwargs) -> None:
        super().__init__()
        self.model = model
        self.softmax_a_bit = kwargs.get("softmax_a_bit", 8)
        self.in_channels = model.in_channels
        if hasattr(model, "image_size"):
            self.image_size = model.image_size
        self.B = b2qb(aq_params["leaf_param"])
        self.quant_module(self.model, wq_params, aq_params, aq_mode=kwargs.get("aq_mode", [QMODE.NORMAL.value]), prev_name=None)
        self.quant_block(self.model, wq_params, aq_params)
        if cali:
            self.get_tib(self.model, wq_params, aq_params)
        self._plan = None          # current (engine, act_names, qtable)
        self._plans = {}           # state key -> plan (FP / weight-only / w4a8 plans coexist during calibration)
        self._act_table = None     # [G, n_q, 2] FSC table installed by set_act_table()
        self._act_step = None

    # ------------------------------------------------------------------ tree rewrite (reference :31-84)
    def get_tib(self, module: nn.Module, wq_params: dict = {}, aq_params: dict = {}):
        for name, child in module.named_children():
            if name == "temb":
                self.tib = QuantTemporalInformationBlockDDIM(child, aq_params, self.model.ch)
            elif name == "time_embed":
                self.tib = QuantTemporalInformationBlock(child, aq_params, self.model.model_channels, None)
            elif isinstance(child, QuantResBlock):
                self.tib.add_emb_layer(child.emb_layers)
            elif isinstance(child, QuantResnetBlock):
                self.tib.add_temb_proj(child.temb_proj)
            else:
                self.get_tib(child, wq_params, aq_params)

    def quant_module(self, module: nn.Module, wq_params: dict = {}, aq_params: dict = {},
                     aq_mode: List[int] = [QMODE.NORMAL.value], prev_name: str = None) -> None:
        """Every Conv2d / Linear becomes a QuantLayer except shortcut / skip convs, down-sample
        convs and `op` (reference :57-58, "refer to PTQD"); temb / emb projections are `quant_emb`."""
        for name, child in module.named_children():
            quantisable = isinstance(child, (nn.Conv2d, nn.Linear)) and not isinstance(child, nn.Conv1d)
            excluded = ("skip" in name or "op" in name or "shortcut" in name or (prev_name == "downsample" and name == "conv"))
            if quantisable and not excluded:
                is_emb = (prev_name is not None and "emb_layers" in prev_name and "1" in name) or "temb_proj" in name
                setattr(module, name, QuantLayer(child, dict(wq_params), dict(aq_params), aq_mode=aq_mode, quant_emb=bool(is_emb)))
            elif isinstance(child, StraightThrough):
                continue
            else:
                self.quant_module(child, wq_params, aq_params, aq_mode=aq_mode, prev_name=name)

    def quant_block(self, module: nn.Module, wq_params: dict = {}, aq_params: dict = {}) -> None:
        for name, child in module.named_children():
            cls = self.B.get(child.__class__.__name__)
            if cls is None:
                self.quant_block(child, wq_params, aq_params)
            elif cls in (QuantBasicTransformerBlock, QuantAttnBlock):
                setattr(module, name, cls(child, aq_params, softmax_a_bit=self.softmax_a_bit))
            elif cls in (QuantResnetBlock, QuantAttentionBlock, QuantResBlock):
                setattr(module, name, cls(child, aq_params))
            elif cls is QuantSMVMatMul:
                setattr(module, name, cls(aq_params, softmax_a_bit=self.softmax_a_bit))
            elif cls is QuantQKMatMul:
                setattr(module, name, cls(aq_params))

    # ------------------------------------------------------------------ state toggles
    def quant_layers(self) -> List[QuantLayer]:
        return [m for m in self.model.modules() if isinstance(m, QuantLayer)]

    def named_quant_layers(self):
        return [(n, m) for n, m in self.model.named_modules() if isinstance(m, QuantLayer)]

    def set_quant_state(self, use_wq: bool = False, use_aq: bool = False) -> None:
        for m in self.model.modules():
            if isinstance(m, (BaseQuantBlock, QuantLayer)):
                m.set_quant_state(use_wq=use_wq, use_aq=use_aq)

    def disable_out_quantization(self) -> None:
        """First / last layers (reference :103-120): [0], [2], [-1] stay FP and leave reconstruction;
        [1], [3] keep 4-bit weights but FP activations."""
        m = self.quant_layers()
        for i in (0, 2, -1):
            m[i].use_wq = False
            m[i].disable_aq = True
            m[i].ignore_recon = True
        m[1].disable_aq = True
        m[3].disable_aq = True

    def set_grad_ckpt(self, grad_ckpt: bool) -> None:
        for module in self.model.modules():
            if hasattr(module, "checkpoint") and module.__class__.__name__ in ("QuantBasicTransformerBlock", "BasicTransformerBlock"):
                module.checkpoint = grad_ckpt

    def synchorize_activation_statistics(self):
        """all-average of every initialised activation delta (reference :127-132; zero-points are not
        synchronised there either)."""
        from tfmq_dm_amd.linklink import dist_helper as dist
        for module in self.modules():
            if isinstance(module, QuantLayer) and module.aqtizer.delta is not None:
                dist.allaverage(module.aqtizer.delta)

    def set_running_stat(self, running_stat: bool = False) -> None:
        for m in self.model.modules():
            if isinstance(m, QuantAttnBlock):
                for q in (m.aqtizer_q, m.aqtizer_k, m.aqtizer_v, m.aqtizer_w):
                    q.running_stat = running_stat
            elif isinstance(m, QuantBasicTransformerBlock):
                for attn in (m.attn1, m.attn2):
                    for q in (attn.aqtizer_q, attn.aqtizer_k, attn.aqtizer_v, attn.aqtizer_w):
                        q.running_stat = running_stat
            elif isinstance(m, QuantLayer):
                m.set_running_stat(running_stat)

    # ------------------------------------------------------------------ lowering
    def _state_key(self):
        key = []
        for n, l in self.named_quant_layers():
            key.append((n, l.use_wq, l.use_aq and not l.disable_aq, id(l.wqtizer), getattr(l.wqtizer, "_version_", 0),
                        l.w.data_ptr(), l.w._version, bool(getattr(self, "_soft_targets", False) and getattr(l.wqtizer, "soft_tgt", False))))
        key.append(tuple(name for _, _, name, _ in self.attn_quantizers()))
        key.append(bool(getattr(self, "_exact_fp", False)))
        return tuple(key)

    def invalidate(self):
        self._plans.clear()
        self._plan = None

    def act_layer_names(self) -> List[str]:
        return [n for n, l in self.named_quant_layers() if l.use_aq and not l.disable_aq]

    def attn_quantizers(self):
        """The attention-matmul quantizers of every block whose `use_aq` was switched on BY HAND (no driver of the reference does:
        SURVEY section 0 fact 2, section 8f-3): [(engine key, role in 'qkvw', qualified quantizer name, UniformAffineQuantizer)], in
        module order.  Engine keys: the QuantAttnBlock path (DDPM), '<transformer block>.attn1' / '.attn2' (cross_attn_forward),
        '<AttentionBlock>.attention' (QuantQKMatMul + QuantSMVMatMul)."""
        out = []
        for n, m in self.model.named_modules():
            if isinstance(m, QuantAttnBlock) and m.use_aq:
                out += [(n, r, f"{n}.aqtizer_{r}", getattr(m, f"aqtizer_{r}")) for r in "qkvw"]
            elif isinstance(m, QuantBasicTransformerBlock):
                for an in ("attn1", "attn2"):
                    a = getattr(m, an)
                    if getattr(a, "use_aq", False):
                        out += [(f"{n}.{an}", r, f"{n}.{an}.aqtizer_{r}", getattr(a, f"aqtizer_{r}")) for r in "qkvw"]
            elif isinstance(m, QuantQKMatMul) and m.use_aq:
                key = n.rsplit(".", 1)[0]
                out += [(key, "q", f"{n}.aqtizer_q", m.aqtizer_q), (key, "k", f"{n}.aqtizer_k", m.aqtizer_k)]
            elif isinstance(m, QuantSMVMatMul) and m.use_aq:
                key = n.rsplit(".", 1)[0]
                out += [(key, "v", f"{n}.aqtizer_v", m.aqtizer_v), (key, "w", f"{n}.aqtizer_w", m.aqtizer_w)]
        return out

    def calibrate_attention_quantizers(self, *inputs) -> None:
        """What the reference's lazy initialisation does on the first forward after `use_aq` of an attention block is switched on
        (UniformAffineQuantizer.forward, quant_layer.py:211-221): every attention quantizer is initialised on the tensor it sees in
        this forward, upstream ones already active, with its scaler (MSE / MINMAX; the softmax quantizer `always_zero`)."""
        aqs = self.attn_quantizers()
        if not aqs:
            return
        dev = next(self.model.parameters()).device
        eng = self.engine(dev)
        _, act_names, _ = self._plan
        ids = set(range(len(act_names), len(act_names) + len(aqs)))
        scaler = getattr(aqs[0][3].scaler, "__name__", "mse")
        eng.calib_mask = ids
        try:
            eng.set_calibration("init" if scaler == "mse" else "init_minmax", 0 if self._act_step is None else int(self._act_step.item()))
            self(*inputs)
        finally:
            eng.set_calibration(None)
            eng.calib_mask = None
        k = 0 if self._act_step is None else int(self._act_step.item())
        rows = eng.qtable[k].cpu()
        for j, (_, _, _, q) in enumerate(aqs):
            d, z = rows[len(act_names) + j]
            q.delta = nn.Parameter(d.clone().to(dev)) if q.leaf_param else d.clone().to(dev)
            q.zero_point = z.clone().to(dev)
            q.init = True

    def _lower(self, device):
        from tfmq_dm_amd.engine import DdimUNetEngine, LayerQ, LdmUNetEngine
        if self.model.__class__.__name__ == "Model" and hasattr(self.model, "temb"):
            engine_cls = DdimUNetEngine          # DDPM pixel-space UNet (ddim/models/diffusion.py)
        elif self.model.__class__.__name__ == "UNetModel" and hasattr(self.model, "time_embed"):
            engine_cls = LdmUNetEngine           # SpatialTransformer UNet (openaimodel.py: SD v1 family)
        else:
            raise TfmqError(f"QuantModel: no engine plan for {self.model.__class__.__name__}")
        sd, wq = {}, {}
        act_names = self.act_layer_names()
        qid = {n: i for i, n in enumerate(act_names)}
        for n, mod in self.model.named_modules():
            if isinstance(mod, QuantLayer):
                if mod.use_wq and getattr(self, "_soft_targets", False) and getattr(mod.wqtizer, "soft_tgt", False):
                    # GetLayerGrad runs the unit under reconstruction with its AdaRound quantizers in the SOFT state (reference
                    # adaptive_rounding.py:54-58: floor(w / delta) + h(alpha), not rounded): an un-quantised layer with those weights
                    if mod.use_aq and not mod.disable_aq:
                        raise TfmqError("QuantModel: soft-target weights together with a live activation quantizer have no engine plan")
                    q = mod.wqtizer
                    sd[n + ".weight"] = ops.adaround_soft_fwd(mod.w.detach().float().contiguous(), q.alpha.detach().float().contiguous(),
                                                              q.delta.detach(), q.zero_point.detach(), q.level, hard=False)
                    if mod.b is not None:
                        sd[n + ".bias"] = mod.b.detach()
                elif mod.use_wq:
                    d, z, a = mod.weight_quant_state()
                    sd[n + ".weight"] = mod.w.detach()
                    if mod.b is not None:
                        sd[n + ".bias"] = mod.b.detach()
                    wq[n] = LayerQ(d, z, a, qid.get(n), level=mod.wqtizer.level, act_level=mod.aqtizer.level)
                else:
                    sd[n + ".weight"] = mod.original_w
                    if mod.original_b is not None:
                        sd[n + ".bias"] = mod.original_b
            elif isinstance(mod, (nn.Conv2d, nn.Conv1d, nn.Linear, nn.GroupNorm, nn.LayerNorm)):
                for pn, p in mod.named_parameters(recurse=False):
                    sd[f"{n}.{pn}"] = p.detach()
        eng = engine_cls(sd, self.model.engine_cfg(), device)
        if getattr(self, "_exact_fp", False):     # save_grad / GetLayerGrad: the differentiated forward runs in the exact-fp32 mode
            eng.exact_fp, eng.stream_f16 = True, False
        if not hasattr(self, "_tiles"):
            self._tiles = {}
        eng.tiles = self._tiles      # measured tile shapes survive re-lowering (keys are shapes, not weights)
        n_steps = 1 if self._act_table is None else self._act_table.shape[0]
        # (an attention block's matmul quantizers follow the block's OWN hand-set `use_aq`, not the model's quant state: the reference's
        # set_quant_state only walks QuantLayers, quant_block.py:20-26 -- so they stay live in the FP passes of save_inout too)
        aqs = self.attn_quantizers()
        attn_q = {}
        for j, (key, role, _, q) in enumerate(aqs):
            attn_q.setdefault(key, {})[role] = len(act_names) + j
            if role == "w":
                attn_q[key]["w_level"] = int(q.level)
        if aqs and self._act_table is not None and self._act_table.shape[1] != len(act_names) + len(aqs):
            raise TfmqError("QuantModel: an activation table from a checkpoint holds no attention-matmul quantizers; switch `use_aq` of the "
                            "attention blocks on only with module-held quantizer state (no table installed)")
        qtable = torch.zeros(n_steps, max(len(act_names) + len(aqs), 1), 2, dtype=torch.float32, device=device)
        if self._act_step is None:
            self._act_step = torch.zeros(1, dtype=torch.int32, device=device)
        eng.prepare(wq, qtable if (act_names or aqs) else None, self._act_step, attn_q=attn_q or None)   # the step counter also indexes the per-step TIB table
        self._plan = (eng, act_names, qtable)
        self._sync_act_params()
        return eng

    def _sync_act_params(self):
        """Module quantizer state (aqtizer.delta / zero_point) -> row(s) of the device table."""
        eng, act_names, qtable = self._plan
        if not act_names and qtable.shape[1] <= 1 and not self.attn_quantizers():
            return
        if self._act_table is not None:
            qtable.copy_(self._act_table)
            return
        layers = dict(self.named_quant_layers())
        rows = []
        for q in [layers[n].aqtizer for n in act_names] + [a[3] for a in self.attn_quantizers()][:qtable.shape[1] - len(act_names)]:
            if q.delta is None:
                rows.append([0.0, 0.0])      # not initialised yet: calibration mode fills it
            else:
                zp = q.zero_point
                rows.append([float(q.delta), float(zp)])
        qtable[0].copy_(torch.tensor(rows, dtype=torch.float32))

    def engine(self, device=None):
        device = device or next(self.model.parameters()).device
        key = (self._state_key(), str(torch.device(device)))
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) >= 4:
                self._plans.clear()
            self._lower(device)
            self._plans[key] = self._plan
        else:
            self._plan = plan
        return self._plan[0]

    def set_act_table(self, cali_ckpt: Optional[dict], rows=None):
        """Install the whole Finite-Set-Calibration table {act_0..act_{G-1}} on the device, replacing
        the per-step `load_state_dict(act_k)` of the reference's sampling loop.  rows: optional list of group indices --
        the installed table is then [act_{rows[0]}, act_{rows[1]}, ...], one row per sampling step in execution order, which is
        what a captured step graph indexes with the device step counter."""
        if cali_ckpt is None:
            self._act_table = None
        else:
            names = self.act_layer_names()
            G = len([k for k in cali_ckpt if k.startswith("act_")])
            tab = torch.zeros(G, len(names), 2)
            for g in range(G):
                act = cali_ckpt[f"act_{g}"]
                for i, n in enumerate(names):
                    tab[g, i, 0] = float(act[f"model.{n}.aqtizer.delta"])
                    tab[g, i, 1] = float(act[f"model.{n}.aqtizer.zero_point"])
            if rows is not None:
                tab = tab[torch.as_tensor(list(rows), dtype=torch.long)].contiguous()
            self._act_table = tab.to(next(self.model.parameters()).device)
        self.invalidate()

    def select_act_group(self, k: int):
        """Finite-Set Calibration: use row k of the table installed by set_act_table() for the following forwards
        (what `load_state_dict(ckpt['act_k'])` does in ddpm.py:1403-1405 / denoising.py:26-29, without the tree walk)."""
        if self._act_table is None:
            raise TfmqError("select_act_group: no activation table installed (set_act_table)")
        if not 0 <= int(k) < self._act_table.shape[0]:
            raise TfmqError(f"select_act_group: group {k} outside the table (0..{self._act_table.shape[0] - 1})")
        if self._act_step is None:
            self._act_step = torch.zeros(1, dtype=torch.int32, device=self._act_table.device)
        self._act_step.fill_(int(k))

    def load_state_dict(self, state_dict, strict: bool = True):
        """`model.load_state_dict(act_k, strict=False)` (ddim/functions/denoising.py:26-29) keeps
        working: the module state is updated as in torch and the device table row is refreshed."""
        out = super().load_state_dict(state_dict, strict=strict)
        if self._plan is not None and self._act_table is None and any("aqtizer" in k for k in state_dict):
            self._sync_act_params()
        if any((".w" in k or "alpha" in k or "wqtizer" in k) for k in state_dict):
            self.invalidate()
        return out

    def forward(self, x: torch.Tensor, timestep=None, context: torch.Tensor = None) -> torch.Tensor:
        if not x.is_cuda:
            raise TfmqError("QuantModel.forward: CPU tensor (the HIP kernels are the only implementation)")
        eng = self.engine(x.device)
        t = timestep if torch.is_tensor(timestep) else torch.full((x.shape[0],), float(timestep), device=x.device)
        xin = ops.nchw_to_nhwc(x.float().contiguous())
        if context is None:
            eps = eng.forward(xin, t.float().contiguous().to(x.device))
        else:
            eps = eng.forward(xin, t.float().contiguous().to(x.device), context.float().contiguous().to(x.device))
        return ops.nhwc_to_nchw(eps)
