"""Stand-alone execution of one quantised block (a reconstruction unit) on the HIP kernels: what
`block(*inputs)` means for QuantResnetBlock / QuantAttnBlock / the DDIM TIB outside a whole-model
plan (quant/data_utill.py hooks, block_reconstruction's forward).  NCHW in / NCHW out."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .ddim_unet import DdimUNetEngine, LayerQ


def _collect(block: nn.Module, prefix: str = "blk"):
    """Module tree of one block -> (sd, wq) in the engine's naming, honouring each QuantLayer's
    current use_wq / use_aq / disable_aq flags; per-layer activation parameters go to a 1-row table."""
    # duck-typed: the mirror may be imported as `quant.*` (drop-in) or `tfmq_dm_amd.quant.*`
    def is_quant_layer(m):
        return hasattr(m, "wqtizer") and hasattr(m, "original_w") and hasattr(m, "weight_quant_state")
    sd, wq, rows = {}, {}, []
    for n, mod in block.named_modules():
        full = f"{prefix}.{n}" if n else prefix
        if is_quant_layer(mod):
            if mod.use_wq:
                d, z, a = mod.weight_quant_state()
                sd[full + ".weight"] = mod.w.detach()
                if mod.b is not None:
                    sd[full + ".bias"] = mod.b.detach()
                qid = None
                if mod.use_aq and not mod.disable_aq and mod.aqtizer.delta is not None:
                    qid = len(rows)
                    rows.append([float(mod.aqtizer.delta), float(mod.aqtizer.zero_point)])
                wq[full] = LayerQ(d, z, a, qid, level=mod.wqtizer.level, act_level=mod.aqtizer.level)
            else:
                sd[full + ".weight"] = mod.original_w
                if mod.original_b is not None:
                    sd[full + ".bias"] = mod.original_b
        elif isinstance(mod, (nn.Conv2d, nn.Conv1d, nn.Linear, nn.GroupNorm, nn.LayerNorm)):
            for pn, p in mod.named_parameters(recurse=False):
                sd[f"{full}.{pn}"] = p.detach()
    return sd, wq, rows


def _engine_for(block: nn.Module, device):
    sd, wq, rows = _collect(block)
    eng = DdimUNetEngine(sd, {}, device)
    qtable = torch.tensor([rows], dtype=torch.float32, device=device) if rows else None
    eng.prepare(wq, qtable, None)
    return eng


def run_resnet_block(block, x: torch.Tensor, temb: torch.Tensor) -> torch.Tensor:
    """QuantResnetBlock.forward (reference quant/quant_block.py:415-444)."""
    eng = _engine_for(block, x.device)
    proj = eng._linear("blk.temb_proj", temb.float().contiguous(), True)
    y = eng._resblock("blk", ops.nchw_to_nhwc(x.float().contiguous()), None, dict(rowadd=proj))
    return ops.nhwc_to_nchw(y)


def run_attn_block(block, x: torch.Tensor) -> torch.Tensor:
    """QuantAttnBlock.forward (reference quant/quant_block.py:474-505) with un-quantised QK^T / PV."""
    eng = _engine_for(block, x.device)
    return ops.nhwc_to_nchw(eng._attnblock("blk", ops.nchw_to_nhwc(x.float().contiguous())))


def _ldm_engine_for(block: nn.Module, device, cfg):
    from .ldm_unet import LdmUNetEngine
    sd, wq, rows = _collect(block)
    eng = LdmUNetEngine(sd, cfg, device)
    qtable = torch.tensor([rows], dtype=torch.float32, device=device) if rows else None
    eng.prepare(wq, qtable, None)
    return eng


def run_res_block(block, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """QuantResBlock.forward (reference quant/quant_block.py:153-206): x NCHW, emb [B, emb_channels]."""
    eng = _ldm_engine_for(block, x.device, {})
    proj = eng._linear("blk.emb_layers.1", emb.float().contiguous(), True)
    y = eng._res("blk", ops.nchw_to_nhwc(x.float().contiguous()), None, dict(rowadd=proj))
    return ops.nhwc_to_nchw(y)


def run_transformer_block(block, x: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
    """QuantBasicTransformerBlock.forward (reference quant/quant_block.py:286-299): x [B,T,C] tokens, context [B,L,D]."""
    eng = _ldm_engine_for(block, x.device, dict(num_heads=block.attn1.heads))
    return eng._tblock("blk", x.float().contiguous(), context.float().contiguous())


def run_attention_block(block, x: torch.Tensor) -> torch.Tensor:
    """QuantAttentionBlock.forward (reference quant/quant_block.py:373-387): x NCHW."""
    eng = _ldm_engine_for(block, x.device, dict(num_heads=block.num_heads))
    return ops.nhwc_to_nchw(eng._attn_block("blk", ops.nchw_to_nhwc(x.float().contiguous())))
