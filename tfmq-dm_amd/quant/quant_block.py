"""Quantised blocks = reconstruction units (reference quant/quant_block.py).

A block re-hosts the sub-modules of the FP block it replaces (same attribute names, so
state-dict keys are unchanged) and defines the unit that block reconstruction optimises.
Execution is not done module by module: QuantModel lowers the whole tree to an engine plan; a
block's own `forward` (used when a unit is evaluated in isolation, e.g. by save_inout or
block_reconstruction) runs the corresponding fused engine routine.

DDPM-UNet blocks (BASELINE configs 1-2) are implemented; the LDM / Stable-Diffusion blocks
(QuantResBlock, QuantBasicTransformerBlock, QuantAttentionBlock, QuantQKMatMul, QuantSMVMatMul,
QuantTemporalInformationBlock) are declared and raise until their engine plan lands (SURVEY §8
rows U2/U3).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn as nn

from tfmq_dm_amd._lib import TfmqError
from .quant_layer import QuantLayer, StraightThrough, UniformAffineQuantizer


class BaseQuantBlock(nn.Module):
    def __init__(self, aq_params: dict = {}) -> None:
        super().__init__()
        self.use_wq = False
        self.use_aq = False
        self.act_func = StraightThrough()
        self.ignore_recon = False

    def quant_layers(self):
        return [m for m in self.modules() if isinstance(m, QuantLayer)]

    def set_quant_state(self, use_wq: bool = False, use_aq: bool = False) -> None:
        # only QuantLayer children are touched: the attention-matmul quantizers below are never
        # enabled by any driver (reference :27-33)
        for m in self.quant_layers():
            m.set_quant_state(use_wq=use_wq, use_aq=use_aq)


class QuantTemporalInformationBlockDDIM(BaseQuantBlock):
    """TIB of the DDPM UNet: timestep-embedding MLP + every ResnetBlock's temb projection,
    optimised as one unit (TIAR; reference :36-75)."""

    def __init__(self, temb: nn.Module, aq_params: dict = {}, ch: int = None) -> None:
        super().__init__(aq_params)
        self.temb = temb
        self.temb_projs = []   # plain list on purpose: the projections stay owned by their ResnetBlocks
        self.ch = ch

    def add_temb_proj(self, temb_proj: nn.Linear) -> None:
        self.temb_projs.append(temb_proj)

    def quant_layers(self):
        return [m for m in self.modules() if isinstance(m, QuantLayer)] + list(self.temb_projs)

    def forward(self, x: torch.Tensor, t: torch.Tensor) -> Tuple[torch.Tensor]:
        from tfmq_dm_amd.engine.tib import tib_forward_ddim
        return tib_forward_ddim(self, t)


class QuantResnetBlock(BaseQuantBlock):
    """reference :391-444"""

    def __init__(self, res: nn.Module, aq_params: dict = {}) -> None:
        super().__init__(aq_params)
        self.in_channels, self.out_channels = res.in_channels, res.out_channels
        self.use_conv_shortcut = res.use_conv_shortcut
        self.norm1, self.conv1, self.temb_proj = res.norm1, res.conv1, res.temb_proj
        self.norm2, self.dropout, self.conv2 = res.norm2, res.dropout, res.conv2
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                self.conv_shortcut = res.conv_shortcut
            else:
                self.nin_shortcut = res.nin_shortcut

    def forward(self, x, temb=None, split: int = 0):
        if temb is None:
            x, temb = x
        from tfmq_dm_amd.engine.blocks import run_resnet_block
        return run_resnet_block(self, x, temb)


class QuantAttnBlock(BaseQuantBlock):
    """reference :447-505.  The q/k/v/softmax quantizers exist for state compatibility; `use_aq`
    of the block is never set, so QK^T and PV run on un-quantised tensors."""

    def __init__(self, attn: nn.Module, aq_params: dict = {}, softmax_a_bit: int = 8) -> None:
        super().__init__(aq_params)
        self.in_channels = attn.in_channels
        self.norm, self.q, self.k, self.v, self.proj_out = attn.norm, attn.q, attn.k, attn.v, attn.proj_out
        self.aqtizer_q = UniformAffineQuantizer(**aq_params)
        self.aqtizer_k = UniformAffineQuantizer(**aq_params)
        self.aqtizer_v = UniformAffineQuantizer(**aq_params)
        aq_w = dict(aq_params)
        aq_w.update(bits=softmax_a_bit, symmetric=False, always_zero=True)
        self.aqtizer_w = UniformAffineQuantizer(**aq_w)

    def forward(self, x):
        if self.use_aq:
            raise NotImplementedError("int8 attention matmuls are a 'next' row (SURVEY §8f-3); no driver enables them")
        from tfmq_dm_amd.engine.blocks import run_attn_block
        return run_attn_block(self, x)


def _ldm_block(name):
    class _Pending(BaseQuantBlock):
        def __init__(self, *a, **k):
            raise TfmqError(f"{name}: the LDM / Stable-Diffusion engine plan is not built yet "
                            "(BASELINE configs 3-5; DESIGN.md 'what comes next')")
    _Pending.__name__ = name
    return _Pending


QuantResBlock = _ldm_block("QuantResBlock")
QuantBasicTransformerBlock = _ldm_block("QuantBasicTransformerBlock")
QuantAttentionBlock = _ldm_block("QuantAttentionBlock")
QuantQKMatMul = _ldm_block("QuantQKMatMul")
QuantSMVMatMul = _ldm_block("QuantSMVMatMul")
QuantTemporalInformationBlock = _ldm_block("QuantTemporalInformationBlock")


def b2qb(use_aq: bool = False) -> Dict[str, type]:
    """FP block class name -> quantised block (reference :508-520)."""
    D = {
        "ResBlock": QuantResBlock,
        "BasicTransformerBlock": QuantBasicTransformerBlock,
        "ResnetBlock": QuantResnetBlock,
        "AttnBlock": QuantAttnBlock,
    }
    if use_aq:
        D["QKMatMul"] = QuantQKMatMul
        D["SMVMatMul"] = QuantSMVMatMul
    else:
        D["AttentionBlock"] = QuantAttentionBlock
    return D
