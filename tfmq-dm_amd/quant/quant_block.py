"""Quantised blocks = reconstruction units (reference quant/quant_block.py).

A block re-hosts the sub-modules of the FP block it replaces (same attribute names, so
state-dict keys are unchanged) and defines the unit that block reconstruction optimises.
Execution is not done module by module: QuantModel lowers the whole tree to an engine plan; a
block's own `forward` (used when a unit is evaluated in isolation, e.g. by save_inout or
block_reconstruction) runs the corresponding fused engine routine.

DDPM-UNet blocks (BASELINE configs 1-2) and the SpatialTransformer-UNet blocks of Stable Diffusion
(QuantResBlock, QuantBasicTransformerBlock, QuantTemporalInformationBlock) and the AttentionBlock seams of the
unconditional LDMs (QuantAttentionBlock, QuantQKMatMul, QuantSMVMatMul) are implemented.
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


class QuantTemporalInformationBlock(BaseQuantBlock):
    """TIB of the latent-diffusion UNets (reference :76-128): `time_embed` MLP + the `emb_layers` of every ResBlock,
    optimised as one unit (TIAR).  The emb_layers stay owned by their ResBlocks (plain list, as in the reference)."""

    def __init__(self, t_emb: nn.Sequential, aq_params: dict = {}, model_channels: int = None, num_classes: int = None) -> None:
        super().__init__(aq_params)
        self.t_emb = t_emb
        self.emb_layers = []
        self.label_emb_layer = None
        self.model_channels = model_channels
        self.num_classes = num_classes
        if num_classes is not None:
            raise TfmqError("QuantTemporalInformationBlock: class-conditional label embedding (cin256) is a next row")

    def add_emb_layer(self, layer: nn.Sequential) -> None:
        self.emb_layers.append(layer)

    def add_label_emb_layer(self, layer: nn.Sequential) -> None:
        self.label_emb = layer

    def quant_layers(self):
        own = [m for m in self.modules() if isinstance(m, QuantLayer)]
        return own + [m for seq in self.emb_layers for m in seq.modules() if isinstance(m, QuantLayer)]

    def forward(self, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor = None) -> Tuple[torch.Tensor]:
        from tfmq_dm_amd.engine.tib import tib_forward_ldm
        return tib_forward_ldm(self, t)


class QuantResBlock(BaseQuantBlock):
    """reference :131-206 (the SD / LDM ResBlock: GN32 -> SiLU -> conv, + emb_layers(emb), GN32 -> SiLU -> conv,
    + skip_connection).  up/down and scale-shift variants are not enabled by any BASELINE config."""

    def __init__(self, res: nn.Module, aq_params: dict = {}) -> None:
        super().__init__(aq_params)
        self.channels, self.emb_channels, self.dropout = res.channels, res.emb_channels, res.dropout
        self.out_channels, self.use_conv = res.out_channels, res.use_conv
        self.use_checkpoint, self.use_scale_shift_norm = res.use_checkpoint, res.use_scale_shift_norm
        self.in_layers = res.in_layers
        self.updown = res.updown
        self.h_upd, self.x_upd = res.h_upd, res.x_upd
        self.emb_layers = res.emb_layers
        self.out_layers = res.out_layers
        self.skip_connection = res.skip_connection
        if self.updown or self.use_scale_shift_norm:
            raise TfmqError("QuantResBlock: resblock up/down and scale-shift norm are not used by the BASELINE configs")

    def forward(self, x, emb=None, split: int = 0):
        if emb is None:
            x, emb = x
        if split != 0:
            raise TfmqError("QuantResBlock: split (QDIFF dual quantizers) is never enabled by the drivers (SURVEY §8f)")
        from tfmq_dm_amd.engine.blocks import run_res_block
        return run_res_block(self, x, emb)


class QuantBasicTransformerBlock(BaseQuantBlock):
    """reference :254-299.  attn1 (self), attn2 (cross, context), GEGLU feed-forward, pre-LayerNorm residuals.  The
    q/k/v/softmax quantizers are created for state compatibility; `attn.use_aq` is never set by any driver."""

    def __init__(self, tran: nn.Module, aq_params: dict = {}, softmax_a_bit: int = 8) -> None:
        super().__init__(aq_params)
        self.attn1, self.ff, self.attn2 = tran.attn1, tran.ff, tran.attn2
        self.norm1, self.norm2, self.norm3 = tran.norm1, tran.norm2, tran.norm3
        self.checkpoint = False
        aq_w = dict(aq_params)
        aq_w.update(bits=softmax_a_bit, symmetric=False, always_zero=True)
        for attn in (self.attn1, self.attn2):
            attn.aqtizer_q = UniformAffineQuantizer(**aq_params)
            attn.aqtizer_k = UniformAffineQuantizer(**aq_params)
            attn.aqtizer_v = UniformAffineQuantizer(**aq_params)
        self.attn1.aqtizer_w = UniformAffineQuantizer(**aq_w)
        self.attn2.aqtizer_w = UniformAffineQuantizer(**aq_w)
        self.attn1.use_aq = False
        self.attn2.use_aq = False

    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        if context is None:
            raise TfmqError("QuantBasicTransformerBlock: context is required (reference asserts the same)")
        if self.attn1.use_aq or self.attn2.use_aq:
            raise NotImplementedError("int8 attention matmuls are a 'next' row (SURVEY §8f-3); no driver enables them")
        from tfmq_dm_amd.engine.blocks import run_transformer_block
        return run_transformer_block(self, x, context)


class QuantQKMatMul(BaseQuantBlock):
    """reference :303-328: the q.k^T seam of QKVAttentionLegacy.  Carries the (never enabled) q/k quantizers; holds no
    QuantLayer, so block reconstruction returns immediately.  Executed inside the engine's attention kernel."""

    def __init__(self, aq_params: dict = {}) -> None:
        super().__init__(aq_params)
        self.scale = None
        self.use_aq = False
        self.aqtizer_q = UniformAffineQuantizer(**aq_params)
        self.aqtizer_k = UniformAffineQuantizer(**aq_params)

    def forward(self, q, k):
        raise TfmqError("QuantQKMatMul: the attention matmuls run inside the engine's attention kernel, not module by module")


class QuantSMVMatMul(BaseQuantBlock):
    """reference :331-354: the softmax.v seam (8-bit `always_zero` softmax quantizer, never enabled)."""

    def __init__(self, aq_params: dict = {}, softmax_a_bit: int = 8) -> None:
        super().__init__(aq_params)
        self.use_aq = False
        self.aqtizer_v = UniformAffineQuantizer(**aq_params)
        aq_w = dict(aq_params)
        aq_w.update(bits=softmax_a_bit, symmetric=False, always_zero=True)
        self.aqtizer_w = UniformAffineQuantizer(**aq_w)

    def forward(self, weight, v):
        raise TfmqError("QuantSMVMatMul: the attention matmuls run inside the engine's attention kernel, not module by module")


class QuantAttentionBlock(BaseQuantBlock):
    """reference :357-387 (weight-only runs, `leaf_param` False): re-hosts the AttentionBlock.  Its Conv1d projections
    are not QuantLayers, so the block has nothing to reconstruct."""

    def __init__(self, attn: nn.Module, aq_params: dict = {}) -> None:
        super().__init__(aq_params)
        self.channels, self.num_heads, self.use_checkpoint = attn.channels, attn.num_heads, attn.use_checkpoint
        self.norm, self.qkv, self.attention, self.proj_out = attn.norm, attn.qkv, attn.attention, attn.proj_out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from tfmq_dm_amd.engine.blocks import run_attention_block
        return run_attention_block(self, x)


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
