"""Parameter containers for the latent-diffusion UNet of the reference
(ldm/modules/diffusionmodules/openaimodel.py UNetModel with SpatialTransformer, ldm/modules/attention.py).
Same module tree and parameter names as the reference, so checkpoints (`model.diffusion_model.*`) load and
the quant/ tree rewrite applies unchanged.  No torch compute: `forward` lowers to engine.LdmUNetEngine."""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from .._lib import TfmqError


class GroupNorm32(nn.GroupNorm):
    """normalization(): GroupNorm(32, C), eps 1e-5 (ldm/modules/diffusionmodules/util.py:214-216)."""


class TimestepEmbedSequential(nn.Sequential):
    pass


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if use_conv:
            self.conv = nn.Conv2d(channels, self.out_channels, 3, padding=padding)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if not use_conv:
            raise TfmqError("avg-pool Downsample is not used by any BASELINE config")
        self.op = nn.Conv2d(channels, self.out_channels, 3, stride=2, padding=padding)


class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or up or down or use_conv:
            raise TfmqError("ResBlock variants (scale-shift norm, resblock up/down, 3x3 skip) are not enabled by any BASELINE config")
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_conv, self.use_checkpoint, self.use_scale_shift_norm, self.updown = use_conv, use_checkpoint, False, False
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(), nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(32, self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1))
        self.skip_connection = nn.Identity() if self.out_channels == channels else nn.Conv2d(channels, self.out_channels, 1)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=True, dropout=0.0):
        super().__init__()
        if not glu:
            raise TfmqError("non-gated FeedForward is not used by the SD / cin256 configs")
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim))


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = context_dim or query_dim
        self.scale, self.heads = dim_head ** -0.5, heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.checkpoint = checkpoint


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None):
        super().__init__()
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim)
                                                 for _ in range(depth)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)


class QKMatMul(nn.Module):
    """Module seam of the reference (openaimodel.py:349-360): swapped for QuantQKMatMul by the tree rewrite."""

    def __init__(self):
        super().__init__()
        self.scale = None


class SMVMatMul(nn.Module):
    pass


class QKVAttentionLegacy(nn.Module):
    """heads are split BEFORE q/k/v: channel (h, {q,k,v}, c) of the qkv projection (openaimodel.py:372-405)."""

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads
        self.qkv_matmul = QKMatMul()
        self.smv_matmul = SMVMatMul()


class AttentionBlock(nn.Module):
    """Self-attention over the spatial positions (openaimodel.py:280-326): GroupNorm32 -> qkv Conv1d -> multi-head
    attention -> proj_out Conv1d, + x.  Conv1d is not in QuantLayer.QMAP, so the block stays un-quantised."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, use_new_attention_order=False):
        super().__init__()
        if use_new_attention_order:
            raise TfmqError("AttentionBlock: use_new_attention_order is not set by any BASELINE config")
        self.channels = channels
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        if num_head_channels != -1 and channels % num_head_channels:
            raise TfmqError(f"q,k,v channels {channels} is not divisible by num_head_channels {num_head_channels}")
        self.use_checkpoint = use_checkpoint
        self.norm = GroupNorm32(32, channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.attention = QKVAttentionLegacy(self.num_heads)
        self.proj_out = nn.Conv1d(channels, channels, 1)


class UNetModel(nn.Module):
    """SpatialTransformer UNet (openaimodel.py:408-780), constructor arguments as in the YAML configs."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 num_heads=-1, num_head_channels=-1, use_spatial_transformer=False, transformer_depth=1, context_dim=None,
                 legacy=True, **unused):
        super().__init__()
        if use_spatial_transformer and context_dim is None:
            raise TfmqError("UNetModel: the SpatialTransformer UNet needs context_dim")
        if num_classes is not None:
            raise TfmqError("class-conditional label embedding is not used by the SD config")
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = num_res_blocks, list(attention_resolutions), list(channel_mult)
        self.num_heads, self.num_head_channels, self.context_dim = num_heads, num_head_channels, context_dim
        self.use_spatial_transformer = use_spatial_transformer
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))

        def heads_for(ch):
            if num_head_channels == -1:
                return num_heads, ch // num_heads
            return ch // num_head_channels, num_head_channels

        def st(ch):
            nh, dh = heads_for(ch)
            if not use_spatial_transformer:      # unconditional LDMs (CelebA-HQ, LSUN): plain AttentionBlock
                return AttentionBlock(ch, use_checkpoint=use_checkpoint, num_heads=nh,
                                      num_head_channels=num_head_channels if legacy else dh)
            if legacy:
                dh = ch // nh
            return SpatialTransformer(ch, nh, dh, depth=transformer_depth, context_dim=context_dim)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, dropout, out_channels=mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(st(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, dropout), st(ch), ResBlock(ch, ted, dropout))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(st(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))
        self._engine = None

    def engine_cfg(self) -> dict:
        return dict(model_channels=self.model_channels, num_heads=self.num_heads, in_channels=self.in_channels,
                    out_channels=self.out_channels, context_dim=self.context_dim, num_head_channels=self.num_head_channels)

    def forward(self, x, timesteps=None, context=None, y=None, **kw):
        from .. import ops
        from ..engine import LdmUNetEngine
        if not x.is_cuda:
            raise TfmqError("UNetModel.forward: the latent UNet only runs on the HIP kernels (no CPU fallback)")
        if self._engine is None:
            self._engine = LdmUNetEngine(self.state_dict(), self.engine_cfg(), x.device)
            self._engine.prepare()
        eps = self._engine.forward(ops.nchw_to_nhwc(x.float().contiguous()), timesteps.float().contiguous(),
                                   None if context is None else context.float().contiguous())
        return ops.nhwc_to_nchw(eps)


SD_V1_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                  num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                  transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False)
"""unet_config.params of stable-diffusion/configs/stable-diffusion/v1-inference.yaml"""

CIN256_V2_UNET = dict(image_size=64, in_channels=3, out_channels=3, model_channels=192, attention_resolutions=[8, 4, 2],
                      num_res_blocks=2, channel_mult=[1, 2, 3, 5], num_heads=1, use_spatial_transformer=True,
                      transformer_depth=1, context_dim=512)
"""unet_config.params of stable-diffusion/configs/latent-diffusion/cin256-v2.yaml (latent_imagenet_diffusion.py:35: class-conditional
ImageNet 256, 64x64x3 latents, ONE attention head per level, cross attention over ONE class-embedding token)"""

CELEBAHQ_LDM_VQ4_UNET = dict(image_size=64, in_channels=3, out_channels=3, model_channels=224, attention_resolutions=[8, 4, 2],
                             num_res_blocks=2, channel_mult=[1, 2, 3, 4], num_head_channels=32)
"""unet_config.params of stable-diffusion/configs/latent-diffusion/celebahq-ldm-vq-4.yaml (sample_diffusion_ldm.py -r
models/ldm/celeba256: unconditional LDM-4, 64x64x3 latents, plain AttentionBlocks with 32-channel heads)"""
