"""Parameter containers for the DDPM UNet of the reference (ddim/models/diffusion.py).

These nn.Modules exist to (a) own parameters under the reference's state-dict names so that
checkpoints and the `quant/` module-tree rewrite (QuantModel) are drop-in, and (b) describe the
architecture to the HIP engine.  They carry no torch compute: `Model.forward` lowers the tree to
`engine.DdimUNetEngine` and runs the HIP kernels; on a CPU tensor it raises (no CPU fallback).
"""
from __future__ import annotations

import argparse
from typing import Optional

import torch
import torch.nn as nn

from .._lib import TfmqError


def _gn(ch: int) -> nn.GroupNorm:
    # Normalize(): 32 groups, eps 1e-6 (ddim/models/diffusion.py:32-33)
    return nn.GroupNorm(32, ch, eps=1e-6, affine=True)


class Upsample(nn.Module):
    """nearest x2 (+ 3x3 conv): ddim/models/diffusion.py:36-53."""

    def __init__(self, in_channels: int, with_conv: bool):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)


class Downsample(nn.Module):
    """pad (0,1,0,1) + 3x3 stride-2 conv: ddim/models/diffusion.py:56-74."""

    def __init__(self, in_channels: int, with_conv: bool):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)


class ResnetBlock(nn.Module):
    """ddim/models/diffusion.py:77-139."""

    def __init__(self, *, in_channels: int, out_channels: Optional[int] = None, conv_shortcut: bool = False,
                 dropout: float = 0.0, temb_channels: int = 512):
        super().__init__()
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels, self.use_conv_shortcut = in_channels, out_channels, conv_shortcut
        self.norm1 = _gn(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = _gn(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            if conv_shortcut:
                raise TfmqError("conv_shortcut=True is not used by any BASELINE config (3x3 shortcut unsupported)")
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)


class AttnBlock(nn.Module):
    """ddim/models/diffusion.py:142-194."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.in_channels = in_channels
        self.norm = _gn(in_channels)
        for n in ("q", "k", "v", "proj_out"):
            setattr(self, n, nn.Conv2d(in_channels, in_channels, 1, 1, 0))


class Model(nn.Module):
    """DDPM UNet container (ddim/models/diffusion.py:197-354); same module/parameter names."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        m = config.model
        ch, ch_mult = m.ch, tuple(m.ch_mult)
        self.ch, self.temb_ch = ch, ch * 4
        self.num_resolutions, self.num_res_blocks = len(ch_mult), m.num_res_blocks
        self.resolution, self.in_channels = config.data.image_size, m.in_channels
        if not m.resamp_with_conv:
            raise TfmqError("resamp_with_conv=False (avg-pool resampling) is not on the TFMQ hot path")
        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(ch, self.temb_ch), nn.Linear(self.temb_ch, self.temb_ch)])
        self.conv_in = nn.Conv2d(m.in_channels, ch, 3, 1, 1)
        res, in_mult = self.resolution, (1,) + ch_mult
        self.down = nn.ModuleList()
        cur = None
        for lvl in range(self.num_resolutions):
            cur, out = ch * in_mult[lvl], ch * ch_mult[lvl]
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            for _ in range(self.num_res_blocks):
                stage.block.append(ResnetBlock(in_channels=cur, out_channels=out, temb_channels=self.temb_ch, dropout=m.dropout))
                cur = out
                if res in m.attn_resolutions:
                    stage.attn.append(AttnBlock(cur))
            if lvl != self.num_resolutions - 1:
                stage.downsample = Downsample(cur, True)
                res //= 2
            self.down.append(stage)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=cur, out_channels=cur, temb_channels=self.temb_ch, dropout=m.dropout)
        self.mid.attn_1 = AttnBlock(cur)
        self.mid.block_2 = ResnetBlock(in_channels=cur, out_channels=cur, temb_channels=self.temb_ch, dropout=m.dropout)
        ups = []
        for lvl in reversed(range(self.num_resolutions)):
            out, skip = ch * ch_mult[lvl], ch * ch_mult[lvl]
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            for i in range(self.num_res_blocks + 1):
                if i == self.num_res_blocks:
                    skip = ch * in_mult[lvl]
                stage.block.append(ResnetBlock(in_channels=cur + skip, out_channels=out, temb_channels=self.temb_ch, dropout=m.dropout))
                cur = out
                if res in m.attn_resolutions:
                    stage.attn.append(AttnBlock(cur))
            if lvl != 0:
                stage.upsample = Upsample(cur, True)
                res *= 2
            ups.insert(0, stage)
        self.up = nn.ModuleList(ups)
        self.norm_out = _gn(cur)
        self.conv_out = nn.Conv2d(cur, m.out_ch, 3, 1, 1)
        self._engine = None

    def engine_cfg(self) -> dict:
        m = self.config.model
        return dict(ch=m.ch, ch_mult=list(m.ch_mult), num_res_blocks=m.num_res_blocks,
                    attn_resolutions=list(m.attn_resolutions), resolution=self.resolution,
                    in_channels=m.in_channels, out_ch=m.out_ch)

    def forward(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """FP forward on the HIP engine (NCHW in / NCHW out, like the reference)."""
        from .. import ops
        from ..engine import DdimUNetEngine
        if not x.is_cuda:
            raise TfmqError("Model.forward: the DDPM UNet only runs on the HIP kernels (no CPU fallback)")
        if self._engine is None:
            self._engine = DdimUNetEngine(self.state_dict(), self.engine_cfg(), x.device)
            self._engine.prepare()
        eps = self._engine.forward(ops.nchw_to_nhwc(x.float().contiguous()), t.float().contiguous())
        return ops.nhwc_to_nchw(eps)


def make_config(ch=128, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,), image_size=32, in_channels=3,
                out_ch=3, dropout=0.1):
    """argparse.Namespace with the fields of ddim/configs/cifar10.yml that the model reads."""
    ns = argparse.Namespace
    return ns(data=ns(image_size=image_size, channels=in_channels),
              model=ns(type="simple", in_channels=in_channels, out_ch=out_ch, ch=ch, ch_mult=list(ch_mult),
                       num_res_blocks=num_res_blocks, attn_resolutions=list(attn_resolutions), dropout=dropout,
                       resamp_with_conv=True),
              diffusion=ns(beta_schedule="linear", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000),
              split_shortcut=True)


def random_init(model: nn.Module, seed: int = 1234, std: float = 0.02) -> nn.Module:
    """Synthetic weights (no checkpoints offline): default module initialisers, then every all-zero
    parameter re-drawn from N(0, std^2) (the reference's own minmax crashes on all-zero
    channels, §0-5a)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            if p.numel() and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return model
