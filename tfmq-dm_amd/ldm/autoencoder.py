"""Decode side of the first-stage models (reference ldm/models/autoencoder.py: AutoencoderKL.decode :329-332,
VQModelInterface.decode :274-282 with force_not_quantize=True) on the HIP engine -- the drop-in for
`LatentDiffusion.first_stage_model` in `decode_first_stage` (ldm/models/diffusion/ddpm.py:694-708).  Encoders, losses,
Lightning plumbing and the VQ codebook snap are glue outside this package."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from .._lib import TfmqError
from ..engine.vae_decoder import VaeDecoderEngine


class FirstStageDecoder(nn.Module):
    """`decode(z)`: z NCHW fp32 latents on the device -> NCHW fp32 images, as `first_stage_model.decode(z)` returns.
    state_dict: the checkpoint's 'first_stage_model.' sub-dict (keys 'decoder.*', 'post_quant_conv.*')."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], ddconfig: dict, device="cuda:0"):
        super().__init__()
        self.ddconfig = dict(ddconfig)
        self.engine = VaeDecoderEngine(state_dict, self.ddconfig, device)

    def decode(self, z: torch.Tensor, force_not_quantize: bool = True) -> torch.Tensor:
        if not force_not_quantize:
            raise TfmqError("FirstStageDecoder.decode: the VQ codebook snap is not part of this package")
        y = self.engine.forward(z.permute(0, 2, 3, 1).contiguous().float())
        return y.permute(0, 3, 1, 2)

    def forward(self, z):
        return self.decode(z)
