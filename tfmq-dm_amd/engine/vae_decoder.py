"""First-stage decoder on the HIP kernels (reference ldm/modules/diffusionmodules/model.py:462-570
Decoder, ldm/models/autoencoder.py:329-332 AutoencoderKL.decode, ldm/models/diffusion/ddpm.py:706-708
decode_first_stage).  Once sampling is fast, decoding 64x64 latents to 512x512 images is the end-to-end bottleneck;
it is outside the timed region of the images/sec metric (sample_diffusion_ldm.py:127-150) and is reported separately.

The Decoder is built from the DDPM UNet's own blocks (ResnetBlock without timestep embedding, single-head AttnBlock,
nearest-2x Upsample + 3x3 conv, GroupNorm(32, eps 1e-6) + swish), so this engine is DdimUNetEngine's kernels in a
different order: every conv is un-quantised (f16 MFMA, fp32 accumulation, LDS-DMA path on fp16 activations written by
the GroupNorm), the 512-channel mid attention over 4096 tokens runs on the exact-fp32 wide-head path."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops
from .._lib import TfmqError
from .ddim_unet import DdimUNetEngine


class VaeDecoderEngine(DdimUNetEngine):
    """sd: the first-stage state dict ('decoder.*' and optionally 'post_quant_conv.*' keys); cfg: the Decoder's
    ddconfig (ch_mult, num_res_blocks, resolution, attn_resolutions)."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: dict, device="cuda:0"):
        dec = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
        if "conv_in.weight" not in dec:
            raise TfmqError("VaeDecoderEngine: state dict has no 'decoder.conv_in.weight'")
        for k, v in sd.items():
            if k.startswith("post_quant_conv."):
                dec[k] = v
        c = dict(cfg)
        c.setdefault("attn_resolutions", [])
        super().__init__(dec, c, device)
        self.res_names = []
        self.prepare()       # every layer FP

    def _forward(self, z: torch.Tensor, scale_factor: float = 1.0, pre_end: bool = False, **_) -> torch.Tensor:
        """z: fp32 NHWC latents [B,h,w,zc] -> fp32 NHWC image [B, h*2^(levels-1), w*2^(levels-1), out_ch]."""
        cfg, L = self.cfg, self.layers
        nlev, nres = len(cfg["ch_mult"]), cfg["num_res_blocks"]
        h = z
        if scale_factor != 1.0:       # z = 1. / self.scale_factor * z  (ddpm.py:707): one multiply by fl32(1/s)
            hz = ops._alloc(z.shape, dtype=torch.float32, device=z.device)
            hz.zero_()
            h = ops.axpy(hz, z.contiguous(), 1.0 / float(scale_factor))
        if "post_quant_conv" in L:
            h = L["post_quant_conv"].run(h, want_stats=False)
        h = L["conv_in"].run(h, pad=(1, 1, 1, 1))
        h = self._resblock("mid.block_1", h, None, {})
        h = self._attnblock("mid.attn_1", h)
        h = self._resblock("mid.block_2", h, None, {})
        res = cfg["resolution"] // 2 ** (nlev - 1)
        for i in reversed(range(nlev)):
            for j in range(nres + 1):
                h = self._resblock(f"up.{i}.block.{j}", h, None, {})
                if res in cfg["attn_resolutions"]:
                    h = self._attnblock(f"up.{i}.attn.{j}", h)
            if i != 0:
                up = L[f"up.{i}.upsample.conv"]
                h = up.run(ops.to_half(h) if self._fp_conv_half_ok(up) else h, pad=(1, 1, 1, 1), up2x=True)
                res *= 2
        if pre_end:
            return h
        h, _ = self._gn("norm_out", h, None, True, None, half=self._fp_conv_half_ok(L["conv_out"]))
        return L["conv_out"].run(h, pad=(1, 1, 1, 1), want_stats=False)
