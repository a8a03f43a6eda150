"""What the latent samplers and the calibration-set generators need from `LatentDiffusion` / `DiffusionWrapper`
(reference ldm/models/diffusion/ddpm.py): the noise schedule buffers, `apply_model`, and the per-call selection of the
Finite-Set activation group.  First stage (VAE), text encoder, Lightning plumbing: glue, out of scope."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .._lib import TfmqError


class DiffusionWrapper(nn.Module):
    """reference :1395-1424.  Attributes `tot`, `t_max`, `ckpt` (set by the drivers, txt2img.py) turn on Finite-Set
    Calibration: every UNet call uses the activation group k = t_max - (t[0]-1)//tot.  The reference walks the module
    tree with `load_state_dict(ckpt['act_k'])` on every call; here the whole table sits on the device once
    (QuantModel.set_act_table) and the call only writes k into the device step scalar."""

    def __init__(self, diffusion_model: nn.Module, conditioning_key: Optional[str] = None):
        super().__init__()
        self.diffusion_model = diffusion_model
        self.conditioning_key = conditioning_key
        if conditioning_key not in (None, "crossattn"):
            raise TfmqError(f"DiffusionWrapper: conditioning_key={conditioning_key!r} is not used by the BASELINE configs")
        self._table_of = None

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None):
        if hasattr(self, "tot"):
            k = int(self.t_max - (float(t[0].item()) - 1) // self.tot)     # t may be fractional (DPM-Solver)
            dm = self.diffusion_model
            if hasattr(dm, "set_act_table"):
                if self._table_of is not self.ckpt:
                    dm.set_act_table(self.ckpt)
                    self._table_of = self.ckpt
                dm.select_act_group(k)
            else:
                dm.load_state_dict(self.ckpt[f"act_{k}"], strict=False)
        if self.conditioning_key is None:
            return self.diffusion_model(x, t)
        return self.diffusion_model(x, t, context=torch.cat(c_crossattn, 1))


class LatentDiffusion(nn.Module):
    """Schedule + apply_model (reference register_schedule :117-169, apply_model :891-900,986-987)."""

    def __init__(self, unet: nn.Module, conditioning_key: Optional[str] = "crossattn", timesteps: int = 1000,
                 linear_start: float = 0.00085, linear_end: float = 0.012, parameterization: str = "eps"):
        super().__init__()
        self.model = DiffusionWrapper(unet, conditioning_key)
        self.parameterization = parameterization
        betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
        alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = int(timesteps)
        self.register_buffer("betas", torch.tensor(betas, dtype=torch.float32))
        self.register_buffer("alphas_cumprod", torch.tensor(alphas_cumprod, dtype=torch.float32))
        self.register_buffer("alphas_cumprod_prev", torch.tensor(np.append(1.0, alphas_cumprod[:-1]), dtype=torch.float32))

    @property
    def device(self):
        return self.betas.device

    def decode_first_stage(self, z, predict_cids: bool = False, force_not_quantize: bool = False):
        """reference :694-708: z / scale_factor, then first_stage_model.decode.  Needs `first_stage_model` (e.g.
        ldm.autoencoder.FirstStageDecoder) and `scale_factor` attributes set by the driver, as the reference's
        instantiate_first_stage / checkpoint loading does."""
        if not hasattr(self, "first_stage_model"):
            raise TfmqError("decode_first_stage: no first_stage_model attached")
        if predict_cids:
            raise TfmqError("decode_first_stage: predict_cids is not used by the BASELINE configs")
        sf = float(getattr(self, "scale_factor", 1.0))
        fs = self.first_stage_model
        if hasattr(fs, "engine"):       # scale on the device inside the engine's first kernel chain
            y = fs.engine.forward(z.permute(0, 2, 3, 1).contiguous().float(), scale_factor=sf)
            return y.permute(0, 3, 1, 2)
        return fs.decode(1.0 / sf * z)

    def apply_model(self, x_noisy, t, cond, return_ids: bool = False):
        if self.model.conditioning_key is None:
            return self.model(x_noisy, t)
        if not isinstance(cond, (list, dict)):
            cond = [cond]
        if isinstance(cond, dict):
            return self.model(x_noisy, t, **cond)
        return self.model(x_noisy, t, c_crossattn=cond)
