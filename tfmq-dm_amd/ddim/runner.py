"""`Diffusion` runner of the pixel-space path (reference ddim/runners/diffusion.py: __init__ :72-107,
sample :203-324, sample_fid :326-364, sample_image :429-476) without the training / dataset /
checkpoint-download glue.  Image files are not written here (torchvision is glue); `sample_fid` returns
the uint8 array the reference dumps to NPZ."""
from __future__ import annotations

import logging
import math
from typing import Optional

import numpy as np
import torch

from .._lib import TfmqError
from .sampler import GraphDdimSampler, generalized_steps, linear_betas, step_sequence

logger = logging.getLogger(__name__)


def inverse_data_transform(x: torch.Tensor) -> torch.Tensor:
    """ddim/datasets/__init__.py:206-215 for rescaled data: clamp((x+1)/2, 0, 1)."""
    return torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)


class Diffusion:
    def __init__(self, args, config, device: Optional[torch.device] = None):
        self.args, self.config = args, config
        self.device = device or (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
        config.device = self.device
        d = config.diffusion
        if d.beta_schedule != "linear":
            raise NotImplementedError("only the linear beta schedule of the BASELINE configs is built")
        self.betas = linear_betas(d.beta_start, d.beta_end, d.num_diffusion_timesteps).to(self.device)
        self.num_timesteps = self.betas.shape[0]

    def _seq(self):
        if getattr(self.args, "sample_type", "generalized") != "generalized":
            raise NotImplementedError("ddpm_noisy sampling is not a BASELINE config")
        return step_sequence(self.args.skip_type, self.args.timesteps, self.num_timesteps)

    def sample_image(self, x, model, last=True, untill_fake_t=114514, tot=None, cali_ckpt=None, t_max=None):
        xs, x0_preds, x_t, t_t = generalized_steps(x, self._seq(), model, self.betas, eta=self.args.eta,
                                                   untill_fake_t=untill_fake_t, tot=tot, cali_ckpt=cali_ckpt, t_max=t_max)
        out = (xs, x0_preds)
        if last:
            out = out[0][-1]
        return out, x_t, t_t

    def sample_fid(self, model, n_images: int, batch_size: int, cali_ckpt=None, use_graph: bool = True, seed: Optional[int] = None):
        """-> uint8 [n, H, W, 3].  With a QuantModel + calibration checkpoint the whole FSC table is installed
        on the device and each batch is sampled by hipGraph replay."""
        cfg = self.config
        shape = (batch_size, cfg.data.channels, cfg.data.image_size, cfg.data.image_size)
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(seed)
        sampler = None
        if use_graph and hasattr(model, "set_act_table"):
            if cali_ckpt is not None:
                model.set_act_table(cali_ckpt)
            eng = model.engine(self.device)
            if eng.step is None:
                eng.step = torch.zeros(1, dtype=torch.int32, device=self.device)
            sampler = GraphDdimSampler(eng, self._seq(), self.betas.cpu(), batch_size, eta=self.args.eta)
        res = []
        for _ in range(math.ceil(n_images / batch_size)):
            x = torch.randn(shape, generator=g).to(self.device)
            if sampler is not None:
                x0 = sampler.sample(x)
            else:
                x0 = self.sample_image(x, model, cali_ckpt=cali_ckpt, tot=None if cali_ckpt is None else 1)[0]
            img = inverse_data_transform(x0).permute(0, 2, 3, 1).cpu().numpy() * 255.0
            res.append(img.round().astype(np.uint8))
        return np.concatenate(res, axis=0)[:n_images]
