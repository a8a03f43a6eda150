"""Calibration-set generators (reference quant/data_generate.py).  The set for timestep i is produced
by running the FP sampler from fresh noise until step i (`untill_fake_t`), i.e. O(T^2/2) UNet forwards
by construction; here they run on the HIP engine, all on the device."""
from __future__ import annotations

from typing import List, Tuple

import torch

from tfmq_dm_amd._lib import TfmqError


def generate_cali_data_ddim(runnr, model, T: int, c: int, batch_size: int, shape: List[int]) -> Tuple[torch.Tensor]:
    """reference :52-72: for i in 1..T (every c-th): x ~ N(0,I) [batch, *shape]; run `runnr.sample_image`
    until step i; keep (x_t, t)."""
    tmp = []
    for i in range(1, T + 1):
        if i % c == 0:
            x = torch.randn((batch_size, *shape), device=runnr.device)
            x_t, t_t = runnr.sample_image(x, model, untill_fake_t=i)[1:]
            tmp.append((x_t, t_t))
    return tuple(torch.cat([p[k] for p in tmp]) for k in range(2))


def _ldm_only(name):
    def fn(*a, **k):
        raise TfmqError(f"{name}: latent-diffusion samplers are the next row (BASELINE configs 3-5)")
    fn.__name__ = name
    return fn


generate_cali_data_ldm = _ldm_only("generate_cali_data_ldm")
generate_cali_data_ldm_imagenet = _ldm_only("generate_cali_data_ldm_imagenet")
generate_cali_text_guided_data = _ldm_only("generate_cali_text_guided_data")
