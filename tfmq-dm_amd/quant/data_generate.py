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


def _real_time(T: int, t: int, ddpm_time_num: int = 1000) -> int:
    """DDPM timestep the sampler would evaluate next after stopping at step t of T (reference :41-42,100-101)."""
    return (T - t) * ddpm_time_num // T + 1


def _stack(tmp):
    return tuple(torch.cat([p[k] for p in tmp]) for k in range(len(tmp[0])))


def _is_ddim_like(sampler) -> bool:
    from tfmq_dm_amd.ldm.ddim import DDIMSampler          # PLMSSampler derives from it
    return isinstance(sampler, DDIMSampler) or sampler.__class__.__name__ in ("DDIMSampler", "PLMSSampler")


def _fast(sampler) -> dict:
    """The package's own DDIM / PLMS samplers replay captured step graphs when neither callbacks nor intermediates are
    asked for (ldm/ddim.py: _graph_sample) -- same recurrence, same kernels; other sampler objects are called as is."""
    from tfmq_dm_amd.ldm.ddim import DDIMSampler
    return {"_graph": True} if isinstance(sampler, DDIMSampler) else {}


def generate_cali_data_ldm(model, T: int, c: int, batch_size: int, shape: List[int], vanilla: bool = False,
                           dpm: bool = False, plms: bool = False, eta: float = 0.0) -> Tuple[torch.Tensor]:
    """reference :75-112 (unconditional LDMs): for every c-th step t, sample from fresh noise until step t with the
    DDIM / PLMS sampler and keep (x_t, the DDPM timestep the next model call would see)."""
    from tfmq_dm_amd.ldm.ddim import DDIMSampler, PLMSSampler
    if vanilla:
        raise NotImplementedError("Vanilla LDM is not implemented yet, because it needs 1000 steps to generate one sample.")
    from tfmq_dm_amd.ldm.dpm_solver import DPMSolverSampler
    sampler = DPMSolverSampler(model) if dpm else (PLMSSampler(model) if plms else DDIMSampler(model))
    tmp = []
    for t in range(1, T + 1):
        if t % c == 0:
            x_t, t_t = sampler.sample(S=T, batch_size=batch_size, shape=shape, verbose=False, eta=eta, untill_fake_t=t, **_fast(sampler))
            if _is_ddim_like(sampler):       # DPM-Solver returns its own (continuous) time labels
                t_t = torch.full((batch_size,), _real_time(T, t), device=sampler.model.betas.device, dtype=torch.long)
            tmp.append((x_t, t_t))
    return _stack(tmp)


def _sample_many(sampler, T: int, until: int, conds, uconds, batch_size: int, shape, scale: float, eta: float, max_batch: int):
    """One sampling per (timestep, prompt / class) is what the reference runs (a host loop around `sampler.sample`).  The
    samples are independent and the engine is batch independent (bit for bit), so several prompts share a sampler call:
    their x_T are drawn one prompt after the other -- the reference's RNG call order -- then stacked.  Returns the
    per-prompt x_t list in prompt order."""
    dev = sampler.model.betas.device
    per = max(1, max_batch // batch_size)
    out = []
    for i in range(0, len(conds), per):
        cs, us = conds[i:i + per], uconds[i:i + per]
        x_T = torch.cat([torch.randn((batch_size, *shape), device=dev) for _ in cs])
        x_t, _ = sampler.sample(S=T, conditioning=torch.cat(cs), batch_size=batch_size * len(cs), shape=shape, verbose=False,
                                eta=eta, unconditional_guidance_scale=scale, unconditional_conditioning=torch.cat(us),
                                x_T=x_T, untill_fake_t=until, **_fast(sampler))
        out += list(x_t.split(batch_size))
    return out


def generate_cali_data_ldm_imagenet(model, T: int, c: int, batch_size: int, shape: List[int], eta: float = 0.0,
                                    scale: float = 3.0, max_batch: int = 64) -> Tuple[torch.Tensor]:
    """reference :115-154 (class-conditional LDM, classifier-free guidance): 32 class labels x every c-th step; both the
    conditional and the unconditional context of each sample enter the set.  `model` supplies the class embedder
    (`get_learned_conditioning`, `cond_stage_key`, `ema_scope`): glue outside this package.  max_batch: samples per
    sampler call (several classes share one, see _sample_many)."""
    from contextlib import nullcontext
    from tfmq_dm_amd.ldm.ddim import DDIMSampler
    sampler = DDIMSampler(model)
    tmp = []
    classes = [i for i in range(0, 1000, 1000 // 31)]
    scope = model.ema_scope() if hasattr(model, "ema_scope") else nullcontext()
    with torch.no_grad(), scope:
        for i in range(1, T + 1):
            if i % c == 0:
                uc_t = model.get_learned_conditioning({model.cond_stage_key: torch.tensor(batch_size * [1000]).to(model.device)})
                c_ts = [model.get_learned_conditioning({model.cond_stage_key: torch.tensor(batch_size * [cl]).to(model.device)})
                        for cl in classes]
                x_ts = _sample_many(sampler, T, i, c_ts, [uc_t] * len(c_ts), batch_size, shape, scale, eta, max_batch)
                t_t = torch.full((batch_size,), _real_time(T, i), device=sampler.model.betas.device, dtype=torch.long)
                for x_t, c_t in zip(x_ts, c_ts):
                    tmp += [(x_t, t_t, c_t), (x_t, t_t, uc_t)]
    return _stack(tmp)


def generate_cali_text_guided_data(model, sampler, T: int, c: int, batch_size: int, prompts: Tuple[str], shape: List[int],
                                   precision_scope=None, max_batch: int = 64) -> Tuple[torch.Tensor]:
    """reference :13-49 (Stable Diffusion): for every c-th step and every prompt, CFG-7.5 sampling from fresh noise until
    step t; (x_t, t, c) and (x_t, t, uc) both enter the set.  `model.get_learned_conditioning` is the text encoder
    (glue outside this package); `precision_scope` is accepted for signature compatibility (the engine's precision is
    fixed: int8 / f16 MFMA with fp32 accumulation).  max_batch: samples per sampler call (several prompts share one, see
    _sample_many; DDIM / PLMS samplers of this package only -- any other sampler object is called once per prompt)."""
    tmp = []
    if hasattr(model, "eval"):
        model.eval()
    with torch.no_grad():
        for t in range(1, T + 1):
            if t % c == 0:
                ucs = [model.get_learned_conditioning(batch_size * [""]) for _ in prompts]
                cts = [model.get_learned_conditioning(batch_size * [p]) for p in prompts]
                if _is_ddim_like(sampler) and _fast(sampler):
                    x_ts = _sample_many(sampler, T, t, cts, ucs, batch_size, shape, 7.5, 0.0, max_batch)
                    t_t = torch.full((batch_size,), _real_time(T, t), device=sampler.model.betas.device, dtype=torch.long)
                    for x_t, c_t, uc_t in zip(x_ts, cts, ucs):
                        tmp += [(x_t, t_t, c_t), (x_t, t_t, uc_t)]
                    continue
                for c_t, uc_t in zip(cts, ucs):
                    x_t, t_t = sampler.sample(S=T, conditioning=c_t, batch_size=batch_size, shape=shape, verbose=False,
                                              unconditional_guidance_scale=7.5, unconditional_conditioning=uc_t,
                                              untill_fake_t=t)
                    if _is_ddim_like(sampler):
                        t_t = torch.full((batch_size,), _real_time(T, t), device=sampler.model.betas.device, dtype=torch.long)
                    tmp += [(x_t, t_t, c_t), (x_t, t_t, uc_t)]
    return _stack(tmp)
