"""Harness that imports the *reference* (read-only, /root/reference) on CPU.

Only used by the golden-vector generator scripts in this directory, which run in the
build container (the reference never travels to the GPU box).  Nothing under tests/
that runs at test time imports this file.

Recipe = SURVEY.md Appendix B: stub the two import-only dependencies, make `.cuda()`
an identity and map 'cuda' device strings to 'cpu'.
"""
import os
import sys
import types

REF = "/root/reference"


_installed = None


def install():
    global _installed
    if _installed is not None:
        return _installed
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present; golden generation runs only in the build container")
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    for p in (REF + "/stable-diffusion", REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.nn as nn

    # type-annotation-only import in quant/calibration.py:2 and quant/data_generate.py
    m = types.ModuleType("ldm.models.diffusion.ddpm")
    m.LatentDiffusion = object
    sys.modules["ldm.models.diffusion.ddpm"] = m
    for name in ("ldm.models.diffusion.dpm_solver", "ldm.models.diffusion.dpm_solver.sampler"):
        mm = types.ModuleType(name)
        mm.DPMSolverSampler = object
        sys.modules[name] = mm
    oc = types.ModuleType("omegaconf")
    ocl = types.ModuleType("omegaconf.listconfig")

    class ListConfig(list):
        pass

    ocl.ListConfig = ListConfig
    oc.listconfig = ocl
    sys.modules["omegaconf"] = oc
    sys.modules["omegaconf.listconfig"] = ocl

    # hard-coded device strings -> cpu
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    _to = torch.Tensor.to

    def _is_cuda(a):
        return (isinstance(a, str) and a.startswith("cuda")) or (isinstance(a, torch.device) and a.type == "cuda")

    def to(self, *args, **kwargs):
        args = tuple("cpu" if _is_cuda(a) else a for a in args)
        if _is_cuda(kwargs.get("device")):
            kwargs["device"] = "cpu"
        return _to(self, *args, **kwargs)

    torch.Tensor.to = to
    _installed = torch
    return torch


def ddim_config(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16,
                in_channels=3, out_ch=3):
    import argparse
    ns = argparse.Namespace
    cfg = ns(
        data=ns(image_size=image_size, channels=in_channels),
        model=ns(type="simple", in_channels=in_channels, out_ch=out_ch, ch=ch, ch_mult=list(ch_mult),
                 num_res_blocks=num_res_blocks, attn_resolutions=list(attn_resolutions), dropout=0.0,
                 resamp_with_conv=True),
        diffusion=ns(beta_schedule="linear", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000),
        split_shortcut=True,
    )
    return cfg


def rerandomize_zero_params(model, std=0.02, seed=7):
    """all-zero params crash the reference's minmax; redraw them."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            if p.numel() > 0 and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * std)
