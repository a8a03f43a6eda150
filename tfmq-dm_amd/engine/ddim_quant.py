"""Quantisation bookkeeping for the DDPM UNet at engine level: which layers are QuantLayers, in
which order, which keep FP / weight-only state, weight-scale search and Finite-Set activation
calibration driven on the device.

Mirrors: QuantModel.quant_module name filter (quant/quant_model.py:57-58), disable_out_quantization
(:103-120), weight quantizer init (quant/calibration.py:86-92) and the activation calibration loop
(:108-152).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import ops
from .ddim_unet import DdimUNetEngine, LayerQ


def quant_layer_names(cfg: dict) -> List[str]:
    """QuantLayers of the DDPM UNet in `model.modules()` order (nin_shortcut and downsample.conv are
    never quantised, quant/quant_model.py:57-58)."""
    nlev, nres = len(cfg["ch_mult"]), cfg["num_res_blocks"]

    def rb(p):
        return [p + ".conv1", p + ".temb_proj", p + ".conv2"]

    def ab(p):
        return [p + s for s in (".q", ".k", ".v", ".proj_out")]

    names = ["temb.dense.0", "temb.dense.1", "conv_in"]
    res = cfg["resolution"]
    for i in range(nlev):
        for j in range(nres):
            names += rb(f"down.{i}.block.{j}")
        if res in cfg["attn_resolutions"]:
            for j in range(nres):
                names += ab(f"down.{i}.attn.{j}")
        if i != nlev - 1:
            res //= 2
    names += rb("mid.block_1") + ab("mid.attn_1") + rb("mid.block_2")
    for i in range(nlev):
        r = cfg["resolution"] // (2 ** i)
        for j in range(nres + 1):
            names += rb(f"up.{i}.block.{j}")
        if r in cfg["attn_resolutions"]:
            for j in range(nres + 1):
                names += ab(f"up.{i}.attn.{j}")
        if i != 0:
            names.append(f"up.{i}.upsample.conv")
    names.append("conv_out")
    return names


def fp_layer_names(cfg) -> List[str]:
    n = quant_layer_names(cfg)
    return [n[0], n[2], n[-1]]            # ignore_recon layers stay FP (quant_model.py:110-120)


def act_layer_names(cfg) -> List[str]:
    n = quant_layer_names(cfg)
    dead = {n[0], n[1], n[2], n[3], n[-1]}  # disable_aq
    return [x for x in n if x not in dead]


def init_weight_quant(sd: Dict[str, torch.Tensor], cfg: dict, scaler: str = "mse", bits: int = 4,
                      device="cuda:0") -> Dict[str, LayerQ]:
    """Per-output-channel weight quantizer init on the device (K2/K3): the reference's Python loop
    over channels x 80 candidates (quant_layer.py:193-204, ~94 s on CPU for this UNet)."""
    names = quant_layer_names(cfg)
    skip = set(fp_layer_names(cfg))
    out = {}
    for n in names:
        if n in skip:
            continue
        w = sd[n + ".weight"].to(device, torch.float32).contiguous()
        rows = w.shape[0]
        if scaler == "mse":
            qp = ops.mse_search(w, rows, 2 ** bits)
        else:
            qp = ops.minmax_to_qparam(ops.minmax(w, rows), 2 ** bits)
        shape = (-1,) + (1,) * (w.dim() - 1)
        out[n] = LayerQ(qp[:, 0].contiguous().view(shape), qp[:, 1].contiguous().view(shape), None, None)
    return out


def attach_act_ids(wq: Dict[str, LayerQ], cfg) -> List[str]:
    names = act_layer_names(cfg)
    for i, n in enumerate(names):
        wq[n].qid = i
    return names


def calibrate_activations(eng: DdimUNetEngine, groups: Sequence[Tuple[torch.Tensor, torch.Tensor]], running_stat: bool = True,
                          init_batch: int = 16, batch: int = 16, rng: Optional[np.random.RandomState] = None,
                          scaler: str = "mse"):
    """Finite-Set Calibration (quant/calibration.py:108-152): for every timestep group k,
    re-initialise all activation quantizers on 16 random samples (MSE scaler), then one running-stat
    pass over the group in batches of 16 (EMA min/max -> MINMAX).  groups[k] = (x NHWC, t) on the device.
    Fills eng.qtable[k]."""
    rng = rng or np.random
    for k, (x, t) in enumerate(groups):
        n = x.shape[0]
        inds = rng.choice(n, min(init_batch, n), replace=False)
        idx = torch.as_tensor(inds, device=x.device)
        eng.set_calibration("init" if scaler == "mse" else "init_minmax", k)
        eng.forward(x.index_select(0, idx).contiguous(), t.index_select(0, idx).contiguous())
        if running_stat:
            order = np.arange(n)
            rng.shuffle(order)
            eng.set_calibration("running", k)
            for i in range(0, n, batch):
                idx = torch.as_tensor(order[i:i + batch], device=x.device)
                eng.forward(x.index_select(0, idx).contiguous(), t.index_select(0, idx).contiguous())
    eng.set_calibration(None)
    return eng.qtable
