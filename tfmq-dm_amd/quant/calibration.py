"""cali_model / cali_model_multi / load_cali_model / uaq2adar (reference quant/calibration.py) on the
HIP engine.  Same signatures, same tree walk, same host RNG calls, same checkpoint schema
{'weight': state_dict, 'act_0': {...}, ...}."""
from __future__ import annotations

import logging
from typing import Any, Dict, Tuple

import numpy as np
import torch
import torch.nn as nn

from tfmq_dm_amd import ops
from .adaptive_rounding import AdaRoundQuantizer, RMODE
from .quant_block import BaseQuantBlock
from .quant_layer import QuantLayer, UniformAffineQuantizer
from .quant_model import QuantModel
from .reconstruction import block_reconstruction, layer_reconstruction, tib_reconstruction

logger = logging.getLogger(__name__)


def uaq2adar(model: nn.Module):
    """Wrap every reconstructable weight quantizer in an AdaRoundQuantizer (checkpoint has `alpha`)."""
    for _, child in model.named_children():
        if isinstance(child, QuantLayer):
            if not child.ignore_recon:
                child.weight_quant_state()
                child.wqtizer = AdaRoundQuantizer(child.wqtizer, rmode=RMODE.LEARNED_HARD_SIGMOID,
                                                  w=child.original_w.data.to(child.w.device))   # plain attribute: .to() of the model does not move it
        elif isinstance(child, BaseQuantBlock):
            if not child.ignore_recon:
                for sub in child.modules():
                    if isinstance(sub, QuantLayer):
                        if hasattr(sub, "wqtizer1"):        # QDIFF-split layer: one AdaRound quantizer per input-channel half (reference :35-40)
                            ow, sp = sub.original_w.data.to(sub.w.device), sub.split
                            sub._wq_state(sub.wqtizer, sub.w.data[:, :sp])
                            sub._wq_state(sub.wqtizer1, sub.w.data[:, sp:])
                            sub.wqtizer = AdaRoundQuantizer(sub.wqtizer, rmode=RMODE.LEARNED_HARD_SIGMOID, w=ow[:, :sp, ...])
                            sub.wqtizer1 = AdaRoundQuantizer(sub.wqtizer1, rmode=RMODE.LEARNED_HARD_SIGMOID, w=ow[:, sp:, ...])
                            continue
                        sub.weight_quant_state()
                        sub.wqtizer = AdaRoundQuantizer(sub.wqtizer, rmode=RMODE.LEARNED_HARD_SIGMOID,
                                                        w=sub.original_w.data.to(sub.w.device))
        else:
            uaq2adar(child)


def _parameterise_weight_quantizers(qnn: QuantModel):
    """delta / zero_point become Parameters so that state_dict() carries them (reference :98-105)."""
    for name, module in qnn.model.named_modules():
        if "wqtizer" in name and isinstance(module, (UniformAffineQuantizer, AdaRoundQuantizer)) and module.delta is not None:
            zp = module.zero_point
            module.zero_point = nn.Parameter(zp.detach().clone() if torch.is_tensor(zp) else torch.tensor(float(zp)))
            module.delta = nn.Parameter(module.delta.detach().clone())


ONLY_UNITS = None      # measurement aid (bench.py --cali-only): tuple of qualified-name prefixes; other units keep nearest rounding


def _unit_wanted(path: str) -> bool:
    return ONLY_UNITS is None or any(path.startswith(p) for p in ONLY_UNITS)


def _recon_walk(qnn: QuantModel, model: nn.Module, cali_data, kwargs, rank0=True, prefix=""):
    """Tree walk of recon_model (reference :56-84): TIB first (at `temb`), single layers, blocks."""
    for name, module in model.named_children():
        path = prefix + name
        if rank0:
            logger.info(f"block name: {name} quant: {isinstance(module, BaseQuantBlock)}")
        # the reference's cache placement (calibration.py:62-67; sticky once set): Stable Diffusion from its first input block on, and the up
        # path from output block 8 on, cache on the CPU.  Here it is a hint -- save_inout keeps a cache on the device whenever it fits
        # (288 GB) and moves it to pinned host memory only when it does not (TFMQ_CACHE_HOST=1 follows the hint literally)
        if name == "0" and cali_data[0].shape[-1] == 64 and len(cali_data) == 3:
            kwargs["keep_gpu"] = False
        if prefix.endswith("output_blocks.") and name.isdigit() and int(name) >= 8:
            kwargs["keep_gpu"] = False
        if name == "tib":
            continue
        if name in ("time_embed", "temb"):
            if _unit_wanted("tib"):
                tib_reconstruction(qnn.tib, cali_data=cali_data, **kwargs)
                qnn.invalidate()
            continue
        if isinstance(module, QuantLayer):
            if not module.ignore_recon and _unit_wanted(path):
                layer_reconstruction(qnn, module, cali_data=cali_data, **kwargs)
        elif isinstance(module, BaseQuantBlock):
            if not module.ignore_recon and _unit_wanted(path):
                block_reconstruction(qnn, module, cali_data=cali_data, **kwargs)
        else:
            _recon_walk(qnn, module, cali_data, kwargs, rank0, path + ".")


def _calibrate_activations(qnn: QuantModel, a_cali_data, interval: int, running_stat: bool, model_dict: dict, rank0=True,
                           sync=None):
    """Finite-Set Calibration (reference :108-152): per timestep group, re-initialise every live activation
    quantizer on 16 random samples (MSE), then one running-stat pass in batches of 16 (EMA -> MINMAX)."""
    dev = next(qnn.model.parameters()).device
    qnn.set_quant_state(use_wq=True, use_aq=True)
    for _, l in qnn.named_quant_layers():     # del module.delta / zero_point; init = False
        l.aqtizer.delta, l.aqtizer.zero_point, l.aqtizer.init = None, None, False
    eng = qnn.engine(dev)
    names = qnn.act_layer_names()
    layers = dict(qnn.named_quant_layers())
    scaler = getattr(layers[names[0]].aqtizer.scaler, "__name__", "mse")
    n_groups = a_cali_data[0].shape[0] // interval
    for time in range(n_groups):
        tx = a_cali_data[0][time * interval:(time + 1) * interval]
        tt = a_cali_data[1][time * interval:(time + 1) * interval]
        tc = a_cali_data[2][time * interval:(time + 1) * interval] if len(a_cali_data) > 2 else None
        n = tx.shape[0]
        batch_size = min(16, n)

        def fwd(sel):
            args = [ops.nchw_to_nhwc(tx[sel].to(dev).float().contiguous()), tt[sel].to(dev).float().contiguous()]
            if tc is not None:       # context-conditioned UNet (Stable Diffusion: text-encoder output)
                args.append(tc[sel].to(dev).float().contiguous())
            eng.forward(*args)
        inds = np.random.choice(n, 16, replace=False)
        eng.set_calibration("init" if scaler == "mse" else "init_minmax", 0)
        fwd(inds)
        if running_stat:
            inds = np.arange(n)
            np.random.shuffle(inds)
            eng.set_calibration("running", 0)
            for i in range(0, n, batch_size):
                fwd(inds[i:i + batch_size])
        eng.set_calibration(None)
        if sync is not None:
            sync(eng.qtable[0, :, 0])            # all-average of the deltas only (quant_model.py:127-132)
        qt = eng.qtable[0].cpu()
        st = eng.act_state.cpu()
        temp = {}
        for i, nme in enumerate(names):
            q = layers[nme].aqtizer
            q.delta = nn.Parameter(qt[i, 0].clone().to(dev))
            q.zero_point = nn.Parameter(qt[i, 1].clone().to(dev))
            q.x_min, q.x_max, q.init = st[i, 0].clone().to(dev), st[i, 1].clone().to(dev), True
            temp[f"model.{nme}.aqtizer.delta"] = qt[i, 0].clone()
            temp[f"model.{nme}.aqtizer.zero_point"] = qt[i, 1].clone()
        if rank0:
            model_dict[f"act_{time}"] = temp


def cali_model(qnn: QuantModel, w_cali_data: Tuple[torch.Tensor], a_cali_data: Tuple[torch.Tensor], use_aq: bool = False,
               path: str = None, running_stat: bool = False, interval: int = 128, **kwargs) -> None:
    logger.info("Calibrating...")
    dev = next(qnn.model.parameters()).device
    # --------- weight initialization (per-channel scale search on the device) -------- #
    cali_data = w_cali_data
    qnn.set_quant_state(use_wq=True, use_aq=False)
    batch_size = min(8, cali_data[0].shape[0])
    qnn(*(x[:batch_size].to(dev) for x in cali_data))
    qnn.disable_out_quantization()
    qnn.invalidate()
    # --------- weight quantization: TIAR + layer / block reconstruction -------- #
    _recon_walk(qnn, qnn, cali_data, kwargs)
    qnn.set_quant_state(use_wq=True, use_aq=False)
    if hasattr(qnn, "tib"):
        delattr(qnn, "tib")
    _parameterise_weight_quantizers(qnn)
    model_dict = {"weight": {k: v.detach().cpu() for k, v in qnn.state_dict().items()}}
    if use_aq:
        _calibrate_activations(qnn, a_cali_data, interval, running_stat, model_dict)
        if path:
            torch.save(model_dict, path)
    logger.info("Calibration done.")
    return model_dict


def shard_for_rank(data: Tuple[torch.Tensor], interval: int, world_size: int, rank: int) -> Tuple[torch.Tensor]:
    """Per-timestep-group shard (reference :269-282): for every group j the rank takes
    [j*I + rank*I//W, j*I + (rank+1)*I//W)."""
    out = []
    for t in data:
        parts = [t[j * interval + rank * interval // world_size: j * interval + (rank + 1) * interval // world_size]
                 for j in range(t.shape[0] // interval)]
        out.append(torch.cat(parts, dim=0))
    return tuple(out)


def cali_model_multi(gpu: int, dist_backend: str, world_size: int, dist_url: str, rank: int, ngpus_per_node: int, model,
                     use_aq: bool, path: str, w_cali_data: Tuple[torch.Tensor], a_cali_data: Tuple[torch.Tensor],
                     interval: int, running_stat: bool, kwargs: Dict[str, Any]) -> None:
    """One process per GPU (mp.spawn target, reference :228-389): shard the calibration sets, replicate the
    model, SUM all-reduce of the unit's gradients every iteration (RCCL over xGMI), all-average of the
    activation deltas, rank 0 writes the checkpoint."""
    from tfmq_dm_amd import linklink as dist
    rank = rank * ngpus_per_node + gpu
    # (the reference selects the device AFTER the rendezvous, :241-245; doing it first costs nothing and every collective --
    # torch's own and the C ABI's lazily created RCCL communicator, linklink.init_comm -- then sees one rank per GPU)
    torch.cuda.set_device(gpu)
    dist.init_process_group(backend=dist_backend, init_method=dist_url, world_size=world_size, rank=rank)
    net = model.diffusion_model if hasattr(model, "diffusion_model") else model
    net.cuda()
    qnn = QuantModel(net, wq_params=kwargs.pop("wq_params"), aq_params=kwargs.pop("aq_params"),
                     softmax_a_bit=kwargs.pop("softmax_a_bit", 8), aq_mode=kwargs.pop("aq_mode", None) or [2])
    kwargs.pop("no_grad_ckpt", None)
    qnn.cuda()
    qnn.eval()
    w_cali_data = shard_for_rank(w_cali_data, interval, world_size, gpu)
    a_cali_data = shard_for_rank(a_cali_data, interval, world_size, gpu)
    dev = next(qnn.model.parameters()).device
    qnn.set_quant_state(use_wq=True, use_aq=False)
    batch_size = min(64, w_cali_data[0].shape[0])
    qnn(*(x[:batch_size].to(dev) for x in w_cali_data))
    qnn.disable_out_quantization()
    qnn.invalidate()
    _recon_walk(qnn, qnn, w_cali_data, kwargs, rank0=rank == 0)
    qnn.set_quant_state(use_wq=True, use_aq=False)
    if hasattr(qnn, "tib"):
        delattr(qnn, "tib")
    model_dict = {}
    if rank == 0:
        _parameterise_weight_quantizers(qnn)
        model_dict = {"weight": {k: v.detach().cpu() for k, v in qnn.state_dict().items()}}
    if use_aq:
        def sync(delta_col):
            # all-average of the deltas (quant_model.py:127-132): the column is a strided view of the table, the
            # collective wants a contiguous buffer
            d = delta_col.contiguous()
            d /= world_size
            dist.allreduce(d)
            delta_col.copy_(d)
        _calibrate_activations(qnn, a_cali_data, interval // world_size, running_stat, model_dict, rank0=rank == 0,
                               sync=sync if ngpus_per_node > 1 else None)
        if path and rank == 0:
            torch.save(model_dict, path)
    logger.info("Calibration done.")
    return qnn      # (the reference returns None; mp.spawn ignores the value -- tests read the replica's final state)


def load_cali_model(qnn: QuantModel, init_data: Tuple[torch.Tensor], use_aq: bool = False, path: str = None) -> None:
    """reference :158-224: init weight quantizers (one sample), disable first/last, wrap in AdaRound if
    the checkpoint has alphas, load w/b/alpha/delta/zero_point; with use_aq create the activation
    quantizers so per-step `act_k` dicts (or set_act_table) can be loaded."""
    logger.info("Loading calibration model...")
    ckpt = torch.load(path, map_location="cpu")["weight"]
    dev = next(qnn.model.parameters()).device
    qnn.set_quant_state(use_wq=True, use_aq=False)
    _ = qnn(*(d.to(dev) for d in init_data))
    qnn.disable_out_quantization()
    if any("alpha" in k for k in ckpt):
        uaq2adar(qnn)
    _parameterise_weight_quantizers(qnn)
    ckpt = {k: v for k, v in ckpt.items() if "aqtizer" not in k}
    qnn.load_state_dict(ckpt, strict=False)
    qnn.set_quant_state(use_wq=True, use_aq=False)
    for module in qnn.model.modules():
        if isinstance(module, (AdaRoundQuantizer, UniformAffineQuantizer)) and isinstance(module.delta, nn.Parameter) \
                and not getattr(module, "leaf_param", False):
            z, d = module.zero_point.data, module.delta.data
            del module.zero_point, module.delta
            module.zero_point, module.delta = z, d
            if isinstance(module, AdaRoundQuantizer):
                module._version_ += 1
    qnn.invalidate()
    if use_aq:
        qnn.set_quant_state(use_wq=True, use_aq=True)
        for _, l in qnn.named_quant_layers():
            if l.use_aq and not l.disable_aq:
                q = l.aqtizer
                q.delta = nn.Parameter(torch.ones((), device=dev))
                q.zero_point = nn.Parameter(torch.zeros((), device=dev))
                q.init = True
    logger.info("Loading calibration model done.")
