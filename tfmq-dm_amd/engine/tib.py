"""Temporal-information block of the DDPM UNet evaluated on its own (TIAR unit)."""
from __future__ import annotations

import torch

from .. import ops
from .blocks import _collect
from .ddim_unet import DdimUNetEngine


def tib_forward_ddim(tib, t: torch.Tensor):
    """QuantTemporalInformationBlockDDIM.forward (reference quant/quant_block.py:52-64): t [m] ->
    tuple of the projections of every ResnetBlock, each [m, Cout_i]."""
    dev = t.device
    sd, wq, rows = _collect(tib.temb, "temb")
    for i, proj in enumerate(tib.temb_projs):
        s2, w2, r2 = _collect(proj, f"proj{i}")
        for k, q in w2.items():
            if q.qid is not None:
                q.qid += len(rows)
        sd.update(s2), wq.update(w2), rows.extend(r2)
    eng = DdimUNetEngine(sd, {}, dev)
    eng.prepare(wq, torch.tensor([rows], dtype=torch.float32, device=dev) if rows else None, None)
    emb = ops.timestep_embedding(t.float().contiguous(), tib.ch)
    h = eng._linear("temb.dense.0", emb, False)
    temb = eng._linear("temb.dense.1", h, True)
    return tuple(eng._linear(f"proj{i}", temb, True) for i in range(len(tib.temb_projs)))


def tib_forward_ldm(tib, t: torch.Tensor):
    """QuantTemporalInformationBlock.forward (reference quant/quant_block.py:99-115): t [m] -> tuple of
    emb_layers(time_embed(timestep_embedding(t))) for every ResBlock, each [m, Cout_i]."""
    dev = t.device
    sd, wq, rows = _collect(tib.t_emb, "time_embed")
    for i, seq in enumerate(tib.emb_layers):
        s2, w2, r2 = _collect(seq, f"emb{i}")
        for q in w2.values():
            if q.qid is not None:
                q.qid += len(rows)
        sd.update(s2), wq.update(w2), rows.extend(r2)
    eng = DdimUNetEngine(sd, {}, dev)
    eng.prepare(wq, torch.tensor([rows], dtype=torch.float32, device=dev) if rows else None, None)
    e0 = ops.timestep_embedding(t.float().contiguous(), tib.model_channels, ldm_order=True)
    h = eng._linear("time_embed.0", e0, False)
    emb = eng._linear("time_embed.2", h, True)
    return tuple(eng._linear(f"emb{i}.1", emb, True) for i in range(len(tib.emb_layers)))
