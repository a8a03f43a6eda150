"""configs[1] at its full size through the drop-in surface: `cali_model` end to end on the CIFAR-10 DDPM UNet (35.7 M, random
init) -- weight-scale search, TIAR, every ResnetBlock / AttnBlock / layer reconstruction unit (a few AdaRound iterations each),
Finite-Set activation calibration, checkpoint -- then `load_cali_model` of that checkpoint into a fresh QuantModel and the
calibrated w4a8 forward against the CPU oracle fed with the SAME calibrated state (hard-rounded AdaRound weights, activation
tables).  (The tiny-UNet version of this is pinned to the reference's own run by fixture F8; the recipe at full length is
scratch/cifar_cali_full.py.)"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_cifar_full_unet_cali_model_round_trip_vs_oracle():
    import tfmq_dm_amd.ddim.models as M
    import tfmq_oracle as O
    from quant.calibration import cali_model, load_cali_model
    from quant.quant_layer import QMODE, QuantLayer, Scaler
    from quant.quant_model import QuantModel
    from quant.reconstruction_util import RLOSS
    N, G, ITERS = 32, 2, 6
    m = M.random_init(M.Model(M.make_config()))
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    qnn = QuantModel(m, wq, aq, cali=True, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).to(DEV).eval()
    g = torch.Generator().manual_seed(7)
    xs = torch.randn(G * N, 3, 32, 32, generator=g)
    ts = torch.cat([torch.full((N,), float(t)) for t in np.linspace(981, 1, G).astype(int)])
    path = os.path.join(tempfile.mkdtemp(), "cifar.pth")
    torch.manual_seed(5)
    np.random.seed(5)
    md = cali_model(qnn, (xs, ts), (xs, ts), use_aq=True, path=path, running_stat=True, interval=N, iters=ITERS, batch_size=32,
                    w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    assert [k for k in md if k.startswith("act_")] == ["act_0", "act_1"]
    n_alpha = sum(1 for k in md["weight"] if k.endswith("alpha"))
    assert n_alpha >= 60                        # every conv / linear of the 22 ResnetBlocks, 6 AttnBlocks and the single layers
    assert all(torch.isfinite(v).all() for v in md["weight"].values() if torch.is_tensor(v) and v.is_floating_point())
    # reload into a fresh model
    m2 = M.Model(M.make_config())
    m2.load_state_dict(sd0)
    q2 = QuantModel(m2, wq, aq, cali=False, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).to(DEV).eval()
    load_cali_model(q2, (torch.randn(1, 3, 32, 32), torch.randint(0, 1000, (1,)).float()), use_aq=True, path=path)
    q2.load_state_dict(torch.load(path, map_location="cpu")["act_1"], strict=False)
    x = torch.randn(4, 3, 32, 32, generator=g).to(DEV)
    t = torch.full((4,), 500.0, device=DEV)
    qnn.set_quant_state(False, False)
    fp = qnn(x, t)
    q2.set_quant_state(True, True)
    qe = q2(x, t)
    assert torch.isfinite(qe).all()
    # the same calibrated state through the CPU oracle
    sdc, wqs, aqs = {}, {}, {}
    for n, mod in q2.model.named_modules():
        if isinstance(mod, QuantLayer):
            sdc[n + ".weight"] = mod.original_w.detach().cpu().float()
            if mod.original_b is not None:
                sdc[n + ".bias"] = mod.original_b.detach().cpu().float()
            if mod.use_wq:
                d, z, a = mod.weight_quant_state()
                shp = (-1,) + (1,) * (sdc[n + ".weight"].dim() - 1)
                wqs[n] = {"delta": d.detach().cpu().float().reshape(shp), "zp": z.detach().cpu().float().reshape(shp),
                          "alpha": None if a is None else a.detach().cpu().float()}
            if mod.use_aq and not mod.disable_aq and mod.aqtizer.delta is not None:
                aqs[n] = (float(mod.aqtizer.delta), float(mod.aqtizer.zero_point))
        elif isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear, torch.nn.GroupNorm)):
            for pn, p in mod.named_parameters(recurse=False):
                sdc[f"{n}.{pn}"] = p.detach().cpu().float()
    with torch.no_grad():
        ref = O.ddim_unet_forward(sdc, q2.model.engine_cfg(), x.cpu(), t.cpu().long(), O.QuantSpec(wq=wqs, aq=aqs))
        ref_fp = O.ddim_unet_forward(sdc, q2.model.engine_cfg(), x.cpu(), t.cpu().long(), None)
    rel_q = float((qe.cpu() - ref).norm() / ref.norm())
    rel_fp = float((fp.cpu() - ref_fp).norm() / ref_fp.norm())
    print("CIFAR full size after cali_model: engine w4a8 vs oracle w4a8", rel_q, "| engine FP vs oracle FP", rel_fp)
    assert rel_fp <= 5e-3        # measured 8.5e-4 (fp16-operand convs of the un-quantised layers)
    assert rel_q <= 5e-2         # measured 2.6e-2: bin flips of the 8-bit activation quantizers
