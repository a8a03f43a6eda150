"""End-to-end `cali_model` of the quant/ mirror on the device vs the reference's own tiny calibration
run (fixture F8: ch=32 DDPM UNet, 3 timestep groups x 16, iters=10, seeds 5/5): same tree walk, same
host RNG calls => same mini-batches and sample subsets.  Then `load_cali_model` round trip."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


def build(g, cali=True):
    import tfmq_dm_amd.ddim.models as M
    from quant.quant_layer import QMODE, Scaler
    from quant.quant_model import QuantModel
    m = M.Model(M.make_config(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16, dropout=0.0))
    m.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")})
    m.to(DEV).eval()
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    return QuantModel(m, wq, aq, cali=cali, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).to(DEV).eval()


def test_cali_model_matches_reference_run(golden):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from quant.calibration import cali_model, load_cali_model
    from quant.reconstruction_util import RLOSS
    g = golden("f8_cali_tiny")
    qnn = build(g)
    xs, ts = T(g["cali_x"]), T(g["cali_t"])
    torch.manual_seed(5)
    np.random.seed(5)
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    md = cali_model(qnn, (xs, ts), (xs, ts), use_aq=True, path=path, running_stat=True, interval=16, iters=10,
                    batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    # ---- checkpoint schema
    ref_keys = set(str(k) for k in g["weight_keys"])
    assert set(md["weight"].keys()) == ref_keys, sorted(ref_keys ^ set(md["weight"].keys()))[:8]
    assert [k for k in md if k.startswith("act_")] == ["act_0", "act_1", "act_2"]
    assert sorted(md["act_0"].keys()) == [str(k) for k in g["act_keys"]]
    # ---- weight quantizers: per-channel MSE scale search on the device vs the reference's Python loop
    n_exact = n_tot = 0
    mask_agree, n_alpha = 0.0, 0
    for k in ref_keys:
        ref = T(g["ck/weight/" + k])
        mine = md["weight"][k].float().reshape(ref.shape)
        if k.endswith("wqtizer.delta") or k.endswith("wqtizer.zero_point"):
            n_tot += ref.numel()
            n_exact += int((mine == ref).sum())
        elif k.endswith("alpha"):
            mask_agree += float(((mine >= 0) == (ref >= 0)).float().sum())
            n_alpha += ref.numel()
            assert float((mine - ref).abs().max()) <= 5e-2, k     # 10 Adam steps of lr 1e-3 from the same init
        elif k.endswith(".w") or k.endswith(".b") or k.endswith(".weight") or k.endswith(".bias"):
            assert torch.equal(mine, ref), k
    assert n_exact / n_tot >= 0.98, n_exact / n_tot   # argmin ties of the 80-candidate search may differ
    assert mask_agree / n_alpha >= 0.99               # AdaRound masks (F8 bar: >= 99 %)
    # ---- activation tables: same sample subsets (np.random) -> close parameters
    for gi in range(3):
        act = md[f"act_{gi}"]
        keys = sorted(act.keys())
        d = torch.stack([act[k].reshape(()) for k in keys if k.endswith("delta")])
        z = torch.stack([act[k].reshape(()) for k in keys if k.endswith("zero_point")])
        rd, rz = T(g[f"ck/act_{gi}/delta"]), T(g[f"ck/act_{gi}/zp"])
        rel = ((d - rd).abs() / rd).numpy()
        assert np.median(rel) <= 5e-3 and rel.max() <= 0.1, (gi, np.median(rel), rel.max())
        assert float((z - rz).abs().max()) <= 3
    # ---- reload into a fresh model and sample one eps with act_1
    qnn2 = build(g, cali=False)
    init = (torch.randn(1, 3, 16, 16), torch.randint(0, 1000, (1,)).float())
    load_cali_model(qnn2, init, use_aq=True, path=path)
    ck = torch.load(path, map_location="cpu")
    qnn2.load_state_dict(ck["act_1"], strict=False)
    xe, te = T(g["reload_x"]).to(DEV), T(g["reload_t"]).to(DEV)
    eps = qnn2(xe, te).cpu()
    ref = T(g["reload_eps_act1"])
    rel = float((eps - ref).norm() / ref.norm())
    print("reload eps rel-L2 vs reference:", rel)
    assert rel <= 5e-2
    # the FSC table path gives the same result as the per-step load_state_dict path
    qnn2.set_act_table(ck)
    qnn2._act_step.fill_(1)
    eps2 = qnn2(xe, te).cpu()
    assert torch.equal(eps2, eps)
