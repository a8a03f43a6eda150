"""End-to-end `cali_model` on the tiny Stable-Diffusion-style UNet vs the reference's own run (fixture F12), and the
hand-written backward of the BasicTransformerBlock reconstruction unit vs autograd on the oracle's functional block."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

from test_quant_mirror_ldm import T, tiny_qnn  # noqa: E402


def test_transformer_unit_gradients_vs_autograd():
    """One iteration of TransformerUnit: reconstruction loss and dL/dW_hat of all 10 layers against torch autograd
    through the same mathematics (soft weights = plain fp32 weights here: alpha does not enter the comparison)."""
    import torch.nn.functional as F
    import tfmq_dm_amd.ops as ops
    from tfmq_dm_amd.engine import recon as R
    gen = torch.Generator().manual_seed(3)
    B, Tn, Cc, L, Dc, H, I = 3, 16, 32, 5, 24, 2, 128

    def w(*s):
        return (torch.randn(*s, generator=gen) * (1.0 / s[-1] ** 0.5)).requires_grad_(True)
    Wq1, Wk1, Wv1, Wo1 = w(Cc, Cc), w(Cc, Cc), w(Cc, Cc), w(Cc, Cc)
    Wf0, Wf2 = w(2 * I, Cc), w(Cc, I)
    Wq2, Wk2, Wv2, Wo2 = w(Cc, Cc), w(Cc, Dc), w(Cc, Dc), w(Cc, Cc)
    bo1, bf0, bf2, bo2 = (torch.randn(n, generator=gen) * 0.1 for n in (Cc, 2 * I, Cc, Cc))
    norms = [(torch.randn(Cc, generator=gen) * 0.2 + 1, torch.randn(Cc, generator=gen) * 0.1) for _ in range(3)]
    x = torch.randn(B, Tn, Cc, generator=gen)
    ctx = torch.randn(B, L, Dc, generator=gen)
    y = torch.randn(B, Tn, Cc, generator=gen)

    def attn(q, k, v):
        d = Cc // H
        qh, kh, vh = (t.reshape(t.shape[0], t.shape[1], H, d).permute(0, 2, 1, 3) for t in (q, k, v))
        p = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1)
        return (p @ vh).permute(0, 2, 1, 3).reshape(q.shape)
    n1 = F.layer_norm(x, (Cc,), *norms[0], 1e-5)
    x1 = attn(n1 @ Wq1.T, n1 @ Wk1.T, n1 @ Wv1.T) @ Wo1.T + bo1 + x
    n2 = F.layer_norm(x1, (Cc,), *norms[1], 1e-5)
    x2 = attn(n2 @ Wq2.T, ctx @ Wk2.T, ctx @ Wv2.T) @ Wo2.T + bo2 + x1
    n3 = F.layer_norm(x2, (Cc,), *norms[2], 1e-5)
    hc = n3 @ Wf0.T + bf0
    a, gate = hc.chunk(2, dim=-1)
    out = (a * F.gelu(gate)) @ Wf2.T + bf2 + x2
    loss = ((out - y) ** 2).sum(1).mean()             # lp_loss(p=2) as the reference states it (quant_layer.py:152-153): on [B,T,C] dim 1 = tokens
    loss.backward()

    class Plain(R.AdaLayer):                          # AdaLayer whose "soft weight" is the weight itself
        def soft_weight_gemm(self):
            return self.w.reshape(self.cout, self.cin)
    ws = [Wq1, Wk1, Wv1, Wo1, Wf0, Wf2, Wq2, Wk2, Wv2, Wo2]
    bs = [None, None, None, bo1, bf0, bf2, None, None, None, bo2]
    layers = []
    for wt, bt in zip(ws, bs):
        wd = wt.detach().to(DEV)
        layers.append(Plain(wd, torch.ones(wd.shape[0], device=DEV), torch.zeros(wd.shape[0], device=DEV),
                            None if bt is None else bt.to(DEV)))
    unit = R.TransformerUnit(layers, [(g.to(DEV), b.to(DEV)) for g, b in norms], H, x.to(DEV), ctx.to(DEV), y.to(DEV), iters=10)
    rec, grads = unit._forward_backward(torch.arange(B, device=DEV))
    assert abs(float(rec) - float(loss)) <= 1e-4 * abs(float(loss))
    for i, (gw, wt) in enumerate(zip(grads, ws)):
        ref = wt.grad
        err = float((gw.cpu() - ref).abs().max() / ref.abs().max())
        assert err <= 2e-4, (i, err)


def test_ldm_cali_model_matches_reference_run(golden):
    from quant.calibration import cali_model, load_cali_model
    from quant.reconstruction_util import RLOSS
    g = golden("f12_ldm_cali_tiny")
    qnn = tiny_qnn(g, device=DEV)
    xs, ts, cs = T(g["cali_x"]), T(g["cali_t"]), T(g["cali_c"])
    torch.manual_seed(5)
    np.random.seed(5)
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    md = cali_model(qnn, (xs, ts, cs), (xs, ts, cs), use_aq=True, path=path, running_stat=True, interval=16, iters=10,
                    batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    # ---- checkpoint schema
    ref_keys = set(str(k) for k in g["weight_keys"])
    assert set(md["weight"].keys()) == ref_keys, sorted(ref_keys ^ set(md["weight"].keys()))[:8]
    assert [k for k in md if k.startswith("act_")] == ["act_0", "act_1", "act_2"]
    assert sorted(md["act_0"].keys()) == [str(k) for k in g["act_keys"]]
    n_exact = n_tot = 0
    mask_agree, n_alpha = 0.0, 0
    worst = 0.0
    for k in ref_keys:
        ref = T(g["ck/weight/" + k])
        mine = md["weight"][k].float().reshape(ref.shape)
        if k.endswith("wqtizer.delta") or k.endswith("wqtizer.zero_point"):
            n_tot += ref.numel()
            n_exact += int((mine == ref).sum())
        elif k.endswith("alpha"):
            mask_agree += float(((mine >= 0) == (ref >= 0)).float().sum())
            n_alpha += ref.numel()
            worst = max(worst, float((mine - ref).abs().max()))
        elif k.endswith(".w") or k.endswith(".b") or k.endswith(".weight") or k.endswith(".bias"):
            assert torch.equal(mine, ref), k
    assert n_exact / n_tot >= 0.98, n_exact / n_tot
    assert mask_agree / n_alpha >= 0.99, mask_agree / n_alpha       # AdaRound masks
    assert worst <= 5e-2, worst                                      # 10 Adam steps of lr 1e-3 from the same init
    for gi in range(3):
        act = md[f"act_{gi}"]
        keys = sorted(act.keys())
        d = torch.stack([act[k].reshape(()) for k in keys if k.endswith("delta")])
        z = torch.stack([act[k].reshape(()) for k in keys if k.endswith("zero_point")])
        rd, rz = T(g[f"ck/act_{gi}/delta"]), T(g[f"ck/act_{gi}/zp"])
        rel = ((d - rd).abs() / rd).numpy()
        assert np.median(rel) <= 5e-3 and rel.max() <= 0.15, (gi, np.median(rel), rel.max())
        assert float((z - rz).abs().max()) <= 4
    # ---- reload into a fresh model and evaluate one eps with act_1
    qnn2 = tiny_qnn(g, cali=False, device=DEV)
    init = (torch.randn(1, 4, 8, 8), torch.randint(0, 1000, (1,)).float(), torch.randn(1, 5, 64))
    load_cali_model(qnn2, init, use_aq=True, path=path)
    ck = torch.load(path, map_location="cpu")
    qnn2.load_state_dict(ck["act_1"], strict=False)
    xe, te, ce = T(g["reload_x"]).to(DEV), T(g["reload_t"]).to(DEV), T(g["reload_c"]).to(DEV)
    eps = qnn2(xe, te, ce).cpu()
    ref = T(g["reload_eps_act1"])
    rel = float((eps - ref).norm() / ref.norm())
    print("LDM reload eps rel-L2 vs reference:", rel)
    assert rel <= 5e-2
