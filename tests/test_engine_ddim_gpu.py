"""DDIM UNet engine (HIP) vs the reference's golden outputs (fixture F5-F7) and the CPU oracle."""
import numpy as np
import pytest
import torch

import tfmq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = dict(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[8], resolution=16)


def T(a):
    return torch.from_numpy(np.asarray(a))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(y):
    return y.permute(0, 3, 1, 2).contiguous().cpu()


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def env(golden):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tfmq_dm_amd.engine import DdimUNetEngine, LayerQ
    g = golden("f7_ddim_tiny")
    sd = {k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")}
    return g, sd, DdimUNetEngine, LayerQ


def layerq_from_fixture(g, LayerQ, with_act):
    act_names = sorted(k[3:-6] for k in g.files if k.startswith("aq/") and k.endswith("/delta"))
    qid = {n: i for i, n in enumerate(act_names)}
    wq = {}
    for k in g.files:
        if k.startswith("wq/") and k.endswith("/delta"):
            n = k[3:-6]
            wq[n] = LayerQ(T(g[k]), T(g[f"wq/{n}/zp"]), None, qid.get(n) if with_act else None)
    qtable = torch.tensor([[[float(g[f"aq/{n}/delta"]), float(g[f"aq/{n}/zp"])] for n in act_names]])
    return wq, qtable, act_names


def test_fp_forward_f16_tolerance(env):
    """FP mode runs every conv on f16 MFMA (fp32 accumulate) and attention in f16: bar 1e-2
    max-normalised on eps and on every block output (measured ~1e-3)."""
    g, sd, Engine, LayerQ = env
    eng = Engine(sd, CFG, DEV)
    eng.prepare()
    x, t = T(g["x"]), T(g["t"])
    taps = {}
    eps = nchw(eng.forward(nhwc(x), t.to(DEV), taps=taps))
    ref = T(g["eps_fp"])
    assert float((eps - ref).abs().max() / ref.abs().max()) <= 1e-2
    for k in g.files:
        if k.startswith("tap_fp/"):
            out = nchw(taps[k[7:]][1])
            r = T(g[k])
            assert float((out - r).abs().max() / r.abs().max()) <= 1e-2, k
    tib = torch.cat([p.cpu() for p in eng.tib(t.to(DEV))], dim=1)
    np.testing.assert_allclose(tib.numpy(), g["tib_fp"], rtol=0, atol=2e-4 * float(np.abs(g["tib_fp"]).max()))


def test_w4_forward(env):
    """Weight-only 4-bit (integer grid exact in f16, activations rounded to f16)."""
    g, sd, Engine, LayerQ = env
    wq, _, _ = layerq_from_fixture(g, LayerQ, with_act=False)
    eng = Engine(sd, CFG, DEV)
    eng.prepare(wq)
    x, t = T(g["x"]), T(g["t"])
    eps = nchw(eng.forward(nhwc(x), t.to(DEV)))
    ref = T(g["eps_w4"])
    assert float((eps - ref).abs().max() / ref.abs().max()) <= 1e-2
    tib = torch.cat([p.cpu() for p in eng.tib(t.to(DEV))], dim=1)
    np.testing.assert_allclose(tib.numpy(), g["tib_w4"], rtol=0, atol=2e-5 * float(np.abs(g["tib_w4"]).max()))


def test_w4a8_forward_and_trajectory(env):
    """w4a8: integer GEMMs are exact; deviations come only from the f16 un-quantised layers /
    attention and the bin flips they cause downstream.  Bars: eps rel-L2 <= 3e-2 per forward,
    10-step DDIM latent rel-L2 <= 5e-2 (stated tolerance for sampled latents)."""
    g, sd, Engine, LayerQ = env
    wq, qtable, act_names = layerq_from_fixture(g, LayerQ, with_act=True)
    eng = Engine(sd, CFG, DEV)
    eng.prepare(wq, qtable.to(DEV))
    x, t = T(g["x"]), T(g["t"])
    eps = nchw(eng.forward(nhwc(x), t.to(DEV)))
    ref = T(g["eps_w4a8"])
    r = rel_l2(eps, ref)
    print("w4a8 eps rel-L2 vs reference:", r)
    assert r <= 3e-2
    tib = torch.cat([p.cpu() for p in eng.tib(t.to(DEV))], dim=1)
    tr = T(g["tib_w4a8"])
    assert rel_l2(tib, tr) <= 2e-2
    # 10-step trajectory (eta = 0), same act table for every step as in the fixture
    betas = O.linear_betas()
    seq = [int(s) for s in g["seq"]]
    import tfmq_dm_amd.ops as ops
    xt = nhwc(T(g["traj_x0"]))
    traj = T(g["traj_w4a8"])
    seq_next = [-1] + seq[:-1]
    for n, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
        at = O.compute_alpha(betas, torch.tensor([i]))
        an = O.compute_alpha(betas, torch.tensor([j]))
        c2 = (1 - an).sqrt()
        coef = torch.tensor([[float((1 - at).sqrt()), float(at.sqrt()), float(an.sqrt()), 0.0, float(c2), 0, 0, 0]], device=DEV)
        tt = torch.full((xt.shape[0],), float(i), device=DEV)
        e = eng.forward(xt, tt)
        xt = ops.ddim_update(xt, e, coef)
        rr = rel_l2(nchw(xt), traj[n + 1])
        assert rr <= 5e-2, (n, rr)
    print("final latent rel-L2:", rr)


def test_tib_table_and_graph_replay(env):
    """Per-step TIB table + device step counter + hipGraph replay give the same eps as the eager
    per-call path."""
    g, sd, Engine, LayerQ = env
    import tfmq_dm_amd.ops as ops
    from tfmq_dm_amd._lib import handle
    import ctypes as C
    wq, qtable, act_names = layerq_from_fixture(g, LayerQ, with_act=True)
    # 3 FSC steps with slightly different parameters
    qt3 = torch.cat([qtable, qtable * torch.tensor([1.05, 1.0]), qtable * torch.tensor([0.9, 1.0])]).to(DEV)
    qt3[..., 1] = qtable[0, :, 1].to(DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    eng = Engine(sd, CFG, DEV)
    eng.prepare(wq, qt3, step)
    tvals = [999.0, 500.0, 3.0]
    eng.build_tib_table(tvals)
    x = nhwc(T(g["x"]))
    ref = []
    for s, tv in enumerate(tvals):
        step.fill_(s)
        ref.append(eng.forward(x, torch.full((x.shape[0],), tv, device=DEV)).clone())
    arena = ops.Arena()
    stream = torch.cuda.Stream()
    h = handle(0)
    with torch.cuda.stream(stream):
        step.zero_()
        with ops.use_arena(arena):
            out = eng.forward(x, None)
        stream.synchronize()
        assert torch.equal(out, ref[0])
        with ops.use_arena(arena):
            h.call("graph_begin", C.c_void_p(stream.cuda_stream))
            out = eng.forward(x, None)
            ops.step_advance(step, 1)
            gid = C.c_int()
            h.call("graph_end", C.c_void_p(stream.cuda_stream), C.byref(gid))
        for s in range(3):
            h.call("graph_launch", gid.value, C.c_void_p(stream.cuda_stream))
            stream.synchronize()
            assert torch.equal(out, ref[s]), s
        assert int(step.item()) == 3


def test_w4a8_bin_flip_rate_per_layer(env):
    """UNet-level bin agreement: the bins every activation quantizer of the tiny DDPM UNet produces in the engine against the
    bins the oracle's fake-quant forward produces for the same layer, same weights, same Finite-Set row.  The engine's un-quantised
    layers and attention run fp16 operands where the reference runs fp32, so bins start to move at the first quantizer behind
    such a layer (measured 0.9 %) and every moved bin perturbs all outputs of the next integer GEMM, so the share compounds with depth
    (measured 22-47 % in the deep layers of this random-weight net, 29 % overall; 2.8 % of bins move by more than one).  What stays
    small is the size of the move: the dequantised layer inputs differ by <= 3.0e-2 rel-L2 at every layer.  The time-embedding
    projections see exactly the reference's bins (fp32 TIB path).  Bars = measured values with headroom."""
    import tfmq_dm_amd.ops as ops
    g, sd, Engine, LayerQ = env
    wq, qtable, act_names = layerq_from_fixture(g, LayerQ, True)
    eng = Engine(sd, CFG, DEV)
    eng.prepare(wq, qtable.to(DEV))
    x, t = T(g["x"]), T(g["t"])
    eng.set_calibration("record", 0)
    eng.forward(nhwc(x), t.to(DEV))
    eng.set_calibration(None)
    owq = {n: {"delta": q.delta.cpu().reshape((-1,) + (1,) * (sd[n + ".weight"].dim() - 1)),
               "zp": q.zp.cpu().reshape((-1,) + (1,) * (sd[n + ".weight"].dim() - 1)), "alpha": None} for n, q in wq.items()}
    oaq = {n: (qtable[0, i, 0], qtable[0, i, 1]) for i, n in enumerate(act_names)}
    qs = O.QuantSpec(wq=owq, aq=oaq)
    qs.trace = {}
    with torch.no_grad():
        O.ddim_unet_forward(sd, dict(CFG), x, t, qs)
    rates, l2, tot_flip, tot_n, big = {}, {}, 0, 0, 0
    for i, n in enumerate(act_names):
        if i not in eng.observed or n not in qs.trace:
            continue
        xe = eng.observed[i]
        be = (ops.quantize_act(xe.contiguous(), ops.qsel(qtable[:, i:i + 1].contiguous().to(DEV))).to(torch.int32) + 128).cpu()
        bo = qs.trace[n].to(torch.int32)
        if bo.dim() == 4:
            bo = bo.permute(0, 2, 3, 1)
        if bo.numel() == 4 * be.numel():   # Upsample conv: the engine quantises before the nearest-neighbour 2x (the two commute exactly)
            bo = bo[:, ::2, ::2, :]
        bo = bo.reshape(be.shape)
        diff = (be - bo).abs()
        rates[n] = float((diff > 0).float().mean())
        zp = float(qtable[0, i, 1])
        l2[n] = float(diff.float().norm() / (bo.float() - zp).norm().clamp_min(1e-9))   # rel-L2 of the dequantised layer inputs
        tot_flip += int((diff > 0).sum())
        tot_n += diff.numel()
        big += int((diff > 1).sum())
    assert len(rates) >= len(act_names) - 2, (len(rates), len(act_names))
    overall = tot_flip / tot_n
    print("per layer flip rate / dequantised rel-L2:", " ".join(f"{n}={r:.3f}/{l2[n]:.3f}" for n, r in rates.items()))
    print("bin flip rate overall", overall, "worst", max(rates.items(), key=lambda kv: kv[1]), "moves by more than one bin", big / tot_n)
    first = next(n for n in rates if not n.endswith("temb_proj"))
    assert rates[first] <= 2e-2, (first, rates[first])
    assert all(r == 0.0 for n, r in rates.items() if n.endswith("temb_proj"))
    assert overall <= 0.40 and max(rates.values()) <= 0.60
    assert big / tot_n <= 5e-2
    assert max(l2.values()) <= 4.5e-2, max(l2.items(), key=lambda kv: kv[1])
