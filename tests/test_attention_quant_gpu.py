"""SURVEY section 8f-3, forward half: the attention-matmul activation quantizers (aqtizer_q / _k / _v and the always-zero softmax
quantizer aqtizer_w of QuantAttnBlock, cross_attn_forward, QuantQKMatMul / QuantSMVMatMul; reference quant/quant_block.py:226-243,
318-323,350-351,487-498) switched ON -- which no driver of the reference does; fixture F21 was produced by importing the reference and
setting `use_aq = True` on those blocks by hand.

The engine then runs ops.attention_quant: fake-quantised q, k, v (8 bit), scores and P V as exact fp32 products of the dequantised values
(strided MFMA GEMMs), fp32 row softmax, softmax quantised with zero point 0.  Checked in the exact-fp32 diagnostics mode (so that the
inputs of the quantizers are the reference's up to summation order) and in the default fast mode:
  * eps against the reference's eps with the attention quantizers on,
  * the BINS every attention quantizer produces against the reference's bins of the same tensor,
  * lazy (MSE) initialisation of those quantizers on the device against the reference's deltas.
Not built: an int8-MFMA kernel for these matmuls and the delta-learning reconstruction mode (reconstruction.py:135-166); see DESIGN.md."""
import numpy as np
import pytest
import torch

import tfmq_oracle as O
from _avalanche import avalanche, engine_bins_vs_trace, first_divergence, tie_distance

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(y):
    return y.permute(0, 3, 1, 2).contiguous().cpu()


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


CASES = {
    "ddim": ("f7_ddim_tiny", dict(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[8], resolution=16)),
    "ldm": ("f11_ldm_tiny", dict(model_channels=32, num_heads=2, in_channels=4)),
    "attnblock": ("f13_ldm_attnblock_tiny", dict(model_channels=32, num_heads=-1, in_channels=3, num_head_channels=16)),
}


def engine_key(which, name):
    """quantizer name of the fixture -> (engine attention key, role)"""
    role = name[-1]
    base = name[:-len(".aqtizer_x")]
    if which == "attnblock":
        base = base.rsplit(".", 1)[0]           # ...attention.qkv_matmul -> ...attention
    return base, role


def _setup(golden, which, monkeypatch, exact, use_ref_attn_params=True):
    from tfmq_dm_amd.engine import DdimUNetEngine, LayerQ, LdmUNetEngine
    g = golden("f21_attention_quant")
    fx, cfg = CASES[which]
    base = golden(fx)
    sd = {k[3:]: T(base[k]) for k in base.files if k.startswith("sd/")}
    if exact:
        monkeypatch.setenv("TFMQ_EXACT_FP", "1")
    else:
        monkeypatch.delenv("TFMQ_EXACT_FP", raising=False)
    eng = (DdimUNetEngine if which == "ddim" else LdmUNetEngine)(sd, cfg, DEV)
    pre = which + "/"
    act_names = sorted(k[len(pre) + 3:-6] for k in g.files if k.startswith(pre + "aq/") and k.endswith("/delta"))
    qid = {n: i for i, n in enumerate(act_names)}
    wq = {}
    for k in g.files:
        if k.startswith(pre + "wq/") and k.endswith("/delta"):
            n = k[len(pre) + 3:-6]
            wq[n] = LayerQ(T(g[k]), T(g[f"{pre}wq/{n}/zp"]), None, qid.get(n))
    rows = [[float(g[f"{pre}aq/{n}/delta"]), float(g[f"{pre}aq/{n}/zp"])] for n in act_names]
    anames = [str(n) for n in g[pre + "attn_q_names"]]
    attn_q = {}
    for j, n in enumerate(anames):
        key, role = engine_key(which, n)
        attn_q.setdefault(key, {})[role] = len(act_names) + j
        if role == "w":
            attn_q[key]["w_level"] = int(g[f"{pre}attn_q/{n}/level"])
        rows.append([float(g[f"{pre}attn_q/{n}/delta"]), float(g[f"{pre}attn_q/{n}/zp"])] if use_ref_attn_params else [1.0, 0.0])
    qtable = torch.tensor([rows])
    x, t = T(base["x"]), T(base["t"]).float()
    args = (nhwc(x), t.to(DEV)) + ((T(base["ctx"]).to(DEV),) if which == "ldm" else ())
    eng.prepare(wq, qtable.to(DEV), None, attn_q=attn_q)
    eng._test_ctx = (sd, cfg, base, act_names, wq)
    return g, eng, args, pre, anames, len(act_names), qtable


@pytest.mark.parametrize("which", ["ddim", "ldm", "attnblock"])
@pytest.mark.parametrize("exact", [True, False])
def test_eps_and_bins_with_attention_quantizers_on(golden, monkeypatch, which, exact):
    import tfmq_dm_amd.ops as ops
    g, eng, args, pre, anames, n_act, qtable = _setup(golden, which, monkeypatch, exact)
    eps = nchw(eng.forward(*args))
    r = rel_l2(eps, T(g[pre + "eps_w4a8_attnq"]))
    eng.set_calibration("record", 0)
    eng.forward(*args)
    eng.set_calibration(None)
    flips, total, worst, arates, aties = 0, 0, 0.0, {}, {}
    for j, n in enumerate(anames):
        xe = eng.observed[n_act + j].float().contiguous()
        level = int(g[f"{pre}attn_q/{n}/level"])
        d, z = float(g[f"{pre}attn_q/{n}/delta"]), float(g[f"{pre}attn_q/{n}/zp"])
        be = torch.clamp(torch.round(xe.cpu() / d) + z, 0, level - 1)
        bo = T(g[f"{pre}attn_q/{n}/bins"]).float()
        be = _to_ref_layout(which, n[-1], be, bo.shape, eng)
        diff = (be - bo).abs()
        rate = float((diff > 0).float().mean())
        arates[n] = rate
        aties[n] = tie_distance(_to_ref_layout(which, n[-1], xe.cpu(), bo.shape, eng), d, diff > 0)
        worst = max(worst, rate)
        flips += int((diff > 0).sum())
        total += diff.numel()
    print(f"[{which}{' exact-fp32 mode' if exact else ''}] attention quantizers on: eps rel-L2 {r:.3e}; bins of the {len(anames)} attention "
          f"quantizer inputs moved {flips / total:.4%} overall, worst quantizer {worst:.3%}")
    assert torch.isfinite(eps).all()
    if exact:
        # The oracle with the same attention quantizers IS the reference on this fixture (tests/test_oracle_r03_fixtures.py); in its call
        # order the engine must be bit-identical in bins up to the first TIE (a value on a rounding boundary, decided by the fp32
        # summation order), and downstream of it stay within the reference's own avalanche (tests/_avalanche.py).
        sd, cfg, base, act_names, wq = eng._test_ctx
        owq = {n: {"delta": q.delta.reshape((-1,) + (1,) * (sd[n + ".weight"].dim() - 1)),
                   "zp": q.zp.reshape((-1,) + (1,) * (sd[n + ".weight"].dim() - 1)), "alpha": None} for n, q in wq.items()}
        kw = dict(wq=owq, aq={n: (qtable[0, i, 0], qtable[0, i, 1]) for i, n in enumerate(act_names)},
                  attn_aq={n: (T(g[f"{pre}attn_q/{n}/delta"]), T(g[f"{pre}attn_q/{n}/zp"]), int(g[f"{pre}attn_q/{n}/level"])) for n in anames})
        x, t = T(base["x"]), T(base["t"])
        if which == "ddim":
            fwd = lambda qs: O.ddim_unet_forward(sd, dict(cfg), x, t, qs)
        else:
            fwd = lambda qs: O.ldm_unet_forward(sd, dict(cfg), x, t.long(), T(base["ctx"]) if "ctx" in base.files else None, qs)
        clean = O.QuantSpec(**kw)
        clean.trace = {}
        with torch.no_grad():
            e_or = fwd(clean)
        same = torch.equal(e_or, T(g[pre + "eps_w4a8_attnq"]))      # bit-identical on the fixture's host; another CPU may avalanche
        print(f"[{which}] oracle on this host vs the fixture: {'bit-identical' if same else 'rel-L2 %.3e' % rel_l2(e_or, T(g[pre + 'eps_w4a8_attnq']))}")
        lrates, lties, _ = engine_bins_vs_trace(eng, args, qtable, act_names, clean.trace)
        rates = {n: (arates[n] if n in arates else lrates[n]) for n in clean.trace if n in arates or n in lrates}
        ties = {n: (aties[n] if n in aties else lties[n]) for n in rates}
        first, n_clean, dist = first_divergence(rates, ties)
        av = avalanche(fwd, kw, rel=1e-6)[1]
        print(f"[{which} exact-fp32 mode] {n_clean} of {len(rates)} quantizers (layers + attention) bit-identical before the first divergence "
              f"({first}, within {dist:.1e} bins of a rounding boundary); the reference under 1e-6 noise: eps moves "
              + ", ".join(f"{a:.2e}" for a, _ in av) + ", bins " + ", ".join(f"{b:.3f}" for _, b in av))
        assert n_clean >= 1 and dist <= 2e-3, (first, n_clean, dist)
        assert r <= max(5e-3, 1.5 * max(a for a, _ in av)) and flips / total <= max(5e-3, 1.5 * max(b for _, b in av))
    else:
        assert r <= 4e-2


def _to_ref_layout(which, role, be, ref_shape, eng):
    """engine tensors are [B, T, heads*d] (q, k, v) and [B, heads, Tq, Tk] (softmax); the reference's are listed per block type"""
    ref_shape = tuple(int(s) for s in ref_shape)
    if which == "ddim":
        # q [b, hw, c]; k, v [b, c, hw]; w [b, hw_k, hw_q] (softmax over keys, then permuted)
        if role == "q":
            return be.reshape(ref_shape)
        if role in "kv":
            return be.permute(0, 2, 1).reshape(ref_shape)
        return be.reshape(be.shape[0], be.shape[2], be.shape[3]).permute(0, 2, 1).reshape(ref_shape)
    heads = ref_shape[0] // be.shape[0]
    if which == "ldm":
        # cross_attn_forward: q, k, v '(b h) n d'; attn '(b h) i j'
        if role == "w":
            return be.reshape(ref_shape)
        B, Tn, C = be.shape
        return be.reshape(B, Tn, heads, C // heads).permute(0, 2, 1, 3).reshape(ref_shape)
    # QKVAttentionLegacy: q, k, v [(b h), ch, T]; weight [(b h), Tq, Tk]
    if role == "w":
        return be.reshape(ref_shape)
    B, Tn, C = be.shape
    return be.reshape(B, Tn, heads, C // heads).permute(0, 2, 3, 1).reshape(ref_shape)


@pytest.mark.parametrize("which", ["ddim", "ldm"])
def test_lazy_init_of_attention_quantizers_matches_reference(golden, monkeypatch, which):
    """MSE initialisation of q / k / v / softmax quantizers on the device (the engine's 'init' calibration restricted to them) in the
    exact mode: deltas within 2 % of the reference's, zero points within one bin, softmax zero point exactly 0."""
    g, eng, args, pre, anames, n_act, qtable = _setup(golden, which, monkeypatch, True, use_ref_attn_params=False)
    eng.calib_mask = set(range(n_act, n_act + len(anames)))
    eng.set_calibration("init", 0)
    eng.forward(*args)
    eng.set_calibration(None)
    eng.calib_mask = None
    rows = eng.qtable[0].cpu()
    rel = []
    for j, n in enumerate(anames):
        d, z = float(rows[n_act + j, 0]), float(rows[n_act + j, 1])
        rd, rz = float(g[f"{pre}attn_q/{n}/delta"]), float(g[f"{pre}attn_q/{n}/zp"])
        rel.append(abs(d - rd) / rd)
        assert abs(z - rz) <= 1.0, (n, z, rz)
        if n.endswith("_w"):
            assert z == 0.0
    print(f"[{which}] attention quantizer deltas vs the reference: median rel. error {np.median(rel):.2e}, max {max(rel):.2e}")
    assert np.median(rel) <= 5e-3 and max(rel) <= 2e-2
