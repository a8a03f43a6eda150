"""Round 5: a reconstruction iteration captured once as a hipGraph and replayed (engine/recon.py: _Unit._graph_iterate;
tfmq_adaround_bwd_adam_dyn reads the optimizer's per-iteration scalars from device memory) against the same iterations issued eagerly
(the default): same kernels in the same order, so the alphas and Adam moments must agree BIT FOR BIT after every iteration count -- the
reported loss values are sums of fp32 atomics and agree to rounding -- across the warm-up boundary where the rounding regulariser switches on (reference quant/reconstruction.py:63-78,
LossFunc / LinearTempDecay of reconstruction_util.py)."""
import pytest
import torch
import torch.nn.functional as F

import tfmq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def mk_layer(gen, cout, cin, k):
    w = torch.randn(cout, cin, k, k, generator=gen) * 0.05
    b = torch.randn(cout, generator=gen) * 0.1
    d, z = O.init_channelwise(w, 16, "minmax")
    return dict(w=w, b=b, delta=d, zp=z, alpha=O.adaround_init_alpha(w, d))


def _build(kind, R, iters):
    gen = torch.Generator().manual_seed(11)
    dev_layer = lambda L: R.AdaLayer(L["w"].to(DEV), L["delta"].to(DEV), L["zp"].to(DEV), L["b"].to(DEV), 16, alpha=L["alpha"].to(DEV))
    if kind == "layer":
        L = mk_layer(gen, 64, 32, 3)
        x = torch.randn(24, 32, 16, 16, generator=gen)
        y = F.conv2d(x, L["w"], L["b"], padding=1)
        dl = dev_layer(L)
        return R.LayerUnit(dl, nhwc(x), nhwc(y), pad=(1, 1, 1, 1), iters=iters, w=0.01, warmup=0.2), [dl], 24
    cin, cout = 64, 64
    c1, c2 = mk_layer(gen, cout, cin, 3), mk_layer(gen, cout, cout, 3)
    g1, b1 = torch.randn(cin, generator=gen), torch.randn(cin, generator=gen) * 0.2
    g2, b2 = torch.randn(cout, generator=gen), torch.randn(cout, generator=gen) * 0.2
    x = torch.randn(20, cin, 16, 16, generator=gen)
    proj = torch.randn(20, cout, generator=gen) * 0.3
    y = torch.randn(20, cout, 16, 16, generator=gen)
    d1, d2 = dev_layer(c1), dev_layer(c2)
    unit = R.ResnetUnit(d1, d2, (g1.to(DEV), b1.to(DEV)), (g2.to(DEV), b2.to(DEV)), None, nhwc(x), proj.to(DEV), nhwc(y),
                        iters=iters, w=0.01, warmup=0.2)
    return unit, [d1, d2], 20


def _run(kind, graph, monkeypatch, gemm):
    from tfmq_dm_amd.engine import recon as R
    monkeypatch.setenv("TFMQ_RECON_GRAPH", "1" if graph else "0")
    monkeypatch.setenv("TFMQ_RECON_GEMM", gemm)
    iters = 40
    unit, layers, n = _build(kind, R, iters)
    assert unit.graph_on == graph
    gen = torch.Generator().manual_seed(5)
    hist = []
    for _ in range(iters):
        idx = torch.randperm(n, generator=gen)[:8].to(DEV)
        rec, rl = unit.iterate(idx)
        hist.append((float(rec), float(rl)))
    assert (unit._graph is not None) == graph
    return hist, [(L.alpha.clone(), L.m.clone(), L.v.clone()) for L in layers]


@pytest.mark.parametrize("gemm", ["bf16x3", "f32"])
@pytest.mark.parametrize("kind", ["layer", "resnet"])
def test_replayed_iterations_equal_eager_iterations_bit_for_bit(kind, gemm, monkeypatch):
    h0, s0 = _run(kind, False, monkeypatch, gemm)
    h1, s1 = _run(kind, True, monkeypatch, gemm)
    for (r0, q0), (r1, q1) in zip(h0, h1):                      # reported losses: fp32 atomic sums, equal to rounding
        assert abs(r0 - r1) <= 1e-5 * abs(r0) and abs(q0 - q1) <= 1e-5 * abs(q0) + 1e-12
    assert any(q > 0 for _, q in h0[8:]) and all(q == 0 for _, q in h0[:7])      # the regulariser switched on at 20 % of the iterations
    for (a0, m0, v0), (a1, m1, v1) in zip(s0, s1):
        assert torch.equal(a0, a1) and torch.equal(m0, m1) and torch.equal(v0, v1)


def test_device_scalars_are_the_librarys_own(monkeypatch):
    """tfmq_adaround_scalars returns what tfmq_adaround_bwd_adam computes internally; one step through each entry point from equal states."""
    import tfmq_dm_amd.ops as ops
    gen = torch.Generator().manual_seed(2)
    w = (torch.randn(48, 96, generator=gen) * 0.05).to(DEV)
    d, z = O.init_channelwise(w.cpu(), 16, "minmax")
    d, z = d.reshape(-1).to(DEV), z.reshape(-1).to(DEV)
    g = torch.randn(48, 96, generator=gen).to(DEV)
    for t, b in ((1, 0.0), (7, 20.0), (1234, 7.5)):
        st = [ops.adaround_init(w, d) for _ in range(2)]
        mv = [[torch.rand_like(w) * 1e-3, torch.rand_like(w) * 1e-6] for _ in range(1)]
        mv.append([mv[0][0].clone(), mv[0][1].clone()])
        rl = [torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)]
        ops.adaround_bwd_adam(w, st[0], d, z, g, mv[0][0], mv[0][1], 16, 0.01, b, 1e-3, t, rl[0])
        sc = torch.tensor(ops.adaround_scalars(0.01, b, 1e-3, t), dtype=torch.float32, device=DEV)
        ops.adaround_bwd_adam_dyn(w, st[1], d, z, g, mv[1][0], mv[1][1], 16, sc, rl[1])
        assert torch.equal(st[0], st[1]) and torch.equal(mv[0][0], mv[1][0]) and torch.equal(mv[0][1], mv[1][1])
        assert abs(float(rl[0]) - float(rl[1])) <= 1e-5 * abs(float(rl[0])) + 1e-12      # (the rounding loss is a sum of fp32 atomics: equal to rounding)
