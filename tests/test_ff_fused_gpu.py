"""tfmq_ff_fused (round 4): `x = self.ff(self.norm3(x)) + x` of a BasicTransformerBlock (ldm/modules/attention.py:37-64, 152-215;
QuantBasicTransformerBlock quant/quant_block.py:248-299) as ONE launch -- LayerNorm -> quantise -> ff.net.0.proj -> value * gelu(gate)
-> quantise -> ff.net.2 (+ x) with a token per lane, the GEGLU bins going from the accumulators straight into the second GEMM's MFMA
operand.  Bar: BIT-IDENTICAL to the three launches it replaces (tfmq_layernorm_h, tfmq_conv2d_w4a8 with TFMQ_OUT_GEGLU_Q8_FAST,
tfmq_conv2d_w4a8 with the fp16 residual), for the fp16 output and for the consumer quantizer's int8 bins, ragged token counts
included; and within the GEGLU bar (bins within 1, < 2e-3 moved after the first quantizer) of the oracle's fp32 arithmetic."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import tfmq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import tfmq_dm_amd.ops as ops
    return ops


def _layers(ops, C, inner, seed, nsteps=1):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(5000, C, generator=g) * 1.7 + 0.3)
    gamma, beta = torch.randn(C, generator=g) * 0.5 + 1.0, torch.randn(C, generator=g) * 0.2
    w1 = torch.randn(2 * inner, C, 1, 1, generator=g) * (2.0 / C ** 0.5)
    b1 = torch.randn(2 * inner, generator=g) * 0.2
    w2 = torch.randn(C, inner, 1, 1, generator=g) * (1.5 / inner ** 0.5)
    b2 = torch.randn(C, generator=g) * 0.2
    wd1, wz1 = O.init_channelwise(w1, 16, "minmax")
    wd2, wz2 = O.init_channelwise(w2, 16, "minmax")
    ln = F.layer_norm(x.half().float(), (C,), gamma, beta, 1e-5)
    ad0, az0 = O.minmax(ln, 256)
    hq = F.conv2d(O.fake_quant(ln.t().reshape(1, C, -1, 1), ad0, az0, 256), O.fake_quant(w1, wd1, wz1, 16), b1)
    a, gt = hq.reshape(2 * inner, -1).t().chunk(2, dim=-1)
    gg = a * F.gelu(gt)
    ad2, az2 = O.minmax(gg, 256)
    y = F.conv2d(O.fake_quant(gg.t().reshape(1, inner, -1, 1), ad2, az2, 256), O.fake_quant(w2, wd2, wz2, 16), b2).reshape(C, -1).t() + x.half().float()
    od, oz = O.minmax(y, 256)
    # quantizer table: three activation quantizers per step row (ff0, ff2, consumer); later rows perturbed (Finite-Set rows)
    rows = []
    for k in range(nsteps):
        f = 1.0 + 0.07 * k
        rows.append([[float(ad0) * f, float(az0)], [float(ad2) * f, float(az2)], [float(od) * f, float(oz)]])
    qt = torch.tensor(rows, dtype=torch.float32, device=DEV)
    perm = ops.geglu_perm(inner)
    pw1 = ops.pack_w4(w1[perm].contiguous().to(DEV), wd1.reshape(-1)[perm].contiguous().to(DEV), wz1.reshape(-1)[perm].contiguous().to(DEV),
                      bias=b1[perm].contiguous().to(DEV))
    pw2 = ops.pack_w4(w2.to(DEV), wd2.to(DEV), wz2.to(DEV), bias=b2.to(DEV))
    ref = dict(x=x, gamma=gamma, beta=beta, w1=w1, b1=b1, wd1=wd1, wz1=wz1, w2=w2, b2=b2, wd2=wd2, wz2=wz2,
               ad0=ad0, az0=az0, ad2=ad2, az2=az2, od=od, oz=oz, y=y)
    return qt, pw1, pw2, ref


def _chain(ops, x16, gamma, beta, sel0, pw1, sel2, pw2, out_q8):
    M, C = x16.shape
    xq = ops.layernorm(x16, gamma, beta, 1e-5, sel0)[0]
    g = ops.conv2d_w4a8(xq.reshape(1, M, 1, C), pw1, sel0, geglu_oq=sel2)
    kw = dict(out_q8=out_q8) if out_q8 is not None else dict(out_f16=True)
    return ops.conv2d_w4a8(g, pw2, sel2, residual=x16.reshape(1, M, 1, C), want_stats=False, **kw).reshape(M, C)


@pytest.mark.parametrize("M,q8", [(4096, False), (4096, True), (256 * 3 + 77, False), (31, True), (1, False), (5000, True)])
def test_ff_fused_equals_the_three_launches(ops, M, q8):
    C, inner = 320, 1280
    qt, pw1, pw2, ref = _layers(ops, C, inner, 5)
    sel0, sel2, selo = ops.qsel(qt, 0), ops.qsel(qt, 1), ops.qsel(qt, 2)
    x16 = ref["x"][:M].half().to(DEV).contiguous()
    gamma, beta = ref["gamma"].to(DEV), ref["beta"].to(DEV)
    assert ops.ff_fused_ok(C, inner, pw1, pw2)
    want = _chain(ops, x16, gamma, beta, sel0, pw1, sel2, pw2, selo if q8 else None)
    got = ops.ff_fused(x16, gamma, beta, 1e-5, sel0, pw1, sel2, pw2, out_q8=selo if q8 else None)
    assert got.dtype == want.dtype and got.shape == want.shape
    assert torch.equal(got, want)
    assert torch.equal(got, ops.ff_fused(x16, gamma, beta, 1e-5, sel0, pw1, sel2, pw2, out_q8=selo if q8 else None))      # run to run
    # against the oracle's fp32 arithmetic (erf GELU): the fp16 row within fp16 rounding + a few moved bins of the two inner quantizers
    yr = ref["y"][:M]
    if q8:
        bins = O.quant_index(yr, ref["od"], ref["oz"], 256)
        d = (got.cpu().float() + 128 - bins).abs()
        assert float(d.max()) <= 2 and float((d > 0).float().mean()) < 3e-2
    else:
        assert float((got.cpu().float() - yr).norm() / yr.norm()) < 5e-3


def test_ff_fused_follows_the_finite_set_step(ops):
    """The activation deltas come from the CURRENT Finite-Set row (device step counter): the folded GEGLU constants are rebuilt per call."""
    C, inner, M = 320, 1280, 777
    qt, pw1, pw2, ref = _layers(ops, C, inner, 9, nsteps=3)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    sel0, sel2 = ops.qsel(qt, 0, step), ops.qsel(qt, 1, step)
    x16 = ref["x"][:M].half().to(DEV).contiguous()
    gamma, beta = ref["gamma"].to(DEV), ref["beta"].to(DEV)
    outs = []
    for k in range(3):
        step.fill_(k)
        want = _chain(ops, x16, gamma, beta, sel0, pw1, sel2, pw2, None)
        got = ops.ff_fused(x16, gamma, beta, 1e-5, sel0, pw1, sel2, pw2)
        assert torch.equal(got, want)
        outs.append(got.clone())
    assert not torch.equal(outs[0], outs[2])


def test_ff_fused_refuses_other_widths(ops):
    from tfmq_dm_amd._lib import TfmqError
    qt, pw1, pw2, ref = _layers(ops, 128, 256, 3)
    assert not ops.ff_fused_ok(128, 256, pw1, pw2)
    with pytest.raises(TfmqError):
        ops.ff_fused(ref["x"][:64].half().to(DEV), ref["gamma"].to(DEV), ref["beta"].to(DEV), 1e-5, ops.qsel(qt, 0), pw1, ops.qsel(qt, 1), pw2)


@pytest.mark.parametrize("M,with_pre,with_post", [(512, True, False), (512, False, True), (4096 * 2, True, True), (256, True, True)])
def test_ff_fused_with_the_linears_in_front_and_behind(ops, M, with_pre, with_post):
    """attn2.to_out (+ residual) in front and proj_out (+ the SpatialTransformer's input, + the next GroupNorm's statistics) behind the
    feed-forward, in the same launch (ldm/modules/attention.py:194, :213, :259-261): every stored tensor and the statistics equal the
    separate launches bit for bit."""
    C, inner, T = 320, 1280, 256
    qt, pw1, pw2, ref = _layers(ops, C, inner, 21)
    g = torch.Generator().manual_seed(33 + M)
    qt2 = torch.tensor([[[0.04, 119.0]]], dtype=torch.float32, device=DEV)
    sel0, sel2, selo, selp = ops.qsel(qt, 0), ops.qsel(qt, 1), ops.qsel(qt, 2), ops.qsel(qt2, 0)
    gamma, beta = ref["gamma"].to(DEV), ref["beta"].to(DEV)

    def lin(bias=True):
        w = torch.randn(C, C, 1, 1, generator=g) * (2.0 / C ** 0.5)
        wd, wz = O.init_channelwise(w, 16, "minmax")
        return ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=(torch.randn(C, generator=g) * 0.2).to(DEV) if bias else None)
    to_out2, proj_out = lin(), lin()
    xres = (torch.randn(M, C, generator=g) * 1.4).half().to(DEV)           # the stream before attn2's residual add
    x_in = (torch.randn(M, C, generator=g) * 1.1).half().to(DEV)           # the SpatialTransformer's input
    o2 = torch.randint(-128, 128, (M, C), generator=g, dtype=torch.int8).to(DEV)
    # the separate launches
    if with_pre:
        x2 = ops.conv2d_w4a8(o2.reshape(1, M, 1, C), to_out2, selp, residual=xres.reshape(1, M, 1, C), out_f16=True, want_stats=False).reshape(M, C)
    else:
        x2 = xres
    if with_post:
        bins = _chain(ops, x2, gamma, beta, sel0, pw1, sel2, pw2, selo)
        want = ops.conv2d_w4a8(bins.reshape(M // T, T, 1, C), proj_out, selo, residual=x_in.reshape(M // T, T, 1, C), out_f16=True, want_stats=True)
        want_stats = want._tfmq_stats
        want = want.reshape(M, C)
    else:
        want = _chain(ops, x2, gamma, beta, sel0, pw1, sel2, pw2, None)
    # one launch
    pre = dict(xq=o2, pw=to_out2, aq=selp, residual=xres) if with_pre else None
    post = dict(pw=proj_out, residual=x_in, stats=True, hw=T) if with_post else None
    got = ops.ff_fused(None if with_pre else x2, gamma, beta, 1e-5, sel0, pw1, sel2, pw2, out_q8=selo if with_post else None, pre=pre, post=post)
    if with_pre:
        assert torch.equal(got[0], x2)
        got = got[1]
    assert got.dtype == want.dtype and torch.equal(got, want)
    if with_post:
        assert got._tfmq_stats[1] == want_stats[1] and torch.equal(got._tfmq_stats[0], want_stats[0])
