"""TFMQ_OUT_GEGLU_Q8_FAST (round 4): the GEGLU epilogue with a GELU sized for its consumer.

x, gate = proj(x).chunk(2); x * gelu(gate) (ldm/modules/attention.py:52-59) is rounded to one of 256 activation bins by the next
QuantLayer's quantizer (quant/quant_layer.py:223-226) in the same epilogue.  The fast mode evaluates Phi(g) as a logistic of an
odd quintic (|g Phi(g) - gelu(g)| <= 2.8e-5), folds the output delta into the value's scale and the zero-point corrections into
the biases.  Bar (the one the exact epilogue carries against the oracle): bins within 1, fewer than 2e-3 of them moved --
against the exact epilogue (TFMQ_OUT_GEGLU_Q8) and against the oracle's fp32 arithmetic with torch's erf GELU."""
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


def qtab(delta, zp):
    return torch.tensor([[float(delta), float(zp)]], dtype=torch.float32, device=DEV)


@pytest.mark.parametrize("B,T,cin,inner,gain", [(2, 200, 128, 256, 1.0), (1, 4096, 320, 1280, 1.0), (2, 77, 64, 64, 1.0),
                                                (3, 1024, 640, 2560, 1.0), (1, 333, 320, 1280, 6.0)])
def test_fast_geglu_epilogue_within_one_bin(ops, B, T, cin, inner, gain):
    """gain = 6: gate pre-activations far beyond the polynomial's clamp (|g| up to ~40) -- the logistic must saturate, not turn over."""
    gen = torch.Generator().manual_seed(11 + inner + int(gain))
    x = torch.randn(B, T, 1, cin, generator=gen) * 1.3 - 0.2
    w = torch.randn(2 * inner, cin, 1, 1, generator=gen) * (2.0 * gain / cin ** 0.5)
    b = torch.randn(2 * inner, generator=gen) * 0.2
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    xq = ops.quantize_act(x.to(DEV), sel)
    ref_h = F.conv2d(O.fake_quant(x.permute(0, 3, 1, 2), ad, az, 256), O.fake_quant(w, wd, wz, 16), b)
    ref_h = ref_h.permute(0, 2, 3, 1).reshape(B, T, 2 * inner)
    a, g = ref_h.chunk(2, dim=-1)
    ref = a * F.gelu(g)
    od, oz = O.minmax(ref, 256)
    osel = ops.qsel(qtab(od, oz))
    perm = ops.geglu_perm(inner)
    pwp = ops.pack_w4(w[perm].contiguous().to(DEV), wd.reshape(-1)[perm].contiguous().to(DEV),
                      wz.reshape(-1)[perm].contiguous().to(DEV), bias=b[perm].contiguous().to(DEV))
    exact = ops.conv2d_w4a8(xq, pwp, sel, geglu_oq=osel, geglu_exact=True)
    fast = ops.conv2d_w4a8(xq, pwp, sel, geglu_oq=osel)
    assert fast.dtype == torch.int8 and fast.shape == exact.shape == (B, T, 1, inner)
    d = (fast.int() - exact.int()).abs()
    assert int(d.max()) <= 1
    assert float((d > 0).float().mean()) < 2e-3
    ref_bins = O.quant_index(ref, od, oz, 256)
    d2 = (fast.reshape(B, T, inner).cpu().float() + 128 - ref_bins).abs()
    assert float(d2.max()) <= 1 and float((d2 > 0).float().mean()) < 2e-3
    # run-to-run identical, and the environment switch restores the exact epilogue
    assert torch.equal(fast, ops.conv2d_w4a8(xq, pwp, sel, geglu_oq=osel))
    os.environ["TFMQ_GELU_EXACT"] = "1"
    try:
        assert torch.equal(exact, ops.conv2d_w4a8(xq, pwp, sel, geglu_oq=osel))
    finally:
        del os.environ["TFMQ_GELU_EXACT"]


@pytest.mark.parametrize("B,T,cin,inner", [(2, 1024, 640, 2560), (1, 700, 320, 1280), (3, 256, 128, 192)])
def test_fast_geglu_on_256_row_tiles_is_bit_identical(ops, B, T, cin, inner):
    """Round 6: the register-direct pointwise kernel on 256 x 128 tiles (TFMQ_TILE_DIRECT256, two blocks per CU) carries the same
    consumer-sized GEGLU epilogue: the same int32 sums, the same arithmetic -- the same bins as the 128-row form (ragged last row tile included)."""
    import tfmq_dm_amd.ops as _o
    gen = torch.Generator().manual_seed(5 + inner)
    x = torch.randn(B, T, 1, cin, generator=gen) * 1.3 - 0.2
    w = torch.randn(2 * inner, cin, 1, 1, generator=gen) * (2.0 / cin ** 0.5)
    b = torch.randn(2 * inner, generator=gen) * 0.2
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    xq = ops.quantize_act(x.to(DEV), sel)
    osel = ops.qsel(qtab(0.04, 131.0))
    perm = ops.geglu_perm(inner)
    pwp = ops.pack_w4(w[perm].contiguous().to(DEV), wd.reshape(-1)[perm].contiguous().to(DEV),
                      wz.reshape(-1)[perm].contiguous().to(DEV), bias=b[perm].contiguous().to(DEV))
    outs = []
    for tile in (6, 9):
        ops.set_conv_autotune({})
        orig = _o._tune_conv
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            outs.append(ops.conv2d_w4a8(xq, pwp, sel, geglu_oq=osel).clone())
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
    assert torch.equal(outs[0], outs[1])
