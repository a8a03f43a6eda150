"""tfmq_row_chain (round 4): the token Linears around the attention of a BasicTransformerBlock chained over resident token tiles in ONE
launch, a token per lane (csrc/row_chain.hip).

  pre   SpatialTransformer.norm (GroupNorm affine + quantise) -> proj_in -> h -> norm1 (LayerNorm + quantise) -> fused to_q | to_k | to_v
        (q | k fp16 rows, v transposed)                     ldm/modules/attention.py:238-261, :212, :168-177
  mid   attention bins -> attn1.to_out + x -> norm2 -> attn2.to_q        ldm/modules/attention.py:194, :212-213
under the QuantLayers of quant/quant_layer.py:306-340 / quant/quant_block.py:178-299.

Bar: BIT-IDENTICAL to the launches it replaces (groupnorm-from-statistics apply pass, tfmq_conv2d_w4a8 with fp16 output / residual /
transposed region, tfmq_layernorm_h), every stored tensor compared."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import tfmq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import tfmq_dm_amd.ops as ops
    return ops


def _lin(ops, g, cout, cin, bias=True, scale=1.0):
    w = torch.randn(cout, cin, 1, 1, generator=g) * (scale * 2.0 / cin ** 0.5)
    b = torch.randn(cout, generator=g) * 0.2 if bias else None
    wd, wz = O.init_channelwise(w, 16, "minmax")
    return ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=None if b is None else b.to(DEV))


@pytest.mark.parametrize("B,T,C", [(2, 256, 320), (1, 4096, 320), (3, 1024, 320), (2, 128, 640), (3, 1024, 640), (1, 256, 640)])
def test_pre_chain_equals_the_four_launches(ops, B, T, C):
    """C = 640 (the 32 x 32 level): two waves share a 32-token group -- one output tile each, half of K per phase, the LayerNorm's row
    statistics cross the pair through LDS in k_layernorm_hs<16>'s summation order."""
    g = torch.Generator().manual_seed(100 + T + C)
    # the stream tensor with its producer's GroupNorm statistics: the fp16 output of a w4a8 pointwise layer
    xin = torch.randn(B, T, 1, 64, generator=g)
    ad, az = O.minmax(xin, 256)
    qt = torch.tensor([[[float(ad), float(az)], [0.031, 131.0], [0.027, 125.0], [0.05, 120.0]]], dtype=torch.float32, device=DEV)
    sel = [ops.qsel(qt, i) for i in range(4)]
    prod = _lin(ops, g, C, 64)
    x = ops.conv2d_w4a8(ops.quantize_act(xin.to(DEV), sel[0]), prod, sel[0], out_f16=True, want_stats=True)     # [B, T, 1, C] fp16 + stats
    assert getattr(x, "_tfmq_stats", None) is not None
    gn_g, gn_b = (torch.randn(C, generator=g) * 0.4 + 1).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    ln_g, ln_b = (torch.randn(C, generator=g) * 0.4 + 1).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    pin, qkv = _lin(ops, g, C, C), _lin(ops, g, 3 * C, C, bias=False)
    # the launches
    xq, _, _ = ops.groupnorm(x, gn_g, gn_b, 1e-6, False, sel[1])
    h = ops.conv2d_w4a8(xq, pin, sel[1], out_f16=True)
    hq = ops.layernorm(h.reshape(B, T, C), ln_g, ln_b, 1e-5, sel[2])[0]
    y16, vt = ops.conv2d_w4a8(hq.reshape(B, T, 1, C), qkv, sel[2], out_f16=True, t_col0=2 * C)
    # the chain
    ab = ops.gn_affine_from_stats(x, gn_g, gn_b, 1e-6)
    assert ab is not None and ops.row_chain_supported(C, B * T, T, True)
    outs = ops.row_chain(x.reshape(B * T, C), T, [dict(pw=pin, aq=sel[1], ln=True), dict(pw=qkv, aq=sel[2], t_col0=2 * C)], gn=ab, ln=(ln_g, ln_b, 1e-5))
    assert torch.equal(outs[0][0], h.reshape(B * T, C))
    assert torch.equal(outs[1][0][:, :2 * C], y16.reshape(B * T, 3 * C)[:, :2 * C])
    assert torch.equal(outs[1][1], vt)
    again = ops.row_chain(x.reshape(B * T, C), T, [dict(pw=pin, aq=sel[1], ln=True), dict(pw=qkv, aq=sel[2], t_col0=2 * C)], gn=ab, ln=(ln_g, ln_b, 1e-5))
    assert torch.equal(again[1][1], vt) and torch.equal(again[0][0], outs[0][0])


@pytest.mark.parametrize("M,C", [(256, 320), (4096 * 2, 320), (1024 * 3, 320), (128, 640), (1024 * 5, 640)])
def test_mid_chain_equals_the_three_launches(ops, M, C):
    g = torch.Generator().manual_seed(7 + M + C)
    qt = torch.tensor([[[0.04, 117.0], [0.033, 129.0]]], dtype=torch.float32, device=DEV)
    sel = [ops.qsel(qt, i) for i in range(2)]
    o = torch.randint(-128, 128, (M, C), generator=g, dtype=torch.int8).to(DEV)          # the attention kernel's output bins
    hres = (torch.randn(M, C, generator=g) * 1.5).half().to(DEV)
    ln_g, ln_b = (torch.randn(C, generator=g) * 0.4 + 1).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    to_out, to_q = _lin(ops, g, C, C), _lin(ops, g, C, C, bias=False)
    x1 = ops.conv2d_w4a8(o.reshape(1, M, 1, C), to_out, sel[0], residual=hres.reshape(1, M, 1, C), out_f16=True, want_stats=False).reshape(M, C)
    xq = ops.layernorm(x1, ln_g, ln_b, 1e-5, sel[1])[0]
    q16 = ops.conv2d_w4a8(xq.reshape(1, M, 1, C), to_q, sel[1], out_f16=True).reshape(M, C)
    outs = ops.row_chain(o, M, [dict(pw=to_out, aq=sel[0], residual=hres, ln=True), dict(pw=to_q, aq=sel[1])], ln=(ln_g, ln_b, 1e-5))
    assert torch.equal(outs[0][0], x1)
    assert torch.equal(outs[1][0], q16)


def test_row_chain_refuses_what_it_cannot_take(ops):
    from tfmq_dm_amd._lib import TfmqError
    C = 320
    assert not ops.row_chain_ok(C, 255, 255, False) and not ops.row_chain_ok(1280, 1024, 1024, False) and not ops.row_chain_ok(C, 256, 64, True)
    assert ops.row_chain_supported(640, 128, 128, True) and not ops.row_chain_supported(640, 192, 192, False)
    assert not ops.row_chain_ok(640, 128, 128, True)          # (the engine's policy: C = 640 measured not faster than the launches)
    with pytest.raises(TfmqError):
        ops.row_chain(torch.zeros(100, C, dtype=torch.int8, device=DEV), 100, [dict(pw=None, aq=None)])
