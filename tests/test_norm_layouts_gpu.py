"""The sub-wave ("rows") layouts of LayerNorm and GroupNorm-apply against the layouts they replace, and against torch.

LayerNorm (ldm/modules/attention.py:196-198 `norm1/2/3` -> the next QuantLayer's activation quantizer, quant_layer.py:223-226):
`k_layernorm_hs` sums a row in a different order than the wave-per-row kernel, so the two agree to fp32 rounding and a bin may
move by one (bars: 2e-6 relative, <= 2e-4 of the bins, never by more than one); fp16 rows and their fp32 images stay bit-identical.
GroupNorm-apply: the per-element arithmetic is the same in both layouts -> bit-identical outputs."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import tfmq_dm_amd.ops as ops
    return ops


def _with_env(name, fn):
    os.environ[name] = "1"
    try:
        return fn()
    finally:
        del os.environ[name]


@pytest.mark.parametrize("rows,C", [(8 * 77 + 3, 320), (1000, 640), (130, 1280), (50, 64), (33, 448), (64, 1288)])
def test_layernorm_rows_layout(ops, rows, C):
    gen = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=gen) * 1.7 + 0.4).to(DEV)
    g, b = torch.randn(C, generator=gen).to(DEV), (torch.randn(C, generator=gen) * 0.3).to(DEV)
    ref = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-5)
    sel = ops.qsel(torch.tensor([[float(ref.abs().max()) * 2 / 255, 128.0]], device=DEV))
    for xin in (x, x.half()):
        r = F.layer_norm(xin.double(), (C,), g.double(), b.double(), 1e-5)
        q_new, f_new = ops.layernorm(xin, g, b, 1e-5, sel, want_f32=True)
        q_old, f_old = _with_env("TFMQ_LN_WAVE_PER_ROW", lambda: ops.layernorm(xin, g, b, 1e-5, sel, want_f32=True))
        scale = float(r.abs().max())
        assert float((f_new.double() - r).abs().max()) <= 2e-6 * scale
        assert float((f_new - f_old).abs().max()) <= 2e-6 * scale
        dq = (q_new.int() - q_old.int()).abs()
        assert int(dq.max()) <= 1 and float((dq > 0).float().mean()) <= 2e-4
        # the quantizer sees exactly the fp32 values the kernel also returns
        assert torch.equal(q_new, ops.quantize_act(f_new, sel))
    # fp16 rows == their fp32 images, bit for bit (both entry points use the same layout)
    qa, fa = ops.layernorm(x.half(), g, b, 1e-5, sel, want_f32=True)
    qb, fb = ops.layernorm(x.half().float(), g, b, 1e-5, sel, want_f32=True)
    assert torch.equal(qa, qb) and torch.equal(fa, fb)


@pytest.mark.parametrize("B,HW,C1,C2", [(3, 64, 320, 320), (2, 256, 640, 320), (2, 64, 1280, 1280), (5, 16, 64, 0), (2, 1024, 320, 0), (1, 64, 1920, 640)])
def test_groupnorm_apply_rows_layout_is_bit_identical(ops, B, HW, C1, C2):
    gen = torch.Generator().manual_seed(B * HW + C1)
    seg = 16

    def src(c):
        x = (torch.randn(B, HW, 1, c, generator=gen) * 1.5 + 0.2).to(DEV).half()
        xf = x.float().reshape(B * HW // seg, seg, c)
        x._tfmq_stats = (torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=-1).contiguous(), seg)
        return x
    x1 = src(C1)
    x2 = src(C2) if C2 else None
    Cc = C1 + C2
    g, b = torch.randn(Cc, generator=gen).to(DEV), (torch.randn(Cc, generator=gen) * 0.3).to(DEV)
    sel = ops.qsel(torch.tensor([[0.03, 9.0]], device=DEV))
    for kw in (dict(want_cat=True, half_out=True), dict(want_cat=True, half_out=True, want_f32=True), dict(want_f32=True), dict()):
        rows = _with_env("TFMQ_GN_APPLY_ROWS", lambda: ops.groupnorm(x1, g, b, 1e-5, True, sel, x2=x2, **kw))
        items = _with_env("TFMQ_GN_APPLY_ITEMS", lambda: ops.groupnorm(x1, g, b, 1e-5, True, sel, x2=x2, **kw))
        for a, c in zip(rows, items):
            assert (a is None) == (c is None)
            if a is not None:
                assert torch.equal(a, c), kw
    # and the values are GroupNorm + SiLU of the concatenated tensor (statistics from the epilogue sums)
    xc = torch.cat([x1] + ([x2] if x2 is not None else []), dim=-1).float()
    ref = F.silu(F.group_norm(xc.reshape(B, HW, Cc).permute(0, 2, 1), 32, g, b, 1e-5)).permute(0, 2, 1).reshape(B, HW, 1, Cc)
    _, yf, xcat = _with_env("TFMQ_GN_APPLY_ROWS", lambda: ops.groupnorm(x1, g, b, 1e-5, True, None, x2=x2, want_f32=True, want_cat=True))
    assert float((yf - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
    assert torch.equal(xcat, xc)
