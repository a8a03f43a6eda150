"""Exact-fp32 fused attention of the reconstruction units (tfmq_attention_f32_fwd / _bwd) against float64 torch:
forward output and log-sum-exp, backward dQ / dK / dV (what autograd gives the reference's einsum / softmax / einsum,
quant_block.py:226-243)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import tfmq_dm_amd.ops as ops
    return ops


def ref_attn(q, k, v, heads, scale):
    B, T, C = q.shape
    L, d = k.shape[1], C // heads
    qh, kh, vh = (t.double().reshape(B, -1, heads, d).permute(0, 2, 1, 3) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * scale
    p = torch.softmax(s, -1)
    o = (p @ vh).permute(0, 2, 1, 3).reshape(B, T, C)
    lse2 = torch.logsumexp(s, -1) * 1.4426950408889634
    return o, lse2


@pytest.mark.parametrize("B,heads,T,L,d", [(2, 8, 256, 256, 40), (1, 2, 128, 96, 64), (2, 3, 64, 160, 32), (1, 8, 1024, 1024, 40),
                                             (2, 8, 128, 77, 40), (1, 2, 96, 200, 80), (2, 1, 64, 5, 32)])
def test_forward_vs_float64(ops, B, heads, T, L, d):
    g = torch.Generator().manual_seed(T + d)
    C = heads * d
    q, k, v = (torch.randn(B, n, C, generator=g) for n in (T, L, L))
    scale = d ** -0.5
    o, lse = ops.attention_f32_fwd(q.to(DEV), k.to(DEV), v.to(DEV), heads, scale)
    ro, rl = ref_attn(q, k, v, heads, scale)
    assert float((o.cpu().double() - ro).abs().max()) <= 2e-6 * max(1.0, float(ro.abs().max()))
    assert float((lse.cpu().double() - rl).abs().max()) <= 2e-5


@pytest.mark.parametrize("B,heads,T,L,d", [(2, 4, 128, 128, 40), (1, 2, 96, 160, 64), (2, 3, 64, 64, 32), (1, 8, 512, 512, 40),
                                             (2, 8, 128, 77, 40), (1, 2, 96, 200, 80), (2, 1, 64, 5, 32)])
def test_backward_vs_float64_autograd(ops, B, heads, T, L, d):
    g = torch.Generator().manual_seed(7 * T + d)
    C = heads * d
    q, k, v = (torch.randn(B, n, C, generator=g) for n in (T, L, L))
    go = torch.randn(B, T, C, generator=g)
    scale = d ** -0.5
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.reshape(B, -1, heads, d).permute(0, 2, 1, 3) for t in (q64, k64, v64))
    o64 = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).permute(0, 2, 1, 3).reshape(B, T, C)
    o64.backward(go.double())
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = ops.attention_f32_fwd(qd, kd, vd, heads, scale)
    dq, dk, dv = ops.attention_f32_bwd(qd, kd, vd, o, lse, go.to(DEV), heads, scale)
    dq2, dk2, dv2 = ops.attention_f32_bwd(qd, kd, vd, o, lse, go.to(DEV), heads, scale)
    for mine, again, ref in ((dq, dq2, q64.grad), (dk, dk2, k64.grad), (dv, dv2, v64.grad)):
        assert torch.equal(mine, again)                                   # no atomics: run-to-run identical
        assert float((mine.cpu().double() - ref).abs().max()) <= 5e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("B,heads,T,L,d", [(2, 8, 256, 256, 40), (1, 2, 128, 96, 64), (2, 3, 64, 160, 32), (1, 8, 1024, 1024, 40),
                                             (2, 8, 128, 77, 40), (1, 2, 96, 200, 80), (2, 1, 64, 5, 32)])
def test_bf16x3_operand_form_vs_float64(ops, B, heads, T, L, d):
    """The same kernels' bf16x3 operand form (tfmq_set_gemm_precision(1): the default of the reconstruction iterations): every fp32
    operand split hi + lo in bf16, three MFMAs per product, fp32 accumulation -- 2^-16 per product.  Forward and backward against
    float64 at 5e-5 of the largest element (the exact-fp32 form: 2e-6 / 5e-6), run-to-run identical."""
    g = torch.Generator().manual_seed(11 * T + d)
    C = heads * d
    q, k, v = (torch.randn(B, n, C, generator=g) for n in (T, L, L))
    go = torch.randn(B, T, C, generator=g)
    scale = d ** -0.5
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.reshape(B, -1, heads, d).permute(0, 2, 1, 3) for t in (q64, k64, v64))
    o64 = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).permute(0, 2, 1, 3).reshape(B, T, C)
    o64.backward(go.double())
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    with ops.gemm_precision("bf16x3", 0):
        o, lse = ops.attention_f32_fwd(qd, kd, vd, heads, scale)
        dq, dk, dv = ops.attention_f32_bwd(qd, kd, vd, o, lse, go.to(DEV), heads, scale)
        o2, _ = ops.attention_f32_fwd(qd, kd, vd, heads, scale)
        dq2, dk2, dv2 = ops.attention_f32_bwd(qd, kd, vd, o, lse, go.to(DEV), heads, scale)
    oe, _ = ops.attention_f32_fwd(qd, kd, vd, heads, scale)           # exact form outside the context
    ro, rl = ref_attn(q, k, v, heads, scale)
    err = float((o.cpu().double() - ro).abs().max()) / max(1.0, float(ro.abs().max()))
    print(f"[bf16x3 d={d} T={T} L={L}] forward error {err:.2e} (exact form {float((oe.cpu().double() - ro).abs().max()):.2e}), "
          f"lse {float((lse.cpu().double() - rl).abs().max()):.2e}")
    assert torch.equal(o, o2) and err <= 5e-5 and float((lse.cpu().double() - rl).abs().max()) <= 1e-4
    for name, mine, again, ref in (("dq", dq, dq2, q64.grad), ("dk", dk, dk2, k64.grad), ("dv", dv, dv2, v64.grad)):
        assert torch.equal(mine, again)
        e = float((mine.cpu().double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        print(f"    {name} error {e:.2e}")
        assert e <= 5e-5, (name, e)


def test_transformer_unit_fused_vs_gemm_path(ops):
    """One AdaRound iteration of a TransformerUnit (heads of 40 channels, 64 tokens) with the fused fp32 attention and
    with the GEMM path: same reconstruction loss and weight gradients (both exact fp32, different summation orders)."""
    from tfmq_dm_amd.engine import recon as R
    gen = torch.Generator().manual_seed(12)
    B, T, C, heads, L, Dc = 4, 64, 80, 2, 77, 48

    def ada(cout, cin, bias=True):
        w = (torch.randn(cout, cin, generator=gen) * 0.08).to(DEV)
        qp = ops.minmax_to_qparam(ops.minmax(w, cout), 16)
        return R.AdaLayer(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), torch.zeros(cout, device=DEV) if bias else None)
    x, y, ctx = (torch.randn(*s, generator=gen).to(DEV) for s in ((B, T, C), (B, T, C), (B, L, Dc)))
    gn = (torch.ones(C, device=DEV), torch.zeros(C, device=DEV))
    outs = []
    for flash in (True, False):
        gen.manual_seed(13)
        layers = [ada(C, C, False), ada(C, C, False), ada(C, C, False), ada(C, C), ada(8 * C, C), ada(C, 4 * C),
                  ada(C, C, False), ada(C, Dc, False), ada(C, Dc, False), ada(C, C)]
        u = R.TransformerUnit(layers, [gn, gn, gn], heads, x, ctx, y, iters=10)
        u.use_flash = flash
        rec, grads = u._forward_backward(torch.arange(B, device=DEV))
        outs.append((float(rec), [g.clone() for g in grads]))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * abs(outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert float((a - b).abs().max()) <= 2e-5 * max(1e-6, float(b.abs().max()))
