"""tfmq_attention_q8 (round 4): the attention of a block with LIVE matmul quantizers on the int8 matrix cores.

  reference  quant/quant_block.py:226-243 (cross_attn_forward), :483-500 (QuantAttnBlock.forward), :318-323 / :350-351 (QuantQKMatMul /
             QuantSMVMatMul): sim = aq_q(q) aq_k(k)^T scale, attn = softmax(sim), out = aq_w(attn) aq_v(v)

Bars
  * against an integer restatement in float64 (bins from the library's own quantizer, exact integer products, float64 softmax): every
    output row whose softmax values all sit away from a rounding boundary of aq_w is BIT-IDENTICAL (the kernel's sums are exact int32,
    its output delta_w delta_v * integer in fp32); the other rows differ by at most a few bins' worth;
  * against ops.attention_quant (the functional path the F21 fixture was pinned with): the same up to boundary values;
  * the engines with TFMQ_ATTN_Q8=1 on F21's models: eps within the functional path's own distance of the reference."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import tfmq_dm_amd.ops as ops
    return ops


def _case(ops, B, heads, Tq, Tk, d, w_level, seed, sharp=1.0):
    g = torch.Generator().manual_seed(seed)
    C = heads * d
    q, k, v = (torch.randn(B, T_, C, generator=g) * s for T_, s in ((Tq, sharp), (Tk, 1.0), (Tk, 1.3)))
    qt = torch.tensor([[[0.031 * sharp, 131.0], [0.027, 125.0], [0.05, 120.0], [1.0 / (w_level - 1), 0.0]]], dtype=torch.float32, device=DEV)
    sel = [ops.qsel(qt, i) for i in range(4)]
    return q.to(DEV), k.to(DEV), v.to(DEV), qt, sel


def _int_reference(ops, q, k, v, heads, scale, qt, sel, w_level, oracle_bins=False):
    """float64 restatement over the input bins -> (out fp32 [B, Tq, C], rows with a value near a boundary [B, heads, Tq]).  The bins of q, k, v
    come from the library's own quantizer, or (oracle_bins) from the oracle's restatement of UniformAffineQuantizer on the host
    (oracle/tfmq_oracle.py: quant_index, reference quant/quant_layer.py:225) -- the form that pins the kernel without trusting the library."""
    B, Tq, C = q.shape
    d = C // heads
    if oracle_bins:
        import tfmq_oracle as O
        bins = [O.quant_index(x.cpu(), qt[0, i, 0].cpu(), qt[0, i, 1].cpu(), 256).to(torch.int32).to(DEV) for i, x in enumerate((q, k, v))]
    else:
        bins = [ops.quantize_act(x.contiguous(), s).to(torch.int32) + 128 for x, s in zip((q, k, v), sel[:3])]
    (dq, zq), (dk, zk), (dv, zv), (dw, _) = [(float(qt[0, i, 0]), float(qt[0, i, 1])) for i in range(4)]
    iq = (bins[0].double() - zq).reshape(B, Tq, heads, d).permute(0, 2, 1, 3)
    ik = (bins[1].double() - zk).reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    iv = (bins[2].double() - zv).reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    s_int = iq @ ik.transpose(-1, -2)                                                        # exact (|.| < 2^53)
    c2 = torch.tensor(dq, dtype=torch.float32) * torch.tensor(dk, dtype=torch.float32) * torch.tensor(scale, dtype=torch.float32)
    p = torch.softmax(s_int * float(c2), dim=-1)
    r = p / float(torch.tensor(dw, dtype=torch.float32))
    bw = torch.clamp(torch.round(r), 0, w_level - 1)                                          # round-half-even, as rintf
    frac = (r - torch.floor(r) - 0.5).abs()
    near = ((frac < 2e-4) & (r < w_level - 0.5)).any(dim=-1)
    o_int = bw @ iv                                                                           # exact
    so = torch.tensor(dw, dtype=torch.float32) * torch.tensor(dv, dtype=torch.float32)
    out = (so.to(DEV) * o_int.float()).permute(0, 2, 1, 3).reshape(B, Tq, C)
    return out, near, float(so)


@pytest.mark.parametrize("B,heads,Tq,Tk,d,w_level", [(2, 8, 256, 256, 40, 256), (2, 4, 200, 77, 40, 256), (1, 8, 1024, 1024, 40, 256), (3, 2, 64, 64, 64, 256),
                                                    (2, 8, 128, 77, 80, 256), (1, 8, 96, 96, 160, 256), (2, 1, 256, 256, 32, 256), (2, 4, 100, 50, 24, 16),
                                                    (1, 2, 130, 33, 96, 256), (1, 3, 64, 640, 128, 64)])
def test_q8_attention_vs_integer_restatement(ops, B, heads, Tq, Tk, d, w_level):
    scale = float(d ** -0.5)
    q, k, v, qt, sel = _case(ops, B, heads, Tq, Tk, d, w_level, 1000 + Tq + d)
    out = ops.attention_q8(q, k, v, heads, scale, *sel, w_level)
    ref, near, so = _int_reference(ops, q, k, v, heads, scale, qt, sel, w_level)
    assert torch.isfinite(out).all()
    clean = ~near                                                                            # [B, heads, Tq]
    o4, r4 = out.reshape(B, Tq, heads, d).permute(0, 2, 1, 3), ref.reshape(B, Tq, heads, d).permute(0, 2, 1, 3)
    same = (o4 == r4).all(dim=-1)
    n_clean, n_same_clean = int(clean.sum()), int((same & clean).sum())
    worst = float((o4 - r4).abs().max()) / so
    print(f"[q8 attention B{B} h{heads} Tq{Tq} Tk{Tk} d{d} L{w_level}] rows bit-identical to the integer restatement: {int(same.sum())} of {same.numel()} "
          f"({n_same_clean} of the {n_clean} rows without a boundary value); worst element {worst:.1f} integer units")
    assert n_clean > 0.5 * clean.numel()
    assert n_same_clean >= n_clean - max(2, n_clean // 500), (n_clean, n_same_clean)          # exp2 vs exp: a value within ~1e-6 of the boundary
    assert worst <= 4 * 255                                                                  # a moved row: a few bins times |b_v - z_v|
    again = ops.attention_q8(q, k, v, heads, scale, *sel, w_level)
    assert torch.equal(again, out)


@pytest.mark.parametrize("B,heads,Tq,Tk,d,w_level", [(2, 8, 256, 256, 40, 256), (2, 4, 200, 77, 40, 256), (2, 8, 128, 77, 80, 256), (1, 3, 64, 640, 128, 64)])
def test_q8_attention_vs_oracle_restatement(ops, B, heads, Tq, Tk, d, w_level):
    """The same bar with the ORACLE deciding the bins of q, k, v (host restatement of quant/quant_layer.py:225) and of the softmax
    (round-half-even of p / delta_w in float64, clamped): what makes tfmq_attention_q8 the default of a block with live quantizers."""
    scale = float(d ** -0.5)
    q, k, v, qt, sel = _case(ops, B, heads, Tq, Tk, d, w_level, 4000 + Tq + d)
    out = ops.attention_q8(q, k, v, heads, scale, *sel, w_level)
    ref, near, so = _int_reference(ops, q, k, v, heads, scale, qt, sel, w_level, oracle_bins=True)
    lib_bins = [ops.quantize_act(x.contiguous(), s).to(torch.int32) + 128 for x, s in zip((q, k, v), sel[:3])]
    import tfmq_oracle as O
    for i, (x, lb) in enumerate(zip((q, k, v), lib_bins)):
        ob = O.quant_index(x.cpu(), qt[0, i, 0].cpu(), qt[0, i, 1].cpu(), 256).to(torch.int32)
        assert torch.equal(ob, lb.cpu()), f"input quantizer {i}: the library's bins differ from the oracle's"
    clean = ~near
    o4, r4 = out.reshape(B, Tq, heads, d).permute(0, 2, 1, 3), ref.reshape(B, Tq, heads, d).permute(0, 2, 1, 3)
    same = (o4 == r4).all(dim=-1)
    n_clean, n_same_clean = int(clean.sum()), int((same & clean).sum())
    print(f"[q8 attention vs oracle bins B{B} h{heads} Tq{Tq} Tk{Tk} d{d} L{w_level}] {n_same_clean} of the {n_clean} rows without a boundary value "
          f"bit-identical ({int(same.sum())} of {same.numel()} overall)")
    assert n_clean > 0.5 * clean.numel()
    assert n_same_clean >= n_clean - max(2, n_clean // 500), (n_clean, n_same_clean)
    assert float((o4 - r4).abs().max()) / so <= 4 * 255


@pytest.mark.parametrize("B,heads,Tq,Tk,d,pre", [(2, 8, 256, 256, 40, 1.0), (2, 8, 128, 77, 40, 1.0), (2, 4, 256, 256, 64, 64 ** -0.25)])
def test_q8_attention_vs_functional_path(ops, B, heads, Tq, Tk, d, pre):
    scale = float(d ** -0.5) if pre == 1.0 else 1.0
    q, k, v, qt, sel = _case(ops, B, heads, Tq, Tk, d, 256, 77 + Tk + d)
    a = ops.attention_q8(q, k, v, heads, scale, *sel, 256, pre)
    b = ops.attention_quant(q, k, v, heads, scale, *sel, 256, pre)
    dw, dv = float(qt[0, 3, 0]), float(qt[0, 2, 0])
    rel = float((a - b).norm() / b.norm())
    moved = float(((a - b).abs() > 0.5 * dw * dv).float().mean())
    print(f"[q8 vs functional d{d} Tk{Tk} pre {pre:.3f}] rel-L2 {rel:.2e}; elements moved by more than half an integer unit: {moved:.3%}")
    assert rel <= 2e-3 and moved <= 0.05


def test_q8_attention_refuses_what_it_cannot_take(ops):
    from tfmq_dm_amd._lib import TfmqError
    q, k, v, qt, sel = _case(ops, 1, 1, 64, 64, 256, 256, 5)
    assert not ops.attention_q8_ok(256, 256) and not ops.attention_q8_ok(40, 65536) and ops.attention_q8_ok(40, 256)
    with pytest.raises(TfmqError):
        ops.attention_q8(q, k, v, 1, 1.0, *sel, 256)


@pytest.mark.parametrize("which", ["ddim", "ldm", "attnblock"])
def test_engines_with_the_q8_attention_kernel(golden, monkeypatch, which):
    """F21's models with every attention quantizer on: TFMQ_ATTN_Q8=1 routes the attention through tfmq_attention_q8 where the head fits
    (d <= 160), and the output stays as close to the reference as the functional path is."""
    import test_attention_quant_gpu as TA
    g, eng, args, pre, anames, n_act, qtable = TA._setup(golden, which, monkeypatch, exact=True)
    monkeypatch.delenv("TFMQ_ATTN_Q8", raising=False)
    e0 = TA.nchw(eng.forward(*args))
    import tfmq_dm_amd.ops as ops_
    calls = []
    orig = ops_.attention_q8
    monkeypatch.setattr(ops_, "attention_q8", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    monkeypatch.setenv("TFMQ_ATTN_Q8", "1")
    e1 = TA.nchw(eng.forward(*args))
    ref = TA.T(g[pre + "eps_w4a8_attnq"])
    r0, r1, r01 = TA.rel_l2(e0, ref), TA.rel_l2(e1, ref), TA.rel_l2(e1, e0)
    print(f"[{which}] eps vs reference: functional {r0:.3e}, q8 kernel {r1:.3e} ({len(calls)} attention launches); q8 vs functional {r01:.3e}")
    assert len(calls) > 0 and torch.isfinite(e1).all()
    # a softmax bin on a rounding boundary moves between the two paths and avalanches through the W4A8 layers behind it, like the functional
    # path's own distance from the reference (tests/_avalanche.py)
    assert r1 <= max(5e-3, 3 * r0) and r01 <= max(5e-3, 2 * r0)


@pytest.mark.parametrize("which", ["ddim", "ldm"])
def test_q8_attention_is_the_default_of_live_quantizers(golden, monkeypatch, which):
    """A (non exact-fp) engine whose attention quantizers are on routes through tfmq_attention_q8 without any switch; TFMQ_ATTN_Q8=0 takes
    it back to the functional path, an observer (calibration) always does."""
    import test_attention_quant_gpu as TA
    g, eng, args, pre, anames, n_act, qtable = TA._setup(golden, which, monkeypatch, exact=False)
    import tfmq_dm_amd.ops as ops_
    calls = []
    orig = ops_.attention_q8
    monkeypatch.setattr(ops_, "attention_q8", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    monkeypatch.delenv("TFMQ_ATTN_Q8", raising=False)
    e1 = TA.nchw(eng.forward(*args))
    n_default = len(calls)
    eng.set_calibration("record", 0)
    eng.forward(*args)
    eng.set_calibration(None)
    n_record = len(calls) - n_default
    monkeypatch.setenv("TFMQ_ATTN_Q8", "0")
    e0 = TA.nchw(eng.forward(*args))
    n_off = len(calls) - n_default - n_record
    ref = TA.T(g[pre + "eps_w4a8_attnq"])
    print(f"[{which}] default: {n_default} tfmq_attention_q8 launches, eps vs reference {TA.rel_l2(e1, ref):.3e}; TFMQ_ATTN_Q8=0: {TA.rel_l2(e0, ref):.3e}")
    assert n_default > 0 and n_record == 0 and n_off == 0
    assert TA.rel_l2(e1, ref) <= 4e-2 and TA.rel_l2(e0, ref) <= 4e-2
