"""fp16-operand attention (tfmq_attention_f16) and the projection GEMM's transposed fp16 output that feeds it.

Reference semantics: QuantAttnBlock / cross_attn_forward (quant/quant_block.py:285-299, 483-500): softmax(q k^T * scale) v
on un-quantised operands.  Bar 3e-3 max-normalised, as for tfmq_attention: q, k, v and P are rounded to f16 for the
MFMA (fp32 accumulation).
"""
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


def qtab(delta, zp):
    return torch.tensor([[float(delta), float(zp)]], dtype=torch.float32, device=DEV)


def maxnorm(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("B,heads,Tq,Tk,d", [(2, 8, 256, 256, 40), (1, 2, 200, 200, 64), (2, 1, 130, 136, 160),
                                             (1, 4, 64, 72, 32), (1, 2, 128, 1000, 80), (2, 1, 256, 256, 128),
                                             (3, 1, 256, 256, 256), (1, 2, 130, 200, 208),
                                             (2, 1, 256, 256, 384), (1, 1, 130, 200, 320), (1, 2, 128, 136, 264)])   # wide heads: sliced output
def test_attention_f16_vs_fp32(ops, B, heads, Tq, Tk, d):
    gen = torch.Generator().manual_seed(Tq + d)
    C = heads * d
    q = torch.randn(B, Tq, C, generator=gen)
    k = torch.randn(B, Tk, C, generator=gen)
    v = torch.randn(B, Tk, C, generator=gen)
    scale = d ** -0.5
    qh = q.reshape(B, Tq, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    ref = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh
    ref = ref.permute(0, 2, 1, 3).reshape(B, Tq, C)
    ad, az = O.minmax(ref, 256)
    assert ops.attention_f16_ok(d, Tk)
    vt = v.transpose(1, 2).contiguous().half().to(DEV)
    out, yq = ops.attention_f16(q.half().to(DEV), k.half().to(DEV), vt, heads, scale, ops.qsel(qtab(ad, az)))
    assert maxnorm(out.cpu(), ref) <= 3e-3
    assert torch.equal(yq.cpu().float() + 128, O.quant_index(out.cpu(), ad, az, 256))
    # agrees with the fp32-operand kernel to the rounding of exp2 / accumulation order
    out32, _ = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), heads, scale)
    assert maxnorm(out.cpu(), out32.cpu()) <= 2e-3
    # q | k column slices of one fused buffer
    if Tq == Tk:
        qk = torch.cat([q, k], -1).half().to(DEV)
        out2, _ = ops.attention_f16(qk[..., :C], qk[..., C:], vt, heads, scale)
        assert torch.equal(out2, out)


@pytest.mark.parametrize("B,T,cin,C", [(2, 256, 128, 128), (1, 64, 64, 320), (2, 100, 320, 64)])
def test_fused_qkv_gemm_writes_v_transposed(ops, B, T, cin, C):
    gen = torch.Generator().manual_seed(C + T)
    x = torch.randn(B, T, 1, cin, generator=gen)
    w = torch.randn(3 * C, cin, 1, 1, generator=gen) * (2.0 / cin ** 0.5)
    b = torch.randn(3 * C, generator=gen) * 0.1
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    xq = ops.quantize_act(x.to(DEV), sel)
    pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
    y32 = ops.conv2d_w4a8(xq, pw, sel).reshape(B, T, 3 * C)
    if (2 * C) % 128:
        with pytest.raises(Exception):
            ops.conv2d_w4a8(xq, pw, sel, out_f16=True, t_col0=2 * C)
        return
    y16, yt = ops.conv2d_w4a8(xq, pw, sel, out_f16=True, t_col0=2 * C)
    y16 = y16.reshape(B, T, 3 * C)
    assert torch.equal(y16[..., :2 * C], y32[..., :2 * C].half())
    assert yt.shape == (B, C, T)
    assert torch.equal(yt, y32[..., 2 * C:].half().transpose(1, 2))


@pytest.mark.parametrize("B,heads,Tq,Tk,d", [(2, 1, 256, 256, 384), (1, 1, 64, 1, 960), (2, 2, 100, 77, 320)])
def test_attention_wide_heads(ops, B, heads, Tq, Tk, d):
    """cin256-style attention (one head of 384..960 channels; cross attention over ONE class token): head dims above the
    flash kernels' limit take the three-launch exact-fp32 path.  Bar 1e-5 max-normalised (fp32 throughout)."""
    gen = torch.Generator().manual_seed(Tq + d)
    C = heads * d
    q = torch.randn(B, Tq, C, generator=gen)
    k = torch.randn(B, Tk, C, generator=gen)
    v = torch.randn(B, Tk, C, generator=gen)
    scale = d ** -0.5
    qh = q.reshape(B, Tq, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).permute(0, 2, 1, 3).reshape(B, Tq, C)
    ad, az = O.minmax(ref, 256)
    out, yq = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), heads, scale, ops.qsel(qtab(ad, az)))
    assert maxnorm(out.cpu(), ref) <= 1e-5
    assert torch.equal(yq.cpu().float() + 128, O.quant_index(out.cpu(), ad, az, 256))
    if Tq == Tk:      # column slices of a fused q|k|v buffer
        qkv = torch.cat([q, k, v], -1).to(DEV)
        out2, _ = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, scale)
        assert torch.equal(out2, out)


def test_attention_f16_padded_keys(ops):
    """Keys stored padded to a multiple of 8 (77 CLIP tokens -> 80 rows): the padding is masked, the result equals the
    fp32-operand kernel on the un-padded operands to the f16 tolerance."""
    gen = torch.Generator().manual_seed(77)
    B, heads, Tq, Tk, d = 2, 8, 256, 77, 40
    C = heads * d
    q = torch.randn(B, Tq, C, generator=gen)
    k = torch.randn(B, Tk, C, generator=gen)
    v = torch.randn(B, Tk, C, generator=gen)
    scale = d ** -0.5
    qh = q.reshape(B, Tq, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).permute(0, 2, 1, 3).reshape(B, Tq, C)
    kp = torch.zeros(B, 80, C)
    kp[:, :Tk] = k
    kp[:, Tk:] = 50.0            # poison: would dominate the softmax if the padding were not masked
    vp = torch.zeros(B, 80, C)
    vp[:, :Tk] = v
    vp[:, Tk:] = 1e4
    out, _ = ops.attention_f16(q.half().to(DEV), kp.half().to(DEV), vp.transpose(1, 2).contiguous().half().to(DEV), heads, scale,
                               n_keys=Tk)
    assert maxnorm(out.cpu(), ref) <= 3e-3


@pytest.mark.parametrize("B,T,cin,C", [(2, 256, 128, 128), (1, 64, 64, 320), (2, 100, 448, 448)])
def test_f16_qkv_conv_writes_v_transposed_on_the_direct_kernel(ops, B, T, cin, C):
    """The un-quantised fused q|k|v conv (LDM AttentionBlock, FP state of the SpatialTransformer): fp16 rows for q | k and V^T for v,
    from the register-direct pointwise kernel on fp16 operands -- bit-identical to the tile kernel's output."""
    import tfmq_dm_amd.ops as _o
    if (2 * C) % 128:
        pytest.skip("transposed region must start at a multiple of 128 channels")
    gen = torch.Generator().manual_seed(C + T)
    x = torch.randn(B, T, 1, cin, generator=gen).to(DEV).half()
    w = (torch.randn(3 * C, cin, generator=gen) * (2.0 / cin ** 0.5)).to(DEV)
    b = (torch.randn(3 * C, generator=gen) * 0.1).to(DEV)
    pf = ops.pack_w_f16(w, b)
    outs = {}
    for tile in (1, 6):
        ops.set_conv_autotune({})
        orig = _o._tune_conv
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            outs[tile] = ops.conv2d_f16(x, pf, out_f16=True, t_col0=2 * C)
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
    (y1, vt1), (y6, vt6) = outs[1], outs[6]
    assert torch.equal(y1[..., :2 * C], y6[..., :2 * C]) and torch.equal(vt1, vt6)
    ref = torch.nn.functional.linear(x.float().reshape(B, T, cin), w.half().float(), b)
    assert float((vt6.float().transpose(1, 2) - ref[..., 2 * C:]).abs().max()) <= 2e-3 * float(ref.abs().max())


def _ref64(q, k, v, heads, scale):
    B, Tq, C = q.shape
    Tk, d = k.shape[1], C // heads
    qh = q.double().reshape(B, Tq, heads, d).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    vh = v.double().reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    return (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).permute(0, 2, 1, 3).reshape(B, Tq, C).float()


@pytest.mark.parametrize("B,heads,T", [(1, 2, 1024), (2, 8, 256), (1, 1, 128), (1, 3, 384)])
def test_attention_d40_pipelined_kernel(ops, B, heads, T):
    """The software-pipelined d = 40 self-attention kernel (k_attention_d40: Tq, Tk multiples of 128) against a float64
    reference on fp16-rounded operands and against the kernel it replaces (TFMQ_ATTN_PIPE=0 selects that one at library
    load; here the ragged-key entry n_keys = Tk - 8 + padding is what still takes it).  Bar 3e-3 max-normalised, as for
    every fp16-operand attention; the two kernels take the same exponentials against shifts that may differ -> 1e-3."""
    d = 40
    gen = torch.Generator().manual_seed(T + heads)
    C = heads * d
    q = torch.randn(B, T, C, generator=gen).half().float()
    k = torch.randn(B, T, C, generator=gen).half().float()
    v = torch.randn(B, T, C, generator=gen).half().float()
    scale = d ** -0.5
    ref = _ref64(q, k, v, heads, scale)
    ad, az = O.minmax(ref, 256)
    vt = v.transpose(1, 2).contiguous().half().to(DEV)
    out, yq = ops.attention_f16(q.half().to(DEV), k.half().to(DEV), vt, heads, scale, ops.qsel(qtab(ad, az)))
    assert maxnorm(out.cpu(), ref) <= 1.5e-3
    assert torch.equal(yq.cpu().float() + 128, O.quant_index(out.cpu(), ad, az, 256))
    # the tile-at-a-time kernel on the same operands (72 padding keys, masked: the ragged-tail entry never takes the pipelined kernel)
    kp = torch.cat([k, torch.full((B, 72, C), 30.0)], 1)
    vp = torch.cat([v, torch.full((B, 72, C), 1e4)], 1)
    old, _ = ops.attention_f16(q.half().to(DEV), kp.half().to(DEV), vp.transpose(1, 2).contiguous().half().to(DEV), heads, scale, n_keys=T)
    assert maxnorm(out.cpu(), old.cpu()) <= 1e-3


@pytest.mark.parametrize("spike_tile,gain", [(0, 40.0), (3, 40.0), (5, 400.0), (7, -60.0)])
def test_attention_d40_rescale_branch(ops, spike_tile, gain):
    """Force the rare branches of the folded softmax shift (cdna guide rule 26: a refcheck on bounded random data never takes
    them): one key of a chosen 64-key tile is aligned with a block of queries so that their scores jump by `gain` there
    (far above the 2^8 slack of the shift -> re-base, O / pending P / next scores rescaled exactly once), or every score of
    the first tile is far BELOW zero (gain < 0: the first tile sets the shift downwards, nothing underflows)."""
    d, heads, B, T = 40, 2, 1, 512
    gen = torch.Generator().manual_seed(11 + spike_tile)
    C = heads * d
    q = torch.randn(B, T, C, generator=gen)
    k = torch.randn(B, T, C, generator=gen)
    v = torch.randn(B, T, C, generator=gen)
    scale = d ** -0.5
    if gain > 0:
        key = spike_tile * 64 + 17
        u = torch.randn(d, generator=gen)
        u = u / u.norm()
        for hd in range(heads):
            k[0, key, hd * d:(hd + 1) * d] = u * (gain / scale) ** 0.5
            q[0, 40:200, hd * d:(hd + 1) * d] = u * (gain / scale) ** 0.5 + 0.1 * q[0, 40:200, hd * d:(hd + 1) * d]
    else:           # all scores of queries 0..127 strongly negative in the first tiles, ordinary later
        for hd in range(heads):
            u = torch.ones(d) / d ** 0.5
            q[0, :128, hd * d:(hd + 1) * d] = u * (-gain / scale) ** 0.5
            k[0, :128, hd * d:(hd + 1) * d] = -u * (-gain / scale) ** 0.5 + 0.05 * k[0, :128, hd * d:(hd + 1) * d]
    q, k, v = q.half().float(), k.half().float(), v.half().float()
    ref = _ref64(q, k, v, heads, scale)
    vt = v.transpose(1, 2).contiguous().half().to(DEV)
    out, _ = ops.attention_f16(q.half().to(DEV), k.half().to(DEV), vt, heads, scale)
    assert torch.isfinite(out).all()
    assert maxnorm(out.cpu(), ref) <= 3e-3


def test_attention_d40_pipelined_kernel_is_deterministic(ops):
    """Run-to-run bit identity, separate vs fused q|k buffers (different leading dimensions).  Round 3 history: the row-maximum tree of
    the pipelined kernel is inline asm (v_max3_f32), and the compiler's hazard recogniser does not count wait states between an MFMA and
    an asm statement that READS its result -- the first shift was sometimes taken from half-accumulated scores: every output still within
    tolerance (any shift near the maximum is valid), but ~9 % of the elements moved by an fp16 rounding from launch to launch."""
    d = 40
    for B, heads, T_ in [(2, 8, 256), (1, 2, 1024), (2, 4, 2048)]:
        gen = torch.Generator().manual_seed(T_)
        C = heads * d
        q = torch.randn(B, T_, C, generator=gen).half().to(DEV)
        k = torch.randn(B, T_, C, generator=gen).half().to(DEV)
        vt = torch.randn(B, C, T_, generator=gen).half().to(DEV)
        qk = torch.cat([q, k], -1).contiguous()
        ref, _ = ops.attention_f16(q, k, vt, heads, d ** -0.5)
        for _ in range(12):
            o1, _ = ops.attention_f16(q, k, vt, heads, d ** -0.5)
            o2, _ = ops.attention_f16(qk[..., :C], qk[..., C:], vt, heads, d ** -0.5)
            assert torch.equal(o1, ref) and torch.equal(o2, ref)


@pytest.mark.parametrize("B,heads,Tq,d", [(2, 8, 256, 40), (3, 8, 1024, 80), (1, 8, 128, 40), (2, 4, 384, 80)])
def test_context_attention_all_heads_per_workgroup(ops, B, heads, Tq, d, monkeypatch):
    monkeypatch.setenv("TFMQ_ATTN_CTX", "2")          # (d = 80 is only taken on request)
    """Round 4, k_attention_ctx (csrc/attention_ctx.hip): cross attention over the 77 CLIP tokens (stored padded to 80 keys, padding
    poisoned) with the int8 output of to_out's quantizer -- a workgroup keeps its 128 queries for all heads, one exact softmax over the
    <= 96 keys.  Taken when only the int8 output is asked for; compared with the quantised fp32 reference and with k_attention_h's bins
    (the same operand precision; the softmax denominator is summed differently): within one bin, a few per mille moved."""
    gen = torch.Generator().manual_seed(d + Tq)
    Tk, C = 77, heads * d
    q = torch.randn(B, Tq, C, generator=gen)
    k = torch.randn(B, Tk, C, generator=gen)
    v = torch.randn(B, Tk, C, generator=gen)
    scale = d ** -0.5
    qh = q.half().float().reshape(B, Tq, heads, d).permute(0, 2, 1, 3)
    kh = k.half().float().reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    vh = v.half().float().reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).permute(0, 2, 1, 3).reshape(B, Tq, C)
    od, oz = O.minmax(ref, 256)
    sel = ops.qsel(qtab(od, oz))
    kp = torch.zeros(B, 80, C)
    kp[:, :Tk] = k
    kp[:, Tk:] = 50.0
    vp = torch.zeros(B, 80, C)
    vp[:, :Tk] = v
    vp[:, Tk:] = 1e4
    args = (q.half().to(DEV), kp.half().to(DEV), vp.transpose(1, 2).contiguous().half().to(DEV), heads, scale)
    out_old, yq_old = ops.attention_f16(*args, sel, want_f32=True, n_keys=Tk)           # k_attention_h (fp32 + int8 outputs)
    _, yq_new = ops.attention_f16(*args, sel, want_f32=False, n_keys=Tk)                # k_attention_ctx
    assert yq_new.dtype == torch.int8 and yq_new.shape == (B, Tq, C)
    assert maxnorm(out_old.cpu(), ref) <= 3e-3
    d_old = (yq_new.int() - yq_old.int()).abs()
    assert int(d_old.max()) <= 1 and float((d_old > 0).float().mean()) < 1e-2
    bins = O.quant_index(ref, od, oz, 256)
    d_ref = (yq_new.cpu().float() + 128 - bins).abs()
    assert float(d_ref.max()) <= 1 and float((d_ref > 0).float().mean()) < 3e-2
    _, again = ops.attention_f16(*args, sel, want_f32=False, n_keys=Tk)
    assert torch.equal(again, yq_new)
