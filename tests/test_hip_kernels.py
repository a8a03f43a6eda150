"""Parity of every HIP kernel (through the C ABI) against the CPU oracle and the golden
vectors of the reference.  Needs an MI355X: `pytest -m gpu`.

Bars: bit-exact for bin indices, packed nibbles, MINMAX delta/zp, masks; stated fp tolerance
elsewhere (written next to each assert)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tfmq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tfmq_dm_amd.ops as ops_
    return ops_


DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


def qtab(delta, zp):
    return torch.tensor([[float(delta), float(zp)]], dtype=torch.float32, device=DEV)


def nhwc(x):  # NCHW cpu -> NHWC gpu
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(y):  # NHWC gpu -> NCHW cpu
    return y.permute(0, 3, 1, 2).contiguous().cpu()


def maxnorm(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ------------------------------------------------------------------ K1/K2
def test_quantize_act_bin_indices_bit_exact(ops, golden):
    g = golden("f1_quantizer")
    x = T(g["acts"])
    for nm in ("minmax", "mse"):
        d, z = float(g[f"acts_{nm}_delta"]), float(g[f"acts_{nm}_zp"])
        q = ops.quantize_act(x.to(DEV), ops.qsel(qtab(d, z)))
        assert np.array_equal((q.cpu().numpy().astype(np.int32) + 128).astype(np.uint8), g[f"acts_{nm}_idx"])
    # large ragged tensor incl. tail, values far outside the clip range
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(1_000_003, generator=gen) * 5 + 0.7
    xp = torch.zeros(1_000_004)
    xp[:1_000_003] = x
    d, z = O.minmax(x[:1000], 256)
    q = ops.quantize_act(xp.to(DEV)[:1_000_003], ops.qsel(qtab(d, z)))
    ref = O.quant_index(x, d, z, 256)
    assert np.array_equal(q.cpu().numpy().astype(np.int32) + 128, ref.numpy().astype(np.int32))


def test_fake_quant_per_channel_weights(ops, golden):
    g = golden("f1_quantizer")
    for tn in ("wts", "lin"):
        w = T(g[tn])
        for nm in ("minmax", "mse"):
            d, z = T(g[f"{tn}_{nm}_delta"]), T(g[f"{tn}_{nm}_zp"])
            y, idx = ops.fake_quant(w.to(DEV), d.to(DEV), z.to(DEV), 16, want_idx=True)
            assert np.array_equal(idx.cpu().numpy(), g[f"{tn}_{nm}_idx"])
            assert np.array_equal(y.cpu().numpy(), g[f"{tn}_{nm}_dq"])


def test_minmax_and_minmax_scaler_bit_exact(ops, golden):
    g = golden("f1_quantizer")
    for name, level, az in (("acts", 256, False), ("pos", 256, True)):
        x = T(g[name])
        mm = ops.minmax(x.to(DEV), 1)
        assert float(mm[0, 0]) == float(x.min()) and float(mm[0, 1]) == float(x.max())
        qp = ops.minmax_to_qparam(mm, level, az).cpu()
        assert float(qp[0, 0]) == float(g[f"{name}_minmax_delta"]) and float(qp[0, 1]) == float(g[f"{name}_minmax_zp"])
    for tn in ("wts", "lin"):
        w = T(g[tn])
        mm = ops.minmax(w.to(DEV), w.shape[0])
        ref = torch.stack([w.reshape(w.shape[0], -1).min(1)[0], w.reshape(w.shape[0], -1).max(1)[0]], 1)
        assert torch.equal(mm.cpu(), ref)
        qp = ops.minmax_to_qparam(mm, 16).cpu()
        assert np.array_equal(qp[:, 0].numpy(), g[f"{tn}_minmax_delta"].reshape(-1))
        assert np.array_equal(qp[:, 1].numpy(), g[f"{tn}_minmax_zp"].reshape(-1))
    # big tensor, odd size
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(3, 1_234_567, generator=gen)
    mm = ops.minmax(x.to(DEV), 3).cpu()
    assert torch.equal(mm[:, 0], x.min(1)[0]) and torch.equal(mm[:, 1], x.max(1)[0])


def test_act_momentum_update_sequence_bit_exact(ops, golden):
    g = golden("f2_momentum")
    xs = T(g["x"]).to(DEV)
    state = torch.zeros(1, 2, device=DEV)
    qp = torch.zeros(1, 2, device=DEV)
    ops.act_range_update(ops.minmax(xs[0]), state, qp, 0.95, 256, init=True)
    for i in range(1, xs.shape[0]):
        ops.act_range_update(ops.minmax(xs[i]), state, qp, 0.95, 256, init=False)
        s, q = state.cpu(), qp.cpu()
        assert float(s[0, 0]) == float(g["x_min"][i - 1]) and float(s[0, 1]) == float(g["x_max"][i - 1]), i
        assert float(q[0, 0]) == float(g["delta"][i - 1]) and float(q[0, 1]) == float(g["zp"][i - 1]), i


# ------------------------------------------------------------------ K3
def _check_mse(qp, losses, best, ref_delta, ref_zp, ref_losses=None):
    """argmin may differ from the CPU only where the two best losses are within 1e-6 relative
    (SURVEY §7-7); candidate losses themselves within 2e-5 relative."""
    qp, losses, best = qp.cpu(), losses.cpu(), best.cpu()
    for r in range(qp.shape[0]):
        if float(qp[r, 0]) == float(ref_delta[r]) and float(qp[r, 1]) == float(ref_zp[r]):
            continue
        lo = torch.sort(losses[r])[0]
        assert float(lo[1] - lo[0]) <= 1e-6 * float(lo[0]), (r, float(qp[r, 0]), float(ref_delta[r]))
    if ref_losses is not None:
        np.testing.assert_allclose(losses.numpy().reshape(-1), np.asarray(ref_losses, np.float32).reshape(-1), rtol=2e-5)


def test_mse_search_per_tensor_and_per_channel(ops, golden):
    g = golden("f1_quantizer")
    x = T(g["acts"])
    qp, losses, best = ops.mse_search(x.to(DEV), 1, 256, want_losses=True)
    _check_mse(qp, losses, best, [g["acts_mse_delta"]], [g["acts_mse_zp"]], g["acts_mse_cand_loss"])
    pos = T(g["pos"])
    qp, losses, best = ops.mse_search(pos.to(DEV), 1, 256, always_zero=True, want_losses=True)
    _check_mse(qp, losses, best, [g["pos_mse_delta"]], [g["pos_mse_zp"]])
    for tn in ("wts", "lin"):
        w = T(g[tn])
        qp, losses, best = ops.mse_search(w.to(DEV), w.shape[0], 16, want_losses=True)
        _check_mse(qp, losses, best, g[f"{tn}_mse_delta"].reshape(-1), g[f"{tn}_mse_zp"].reshape(-1))
    # a large activation tensor (multi-block reduction)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(16, 256, 16, 16, generator=gen) * 1.7 + 0.3
    d, z, b, ls = O.mse(x, 256, return_losses=True)
    qp, losses, best = ops.mse_search(x.to(DEV), 1, 256, want_losses=True)
    _check_mse(qp, losses, best, [float(d)], [float(z)], ls.numpy())


# ------------------------------------------------------------------ K4
def test_pack_w4_nibbles_bit_exact(ops, golden):
    g = golden("f3_quantlayer")
    for tag in ("conv3", "conv1", "lin2d", "lin3d"):
        w = T(g[f"{tag}_w"])
        d, z = T(g[f"{tag}_wdelta"]), T(g[f"{tag}_wzp"])
        pw = ops.pack_w4(w.to(DEV), d.to(DEV), z.to(DEV))
        idx = ops.unpack_w4(pw).cpu().reshape(w.shape)
        ref = O.quant_index(w, d, z, 16)
        assert torch.equal(idx.float(), ref), tag
        meta = pw.wmeta.cpu()
        assert torch.equal(meta[:, 0].float(), z.reshape(-1))
        assert torch.equal(meta[:, 1], ref.reshape(w.shape[0], -1).sum(1).to(torch.int32))
    # AdaRound hard masks (fixture F4: learned alpha after 20 iterations)
    g = golden("f4_adaround")
    w, d, z = T(g["w"]), T(g["wdelta"]), T(g["wzp"])
    alpha = T(g["alphas"][-1])
    w16 = torch.zeros(24, 16, 3, 3)
    w16[:] = w
    pw = ops.pack_w4(w16.to(DEV), d.to(DEV), z.to(DEV), alpha=alpha.contiguous().to(DEV))
    idx = ops.unpack_w4(pw).cpu().float()
    ref = O.adaround_index(w, alpha, d, z, 16, soft=False)
    assert torch.equal(idx, ref)
    assert torch.equal(d * (idx - z), T(g["w_hard_final"]))


# ------------------------------------------------------------------ K5/K6
def _w4a8_ref(x, w, b, wd, wz, ad, az, **kw):
    xq, wq = O.fake_quant(x, ad, az, 256), O.fake_quant(w, wd, wz, 16)
    return F.conv2d(xq, wq, b, **kw)


def test_conv_w4a8_golden_quantlayer(ops, golden):
    """F3: QuantLayer outputs of the reference (w4a8).  Bar: max-normalised error <= 1e-5."""
    g = golden("f3_quantlayer")
    for tag, pad in (("conv3", 1), ("conv1", 0)):
        x, w, b = T(g[f"{tag}_x"]), T(g[f"{tag}_w"]), T(g[f"{tag}_b"])
        wd, wz = T(g[f"{tag}_wdelta"]), T(g[f"{tag}_wzp"])
        ad, az = float(g[f"{tag}_adelta"]), float(g[f"{tag}_azp"])
        sel = ops.qsel(qtab(ad, az))
        xq = ops.quantize_act(nhwc(x), sel)
        pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
        y = ops.conv2d_w4a8(xq, pw, sel, pad=(pad, pad, pad, pad))
        assert maxnorm(nchw(y), T(g[f"{tag}_y"])) <= 1e-5, tag
    for tag in ("lin2d", "lin3d"):
        x, w = T(g[f"{tag}_x"]), T(g[f"{tag}_w"])
        b = T(g[f"{tag}_b"]) if f"{tag}_b" in g.files else None
        wd, wz = T(g[f"{tag}_wdelta"]), T(g[f"{tag}_wzp"])
        ad, az = float(g[f"{tag}_adelta"]), float(g[f"{tag}_azp"])
        sel = ops.qsel(qtab(ad, az))
        x4 = x.reshape(-1, 1, 1, x.shape[-1]).contiguous().to(DEV)
        xq = ops.quantize_act(x4, sel)
        pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=None if b is None else b.to(DEV))
        y = ops.conv2d_w4a8(xq, pw, sel).reshape(x.shape[:-1] + (w.shape[0],))
        assert maxnorm(y.cpu(), T(g[f"{tag}_y"])) <= 1e-5, tag


@pytest.mark.parametrize("B,H,W,cin,cout,k,stride,pad,up2x", [
    (3, 16, 16, 128, 128, 3, 1, (1, 1, 1, 1), False),
    (2, 9, 7, 64, 200, 3, 1, (1, 1, 1, 1), False),      # ragged M and N tails
    (2, 8, 8, 192, 160, 1, 1, (0, 0, 0, 0), False),
    (2, 8, 8, 64, 96, 3, 1, (1, 1, 1, 1), True),         # fused nearest-2x upsample
    (2, 17, 17, 64, 64, 3, 2, (0, 0, 1, 1), False),      # stride 2, asymmetric pad
    (1, 1, 1, 512, 24, 1, 1, (0, 0, 0, 0), False),       # single token, narrow-N tile
])
def test_conv_w4a8_vs_oracle(ops, B, H, W, cin, cout, k, stride, pad, up2x):
    gen = torch.Generator().manual_seed(B * 1000 + cin + cout)
    x = torch.randn(B, cin, H, W, generator=gen) * 1.7 + 0.3
    w = torch.randn(cout, cin, k, k, generator=gen) * 0.02
    b = torch.randn(cout, generator=gen) * 0.1
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    xq = ops.quantize_act(nhwc(x), sel)
    assert torch.equal(nchw(xq.float()) + 128, O.quant_index(x, ad, az, 256))
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up2x else x
    xqd = O.fake_quant(xin, ad, az, 256)
    xqd = F.pad(xqd, (pad[1], pad[3], pad[0], pad[2]))
    ref = F.conv2d(xqd, O.fake_quant(w, wd, wz, 16), b, stride=stride)
    rowadd = torch.randn(B, cout, generator=gen)
    res = torch.randn(ref.shape, generator=gen)
    pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
    y = ops.conv2d_w4a8(xq, pw, sel, stride=stride, pad=pad, up2x=up2x, rowadd=rowadd.to(DEV), residual=nhwc(res))
    ref = ref + rowadd[:, :, None, None] + res
    assert y.shape[1:3] == ref.shape[2:]
    assert maxnorm(nchw(y), ref) <= 1e-5


@pytest.mark.parametrize("B,H,W,cin,cout,k,stride,pad", [
    (2, 16, 16, 3, 128, 3, 1, (1, 1, 1, 1)),     # conv_in: Cin=3
    (2, 16, 16, 128, 3, 3, 1, (1, 1, 1, 1)),     # conv_out: Cout=3 (narrow tile)
    (2, 8, 8, 384, 256, 1, 1, (0, 0, 0, 0)),     # nin_shortcut
    (2, 16, 16, 128, 128, 3, 2, (0, 0, 1, 1)),   # downsample: pad (0,1,0,1), stride 2
])
def test_conv_f16_vs_fp32(ops, B, H, W, cin, cout, k, stride, pad):
    """Un-quantised layers run on f16 MFMA with fp32 accumulation.  Bar: 2e-3 max-normalised
    (f16 operand rounding, 2^-11 relative per operand)."""
    gen = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(B, cin, H, W, generator=gen)
    w = torch.randn(cout, cin, k, k, generator=gen) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=gen) * 0.1
    ref = F.conv2d(F.pad(x, (pad[1], pad[3], pad[0], pad[2])), w, b, stride=stride)
    pf = ops.pack_w_f16(w.to(DEV), b.to(DEV))
    y = ops.conv2d_f16(nhwc(x), pf, stride=stride, pad=pad)
    assert maxnorm(nchw(y), ref) <= 2e-3


# ------------------------------------------------------------------ K8
@pytest.mark.parametrize("B,HW,C1,C2,silu", [(3, 64, 128, 0, True), (2, 256, 256, 128, True), (2, 16, 32, 0, False),
                                             (2, 100, 320, 0, True), (2, 64, 64, 64, True)])
def test_groupnorm_silu_quant(ops, B, HW, C1, C2, silu):
    gen = torch.Generator().manual_seed(C1 + C2 + HW)
    side = int(HW ** 0.5)
    x1 = torch.randn(B, C1, side, HW // side, generator=gen) * 2 + 0.5
    x2 = torch.randn(B, C2, side, HW // side, generator=gen) if C2 else None
    xc = x1 if x2 is None else torch.cat([x1, x2], 1)
    Cc = C1 + C2
    gamma, beta = torch.randn(Cc, generator=gen), torch.randn(Cc, generator=gen) * 0.3
    ref = F.group_norm(xc, 32, gamma, beta, 1e-6)
    if silu:
        ref = O.swish(ref)
    ad, az = O.minmax(ref, 256)
    yq, yf, xcat = ops.groupnorm(nhwc(x1), gamma.to(DEV), beta.to(DEV), 1e-6, silu, ops.qsel(qtab(ad, az)),
                                 x2=None if x2 is None else nhwc(x2), want_f32=True, want_cat=True)
    assert torch.equal(nchw(xcat), xc)
    # fp32 result: 1e-5 max-normalised (different summation order / exp implementation)
    assert maxnorm(nchw(yf), ref) <= 1e-5
    # bin indices: a value within ~1e-6*range of a rounding boundary may land in the neighbouring bin
    idx, ridx = nchw(yq.float()) + 128, O.quant_index(ref, ad, az, 256)
    diff = (idx - ridx).abs()
    assert float(diff.max()) <= 1 and float((diff > 0).float().mean()) <= 2e-4
    # given identical fp32 inputs the quantiser itself is bit-exact
    assert torch.equal(idx, O.quant_index(nchw(yf), ad, az, 256))


# ------------------------------------------------------------------ K10
@pytest.mark.parametrize("B,heads,Tq,Tk,d", [(2, 1, 256, 256, 256), (2, 1, 16, 16, 256), (1, 8, 200, 77, 40),
                                             (2, 4, 130, 130, 32), (1, 2, 64, 64, 160)])
def test_attention_vs_fp32(ops, B, heads, Tq, Tk, d):
    """Bar 3e-3 max-normalised: q,k,v and P are rounded to f16 for the MFMA (fp32 accumulation)."""
    gen = torch.Generator().manual_seed(Tq + d)
    q = torch.randn(B, Tq, heads * d, generator=gen)
    k = torch.randn(B, Tk, heads * d, generator=gen)
    v = torch.randn(B, Tk, heads * d, generator=gen)
    scale = d ** -0.5
    qh = q.reshape(B, Tq, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(B, Tk, heads, d).permute(0, 2, 1, 3)
    ref = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh
    ref = ref.permute(0, 2, 1, 3).reshape(B, Tq, heads * d)
    ad, az = O.minmax(ref, 256)
    out, yq = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), heads, scale, ops.qsel(qtab(ad, az)))
    assert maxnorm(out.cpu(), ref) <= 3e-3
    assert torch.equal(yq.cpu().float() + 128, O.quant_index(out.cpu(), ad, az, 256))
    # fused-qkv layout: column slices of one [B,T,3C] buffer
    if Tq == Tk:
        qkv = torch.cat([q, k, v], -1).to(DEV)
        Cc = heads * d
        out2, _ = ops.attention(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], heads, scale)
        assert torch.equal(out2, out)


# ------------------------------------------------------------------ K7 / K11
def test_timestep_embedding_and_small_linears(ops, golden):
    g = golden("f9_schedules")
    t = T(g["temb_t"])
    for dim in (128, 32):
        emb = ops.timestep_embedding(t.to(DEV), dim).cpu()
        # arguments reach ~1e3: a 1-ulp difference in the fp32 frequency (device exp vs CPU libm)
        # moves sin/cos by up to 1e3 * 6e-8 = 6e-5; everything else agrees to ~1e-6.
        err = np.abs(emb.numpy() - g[f"temb_{dim}"])
        assert err.max() <= 1e-4 and (err <= 2e-6).mean() >= 0.97
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(5, 128, generator=gen)
    w = torch.randn(512, 128, generator=gen) * 0.05
    b = torch.randn(512, generator=gen) * 0.1
    y = ops.linear_small_f32(x.to(DEV), w.to(DEV), b.to(DEV), silu_in=True).cpu()
    assert maxnorm(y, F.linear(O.swish(x), w, b)) <= 1e-5
    # int4 weights, weight-only and w4a8
    x = torch.randn(11, 512, generator=gen)
    w = torch.randn(256, 512, generator=gen) * 0.03
    wd, wz = O.init_channelwise(w, 16, "minmax")
    pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b[:256].contiguous().to(DEV))
    y = ops.linear_small_w4(x.to(DEV), pw, ops.qsel(None), silu_in=False).cpu()
    assert maxnorm(y, F.linear(x, O.fake_quant(w, wd, wz, 16), b[:256])) <= 1e-5
    xs = O.swish(x)
    ad, az = O.minmax(xs, 256)
    y = ops.linear_small_w4(x.to(DEV), pw, ops.qsel(qtab(ad, az)), silu_in=True).cpu()
    ref = F.linear(O.fake_quant(xs, ad, az, 256), O.fake_quant(w, wd, wz, 16), b[:256])
    # SiLU on device vs CPU can move a value across a bin edge: allow one delta on isolated outputs
    assert maxnorm(y, ref) <= 1e-5 or float(((y - ref).abs() > 1e-5 * ref.abs().max()).float().mean()) < 0.02


def test_ddim_update_bit_exact_and_layouts(ops):
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(4, 3, 32, 32, generator=gen)
    e = torch.randn(4, 3, 32, 32, generator=gen)
    betas = O.linear_betas()
    seq = O.ddim_seq("quad", 100)
    i, j = seq[57], seq[56]
    at = O.compute_alpha(betas, torch.tensor([i]))
    an = O.compute_alpha(betas, torch.tensor([j]))
    x0 = (x - e * (1 - at).sqrt()) / at.sqrt()
    c2 = ((1 - an) - 0.0 ** 2).sqrt()
    ref = an.sqrt() * x0 + 0.0 * torch.randn_like(x) + c2 * e
    coef = torch.tensor([[float((1 - at).sqrt()), float(at.sqrt()), float(an.sqrt()), 0.0, float(c2), 0, 0, 0]], device=DEV)
    xn, x0d = ops.ddim_update(x.to(DEV), e.to(DEV), coef, want_x0=True)
    assert torch.equal(x0d.cpu(), x0) and torch.equal(xn.cpu(), ref)
    y = ops.nchw_to_nhwc(x.to(DEV))
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw(y).cpu(), x)
    x = torch.randn(3, 37, 5, 7, generator=gen)
    assert torch.equal(ops.nhwc_to_nchw(ops.nchw_to_nhwc(x.to(DEV))).cpu(), x)


# ------------------------------------------------------------------ K12-K14
def test_adaround_kernels_vs_golden(ops, golden):
    g = golden("f4_adaround")
    w, b, x, y_fp = T(g["w"]), T(g["b"]), T(g["x"]), T(g["y_fp"])
    wd, wz = T(g["wdelta"]), T(g["wzp"])
    wdev, dd, zd = w.to(DEV), wd.to(DEV), wz.to(DEV)
    alpha = ops.adaround_init(wdev, dd)
    np.testing.assert_allclose(alpha.cpu().numpy(), g["alpha0"], rtol=1e-5, atol=1e-6)  # logf ulps
    a0 = T(g["alpha0"]).to(DEV)
    w_soft = ops.adaround_soft_fwd(wdev, a0, dd, zd, 16)
    np.testing.assert_allclose(w_soft.cpu().numpy(), g["w_soft0"], rtol=1e-5, atol=1e-8)
    # 20 iterations: conv fwd / weight-grad on the CPU oracle side, AdaRound + Adam on the GPU.
    iters = 20
    alpha = a0.clone()
    m, v = torch.zeros_like(alpha), torch.zeros_like(alpha)
    gi, ai = list(g["grad_iters"]), list(g["alpha_iters"])
    for it in range(iters):
        w_hat = ops.adaround_soft_fwd(wdev, alpha, dd, zd, 16).cpu().requires_grad_(True)
        yq = F.conv2d(x, w_hat, b, padding=1)
        rec = O.lp_loss(yq, y_fp)
        rec.backward()
        count = it + 1
        btemp = O.temp_decay(count, iters, 0.2) if count >= iters * 0.2 else 0.0
        rl = torch.zeros(1, device=DEV)
        ops.adaround_bwd_adam(wdev, alpha, dd, zd, w_hat.grad.contiguous().to(DEV), m, v, 16, 0.01, btemp, 1e-3, count, rl)
        tot = float(rec) + float(rl)
        assert abs(tot - float(g["loss"][it])) <= 2e-5 * abs(float(g["loss"][it])), it
        if it in ai:
            np.testing.assert_allclose(alpha.cpu().numpy(), g["alphas"][ai.index(it)], rtol=2e-5, atol=2e-6)
    mask = (alpha.cpu() >= 0).numpy().astype(np.uint8)
    assert (mask == g["mask_final"]).mean() >= 0.9999


def test_recon_loss_kernel(ops):
    gen = torch.Generator().manual_seed(8)
    p = torch.randn(8, 24, 6, 6, generator=gen)
    t = torch.randn(8, 24, 6, 6, generator=gen)
    pr = p.clone().requires_grad_(True)
    ref = O.lp_loss(pr, t)
    ref.backward()
    loss, gr = ops.recon_loss(nhwc(p), nhwc(t), denom=8 * 6 * 6)
    assert abs(float(loss) - float(ref)) <= 1e-5 * float(ref)
    np.testing.assert_allclose(nchw(gr).numpy(), pr.grad.numpy(), rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_histogram_scalers_kl_and_hist_vs_reference(golden):
    """Scaler.KL / Scaler.HIST of the drop-in quant_layer on the device histogram (tfmq_np_histogram: numpy's bin arithmetic,
    fp32 for the raw data, float64 for the clipped data) against the reference's own functions (fixture F18: five input
    distributions, 256 / 16 levels, with and without always_zero): delta and zero point bit for bit; and the histogram kernel
    itself against np.histogram."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tfmq-dm_amd"))
    from quant.quant_layer import Scaler, UniformAffineQuantizer, hist, kl
    import tfmq_dm_amd.ops as ops
    g = golden("f18_hist_scalers")
    n = 0
    for k in g.files:
        if not (k.startswith("kl/") or k.startswith("hist/")):
            continue
        fn, name, level, az = k.split("/")
        x = torch.from_numpy(g[f"x/{name}"]).to(DEV)
        d, z = (kl if fn == "kl" else hist)(x, False, int(level), bool(int(az)))
        assert float(d) == float(np.float32(g[k][0])) and float(z) == g[k][1], (k, float(d), float(z), g[k])
        n += 1
    assert n == 28
    a = g["x/heavy"]
    ref, edges = np.histogram(a, bins=256)
    assert np.array_equal(ops.np_histogram(torch.from_numpy(a).to(DEV), edges.astype(np.float32)), ref)
    lo, hi = np.min(a) * np.float64(0.7), np.max(a) * np.float64(0.7)      # float64 bounds, as linspace's ratios make them
    ref, edges = np.histogram(np.clip(a, lo, hi), bins=64)
    assert edges.dtype == np.float64
    assert np.array_equal(ops.np_histogram(torch.from_numpy(a).to(DEV), edges, clip=(lo, hi)), ref)
    # through the quantizer's lazy initialisation, per tensor and per channel
    q = UniformAffineQuantizer(bits=8, scaler=Scaler.KL, leaf_param=True)
    q(torch.from_numpy(g["x/normal"]).to(DEV))
    assert float(q.delta) == float(np.float32(g["kl/normal/256/0"][0])) and float(q.zero_point) == g["kl/normal/256/0"][1]
    w = torch.stack([torch.from_numpy(g["x/small"])[:600], torch.from_numpy(g["x/positive"])[:600]]).to(DEV)
    qc = UniformAffineQuantizer(bits=4, scaler=Scaler.HIST, channel_wise=True)
    qc(w)
    d0, z0 = hist(w[0], False, 16, False)
    assert qc.delta.shape == (2, 1) and float(qc.delta[0]) == float(d0) and float(qc.zero_point[0]) == float(z0)


@pytest.mark.gpu
def test_channel_wise_histogram_scalers_batched_equal_the_channel_loop(monkeypatch):
    """Channel-wise Scaler.KL / Scaler.HIST (the reference loops the output channels through the scaler, quant_layer.py:193-204): all
    channels per launch (kl_rows / hist_rows: 51 launches in all, the bin bookkeeping vectorised over the rows) against the channel loop
    over the per-tensor functions that F18 pins to the reference -- delta and zero point bit for bit."""
    from quant.quant_layer import Scaler, UniformAffineQuantizer
    g = torch.Generator().manual_seed(18)
    w = torch.randn(24, 16, 3, 3, generator=g) * 0.1
    w[3] *= 4.0
    w[7, 0, 0, 0] = 2.5                 # an outlier channel: the clip search has something to find
    w[11] = w[11].abs()                 # a one-sided channel
    for scaler in (Scaler.KL, Scaler.HIST):
        for bits, az in ((4, False), (8, False), (8, True)):
            res = []
            for loop in (False, True):
                if loop:
                    monkeypatch.setenv("TFMQ_SCALER_ROW_LOOP", "1")
                else:
                    monkeypatch.delenv("TFMQ_SCALER_ROW_LOOP", raising=False)
                q = UniformAffineQuantizer(bits=bits, channel_wise=True, scaler=scaler, always_zero=az)
                d, z = q._init_quantization_param(w.to(DEV), True)
                res.append((d.cpu(), z.cpu()))
            assert res[0][0].shape == (24, 1, 1, 1)
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (scaler.__name__, bits, az)
            assert len(torch.unique(res[0][0])) > 4
