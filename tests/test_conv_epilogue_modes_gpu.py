"""Fused output modes of the w4a8 GEMM epilogue (include/tfmq_hip.h: tfmq_conv_desc.out_mode).

TFMQ_OUT_F16       the fp32 result rounded once to fp16 -- bit-identical to casting the fp32 output.
TFMQ_OUT_GEGLU_Q8  GEGLU (ldm/modules/attention.py:52-59: x, gate = proj(x).chunk(2); x * gelu(gate)) followed by
                   the next QuantLayer's 8-bit activation quantizer (quant_layer.py:223-226), checked bit-exactly
                   against the unfused kernels and against the oracle's arithmetic.
"""
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


def _setup(ops, B, T, cin, cout, seed):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, 1, cin, generator=gen) * 1.3 - 0.2
    w = torch.randn(cout, cin, 1, 1, generator=gen) * (2.0 / cin ** 0.5)
    b = torch.randn(cout, generator=gen) * 0.2
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    xq = ops.quantize_act(x.to(DEV), sel)
    return x, w, b, wd, wz, ad, az, sel, xq


@pytest.mark.parametrize("B,T,cin,cout", [(2, 256, 128, 384), (1, 77, 64, 200), (3, 64, 320, 64)])
def test_f16_output_is_the_rounded_fp32_output(ops, B, T, cin, cout):
    x, w, b, wd, wz, ad, az, sel, xq = _setup(ops, B, T, cin, cout, 5 + cout)
    pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
    y32 = ops.conv2d_w4a8(xq, pw, sel)
    y16 = ops.conv2d_w4a8(xq, pw, sel, out_f16=True)
    assert y16.dtype == torch.float16 and y16.shape == y32.shape
    assert torch.equal(y16, y32.half())


@pytest.mark.parametrize("B,T,cin,inner", [(2, 200, 128, 256), (1, 64, 320, 1280), (2, 77, 64, 64)])
def test_geglu_epilogue_bit_exact(ops, B, T, cin, inner):
    x, w, b, wd, wz, ad, az, sel, xq = _setup(ops, B, T, cin, 2 * inner, 11 + inner)
    pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
    h = ops.conv2d_w4a8(xq, pw, sel)                                   # fp32 [B,T,1,2*inner]
    # consumer quantizer from the unfused fp32 GEGLU output
    gf = ops.geglu(h.reshape(B, T, 2 * inner), None)[1]
    od, oz = O.minmax(gf.cpu(), 256)
    osel = ops.qsel(qtab(od, oz))
    want = ops.geglu(h.reshape(B, T, 2 * inner), osel)[0]
    perm = ops.geglu_perm(inner)
    pwp = ops.pack_w4(w[perm].contiguous().to(DEV), wd.reshape(-1)[perm].contiguous().to(DEV),
                      wz.reshape(-1)[perm].contiguous().to(DEV), bias=b[perm].contiguous().to(DEV))
    got = ops.conv2d_w4a8(xq, pwp, sel, geglu_oq=osel, geglu_exact=True)
    assert got.dtype == torch.int8 and got.shape == (B, T, 1, inner)
    assert torch.equal(got.reshape(B, T, inner), want)
    # and against the oracle's arithmetic on the same fake-quantised operands (bins may differ only where the
    # fp32 GEMM rounding moves a value across a bin edge)
    ref_h = F.conv2d(O.fake_quant(x.permute(0, 3, 1, 2), ad, az, 256), O.fake_quant(w, wd, wz, 16), b)
    ref_h = ref_h.permute(0, 2, 3, 1).reshape(B, T, 2 * inner)
    a, g = ref_h.chunk(2, dim=-1)
    ref_bins = O.quant_index(a * F.gelu(g), od, oz, 256)
    diff = (got.reshape(B, T, inner).cpu().float() + 128 - ref_bins).abs()
    assert diff.max() <= 1 and (diff > 0).float().mean() < 2e-3


@pytest.mark.parametrize("B,H,W,cin,cout,k,stride,pad", [
    (2, 16, 16, 128, 96, 1, 1, (0, 0, 0, 0)),       # skip_connection / nin_shortcut
    (2, 17, 17, 64, 64, 3, 2, (1, 1, 1, 1)),        # Downsample.op: stride 2, ragged
    (1, 8, 8, 320, 4, 3, 1, (1, 1, 1, 1)),          # out.2: narrow Cout
    (3, 8, 8, 2560, 1280, 1, 1, (0, 0, 0, 0)),      # widest SD skip conv, 64x64 tiles
])
def test_f16_input_conv_is_bit_identical_to_fp32_input(ops, B, H, W, cin, cout, k, stride, pad):
    """tfmq_conv_desc.x_f16: the un-quantised convs on fp16 activations (LDS-DMA pipeline) -- the fp32-input path rounds
    its input to fp16 while staging, so both paths multiply the same operands in the same order."""
    gen = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, H, W, cin, generator=gen).to(DEV)
    w = torch.randn(cout, cin, k, k, generator=gen) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=gen) * 0.1
    pf = ops.pack_w_f16(w.to(DEV), b.to(DEV))
    assert ops.f16_dma_ok(cin, k, k)
    y32 = ops.conv2d_f16(x, pf, stride=stride, pad=pad)
    xh = ops.to_half(x)
    assert torch.equal(xh, x.half())
    y16 = ops.conv2d_f16(xh, pf, stride=stride, pad=pad)
    assert torch.equal(y16, y32)
    res = torch.randn(y32.shape, generator=gen).to(DEV)
    assert torch.equal(ops.conv2d_f16(xh, pf, stride=stride, pad=pad, residual=res), ops.conv2d_f16(x, pf, stride=stride, pad=pad, residual=res))


def test_groupnorm_half_outputs(ops):
    gen = torch.Generator().manual_seed(2)
    x1 = torch.randn(2, 8, 8, 64, generator=gen).to(DEV)
    x2 = torch.randn(2, 8, 8, 32, generator=gen).to(DEV)
    gm, bt = torch.randn(96, generator=gen).to(DEV), torch.randn(96, generator=gen).to(DEV)
    _, yf, xc = ops.groupnorm(x1, gm, bt, 1e-5, True, None, x2=x2, want_cat=True)
    _, yh, xh = ops.groupnorm(x1, gm, bt, 1e-5, True, None, x2=x2, want_cat=True, half_out=True)
    assert yh.dtype == torch.float16 and xh.dtype == torch.float16
    assert torch.equal(yh, yf.half()) and torch.equal(xh, xc.half()) and torch.equal(xc, torch.cat([x1, x2], -1))
    # split form (statistics from the producing conv's epilogue)
    qt = torch.tensor([[0.05, 120.0]], device=DEV)
    sel = ops.qsel(qt)
    xq = (torch.randn(2, 8, 8, 64, generator=gen) * 30).clamp(-128, 127).to(torch.int8).to(DEV)
    w = torch.randn(64, 64, 3, 3, generator=gen) * 0.05
    qp = ops.minmax_to_qparam(ops.minmax(w.to(DEV), 64), 16)
    pw = ops.pack_w4(w.to(DEV), qp[:, 0].contiguous(), qp[:, 1].contiguous(), bias=torch.zeros(64, device=DEV))
    y = ops.conv2d_w4a8(xq, pw, sel, pad=(1, 1, 1, 1), want_stats=True)
    assert getattr(y, "_tfmq_stats", None) is not None
    g2, b2 = torch.randn(64, generator=gen).to(DEV), torch.randn(64, generator=gen).to(DEV)
    yq_a, _, xc_a = ops.groupnorm(y, g2, b2, 1e-5, True, sel, want_cat=True)
    yq_b, _, xc_b = ops.groupnorm(y, g2, b2, 1e-5, True, sel, want_cat=True, half_out=True)
    assert torch.equal(yq_a, yq_b) and torch.equal(xc_b, xc_a.half())


# ---- tile shape hint (tfmq_conv_desc.tile) and the measured per-shape selection (ops.set_conv_autotune): the result
# must not depend on the tile shape
@pytest.mark.parametrize("k,res,mode,B,H", [(3, True, "f32", 4, 32), (1, False, "f16", 4, 32), (1, True, "q8", 4, 32),
                                              (3, True, "f32", 64, 8), (3, False, "f32", 256, 4), (1, True, "f32", 16, 16)])
def test_tile_variants_are_bit_identical(ops, k, res, mode, B, H):
    # (8x8, 4x4, 16x16 maps: GroupNorm statistics segments of 64, 16, 128 pixels -- one summation order for every tile)
    W, cin, cout = H, 128, 384
    g = torch.Generator().manual_seed(7 + k)
    x = torch.randn(B, H, W, cin, generator=g) * 1.3 - 0.2
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.2
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    oq = ops.qsel(qtab(0.03, 117.0))
    xq = ops.quantize_act(x.to(DEV), sel)
    pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
    r = torch.randn(B, H, W, cout, generator=g).to(DEV) if res else None
    kw = dict(pad=(k // 2,) * 4, residual=r)
    if mode == "f16":
        kw["out_f16"] = True
    elif mode == "q8":
        kw["out_q8"] = oq
    else:
        kw["want_stats"] = True
    outs = []
    # codes >= 256: the split-K form of a tile kernel, ksplit = code >> 8 workgroups per output tile (tfmq_conv_desc.ksplit); K-steps: 18
    # for the 3x3 layers here, 2 for the pointwise ones
    splits = (1 | 2 << 8, 1 | 5 << 8, 2 | 9 << 8, 4 | 18 << 8, 3 | 3 << 8) if k == 3 else (1 | 2 << 8, 2 | 2 << 8)
    def fits(code):       # the handle's slab workspace: tiles * ksplit * tile elements <= 16 Mi (else TFMQ_ERR_ARG, by design)
        bm, bn = {1: (128, 128), 2: (128, 128), 3: (256, 128), 4: (128, 64)}[code & 0xff]
        return -(-B * H * W // bm) * -(-cout // bn) * bm * bn * (code >> 8) <= (16 << 20)
    splits = tuple(c for c in splits if fits(c))
    for tile in (1, 2, 3, 4) + splits + splits[:1]:          # (a split form twice: the arrival tickets are zero again after a launch)
        ops.set_conv_autotune({})
        try:
            import tfmq_dm_amd.ops as _o
            orig = _o._tune_conv
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            y = ops.conv2d_w4a8(xq, pw, sel, **kw)
            outs.append((y.clone(), y._tfmq_stats[0].clone() if mode == "f32" else None))
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
    for y, st in outs[1:]:
        assert torch.equal(y, outs[0][0])
        if st is not None:   # statistics: per-segment sums in a fixed order inside a tile; segments of 64 here
            assert torch.equal(st, outs[0][1])
    # the measured selection fills its cache with one of the eligible variants and returns the same result
    cache = {}
    ops.set_conv_autotune(cache)
    try:
        y = ops.conv2d_w4a8(xq, pw, sel, **kw)
    finally:
        ops.set_conv_autotune(None)
    assert len(cache) == 1 and (list(cache.values())[0] & 0xff) in (1, 2, 3, 4, 5, 6, 7, 8)
    assert torch.equal(y, outs[0][0])


def test_split_k_small_batch_shapes_are_bit_identical(ops):
    """The launches split-K is for: the 1280-channel 3x3 convs at 8x8 / 16x16 of a UNet(2) forward (1 image under guidance) -- one or four
    128-pixel row tiles, 180 / 360 K-steps.  Every split form, launched repeatedly (slab workspace and tickets are reused), equals the
    unsplit tile kernel bit for bit: fp16-stream output with residual and GroupNorm statistics, and the int8 output mode."""
    import tfmq_dm_amd.ops as _o
    g = torch.Generator().manual_seed(31)
    for (B, H, cin, cout, mode) in ((2, 8, 1280, 1280, "f16"), (2, 16, 2560, 1280, "f16"), (2, 8, 1280, 1280, "q8"), (3, 8, 640, 320, "f32")):
        x = torch.randn(B, H, H, cin, generator=g) * 1.3 - 0.2
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9) ** 0.5)
        b = torch.randn(cout, generator=g) * 0.2
        wd, wz = O.init_channelwise(w, 16, "minmax")
        ad, az = O.minmax(x, 256)
        sel = ops.qsel(qtab(ad, az))
        xq = ops.quantize_act(x.to(DEV), sel)
        pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
        r = torch.randn(B, H, H, cout, generator=g).to(DEV)
        kw = dict(pad=(1, 1, 1, 1))
        if mode == "f16":
            kw.update(out_f16=True, residual=r.half(), want_stats=True)
        elif mode == "q8":
            kw.update(out_q8=ops.qsel(qtab(0.03, 117.0)), residual=r)
        else:
            kw.update(residual=r, want_stats=True)
        ref = None
        for code in (1, 1 | 16 << 8, 2 | 6 << 8, 4 | 12 << 8, 1 | 16 << 8, 1 | 7 << 8, 1 | 16 << 8):
            ops.set_conv_autotune({})
            try:
                orig = _o._tune_conv
                _o._tune_conv = lambda h, name, kind, d, dsc, t=code: t
                y = ops.conv2d_w4a8(xq, pw, sel, **kw)
                st = y._tfmq_stats[0].clone() if hasattr(y, "_tfmq_stats") else None
            finally:
                _o._tune_conv = orig
                ops.set_conv_autotune(None)
            if ref is None:
                ref = (y.clone(), st)
            else:
                assert torch.equal(y, ref[0]), (B, H, cin, mode, code)
                assert st is None or torch.equal(st, ref[1]), (B, H, cin, mode, code)
    # the measured selection may now return a split form; it must still be the same bits
    cache = {}
    ops.set_conv_autotune(cache)
    try:
        y = ops.conv2d_w4a8(xq, pw, sel, **kw)
    finally:
        ops.set_conv_autotune(None)
    assert torch.equal(y, ref[0])
    print("selected for", (B, H, cin, cout), ":", _o.tile_name(list(cache.values())[0]))


@pytest.mark.parametrize("B,H,cin,cout,res,mode", [
    (2, 64, 64, 320, True, "f32"),        # 4 image rows per tile, 320-wide tiles (WN = 5: waves 0-3 stage 3 weight pieces)
    (3, 32, 128, 640, False, "f32"),      # 8 rows per tile, two 320-wide column tiles
    (5, 16, 192, 256, True, "q8"),        # one image per tile, 256-wide tiles, int8 output
    (7, 8, 128, 96, True, "f32"),         # four images per tile (7 images: ragged last tile), 128-wide tile, ragged columns
    (2, 32, 64, 384, True, "f16")])       # 256-wide tiles with a ragged second column tile, fp16 output
def test_slab_kernel_is_bit_identical_to_the_tile_kernels(ops, B, H, cin, cout, res, mode):
    """The 3x3 slab kernel (TFMQ_TILE_SLAB: K order (chunk, tap), activation slab staged once per chunk, 8 waves) against
    the 128x128 kernel on the same launch: int32 sums are exact, the epilogue arithmetic and the statistics order are
    shared, so outputs and GroupNorm statistics must agree bit for bit."""
    import tfmq_dm_amd.ops as _o
    W = H
    g = torch.Generator().manual_seed(31 + H)
    x = torch.randn(B, H, W, cin, generator=g) * 1.3 - 0.2
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.2
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    xq = ops.quantize_act(x.to(DEV), sel)
    pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
    r = torch.randn(B, H, W, cout, generator=g).to(DEV) if res else None
    ra = torch.randn(B, cout, generator=g).to(DEV)
    kw = dict(pad=(1, 1, 1, 1), residual=r, rowadd=ra)
    if mode == "q8":
        kw["out_q8"] = ops.qsel(qtab(0.05, 120.0))
    elif mode == "f16":
        kw.update(out_f16=True, rowadd=None, residual=None)
    else:
        kw["want_stats"] = True
    outs = []
    for tile in (1, 5):
        ops.set_conv_autotune({})
        orig = _o._tune_conv
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            y = ops.conv2d_w4a8(xq, pw, sel, **kw)
            outs.append((y.clone(), y._tfmq_stats[0].clone() if mode == "f32" else None))
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
    assert torch.equal(outs[1][0], outs[0][0])
    if mode == "f32":
        assert torch.equal(outs[1][1], outs[0][1])
    # and against the oracle's conv on the fake-quantised operands
    if mode == "f32":
        import torch.nn.functional as F
        xh = O.fake_quant(x.permute(0, 3, 1, 2).contiguous(), ad, az, 256)
        wh = O.fake_quant(w, wd.reshape(-1, 1, 1, 1), wz.reshape(-1, 1, 1, 1), 16)
        ref = F.conv2d(xh, wh, b, padding=1) + ra.cpu()[:, :, None, None]
        if r is not None:
            ref = ref + r.cpu().permute(0, 3, 1, 2)
        yy = outs[1][0].cpu().permute(0, 3, 1, 2)
        assert float((yy - ref).abs().max() / ref.abs().max()) <= 1e-5


@pytest.mark.parametrize("B,H,cin,cout,mode", [(2, 32, 64, 320, "f32"), (3, 16, 128, 256, "f16"), (5, 8, 192, 640, "f32"), (9, 4, 64, 128, "q8")])
def test_slab_kernel_fused_upsample_is_bit_identical(ops, B, H, cin, cout, mode):
    """Nearest-2x upsample fused into the 3x3 conv (Upsample.conv of the UNets): the slab kernel stages the UPSAMPLED rows
    (virtual pixel (y, x) reads input (y >> 1, x >> 1)); against the 128x128 tile kernel and against upsampling first."""
    import tfmq_dm_amd.ops as _o
    g = torch.Generator().manual_seed(77 + H)
    x = torch.randn(B, H, H, cin, generator=g) * 1.1 + 0.1
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.2
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    xq = ops.quantize_act(x.to(DEV), sel)
    pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
    kw = dict(pad=(1, 1, 1, 1), up2x=True)
    if mode == "q8":
        kw["out_q8"] = ops.qsel(qtab(0.05, 120.0))
    elif mode == "f16":
        kw["out_f16"] = True
    else:
        kw["want_stats"] = True
    outs = []
    for tile in (1, 5):
        ops.set_conv_autotune({})
        orig = _o._tune_conv
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            y = ops.conv2d_w4a8(xq, pw, sel, **kw)
            outs.append((y.clone(), y._tfmq_stats[0].clone() if mode == "f32" else None))
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
    assert torch.equal(outs[1][0], outs[0][0])
    if mode == "f32":
        assert torch.equal(outs[1][1], outs[0][1])
    kw.pop("up2x")
    x2 = xq.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous()
    y2 = ops.conv2d_w4a8(x2, pw, sel, **kw)
    assert torch.equal(y2, outs[1][0])


@pytest.mark.parametrize("k,B,H,cin,cout", [(3, 2, 32, 64, 320), (1, 3, 16, 128, 256), (3, 5, 8, 128, 640), (1, 2, 64, 64, 64)])
def test_f16_stream_epilogue_equals_rounded_f32_epilogue(ops, k, B, H, cin, cout):
    """fp16 activation stream: a conv writing fp16 (+ temb row + fp16 residual, statistics from the fp32 values) must give
    exactly fp16(fp32 result) and the same statistics as the fp32 launch fed the same (fp16-representable) residual --
    for every tile kernel and the slab kernel (8-channel / 16-byte items in the store pass)."""
    import tfmq_dm_amd.ops as _o
    W = H
    g = torch.Generator().manual_seed(77 + H)
    x = torch.randn(B, H, W, cin, generator=g) * 1.3 - 0.2
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.2
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    xq = ops.quantize_act(x.to(DEV), sel)
    pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
    r16 = torch.randn(B, H, W, cout, generator=g).half().to(DEV)
    ra = torch.randn(B, cout, generator=g).to(DEV)
    pad = (k // 2,) * 4
    ref = ops.conv2d_w4a8(xq, pw, sel, pad=pad, residual=r16.float(), rowadd=ra, want_stats=True)
    ref_st = ref._tfmq_stats[0].clone()
    tiles = (1, 2, 3, 4, 5, 7) if k == 3 else (1, 2, 3, 4)
    for tile in tiles:
        ops.set_conv_autotune({})
        orig = _o._tune_conv
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            y = ops.conv2d_w4a8(xq, pw, sel, pad=pad, residual=r16, rowadd=ra, want_stats=True, out_f16=True)
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
        assert y.dtype == torch.float16 and torch.equal(y, ref.half()), tile
        assert torch.equal(y._tfmq_stats[0], ref_st), tile
    # consumers of the fp16 stream read exactly the values their fp32 forms would: GroupNorm from the epilogue statistics
    # (8-channel fp16 kernel), LayerNorm over the channel rows, the plain activation quantizer
    gq = torch.Generator().manual_seed(5)
    g2, b2 = torch.randn(cout, generator=gq).to(DEV), torch.randn(cout, generator=gq).to(DEV)
    y32 = y.float()
    y32._tfmq_stats = y._tfmq_stats
    qa, _, ca = ops.groupnorm(y, g2, b2, 1e-5, True, sel, want_cat=True, half_out=True)
    qb, _, cb = ops.groupnorm(y32, g2, b2, 1e-5, True, sel, want_cat=True, half_out=True)
    assert torch.equal(qa, qb) and torch.equal(ca, cb) and torch.equal(ca, y)
    qa2, fa2 = ops.layernorm(y.reshape(B, H * W, cout), g2, b2, 1e-5, sel, want_f32=True)
    qb2, fb2 = ops.layernorm(y32.reshape(B, H * W, cout), g2, b2, 1e-5, sel, want_f32=True)
    assert torch.equal(qa2, qb2) and torch.equal(fa2, fb2)
    assert torch.equal(ops.quantize_act(y, sel), ops.quantize_act(y32, sel))
    # int8 output with an fp16 residual (16-channel items): the consumer quantizer's bins of the same fp32 values
    oq = ops.qsel(qtab(0.04, 119.0))
    q_ref = ops.quantize_act(ref, oq)
    for tile in tiles:
        ops.set_conv_autotune({})
        orig = _o._tune_conv
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            yq = ops.conv2d_w4a8(xq, pw, sel, pad=pad, residual=r16, rowadd=ra, out_q8=oq)
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
        assert torch.equal(yq, q_ref), tile


@pytest.mark.parametrize("T,cin,cout,mode", [(4096, 320, 320, "f16res"), (1000, 64, 192, "f16"), (520, 1280, 320, "q8res"),
                                              (300, 128, 100, "q8"), (777, 320, 2560, "geglu"), (256, 640, 5120, "geglu"),
                                              (40000, 320, 320, "f16res"), (9000, 320, 2560, "geglu"), (33000, 192, 200, "q8res"),
                                              (20000, 640, 1920, "f16"), (4096, 320, 320, "f16res+stats"), (1024, 640, 640, "f16res+stats"),
                                              (64, 1280, 1280, "f16+stats")])
def test_direct_pointwise_kernel_is_bit_identical_to_the_tile_kernels(ops, T, cin, cout, mode):
    """The pointwise kernel with the register-direct epilogue (TFMQ_TILE_DIRECT: swapped MFMA operands, a lane owns 4
    consecutive channels of one pixel, no LDS staging) against the 128x128 tile kernel: fp16 output (+ fp16 residual),
    int8 output (+ residual), fused GEGLU -> int8; ragged rows and columns."""
    import tfmq_dm_amd.ops as _o
    B = 2
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, T, 1, cin, generator=g) * 1.3 - 0.2
    w = torch.randn(cout, cin, generator=g) * (2.0 / cin ** 0.5)
    b = torch.randn(cout, generator=g) * 0.2
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    oq = ops.qsel(qtab(0.05, 121.0))
    xq = ops.quantize_act(x.to(DEV), sel)
    kw = {}
    if mode == "geglu":
        perm = ops.geglu_perm(cout // 2, DEV)
        pw = ops.pack_w4(w.to(DEV)[perm].contiguous(), wd.to(DEV).reshape(-1)[perm].contiguous(), wz.to(DEV).reshape(-1)[perm].contiguous(),
                         bias=b.to(DEV)[perm].contiguous())
        kw["geglu_oq"] = oq
        kw["geglu_exact"] = True          # (the consumer-sized GELU of round 4 lives on the register-direct kernel only: tests/test_geglu_fast_gpu.py)
    else:
        pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
        if mode.startswith("f16"):
            kw["out_f16"] = True
        else:
            kw["out_q8"] = oq
        if "res" in mode:
            kw["residual"] = torch.randn(B, T, 1, cout, generator=g).half().to(DEV)
        if mode.endswith("+stats"):      # the SpatialTransformer's proj_out: GroupNorm statistics of the consumer (DPP sums)
            kw["want_stats"] = True
    outs, stats = [], []
    for tile in (1, 6, 9):       # the tile kernel, the register-direct pointwise kernel and its 256 x 128 form (round 6; layers with a residual: the 128-row form)
        ops.set_conv_autotune({})
        orig = _o._tune_conv
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            y = ops.conv2d_w4a8(xq, pw, sel, **kw)
            outs.append(y.clone())
            if mode.endswith("+stats"):
                stats.append(y._tfmq_stats[0].clone())
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    if stats:
        assert torch.equal(stats[0], stats[1]) and torch.equal(stats[0], stats[2])
    if mode == "geglu":     # and against the arithmetic spelled out: x * gelu(gate) of the un-fused projection, then the quantizer
        h = ops.conv2d_w4a8(xq, ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV)), sel)
        ref = ops.geglu(h.reshape(B * T, cout), oq)[0].reshape(B, T, 1, cout // 2)
        assert torch.equal(outs[1], ref)


@pytest.mark.parametrize("B,T,C", [(2, 1024, 128), (3, 4096, 320), (1, 64, 640), (2, 100, 128)])
def test_direct_kernel_transposed_region_is_bit_identical(ops, B, T, C):
    """Fused q|k|v projection: q|k as fp16 rows, V transposed ([B, C, T]: the attention kernel's V^T operand) -- the
    register-direct pointwise kernel (lane = pixel: one channel's 32 pixels are 64 contiguous bytes per store) against the
    tile kernel's LDS-staged transposed pass; T a multiple of 4 only."""
    import tfmq_dm_amd.ops as _o
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(B, T, 1, C, generator=g)
    w = torch.randn(3 * C, C, generator=g) * (2.0 / C ** 0.5)
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    xq = ops.quantize_act(x.to(DEV), sel)
    pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=None)
    t0 = 2 * C if (2 * C) % 128 == 0 else 256
    outs = []
    for tile in (1, 6, 9):       # (9: the 256 x 128 form of round 6)
        ops.set_conv_autotune({})
        orig = _o._tune_conv
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            y, yt = ops.conv2d_w4a8(xq, pw, sel, out_f16=True, t_col0=t0)
            outs.append((y[..., :t0].clone(), yt.clone()))
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0])
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][1], outs[2][1])


def test_f16_conv_tile_variants_are_bit_identical(ops):
    """The fp16-activation DMA conv (un-quantised / weight-only layers) in its four tile shapes: same K order per output."""
    import tfmq_dm_amd.ops as _o
    g = torch.Generator().manual_seed(19)
    B, H, W, cin, cout = 8, 32, 32, 64, 320
    x = torch.randn(B, H, W, cin, generator=g).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV)
    pf = ops.pack_w_f16(w, torch.randn(cout, generator=g).to(DEV))
    xh = ops.to_half(x)
    res = torch.randn(B, H, W, cout, generator=g).to(DEV)
    outs = []
    orig = _o._tune_conv
    for tile in (1, 2, 3, 4):
        ops.set_conv_autotune({})
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            y = ops.conv2d_f16(xh, pf, pad=(1, 1, 1, 1), residual=res, want_stats=True)
            outs.append((y.clone(), y._tfmq_stats[0].clone()))
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
    for y, st in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(st, outs[0][1])
    assert torch.equal(outs[0][0], ops.conv2d_f16(x, pf, pad=(1, 1, 1, 1), residual=res))      # == the fp32-input path


@pytest.mark.parametrize("B,T,cin,cout,res,wq", [(2, 4096, 640, 320, False, False), (3, 1000, 96, 200, True, False), (1, 256, 2560, 1280, False, True),
                                                 (4, 64, 320, 648, True, True)])
def test_direct_pointwise_kernel_on_fp16_operands(ops, B, T, cin, cout, res, wq):
    """Un-quantised / weight-only pointwise layers (skip-connection 1x1 convs: fp16 activations written by the producing
    GroupNorm, fp16 weights, f16 MFMA): the register-direct kernel (tile 6) against the LDS-staged tile kernel (tile 1) --
    fp16 output, optional fp16 residual, with and without a weight-only integer grid + scale; ragged rows and columns."""
    import tfmq_dm_amd.ops as _o
    g = torch.Generator().manual_seed(cin + cout)
    x = (torch.randn(B, T, 1, cin, generator=g) * 1.2).to(DEV)
    w = torch.randn(cout, cin, generator=g) * (2.0 / cin ** 0.5)
    b = (torch.randn(cout, generator=g) * 0.2).to(DEV)
    if wq:
        wd, wz = O.init_channelwise(w, 16, "minmax")
        pf = ops.pack_w_f16(w.to(DEV), b, delta=wd.to(DEV), zp=wz.to(DEV))
    else:
        pf = ops.pack_w_f16(w.to(DEV), b)
    xh = ops.to_half(x)
    r16 = torch.randn(B, T, 1, cout, generator=g).half().to(DEV) if res else None
    outs = []
    orig = _o._tune_conv
    for tile in (1, 6):
        ops.set_conv_autotune({})
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            outs.append(ops.conv2d_f16(xh, pf, residual=r16, out_f16=True).clone())
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
    assert outs[0].dtype == torch.float16 and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("B,T,c1,c2,cout,res", [(2, 256, 320, 320, 320, False), (3, 77, 640, 320, 320, True), (1, 300, 1280, 640, 640, False), (2, 64, 32, 96, 64, True)])
def test_f16_pointwise_reads_a_virtual_concat(ops, B, T, c1, c2, cout, res):
    """tfmq_conv_desc.x2: the shortcut conv of an up-path ResBlock (openaimodel.py:771 th.cat([h, hs.pop()], dim=1) ->
    skip_connection) reads its two fp16 sources directly -- bit-identical to the conv over the materialised concat, with and
    without a residual and GroupNorm statistics; launches the kernel cannot take are refused, not computed some other way."""
    from tfmq_dm_amd._lib import TfmqError
    gen = torch.Generator().manual_seed(B * T + c1)
    x1 = (torch.randn(B, T, 1, c1, generator=gen) * 1.2).to(DEV).half()
    x2 = (torch.randn(B, T, 1, c2, generator=gen) * 0.7 + 0.1).to(DEV).half()
    w = (torch.randn(cout, c1 + c2, 1, 1, generator=gen) * 0.05).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    pf = ops.pack_w_f16(w.reshape(cout, c1 + c2), bias)
    kw = {}
    if res:
        kw["residual"] = torch.randn(B, T, 1, cout, generator=gen).to(DEV).half()
    xc = torch.cat([x1, x2], dim=-1).contiguous()
    ref = ops.conv2d_f16(xc, pf, out_f16=True, want_stats=True, **kw)
    got = ops.conv2d_f16(x1, pf, out_f16=True, want_stats=True, x2=x2, **kw)
    assert torch.equal(got, ref)
    assert hasattr(got, "_tfmq_stats") == hasattr(ref, "_tfmq_stats")
    if hasattr(ref, "_tfmq_stats"):
        assert torch.equal(got._tfmq_stats[0], ref._tfmq_stats[0])
    # value check against torch (fp16 operands, fp32 accumulate)
    y = torch.nn.functional.linear(xc.float().reshape(-1, c1 + c2), w.reshape(cout, -1).half().float(), bias).reshape(B, T, 1, cout)
    if res:
        y = y + kw["residual"].float()
    assert float((got.float() - y).abs().max()) <= 4e-3 * float(y.abs().max())
    with pytest.raises(TfmqError):
        ops.conv2d_f16(x1, pf, x2=x2)                       # fp32 output: not a launch of the pointwise kernel
    with pytest.raises(TfmqError):
        ops.conv2d_f16(x1.float(), pf, out_f16=True, x2=x2)  # fp32 first source


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 16, 4, 320), (3, 8, 12, 3, 128), (1, 32, 32, 4, 64), (2, 5, 7, 7, 96)])
def test_narrow_input_conv_as_im2col_gemm(ops, B, H, W, cin, cout):
    """The UNet's first conv (conv_in, ddim/models/diffusion.py:310; input_blocks.0.0, openaimodel.py:502-506) in the fp16 stream:
    tfmq_im2col_f16 rows + the pointwise fp16 kernel.  Same fp16 operand values and fp32 accumulation as the 3x3 tile kernel in
    another K order: equal to it within fp32 summation noise (<= 2e-6 of the output scale before the fp16 rounding, i.e. at
    most one fp16 ulp apart), and to torch's conv on the fp16-rounded operands within 1e-3."""
    gen = torch.Generator().manual_seed(B * H + cin)
    x = (torch.randn(B, H, W, cin, generator=gen) * 1.1).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=gen) * 0.2).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    pf = ops.pack_w_f16(w, bias)
    g = ops.narrow_conv_as_gemm(pf)
    assert g is not None and g.cin % 32 == 0 and g.cin >= 9 * cin
    col = ops.im2col_f16(x, 3, 3, 1, 1, g.cin)
    # the rows are the zero-padded patches, fp16-rounded, (tap, channel) order
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1))
    pat = torch.stack([xp[:, :, dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)], dim=1)      # [B, 9, C, H, W]
    ref_col = pat.permute(0, 3, 4, 1, 2).reshape(B, H, W, 9 * cin).half()
    assert torch.equal(col[..., :9 * cin], ref_col) and not bool(col[..., 9 * cin:].any())
    y_gemm = ops.conv2d_f16(col, g, out_f16=True, want_stats=True)
    y_tile = ops.conv2d_f16(x, pf, pad=(1, 1, 1, 1), out_f16=True, want_stats=True)
    ref = torch.nn.functional.conv2d(x.half().float().permute(0, 3, 1, 2), w.half().float(), bias, padding=1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    assert float((y_gemm.float() - ref).abs().max()) <= 1e-3 * scale
    assert float((y_gemm.float() - y_tile.float()).abs().max()) <= 1.0e-3 * scale       # one fp16 ulp at the output scale
    assert float((y_gemm.float() - y_tile.float()).abs().mean()) <= 2e-5 * scale        # ... and rarely: most values identical
    if hasattr(y_tile, "_tfmq_stats"):
        a, b = y_gemm._tfmq_stats[0], y_tile._tfmq_stats[0]
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 16, 320, 4), (3, 8, 12, 128, 3), (1, 32, 32, 64, 4), (2, 5, 7, 96, 1)])
def test_narrow_output_conv_as_gemm_plus_tap_gather(ops, B, H, W, cin, cout):
    """The UNet's last conv (conv_out, ddim/models/diffusion.py:353; out.2, openaimodel.py:700-704): one pointwise fp16 GEMM to the
    kh*kw*cout per-tap partial sums (fp32) + tfmq_tap_gather_sum.  Same fp16 operands, fp32 accumulation per tap, taps added in
    ascending order: equal to the 3x3 tile kernel within fp32 summation noise and to torch's conv on the rounded operands."""
    gen = torch.Generator().manual_seed(B * W + cin)
    x = (torch.randn(B, H, W, cin, generator=gen) * 0.9).to(DEV).half()
    w = (torch.randn(cout, cin, 3, 3, generator=gen) * 0.05).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    pf = ops.pack_w_f16(w, bias)
    g = ops.narrow_out_conv_as_gemm(pf)
    assert g is not None and g.cout % 8 == 0 and g.cout >= 9 * cout
    y9 = ops.conv2d_f16(x, g)
    got = ops.tap_gather_sum(y9, 3, 3, cout, 1, 1, pf.bias)
    tile = ops.conv2d_f16(x, pf, pad=(1, 1, 1, 1))
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.half().float(), bias, padding=1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * scale
    assert float((got - tile).abs().max()) <= 2e-5 * scale
    # the gather itself, on arbitrary partial sums, against a plain restatement
    y = torch.randn(B, H, W, g.cout, generator=gen).to(DEV)
    yp = torch.nn.functional.pad(y.permute(0, 3, 1, 2), (1, 1, 1, 1))
    want = bias.view(1, 1, 1, cout).expand(B, H, W, cout).clone()
    for tap in range(9):
        dy, dx = tap // 3, tap % 3
        want = want + yp[:, tap * cout:(tap + 1) * cout, dy:dy + H, dx:dx + W].permute(0, 2, 3, 1)
    assert torch.equal(ops.tap_gather_sum(y, 3, 3, cout, 1, 1, bias), want)


@pytest.mark.parametrize("B,H,W,cin,cout,res,o16,up", [(2, 16, 16, 64, 320, False, True, False), (4, 8, 8, 96, 128, True, True, False),
                                                      (1, 32, 32, 32, 64, False, False, False), (2, 16, 16, 64, 640, True, False, False),
                                                      (2, 8, 8, 64, 320, False, True, True)])
def test_f16_slab_kernel_vs_tile_kernel(ops, B, H, W, cin, cout, res, o16, up):
    """The 3x3 slab kernel on fp16 operands (un-quantised / weight-only layers: the first ResBlock conv of the SD UNet, every 3x3
    conv of the FP passes that generate and capture calibration data).  Same fp16 operand values and fp32 accumulation as the
    tile kernel in (channel chunk, tap) instead of (tap, channel chunk) order: equal within fp32 summation noise; GroupNorm
    statistics, fp16 residual, fp16 / fp32 output, fused nearest-2x upsample."""
    import tfmq_dm_amd.ops as _o
    gen = torch.Generator().manual_seed(H * W + cin + cout)
    x = (torch.randn(B, H, W, cin, generator=gen) * 0.8).to(DEV).half()
    w = (torch.randn(cout, cin, 3, 3, generator=gen) * (1.5 / (9 * cin) ** 0.5)).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    pf = ops.pack_w_f16(w, bias)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    kw = dict(pad=(1, 1, 1, 1), out_f16=o16, want_stats=True, up2x=up)
    if res:
        kw["residual"] = torch.randn(B, Ho, Wo, cout, generator=gen).to(DEV).half()
    outs = {}
    for tile in (1, 5):
        ops.set_conv_autotune({})
        orig = _o._tune_conv
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            outs[tile] = ops.conv2d_f16(x, pf, **kw)
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
    xin = x.float().permute(0, 3, 1, 2)
    if up:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(xin, w.half().float(), bias, padding=1).permute(0, 2, 3, 1)
    if res:
        ref = ref + kw["residual"].float()
    scale = float(ref.abs().max())
    a, b = outs[5].float(), outs[1].float()
    tol = 1.0e-3 if o16 else 2e-6          # one fp16 ulp at the output scale / fp32 summation noise
    assert float((a - ref).abs().max()) <= (1.5e-3 if o16 else 2e-5) * scale
    assert float((a - b).abs().max()) <= tol * scale
    sa, sb = outs[5]._tfmq_stats[0], outs[1]._tfmq_stats[0]
    assert float((sa - sb).abs().max()) <= 1e-5 * float(sb.abs().max())


@pytest.mark.parametrize("B,H,W,cin,cout,k,res", [(2, 16, 16, 224, 224, 3, True), (3, 8, 8, 96, 128, 3, False), (2, 20, 1, 160, 320, 1, True),
                                                  (1, 32, 32, 224, 448, 1, False), (2, 16, 16, 672, 224, 3, False)])
def test_w4a8_k_padded_operand_cin_32_mod_64(ops, B, H, W, cin, cout, k, res):
    """Cin % 64 == 32 (the 224-channel multiples of the LDM-4 CelebA UNet): tfmq_expand_w4_k64 pads every tap's channels to 64-channel
    K-steps with zero weights, so the LDS-DMA kernels (slab, register-direct pointwise, DMA tile kernels) take the layer; their last
    K-step of a pixel reads 32 bytes of the NEXT pixel, times zero.  Integer sums: bit-identical to the register-staged kernel with
    32-channel steps (TFMQ_W4_KPAD=0 packs without the padded operand), whatever the tile."""
    import tfmq_dm_amd.ops as _o
    gen = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(B, H, W, cin, generator=gen) * 1.1
    w = torch.randn(cout, cin, k, k, generator=gen) * (1.5 / (k * k * cin) ** 0.5)
    bias = torch.randn(cout, generator=gen) * 0.1
    wd, wz = O.init_channelwise(w, 16, "minmax")
    ad, az = O.minmax(x, 256)
    sel = ops.qsel(qtab(ad, az))
    xq = ops.quantize_act(x.to(DEV), sel)
    os.environ["TFMQ_W4_KPAD"] = "0"
    try:
        pw_ref = ops.pack_w4(w.to(DEV), wd.reshape(-1).to(DEV), wz.reshape(-1).to(DEV), None, bias.to(DEV))
    finally:
        del os.environ["TFMQ_W4_KPAD"]
    pw = ops.pack_w4(w.to(DEV), wd.reshape(-1).to(DEV), wz.reshape(-1).to(DEV), None, bias.to(DEV))
    assert pw_ref.w8p is None and pw.w8p is not None
    pad = (1, 1, 1, 1) if k == 3 else (0, 0, 0, 0)
    kw = dict(pad=pad, want_stats=True, out_f16=True)
    if res:
        kw["residual"] = torch.randn(B, H, W, cout, generator=gen).to(DEV).half()
    ref = ops.conv2d_w4a8(xq, pw_ref, sel, **kw)
    tiles = (1, 2, 4, 5, 7) if k == 3 else (1, 2, 4, 6)
    for tile in tiles:
        ops.set_conv_autotune({})
        orig = _o._tune_conv
        try:
            _o._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            y = ops.conv2d_w4a8(xq, pw, sel, **kw)
        finally:
            _o._tune_conv = orig
            ops.set_conv_autotune(None)
        assert torch.equal(y, ref), tile
        if hasattr(ref, "_tfmq_stats"):
            assert torch.equal(y._tfmq_stats[0], ref._tfmq_stats[0]), tile
    # the last pixel's padded K-step reads past the tensor: a tensor that ends exactly at its last channel gives the same result
    # whatever follows it in memory
    big = torch.full((xq.numel() + 4096,), 77, dtype=torch.int8, device=DEV)
    big[:xq.numel()] = xq.reshape(-1)
    y2 = ops.conv2d_w4a8(big[:xq.numel()].view(xq.shape), pw, sel, **kw)
    assert torch.equal(y2, ref)
