"""CPU-only: the C-ABI library loads and exports every symbol include/tfmq_hip.h declares;
the product path fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import tfmq_dm_amd._lib as L
    if not os.path.exists(L.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("tfmq_build", os.path.join(ROOT, "tfmq-dm_amd", "build.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        m.build(verbose=False)
    return L


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "tfmq_hip.h")).read()
    declared = set(re.findall(r"\b(tfmq_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tfmq_ctx"}
    assert declared == set(lib.EXPORTED_SYMBOLS), declared ^ set(lib.EXPORTED_SYMBOLS)
    l = lib.load()
    for name in declared:
        assert hasattr(l, name), name
    assert l.tfmq_abi_version() == 9


def test_docs_quote_the_real_abi_size_and_version(lib):
    """DESIGN.md / README.md state the number of entry points and the ABI version; the round-4 review found them two versions stale."""
    hdr = open(os.path.join(ROOT, "include", "tfmq_hip.h")).read()
    n = len(set(re.findall(r"\b(tfmq_[a-z0-9_]+)\s*\(", hdr)) - {"tfmq_ctx"})
    v = lib.load().tfmq_abi_version()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    readme = open(os.path.join(ROOT, "README.md")).read()
    assert f'({n} `extern "C"` entry points, `tfmq_abi_version()` = {v}' in design
    assert f"C ABI version {v}, {n} entry points" in readme


def test_struct_layouts_match_header(lib):
    import ctypes as C
    # tfmq_qsel: 2 pointers + 2 int32; tfmq_conv_desc / tfmq_gn_desc sizes as laid out by the C compiler
    assert C.sizeof(lib.QSel) == 24
    assert C.sizeof(lib.ConvDesc) == 13 * 4 + 4 + 5 * 8 + 24 + 2 * 8 + 8 + 2 * 8 + 8 + 16 + 24 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8   # ... + x2 + cin1 (padded) + w64 + ksplit (padded)
    assert C.sizeof(lib.GnDesc) == 16 + 32 + 12 + 4 + 24 + 24 + 8
    assert C.sizeof(lib.FfDesc) == 16 + 3 * 8 + 8 + 24 + 4 * 8 + 24 + 4 * 8 + 8 + 24 + 8 + 8 + (5 * 8 + 24 + 2 * 8) + (7 * 8 + 8)      # tfmq_ff_desc (round 4): + the Linears in front / behind
    assert C.sizeof(lib.ChainGemm) == 104 and C.sizeof(lib.ChainDesc) == 384      # tfmq_chain_gemm / tfmq_chain_desc as g++ lays them out


def test_cpu_tensor_is_refused_loudly(lib):
    import tfmq_dm_amd.ops as ops
    with pytest.raises(lib.TfmqError, match="no CPU fallback"):
        ops.minmax(torch.zeros(8))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_no_device_no_handle(lib):
    with pytest.raises(lib.TfmqError):
        lib.Handle(0)


def test_slab_kernel_launch_geometry_rule(lib):
    """Host-side rule deciding whether the 3x3 slab kernel can take a launch (ops.slab_ok mirrors launch_conv_slab): 256-pixel
    tiles made of whole image rows or whole images, a slab of at most 512 pixel rows, also with the fused upsample."""
    import tfmq_dm_amd.ops as ops

    def desc(H, W, cin=320, up=0, k=3, stride=1, pad=1, out_mode=1):
        d = lib.ConvDesc()
        d.H, d.W, d.Cin, d.KH, d.KW, d.stride, d.up2x, d.pad_t, d.pad_l, d.out_mode = H, W, cin, k, k, stride, up, pad, pad, out_mode
        d.Ho, d.Wo = (2 * H, 2 * W) if up else (H, W)
        return d
    assert ops.slab_ok(desc(64, 64)) and ops.slab_ok(desc(32, 32)) and ops.slab_ok(desc(16, 16)) and ops.slab_ok(desc(8, 8))
    assert ops.slab_ok(desc(32, 32, up=1)) and ops.slab_ok(desc(8, 8, up=1))          # Upsample convs: upsampled rows are staged
    assert not ops.slab_ok(desc(64, 64, cin=96))                                       # Cin % 64
    assert not ops.slab_ok(desc(64, 64, k=1, pad=0))                                   # pointwise layers go to the direct kernel
    assert not ops.slab_ok(desc(64, 64, stride=2))
    assert not ops.slab_ok(desc(24, 24))                                               # 576 pixels: neither 256 | HW nor HW | 256
    assert not ops.slab_ok(desc(128, 128))                                             # (2 + 2) * 130 = 520 slab rows > 512
    assert not ops.slab_ok(desc(64, 64, out_mode=2))                                   # GEGLU output is a pointwise mode
    # the 128-pixel form (TFMQ_TILE_SLAB128): slabs of at most 320 rows
    assert all(ops.slab_ok(desc(r, r), 128) for r in (64, 32, 16, 8, 4)) and ops.slab_ok(desc(32, 32, up=1), 128)
    assert not ops.slab_ok(desc(128, 128), 128)                                        # (1 + 2) * 130 = 390 slab rows > 320
    assert not ops.slab_ok(desc(24, 24), 128)
    assert set(ops._TILE_NAMES) == {1, 2, 3, 4, 5, 6, 7, 9}      # (9: the 256 x 128 form of the register-direct pointwise kernel, round 6)


def test_split_k_candidate_rule(lib):
    """Host-side rule that offers split-K forms of the w4a8 tile kernels to the per-shape measurement (ops._ksplit_candidates): only
    launches whose output grid leaves CUs idle, at least three K-steps per slice, between 160 and 1280 workgroups, and never more slab
    elements than the handle's 64 MiB workspace holds (the launcher would answer TFMQ_ERR_ARG)."""
    import tfmq_dm_amd.ops as ops

    def desc(B, H, cin, cout, k=3, out_mode=1):
        d = lib.ConvDesc()
        d.B, d.H, d.W, d.Ho, d.Wo, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.out_mode = B, H, H, H, H, cin, cout, k, k, 1, out_mode
        return d
    tiles = {1: (128, 128), 4: (128, 64), 2: (64, 64)}
    for d in (desc(2, 8, 1280, 1280), desc(2, 16, 2560, 1280), desc(2, 32, 640, 640), desc(2, 64, 320, 320), desc(2, 16, 1280, 1280, k=1)):
        cands = ops._ksplit_candidates("w4a8", d)
        assert cands, (d.H, d.Cin)
        M, nsteps = d.B * d.Ho * d.Wo, d.KH * d.KW * ((d.Cin + 63) // 64)
        for c in cands:
            bm, bn = tiles[c & 0xff]
            ks = c >> 8
            nb = -(-M // bm) * -(-d.Cout // bn)
            assert ks >= 2 and nsteps // ks >= 3 and 160 <= nb * ks <= 1280 and nb * ks * bm * bn <= (16 << 20)
    assert ops._ksplit_candidates("w4a8", desc(128, 64, 320, 320)) == []          # the metric batch: the grid fills the chip
    assert ops._ksplit_candidates("f16", desc(2, 8, 1280, 1280)) == []            # fp32 sums: the K order is part of the result
    assert ops._ksplit_candidates("w4a8", desc(2, 8, 1280, 2560, k=1, out_mode=2)) == []      # fused GEGLU epilogue pairs columns inside a tile
    assert ops.tile_name(2 | 12 << 8) == "64x64, split-K 12" and ops.tile_name(5).startswith("slab")
