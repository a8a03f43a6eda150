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
    assert l.tfmq_abi_version() == 7


def test_struct_layouts_match_header(lib):
    import ctypes as C
    # tfmq_qsel: 2 pointers + 2 int32; tfmq_conv_desc / tfmq_gn_desc sizes as laid out by the C compiler
    assert C.sizeof(lib.QSel) == 24
    assert C.sizeof(lib.ConvDesc) == 13 * 4 + 4 + 5 * 8 + 24 + 2 * 8 + 8 + 2 * 8 + 8 + 16 + 24 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8   # ... + x2 + cin1 (padded) + w64 + ksplit (padded)
    assert C.sizeof(lib.GnDesc) == 16 + 32 + 12 + 4 + 24 + 24 + 8


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
    assert set(ops._TILE_NAMES) == {1, 2, 3, 4, 5, 6, 7}
