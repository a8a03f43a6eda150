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
    assert l.tfmq_abi_version() == 2


def test_struct_layouts_match_header(lib):
    import ctypes as C
    # tfmq_qsel: 2 pointers + 2 int32; tfmq_conv_desc / tfmq_gn_desc sizes as laid out by the C compiler
    assert C.sizeof(lib.QSel) == 24
    assert C.sizeof(lib.ConvDesc) == 13 * 4 + 4 + 5 * 8 + 24 + 2 * 8 + 8 + 2 * 8 + 8 + 16 + 24 + 8 + 8 + 8 + 8
    assert C.sizeof(lib.GnDesc) == 16 + 32 + 12 + 4 + 24 + 24 + 8


def test_cpu_tensor_is_refused_loudly(lib):
    import tfmq_dm_amd.ops as ops
    with pytest.raises(lib.TfmqError, match="no CPU fallback"):
        ops.minmax(torch.zeros(8))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_no_device_no_handle(lib):
    with pytest.raises(lib.TfmqError):
        lib.Handle(0)
