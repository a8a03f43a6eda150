"""Nothing on the launch path may reach for device 0 when the tensors live on another GPU (one process per GPU: rank k works on cuda:k).

Round 5 found the conv autotuner reading its timing events through `handle(0)`: on a rank whose device is not 0 that creates a stray handle
on GPU 0 and fails with "event_elapsed: bad id".  No multi-GPU box is available to the suite, so the host logic is driven here with fake
handles for device 3 and every handle request is recorded."""
import ctypes as C

import torch


class FakeHandle:
    def __init__(self, dev):
        self.dev, self.n, self.calls = dev, 0, []

    def call(self, name, *a):
        self.calls.append(name)
        if name == "event_create":
            a[0]._obj.value = self.n
            self.n += 1
        elif name == "event_elapsed_ms":
            assert 0 <= a[0] < self.n and 0 <= a[1] < self.n, "event ids read on a handle that does not own them"
            a[2]._obj.value = 1.0 + 0.01 * len(self.calls)


def _fakes(monkeypatch):
    import tfmq_dm_amd._lib as _lib
    import tfmq_dm_amd.ops as ops
    asked, made = [], {}

    def fake(dev=0, gemm_precision=0):
        asked.append((dev, gemm_precision))
        return made.setdefault((dev, gemm_precision), FakeHandle(dev))
    monkeypatch.setattr(ops, "_lib_handle", fake)
    monkeypatch.setattr(_lib, "handle", fake)
    monkeypatch.setattr(ops, "_stream", lambda d: None)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    return ops, asked, made


def _desc(ops):
    d = ops.ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.up2x, d.out_mode = 2, 16, 16, 320, 320, 1, 1, 1, 0, 1
    return d


def test_autotune_times_on_the_launch_device(monkeypatch):
    ops, asked, made = _fakes(monkeypatch)
    monkeypatch.setattr(ops, "_AUTOTUNE", {})
    t = ops._tune_conv(ops.handle(3), "conv2d_w4a8", "w4a8", 3, _desc(ops))
    assert {d for d, _ in asked} == {3}, asked
    assert "event_elapsed_ms" in made[(3, 0)].calls and ((t & 0xff) in (1, 2, 3, 4, 5, 6, 7))


def test_profile_events_live_on_the_base_handle_of_the_launch_device(monkeypatch):
    ops, asked, made = _fakes(monkeypatch)
    monkeypatch.setattr(ops, "_AUTOTUNE", None)
    rec = []
    monkeypatch.setattr(ops, "_conv_prof", rec)
    with ops.gemm_precision("bf16x3", 3):                      # a launch from inside a precision context: events still on (3, 0)
        ops._profiled_conv("conv2d_w4a8", "w4a8", 3, _desc(ops), 1.0)
    assert len(rec) == 1 and {d for d, _ in asked} == {3}
    assert made[(3, 0)].calls.count("event_create") == 2 and "conv2d_w4a8" in made[(3, 1)].calls
    assert ops.event_elapsed_ms(rec[0][0], rec[0][1]) > 0          # default device = the current one (3), base handle
    assert {d for d, _ in asked} == {3}


def test_gemm_precision_selects_a_handle_and_toggles_nothing(monkeypatch):
    ops, asked, made = _fakes(monkeypatch)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    assert ops.handle(3) is made[(3, 0)]
    with ops.gemm_precision("bf16x3", 3):
        assert ops.handle(3) is made[(3, 1)]
        with ops.gemm_precision("f32", 3):
            assert ops.handle(3) is made[(3, 0)]
        assert ops.handle(3) is made[(3, 1)]
    assert ops.handle(3) is made[(3, 0)]
    assert all("set_gemm_precision" not in h.calls for h in made.values())      # (the real _lib.handle sets it once, at creation)


def test_lib_handle_sets_the_precision_once_at_creation(monkeypatch):
    """_lib.handle(device, precision): one handle per (device, precision); tfmq_set_gemm_precision is called on it exactly once, when it is
    created, never on the base handle."""
    import tfmq_dm_amd._lib as _lib
    made = []

    class H(FakeHandle):
        def __init__(self, dev):
            super().__init__(dev)
            made.append(self)
    monkeypatch.setattr(_lib, "Handle", H)
    monkeypatch.setattr(_lib, "_handles", {})
    monkeypatch.setattr(_lib, "_prec_handles", {})
    base = _lib.handle(2)
    p1 = _lib.handle(2, 1)
    assert _lib.handle(2) is base and _lib.handle(2, 1) is p1 and _lib.handle(2, 0) is base and p1 is not base
    p2 = _lib.handle(2, 2)
    other = _lib.handle(5, 1)
    assert len(made) == 4 and len({id(h) for h in (base, p1, p2, other)}) == 4
    assert base.calls == [] and p1.calls == ["set_gemm_precision"] and p2.calls == ["set_gemm_precision"] and other.dev == 5
    assert (p1.gemm_precision, p2.gemm_precision) == (1, 2)
    assert set(_lib._handles) == {2}                      # linklink.comm_device walks the base handles only
