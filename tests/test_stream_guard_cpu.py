"""Decision rule of the fp16-stream guard (ddim/sampler.py: streams_disagree): pure function over the (fp16-stream eps, fp32-stream eps) pairs
of the probed steps -- first AND last step of the trajectory since round 5.  The integration (fallback to the fp32 stream, re-sampling, the
loud error when even that overflows) is tests/test_engine_ldm_gpu.py::test_fp16_stream_overflow_falls_back_to_the_fp32_stream."""
import torch


def test_streams_disagree_rule():
    from tfmq_dm_amd.ddim.sampler import STREAM_GUARD_REL_L2, streams_disagree
    g = torch.Generator().manual_seed(0)
    e = torch.randn(2, 8, 8, 4, generator=g)
    near = e + 0.03 * torch.randn(e.shape, generator=g)          # bin flips: a few per cent apart
    far = e + 0.5 * torch.randn(e.shape, generator=g)
    inf = e.clone()
    inf[0, 0, 0, 0] = float("inf")
    assert STREAM_GUARD_REL_L2 == 0.25
    assert not streams_disagree([])
    assert not streams_disagree([(near, e)]) and not streams_disagree([(near, e), (near, e)])
    assert streams_disagree([(far, e)])
    assert streams_disagree([(near, e), (far, e)])               # only the LAST step disagrees: caught since both ends are probed
    assert streams_disagree([(inf, e)]) and streams_disagree([(near, e), (inf, e)])
    assert not streams_disagree([(near, inf)])                   # the fp32 probe itself is not finite: not the fp16 stream's doing


def test_guard_probes_both_ends_of_the_trajectory(monkeypatch):
    """fp16_stream_overflowed with a stand-in sampler: the probes run at step 0 and at step n - 1, each with both streams; a disagreement at
    the last step alone switches the engine to the fp32 stream."""
    import contextlib
    import types
    import tfmq_dm_amd.ddim.sampler as S
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(S.ops, "use_arena", lambda a: contextlib.nullcontext())
    monkeypatch.setattr(S.ops, "Arena", lambda: "fresh-arena")
    eng = types.SimpleNamespace(stream_f16=True)
    sp = types.SimpleNamespace(eng=eng, stream=types.SimpleNamespace(synchronize=lambda: None), x=torch.zeros(2, 4), step=torch.zeros(1, dtype=torch.int32),
                               n_steps=7, arena="old", gid=5)
    seen = []
    base = torch.ones(2, 4)

    def probe():
        seen.append(("first", int(sp.step), eng.stream_f16))
        return base.clone()

    def probe_last():
        seen.append(("last", int(sp.step), eng.stream_f16))
        return base.clone() if not eng.stream_f16 else base * 3.0      # the fp16 stream is far off at the last step only
    sp.step.fill_(7)
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert S.fp16_stream_overflowed(sp, probe, probe_last) is True
    assert seen == [("first", 0, True), ("first", 0, False), ("last", 6, True), ("last", 6, False)]
    assert eng.stream_f16 is False and sp.gid is None and sp.arena == "fresh-arena" and int(sp.step) == 7
    assert any("fp16 activation stream" in str(x.message) for x in w)
    assert S.fp16_stream_overflowed(sp, probe, probe_last) is False          # checked once per sampler
    # agreement at both ends: nothing changes
    eng2 = types.SimpleNamespace(stream_f16=True)
    sp2 = types.SimpleNamespace(eng=eng2, stream=sp.stream, x=torch.zeros(2, 4), step=torch.zeros(1, dtype=torch.int32), n_steps=3, arena="old", gid=5)
    assert S.fp16_stream_overflowed(sp2, lambda: base.clone(), lambda: base.clone()) is False
    assert eng2.stream_f16 is True and sp2.gid == 5
