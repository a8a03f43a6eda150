"""Reconstruction-cache placement helpers of quant/data_utill.py that need no GPU (round 6): the cap on pinned host memory and the
opt-in fp16 device cache of last resort (HalfRows)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))


def test_host_room_honours_the_cap(monkeypatch):
    import tfmq_dm_amd.quant.data_utill as D
    monkeypatch.setenv("TFMQ_CACHE_HOST_MAX_GB", "1")
    assert 0 <= D.host_room() <= (1 << 30)
    monkeypatch.setenv("TFMQ_CACHE_HOST_MAX_GB", "0")
    assert D.host_room() == 0
    monkeypatch.delenv("TFMQ_CACHE_HOST_MAX_GB")
    assert D.host_room() <= (64 << 30)


def test_half_rows_round_trip_and_inexact_count():
    import tfmq_dm_amd.quant.data_utill as D
    g = torch.Generator().manual_seed(3)
    exact = torch.randn(6, 4, 5, generator=g).half().float()          # values of the fp16 activation stream: the narrowing changes nothing
    r = D.HalfRows(exact.shape, torch.float32, "cpu")
    r.fill(0, exact[:4])
    r.fill(4, exact[4:])
    assert r.inexact == 0 and len(r) == 6 and r.size(1) == 4
    idx = torch.tensor([5, 0, 3])
    out = r.index_select(0, idx)
    assert out.dtype == torch.float32 and torch.equal(out, exact[idx])
    rough = torch.randn(3, 8, generator=g) * 1.000123
    r2 = D.HalfRows(rough.shape, torch.float32, "cpu")
    r2.fill(0, rough)
    assert r2.inexact == int((rough.half().float() != rough).sum()) > 0
    assert torch.equal(r2.index_select(0, torch.tensor([2])), rough[2:3].half().float())
