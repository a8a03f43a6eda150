"""GradTape's view classification (engine/fisher.py: _root) on plain CPU tensors: which views of a taped tensor are differentiable column
slices, which are refused.  (save_grad, quant/data_utill.py:191-256, needs dL/d(unit output) through fused q|k|v / k|v buffers.)"""
import pytest
import torch

from tfmq_dm_amd._lib import TfmqError
from tfmq_dm_amd.engine.fisher import _root


def test_column_slices_and_their_roots():
    base = torch.zeros(4, 7, 12)
    b, off, rowlen = _root(base)
    assert b is base and off == 0 and rowlen == 12
    b, off, rowlen = _root(base[..., 4:8])
    assert b.data_ptr() == base.data_ptr() and off == 4 and rowlen == 12
    r = base.reshape(28, 12)                      # a reshape of a contiguous tensor is its own root
    b, off, rowlen = _root(r)
    assert b is r and off == 0 and rowlen == 12


def test_single_row_column_slice_is_not_a_batch_slice():
    """is_contiguous() ignores size-1 dimensions, so kv[..., :C] of a [1, 1, 2C] fused k|v projection (batch 1, one context
    token) is 'contiguous and smaller than its base' -- a column slice all the same."""
    kv = torch.zeros(1, 1, 16)
    for lo, hi in ((0, 8), (8, 16)):
        v = kv[..., lo:hi]
        assert v.is_contiguous() and v.numel() < kv.numel()
        b, off, rowlen = _root(v)
        assert b.data_ptr() == kv.data_ptr() and off == lo and rowlen == 16
    kvb = torch.zeros(3, 1, 16)                    # batch 3, one token: not contiguous, the row stride of the size-1 dimension is arbitrary
    b, off, rowlen = _root(kvb[..., 8:])
    assert off == 8 and rowlen == 16


def test_batch_slices_are_refused_loudly():
    base = torch.zeros(4, 7, 12)
    with pytest.raises(TfmqError):
        _root(base[:2])
    with pytest.raises(TfmqError):
        _root(base[2:])
    with pytest.raises(TfmqError):
        _root(base[..., ::2])                      # strided columns
