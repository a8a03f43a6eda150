"""Round 6: reconstruction caches in pinned host memory (quant/data_utill.py: HostRows) -- the reference's `keep_gpu=False`
(quant/calibration.py:62-67 for the widest Stable-Diffusion units; quant/data_utill.py:39-46 keeps the cache on the CPU and
quant/reconstruction.py:66,184 moves each mini-batch to the device).  Where a cache lives must not change a single bit: the rows the
iterations see are the same fp32 values either way, and the host RNG stream that draws the mini-batches is untouched."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_host_rows_answer_index_select_like_a_device_tensor():
    from quant.data_utill import HostRows
    gen = torch.Generator().manual_seed(3)
    full = torch.randn(37, 5, 6, 7, generator=gen)
    hr = HostRows(full.shape, full.dtype, DEV)
    for i in range(0, 37, 8):                        # filled batch by batch from DEVICE tensors, as save_inout does
        hr.fill(i, full[i:i + 8].to(DEV))
    assert hr.size(0) == 37 and tuple(hr.shape) == tuple(full.shape) and hr.buf.is_pinned()
    idx_host = torch.randperm(37, generator=gen)[:8]
    idx = idx_host.to(DEV)
    a = hr.index_select(0, idx)                      # no host copy attached: indices come back from the device
    idx._host = idx_host
    b = hr.index_select(0, idx)                      # the path reconstruction._run takes
    torch.cuda.synchronize()
    ref = full.to(DEV).index_select(0, idx)
    assert a.device.type == "cuda" and torch.equal(a, ref) and torch.equal(b, ref)


def _build(g):
    import tfmq_dm_amd.ddim.models as M
    from quant.quant_layer import QMODE, Scaler
    from quant.quant_model import QuantModel
    m = M.Model(M.make_config(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16, dropout=0.0))
    m.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")})
    m.to(DEV).eval()
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    return QuantModel(m, wq, aq, cali=True, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).to(DEV).eval()


def _calibrate(g, monkeypatch, host):
    from quant.calibration import cali_model
    from quant.reconstruction_util import RLOSS
    if host:
        monkeypatch.setenv("TFMQ_CACHE_HOST", "all")
    else:
        monkeypatch.delenv("TFMQ_CACHE_HOST", raising=False)
    qnn = _build(g)
    xs, ts = T(g["cali_x"]), T(g["cali_t"])
    torch.manual_seed(5)
    np.random.seed(5)
    return cali_model(qnn, (xs, ts), (xs, ts), use_aq=True, path=None, running_stat=True, interval=16, iters=10,
                      batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)


def test_calibration_with_every_cache_on_the_host_is_bit_identical(golden, monkeypatch, caplog):
    """The whole tiny-UNet `cali_model` job of fixture F8 (every unit type of the DDPM UNet: single layers, ResnetBlocks, the attention
    block, the TIB unit) with every reconstruction cache in pinned host memory against the same job with the caches on the device."""
    import logging
    g = golden("f8_cali_tiny")
    dev = _calibrate(g, monkeypatch, host=False)
    with caplog.at_level(logging.INFO, logger="quant.data_utill"):
        host = _calibrate(g, monkeypatch, host=True)
    assert any("pinned host memory" in r.getMessage() for r in caplog.records)      # the host path really ran
    assert set(dev["weight"].keys()) == set(host["weight"].keys())
    for k in dev["weight"]:
        assert torch.equal(dev["weight"][k], host["weight"][k]), k
    for k in (k for k in dev if k.startswith("act_")):
        for n in dev[k]:
            assert torch.equal(torch.as_tensor(dev[k][n]), torch.as_tensor(host[k][n])), (k, n)
