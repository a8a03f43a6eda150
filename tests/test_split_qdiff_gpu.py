"""`split` / QMODE.QDIFF dual quantizers of QuantLayer (reference quant/quant_layer.py:296-334, calibration.py:35-40): the
concatenated input of an up-path shortcut conv gets one (activation, weight) quantizer pair per channel half.  The released tree rewrite
never wraps skip / shortcut convs (quant_model.py:57-58), so fixture F24 builds the QuantLayer by hand on the reference (tests/golden/
gen_golden_r03d.py); here the same hand-built layer runs on the HIP kernels: two w4a8 launches, the second accumulating onto the first.

Bars: the four quantizers' delta / zero_point bit-exact (they are MSE-initialised lazily on the device, per half); outputs within 1e-5 of
the largest output (integer sums per half, one fp32 affine map each, vs one fp32 conv over the concatenation); FP state at the fp16-operand
bar of the un-quantised layers (2e-3); AdaRound wrapping of both halves (uaq2adar): alpha signs exact, values to 1e-5, output 1e-5."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_split_qdiff_layer_matches_reference(golden):
    from quant.quant_layer import QMODE, QuantLayer, Scaler
    from quant.quant_block import BaseQuantBlock
    from quant.calibration import uaq2adar
    from tfmq_dm_amd._lib import TfmqError
    g = golden("f24_split_qdiff")
    conv = nn.Conv2d(64, 32, 1)
    with torch.no_grad():
        conv.weight.copy_(T(g["w"]))
        conv.bias.copy_(T(g["b"]))
    conv = conv.to(DEV)
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    layer = QuantLayer(conv, wq, aq, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
    x, x2, split = T(g["x"]).to(DEV), T(g["x2"]).to(DEV), int(g["split"])

    def close(y, key, tol):
        ref = T(g[key])
        err = float((y.cpu() - ref).abs().max() / ref.abs().max())
        print(f"[{key}] max error / max |y| = {err:.2e}")
        assert err <= tol, (key, err)

    layer.set_quant_state(False, False)
    close(layer(x, split=split), "y_fp", 2e-3)
    assert layer.split == split and hasattr(layer, "aqtizer1") and hasattr(layer, "wqtizer1")
    layer.set_quant_state(True, False)
    close(layer(x, split=split), "y_w", 2e-3)                       # weight-only: fp16-operand kernel on the exact integer grids
    layer.set_quant_state(True, True)
    close(layer(x, split=split), "y_wa", 1e-5)
    for n in ("wqtizer", "wqtizer1", "aqtizer", "aqtizer1"):
        q = getattr(layer, n)
        zp = q.zero_point if torch.is_tensor(q.zero_point) else torch.tensor(float(q.zero_point))
        assert torch.equal(q.delta.detach().cpu().reshape(-1), T(g[f"{n}/delta"])), n
        assert torch.equal(zp.detach().cpu().reshape(-1).float(), T(g[f"{n}/zp"]).float()), n
    assert float(layer.aqtizer1.delta) > 2.0 * float(layer.aqtizer.delta)          # the halves do need their own grids on this input
    close(layer(x2, split=split), "y_wa_x2", 1e-5)
    # a single pair for the whole layer is the NORMAL mode: it must NOT reproduce the split output
    ref_single = T(g["y_wa_single_pair"])
    assert float((layer(x, split=split).cpu() - ref_single).abs().max() / ref_single.abs().max()) > 1e-2
    blk = BaseQuantBlock(aq)
    blk.sc = layer
    box = nn.Module()
    box.b = blk
    uaq2adar(box)
    for mine, key in ((layer.wqtizer.alpha, "alpha"), (layer.wqtizer1.alpha, "alpha1")):
        a, ra = mine.detach().cpu(), T(g[key])
        # the device's log against torch's: a few ulp; what the hard rounding reads is the sign
        assert a.shape == ra.shape and torch.equal(a >= 0, ra >= 0) and float((a - ra).abs().max()) <= 1e-5 * float(ra.abs().max())
    close(layer(x, split=split), "y_wa_adaround", 1e-5)
    layer.set_running_stat(True)
    assert layer.aqtizer.running_stat and layer.aqtizer1.running_stat
    with pytest.raises(TfmqError):          # the fused engine plan / reconstruction units take un-split layers only: loud, not silent
        layer.weight_quant_state()
