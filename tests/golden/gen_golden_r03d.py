"""Round-3 golden vectors, part d, produced by importing the reference (build container only):

    python tests/golden/gen_golden_r03d.py

  f24  `split` / QMODE.QDIFF dual quantizers of QuantLayer (reference quant/quant_layer.py:296-334).  The tree rewrite of the
       released code never wraps skip / shortcut convs (quant_model.py:57-58), so `split` never reaches a QuantLayer there; the fixture
       builds the layer by hand, the way an edited name filter would: QuantLayer(Conv2d(64, 32, 1), w4 channel-wise MSE, a8 MSE,
       aq_mode=[NORMAL, QDIFF]) called with split=32 on a [4, 64, 8, 8] input whose two channel halves have different ranges.
       Recorded: the outputs in the FP, weight-only and weight+activation states (the quantizers initialise lazily in that order), the
       four quantizers' delta / zero_point, a second input through the initialised state, and the same after calibration.uaq2adar wrapped
       both weight quantizers in AdaRoundQuantizers (alphas recorded)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refharness as H  # noqa: E402

torch = H.install()
import torch.nn as nn  # noqa: E402
from gen_golden import save  # noqa: E402
from quant.quant_layer import QMODE, QuantLayer, Scaler  # noqa: E402
from quant.quant_block import BaseQuantBlock  # noqa: E402
from quant.calibration import uaq2adar  # noqa: E402


def f24():
    g = torch.Generator().manual_seed(24)
    conv = nn.Conv2d(64, 32, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(32, 64, 1, 1, generator=g) * 0.08)
        conv.weight[:, 32:] *= 0.35                 # the second half's weights live on another scale: two weight quantizers matter
        conv.bias.copy_(torch.randn(32, generator=g) * 0.1)
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    x = torch.randn(4, 64, 8, 8, generator=g)
    x[:, 32:] = x[:, 32:] * 3.0 + 0.5               # the skip half has another range: two activation quantizers matter
    x2 = torch.randn(4, 64, 8, 8, generator=g)
    x2[:, 32:] = x2[:, 32:] * 2.5 + 0.3
    layer = QuantLayer(conv, wq, aq, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
    out = {"w": conv.weight.detach().clone(), "b": conv.bias.detach().clone(), "x": x, "x2": x2, "split": np.array(32)}
    with torch.no_grad():
        layer.set_quant_state(False, False)
        out["y_fp"] = layer(x, split=32)
        assert layer.split == 32 and hasattr(layer, "aqtizer1") and hasattr(layer, "wqtizer1")
        layer.set_quant_state(True, False)
        out["y_w"] = layer(x, split=32)
        layer.set_quant_state(True, True)
        out["y_wa"] = layer(x, split=32)
        out["y_wa_x2"] = layer(x2, split=32)
        for n in ("wqtizer", "wqtizer1", "aqtizer", "aqtizer1"):
            q = getattr(layer, n)
            out[f"{n}/delta"] = q.delta.detach().clone().reshape(-1)
            zp = q.zero_point
            out[f"{n}/zp"] = (zp.detach().clone() if torch.is_tensor(zp) else torch.tensor(float(zp))).reshape(-1)
        # one quantizer for the whole layer (NORMAL only) for contrast: how much the split pair buys on this input
        plain = QuantLayer(conv, dict(wq), dict(aq), aq_mode=[QMODE.NORMAL.value]).eval()
        plain.set_quant_state(True, True)
        out["y_wa_single_pair"] = plain(x, split=32)
        blk = BaseQuantBlock(aq)
        blk.sc = layer
        box = nn.Module()
        box.b = blk
        uaq2adar(box)
        out["alpha"], out["alpha1"] = layer.wqtizer.alpha.detach().clone(), layer.wqtizer1.alpha.detach().clone()
        out["y_wa_adaround"] = layer(x, split=32)
    e1 = float((out["y_wa"] - out["y_fp"]).norm() / out["y_fp"].norm())
    e0 = float((out["y_wa_single_pair"] - out["y_fp"]).norm() / out["y_fp"].norm())
    print(f"w4a8 error vs FP: split pair {e1:.4f}, single pair {e0:.4f}; aq delta {float(layer.aqtizer.delta):.5f} / aq1 {float(layer.aqtizer1.delta):.5f}")
    save("f24_split_qdiff", **out)


if __name__ == "__main__":
    f24()
