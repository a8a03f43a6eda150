"""Eager forwards of the SD UNet in FP / weight-only state (the calibration data passes: save_inout, FP sampling of the
calibration set) -- target of `rocprofv3 --kernel-trace --stats`."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
import torch
from tfmq_dm_amd.ldm.unet import UNetModel, SD_V1_UNET
from tfmq_dm_amd import ops
from quant.quant_layer import QMODE, Scaler
from quant.quant_model import QuantModel
DEV = "cuda:0"
torch.manual_seed(1234)
m = UNetModel(**SD_V1_UNET)
g = torch.Generator().manual_seed(7)
with torch.no_grad():
    for p in m.parameters():
        if p.numel() and float(p.abs().max()) == 0.0:
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
m = m.to(DEV)
wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MINMAX}
aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": True}
qnn = QuantModel(m, wq, aq, cali=True, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
B = int(os.environ.get("B", "64"))
x = ops.nchw_to_nhwc(torch.randn(B, 4, 64, 64, generator=g).to(DEV)); t = torch.full((B,), 500.0, device=DEV); c = torch.randn(B, 77, 768, generator=g).to(DEV)
state = os.environ.get("STATE", "fp")
qnn.set_quant_state(state != "fp", False)
if state != "fp":
    qnn(torch.randn(8, 4, 64, 64).to(DEV), torch.full((8,), 500.0, device=DEV), torch.randn(8, 77, 768).to(DEV)); qnn.disable_out_quantization(); qnn.invalidate()
eng = qnn.engine(DEV)
eng.forward(x, t, c); torch.cuda.synchronize()
ops.upsample2x(torch.zeros(1, 2, 2, 4, device=DEV)); torch.cuda.synchronize()     # marker
t0 = time.time()
for _ in range(3): eng.forward(x, t, c)
torch.cuda.synchronize()
print(f"{state} forward, batch {B}: {(time.time()-t0)/3*1e3:.1f} ms")
