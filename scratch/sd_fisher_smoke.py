"""Fisher-weighted reconstruction at production size: save_grad (GetLayerGrad on the backward tape over the exact-fp32 engine) for units
of the full SD v1 UNet (859.5 M, random init), then a few FISHER_DIAG AdaRound iterations of one of them.  Prints times, the size of
the tape and sanity properties (finite, non-zero, |g| + 1 >= 1, deterministic)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
import torch
from tfmq_dm_amd.ldm.unet import UNetModel, SD_V1_UNET
from quant.quant_layer import QMODE, Scaler
from quant.quant_model import QuantModel
from quant.reconstruction_util import RLOSS
import quant.data_utill as DU, quant.reconstruction as REC
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
N = int(os.environ.get("N", "4"))
xs, ts, cs = torch.randn(N, 4, 64, 64, generator=g), torch.full((N,), 501.0), torch.randn(N, 77, 768, generator=g)
qnn.set_quant_state(True, False)
qnn(xs[:1].to(DEV), ts[:1].to(DEV), cs[:1].to(DEV))            # weight quantizers initialise (what cali_model does first)
mods = dict(qnn.model.named_modules())
for name in os.environ.get("UNITS", "input_blocks.1.0,input_blocks.4.1.transformer_blocks.0,output_blocks.11.1.proj_out").split(","):
    unit = mods[name]
    torch.cuda.synchronize(); t0 = time.time()
    w1 = DU.save_grad(qnn, unit, (xs, ts, cs), 1.0, False, 2, True)
    torch.cuda.synchronize(); dt = time.time() - t0
    w2 = DU.save_grad(qnn, unit, (xs, ts, cs), 1.0, False, 2, True)
    print(f"{name}: save_grad for {N} samples {dt:.2f}s; weights {tuple(w1.shape)} min {float(w1.min()):.6f} max {float(w1.max()):.6f} "
          f"finite {bool(torch.isfinite(w1).all())} deterministic {bool(torch.equal(w1, w2))}", flush=True)
unit = mods["input_blocks.1.0"]
torch.cuda.synchronize(); t0 = time.time()
REC.LOSS_TRACE = {"counts": (1, 5, 10), "rows": [], "unit": 0}
REC.block_reconstruction(qnn, unit, cali_data=(xs, ts, cs), batch_size=2, iters=10, w=0.01, opt_mode=RLOSS.FISHER_DIAG, asym=True, warmup=0.2,
                         use_aq=False, multi_gpu=False)
torch.cuda.synchronize()
print(f"block_reconstruction(input_blocks.1.0, FISHER_DIAG, 10 iterations): {time.time() - t0:.2f}s; (count, rec, round) {[(r[1], round(r[2], 6), round(r[3], 3)) for r in REC.LOSS_TRACE['rows']]}")
