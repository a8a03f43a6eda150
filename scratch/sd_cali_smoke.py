"""cali_model end to end on the full SD v1 UNet (859.5 M, random init) with a tiny calibration set and a few AdaRound
iterations per unit: every reconstruction unit, TIAR, Finite-Set calibration and the checkpoint at production sizes.
Prints the wall-clock of the whole run and the per-unit iteration rate the run implies."""
import sys, os, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
import numpy as np, torch
from tfmq_dm_amd.ldm.unet import UNetModel, SD_V1_UNET
from quant.quant_layer import QMODE, Scaler
from quant.quant_model import QuantModel
from quant.calibration import cali_model, load_cali_model
from quant.reconstruction_util import RLOSS
DEV = "cuda:0"
N, G, ITERS = int(os.environ.get("N", "8")), int(os.environ.get("G", "2")), int(os.environ.get("ITERS", "6"))
torch.manual_seed(1234)
t0 = time.time()
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
print(f"model + QuantModel: {time.time()-t0:.1f}s", flush=True)
xs = torch.randn(G * N, 4, 64, 64, generator=g)
ts = torch.cat([torch.full((N,), float(t)) for t in np.linspace(981, 1, G).astype(int)])
cs = torch.randn(G * N, 77, 768, generator=g)
path = os.path.join(tempfile.mkdtemp(), "sd.pth")
import collections
import quant.calibration as QC, quant.reconstruction as QR, quant.data_utill as QD
acc = collections.defaultdict(float)
def timed(mod, name):
    f = getattr(mod, name)
    def g_(*a, **k):
        torch.cuda.synchronize(); t = time.time()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[name] += time.time() - t
        return r
    setattr(mod, name, g_)
for mod, name in ((QC, "tib_reconstruction"), (QC, "block_reconstruction"), (QC, "layer_reconstruction"), (QC, "_calibrate_activations"),
                  (QR, "save_inout")):
    if hasattr(mod, name): timed(mod, name)
import quant.quant_model as QM
timed(QM.QuantModel, "_lower")
ncall = collections.Counter()
_si = QR.save_inout
def _cnt(*a, **k):
    ncall["save_inout"] += 1
    return _si(*a, **k)
QR.save_inout = _cnt
t0 = time.time()
md = cali_model(qnn, (xs, ts, cs), (xs, ts, cs), use_aq=True, path=path, running_stat=True, interval=N, iters=ITERS,
                batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
torch.cuda.synchronize()
dt = time.time() - t0
nw = sum(1 for k in md["weight"] if k.endswith("alpha"))
print(f"cali_model: {dt:.1f}s for {G*N} samples, {ITERS} iterations/unit; {nw} AdaRound tensors, act groups {[k for k in md if k.startswith('act_')]}", flush=True)
print("phases (s):", {k: round(v, 1) for k, v in acc.items()})
print("calls:", dict(ncall))
print("checkpoint MB:", os.path.getsize(path) / 1e6)
bad = [k for k, v in md["weight"].items() if torch.is_tensor(v) and v.is_floating_point() and not torch.isfinite(v).all()]
print("non-finite entries:", bad[:5])
