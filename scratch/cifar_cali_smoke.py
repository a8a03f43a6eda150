"""cali_model end to end on the full CIFAR-10 DDPM UNet (35.7 M, random init): every reconstruction unit, TIAR,
Finite-Set calibration, checkpoint round trip -- the DDPM side of scratch/sd_cali_smoke.py."""
import sys, os, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
import numpy as np, torch
import tfmq_dm_amd.ddim.models as M
from quant.quant_layer import QMODE, Scaler
from quant.quant_model import QuantModel
from quant.calibration import cali_model, load_cali_model
from quant.reconstruction_util import RLOSS
DEV = "cuda:0"
N, G, ITERS = int(os.environ.get("N", "64")), int(os.environ.get("G", "3")), int(os.environ.get("ITERS", "20"))
m = M.random_init(M.Model(M.make_config()))
sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
qnn = QuantModel(m, wq, aq, cali=True, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).to(DEV).eval()
g = torch.Generator().manual_seed(7)
xs = torch.randn(G * N, 3, 32, 32, generator=g)
ts = torch.cat([torch.full((N,), float(t)) for t in np.linspace(981, 1, G).astype(int)])
path = os.path.join(tempfile.mkdtemp(), "cifar.pth")
torch.manual_seed(5); np.random.seed(5)
t0 = time.time()
md = cali_model(qnn, (xs, ts), (xs, ts), use_aq=True, path=path, running_stat=True, interval=N, iters=ITERS, batch_size=32,
                w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
torch.cuda.synchronize()
print(f"cali_model: {time.time()-t0:.1f}s for {G*N} samples, {ITERS} iterations/unit; act groups {[k for k in md if k.startswith('act_')]}; "
      f"{sum(1 for k in md['weight'] if k.endswith('alpha'))} AdaRound tensors; checkpoint {os.path.getsize(path)/1e6:.0f} MB", flush=True)
m2 = M.Model(M.make_config()); m2.load_state_dict(sd0)
q2 = QuantModel(m2, wq, aq, cali=False, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).to(DEV).eval()
load_cali_model(q2, (torch.randn(1, 3, 32, 32), torch.randint(0, 1000, (1,)).float()), use_aq=True, path=path)
ck = torch.load(path, map_location="cpu")
q2.load_state_dict(ck["act_1"], strict=False)
x = torch.randn(4, 3, 32, 32, generator=g).to(DEV); t = torch.full((4,), 500.0, device=DEV)
qnn.set_quant_state(False, False); fp = qnn(x, t)
q2.set_quant_state(True, True); qe = q2(x, t)
print("reloaded w4a8 eps vs FP eps rel-L2:", float((qe - fp).norm() / fp.norm()), "finite", bool(torch.isfinite(qe).all()))

# ---- the same calibrated state through the CPU oracle (the reference's fake-quant forward): engine == oracle up to bin flips
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tfmq_oracle as O
from quant.quant_layer import QuantLayer
sdc, wqs, aqs = {}, {}, {}
for n, mod in q2.model.named_modules():
    if isinstance(mod, QuantLayer):
        sdc[n + ".weight"] = mod.original_w.detach().cpu().float()
        if mod.original_b is not None:
            sdc[n + ".bias"] = mod.original_b.detach().cpu().float()
        if mod.use_wq:
            d, z, a = mod.weight_quant_state()
            shp = (-1,) + (1,) * (sdc[n + ".weight"].dim() - 1)
            wqs[n] = {"delta": d.detach().cpu().float().reshape(shp), "zp": z.detach().cpu().float().reshape(shp),
                      "alpha": None if a is None else a.detach().cpu().float()}
        if mod.use_aq and not mod.disable_aq and mod.aqtizer.delta is not None:
            aqs[n] = (float(mod.aqtizer.delta), float(mod.aqtizer.zero_point))
    elif isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear, torch.nn.GroupNorm)):
        for pn, p in mod.named_parameters(recurse=False):
            sdc[f"{n}.{pn}"] = p.detach().cpu().float()
with torch.no_grad():
    ref = O.ddim_unet_forward(sdc, q2.model.engine_cfg(), x.cpu(), t.cpu().long(), O.QuantSpec(wq=wqs, aq=aqs))
    ref_fp = O.ddim_unet_forward(sdc, q2.model.engine_cfg(), x.cpu(), t.cpu().long(), None)
print("engine w4a8 vs oracle w4a8 rel-L2:", float((qe.cpu() - ref).norm() / ref.norm()),
      "| oracle w4a8 vs oracle FP:", float((ref - ref_fp).norm() / ref_fp.norm()),
      "| engine FP vs oracle FP:", float((fp.cpu() - ref_fp).norm() / ref_fp.norm()))
