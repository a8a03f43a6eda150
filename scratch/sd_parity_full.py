"""Engine vs CPU oracle on the FULL Stable Diffusion v1 UNet (859.5 M, random init, w4a8 with the bench's synthetic
Finite-Set tables): one CFG pair (UNet batch 2) at the first DDIM step, FP and w4a8.  The oracle is the reference's
fake-quant forward restated on torch-CPU (pinned to the reference by fixtures F11-F13 at tiny sizes); ~1 min of CPU."""
import sys, os, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import bench
import tfmq_oracle as O
from tfmq_dm_amd.ldm.sampler import ddim_timesteps
DEV = torch.device("cuda", 0)
args = argparse.Namespace(batch=1, ddim_steps=2)
run, fwd, cpu, info = bench.setup_sd(args, DEV, 0, lambda *a: print("[setup]", *a, file=sys.stderr))
st = info["oracle_state"]
eng, sd, wq, act_names, cfg = st["eng"], st["sd"], st["wq"], st["act_names"], st["cfg"]
g = torch.Generator().manual_seed(123)
x = torch.randn(2, 4, 64, 64, generator=g); ctx = torch.randn(2, 77, 768, generator=g)
tv = float(np.flip(ddim_timesteps(2))[0])
t = torch.full((2,), tv)
with torch.cuda.stream(info["stream"]):
    info["step"].zero_()
    e = eng.forward(x.permute(0, 2, 3, 1).contiguous().to(DEV), t.to(DEV), ctx.to(DEV)).permute(0, 3, 1, 2).clone()
    info["stream"].synchronize()
sdc = {k: v.cpu() for k, v in sd.items()}
def shp(n, v):
    return v.cpu().reshape((-1,) + (1,) * (sdc[n + ".weight"].dim() - 1))
wqc = {n: {"delta": shp(n, q.delta), "zp": shp(n, q.zp), "alpha": None} for n, q in wq.items()}
qt = eng.qtable.cpu()
aq = {n: (qt[0, j, 0], qt[0, j, 1]) for j, n in enumerate(act_names)}
t0 = time.time()
with torch.no_grad():
    ref = O.ldm_unet_forward(sdc, dict(cfg), x, t.long(), ctx, O.QuantSpec(wq=wqc, aq=aq))
print(f"oracle w4a8 forward: {time.time()-t0:.1f}s")
rel = float((e.cpu() - ref).norm() / ref.norm())
print("FULL SD v1 UNet, w4a8: engine vs oracle eps rel-L2 =", rel, "| max-normalised", float((e.cpu() - ref).abs().max() / ref.abs().max()))
