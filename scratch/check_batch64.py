"""UNet batch 128 (bench default: 64 images x CFG) vs two UNet-batch-64 forwards on the same rows: bit-exact."""
import sys, os, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
DEV = "cuda:0"
NB = int(os.environ.get("CHECK_UNET_BATCH", "128"))
args = argparse.Namespace(batch=NB // 2, ddim_steps=2)
run, fwd, cpu, info = bench.setup_sd(args, torch.device(DEV), 0, lambda *a: print(*a, file=sys.stderr))
eng = [c.cell_contents for c in fwd.__closure__ if hasattr(c.cell_contents, "qtable") and hasattr(c.cell_contents, "forward")][0]
g = torch.Generator().manual_seed(3)
x = torch.randn(NB, 64, 64, 4, generator=g).to(DEV); ctx = torch.randn(NB, 77, 768, generator=g).to(DEV)
t = torch.full((NB,), 981.0, device=DEV)
with torch.cuda.stream(info["stream"]):
    info["step"].zero_()
    e = eng.forward(x, t, ctx).clone()
    a = eng.forward(x[:NB // 2].contiguous(), t[:NB // 2], ctx[:NB // 2].contiguous()).clone()
    b = eng.forward(x[NB // 2:].contiguous(), t[NB // 2:], ctx[NB // 2:].contiguous()).clone()
    info["stream"].synchronize()
print("finite", bool(torch.isfinite(e).all()), "first half equal", torch.equal(e[:NB // 2], a), "second half equal", torch.equal(e[NB // 2:], b))
