"""Find the first block whose output differs between a UNet-batch-NB forward and the forward of its first half."""
import sys, os, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
DEV = "cuda:0"
NB = int(os.environ.get("CHECK_UNET_BATCH", "128"))
args = argparse.Namespace(batch=NB // 2, ddim_steps=2)
run, fwd, cpu, info = bench.setup_sd(args, torch.device(DEV), 0, lambda *a: None)
eng = [c.cell_contents for c in fwd.__closure__ if hasattr(c.cell_contents, "qtable") and hasattr(c.cell_contents, "forward")][0]
g = torch.Generator().manual_seed(3)
x = torch.randn(NB, 64, 64, 4, generator=g).to(DEV); ctx = torch.randn(NB, 77, 768, generator=g).to(DEV)
t = torch.full((NB,), 981.0, device=DEV)
h = NB // 2
with torch.cuda.stream(info["stream"]):
    info["step"].zero_()
    ta, tb = {}, {}
    e = eng.forward(x, t, ctx, taps=ta)
    info["stream"].synchronize()
    ta = {k: (v[1].clone() if torch.is_tensor(v[1]) else None) for k, v in ta.items() if isinstance(v, tuple)}
    a = eng.forward(x[:h].contiguous(), t[:h], ctx[:h].contiguous(), taps=tb)
    info["stream"].synchronize()
for k, v in tb.items():
    if not isinstance(v, tuple) or not torch.is_tensor(v[1]) or ta.get(k) is None:
        continue
    full, half = ta[k], v[1]
    if full.shape[0] != NB:
        continue
    same = torch.equal(full[:h], half)
    print(f"{k:50s} {tuple(half.shape)} {'same' if same else 'DIFF max ' + str(float((full[:h] - half).abs().max()))}")
    if not same:
        d = (full[:h] != half)
        idx = d.nonzero()
        print("   first differing indices", idx[:3].tolist(), "count", int(d.sum()), "of", d.numel())
        break
