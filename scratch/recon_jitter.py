"""Per-iteration wall-clock of 80 AdaRound iterations of one SD-size transformer unit: are there recurring outliers?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
from tfmq_dm_amd.engine import recon as R
DEV = "cuda:0"
gen = torch.Generator().manual_seed(0)
def ada(cout, cin, k=1, bias=True):
    w = (torch.randn(cout, cin, generator=gen) * 0.05).to(DEV)
    qp = ops.minmax_to_qparam(ops.minmax(w.reshape(cout, -1).contiguous(), cout), 16)
    return R.AdaLayer(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), torch.zeros(cout, device=DEV) if bias else None)
C, HW, heads, N = 640, 32, 8, 8
T = HW * HW
x = torch.randn(N, T, C, device=DEV); y = torch.randn(N, T, C, device=DEV)
gn = (torch.ones(C, device=DEV), torch.zeros(C, device=DEV))
layers = [ada(C, C, 1, False), ada(C, C, 1, False), ada(C, C, 1, False), ada(C, C), ada(8 * C, C), ada(C, 4 * C),
          ada(C, C, 1, False), ada(C, 768, 1, False), ada(C, 768, 1, False), ada(C, C)]
tu = R.TransformerUnit(layers, [gn, gn, gn], heads, x, torch.randn(N, 77, 768, device=DEV), y, iters=1000)
idx = torch.arange(8, device=DEV)
tu.iterate(idx); torch.cuda.synchronize()
ts = []
for i in range(80):
    t0 = time.perf_counter(); tu.iterate(idx); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("median %.2f ms, mean %.2f ms" % (sorted(ts)[40], sum(ts) / len(ts)))
print("outliers (> 1.5 x median):", [(i, round(t, 1)) for i, t in enumerate(ts) if t > 1.5 * sorted(ts)[40]])
# back to back without per-iteration sync
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(80): tu.iterate(idx)
torch.cuda.synchronize(); print("80 iterations back to back: %.2f ms/iter" % ((time.perf_counter() - t0) * 1e3 / 80))
