"""Time per AdaRound iteration of SD-size reconstruction units (batch 8 = the reference's single-GPU SD setting)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
from tfmq_dm_amd.engine import recon as R
DEV = "cuda:0"
gen = torch.Generator().manual_seed(0)
def ada(cout, cin, k=1, bias=True):
    w = (torch.randn(cout, cin, k, k, generator=gen) * 0.05).to(DEV) if k > 1 else (torch.randn(cout, cin, generator=gen) * 0.05).to(DEV)
    qp = ops.minmax_to_qparam(ops.minmax(w.reshape(cout, -1).contiguous(), cout), 16)
    return R.AdaLayer(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), torch.zeros(cout, device=DEV) if bias else None)
def timeit(unit, n, bs, iters=5):
    idx = torch.arange(bs, device=DEV)
    unit.iterate(idx); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.time(); unit.iterate(idx); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    if max(ts) > 1.5 * min(ts): print("   uneven iterations:", [round(t, 1) for t in ts], flush=True)
    return sorted(ts)[len(ts) // 2]
N, bs = 8, 8
for (C, HW, heads) in [(320, 64, 8), (640, 32, 8), (1280, 16, 8)]:
    T = HW * HW
    x = torch.randn(N, HW, HW, C, device=DEV); emb = torch.randn(N, C, device=DEV); y = torch.randn(N, HW, HW, C, device=DEV)
    gn = (torch.ones(C, device=DEV), torch.zeros(C, device=DEV))
    ru = R.ResnetUnit(ada(C, C, 3), ada(C, C, 3), gn, gn, None, x, emb, y, eps=1e-5, iters=100)
    ms_r = timeit(ru, N, bs)
    fl_r = 2 * 3 * 2.0 * bs * T * C * C * 9
    layers = [ada(C, C, 1, False), ada(C, C, 1, False), ada(C, C, 1, False), ada(C, C), ada(8 * C, C), ada(C, 4 * C),
              ada(C, C, 1, False), ada(C, 768, 1, False), ada(C, 768, 1, False), ada(C, C)]
    tu = R.TransformerUnit(layers, [gn, gn, gn], heads, x.reshape(N, T, C), torch.randn(N, 77, 768, device=DEV), y.reshape(N, T, C), iters=100)
    ms_t = timeit(tu, N, bs)
    fl_t = 3 * 2.0 * bs * T * C * C * (4 + 16 + 2) + 3 * 2 * 2.0 * bs * T * T * C
    print(f"C={C} {HW}x{HW}: ResBlock unit {ms_r:8.1f} ms/iter ({fl_r/ms_r/1e9:6.1f} TFLOP/s)   transformer unit {ms_t:8.1f} ms/iter ({fl_t/ms_t/1e9:6.1f} TFLOP/s)", flush=True)
