"""Per-call timing (device sync per call) of every GEMM / kernel-level op of one AdaRound iteration of an SD-size
transformer reconstruction unit: which shapes the 67-91 ms go to."""
import sys, os, time, collections
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
C, HW, heads, N, bs = [int(v) for v in os.environ.get("UNIT", "320,64,8,8,8").split(",")]
T = HW * HW
x = torch.randn(N, T, C, device=DEV); y = torch.randn(N, T, C, device=DEV)
gn = (torch.ones(C, device=DEV), torch.zeros(C, device=DEV))
layers = [ada(C, C, 1, False), ada(C, C, 1, False), ada(C, C, 1, False), ada(C, C), ada(8 * C, C), ada(C, 4 * C),
          ada(C, C, 1, False), ada(C, 768, 1, False), ada(C, 768, 1, False), ada(C, C)]
tu = R.TransformerUnit(layers, [gn, gn, gn], heads, x, torch.randn(N, 77, 768, device=DEV), y, iters=100)
idx = torch.arange(bs, device=DEV)
tu.iterate(idx); torch.cuda.synchronize()
agg = collections.OrderedDict()
def wrap(name, keyfn):
    f = getattr(ops, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        key = (name,) + keyfn(*a, **k)
        e = agg.setdefault(key, [0, 0.0]); e[0] += 1; e[1] += dt
        return r
    setattr(ops, name, g)
wrap("gemm", lambda A, B, trans_a=False, trans_b=False, **k: (tuple(A.shape), tuple(B.shape), trans_a, trans_b))
wrap("gemm_strided", lambda *a, **k: tuple(a[12:16]) if len(a) >= 16 else (k.get("M"), k.get("N"), k.get("K"), k.get("batch")))
for n in ("softmax_rows", "softmax_bwd_rows", "layernorm_bwd", "geglu_bwd", "layernorm", "geglu", "adaround_soft_fwd", "adaround_bwd_adam"):
    if hasattr(ops, n): wrap(n, lambda *a, **k: ())
torch.cuda.synchronize(); t0 = time.perf_counter()
tu.iterate(idx); torch.cuda.synchronize()
tot = time.perf_counter() - t0
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t*1e3:8.2f} ms  n={n:3d}  {k}")
print("iteration (with per-call syncs):", tot * 1e3, "ms; sum of timed ops", sum(v[1] for v in agg.values()) * 1e3)
