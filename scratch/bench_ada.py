"""AdaRound per-iteration elementwise kernels at the SD layer shapes: soft weights forward, backward + Adam, OIHW <-> GEMM relayout.
Prints us and TB/s of the algorithmic bytes (12 / 28 / 8 B per weight) plus a checksum of the results (A/B runs: TFMQ_LIB_PATH)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (co, ci, k) in [(1280, 1280, 3), (640, 640, 3), (320, 320, 3), (320, 960, 3), (10240, 1280, 1), (320, 320, 1), (1280, 5120, 1)]:
    g = torch.Generator().manual_seed(co + ci)
    w = (torch.randn(co, ci, k, k, generator=g) * 0.05).to(DEV)
    wf = w.reshape(co, -1)
    mn, mx = wf.min(1).values, wf.max(1).values
    delta = ((mx - mn) / 15).contiguous(); zp = torch.round(-mn / delta).contiguous()
    alpha = torch.randn(w.shape, generator=g).to(DEV)
    gw = (torch.randn(w.shape, generator=g) * 1e-3).to(DEV)
    m = torch.zeros_like(w); v = torch.zeros_like(w); rl = torch.zeros(1, device=DEV)
    n = w.numel()
    us_f = t(lambda: ops.adaround_soft_fwd(w, alpha, delta, zp, 16))
    a2, m2, v2 = alpha.clone(), m.clone(), v.clone()
    ops.adaround_bwd_adam(w, a2, delta, zp, gw, m2, v2, 16, 0.01, 11.0, 1e-3, 3, rl)
    chk = float(a2.double().sum()), float(m2.double().abs().sum()), float(v2.double().sum()), float(rl)
    us_b = t(lambda: ops.adaround_bwd_adam(w, alpha, delta, zp, gw, m, v, 16, 0.01, 11.0, 1e-3, 3, rl))
    wh = ops.adaround_soft_fwd(w, alpha, delta, zp, 16)
    us_r0 = t(lambda: ops.w_relayout(wh, co, ci, k, k, True))
    gg = ops.w_relayout(wh, co, ci, k, k, True)
    us_r1 = t(lambda: ops.w_relayout(gg, co, ci, k, k, False))
    back = ops.w_relayout(gg, co, ci, k, k, False)
    ok = torch.equal(back, wh) and torch.equal(gg.reshape(co, k * k, ci), wh.reshape(co, ci, k * k).permute(0, 2, 1))
    print(f"{co}x{ci}x{k}x{k}: soft_fwd {us_f:7.1f} us ({12*n/us_f/1e6:5.2f} TB/s)  bwd_adam {us_b:7.1f} us ({28*n/us_b/1e6:5.2f} TB/s)  "
          f"relayout {us_r0:6.1f} / {us_r1:6.1f} us ({8*n/us_r0/1e6:5.2f} TB/s) round trip {'ok' if ok else 'MISMATCH'}  chk {chk[0]:.6f} {chk[1]:.6e} {chk[2]:.6e} {chk[3]:.5f}", flush=True)
