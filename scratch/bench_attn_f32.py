"""Exact-fp32 vs bf16x3 operand form of the reconstruction units' fused attention (forward, backward) at the SD unit shapes:
mini-batch 8, 8 heads; ms per call and nominal TFLOP/s (4 B h T L d forward, 10 B h T L d backward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for (B, heads, T, L, d) in ((8, 8, 4096, 4096, 40), (8, 8, 1024, 1024, 80), (8, 8, 4096, 77, 40)):
    C = heads * d
    q, k, v = (torch.randn(B, n, C, generator=g).to(DEV) for n in (T, L, L))
    go = torch.randn(B, T, C, generator=g).to(DEV)
    for mode in ("f32", "bf16x3"):
        with ops.gemm_precision(mode, 0):
            o, lse = ops.attention_f32_fwd(q, k, v, heads, d ** -0.5)
            ops.attention_f32_bwd(q, k, v, o, lse, go, heads, d ** -0.5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                o, lse = ops.attention_f32_fwd(q, k, v, heads, d ** -0.5)
            torch.cuda.synchronize()
            tf = (time.perf_counter() - t0) / 5
            t0 = time.perf_counter()
            for _ in range(5):
                ops.attention_f32_bwd(q, k, v, o, lse, go, heads, d ** -0.5)
            torch.cuda.synchronize()
            tb = (time.perf_counter() - t0) / 5
        fl = 4.0 * B * heads * T * L * d
        print(f"T={T} L={L} d={d} {mode:7s}: forward {tf * 1e3:7.3f} ms ({fl / tf / 1e12:6.1f} TFLOP/s)  backward {tb * 1e3:7.3f} ms ({2.5 * fl / tb / 1e12:6.1f} TFLOP/s)")
