"""Skinny fp32 GEMMs of the TIB unit (mini-batch rows x 320..1280 -> 1280): FMA tile kernel (M < 32) against the MFMA kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, tb) in [(8, 1280, 320, True), (8, 1280, 1280, True), (8, 320, 1280, True), (8, 1280, 1280, False), (32, 1280, 1280, True), (32, 1280, 1280, False),
                      (1280, 1280, 8, None), (1280, 320, 8, None)]:
    g = torch.Generator().manual_seed(M + N + K)
    if tb is None:      # wgrad: gw[N_out, K_in] = g^T [M rows] x  -> trans_a
        a = torch.randn(K, M, generator=g).to(DEV); b = torch.randn(K, N, generator=g).to(DEV)
        fn = lambda: ops.gemm(a, b, trans_a=True)
        ref = a.double().t() @ b.double()
    else:
        a = torch.randn(M, K, generator=g).to(DEV)
        b = (torch.randn(N, K, generator=g) if tb else torch.randn(K, N, generator=g)).to(DEV)
        fn = lambda: ops.gemm(a, b, trans_b=tb)
        ref = a.double() @ (b.double().t() if tb else b.double())
    us = t(fn)
    err = float((fn().double() - ref).abs().max() / ref.abs().max())
    print(f"M={M} N={N} K={K} trans_b={tb}: {us:7.1f} us  rel err {err:.2e}", flush=True)
