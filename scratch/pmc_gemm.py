"""One reconstruction-GEMM shape, a few launches -- target of `rocprofv3 --pmc` passes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
M, N, K = [int(v) for v in os.environ.get("SHAPE", "8192,640,5760").split(",")]
A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); out = torch.empty(M, N, device=DEV)
import contextlib
ctx = ops.gemm_precision(os.environ["GEMM_PREC"]) if os.environ.get("GEMM_PREC") else contextlib.nullcontext()
with ctx:
    for _ in range(5):
        ops.gemm(A, B, trans_a=False, trans_b=True, out=out)
torch.cuda.synchronize()
