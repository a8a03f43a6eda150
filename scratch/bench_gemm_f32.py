"""TFLOP/s of the reconstruction GEMM (tfmq_gemm_f32, fp32 MFMA) at the shapes of SD-size units."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
MODE = os.environ.get("GEMM_PREC", "f32")      # f32 | bf16x3 | f16 (ops.gemm_precision)
print("operand precision:", MODE)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
SHAPES = [(32768, 320, 2880, False, True), (32768, 2880, 320, False, False), (2880, 320, 32768, True, False),
          (32768, 320, 320, False, True), (32768, 2560, 320, False, True), (8192, 640, 5760, False, True),
          (2048, 1280, 11520, False, True), (4096, 4096, 40, False, True), (4096, 40, 4096, False, False)]
if os.environ.get("ONLY"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["ONLY"].split(",")]
NOLIB = os.environ.get("NOLIB") == "1"
for (M, N, K, ta, tb) in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=DEV)
    B = torch.randn((N, K) if tb else (K, N), device=DEV)
    out = torch.empty(M, N, device=DEV)
    with ops.gemm_precision(MODE):
        ms = t(lambda: ops.gemm(A, B, trans_a=ta, trans_b=tb, out=out))
    # library sgemm at the same shape / layouts (headroom check only; the product never calls it)
    At, Bt = (A.t() if ta else A), (B.t() if tb else B)
    ms_lib = float("nan") if NOLIB else t(lambda: torch.matmul(At, Bt, out=out))
    print(f"M={M:6d} N={N:5d} K={K:6d} ta={int(ta)} tb={int(tb)}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s   "
          f"(torch.matmul {ms_lib*1e3:8.1f} us {2.0*M*N*K/ms_lib/1e9:7.1f})", flush=True)
