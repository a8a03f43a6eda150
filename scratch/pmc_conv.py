"""One w4a8 conv shape, a few launches -- target of `rocprofv3 --pmc` passes (per-dispatch counters of the K loop)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
B, H, W, cin, cout, k = [int(v) for v in os.environ.get("SHAPE", "32,16,16,2560,1280,3").split(",")]
qt = torch.tensor([[0.05, 120.0]], device=DEV); sel = ops.qsel(qt)
x = (torch.randn(B, H, W, cin, device=DEV) * 40).clamp(-128, 127).to(torch.int8)
w = torch.randn(cout, cin, k, k, device=DEV) * 0.02
qp = ops.minmax_to_qparam(ops.minmax(w, cout), 16)
pw = ops.pack_w4(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), bias=torch.zeros(cout, device=DEV))
pad = (k // 2,) * 4
y = ops.conv2d_w4a8(x, pw, sel, pad=pad)
for _ in range(4):
    ops.conv2d_w4a8(x, pw, sel, pad=pad, out=y)
torch.cuda.synchronize()
