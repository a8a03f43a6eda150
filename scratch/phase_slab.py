"""Diagnostics build only (TFMQ_EXTRA_HIPCC_FLAGS=-DTFMQ_PHASE_TIMERS python tfmq-dm_amd/build.py): K-loop / epilogue wall time per
block of the 3x3 slab kernel and how synchronised the blocks' epilogues are (one 8-wave block per CU: when every CU reaches its
epilogue at the same moment, the residual loads + output stores of all 256 tiles hit HBM together)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "128"))
shapes = [(64, 320, 320, True), (64, 640, 320, False), (32, 640, 640, True), (32, 1280, 640, False), (16, 1280, 1280, True)]
gen = torch.Generator().manual_seed(0)
sel = ops.qsel(torch.tensor([[[0.05, 120.0]]], device=dev))
ops.set_conv_autotune({})
ops._tune_conv = lambda h, name, kind, d, dsc: 5
for (H, cin, cout, res) in shapes:
    xq = torch.randint(-128, 128, (B, H, H, cin), dtype=torch.int8, device=dev)
    w = (torch.randn(cout, cin, 3, 3, generator=gen) * 0.02).to(dev)
    qp = ops.minmax_to_qparam(ops.minmax(w.reshape(cout, -1).contiguous(), cout), 16)
    pw = ops.pack_w4(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), None, torch.zeros(cout, device=dev))
    r = torch.randn(B, H, H, cout, device=dev).half() if res else None
    ra = torch.randn(B, cout, device=dev)
    kw = dict(pad=(1, 1, 1, 1), residual=r, rowadd=ra, want_stats=True, out_f16=True)
    for _ in range(2):
        ops.conv2d_w4a8(xq, pw, sel, **kw)
    torch.cuda.synchronize()
    os.environ["TFMQ_PHASE_PRINT"] = "1"
    ops.conv2d_w4a8(xq, pw, sel, **kw)
    torch.cuda.synchronize()
    del os.environ["TFMQ_PHASE_PRINT"]
