"""Diagnostics build only (TFMQ_EXTRA_HIPCC_FLAGS=-DTFMQ_PHASE_TIMERS python tfmq-dm_amd/build.py): where the producer wave and a
consumer wave of k_lin_stream (tile 7) spend their cycles, per block, on the SD pointwise shapes at UNet batch 128."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "128"))
shapes = [(4096, 320, 2560, "geglu"), (4096, 320, 320, "f16res"), (4096, 320, 320, "f16"), (4096, 320, 960, "f16"), (1024, 2560, 640, "f16res"), (256, 1280, 10240, "geglu")]
gen = torch.Generator().manual_seed(0)
sel = ops.qsel(torch.tensor([[[0.05, 120.0]]], device=dev))
ops.set_conv_autotune({})
TILE = int(os.environ.get('TILE', '7'))
ops._tune_conv = lambda h, name, kind, d, dsc: TILE
for (T, cin, cout, mode) in shapes:
    xq = torch.randint(-128, 128, (B, T, 1, cin), dtype=torch.int8, device=dev)
    w = (torch.randn(cout, cin, generator=gen) * 0.02).to(dev)
    qp = ops.minmax_to_qparam(ops.minmax(w, cout), 16)
    pw = ops.pack_w4(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), None, torch.zeros(cout, device=dev))
    kw = {"geglu_oq": sel} if mode == "geglu" else {"out_f16": True}
    if mode.endswith("res"):
        kw["residual"] = torch.randn(B, T, 1, cout, device=dev).half()
    for _ in range(2):
        ops.conv2d_w4a8(xq, pw, sel, **kw)
    torch.cuda.synchronize()
    os.environ["TFMQ_PHASE_PRINT"] = "1"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d_w4a8(xq, pw, sel, **kw)
    e1.record()
    torch.cuda.synchronize()
    del os.environ["TFMQ_PHASE_PRINT"]
    print(f"{B}x{T} {cin}->{cout} {mode}: {e0.elapsed_time(e1) * 1e3:.0f} us", file=sys.stderr, flush=True)
