"""Pointwise w4a8 layers of the SD UNet at UNet batch 128: tile kernels (1..4) vs the register-direct-epilogue kernel (6)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "128"))
# (tokens per image, Cin, Cout, mode)
shapes = [(4096, 320, 2560, "geglu"), (4096, 320, 320, "f16res"), (4096, 1280, 320, "f16res"), (4096, 320, 320, "f16"),
          (1024, 640, 5120, "geglu"), (1024, 640, 640, "f16res"), (1024, 2560, 640, "f16res"),
          (256, 1280, 10240, "geglu"), (256, 1280, 1280, "f16res"), (256, 5120, 1280, "f16res"), (4096, 1280, 320, "q8res")]
if os.environ.get("SHAPES") == "qkv":        # fused q|k|v projections: q|k fp16 rows + V^T
    shapes = [(4096, 320, 960, "qkv"), (1024, 640, 1920, "qkv"), (256, 1280, 3840, "qkv"), (64, 1280, 3840, "qkv")]
if os.environ.get("SHAPES") == "qkvcmp":     # what the transposed V^T third costs: the same GEMM with plain fp16 rows
    shapes = [(4096, 320, 960, "qkv"), (4096, 320, 960, "f16"), (1024, 640, 1920, "qkv"), (1024, 640, 1920, "f16"), (256, 1280, 3840, "qkv"), (256, 1280, 3840, "f16")]
if os.environ.get("SHAPES") == "modes":      # one GEMM shape, the three epilogues: what the epilogue arithmetic / stores cost
    shapes = [(4096, 320, 2560, "geglu"), (4096, 320, 2560, "q8"), (4096, 320, 2560, "f16"), (4096, 320, 1280, "q8"), (4096, 320, 1280, "f16")]
if os.environ.get("SHAPES") == "f16":        # un-quantised skip-connection 1x1 convs (fp16 operands)
    shapes = [(4096, 640, 320, "h16"), (4096, 960, 320, "h16"), (1024, 1920, 640, "h16"), (1024, 1280, 640, "h16"), (256, 2560, 1280, "h16"),
              (64, 2560, 1280, "h16")]
if os.environ.get("ONLY"):
    shapes = [shapes[int(i)] for i in os.environ["ONLY"].split(",")]
TILES = tuple(int(t) for t in os.environ.get("TILES", "1,4,6,8").split(","))
gen = torch.Generator().manual_seed(0)
qt = torch.tensor([[[0.05, 120.0]]], device=dev)
sel = ops.qsel(qt)
for (T, cin, cout, mode) in shapes:
    if mode == "h16":
        xh = torch.randn(B, T, 1, cin, device=dev).half()
        pf = ops.pack_w_f16((torch.randn(cout, cin, generator=gen) * 0.02).to(dev), torch.zeros(cout, device=dev))
        line = f"{B}x{T} {cin}->{cout} f16 operands:"
        ref = None
        for tile in TILES:
            orig = ops._tune_conv
            ops.set_conv_autotune({})
            ops._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
            try:
                y = ops.conv2d_f16(xh, pf, out_f16=True)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    y = ops.conv2d_f16(xh, pf, out_f16=True)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 5 * 1e3
            finally:
                ops._tune_conv = orig
                ops.set_conv_autotune(None)
            if ref is None:
                ref = y.clone()
            line += f"  t{tile}: {us:7.1f} us{'' if torch.equal(y, ref) else ' MISMATCH'}"
        print(line, flush=True)
        continue
    xq = torch.randint(-128, 128, (B, T, 1, cin), dtype=torch.int8, device=dev)
    w = (torch.randn(cout, cin, generator=gen) * 0.02).to(dev)
    qp = ops.minmax_to_qparam(ops.minmax(w, cout), 16)
    pw = ops.pack_w4(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), None, torch.zeros(cout, device=dev))
    kw = {}
    if mode == "qkv":
        kw["out_f16"], kw["t_col0"] = True, 2 * cin
    elif mode == "geglu":
        kw["geglu_oq"] = sel
    elif mode.startswith("q8"):
        kw["out_q8"] = sel
    else:
        kw["out_f16"] = True
    if mode.endswith("res"):
        kw["residual"] = torch.randn(B, T, 1, cout, device=dev).half()
    nops = 2.0 * B * T * cout * cin
    line = f"{B}x{T} {cin}->{cout} {mode}:"
    ref = None
    for tile in TILES:
        orig = ops._tune_conv
        ops.set_conv_autotune({})
        ops._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
        try:
            y = ops.conv2d_w4a8(xq, pw, sel, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                y = ops.conv2d_w4a8(xq, pw, sel, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
        finally:
            ops._tune_conv = orig
            ops.set_conv_autotune(None)
        if isinstance(y, tuple):
            y = y[1]                    # fused q|k|v: compare the transposed V part
        if ref is None:
            ref = y.clone()
        line += f"  t{tile}: {us:7.1f} us{'' if torch.equal(y, ref) else ' MISMATCH'}"
    print(line, flush=True)
