"""3x3 w4a8 convolutions of the SD UNet at UNet batch 128: every tile kernel (tfmq_conv_desc.tile 1..4) vs the slab kernel (5) and its 128-pixel two-blocks-per-CU form (7)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "128"))
shapes = [(64, 320, 320, True), (64, 640, 320, False), (64, 960, 320, False), (32, 640, 640, True), (32, 1280, 640, False),
          (32, 1920, 640, False), (16, 1280, 1280, True), (16, 2560, 1280, False), (8, 1280, 1280, True), (8, 2560, 1280, False)]
STATS = os.environ.get("STATS", "1") == "1"      # GroupNorm statistics from the epilogue (timing ablation with 0)
TILES = tuple(int(t) for t in os.environ.get("TILES", "1,3,4,5,7").split(","))
UP = os.environ.get("UP") == "1"                 # the Upsample convs: nearest-2x fused into the 3x3 conv (H = input size)
if UP:
    shapes = [(32, 640, 640, False), (16, 1280, 1280, False), (8, 1280, 1280, False)]
if os.environ.get("SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["SHAPES"].split(",")]
gen = torch.Generator().manual_seed(0)
qt = torch.tensor([[[0.05, 120.0]]], device=dev)
sel = ops.qsel(qt)
for (H, cin, cout, res) in shapes:
    xq = torch.randint(-128, 128, (B, H, H, cin), dtype=torch.int8, device=dev)
    w = (torch.randn(cout, cin, 3, 3, generator=gen) * 0.02).to(dev)
    qp = ops.minmax_to_qparam(ops.minmax(w.reshape(cout, -1).contiguous(), cout), 16)
    pw = ops.pack_w4(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), None, torch.zeros(cout, device=dev))
    F16 = os.environ.get("F16", "1") == "1"          # fp16 activation stream (the sampling path) or the fp32 stream
    Ho = 2 * H if UP else H
    r = torch.randn(B, Ho, Ho, cout, device=dev) if res else None
    if r is not None and F16:
        r = r.half()
    ra = torch.randn(B, cout, device=dev)
    nops = 2.0 * B * Ho * Ho * cout * 9 * cin
    line = f"{B}x{H}x{H} {cin}->{cout} res={int(res)}:"
    ref = None
    for tile in TILES:
        orig = ops._tune_conv
        ops.set_conv_autotune({})
        ops._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
        try:
            y = ops.conv2d_w4a8(xq, pw, sel, pad=(1, 1, 1, 1), residual=r, rowadd=None if UP else ra, want_stats=STATS, out_f16=F16, up2x=UP)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                y = ops.conv2d_w4a8(xq, pw, sel, pad=(1, 1, 1, 1), residual=r, rowadd=None if UP else ra, want_stats=STATS, out_f16=F16, up2x=UP)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
        finally:
            ops._tune_conv = orig
            ops.set_conv_autotune(None)
        if ref is None:
            ref = (y.clone(), y._tfmq_stats[0].clone() if STATS else None)
        ok = torch.equal(y, ref[0]) and (not STATS or torch.equal(y._tfmq_stats[0], ref[1]))
        line += f"  t{tile}: {us:7.1f} us {nops / us / 1e6:6.0f} TOP/s{'' if ok else ' MISMATCH'}"
    print(line, flush=True)
