"""fp16-operand 3x3 layers (un-quantised / weight-only) at UNet batch 128: tile kernels (1, 3, 4) vs the slab kernel (5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "128"))
for (H, cin, cout) in [(64, 320, 320), (32, 640, 640), (32, 320, 640), (16, 1280, 1280), (16, 640, 1280), (8, 1280, 1280)]:
    x = torch.randn(B, H, H, cin, device=dev).half()
    pf = ops.pack_w_f16((torch.randn(cout, cin, 3, 3) * 0.02).to(dev), torch.zeros(cout, device=dev))
    line = f"{B}x{H}x{H} {cin}->{cout} f16 3x3:"
    ref = None
    for tile in (1, 3, 4, 5):
        orig = ops._tune_conv
        ops.set_conv_autotune({})
        ops._tune_conv = lambda h, name, kind, d, dsc, t=tile: t
        try:
            y = ops.conv2d_f16(x, pf, pad=(1, 1, 1, 1), out_f16=True, want_stats=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                y = ops.conv2d_f16(x, pf, pad=(1, 1, 1, 1), out_f16=True, want_stats=True)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
        finally:
            ops._tune_conv = orig; ops.set_conv_autotune(None)
        if ref is None: ref = y.float()
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        line += f"  t{tile}: {us:7.1f} us ({2.0*B*H*H*cout*9*cin/us/1e6:5.0f} TF/s, d {err:.1e})"
    print(line, flush=True)
