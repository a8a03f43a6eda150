"""Life of ONE block of the pointwise w4a8 kernel (k_lin_direct, tile 6): launches of exactly 1, 3 and 6 tiles per CU
(256 CUs) at K = 320 / 640 / 1280, fp16 and GEGLU output.  Time of the 1-tile-per-CU launch ~ the serial latency chain of a
block (prologue loads -> K loop -> epilogue); how it grows with tiles per CU says what the CU is bound by."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(0)
qt = torch.tensor([[[0.05, 120.0]]], device=dev)
sel = ops.qsel(qt)
orig = ops._tune_conv
ops.set_conv_autotune({})
TILE = int(os.environ.get('TILE', '6'))
ops._tune_conv = lambda h, name, kind, d, dsc: TILE
for cin in (320, 640, 1280):
    for mode in ("f16", "geglu"):
        cout = 128
        w = (torch.randn(cout, cin, generator=gen) * 0.02).to(dev)
        qp = ops.minmax_to_qparam(ops.minmax(w, cout), 16)
        pw = ops.pack_w4(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), None, torch.zeros(cout, device=dev))
        line = f"K={cin} {mode}:"
        for per_cu in (1, 2, 3, 6, 12, 24):
            M = 256 * per_cu * 128
            xq = torch.randint(-128, 128, (M // 1024, 1024, 1, cin), dtype=torch.int8, device=dev)
            kw = {"geglu_oq": sel} if mode == "geglu" else {"out_f16": True}
            for _ in range(3):
                y = ops.conv2d_w4a8(xq, pw, sel, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = ops.conv2d_w4a8(xq, pw, sel, **kw)
            e1.record()
            torch.cuda.synchronize()
            line += f"  {per_cu}/CU: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us"
        print(line, flush=True)
