"""GroupNorm-apply (from conv-epilogue statistics) at the SD 64x64 shapes: us and effective TB/s."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
qt = torch.tensor([[0.05, 120.0]], device=DEV); sel = ops.qsel(qt)
SHAPES = [(256, 1024, 128, 0), (256, 256, 256, 0), (256, 64, 256, 256), (128, 4096, 320, 0), (128, 4096, 320, 320), (128, 1024, 640, 0), (128, 256, 1280, 0), (32, 4096, 320, 0)]
if os.environ.get("SHAPES") == "cat":         # the SD up path's ResBlock inputs: GroupNorm over cat(h, skip) + the fp16 copy for the shortcut conv
    SHAPES = [(128, 4096, 320, 320), (128, 4096, 640, 320), (128, 1024, 640, 320), (128, 1024, 640, 640), (128, 1024, 1280, 640),
              (128, 256, 1280, 640), (128, 256, 1280, 1280), (128, 64, 1280, 1280)]
for (B, HW, C, C2) in SHAPES:
    H16 = os.environ.get("F16", "1") == "1"        # the fp16 activation stream
    x = torch.randn(B, HW, 1, C, device=DEV)
    if H16:
        x = x.half()
    seg = 64
    def stats(c):
        st = torch.zeros(B * HW // seg, c, 2, device=DEV); st[..., 1] = seg      # sum 0, sum of squares seg
        return st
    x._tfmq_stats = (stats(C), seg)
    x2 = None
    if C2:
        x2 = torch.randn(B, HW, 1, C2, device=DEV)
        if H16:
            x2 = x2.half()
        x2._tfmq_stats = (stats(C2), seg)
    g = torch.ones(C + C2, device=DEV); b = torch.zeros(C + C2, device=DEV)
    SILU = os.environ.get("SILU", "1") == "1"
    fn = lambda: ops.groupnorm(x, g, b, 1e-5, SILU, sel, x2=x2, want_cat=bool(C2), half_out=bool(C2))
    ms = t(fn)
    byts = B * HW * (C + C2) * ((3 if H16 else 5) + (2 if C2 else 0))
    print(f"B={B} HW={HW} C={C}+{C2}: gn_apply from stats {ms*1e3:8.1f} us  ({byts/ms/1e9:6.2f} TB/s algorithmic)", flush=True)
