"""Known-byte launches for calibrating FETCH_SIZE / WRITE_SIZE on THIS library's access patterns (MI355X_MICROARCH.md: the 2x FETCH_SIZE
rule is for wide coalesced streaming reads; other patterns 'calibrate on a known byte count in your own access pattern'):
  copy      torch copy of 1 GiB fp32 (wide coalesced: the rule's own case)                       read 1073.7 MB, write 1073.7 MB
  lin       pointwise w4a8, M = 524288, K = 320, N = 128 (ONE column tile: every activation row is read exactly once by LDS-DMA,
            16 lanes x 64-byte row slices per wave instruction), fp16 output                       read 167.8 MB (+0.04 weights), write 134.2 MB
  lin_res   the same with an fp16 residual (16-byte per-lane loads of 256-byte rows)               read 167.8 + 134.2 MB, write 134.2 MB
  slab      3x3 w4a8 320 -> 320 at 64 x 64, UNet batch 32, fp16 residual + fp16 output            read 41.9 + 83.9 MB (+0.9 weights), write 83.9 MB"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
qt = torch.tensor([[[0.05, 120.0]]], device=dev)
sel = ops.qsel(qt)


def pw(cout, cin, k):
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.02).to(dev)
    qp = ops.minmax_to_qparam(ops.minmax(w.reshape(cout, -1).contiguous(), cout), 16)
    return ops.pack_w4(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), None, torch.zeros(cout, device=dev))


M = 524288
x = torch.randint(-128, 128, (1, M, 1, 320), dtype=torch.int8, device=dev)
p1 = pw(128, 320, 1)
r1 = torch.randn(1, M, 1, 128, device=dev).half()
xs = torch.randint(-128, 128, (32, 64, 64, 320), dtype=torch.int8, device=dev)
p3 = pw(320, 320, 3)
r3 = torch.randn(32, 64, 64, 320, device=dev).half()
a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
a.fill_(1.0)
torch.cuda.synchronize()
for rep in range(3):
    b.copy_(a)
    ops.conv2d_w4a8(x, p1, sel, out_f16=True, want_stats=False)
    ops.conv2d_w4a8(x, p1, sel, residual=r1, out_f16=True, want_stats=False)
    ops.conv2d_w4a8(xs, p3, sel, pad=(1, 1, 1, 1), residual=r3, out_f16=True, want_stats=False)
torch.cuda.synchronize()
print("done")
