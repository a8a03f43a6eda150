import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
B, heads, T, d = 16, 8, 4096, 40
C = heads*d
q = torch.randn(B, T, C, device=DEV).half(); k = torch.randn(B, T, C, device=DEV).half()
vt = torch.randn(B, C, T, device=DEV).half()
qt = torch.tensor([[0.02, 128.0]], device=DEV); sel = ops.qsel(qt)
for _ in range(3):
    ops.attention_f16(q, k, vt, heads, d**-0.5, sel, want_f32=False)
torch.cuda.synchronize()
