import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
BB = int(os.environ.get('BATCH', '16'))
SHAPES = [(BB, 8, 4096, 40)] if os.environ.get("ONLY40") else [(BB, 8, 4096, 40), (BB, 8, 1024, 80), (BB, 8, 256, 160)]
for (B, heads, T, d) in SHAPES:
    C = heads*d
    q = torch.randn(B, T, C, device=DEV); k = torch.randn(B, T, C, device=DEV); v = torch.randn(B, T, C, device=DEV)
    qh, kh, vt = q.half(), k.half(), v.transpose(1, 2).contiguous().half()
    qt = torch.tensor([[0.02, 128.0]], device=DEV); sel = ops.qsel(qt)
    t32 = timeit(lambda: ops.attention(q, k, v, heads, d**-0.5, sel, want_f32=False)) if os.environ.get('F32') else float('nan')
    t16 = timeit(lambda: ops.attention_f16(qh, kh, vt, heads, d**-0.5, sel, want_f32=False))
    fl = 4.0*B*heads*T*T*d
    print(f"B{B} h{heads} T{T} d{d}: fp32-in {t32*1e3:8.1f} us ({fl/t32/1e9:6.1f} TF/s)   f16-in {t16*1e3:8.1f} us ({fl/t16/1e9:6.1f} TF/s)")
