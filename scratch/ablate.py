import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
qt = torch.tensor([[0.05, 120.0]], device=DEV); sel = ops.qsel(qt)
B,H,W,cin,cout,k = 256,32,32,128,128,3
x = (torch.randn(B,H,W,cin, device=DEV)*40).clamp(-128,127).to(torch.int8)
w = torch.randn(cout,cin,k,k, device=DEV)*0.02
qp = ops.minmax_to_qparam(ops.minmax(w, cout), 16)
pw = ops.pack_w4(w, qp[:,0].contiguous(), qp[:,1].contiguous(), bias=torch.zeros(cout, device=DEV))
y = ops.conv2d_w4a8(x, pw, sel, pad=(1,1,1,1))
ms = timeit(lambda: ops.conv2d_w4a8(x, pw, sel, pad=(1,1,1,1), out=y))
print(os.environ.get("TFMQ_ABLATE","0"), f"{ms*1e3:.1f} us")
