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
qt = torch.tensor([[0.05, 120.0]], device=DEV); sel = ops.qsel(qt)
for (B,H,W,cin,cout,k) in [(32,64,64,320,320,3), (32,32,32,640,640,3), (32,16,16,1280,1280,3)]:
    x = (torch.randn(B,H,W,cin, device=DEV)*40).clamp(-128,127).to(torch.int8)
    w = torch.randn(cout,cin,k,k, device=DEV)*0.02
    qp = ops.minmax_to_qparam(ops.minmax(w, cout), 16)
    pw = ops.pack_w4(w, qp[:,0].contiguous(), qp[:,1].contiguous(), bias=torch.zeros(cout, device=DEV))
    y = ops.conv2d_w4a8(x, pw, sel, pad=(1,1,1,1))
    fn = lambda: ops.conv2d_w4a8(x, pw, sel, pad=(1,1,1,1), out=y)
    for same in (0, 1):
        if same: os.environ["TFMQ_DBG_SAME"] = "1"
        else: os.environ.pop("TFMQ_DBG_SAME", None)
        os.environ.pop("TFMQ_PHASE_PRINT", None)
        ms = timeit(fn)
        print((B,H,W,cin,cout,k), "same" if same else "real", f"{ms*1e3:.1f} us  {2.0*B*H*W*cout*cin*k*k/ms/1e9:.0f} TOP/s", flush=True)
        os.environ["TFMQ_PHASE_PRINT"] = "1"
        fn(); torch.cuda.synchronize()
    os.environ.pop("TFMQ_DBG_SAME", None); os.environ.pop("TFMQ_PHASE_PRINT", None)
