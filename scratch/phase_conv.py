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
cases = [(32,4096,1,320,2560,1,"geglu"), (32,4096,1,1280,320,1,"res"), (32,1024,1,2560,640,1,"res"), (32,256,1,5120,1280,1,"res"),
         (32,64,64,320,320,3,"res"), (32,32,32,640,640,3,"res"), (32,16,16,1280,1280,3,"res"), (32,16,16,2560,1280,3,"f32")]
if os.environ.get("PHASE_CASES") == "all":
    cases += [(32,4096,1,320,320,1,"res"), (32,4096,1,320,960,1,"qkv"), (32,1024,1,640,5120,1,"geglu"), (32,1024,1,640,640,1,"res"),
              (32,256,1,1280,10240,1,"geglu"), (32,256,1,1280,1280,1,"res"), (32,8,8,1280,1280,3,"res")]
if os.environ.get("PHASE_NO_GEGLU"): cases = [c for c in cases if c[6] != "geglu"]
for (B,H,W,cin,cout,k,mode) in cases:
    x = (torch.randn(B,H,W,cin, device=DEV)*40).clamp(-128,127).to(torch.int8)
    w = torch.randn(cout,cin,k,k, device=DEV)*0.02
    qp = ops.minmax_to_qparam(ops.minmax(w, cout), 16)
    pw = ops.pack_w4(w, qp[:,0].contiguous(), qp[:,1].contiguous(), bias=torch.zeros(cout, device=DEV))
    pad = (k//2,)*4
    res = torch.randn(B,H,W,cout, device=DEV) if mode == "res" else None
    if mode == "geglu": fn = lambda: ops.conv2d_w4a8(x, pw, sel, geglu_oq=sel)
    elif mode == "qkv": fn = lambda: ops.conv2d_w4a8(x, pw, sel, out_f16=True, t_col0=640)
    else:
        y = ops.conv2d_w4a8(x, pw, sel, pad=pad, residual=res)
        fn = lambda: ops.conv2d_w4a8(x, pw, sel, pad=pad, residual=res, out=y)
    os.environ.pop("TFMQ_PHASE_PRINT", None)
    ms = timeit(fn)
    print((B,H,W,cin,cout,k,mode), f"{ms*1e3:.1f} us", flush=True)
    os.environ["TFMQ_PHASE_PRINT"] = "1"
    fn(); torch.cuda.synchronize()
    os.environ.pop("TFMQ_PHASE_PRINT", None)
