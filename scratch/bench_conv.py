import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ctypes as C
import tfmq_dm_amd.ops as ops
from tfmq_dm_amd._lib import handle
DEV = "cuda:0"
def qtab(d, z): return torch.tensor([[float(d), float(z)]], dtype=torch.float32, device=DEV)
shapes = [  # B,H,W,cin,cout,k
 (256,32,32,128,128,3), (256,32,32,256,128,3), (256,16,16,256,256,3), (256,16,16,512,256,3),
 (256,8,8,256,256,3), (256,4,4,256,256,3), (256,16,16,256,768,1), (256,16,16,256,256,1), (256,32,32,384,128,3),
]
h = handle(0)
for (B,H,W,cin,cout,k) in shapes:
    x = (torch.randn(B,H,W,cin, device=DEV)*40).clamp(-128,127).to(torch.int8)
    w = torch.randn(cout,cin,k,k, device=DEV)*0.02
    mm = ops.minmax(w, cout); qp = ops.minmax_to_qparam(mm, 16)
    pw = ops.pack_w4(w, qp[:,0].contiguous(), qp[:,1].contiguous(), bias=torch.zeros(cout, device=DEV))
    sel = ops.qsel(qtab(0.05, 120.0))
    pad = (k//2,)*4
    y = ops.conv2d_w4a8(x, pw, sel, pad=pad)
    torch.cuda.synchronize()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.conv2d_w4a8(x, pw, sel, pad=pad, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/n
    gop = 2.0*B*H*W*cout*k*k*cin/1e9
    print(f"w4a8 B{B} {H}x{W} {cin}->{cout} k{k}: {ms*1e3:8.1f} us  {gop/ms:8.1f} TOP/s  ({gop:.1f} GOP)")
# groupnorm
for (B,HW,C1,C2) in [(256,1024,128,0),(256,1024,256,128),(256,256,256,0),(256,256,256,256),(256,64,256,0)]:
    side=int(HW**0.5)
    x1 = torch.randn(B,side,side,C1, device=DEV); x2 = torch.randn(B,side,side,C2, device=DEV) if C2 else None
    g = torch.ones(C1+C2, device=DEV); b = torch.zeros(C1+C2, device=DEV)
    sel = ops.qsel(qtab(0.05, 120.0))
    ops.groupnorm(x1,g,b,1e-6,True,sel,x2=x2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.groupnorm(x1,g,b,1e-6,True,sel,x2=x2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    byts = B*HW*(C1+C2)*5
    print(f"gn B{B} HW{HW} C{C1}+{C2}: {ms*1e3:8.1f} us  {byts/ms/1e9:6.2f} TB/s (algorithmic 5 B/elem)")
