import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch, numpy as np
import torch.nn.functional as F
import tfmq_dm_amd.ops as ops
import tfmq_oracle as O
DEV = "cuda:0"
def qtab(d, z): return torch.tensor([[float(d), float(z)]], dtype=torch.float32, device=DEV)
torch.manual_seed(0)
# 1x1 conv, integer-valued everything: delta_a = 1, za = 128 -> a' = x ; delta_w = 1, zw = 0 -> qw = w
B, H, W, cin, cout = 1, 4, 8, 64, 32+16
x = torch.randint(-5, 6, (B, cin, H, W)).float()
w = torch.randint(0, 4, (cout, cin, 1, 1)).float()
sel = ops.qsel(qtab(1.0, 128.0))
xq = ops.quantize_act(x.permute(0,2,3,1).contiguous().to(DEV), sel)
print("xq ok", torch.equal(xq.cpu().float(), x.permute(0,2,3,1)))
pw = ops.pack_w4(w.to(DEV), torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV))
print("unpack ok", torch.equal(ops.unpack_w4(pw).cpu().float(), w))
y = ops.conv2d_w4a8(xq, pw, sel)
ref = F.conv2d(x, w).permute(0,2,3,1)
print("1x1 za=128 zw=0 maxerr", float((y.cpu()-ref).abs().max()), "ref max", float(ref.abs().max()))
if float((y.cpu()-ref).abs().max()) > 0:
    yc = y.cpu()
    print("y[0,0,0,:8]", yc[0,0,0,:8].tolist()); print("ref      ", ref[0,0,0,:8].tolist())
    print("y[0,0,:8,0]", yc[0,0,:8,0].tolist()); print("ref      ", ref[0,0,:8,0].tolist())
    # is y a permutation of ref?
    print("sorted equal", torch.equal(torch.sort(yc.reshape(-1))[0], torch.sort(ref.reshape(-1))[0]))
    # single-nonzero probes: x one-hot at (pixel p, channel c), w one-hot
    for (p, c, n) in [(0,0,0),(1,0,0),(0,1,0),(0,0,1),(5,17,9),(3,40,33)]:
        xx = torch.zeros(B, cin, H, W); xx.view(B,cin,-1)[0,c,p] = 1
        ww = torch.zeros(cout, cin, 1, 1); ww[n,c] = 1
        xq2 = ops.quantize_act(xx.permute(0,2,3,1).contiguous().to(DEV), sel)
        pw2 = ops.pack_w4(ww.to(DEV), torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV))
        y2 = ops.conv2d_w4a8(xq2, pw2, sel).cpu().reshape(-1, cout)
        nz = torch.nonzero(y2)
        print("probe", (p,c,n), "->", nz.tolist()[:6], y2[y2!=0][:6].tolist())
# nonzero zero points
x = torch.randn(2, 64, 8, 8)*1.7+0.3; w = torch.randn(48,64,3,3)*0.05; b = torch.randn(48)*0.1
wd, wz = O.init_channelwise(w, 16, "minmax"); ad, az = O.minmax(x, 256)
sel = ops.qsel(qtab(ad, az))
xq = ops.quantize_act(x.permute(0,2,3,1).contiguous().to(DEV), sel)
pw = ops.pack_w4(w.to(DEV), wd.to(DEV), wz.to(DEV), bias=b.to(DEV))
y = ops.conv2d_w4a8(xq, pw, sel, pad=(1,1,1,1)).cpu()
ref = F.conv2d(O.fake_quant(x,ad,az,256), O.fake_quant(w,wd,wz,16), b, padding=1).permute(0,2,3,1)
print("3x3 general maxnorm", float((y-ref).abs().max()/ref.abs().max()))
y = ops.conv2d_w4a8(xq, pw, sel, pad=(1,1,1,1)).cpu()
# f16
pf = ops.pack_w_f16(w.to(DEV), b.to(DEV))
yf = ops.conv2d_f16(x.permute(0,2,3,1).contiguous().to(DEV), pf, pad=(1,1,1,1)).cpu()
reff = F.conv2d(x, w, b, padding=1).permute(0,2,3,1)
print("f16 3x3 maxnorm", float((yf-reff).abs().max()/reff.abs().max()))
