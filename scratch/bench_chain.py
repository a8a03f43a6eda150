"""tfmq_row_chain vs the launches it replaces at the SD 64x64 level (UNet batch BATCH, default 128): the pre-attention chain (GroupNorm
apply + quantise, proj_in, LayerNorm + quantise, fused q|k|v) and the chain between the two attentions (to_out + residual, LayerNorm +
quantise, to_q): us per call, bit identity."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "128"))
C = int(os.environ.get("CH", "320"))
T = 4096 * 320 * 320 // (C * C)       # 4096 tokens at C = 320, 1024 at C = 640
M = B * T
g = torch.Generator().manual_seed(0)


def lin(cout, cin, bias=True):
    w = (torch.randn(cout, cin, generator=g) * 0.08).to(dev)
    qp = ops.minmax_to_qparam(ops.minmax(w, cout), 16)
    return ops.pack_w4(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), None, torch.zeros(cout, device=dev) if bias else None)


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, y


qt = torch.tensor([[[0.03, 128.0], [0.031, 131.0], [0.027, 125.0], [0.05, 120.0]]], device=dev)
sel = [ops.qsel(qt, i) for i in range(4)]
prod = lin(C, 64)
xq0 = torch.randint(-128, 128, (B, T, 1, 64), dtype=torch.int8, device=dev)
x = ops.conv2d_w4a8(xq0, prod, sel[0], out_f16=True, want_stats=True)
gn_g, gn_b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
ln_g, ln_b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
pin, qkv, to_out, to_q = lin(C, C), lin(3 * C, C, False), lin(C, C), lin(C, C, False)


def pre_launches():
    xq, _, _ = ops.groupnorm(x, gn_g, gn_b, 1e-6, False, sel[1])
    h = ops.conv2d_w4a8(xq, pin, sel[1], out_f16=True)
    hq = ops.layernorm(h.reshape(B, T, C), ln_g, ln_b, 1e-5, sel[2])[0]
    y16, vt = ops.conv2d_w4a8(hq.reshape(B, T, 1, C), qkv, sel[2], out_f16=True, t_col0=2 * C)
    return h, y16, vt


def pre_chain():
    ab = ops.gn_affine_from_stats(x, gn_g, gn_b, 1e-6)
    o = ops.row_chain(x.reshape(M, C), T, [dict(pw=pin, aq=sel[1], ln=True), dict(pw=qkv, aq=sel[2], t_col0=2 * C)], gn=ab, ln=(ln_g, ln_b, 1e-5))
    return o[0][0], o[1][0], o[1][1]


ul, yl = t(pre_launches)
uc, yc = t(pre_chain)
same = torch.equal(yl[0].reshape(M, C), yc[0]) and torch.equal(yl[1].reshape(M, 3 * C)[:, :2 * C], yc[1][:, :2 * C]) and torch.equal(yl[2], yc[2])
print(f"pre chain  (GN apply, proj_in, LN, q|k|v) UNet batch {B}: launches {ul:8.1f} us   chain {uc:8.1f} us   {'bit-identical' if same else 'MISMATCH'}", flush=True)

o = torch.randint(-128, 128, (M, C), dtype=torch.int8, device=dev)
hres = (torch.randn(M, C, device=dev) * 1.5).half()


def mid_launches():
    x1 = ops.conv2d_w4a8(o.reshape(1, M, 1, C), to_out, sel[0], residual=hres.reshape(1, M, 1, C), out_f16=True, want_stats=False).reshape(M, C)
    xq = ops.layernorm(x1, ln_g, ln_b, 1e-5, sel[1])[0]
    return x1, ops.conv2d_w4a8(xq.reshape(1, M, 1, C), to_q, sel[1], out_f16=True).reshape(M, C)


def mid_chain():
    r = ops.row_chain(o, T, [dict(pw=to_out, aq=sel[0], residual=hres, ln=True), dict(pw=to_q, aq=sel[1])], ln=(ln_g, ln_b, 1e-5))
    return r[0][0], r[1][0]


ul, yl = t(mid_launches)
uc, yc = t(mid_chain)
same = torch.equal(yl[0], yc[0]) and torch.equal(yl[1], yc[1])
print(f"mid chain  (to_out + res, LN, to_q)       UNet batch {B}: launches {ul:8.1f} us   chain {uc:8.1f} us   {'bit-identical' if same else 'MISMATCH'}", flush=True)
