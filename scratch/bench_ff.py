"""tfmq_ff_fused vs the three launches it replaces (LayerNorm + quantise, GEGLU projection, ff.net.2 + residual) at the SD 64x64 level:
us per call, effective TOP/s of the two GEMMs, bit identity.  BATCH (UNet batch, default 128), Q8=1: int8 output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "128"))
C, inner, T = 320, 1280, 4096
M = B * T
g = torch.Generator().manual_seed(0)
x16 = (torch.randn(M, C, generator=g) * 1.5).half().to(dev)
gamma, beta = (torch.randn(C, generator=g) * 0.3 + 1).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
w1 = (torch.randn(2 * inner, C, generator=g) * 0.08).to(dev)
w2 = (torch.randn(C, inner, generator=g) * 0.04).to(dev)
perm = ops.geglu_perm(inner, dev)
qp1 = ops.minmax_to_qparam(ops.minmax(w1, 2 * inner), 16)
qp2 = ops.minmax_to_qparam(ops.minmax(w2, C), 16)
pw1 = ops.pack_w4(w1[perm].contiguous(), qp1[:, 0][perm].contiguous(), qp1[:, 1][perm].contiguous(), None, torch.zeros(2 * inner, device=dev))
pw2 = ops.pack_w4(w2, qp2[:, 0].contiguous(), qp2[:, 1].contiguous(), None, torch.zeros(C, device=dev))
qt = torch.tensor([[[0.03, 128.0], [0.04, 100.0], [0.05, 120.0]]], device=dev)
s0, s2, so = ops.qsel(qt, 0), ops.qsel(qt, 1), ops.qsel(qt, 2)
q8 = os.environ.get("Q8") == "1"


def chain():
    xq = ops.layernorm(x16, gamma, beta, 1e-5, s0)[0]
    gg = ops.conv2d_w4a8(xq.reshape(1, M, 1, C), pw1, s0, geglu_oq=s2)
    kw = dict(out_q8=so) if q8 else dict(out_f16=True)
    return ops.conv2d_w4a8(gg, pw2, s2, residual=x16.reshape(1, M, 1, C), want_stats=False, **kw).reshape(M, C)


def fused():
    return ops.ff_fused(x16, gamma, beta, 1e-5, s0, pw1, s2, pw2, out_q8=so if q8 else None)


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


uc, yc = t(chain)
uf, yf = t(fused)
nops = 2.0 * M * (2 * inner * C + inner * C)
print(f"UNet batch {B} x {T} tokens, C {C}, inner {inner}, {'int8' if q8 else 'fp16'} out: three launches {uc:8.1f} us ({nops / uc / 1e6:6.0f} TOP/s)   "
      f"fused {uf:8.1f} us ({nops / uf / 1e6:6.0f} TOP/s)   {'bit-identical' if torch.equal(yc, yf) else 'MISMATCH'}", flush=True)

# ---- the Linears in front of / behind the feed-forward in the same launch (attn2.to_out + residual; proj_out + x_in + statistics)
if os.environ.get("CHAIN", "1") == "1":
    w0 = (torch.randn(C, C, generator=g) * 0.08).to(dev)
    w3 = (torch.randn(C, C, generator=g) * 0.08).to(dev)
    q0, q3 = ops.minmax_to_qparam(ops.minmax(w0, C), 16), ops.minmax_to_qparam(ops.minmax(w3, C), 16)
    pw0 = ops.pack_w4(w0, q0[:, 0].contiguous(), q0[:, 1].contiguous(), None, torch.zeros(C, device=dev))
    pw3 = ops.pack_w4(w3, q3[:, 0].contiguous(), q3[:, 1].contiguous(), None, torch.zeros(C, device=dev))
    o2 = torch.randint(-128, 128, (M, C), dtype=torch.int8, device=dev)
    xin = (torch.randn(M, C, device=dev)).half()

    def launches():
        x2 = ops.conv2d_w4a8(o2.reshape(1, M, 1, C), pw0, so, residual=x16.reshape(1, M, 1, C), out_f16=True, want_stats=False).reshape(M, C)
        b = ops.ff_fused(x2, gamma, beta, 1e-5, s0, pw1, s2, pw2, out_q8=so)
        return ops.conv2d_w4a8(b.reshape(B, T, 1, C), pw3, so, residual=xin.reshape(B, T, 1, C), out_f16=True, want_stats=True).reshape(M, C)

    def one():
        return ops.ff_fused(None, gamma, beta, 1e-5, s0, pw1, s2, pw2, out_q8=so, pre=dict(xq=o2, pw=pw0, aq=so, residual=x16),
                            post=dict(pw=pw3, residual=xin, stats=True, hw=T))[1]
    ul, yl = t(launches)
    uo, yo = t(one)
    print(f"to_out + FF + proj_out: three launches {ul:8.1f} us   one launch {uo:8.1f} us   {'bit-identical' if torch.equal(yl, yo) else 'MISMATCH'}", flush=True)
