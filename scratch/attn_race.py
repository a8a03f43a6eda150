import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
torch.manual_seed(0)
for (B, heads, T) in [(2, 8, 256), (1, 2, 1024), (4, 8, 4096)]:
    d = 40; C = heads * d
    q = torch.randn(B, T, C).half().to(DEV); k = torch.randn(B, T, C).half().to(DEV); v = torch.randn(B, T, C)
    vt = v.transpose(1, 2).contiguous().half().to(DEV)
    qk = torch.cat([q, k], -1).contiguous()
    ref, _ = ops.attention_f16(q, k, vt, heads, d ** -0.5)
    bad = 0
    for it in range(30):
        o1, _ = ops.attention_f16(q, k, vt, heads, d ** -0.5)
        o2, _ = ops.attention_f16(qk[..., :C], qk[..., C:], vt, heads, d ** -0.5)
        for name, o in (("sep", o1), ("fused", o2)):
            ne = (o != ref)
            if ne.any():
                bad += 1
                idx = ne.nonzero()
                if bad <= 3: print(f"B{B} T{T} it{it} {name}: {int(ne.sum())} mismatches; first {idx[0].tolist()} last {idx[-1].tolist()}; tokens {sorted(set(idx[:,1].tolist()))[:10]} chans {sorted(set((idx[:,2] % d).tolist()))[:12]} maxdiff {float((o-ref).abs().max()):.3e}")
    print(f"B{B} h{heads} T{T}: {bad} bad of 60")
