"""Narrowing scratch/alloc_dependence.py: which product of a fresh sampler depends on the allocator's leftovers -- the TIB table built by
its constructor, or one eager UNet forward (plain allocations, no arena)?"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
args = argparse.Namespace(batch=int(os.environ.get("SD_BATCH", "8")), ddim_steps=int(os.environ.get("SD_STEPS", "20")))
run, fwd, cpu, info = bench.setup_sd(args, dev, 0, lambda *a: None)
smp = info["sampler"]
eng = smp.eng
x_T, cond, uncond = info["inputs"]
ctx2 = torch.cat([uncond, cond]).contiguous()


def poison(pattern):
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    t = torch.empty(int(free * 0.9) // 4, dtype=torch.int32, device=dev)
    t.fill_(pattern)
    torch.cuda.synchronize()
    del t


def one_forward(pair):
    with torch.cuda.stream(smp.stream):
        eng.step.zero_()
        y = eng.forward(x_T, None, ctx2, pair_prefix=True) if pair else eng.forward(torch.cat([x_T, x_T]).contiguous(), None, ctx2)
        smp.stream.synchronize()
    return y.clone()


tib0 = eng.tib_table.clone()
f0 = {p: one_forward(p) for p in (True, False)}
for pat in (0x7f7f7f7f, -1, 0x7f7f7f7f, 0x3c003c00, 0x7f7f7f7f, 0x01010101, 0x3c003c00):
    poison(pat)
    s2 = info["new_sampler"]()                                  # the constructor rebuilds the engine's TIB table
    tib_same = bool(torch.equal(eng.tib_table, tib0))
    res = []
    for p in (True, False):
        y = one_forward(p)
        d = (y - f0[p]).abs()
        res.append("same" if float(d.max()) == 0 else f"{float(d.max()):.1e}")
    import tfmq_dm_amd.ops as ops
    ar = ops.Arena()
    for tag in ("arena, recorded pass", "arena, replayed pass"):
        ops.set_conv_autotune(eng.tiles)
        try:
            with ops.use_arena(ar):
                y = one_forward(True)
        finally:
            ops.set_conv_autotune(None)
        d = (y - f0[True]).abs()
        res.append(f"{tag}: " + ("same" if float(d.max()) == 0 else f"{float(d.max()):.1e}"))
    del ar
    print(f"{pat & 0xffffffff:08x}: TIB table {'same' if tib_same else 'DIFFERENT %.2e' % float((eng.tib_table - tib0).abs().max())}; eager forward pair-prefix {res[0]}, materialised {res[1]}; {res[2]}; {res[3]}", flush=True)
    del s2
