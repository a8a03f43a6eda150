"""Does the materialised-pair sampling still end on the metric run's latents bit for bit -- and does a leg in between (the live calibration
slice of bench.py) disturb it?  usage: python scratch/pair_equal_check.py [live]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
args = argparse.Namespace(batch=int(os.environ.get("SD_BATCH", "8")), ddim_steps=int(os.environ.get("SD_STEPS", "10")))
run, fwd, cpu, info = bench.setup_sd(args, dev, 0, lambda *a: print(*a, file=sys.stderr))
smp = info["sampler"]
outs = []
for i in range(int(os.environ.get("REPEAT", "3"))):
    run(); info["stream"].synchronize()
    outs.append(smp.x.clone())
print("the metric sampling repeated: equal to the first run:", [bool(torch.equal(o, outs[0])) for o in outs],
      "max abs diff", [float((o - outs[0]).abs().max()) for o in outs])
for i in range(2):
    print("materialised right after the metric run:", info["materialised"]())
def poison(pattern):
    """fill (nearly) all free device memory with a byte pattern and release it to torch's caching allocator: later allocations see it"""
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    n = int(free * 0.9) // 4
    t = torch.empty(n, dtype=torch.int32, device=dev)
    t.fill_(pattern)
    torch.cuda.synchronize()
    del t
for name, pat in (("zeros", 0), ("NaN bits", -1), ("0x7f7f7f7f", 0x7f7f7f7f)):
    if "poison" in sys.argv:
        poison(pat)
        print(f"free memory poisoned with {name}:", info["materialised"]())
        s2 = info["new_sampler"]().capture()
        for rep in range(2):
            o2 = s2.sample_nhwc(*info["inputs"]); s2.stream.synchronize()
            print(f"   a NEW sampler of the metric's own kind after the poisoning, sampling {rep}: equal to the metric run {bool(torch.equal(o2, outs[0]))}, "
                  f"max abs diff {float((o2 - outs[0]).abs().max()):.3e}")
        del s2
if "live" in sys.argv:
    lj = bench.run_cali_workload(argparse.Namespace(cali_samples=16, cali_groups=2, cali_iters=50, cali_only="model.input_blocks.1,model.middle_block.1",
                                                    cali_generate=True), dev, 0, 0, 1, lambda *a: None)
    print("live slice:", lj["value"], "s")
    print("materialised after the live slice (metric buffer untouched):", info["materialised"]())
    run(); info["stream"].synchronize()
    print("materialised after the live slice and a fresh metric run:", info["materialised"]())
