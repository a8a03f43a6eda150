#!/usr/bin/env python
"""Headline benchmark of the TFMQ-DM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): DDIM CIFAR-10 w4a8 -- DDPM UNet (35.7 M params, 32x32x3),
DDIM-100 'quad' schedule, eta = 0, 256-image batch per GPU, one Finite-Set-Calibration activation
table per step, synthetic N(0,1) latents and random-init weights (no checkpoints offline).
One "step" = one full 100-step sampling of one 256-image batch on every rank (UNet evals + DDIM
updates only, the timing region of sample_diffusion_ldm.py:127-150).  Sampling shards with no
exchange (images are independent): weak scaling, value = total images / max-over-ranks time.

The JSON line also carries
  roofline    : the dominant kernel (w4a8 implicit-GEMM conv, int8 MFMA) -- algorithmic int8 ops of
                every launch of one UNet forward / its HIP-event-measured duration, vs 5 POP/s dense.
  cpu_baseline: the CPU oracle (torch-CPU restatement of the reference's fake-quant path) timed on
                this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_PEAK_TOPS = 5000.0  # dense int8 MFMA peak of MI355X (2x the 2.5 PF bf16 dense peak, MI355X_MICROARCH.md)


def build_quantized_engine(dev, batch, n_steps, seed=1234, log=lambda *a: None):
    import tfmq_dm_amd.ddim.models as M
    from tfmq_dm_amd.ddim.sampler import linear_betas, step_sequence
    from tfmq_dm_amd.engine import DdimUNetEngine, ddim_quant as Q

    torch.manual_seed(seed)
    model = M.random_init(M.Model(M.make_config()), seed)
    cfg = model.engine_cfg()
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    t0 = time.time()
    wq = Q.init_weight_quant(sd, cfg, "mse", 4, dev)
    torch.cuda.synchronize()
    log(f"weight-scale search (mse, per channel): {time.time() - t0:.2f}s")
    names = Q.attach_act_ids(wq, cfg)
    seq = step_sequence("quad", n_steps)
    qtable = torch.zeros(n_steps, len(names), 2, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    eng = DdimUNetEngine(sd, cfg, dev)
    eng.prepare(wq, qtable, step)
    # synthetic Finite-Set Calibration: one group of N(0,1) latents per sampling step, MINMAX scaler
    # (what the reference's running-stat pass ends with), at the benchmark batch so that every conv
    # launch of the process has the shapes of the timed region (keeps rocprof averages comparable).
    calib_batch = batch
    t0 = time.time()
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    groups = []
    for i in reversed(seq):
        x = torch.randn(calib_batch, cfg["resolution"], cfg["resolution"], 3, generator=g).to(dev)
        groups.append((x, torch.full((calib_batch,), float(i), device=dev)))
    Q.calibrate_activations(eng, groups, running_stat=False, init_batch=calib_batch, scaler="minmax")
    torch.cuda.synchronize()
    log(f"synthetic activation calibration ({n_steps} groups x {calib_batch}, minmax): {time.time() - t0:.2f}s")
    return eng, cfg, sd, wq, names, seq, linear_betas()


def conv_roofline(eng, x, n_fwd=3):
    """Per-launch HIP-event timing (on the launch stream) of every w4a8 conv launch of a UNet forward."""
    import tfmq_dm_amd.ops as ops
    rec = []
    ops.set_conv_profile(rec)
    stream = torch.cuda.current_stream()
    try:
        for _ in range(n_fwd):
            eng.forward(x, None)
        stream.synchronize()
    finally:
        ops.set_conv_profile(None)
    tot_ops, tot_ms, tot_bytes, n = 0.0, 0.0, 0.0, 0
    for (e0, e1, nops, kind, nbytes) in rec:
        if kind != "w4a8":
            continue
        tot_ops += nops
        tot_bytes += nbytes
        tot_ms += ops.event_elapsed_ms(e0, e1)
        n += 1
    return tot_ops, tot_ms, n, tot_bytes


def cpu_baseline(cfg, sd, wq, names, qtable, seq, betas, batch=8, steps=4):
    """Oracle (torch-CPU fake-quant UNet + DDIM update) on a bounded sample: `batch` images x `steps`
    of the 100 DDIM steps; images/s extrapolated to the full 100-step schedule."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tfmq_oracle as O
    qt = qtable.cpu()
    sdc = {k: v.cpu() for k, v in sd.items()}
    wqc = {n: {"delta": q.delta.cpu(), "zp": q.zp.cpu(), "alpha": None} for n, q in wq.items()}
    ocfg = dict(cfg)
    x = torch.randn(batch, 3, cfg["resolution"], cfg["resolution"])

    def model_fn(xt, t, cnt):
        aq = {n: (qt[cnt, i, 0], qt[cnt, i, 1]) for i, n in enumerate(names)}
        return O.ddim_unet_forward(sdc, ocfg, xt, t, O.QuantSpec(wq=wqc, aq=aq))

    with torch.no_grad():
        t0 = time.time()
        O.generalized_steps(x, seq, model_fn, betas, until=steps + 1)
        dt = time.time() - t0
    per_image_full = dt / steps * len(seq) / batch
    return 1.0 / per_image_full, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ddim-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the TFMQ hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    from tfmq_dm_amd.ddim.sampler import GraphDdimSampler
    eng, cfg, sd, wq, names, seq, betas = build_quantized_engine(dev, args.batch, args.ddim_steps, log=log)
    sampler = GraphDdimSampler(eng, seq, betas, args.batch).capture()
    log(f"captured DDIM step graph; activations arena {sampler.arena.nbytes() / 2**30:.2f} GiB")
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x_T = torch.randn(args.batch, cfg["resolution"], cfg["resolution"], 3, generator=g).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sampler.sample_nhwc(x_T)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sampler.sample_nhwc(x_T)
    sampler.stream.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    x0 = sampler.x
    finite = bool(torch.isfinite(x0).all().item())

    out = None
    if rank == 0:
        images = args.batch * world * args.steps
        value = images / dt
        # ---- roofline of the dominant kernel (HIP events around every launch, launch stream)
        with torch.cuda.stream(sampler.stream):
            eng.step.zero_()
            tot_ops, tot_ms, n_launch, tot_bytes = conv_roofline(eng, sampler.x)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):   # PMC bytes (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes) of the same kernel
            tj = json.load(open(tpath))["kernels"]
            ks = [v for k, v in tj.items() if k.startswith("void k_conv_igemm<true")]
            if ks:
                traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ks) / sum(v["launches"] for v in ks)
                traffic_src = "profiles/r01_traffic.json (rocprofv3 --pmc, eager forwards of the same workload)"
        achieved = tot_ops / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        roof = {"bound": "mfma", "kernel": "k_conv_igemm<int8> (w4a8 implicit-GEMM conv / linear)",
                "achieved": round(achieved, 2), "peak": INT8_PEAK_TOPS, "unit": "TOP/s",
                "frac": round(achieved / INT8_PEAK_TOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch (PMC)",
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(tot_bytes / max(n_launch, 1)),
                "hbm_achieved_TBps": round(tot_bytes / (tot_ms * 1e-3) / 1e12, 3) if tot_ms > 0 else None,
                "launches_timed": n_launch, "avg_launch_us": round(tot_ms * 1e3 / max(n_launch, 1), 2),
                "algorithmic_ops_per_forward": tot_ops / 3.0}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            v, secs = cpu_baseline(cfg, sd, wq, names, eng.qtable, seq, betas)
            cpu = {"value": round(v, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"oracle (torch-CPU fake-quant UNet, same weights/act tables): 8 images x 4 of the "
                             f"{len(seq)} DDIM steps = {secs:.1f}s, extrapolated to the full schedule"}
        out = {
            "metric": "DDIM images/sec, w4a8 (headline metric of BASELINE.json on its configs[1] workload)",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int8 (u8 act bins x int4 weights, int32 accumulate; f16 for the un-quantised layers/attention)",
            "data": "synthetic: N(0,1) latents, random-init weights (zero params re-drawn N(0,0.02^2)), synthetic FSC tables",
            "config": {"workload": "DDIM CIFAR-10 w4a8 on MI355X: DDPM UNet 35.7M, 32x32x3, DDIM-100 quad eta=0, "
                                   f"{args.batch}-image batch per GPU (BASELINE.json configs[1])",
                       "batch_per_gpu": args.batch, "ddim_steps": len(seq), "unet_evals_per_step": len(seq),
                       "parallelism": f"replicas x{world} (no data-path collective)"},
            "finite": finite, "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
