#!/usr/bin/env python
"""Headline benchmark of the TFMQ-DM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload sd|cifar]
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workloads
  sd     (default; BASELINE.json's metric config, configs[3]): Stable Diffusion v1-4 UNet (859.5 M params),
         w4a8, DDIM-50 eta 0, classifier-free guidance 7.5 (UNet batch = 2 x images), 64x64x4 latents,
         77x768 context, one Finite-Set-Calibration activation table per step.  One "step" = one full
         50-step sampling of `--batch` images on every rank.
  cifar  (configs[1]): DDPM UNet 35.7 M, 32x32x3, DDIM-100 quad, 256-image batch per GPU.
Synthetic data in both cases (no checkpoints / datasets offline): N(0,1) latents and context, random-init
weights with the reference's initialisers (zero parameters re-drawn N(0, 0.02^2)), per-channel MSE weight scales
computed on the device, synthetic FSC tables (MINMAX of the actual activations at every step).
Timing region = UNet evaluations + sampler updates only (sample_diffusion_ldm.py:127-150); text encoder and
VAE are glue and excluded.  Sampling shards with no exchange: weak scaling, value = total images / max time.

The JSON line also carries
  roofline    : dominant kernel = the w4a8 implicit-GEMM (int8 MFMA): algorithmic int8 ops of every launch of
                UNet forwards / their HIP-event-measured durations (launch stream), vs the 5 POP/s dense int8
                peak; `traffic` = PMC HBM bytes per launch from profiles/ (rocprofv3, separate passes).
  cpu_baseline: the CPU oracle (torch-CPU restatement of the reference's fake-quant path) on this box's host
                cores, bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this host driver needs dmabuf IPC (RCCL / tensor sharing across ranks fail with the legacy mode); set before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# TFMQ_BENCH_ONE_DEVICE=1: exercise the whole N > 1 flow (self-spawn, barriers, max-over-ranks time, sharded-calibration leg, one
# JSON line from rank 0) on a box with ONE GPU -- all ranks share cuda:0, torch.distributed runs on gloo, the C ABI's RCCL
# communicator is not created (linklink falls back to torch.distributed).  A test mode: its number means nothing.
ONE_DEVICE = os.environ.get("TFMQ_BENCH_ONE_DEVICE") == "1"
INT8_PEAK_TOPS = 5000.0  # dense int8 MFMA peak of MI355X (2x the 2.5 PF bf16 dense peak, MI355X_MICROARCH.md)
F16_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (the un-quantised convs run fp16 operands)
HBM_ACHIEVABLE_TBPS = 6.3  # what a float4 copy reaches of the 8 TB/s (MI355X_MICROARCH.md): the HBM roof the per-launch floors are priced at


# ------------------------------------------------------------------------------------------------ cifar
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
    # synthetic Finite-Set Calibration at the benchmark batch (every conv launch of the process then has the
    # shapes of the timed region, which keeps rocprof per-kernel averages comparable with the live ones)
    t0 = time.time()
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    groups = []
    for i in reversed(seq):
        x = torch.randn(batch, cfg["resolution"], cfg["resolution"], 3, generator=g).to(dev)
        groups.append((x, torch.full((batch,), float(i), device=dev)))
    Q.calibrate_activations(eng, groups, running_stat=False, init_batch=batch, scaler="minmax")
    torch.cuda.synchronize()
    log(f"synthetic activation calibration ({n_steps} groups x {batch}, minmax): {time.time() - t0:.2f}s")
    return eng, cfg, sd, wq, names, seq, linear_betas()


def setup_cifar(args, dev, rank, log):
    from tfmq_dm_amd.ddim.sampler import GraphDdimSampler
    batch = args.batch or 256
    n_steps = args.ddim_steps or 100
    eng, cfg, sd, wq, names, seq, betas = build_quantized_engine(dev, batch, n_steps, log=log)
    sampler = GraphDdimSampler(eng, seq, betas, batch).capture()
    log(f"captured DDIM step graph; activations arena {sampler.arena.nbytes() / 2**30:.2f} GiB")
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x_T = torch.randn(batch, cfg["resolution"], cfg["resolution"], 3, generator=g).to(dev)

    def run():
        sampler.sample_nhwc(x_T)

    # the sampler's first sampling carries its one-off fp16-stream check (ddim/sampler.py: fp16_stream_overflowed): here, outside
    # the timed region whatever --warmup is
    if getattr(args, "first_sampling", True):      # (scratch/pmc_forward.py: counter collection cannot take a graph replay)
        t0 = time.time()
        run()
        sampler.stream.synchronize()
        log(f"first sampling (graph upload + fp16-stream check): {time.time() - t0:.2f}s, fp16 stream {'on' if eng.stream_f16 else 'OFF (fallback)'}")

    def fwd():
        eng.forward(sampler.x, None)

    def cpu():
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import tfmq_oracle as O
        qt = eng.qtable.cpu()
        sdc = {k: v.cpu() for k, v in sd.items()}
        wqc = {n: {"delta": q.delta.cpu(), "zp": q.zp.cpu(), "alpha": None} for n, q in wq.items()}
        cb, cs = 16, 8                   # bounded sample: ~15 s of host work on this box
        x = torch.randn(cb, 3, cfg["resolution"], cfg["resolution"])

        def model_fn(xt, t, cnt):
            aq = {n: (qt[cnt, i, 0], qt[cnt, i, 1]) for i, n in enumerate(names)}
            return O.ddim_unet_forward(sdc, dict(cfg), xt, t, O.QuantSpec(wq=wqc, aq=aq))
        with torch.no_grad():
            t0 = time.time()
            O.generalized_steps(x, seq, model_fn, betas, until=cs + 1)
            dt = time.time() - t0
        return 1.0 / (dt / cs * len(seq) / cb), f"{cb} images x {cs} of the {len(seq)} DDIM steps = {dt:.1f}s, extrapolated to the full schedule"

    info = dict(batch=batch, sync=sampler.stream.synchronize, finite=lambda: bool(torch.isfinite(sampler.x).all().item()),
                stream=sampler.stream, step=eng.step,
                workload=("DDIM CIFAR-10 w4a8 on MI355X: DDPM UNet 35.7M, 32x32x3, DDIM-100 quad eta=0, "
                          f"{batch}-image batch per GPU (BASELINE.json configs[1])"),
                extra={"batch_per_gpu": batch, "ddim_steps": len(seq), "unet_evals_per_step": len(seq)})
    return run, fwd, cpu, info


# ------------------------------------------------------------------------------------------------ stable diffusion
LDM_PRESETS = {
    # BASELINE.json configs[3] (the metric's config), configs[4] and configs[2]
    "sd": dict(unet="SD_V1_UNET", latent=(4, 64, 64), ctx=(77, 768), scale=7.5, steps=50, batch=64,
               name="Stable Diffusion v1-4 UNet", tail="BASELINE.json configs[3] = the metric's config"),
    "cin256": dict(unet="CIN256_V2_UNET", latent=(3, 64, 64), ctx=(1, 512), scale=3.0, steps=20, batch=64,
                   name="LDM ImageNet-256 class-conditional UNet (cin256-v2)", tail="BASELINE.json configs[4]; latent_imagenet_diffusion.py --ddim_steps 20 --scale 3.0"),
    "celeba": dict(unet="CELEBAHQ_LDM_VQ4_UNET", latent=(3, 64, 64), ctx=None, scale=1.0, steps=200, batch=64,
                   name="LDM-4 CelebA-HQ 256 unconditional UNet", tail="BASELINE.json configs[2]; sample_diffusion_ldm.py -c 200 -e 0.0"),
}


def setup_sd(args, dev, rank, log, preset="sd"):
    import numpy as np
    import tfmq_dm_amd.ldm.unet as U
    import tfmq_dm_amd.ops as ops
    from tfmq_dm_amd.ddim.models import random_init
    from tfmq_dm_amd.engine import LayerQ, LdmUNetEngine
    from tfmq_dm_amd.ldm.sampler import GraphLatentDdimSampler, alphas_cumprod_linear, ddim_timesteps

    P = LDM_PRESETS[preset]
    batch = args.batch or P["batch"]
    S = args.ddim_steps or P["steps"]
    scale = P["scale"]
    LAT, CTX = P["latent"], P["ctx"]
    LC, LH, LW = LAT
    NB = batch * (2 if CTX is not None else 1)         # UNet batch: cond + uncond halves under guidance
    t0 = time.time()
    torch.manual_seed(40)
    model = random_init(U.UNetModel(**getattr(U, P["unet"])), 40)
    cfg = model.engine_cfg()
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    n_params = sum(v.numel() for v in sd.values()) / 1e6
    log(f"{P['name']} random init ({n_params:.1f} M params): {time.time() - t0:.1f}s")
    # QuantLayers in module order = every Conv2d / Linear except skip_connection / op (quant_model.py:57-58)
    qnames = [k[:-7] for k in sd if k.endswith(".weight") and sd[k].dim() in (2, 4) and "skip_connection" not in k
              and not k.endswith(".op.weight")]
    fp = {qnames[0], qnames[2], qnames[-1]}
    no_act = fp | {qnames[1], qnames[3]}
    t0 = time.time()
    wq = {}
    for n in qnames:
        if n in fp:
            continue
        w = sd[n + ".weight"].to(dev, torch.float32).contiguous()
        qp = ops.mse_search(w, w.shape[0], 16)
        wq[n] = LayerQ(qp[:, 0].contiguous(), qp[:, 1].contiguous(), None, None)
    torch.cuda.synchronize()
    log(f"weight-scale search (mse, per channel, {len(wq)} layers): {time.time() - t0:.2f}s")
    act_names = [n for n in qnames if n not in no_act]
    for i, n in enumerate(act_names):
        wq[n].qid = i
    qtable = torch.zeros(S, len(act_names), 2, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    eng = LdmUNetEngine(sd, cfg, dev)
    eng.prepare(wq, qtable, step)
    # synthetic FSC: MINMAX of the activations at every step, on UNet batch 2 x batch
    t0 = time.time()
    g = torch.Generator(device="cpu").manual_seed(41 + rank)
    ctx = None if CTX is None else torch.randn(NB, CTX[0], CTX[1], generator=g).to(dev)
    ts = np.flip(ddim_timesteps(S))
    for k, tv in enumerate(ts):
        eng.set_calibration("init_minmax", k)
        x = torch.randn(NB, LH, LW, LC, generator=g).to(dev)
        eng.forward(x, torch.full((NB,), float(tv), device=dev), ctx)
    eng.set_calibration(None)
    eng.prepare(wq, eng.qtable, step)   # re-evaluate the sibling-quantizer fusion with the calibrated table
    torch.cuda.synchronize()
    log(f"synthetic activation calibration ({S} groups x {NB}, minmax): {time.time() - t0:.2f}s")
    sampler = GraphLatentDdimSampler(eng, S, batch, LAT, CTX, scale=scale, alphas_cumprod=alphas_cumprod_linear()).capture()
    log(f"captured DDIM step graph; activations arena {sampler.arena.nbytes() / 2**30:.2f} GiB")
    x_T = torch.randn(batch, LH, LW, LC, generator=g).to(dev)
    cond, uncond = (None, None) if CTX is None else (ctx[batch:].contiguous(), ctx[:batch].contiguous())

    def run():
        if CTX is None:
            sampler.sample_nhwc(x_T)
        else:
            sampler.sample_nhwc(x_T, cond, uncond)

    if getattr(args, "first_sampling", True):      # (scratch/pmc_forward.py: counter collection cannot take a graph replay)
        t0 = time.time()
        run()        # one-off fp16-stream check of the sampler's first sampling: outside the timed region whatever --warmup is
        sampler.stream.synchronize()
        log(f"first sampling (graph upload + fp16-stream check): {time.time() - t0:.2f}s, fp16 stream {'on' if eng.stream_f16 else 'OFF (fallback)'}")

    def fwd():
        if CTX is None:
            eng.forward(sampler.x, None)
        elif sampler.pair_prefix:
            eng.forward(sampler.x, None, sampler.ctx2, pair_prefix=True)
        else:
            eng.forward(sampler.x2, None, sampler.ctx2)

    def materialised():
        """The same sampling with the guidance pair materialised as a 2B batch in front of the UNet (TFMQ_PAIR_PREFIX=0: the two
        members' shared prefix -- conv_in ... first self attention -- computed twice, as the reference's cat([x] * 2) does); the
        metric's run computes it once per pair, bit-identical output (tests/test_engine_ldm_gpu.py)."""
        old = os.environ.get("TFMQ_PAIR_PREFIX")
        os.environ["TFMQ_PAIR_PREFIX"] = "0"
        try:
            ms = GraphLatentDdimSampler(eng, S, batch, LAT, CTX, scale=scale, alphas_cumprod=alphas_cumprod_linear()).capture()
        finally:
            if old is None:
                del os.environ["TFMQ_PAIR_PREFIX"]
            else:
                os.environ["TFMQ_PAIR_PREFIX"] = old
        ms.sample_nhwc(x_T, cond, uncond)
        ms.stream.synchronize()
        t0 = time.perf_counter()
        out = ms.sample_nhwc(x_T, cond, uncond)
        ms.stream.synchronize()
        dt = time.perf_counter() - t0
        same = bool(torch.equal(out, sampler.x)) if sampler.gid is not None else None
        if same is False:
            df = (out - sampler.x).abs()
            log(f"materialised pair vs metric run: {int((df > 0).sum())} of {df.numel()} latent values differ, max abs {float(df.max()):.3e}, "
                f"images touched {int((df.reshape(df.shape[0], -1).amax(dim=1) > 0).sum())} of {df.shape[0]}")
        return {"images_per_s": round(batch / dt, 3), "final_latents_equal_to_the_metric_run": same,
                "note": "guidance pair materialised as a 2B batch (TFMQ_PAIR_PREFIX=0); the metric's run shares the pair's common prefix"}

    def gelu_exact():
        """The same sampling with the 5e-7 erf GELU in every fused GEGLU epilogue (TFMQ_GELU_EXACT=1: out_mode 2 of the pointwise kernel, the
        feed-forward as three launches) instead of the consumer-sized form the metric's run uses (TFMQ_OUT_GEGLU_Q8_FAST: |dPhi| <= 2.8e-5,
        bins within 1, < 2e-3 of them moved on identical inputs -- tests/test_geglu_fast_gpu.py).  Reported beside `value`."""
        old = os.environ.get("TFMQ_GELU_EXACT")
        os.environ["TFMQ_GELU_EXACT"] = "1"
        try:
            ms = GraphLatentDdimSampler(eng, S, batch, LAT, CTX, scale=scale, alphas_cumprod=alphas_cumprod_linear()).capture()
        finally:
            if old is None:
                del os.environ["TFMQ_GELU_EXACT"]
            else:
                os.environ["TFMQ_GELU_EXACT"] = old
        args_ = (x_T,) if CTX is None else (x_T, cond, uncond)
        ms.sample_nhwc(*args_)
        ms.stream.synchronize()
        t0 = time.perf_counter()
        out = ms.sample_nhwc(*args_)
        ms.stream.synchronize()
        dt = time.perf_counter() - t0
        ref = sampler.x.float()
        rel = float((out.float() - ref).norm() / ref.norm()) if sampler.gid is not None else None
        return {"images_per_s": round(batch / dt, 3), "final_latents_rel_l2_vs_metric_run": None if rel is None else round(rel, 5),
                "note": "TFMQ_GELU_EXACT=1: erf-form GELU (|error| <= 5e-7) in the GEGLU epilogues, feed-forward as three launches; one sampling after a warm one"}

    def cpu():
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import tfmq_oracle as O
        qt = eng.qtable.cpu()
        sdc = {k: v.cpu() for k, v in sd.items()}

        def shp(n, v):
            return v.cpu().reshape((-1,) + (1,) * (sdc[n + ".weight"].dim() - 1))
        wqc = {n: {"delta": shp(n, q.delta), "zp": shp(n, q.zp), "alpha": None} for n, q in wq.items()}
        cs, ci = int(os.environ.get("TFMQ_BENCH_CPU_STEPS", "4")), int(os.environ.get("TFMQ_BENCH_CPU_IMAGES", "1"))   # 4 DDIM steps of 1 image (UNet batch 2 with guidance): bounded sample (~70 s), host cores busy
        gcpu = torch.Generator().manual_seed(4242)
        xc = torch.randn(ci, LC, LH, LW, generator=gcpu)
        tsn, _, _ = O.ldm_ddim_schedule(O.ldm_alphas_cumprod(), S)
        eps_ref, inputs = [], []
        with torch.no_grad():
            t0 = time.time()
            for i, stp in enumerate(list(np.flip(tsn))[:cs]):
                aq = {n: (qt[i, j, 0], qt[i, j, 1]) for j, n in enumerate(act_names)}
                tt = torch.full((ci * (1 if CTX is None else 2),), int(stp), dtype=torch.long)
                if CTX is None:
                    xin, cin_ = xc, None
                else:
                    c1, u1 = torch.randn(ci, CTX[0], CTX[1], generator=gcpu), torch.randn(ci, CTX[0], CTX[1], generator=gcpu)
                    xin, cin_ = torch.cat([xc] * 2), torch.cat([u1, c1])
                e = O.ldm_unet_forward(sdc, dict(cfg), xin, tt, cin_, O.QuantSpec(wq=wqc, aq=aq))
                if i < 2:
                    eps_ref.append(e)
                    inputs.append((xin, tt, cin_))
            dt = time.time() - t0
        info["parity_inputs"] = (inputs, eps_ref)          # the engine is evaluated on the same inputs / rows (parity leg below)
        return ci / (dt / cs * S), (f"{ci} images (UNet batch {ci * (1 if CTX is None else 2)}) x {cs} of the {S} DDIM steps = {dt:.1f}s on "
                                    f"{torch.get_num_threads()} threads, extrapolated to the full schedule")

    def parity():
        """eps of the HIP engine against the CPU oracle (the reference's fp32 fake-quant arithmetic) on the inputs the cpu_baseline leg just
        evaluated: the benchmarked fast mode (fp16 activation stream, fp16-operand attention / un-quantised convs, consumer-sized GELU) and
        the exact mode (TFMQ_EXACT_FP=1: fp32 stream, exact-fp32 MFMA for every un-quantised product) built on the same weights and
        tables, plus the exact mode's forward time.  rel-L2 over the first two DDIM steps' UNet calls."""
        inputs, eps_ref = info.get("parity_inputs", (None, None))
        if not inputs:
            return None

        def rel(e_, r_):
            return float((e_ - r_).norm() / r_.norm())

        def run_engine(e_):
            out = []
            with torch.cuda.stream(sampler.stream):
                for i, (xin, tt, cin_) in enumerate(inputs):
                    e_.step.fill_(i)
                    y = e_.forward(xin.permute(0, 2, 3, 1).contiguous().to(dev), tt.float().to(dev), None if cin_ is None else cin_.to(dev))
                    out.append(y.permute(0, 3, 1, 2).float().cpu())
                e_.step.zero_()
                sampler.stream.synchronize()
            return out
        res = {"inputs": f"UNet batch {inputs[0][0].shape[0]}, the first {len(inputs)} DDIM steps' tables", "reference": "oracle/tfmq_oracle.py on the host (fp32 fake-quant path)"}
        fast = run_engine(eng)
        res["eps_rel_l2_fast"] = [round(rel(a, b), 5) for a, b in zip(fast, eps_ref)]
        old = os.environ.get("TFMQ_EXACT_FP")
        os.environ["TFMQ_EXACT_FP"] = "1"
        try:
            ex = LdmUNetEngine(sd, cfg, dev)
            ex.prepare(wq, eng.qtable, torch.zeros(1, dtype=torch.int32, device=dev))
            exact = run_engine(ex)
            res["eps_rel_l2_exact"] = [round(rel(a, b), 5) for a, b in zip(exact, eps_ref)]
            res["gelu"] = {"fast_leg": "consumer-sized GELU (TFMQ_OUT_GEGLU_Q8_FAST), as in the timed region",
                           "exact_leg": "erf form, |error| <= 5e-7 (the exact-fp engine passes geglu_exact; round 5)"}
            xin, tt, cin_ = inputs[0]
            nb = 16
            xb = torch.randn(nb, LH, LW, LC, device=dev)
            cb = None if cin_ is None else torch.randn(nb, CTX[0], CTX[1], device=dev)
            tb = torch.full((nb,), float(tt[0]), device=dev)
            ex.forward(xb, tb, cb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                ex.forward(xb, tb, cb)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 2 * 1e3
            per_img = nb // (1 if CTX is None else 2)
            res["exact_fp"] = {"ms_per_unet_forward": round(ms, 1), "unet_batch": nb,
                               "images_per_s_equivalent": round(per_img / (S * ms * 1e-3), 3),
                               "note": (f"TFMQ_EXACT_FP=1 engine, eager forwards at UNet batch {nb} ({per_img} images): images/s = images / ({S} steps x forward time); "
                                        "the sampler update is negligible beside it.  Diagnostics mode, not the metric")}
            del ex
        except Exception as e_:       # noqa: BLE001 -- a diagnostics leg must not take the line down
            res["exact_error"] = f"{type(e_).__name__}: {e_}"
        finally:
            if old is None:
                del os.environ["TFMQ_EXACT_FP"]
            else:
                os.environ["TFMQ_EXACT_FP"] = old
            torch.cuda.empty_cache()
        return res

    def plms():
        """The README's SD recipe samples with PLMS (S + 1 UNet calls): reported beside the metric, one
        sampling after a warm one, on the same engine, tables and inputs."""
        from tfmq_dm_amd.ldm.sampler import GraphLatentPlmsSampler
        ps = GraphLatentPlmsSampler(eng, S, batch, LAT, CTX, scale=scale, alphas_cumprod=alphas_cumprod_linear()).capture()
        ps.sample_nhwc(x_T, cond, uncond)
        ps.stream.synchronize()
        t0 = time.perf_counter()
        out = ps.sample_nhwc(x_T, cond, uncond)
        ps.stream.synchronize()
        dt = time.perf_counter() - t0
        return {"images_per_s": round(batch / dt, 3), "unet_evals": S + 1, "finite": bool(torch.isfinite(out).all().item()),
                "sampler": "PLMS-50 (Adams-Bashforth 1-4), CFG 7.5, four captured step graphs"}

    def sweep(batches=(1, 4, 8, 16, 32)):
        """images/s by images per GPU (SURVEY 8d asks for 1-32; the metric batch itself is the timed region): one full DDIM-50
        sampling per batch size on its own captured graph, after a warm one.  UNet forward time per step and the time the
        int8 weight operand alone would need from HBM are printed beside it: below ~8 images the forward is bound by the
        latency of its ~600 launches, not by weight bytes."""
        out = {}
        wbytes = sum(l.p.w8.numel() for l in eng.layers.values() if getattr(l.p, "w8", None) is not None)
        for b in batches:
            sp = GraphLatentDdimSampler(eng, S, b, LAT, CTX, scale=scale, alphas_cumprod=alphas_cumprod_linear()).capture()
            xb, cb, ub = x_T[:b].contiguous(), cond[:b].contiguous(), uncond[:b].contiguous()
            sp.sample_nhwc(xb, cb, ub)
            sp.stream.synchronize()
            t0 = time.perf_counter()
            sp.sample_nhwc(xb, cb, ub)
            sp.stream.synchronize()
            dt = time.perf_counter() - t0
            out[str(b)] = {"images_per_s": round(b / dt, 3), "ms_per_unet_forward": round(dt / S * 1e3, 3)}
            del sp
            torch.cuda.empty_cache()
        out["int8_weight_operand_MB"] = round(wbytes / 1e6, 1)
        out["weight_stream_floor_ms_at_5TBps"] = round(wbytes / 5e12 * 1e3, 3)
        return out

    info = dict(batch=batch, sync=sampler.stream.synchronize, finite=lambda: bool(torch.isfinite(sampler.x).all().item()),
                stream=sampler.stream, step=eng.step, sampler=sampler, inputs=(x_T, cond, uncond),
                new_sampler=lambda: GraphLatentDdimSampler(eng, S, batch, LAT, CTX, scale=scale, alphas_cumprod=alphas_cumprod_linear()), plms=plms, sweep=sweep, parity=parity, gelu_exact=gelu_exact, materialised=materialised if CTX is not None and sampler.pair_prefix else None,
                oracle_state=dict(sd=sd, wq=wq, act_names=act_names, cfg=cfg, eng=eng),     # scratch/sd_parity_full.py
                workload=(f"{P['name']} ({n_params:.1f}M) w4a8 on MI355X: {LH}x{LW}x{LC} latents, DDIM-{S} eta=0, "
                          + (f"CFG {scale} (UNet batch 2x{batch}), {CTX[0]}x{CTX[1]} context, " if CTX is not None else "unconditional, ")
                          + f"{batch} images per GPU ({P['tail']})"),
                preset=preset,
                extra={"batch_per_gpu": batch, "ddim_steps": S, "unet_evals_per_step": S, "unet_batch": NB, "guidance_scale": scale,
                       "guidance_pair": ("shared prefix computed once per pair (conv_in ... first self attention; bit-identical to the materialised 2B batch)"
                                         if (CTX is not None and sampler.pair_prefix) else ("materialised 2B batch" if CTX is not None else None))})
    return run, fwd, cpu, info


def calibration_sample(dev):
    """Second half of BASELINE.json's metric ("... + calibration wall-clock"): milliseconds per AdaRound iteration of
    SD-v1-size reconstruction units (the reference's single-GPU SD setting: mini-batch 8), measured on a bounded sample
    (median of 5 timed iterations per unit, synthetic cached inputs), and the wall-clock those rates imply for the block
    reconstructions of the whole UNet at the recipe's 20 000 iterations per unit (16 ResBlocks + 16 transformer blocks
    with at least these sizes' cost classes; layer units, TIB and activation calibration are minutes and left out)."""
    import tfmq_dm_amd.ops as ops
    from tfmq_dm_amd.engine import recon as R
    gen = torch.Generator().manual_seed(0)

    def ada(cout, cin, k=1, bias=True):
        shape = (cout, cin, k, k) if k > 1 else (cout, cin)
        w = (torch.randn(*shape, generator=gen) * 0.05).to(dev)
        qp = ops.minmax_to_qparam(ops.minmax(w.reshape(cout, -1).contiguous(), cout), 16)
        return R.AdaLayer(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), torch.zeros(cout, device=dev) if bias else None)

    def timeit(unit, bs, iters=5):
        idx = torch.arange(bs, device=dev)
        unit.iterate(idx)
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(iters):          # median: the first iterations after a unit is built still grow allocator pools
            t0 = time.perf_counter()
            unit.iterate(idx)
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]
    res = {}
    # (channels, resolution, #ResBlocks, #transformer blocks) of SD v1: input + output path per level; middle at 8x8
    for Cc, HW, n_res, n_tb in ((320, 64, 5, 5), (640, 32, 5, 5), (1280, 16, 5, 5), (1280, 8, 7, 1)):
        N = 8
        x = torch.randn(N, HW, HW, Cc, device=dev)
        y = torch.randn(N, HW, HW, Cc, device=dev)
        gn = (torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev))
        ru = R.ResnetUnit(ada(Cc, Cc, 3), ada(Cc, Cc, 3), gn, gn, None, x, torch.randn(N, Cc, device=dev), y, eps=1e-5, iters=100)
        ms_r = timeit(ru, N)
        layers = [ada(Cc, Cc, 1, False), ada(Cc, Cc, 1, False), ada(Cc, Cc, 1, False), ada(Cc, Cc), ada(8 * Cc, Cc), ada(Cc, 4 * Cc),
                  ada(Cc, Cc, 1, False), ada(Cc, 768, 1, False), ada(Cc, 768, 1, False), ada(Cc, Cc)]
        tu = R.TransformerUnit(layers, [gn, gn, gn], 8, x.reshape(N, HW * HW, Cc), torch.randn(N, 77, 768, device=dev),
                               y.reshape(N, HW * HW, Cc), iters=100)
        ms_t = timeit(tu, N)
        res[f"resblock_{Cc}ch_{HW}x{HW}_ms_per_iter"] = round(ms_r, 2)
        res[f"transformer_{Cc}ch_{HW}x{HW}_ms_per_iter"] = round(ms_t, 2)
        del ru, tu, layers, x, y
        torch.cuda.empty_cache()
    res["recipe"] = ("AdaRound block reconstruction, mini-batch 8, exact-fp32 MFMA GEMMs (K15); the per-unit rates above are timed live; the "
                     "recipe's 20000 iterations per unit were RUN, not projected: see measured_sd_recipe_20000_iterations")
    return res


def link_comm_ready():
    import tfmq_dm_amd.linklink as link
    return link.comm_device() is not None


def calibration_sharded(dev, world):
    """The exchange step of the sharded calibration on the real interconnect (BASELINE configs[4]): one SD-size
    ResBlock unit and one transformer unit (320 ch @ 64x64) iterated with mini-batch 8 PER RANK -- every rank owns its
    shard's cached inputs -- and ONE SUM all-reduce of the flattened dL/dW_hat buffer per Adam iteration through the C
    ABI's RCCL wrapper (tfmq_allreduce_sum_f32 on the unit's stream, between the backward GEMMs and the fused
    AdaRound-backward + Adam kernel).  Collective: every rank calls it.  world 1: the same units without the exchange."""
    import tfmq_dm_amd.linklink as link
    import tfmq_dm_amd.ops as ops
    from tfmq_dm_amd.engine import recon as R
    gen = torch.Generator().manual_seed(3)          # identical weights on every rank (replicas); shard data differs by rank

    def ada(cout, cin, k=1, bias=True):
        shape = (cout, cin, k, k) if k > 1 else (cout, cin)
        w = (torch.randn(*shape, generator=gen) * 0.05).to(dev)
        qp = ops.minmax_to_qparam(ops.minmax(w.reshape(cout, -1).contiguous(), cout), 16)
        return R.AdaLayer(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), torch.zeros(cout, device=dev) if bias else None)
    Cc, HW, N = 320, 64, 8
    kw = dict(iters=100, world_size=world, allreduce=link.allreduce if world > 1 else None)
    c1, c2 = ada(Cc, Cc, 3), ada(Cc, Cc, 3)
    layers = [ada(Cc, Cc, 1, False), ada(Cc, Cc, 1, False), ada(Cc, Cc, 1, False), ada(Cc, Cc), ada(8 * Cc, Cc), ada(Cc, 4 * Cc),
              ada(Cc, Cc, 1, False), ada(Cc, 768, 1, False), ada(Cc, 768, 1, False), ada(Cc, Cc)]
    x = torch.randn(N, HW, HW, Cc, device=dev)      # device RNG: each rank draws its own shard
    y = torch.randn(N, HW, HW, Cc, device=dev)
    gn = (torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev))
    ru = R.ResnetUnit(c1, c2, gn, gn, None, x, torch.randn(N, Cc, device=dev), y, eps=1e-5, **kw)
    tu = R.TransformerUnit(layers, [gn, gn, gn], 8, x.reshape(N, HW * HW, Cc), torch.randn(N, 77, 768, device=dev),
                           y.reshape(N, HW * HW, Cc), **kw)
    idx = torch.arange(N, device=dev)
    rccl_ranks = None
    if world > 1 and link_comm_ready():
        import ctypes as C
        from tfmq_dm_amd._lib import handle
        r_, w_ = C.c_int(-1), C.c_int(-1)
        handle(link.comm_device()).call("comm_info", C.byref(r_), C.byref(w_))
        rccl_ranks = int(w_.value)               # ranks the C ABI's RCCL communicator actually spans
    res = {"world": world, "rccl_ranks": rccl_ranks, "mini_batch_per_rank": N,
           "collective": (None if world == 1 else
                          "RCCL ncclAllReduce(SUM, fp32) via the C ABI (tfmq_allreduce_sum_f32), one per iteration" if link_comm_ready()
                          else "torch.distributed all_reduce (the C ABI communicator could not be created), one per iteration")}
    for name, unit in (("resblock_320ch_64x64", ru), ("transformer_320ch_64x64", tu)):
        for _ in range(2):
            unit.iterate(idx)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(8):
            unit.iterate(idx)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / 8 * 1e3
        nbytes = 4 * sum(l.alpha.numel() for l in unit.layers)
        ent = {"ms_per_iter": round(ms, 3), "allreduce_bytes": nbytes}
        if world > 1:
            buf = torch.empty(nbytes // 4, device=dev).normal_()
            for _ in range(3):
                link.allreduce(buf)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(10):
                link.allreduce(buf)
            torch.cuda.synchronize(dev)
            us = (time.perf_counter() - t0) / 10 * 1e6
            ent["allreduce_us"] = round(us, 1)
            ent["allreduce_algbw_GBps"] = round(nbytes / us / 1e3, 2)
            ent["allreduce_busbw_GBps"] = round(nbytes / us / 1e3 * 2 * (world - 1) / world, 2)
            # share of an iteration the exchange would take if nothing overlapped it (the pieces after the first overlap the
            # previous piece's Adam kernels: engine/recon.py, TFMQ_EXCHANGE_CHUNKS)
            ent["exchange_share_of_iteration"] = round(us / (ms * 1e3), 4)
            ent["exchange_pieces"] = len(getattr(unit, "_cuts", [0]))
        res[name] = ent
    del ru, tu, layers, x, y
    torch.cuda.empty_cache()
    return res


def run_cali_workload(args, dev, rank, local_rank, world, log):
    """`--workload cali`: the calibration half of the metric as ONE measured job -- `cali_model` (N = 1) or `cali_model_multi`
    (N > 1: per-timestep-group shards, one RCCL SUM all-reduce per AdaRound iteration through the C ABI, all-averaged
    activation deltas) end to end on the full SD v1 UNet (859.5 M, random init): weight-scale search, TIAR, every block and
    single-layer reconstruction unit, Finite-Set activation calibration, checkpoint.  The recipe is the README's with the
    calibration set and the iteration count cut so that the run takes minutes (`--cali-iters`, `--cali-samples`,
    `--cali-groups`); the line says what was run and scales nothing."""
    import collections, tempfile
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))      # the drop-in module names (`quant`, `linklink`), as the reference's scripts import them
    from tfmq_dm_amd.ldm.unet import UNetModel, SD_V1_UNET
    from quant.quant_layer import QMODE, Scaler
    import quant.calibration as QC, quant.reconstruction as QR
    from quant.reconstruction_util import RLOSS
    N, G, ITERS = args.cali_samples, args.cali_groups, args.cali_iters
    if N // world < 16:
        raise SystemExit(f"--workload cali: {N} samples per group over {world} rank(s) -- the activation calibration draws 16 per group and rank "
                         "without replacement (quant/calibration.py, as the reference does)")
    torch.manual_seed(1234)
    m = UNetModel(**SD_V1_UNET)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for p in m.parameters():
            if p.numel() and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    m = m.to(dev)
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MINMAX}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": True}
    path = os.path.join(tempfile.mkdtemp(), "sd_w4a8.pth")
    acc, calls = collections.defaultdict(float), collections.Counter()
    qnn, gen_note = None, "calibration set: synthetic normal latents / contexts at fixed timesteps"
    if getattr(args, "cali_generate", False):
        # The set as the reference's driver makes it (txt2img.py:421-487 -> quant/data_generate.py:13-49, generate_cali_text_guided_data): for every
        # c-th of the T = 50 sampler steps and every prompt, CFG-7.5 sampling with the FULL-PRECISION model from fresh noise until that step;
        # (x_t, t, c) and (x_t, t, uc) both enter.  The text encoder is outside this package: a fixed table of random embeddings stands in.
        if world > 1:
            raise SystemExit("--cali-generate: single-GPU job (the reference generates the set before it spawns the workers)")
        from tfmq_dm_amd.ldm.ddpm import LatentDiffusion
        from tfmq_dm_amd.ldm.ddim import DDIMSampler
        from quant.quant_model import QuantModel
        from quant.data_generate import generate_cali_text_guided_data
        T_, c_ = 50, 50 // G
        if 50 // c_ != G or N % 2:
            raise SystemExit(f"--cali-generate: {G} groups do not divide the 50 sampler steps evenly / odd group size {N}")
        qnn = QuantModel(m, wq, aq, cali=True, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
        qnn.set_quant_state(False, False)
        ld = LatentDiffusion(qnn, conditioning_key="crossattn").to(dev)
        table = {}
        ld.get_learned_conditioning = lambda prompts: torch.stack([table.setdefault(p_, torch.randn(77, 768, generator=g)) for p_ in prompts]).to(dev)
        torch.cuda.synchronize()
        tg = time.perf_counter()
        xs, ts, cs = generate_cali_text_guided_data(ld, DDIMSampler(ld), T_, c_, 1, tuple(f"prompt {i}" for i in range(N // 2)), [4, 64, 64])
        torch.cuda.synchronize()
        acc["generate_cali_text_guided_data"] = time.perf_counter() - tg
        xs, ts, cs = xs.float().cpu(), ts.float().cpu(), cs.float().cpu()
        assert xs.shape[0] == G * N, (xs.shape, G, N)
        gen_note = (f"calibration set GENERATED inside the timed run (generate_cali_text_guided_data: DDIM-50, CFG 7.5, full-precision UNet, {N // 2} prompts "
                    f"x every {c_}th step; random text embeddings stand in for the text encoder)")
        del ld
    else:
        xs = torch.randn(G * N, 4, 64, 64, generator=g)
        ts = torch.cat([torch.full((N,), float(t)) for t in np.linspace(981, 1, G).astype(int)])
        cs = torch.randn(G * N, 77, 768, generator=g)

    def timed(mod, name):
        f = getattr(mod, name)

        def g_(*a, **k):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = f(*a, **k)
            torch.cuda.synchronize()
            acc[name] += time.perf_counter() - t
            calls[name] += 1
            if rank == 0 and ITERS >= 1000:      # a long job keeps its per-unit times even when it is cut off
                print("[cali] %s #%d: %.1f s" % (name, calls[name], time.perf_counter() - t), file=sys.stderr, flush=True)
            return r
        setattr(mod, name, g_)
    for mod, name in ((QC, "tib_reconstruction"), (QC, "block_reconstruction"), (QC, "layer_reconstruction"), (QC, "_calibrate_activations")):
        timed(mod, name)
    kw = dict(iters=ITERS, batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE)
    if args.cali_only:
        QC.ONLY_UNITS = tuple(p for p in args.cali_only.split(",") if p)
    # roofline of the calibration half: HIP events around a bounded SAMPLE of the tfmq_gemm_f32 launches of the job (every
    # n-th launch, at most 2048 -- reconstruction forwards / backwards and the capture passes' exact-fp32 GEMMs alike), on their launch stream
    import tfmq_dm_amd.ops as ops
    gemm_rec = []
    ops.set_gemm_profile(gemm_rec, every=max(1, (74 if not args.cali_only else 8) * ITERS * 20 // 2048), cap=2048)
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    if world > 1:
        kw.update(wq_params=wq, aq_params=aq, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value], multi_gpu=True)
        if ONE_DEVICE:       # dry run on one GPU: every rank on cuda:0 over gloo (cali_model_multi's torch.cuda.set_device(gpu) -> device 0)
            real_set = torch.cuda.set_device
            torch.cuda.set_device = lambda d: real_set(0)
        QC.cali_model_multi(local_rank if not ONE_DEVICE else rank, "gloo" if ONE_DEVICE else "nccl", world, "env://", 0, world, m, True, path,
                            (xs, ts, cs), (xs, ts, cs), N, True, kw)
    else:
        from quant.quant_model import QuantModel
        if qnn is None:
            qnn = QuantModel(m, wq, aq, cali=True, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
        md = QC.cali_model(qnn, (xs, ts, cs), (xs, ts, cs), use_aq=True, path=path, running_stat=True, interval=N, multi_gpu=False, **kw)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0 + acc.get("generate_cali_text_guided_data", 0.0)
    QC.ONLY_UNITS = None
    ops.set_gemm_profile(None)
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if ONE_DEVICE else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        return None
    # (the measured numbers go to stderr first: a long job must not lose them to a formatting slip further down)
    print("[cali] wall-clock %.2f s; phases %s; calls %s" % (dt, json.dumps({k: round(v, 2) for k, v in acc.items()}), json.dumps(dict(calls))), file=sys.stderr, flush=True)
    ck = torch.load(path, map_location="cpu")
    n_units = calls["tib_reconstruction"] + calls["block_reconstruction"] + calls["layer_reconstruction"]
    rec_s = acc["tib_reconstruction"] + acc["block_reconstruction"] + acc["layer_reconstruction"]
    finite = all(bool(torch.isfinite(v).all()) for v in ck["weight"].values() if torch.is_tensor(v) and v.is_floating_point())
    mode = os.environ.get("TFMQ_RECON_GEMM", "bf16x3")
    # 2.5 PFLOP/s dense bf16 / f16 MFMA (MI355X_MICROARCH.md); bf16x3 spends three MFMAs per fp32 product; exact fp32 products run at the vector rate
    peak = {"bf16x3": 2500.0 / 3.0, "f16": 2500.0, "f32": 157.3}.get(mode, 2500.0 / 3.0)
    big = [(ops.event_elapsed_ms(e0, e1, dev.index or 0), fl, sh) for (e0, e1, fl, sh) in gemm_rec]
    roof = None
    if big:
        tot_ms, tot_fl = sum(b[0] for b in big), sum(b[1] for b in big)
        lg = [b for b in big if b[1] >= 2e9]           # the conv / Linear products of the units (per-head attention products are far smaller)
        lg_ms, lg_fl = sum(b[0] for b in lg), sum(b[1] for b in lg)
        ach = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        roof = {"bound": "mfma", "kernel": ("tfmq_gemm_f32 (csrc/gemm_f32_mfma.hip: k_gemm_bx3 -- bf16x3 operands split once per block into LDS, 128 x 128 tiles -- for "
                                            "the reconstruction iterations; k_gemm_f32_mfma for exact-fp32 products and skinny shapes)"),
                "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "peak_note": ("2.5 PFLOP/s dense bf16 MFMA / 3 MFMAs per fp32 product (hi hi' + hi lo' + lo hi')" if mode == "bf16x3" else f"operand mode {mode}"),
                "launches_timed": len(big), "sampled": "every n-th tfmq_gemm_f32 launch of the whole job (capture passes included), HIP events on the launch stream",
                "avg_launch_us": round(tot_ms * 1e3 / len(big), 2),
                "launches_of_2_GFLOP_or_more": {"n": len(lg), "achieved": round(lg_fl / (lg_ms * 1e-3) / 1e12, 1) if lg_ms > 0 else None,
                                                "frac": round(lg_fl / (lg_ms * 1e-3) / 1e12 / peak, 4) if lg_ms > 0 else None,
                                                "share_of_sampled_gemm_time": round(lg_ms / tot_ms, 3) if tot_ms > 0 else None}}
    return {
        "metric": "w4a8 calibration wall-clock, SD-v1-4 UNet (reduced recipe, see config)", "value": round(dt, 2), "unit": "s",
        "roofline": roof,
        "n_gpus": world, "steps": 1, "warmup": 0, "ms_per_step": round(dt * 1e3, 1), "higher_is_better": False,
        "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
        "dtype": {"f32": "f32 (AdaRound iterations: exact fp32 MFMA GEMMs)", "f16": "f32 values, fp16-operand MFMA GEMMs with fp32 accumulation (AdaRound iterations)"}.get(
            os.environ.get("TFMQ_RECON_GEMM", "bf16x3"), "f32 values, split-bf16 (hi + lo, 3 MFMAs per product, fp32 accumulation: 2^-16 per product) GEMMs (AdaRound iterations)")
        + " + int8/f16 (capture forwards)", "data": "synthetic",
        "config": {"workload": (f"cali_model{'_multi' if world > 1 else ''} on the SD v1-4 UNet (859.5M, random init): {G} timestep groups x {N} "
                                f"samples, {ITERS} AdaRound iterations per unit at mini-batch 8/rank (the recipe, txt2img.py:421-429,486: 50 DDIM steps x 256 samples -- 128 prompts x (cond, uncond) --, 20000), "
                                "w4 channel-wise + a8 Finite-Set, running_stat; " + gen_note
                                + (f"; reconstruction restricted to the units under {args.cali_only}" if args.cali_only else "")),
                   "parallelism": "single GPU" if world == 1 else f"timestep-group shards x{world}, one RCCL SUM all-reduce per iteration"},
        "finite": finite,
        "calibration": {"measured": True, "wall_clock_s": round(dt, 2), "reconstruction_units": n_units,
                        "iterations_per_unit": ITERS, "adaround_iterations_per_s": round(n_units * ITERS / max(rec_s, 1e-9), 1),
                        "phases_s": {**({"generate_cali_text_guided_data": round(acc["generate_cali_text_guided_data"], 2)} if "generate_cali_text_guided_data" in acc else {}),
                                     "tib_reconstruction": round(acc["tib_reconstruction"], 2),
                                     "block_reconstruction (incl. input/target capture)": round(acc["block_reconstruction"], 2),
                                     "layer_reconstruction (incl. capture)": round(acc["layer_reconstruction"], 2),
                                     "finite_set_activation_calibration": round(acc["_calibrate_activations"], 2)},
                        "adaround_tensors": sum(1 for k in ck["weight"] if k.endswith("alpha")),
                        "act_groups": len([k for k in ck if k.startswith("act_")]),
                        "checkpoint_MB": round(os.path.getsize(path) / 1e6, 1)},
    }


def sd_first_stage_state(gen):
    """Random-init state dict of the SD v1 KL-f8 first stage's decode side (ch 128, mult 1-2-4-4, 2 res blocks)."""
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=gen) * (1.0 / (ci * k * k) ** 0.5)
        sd[name + ".bias"] = torch.randn(co, generator=gen) * 0.02

    def norm(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=gen)
        sd[name + ".bias"] = 0.05 * torch.randn(c, generator=gen)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".nin_shortcut", co, ci, 1)

    ch, mult = 128, (1, 2, 4, 4)
    conv("post_quant_conv", 4, 4, 1)
    bi = ch * mult[-1]
    conv("decoder.conv_in", bi, 4, 3)
    res("decoder.mid.block_1", bi, bi)
    norm("decoder.mid.attn_1.norm", bi)
    for nm in ("q", "k", "v", "proj_out"):
        conv("decoder.mid.attn_1." + nm, bi, bi, 1)
    res("decoder.mid.block_2", bi, bi)
    for i in reversed(range(4)):
        bo = ch * mult[i]
        for j in range(3):
            res(f"decoder.up.{i}.block.{j}", bi, bo)
            bi = bo
        if i != 0:
            conv(f"decoder.up.{i}.upsample.conv", bi, bi, 3)
    norm("decoder.norm_out", bi)
    conv("decoder.conv_out", 3, bi, 3)
    return sd, dict(ch=ch, ch_mult=mult, num_res_blocks=2, resolution=256, attn_resolutions=[])


def first_stage_sample(dev, n_img=4):
    """Outside the metric's timed region (sample_diffusion_ldm.py:127-150 times the sampler loop only) and reported
    separately: decode of 64x64x4 latents to 512x512x3 images on the HIP first-stage decoder."""
    from tfmq_dm_amd.engine.vae_decoder import VaeDecoderEngine
    gen = torch.Generator().manual_seed(11)
    sd, cfg = sd_first_stage_state(gen)
    eng = VaeDecoderEngine(sd, cfg, dev)
    z = (torch.randn(n_img, 64, 64, 4, generator=gen) * 0.9).to(dev)
    eng.forward(z, scale_factor=0.18215)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(2):
        y = eng.forward(z, scale_factor=0.18215)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / 2 / n_img * 1e3
    res = {"ms_per_image": round(ms, 2), "images_per_s": round(1e3 / ms, 1), "tflops": round(2.5145 / ms * 1e3, 1), "batch": n_img,
           "finite": bool(torch.isfinite(y).all()),
           "config": "SD v1 KL-f8 decoder (49.5 M params, 2.51 TFLOP/image), fp16-operand MFMA convs, exact-fp32 512-channel mid attention"}
    del eng, y, z
    torch.cuda.empty_cache()
    return res


def conv_roofline(fwd, stream, n_fwd=2):
    """Per-launch HIP-event timing (on the launch stream) of every w4a8 GEMM launch of UNet forwards."""
    import tfmq_dm_amd.ops as ops
    rec = []
    ops.set_conv_profile(rec)
    try:
        for _ in range(n_fwd):
            fwd()
        stream.synchronize()
    finally:
        ops.set_conv_profile(None)
    tot_ops, tot_ms, tot_bytes, n = 0.0, 0.0, 0.0, 0
    fam = {}
    for (e0, e1, nops, kind, nbytes, family) in rec:
        ms = ops.event_elapsed_ms(e0, e1)
        # per launch: the floor each roof sets (dense int8 / fp16 MFMA peak; the 6.3 TB/s a float4 copy reaches on this chip,
        # MI355X_MICROARCH.md) -- the larger one is the launch's binding roof
        mfma_ms = nops / ((INT8_PEAK_TOPS if kind == "w4a8" else F16_PEAK_TFLOPS) * 1e12) * 1e3
        hbm_ms = nbytes / (HBM_ACHIEVABLE_TBPS * 1e12) * 1e3
        f = fam.setdefault(family, {"launches": 0, "ms": 0.0, "mfma_floor_ms": 0.0, "hbm_floor_ms": 0.0, "binding_floor_ms": 0.0, "hbm_bound_launches": 0})
        f["launches"] += 1
        f["ms"] += ms
        f["mfma_floor_ms"] += mfma_ms
        f["hbm_floor_ms"] += hbm_ms
        f["binding_floor_ms"] += max(mfma_ms, hbm_ms)
        f["hbm_bound_launches"] += int(hbm_ms > mfma_ms)
        if kind != "w4a8":
            continue
        tot_ops += nops
        tot_bytes += nbytes
        tot_ms += ms
        n += 1
    global _FAMILY_TABLE
    _FAMILY_TABLE = [{"family": k, "launches_per_forward": round(v["launches"] / n_fwd, 1), "hbm_bound_launches_per_forward": round(v["hbm_bound_launches"] / n_fwd, 1),
                      "ms_per_forward": round(v["ms"] / n_fwd, 3), "mfma_floor_ms": round(v["mfma_floor_ms"] / n_fwd, 3),
                      "hbm_floor_ms": round(v["hbm_floor_ms"] / n_fwd, 3),
                      "frac_of_binding_roof": round(v["binding_floor_ms"] / v["ms"], 4) if v["ms"] > 0 else None}
                     for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])]
    return tot_ops, tot_ms, n, tot_bytes, n_fwd


_FAMILY_TABLE = None


def _partial_line(args, info, world, dt, finite):
    """The sampling half of the JSON line (what rank 0 still prints if the calibration leg wedges at N > 1)."""
    cfgd = {"workload": info["workload"], "parallelism": f"replicas x{world} (no data-path collective)"}
    cfgd.update(info["extra"])
    return {"metric": "DDIM-50 images/sec, w4a8 SD-v1-4", "value": round(info["batch"] * world * args.steps / dt, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": cfgd, "finite": finite}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["sd", "cifar", "cali", "cin256", "celeba"], default="sd")
    ap.add_argument("--cali-iters", type=int, default=100, help="--workload cali: AdaRound iterations per unit (recipe: 20000)")
    ap.add_argument("--cali-samples", type=int, default=32, help="--workload cali: samples per timestep group (recipe: 256 = 128 prompts x {cond, uncond})")
    ap.add_argument("--cali-groups", type=int, default=2, help="--workload cali: timestep groups (recipe: 50, one per DDIM step)")
    ap.add_argument("--cali-generate", action="store_true", help="--workload cali: build the calibration set with generate_cali_text_guided_data (FP sampling) "
                    "inside the timed run instead of drawing synthetic latents")
    ap.add_argument("--cali-only", default="", help="--workload cali: comma-separated unit-name prefixes (e.g. model.middle_block,model.input_blocks.10); "
                    "only these reconstruction units run (at --cali-iters), the others keep nearest rounding -- for measuring a resolution level at the recipe's length")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default 64 for sd, 256 for cifar)")
    ap.add_argument("--ddim-steps", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cali-leg", action="store_true", help="skip the sharded-calibration exchange-step leg")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the TFMQ hot path has no CPU fallback")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become N ranks (one process per GPU) under torch.distributed.run,
        # exactly the command the driver uses
        if torch.cuda.device_count() < args.gpus and not ONE_DEVICE:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node has {torch.cuda.device_count()} GPU(s)")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    if ONE_DEVICE:                       # dry run of the N > 1 path on a 1-GPU box: every rank on cuda:0, gloo collectives
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if ONE_DEVICE else "nccl", rank=rank, world_size=world)
        import tfmq_dm_amd.linklink as link

    def init_c_abi_comm():
        """The C ABI's own RCCL communicator (calibration exchange step).  Sampling needs no collective, so for the sampling workloads it is
        created AFTER the timed region, under the calibration leg's watchdog: a communicator that cannot be built (or a bootstrap that never
        returns) costs that leg, never the sampling number."""
        try:
            if world > 1 and not ONE_DEVICE:
                link.init_comm(local_rank)
        except Exception as e:           # noqa: BLE001 -- keep the number, report the leg's error
            print(f"[bench] rank {rank}: tfmq_comm_init failed ({type(e).__name__}: {e}); the calibration leg falls back to torch.distributed",
                  file=sys.stderr, flush=True)

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    if args.workload == "cali":
        init_c_abi_comm()
        out = run_cali_workload(args, dev, rank, local_rank, world, log)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            link.destroy_comm()
            dist.destroy_process_group()
        return
    if args.workload == "cifar":
        run, fwd, cpu, info = setup_cifar(args, dev, rank, log)
    else:
        run, fwd, cpu, info = setup_sd(args, dev, rank, log, preset=args.workload)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    info["sync"]()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cpu" if ONE_DEVICE else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    finite = info["finite"]()

    # ---- second half of the metric: the sharded-calibration exchange step, every rank takes part.  A watchdog keeps a
    # wedged collective from costing the sampling number: after the deadline rank 0 prints the line without this leg.
    sharded = {"error": "skipped"}
    done = {"printed": False}

    def emit(out):
        if not done["printed"]:
            done["printed"] = True
            print(json.dumps(out), flush=True)
    if args.workload == "sd" and not args.no_cali_leg:
        import threading
        state = {"partial": None}

        def bail():
            if rank == 0 and state["partial"] is not None:
                state["partial"]["calibration"] = {"sharded": {"error": "calibration leg exceeded its deadline"}}
                # (exit status stays 0 so that the launcher keeps the sampling number; launchers / CI tell this run from a clean one by the field)
                state["partial"]["status"] = "calibration_leg_deadline_exceeded"
                emit(state["partial"])
            # (the watchdog only runs at world > 1, where rank 0 always holds the partial line: the other ranks leave quietly as well, so the
            # launcher reports the run whose line was printed as a success)
            os._exit(0 if world > 1 or (rank == 0 and state["partial"] is not None) else 3)
        wd = threading.Timer(420.0, bail)
        wd.daemon = True
        if world > 1:
            state["partial"] = dict(_partial_line(args, info, world, dt, finite)) if rank == 0 else None
            wd.start()
        try:
            init_c_abi_comm()
            sharded = calibration_sharded(dev, world)
        except Exception as e:        # noqa: BLE001 -- reported in the line, never fatal for the sampling number
            sharded = {"error": f"{type(e).__name__}: {e}"}
        wd.cancel()

    if rank == 0:
        images = info["batch"] * world * args.steps
        value = images / dt
        with torch.cuda.stream(info["stream"]):
            info["step"].zero_()
            tot_ops, tot_ms, n_launch, tot_bytes, n_fwd = conv_roofline(fwd, info["stream"])
        achieved = tot_ops / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        traffic, traffic_src = None, None
        tname = next((f"r{r:02d}_traffic_{args.workload}.json" for r in (9, 8, 7, 6, 5, 4, 3, 2, 1)
                      if os.path.exists(os.path.join(ROOT, "profiles", f"r{r:02d}_traffic_{args.workload}.json"))), None)
        if tname is not None:
            tj = json.load(open(os.path.join(ROOT, "profiles", tname)))["kernels"]
            ks = [v for k, v in tj.items() if ("k_conv_dma<false" in k or "k_conv_igemm<true" in k or "k_conv3_slab" in k
                                                or "k_lin_direct" in k or "k_ff_fused" in k or "k_row_chain" in k)]
            if ks:
                traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ks) / sum(v["launches"] for v in ks)
                traffic_src = (f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                               "separate passes, eager forwards of the same workload)")
        roof = {"bound": "mfma", "kernel": ("the w4a8 implicit-GEMM family: k_conv3_slab<WN> (3x3, activation slab staged once per channel chunk), "
                                            "k_lin_direct<mode> (pointwise, register-direct epilogue), k_conv_dma<false,...> (tile kernels), and the token-per-lane "
                                            "launches of round 4 -- k_ff_fused (attn2.to_out + norm3 + GEGLU feed-forward + proj_out) and k_row_chain (the Linears "
                                            "around the attentions), counted with the int8 ops of all their GEMMs"),
                "achieved": round(achieved, 2), "peak": INT8_PEAK_TOPS, "unit": "TOP/s",
                "frac": round(achieved / INT8_PEAK_TOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch (PMC)",
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(tot_bytes / max(n_launch, 1)),
                "hbm_achieved_TBps": round(tot_bytes / (tot_ms * 1e-3) / 1e12, 3) if tot_ms > 0 else None,
                "launches_timed": n_launch, "avg_launch_us": round(tot_ms * 1e3 / max(n_launch, 1), 2),
                "algorithmic_ops_per_forward": tot_ops / n_fwd,
                # which roof binds, per launch: floor = max(ops / MFMA peak, algorithmic bytes / 6.3 TB/s), summed per kernel family over the
                # same live HIP events; "bound" above names the roof of the family as a whole
                "families": _FAMILY_TABLE}
        cpu_b = None
        if world == 1 and not args.no_cpu_baseline:
            v, sample = cpu()
            cpu_b = {"value": round(v, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                     "sample": "oracle (torch-CPU fake-quant UNet, same weights / act tables): " + sample}
        cali = None
        if args.workload == "sd":
            cali = {"sharded": sharded}
        if world == 1 and args.workload == "sd" and not args.no_cpu_baseline:
            cali.update(calibration_sample(dev))
            # LIVE slice of the calibration job itself, timed inside this run: cali_model end to end on the SD UNet -- calibration-set generation
            # (FP DDIM sampling), weight-scale search, the reconstruction units under two prefixes (a ResBlock, a SpatialTransformer with its
            # BasicTransformerBlock, the middle attention) with input / target capture, Finite-Set activation calibration, checkpoint.
            # The full-length runs below are RECORDED (run once with `--workload cali`, committed under profiles/).
            try:
                if os.environ.get("TFMQ_BENCH_NO_LIVE_CALI") == "1":
                    raise RuntimeError("skipped (TFMQ_BENCH_NO_LIVE_CALI=1)")
                lj = run_cali_workload(argparse.Namespace(cali_samples=16, cali_groups=2, cali_iters=int(os.environ.get("TFMQ_BENCH_LIVE_CALI_ITERS", "500")),
                                                          cali_only="model.input_blocks.1,model.middle_block.1", cali_generate=True), dev, 0, 0, 1, log)
                cali["live_slice"] = {"live": True, "workload": lj["config"]["workload"], "finite": lj["finite"], **lj["calibration"]}
            except Exception as e:          # the slice must not take the sampling line down
                cali["live_slice"] = {"live": True, "error": repr(e)}
            torch.cuda.empty_cache()
            cali["first_stage_decode"] = first_stage_sample(dev)
            cali["plms"] = info["plms"]()
            # SD calibration at the recipe's iteration count, run once with this code by `bench.py --workload cali --cali-iters 20000`
            # (a run of this length cannot sit inside the default bench; the committed lines carry their own config and phase split)
            # (a gpurun call is limited to one hour, so the UNet was measured one resolution level per call; the levels partition the
            # reconstruction units, and `sum_of_levels` adds their wall-clocks)
            for key, fn in (("64x64_level_units", "r03_bench_line_cali_sd_64x64level_20000.json"), ("32x32_level_units", "r03_bench_line_cali_sd_32x32level_20000.json"),
                            ("16x16_level_units", "r03_bench_line_cali_sd_16x16level_20000.json"), ("8x8_level_units_and_tib", "r03_bench_line_cali_sd_8x8level_20000.json")):
                fpath = os.path.join(ROOT, "profiles", fn)
                if os.path.exists(fpath):
                    try:
                        mj = json.loads(open(fpath).read().strip().splitlines()[-1])
                        ent = dict(mj["calibration"])
                        ent["workload"], ent["source"], ent["recorded_run"] = mj["config"]["workload"], "profiles/" + fn, True
                        cali.setdefault("measured_sd_recipe_20000_iterations", {})[key] = ent
                    except Exception as e:      # a damaged record must not take the sampling line down
                        cali.setdefault("measured_sd_recipe_20000_iterations", {})[key] = {"error": repr(e)}
            # round 4: ONE job over all 74 units with the calibration set generated inside it (`--workload cali --cali-generate`)
            fname = next((n for n in ("r05_bench_line_cali_sd_full_20000.json", "r04_bench_line_cali_sd_full_20000.json")
                          if os.path.exists(os.path.join(ROOT, "profiles", n))), "r04_bench_line_cali_sd_full_20000.json")
            fpath = os.path.join(ROOT, "profiles", fname)
            if os.path.exists(fpath):
                try:
                    mj = json.loads(open(fpath).read().strip().splitlines()[-1])
                    cali["measured_sd_recipe_20000_iterations_one_job"] = {**mj["calibration"], "workload": mj["config"]["workload"],
                                                                           "source": "profiles/" + fname, "recorded_run": True}
                except Exception as e:
                    cali["measured_sd_recipe_20000_iterations_one_job"] = {"error": repr(e)}
            # round 6: the SD job ON THE RECIPE'S OWN SET (txt2img.py:421-429,486: 50 DDIM steps x 256 samples = 12 800, generated inside the job),
            # all 74 units x 20 000 iterations, measured in three jobs that partition the units (scratch/r06_sd_cali_full.sh); every job repeats set
            # generation and the Finite-Set pass, so ONE job = the reconstruction phases of all three + one generation + one Finite-Set pass
            try:
                parts = {}
                for part in "abc":
                    mj = json.loads(open(os.path.join(ROOT, "profiles", f"r06_sd_calibration_50x256_part_{part}.json")).read().strip().splitlines()[-1])
                    parts[part] = {"wall_clock_s": mj["calibration"]["wall_clock_s"], "reconstruction_units": mj["calibration"]["reconstruction_units"],
                                   "phases_s": mj["calibration"]["phases_s"], "source": f"profiles/r06_sd_calibration_50x256_part_{part}.json"}
                rec = sum(v for pt in parts.values() for k, v in pt["phases_s"].items() if "reconstruction" in k)
                gen = [pt["phases_s"]["generate_cali_text_guided_data"] for pt in parts.values()]
                fsc = [pt["phases_s"]["finite_set_activation_calibration"] for pt in parts.values()]
                cali["measured_sd_recipe_50x256_set_20000_iterations"] = {
                    "recorded_run": True, "reconstruction_units": sum(pt["reconstruction_units"] for pt in parts.values()), "iterations_per_unit": 20000,
                    "calibration_set": "50 DDIM steps x 256 samples (128 prompts x {cond, uncond}) = 12 800, FP sampling inside each job",
                    "sum_of_the_three_jobs_s": round(sum(pt["wall_clock_s"] for pt in parts.values()), 1),
                    "one_job_s": round(rec + sum(gen) / 3 + sum(fsc) / 3, 1),
                    "one_job_is": "sum of the reconstruction phases of the three jobs + one set generation + one Finite-Set pass (means of the three)",
                    "reconstruction_s": round(rec, 1), "parts": parts,
                    "note": "two cached tensors of part c (the 187.5 GiB input of output_blocks.9.0, the 125 GiB target of output_blocks.8.2.conv) were held "
                            "on the device as fp16 (TFMQ_CACHE_F16=1): the box's sandbox does not survive a pinned host allocation of that size"}
            except Exception as e:
                cali["measured_sd_recipe_50x256_set_20000_iterations"] = {"error": repr(e)}
            lv = cali.get("measured_sd_recipe_20000_iterations", {})
            if len(lv) == 4 and all("wall_clock_s" in v for v in lv.values()):
                lv["sum_of_levels"] = {"wall_clock_s": round(sum(v["wall_clock_s"] for v in lv.values()), 1),
                                       "reconstruction_units": sum(v["reconstruction_units"] for v in lv.values()),
                                       "note": "8 timestep groups x 128 samples, 20000 iterations per unit, 1 GPU; each level's run repeats the weight "
                                               "initialisation and the Finite-Set pass of the whole UNet"}
            mname = next((n for n in ("r05_cifar_calibration_full.json", "r04_cifar_calibration_full.json", "r03_cifar_calibration_full.json", "r02_cifar_calibration_full.json")
                          if os.path.exists(os.path.join(ROOT, "profiles", n))), "r02_cifar_calibration_full.json")
            mpath = os.path.join(ROOT, "profiles", mname)
            if os.path.exists(mpath):       # the whole CIFAR recipe, measured once end to end with this code (scratch/cifar_cali_full.py)
                mj = json.load(open(mpath))
                cali["measured_full_recipe_cifar"] = {k: mj[k] for k in ("recipe", "wall_clock_s", "phases_s", "reconstruction_units",
                                                                       "iterations_per_unit", "ms_per_iteration_all_units") if k in mj}
                cali["measured_full_recipe_cifar"]["source"] = "profiles/" + mname
            # round 5: the full recipes of the other two LDM drivers BASELINE.json names (configs[2] CelebA-HQ LDM-4, configs[4] cin256-v2), each
            # run once end to end through LatentRunner.quantize with this code (scratch/ldm_cali_full.py); recorded runs with their phase split
            for key, fn in (("measured_full_recipe_celeba_ldm4", "r05_celeba_calibration_full.json"), ("measured_full_recipe_cin256", "r05_cin256_calibration_full.json")):
                fpath = os.path.join(ROOT, "profiles", fn)
                if os.path.exists(fpath):
                    try:
                        mj = json.load(open(fpath))
                        cali[key] = {**{k: mj[k] for k in ("recipe", "wall_clock_s", "phases_s", "reconstruction_units", "iterations_per_unit",
                                                             "adaround_iterations_per_s", "parts") if k in mj}, "source": "profiles/" + fn, "recorded_run": True}
                    except Exception as e:
                        cali[key] = {"error": repr(e)}
        cfgd = {"workload": info["workload"], "parallelism": f"replicas x{world} (no data-path collective)"}
        cfgd.update(info["extra"])
        out = {
            "metric": {"sd": "DDIM-50 images/sec, w4a8 SD-v1-4", "cifar": "DDIM-100 images/sec, w4a8 CIFAR-10 DDPM",
                       "cin256": "DDIM-20 images/sec, w4a8 LDM ImageNet-256 (cin256-v2, CFG 3.0)",
                       "celeba": "DDIM-200 images/sec, w4a8 LDM-4 CelebA-HQ 256"}[args.workload],
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int8 (u8 activation bins x int4 weights, int32 accumulate; f16 MFMA for un-quantised layers / attention, fp16 activation stream with fp32 statistics / arithmetic)",
            "data": "synthetic: N(0,1) latents / context, random-init weights (zero params re-drawn N(0,0.02^2)), synthetic FSC tables",
            "config": cfgd, "finite": finite, "roofline": roof, "cpu_baseline": cpu_b, "calibration": cali,
            "status": "ok" if not (isinstance(sharded, dict) and "error" in sharded and sharded["error"] != "skipped") else "calibration_leg_error",
        }
        if cpu_b is not None and "parity" in info:
            try:
                out["parity"] = info["parity"]()
            except Exception as e:      # noqa: BLE001
                out["parity"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline and info.get("materialised") is not None:
            out["guidance_pair_materialised"] = info["materialised"]()
        if world == 1 and not args.no_cpu_baseline and info.get("gelu_exact") is not None:
            try:
                ge = info["gelu_exact"]()
                out["value_gelu_exact"] = ge["images_per_s"]
                out["gelu_exact"] = ge
            except Exception as e:      # noqa: BLE001 -- a side leg must not take the line down
                out["gelu_exact"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.workload == "sd" and not args.no_cpu_baseline and args.batch in (0, 64) and "sweep" in info:
            out["batch_sweep"] = info["sweep"]()
        emit(out)
    if dist is not None:
        dist.barrier()
        link.destroy_comm()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
