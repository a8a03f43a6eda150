"""Fixture F27: a full DDIM-50 / CFG 7.5 trajectory of the Stable-Diffusion-size w4a8 UNet, device samplers against the CPU oracle
(tests/golden/gen_golden_sd_traj.py: oracle/tfmq_oracle.py -- the reference's fp32 fake-quant UNet + DDIMSampler arithmetic,
ldm/models/diffusion/ddim.py:118-212 -- run once for all 50 steps on bench.py's SD workload at 1 image).

BASELINE.json: "stated fp tolerance for sampled latents".  The tolerance is stated HERE, as a fraction of the yardstick
    Y = rel_l2(final latents of the w4a8 oracle, final latents of the un-quantised model on the same inputs)
-- the distance quantisation itself moves the sample.  A quantised forward is an avalanche (1e-6 noise on the quantizer
inputs moves the oracle's own eps by 2e-2), so 50 steps of any implementation that is not bit-identical in summation order end up a
quantisation-noise radius from the oracle; what is asserted is that every device mode stays well inside Y, and that the per-step
eps / x norms follow the oracle's."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "f27_sd_traj.npz")
DEV = torch.device("cuda", 0)

# stated tolerances (fractions of the yardstick Y; measured values are printed and recorded in DESIGN section 5)
FRAC_METRIC, FRAC_GELU_EXACT, FRAC_EXACT_FP = 0.75, 0.75, 0.75
ABS_BAR = 5e-2            # and in absolute terms: the bar the full-size eps comparisons use (tests/test_full_size_properties_gpu.py)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.fixture(scope="module")
def world():
    return _world(FIX)


def _world(FIX):
    if not os.path.exists(FIX):
        pytest.skip(f"{os.path.relpath(FIX, ROOT)} not generated yet (tests/golden/gen_golden_sd_traj.py, on the GPU box's host cores)")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import argparse
    import bench
    import gen_golden_sd_traj as G
    from tfmq_dm_amd.engine import LayerQ
    g = np.load(FIX)
    S = int(g["steps"])
    run, fwd, cpu, info = bench.setup_sd(argparse.Namespace(batch=1, ddim_steps=S, first_sampling=False), DEV, 0, lambda *a: None)
    st = info["oracle_state"]
    eng, sd, wq, act_names = st["eng"], st["sd"], st["wq"], st["act_names"]
    # the fixture's inputs and weights are the ones this process just rebuilt from their seeds
    np.testing.assert_allclose(G.weight_checksum({k: v.cpu() for k, v in sd.items()}), g["weight_checksum"], rtol=1e-12)
    seeds = [int(v) for v in np.atleast_1d(g["seed"])]
    trip = [G.inputs(sd_) for sd_ in seeds]
    x_T, cond, uncond = (torch.cat([t[i] for t in trip]) for i in range(3))
    np.testing.assert_allclose([float(x_T.double().sum()), float(cond.double().sum()), float(uncond.double().sum())], g["input_checksum"], rtol=1e-12)
    assert json.loads(str(g["act_names"])) == act_names
    # weight scales and Finite-Set table: the FIXTURE's (what the oracle ran with); how far a fresh device search is from them is printed
    names, sizes = json.loads(str(g["wq_names"])), g["wq_sizes"]
    assert names == sorted(wq)
    off, same = 0, 0
    wqf = {}
    for n, sz in zip(names, sizes):
        d = torch.from_numpy(g["wq_delta"][off:off + sz].copy()).to(DEV)
        z = torch.from_numpy(g["wq_zp"][off:off + sz].astype(np.float32)).to(DEV)
        same += int(torch.equal(d, wq[n].delta.reshape(-1)) and torch.equal(z, wq[n].zp.reshape(-1)))
        wqf[n] = LayerQ(d.reshape(wq[n].delta.shape), z.reshape(wq[n].zp.shape), None, wq[n].qid)
        off += sz
    qt = torch.from_numpy(g["qtable"]).to(DEV)
    print(f"\n[F27] weight scales of this box's search equal to the fixture's: {same} of {len(names)} layers; "
          f"Finite-Set table max rel. difference {float(((eng.qtable - qt).abs() / qt.abs().clamp_min(1e-12)).max()):.2e}")
    return dict(g=g, S=S, sd=sd, cfg=st["cfg"], wq=wqf, qt=qt, x_T=x_T, cond=cond, uncond=uncond, final=torch.from_numpy(g["final"]))


def _sample(w, env=None, quantised=True, record_steps=()):
    """final latents (NCHW, cpu) of a fresh engine + graph sampler under the given environment; optionally the latents entering some steps"""
    from tfmq_dm_amd.engine import LdmUNetEngine
    from tfmq_dm_amd.ldm.sampler import GraphLatentDdimSampler, alphas_cumprod_linear
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        eng = LdmUNetEngine(w["sd"], w["cfg"], DEV)
        step = torch.zeros(1, dtype=torch.int32, device=DEV)
        if quantised:
            eng.prepare(w["wq"], w["qt"], step)
        else:
            eng.prepare(None, None, step)
        sp = GraphLatentDdimSampler(eng, w["S"], 1, (4, 64, 64), (77, 768), scale=7.5, alphas_cumprod=alphas_cumprod_linear())
        xT = w["x_T"].permute(0, 2, 3, 1).contiguous().to(DEV)
        eager = bool(getattr(eng, "exact_fp", False))      # the exact-fp diagnostics engine reads its step counter on the host: no capture

        def sample(steps=None):
            if not eager:
                return sp.sample_nhwc(xT, w["cond"].to(DEV), w["uncond"].to(DEV), steps=steps)
            with torch.cuda.stream(sp.stream):             # the sampler's own step body (UNet pair, fused CFG + DDIM update, step advance), un-captured
                sp.x.copy_(xT)
                sp.ctx2[:1].copy_(w["uncond"].to(DEV))
                sp.ctx2[1:].copy_(w["cond"].to(DEV))
                sp.step.zero_()
                for _ in range(sp.coef.shape[0] if steps is None else steps):
                    sp._step_body()
            return sp.x
        if not eager:
            sp.capture()
        inter = {}
        for k in record_steps:
            x = sample(int(k))
            sp.stream.synchronize()
            inter[int(k)] = x.permute(0, 3, 1, 2).float().cpu().clone()
        x = sample()
        sp.stream.synchronize()
        out = x.permute(0, 3, 1, 2).float().cpu().clone()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        torch.cuda.empty_cache()
    return (out, inter) if record_steps else out


def test_final_latents_of_every_device_mode_within_the_stated_fraction_of_the_quantisation_yardstick(world):
    w = world
    ref = w["final"]
    assert torch.isfinite(ref).all()
    fp = _sample(w, env={"TFMQ_EXACT_FP": "1", "TFMQ_PAIR_PREFIX": "0"}, quantised=False)          # un-quantised model, exact-fp32 device path
    Y = rel(ref, fp)
    keep_at = [int(k) for k in w["g"]["keep_at"] if int(k) > 0]
    metric, inter = _sample(w, record_steps=keep_at)
    r_metric = rel(metric, ref)
    r_gelu = rel(_sample(w, env={"TFMQ_GELU_EXACT": "1"}), ref)
    r_exact = rel(_sample(w, env={"TFMQ_EXACT_FP": "1", "TFMQ_PAIR_PREFIX": "0"}), ref)
    print(f"\n[F27] DDIM-{w['S']} final latents, rel-L2 vs the CPU oracle: metric mode {r_metric:.4f}, TFMQ_GELU_EXACT=1 {r_gelu:.4f}, "
          f"TFMQ_EXACT_FP=1 {r_exact:.4f}; yardstick rel_l2(w4a8 oracle, un-quantised model) = {Y:.4f} "
          f"-> fractions {r_metric / Y:.3f} / {r_gelu / Y:.3f} / {r_exact / Y:.3f}")
    for k, kept in zip(w["g"]["keep_at"], w["g"]["keep"]):
        if int(k) in inter:
            print(f"[F27]   latent entering step {int(k) + 1}: rel-L2 vs the oracle {rel(inter[int(k)], torch.from_numpy(kept)):.4f}")
    assert Y > 0.02, "the quantised and the un-quantised trajectories should differ visibly (else the yardstick says nothing)"
    assert r_metric <= FRAC_METRIC * Y and r_gelu <= FRAC_GELU_EXACT * Y and r_exact <= FRAC_EXACT_FP * Y
    assert max(r_metric, r_gelu, r_exact) <= ABS_BAR
    # the drift grows along the trajectory, it does not jump: the first recorded latents are far closer than the last
    first = min(inter)
    assert rel(inter[first], torch.from_numpy(w["g"]["keep"][list(w["g"]["keep_at"]).index(first)])) <= 0.5 * max(r_metric, 1e-3) + 5e-3


def test_per_step_norms_follow_the_oracle(world):
    """|x_t| entering every step (the sampler's own arithmetic + the UNet) within 2 % of the oracle's along the whole trajectory."""
    w = world
    S = w["S"]
    xn = w["g"]["x_norm"]
    ks = sorted(set(range(1, S, max(1, S // 10))) | {S - 1})
    _, inter = _sample(w, record_steps=ks)
    dev = {k: float(inter[k].norm()) for k in ks}
    worst = max(abs(dev[k] - xn[k]) / xn[k] for k in ks)
    print(f"\n[F27] |x_t| entering steps {ks}: worst relative deviation from the oracle {worst:.4f}")
    assert worst <= 0.02


# ---- fixture F27b (round 6): the same trajectory for MORE images (other latents, other contexts), one oracle batch.  The
# verdict asked for four; the oracle's DDIM-50 run costs ~20 minutes of the GPU box's host cores per image and the round's lease held two
# (gen_golden_sd_traj.py --seeds 2026,2027): with F27 the stated tolerance rests on three images.
FIX4 = os.path.join(ROOT, "tests", "golden", "f27b_sd_traj_multi.npz")
FRAC_MULTI = 0.6          # tightened from 0.75: every image measured so far sits at 0.45 ... 0.55 of its own yardstick


def test_more_images_stay_within_the_tightened_fraction_of_their_yardsticks():
    w = _world(FIX4)
    n = w["x_T"].shape[0]
    assert n >= 2 and w["final"].shape[0] == n
    rows = []
    for i in range(n):
        wi = dict(w, x_T=w["x_T"][i:i + 1], cond=w["cond"][i:i + 1], uncond=w["uncond"][i:i + 1])
        ref = w["final"][i:i + 1]
        Y = rel(ref, _sample(wi, env={"TFMQ_EXACT_FP": "1", "TFMQ_PAIR_PREFIX": "0"}, quantised=False))
        r_metric = rel(_sample(wi), ref)
        r_gelu = rel(_sample(wi, env={"TFMQ_GELU_EXACT": "1"}), ref)
        rows.append((r_metric, r_gelu, Y))
        print(f"\n[F27b] image {i}: final latents rel-L2 vs the CPU oracle: metric mode {r_metric:.4f}, TFMQ_GELU_EXACT=1 {r_gelu:.4f}; "
              f"yardstick {Y:.4f} -> fractions {r_metric / Y:.3f} / {r_gelu / Y:.3f}")
    for r_metric, r_gelu, Y in rows:
        assert Y > 0.02
        assert r_metric <= FRAC_MULTI * Y and r_gelu <= FRAC_MULTI * Y and max(r_metric, r_gelu) <= ABS_BAR
