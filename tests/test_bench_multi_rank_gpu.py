"""The N > 1 flow of bench.py on a box with ONE GPU (TFMQ_BENCH_ONE_DEVICE=1: every rank on cuda:0, gloo collectives): the
self-spawn under torch.distributed.run, barrier + max-over-ranks timing, the sharded-calibration exchange leg with a real
all-reduce between two processes, exactly one JSON line from rank 0 with n_gpus following the world size.  (The RCCL
communicator itself is covered at world 1 by tests/test_calibration_multi_gpu.py; 2...8 real GPUs are the driver's run.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ, TFMQ_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"] + extra,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_cifar_sampling():
    j = _run(["--workload", "cifar", "--no-cpu-baseline", "--batch", "64"])
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["scaling"] == "weak" and j["finite"] and j["value"] > 0
    assert j["config"]["parallelism"].startswith("replicas x2")
    assert j["roofline"]["frac"] > 0 and j["cpu_baseline"] is None


def test_two_ranks_sd_with_sharded_calibration_leg():
    j = _run(["--workload", "sd", "--batch", "2", "--ddim-steps", "4", "--no-cpu-baseline"])
    assert j["n_gpus"] == 2 and j["finite"] and j["value"] > 0
    sh = j["calibration"]["sharded"]
    assert sh.get("world") == 2, sh
    assert sh["resblock_320ch_64x64"]["ms_per_iter"] > 0 and sh["transformer_320ch_64x64"]["allreduce_bytes"] > 0
    assert "torch.distributed" in sh["collective"] or "RCCL" in sh["collective"]


def test_two_ranks_calibration_workload():
    """`--workload cali` with two ranks: cali_model_multi on the full SD UNet -- shards by timestep group, one all-reduce per AdaRound
    iteration, all-averaged activation deltas, rank 0 writes the checkpoint -- at a few iterations per unit."""
    j = _run(["--workload", "cali", "--cali-iters", "3", "--cali-samples", "32", "--cali-groups", "2"])      # >= 16 samples per group and rank (calibration.py:97)
    assert j["n_gpus"] == 2 and j["finite"] and j["scaling"] == "strong" and j["higher_is_better"] is False
    c = j["calibration"]
    assert c["measured"] and c["reconstruction_units"] == 74 and c["adaround_tensors"] == 263 and c["act_groups"] == 2
