"""Properties of the generated gfx950 code that the measurements rely on (round 6; hipcc cross-compiles without a GPU):

  * no scratch (spill) instruction between a loop header and its back-branch in the shipped 8-wave slab kernel -- the first ping-pong build
    reloaded a hoisted pointer inside the K loop behind an s_waitcnt vmcnt(0), i.e. drained the LDS-DMA queue once per chunk;
  * the compute phase of the ping-pong loop is MFMAs only: between the barrier that closes a load phase and the next barrier there is no
    ds_read, no LDS-DMA and no VALU arithmetic (what `__builtin_amdgcn_sched_barrier(0)` and the pinned operands are there for).
"""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = next((c for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc") or "") if c and os.path.exists(c)), None)
KERNEL = "k_conv3_slabILi5ELb0ELi4ELi1E"          # k_conv3_slab<5, false, 4, 1>: 256 x 320 tiles, int8, ping-pong


@pytest.fixture(scope="module")
def slab_isa(tmp_path_factory):
    if HIPCC is None:
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("isa") / "conv_slab.s"
    sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
    import build as B          # the product's own flags
    src = os.path.join(ROOT, "tfmq-dm_amd", "csrc", "conv_slab.hip")
    flags = [f for f in B.FLAGS if f != "-fPIC"]
    r = subprocess.run([HIPCC] + flags + ["-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = out.read_text().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for k, (i, n) in enumerate(starts):
        if KERNEL in n:
            return lines[i:(starts[k + 1][0] if k + 1 < len(starts) else len(lines))]
    pytest.fail(f"{KERNEL} not in the listing")


def _loops(body):
    """The K loops: innermost backward-branch regions that hold a whole chunk's MFMAs (9 taps x 20).  (A backward branch to a join block that
    the code layout placed in front of the second loop copy spans that copy without being a loop: such regions contain another candidate.)"""
    labels = {m.group(1): q for q, l in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", l))}
    cand = {}
    for q, l in enumerate(body):
        t = l.strip()
        if t.startswith(("s_cbranch", "s_branch")):
            tgt = labels.get(t.split()[-1])
            if tgt is not None and tgt < q and sum("v_mfma" in x for x in body[tgt:q]) >= 180:
                cand[tgt] = min(q, cand.get(tgt, q))
    regions = sorted(cand.items())
    return [(a, b) for a, b in regions if not any((a2, b2) != (a, b) and a <= a2 and b2 <= b for a2, b2 in regions)]


def test_no_scratch_traffic_inside_the_k_loops(slab_isa):
    loops = _loops(slab_isa)
    assert loops, "no MFMA loop found"
    inside = [(q, slab_isa[q].strip()) for a, b in loops for q in range(a, b) if slab_isa[q].strip().startswith("scratch_")]
    assert not inside, inside[:5]


def test_compute_phase_is_mfma_only(slab_isa):
    a, b = max(_loops(slab_isa), key=lambda ab: ab[1] - ab[0])
    body = [l.strip() for l in slab_isa[a:b] if l.strip() and not l.strip().startswith((";", "."))]
    # phases = runs between s_barrier instructions; a compute phase is one that holds the 20 MFMAs of a step
    runs, cur = [], []
    for t in body:
        if t.startswith("s_barrier"):
            runs.append(cur)
            cur = []
        else:
            cur.append(t)
    compute = [r for r in runs if sum(x.startswith("v_mfma") for x in r) >= 20]
    assert len(compute) >= 8, len(compute)          # nine taps per chunk (the first one's phase starts before the loop header)
    for r in compute:
        other = [x for x in r if x.startswith(("ds_", "global_load", "buffer_load", "v_")) and not x.startswith("v_mfma")]
        assert not other, other[:5]
