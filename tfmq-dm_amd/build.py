"""In-tree build of libtfmq_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python tfmq-dm_amd/build.py            # incremental
    python tfmq-dm_amd/build.py --force

Flags: no fast-math and -ffp-contract=off -- quantiser arithmetic must be true IEEE
division / round-half-even / un-fused multiply-add to reproduce the reference's bin
indices (SURVEY.md §7 hard part 2).
"""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# TFMQ_BUILD_DIR / TFMQ_LIB_OUT: a variant build (diagnostics flags, A/B arms) beside the product library, loaded with TFMQ_LIB_PATH
OBJ = os.environ.get("TFMQ_BUILD_DIR") or os.path.join(HERE, "build")
LIB = os.environ.get("TFMQ_LIB_OUT") or os.path.join(HERE, "libtfmq_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"] + os.environ.get("TFMQ_EXTRA_HIPCC_FLAGS", "").split()
# Per-file additions.  VGPR-form MFMA: the softmax works on the score accumulators with VALU instructions, and with
# the accumulators in AGPRs every key tile paid 64 v_accvgpr_read/write (a quarter of the kernel's VALU time).
FILE_FLAGS = {"attention_f16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(path, extra):
    h = hashlib.sha1()
    h.update(" ".join(FLAGS + FILE_FLAGS.get(os.path.basename(path), [])).encode())
    for p in [path] + extra:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")]
    hdrs.append(os.path.join(HERE, "..", "include", "tfmq_hip.h"))
    hipcc = _hipcc()
    jobs, objs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        stamp = obj + ".sha1"
        dig = _digest(src, hdrs)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((src, obj, stamp, dig))

    def compile_one(job):
        src, obj, stamp, dig = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(dig)
        return src

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for done in ex.map(compile_one, jobs):
                if verbose:
                    print("compiled", os.path.basename(done))
    if jobs or not os.path.exists(LIB) or force:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
