"""Build oracle/_build/liboracle_c.so (gcc).  Test infrastructure; see tfmq_oracle_c.c."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "liboracle_c.so")


def build(force=False):
    src = os.path.join(HERE, "tfmq_oracle_c.c")
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(src):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", OUT, src, "-lm"], check=True)
    return OUT


if __name__ == "__main__":
    print(build(True))
