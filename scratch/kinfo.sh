#!/bin/bash
# kernel metadata (VGPR / AGPR / spills / LDS / scratch) of one .hip file, compiled device-only for gfx950 -- works without a GPU
# usage: scratch/kinfo.sh tfmq-dm_amd/csrc/conv_lin.hip [extra hipcc flags]
f=$1; shift
out=/tmp/kinfo_$(basename $f .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S "$@" -o $out $f || exit 1
python3 - $out <<'PY'
import re,sys
s=open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - \.a|\namdhsa\.target|\Z)", s, re.S):
    pass
# parse amdhsa.kernels metadata
import subprocess
blocks=s.split("- .agpr_count:")
for b in blocks[1:]:
    d=dict(re.findall(r"\.(\w+):\s+(\S+)", "agpr_count:"+b.split("\n    .args")[0] if False else ".agpr_count:"+b))
    name=d.get("name","?")
    try:
        name=subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt",name],capture_output=True,text=True).stdout.strip()[:110]
    except Exception: pass
    print(f"{name}\n    vgpr {d.get('vgpr_count')} agpr {d.get('agpr_count')} sgpr {d.get('sgpr_count')} vspill {d.get('vgpr_spill_count')} sspill {d.get('sgpr_spill_count')} lds {d.get('group_segment_fixed_size')} scratch {d.get('private_segment_fixed_size')}")
PY
