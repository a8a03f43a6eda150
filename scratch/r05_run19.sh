#!/bin/bash
# final state check: smoke, whole GPU suite, the plain default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/run19_smoke.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/run19_suite.txt
timeout 1200 python bench.py 2>$O/run19_bench.err | grep '^{"metric"' | tail -1 > $O/run19_bench_line_sd.json
python -c "import json; j=json.load(open('$O/run19_bench_line_sd.json')); print(j['value'], j.get('value_gelu_exact'), j['roofline']['frac'], j['parity']['eps_rel_l2_fast'], j['calibration']['live_slice'].get('wall_clock_s'))" | tee $O/run19_line.txt
