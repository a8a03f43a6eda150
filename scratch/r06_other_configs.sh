#!/bin/bash
# round 6: plain bench lines of the other BASELINE configs at the last commit
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for wl in cifar cin256 celeba; do
  timeout 400 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-cali-leg 2> $O/other_$wl.err | grep '^{"metric"' | tail -1 > $O/r06_bench_line_$wl.json
  python - $wl <<'PY'
import json, sys
try:
    d = json.loads(open(f'gpurun_out/r06/r06_bench_line_{sys.argv[1]}.json').read().strip().splitlines()[-1])
    print(sys.argv[1], d["metric"], d["value"], d.get("value_gelu_exact"), d.get("status"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
