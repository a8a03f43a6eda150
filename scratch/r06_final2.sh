#!/bin/bash
# round 6, last check at HEAD: the -m gpu suite and the plain default bench line (with the recorded 50x256 calibration in it)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -a -E "passed|failed|error|FAILED|ERROR|^E  " | tail -15 > $O/r06_final_gpu_suite.txt
cat $O/r06_final_gpu_suite.txt
python bench.py --steps 2 --warmup 1 2> $O/final2_bench.err | grep '^{"metric"' | tail -1 > $O/r06_bench_line_sd_head.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06/r06_bench_line_sd_head.json').read().strip().splitlines()[-1])
print("VALUE", d["value"], "gelu_exact", d.get("value_gelu_exact"), "frac", d["roofline"]["frac"], "status", d.get("status"))
c = d["calibration"].get("measured_sd_recipe_50x256_set_20000_iterations", {})
print({k: c.get(k) for k in ("reconstruction_units", "one_job_s", "sum_of_the_three_jobs_s", "error")})
PY
