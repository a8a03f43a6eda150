#!/bin/bash
# round 6, final state: smoke, the -m gpu suite, refreshed profiles (PMC traffic, rocprofv3 stats, the plain bench line), kernel shares of the timed sampling
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/final_smoke.txt 2>&1; tail -1 $O/final_smoke.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -a -E "passed|failed|error|FAILED|ERROR|^E  " | tail -15 > $O/r06_final_gpu_suite.txt
cat $O/r06_final_gpu_suite.txt
RR=r06 WLS=sd bash scratch/refresh_profiles.sh > $O/final_refresh.log 2>&1
tail -3 $O/final_refresh.log
RR=r06 bash scratch/r03_shares.sh > $O/final_shares.log 2>&1
head -12 $O/r06_trace_kernel_shares_sd.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06/final/r06_bench_line_sd.json').read().strip().splitlines()[-1])
print("VALUE", d["value"], "gelu_exact", d.get("value_gelu_exact"), "frac", d["roofline"]["frac"], "ms", d["ms_per_step"], "status", d.get("status"))
print("sweep", {k: v.get("images_per_s") for k, v in d["batch_sweep"].items() if isinstance(v, dict)})
for f in d["roofline"].get("families", []): print(f["family"], f["ms_per_forward"], f["frac_of_binding_roof"])
PY
