#!/bin/bash
# usage (on the GPU box): ONLY=0 bash scratch/pmc_lin.sh -> HBM / L2 counters of the direct pointwise kernel on one shape of scratch/bench_lin.py
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r02; cd /tmp; export TMPDIR=/tmp
export ONLY=${ONLY:-0} TILES=${TILES:-6}
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcl$i -- python $R/scratch/bench_lin.py > /tmp/pmcl$i.log 2>&1
  f=$(ls /tmp/pmcl$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep "k_lin_" $f | tail -12) > $R/gpurun_out/r02/pmc_lin_$i.csv || tail -5 /tmp/pmcl$i.log
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R + "/gpurun_out/r02/pmc_lin_*.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(f), {k: sum(v) / len(v) for k, v in agg.items()})
PY
