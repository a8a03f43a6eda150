#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
rm -f $O/run18.txt
for v in 0 1 0 1; do
  echo "-- TFMQ_ROW_CHAIN_640=$v" | tee -a $O/run18.txt
  TFMQ_ROW_CHAIN_640=$v timeout 600 python bench.py --no-cpu-baseline --no-cali-leg --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'])" | tee -a $O/run18.txt
done
