#!/bin/bash
# usage (on the GPU box): bash scratch/pmc_ff.sh [out-name] -> issue / busy / LDS counters of k_ff_fused on scratch/bench_ff.py's shape
R=$GRAFT_REPO_ROOT; OUT=${1:-pmc_ff}; mkdir -p $R/gpurun_out/r04; cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcf$i -- python $R/scratch/bench_ff.py > /tmp/pmcf$i.log 2>&1
  f=$(ls /tmp/pmcf$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep "k_ff_fused" $f | tail -40) > $R/gpurun_out/r04/${OUT}_$i.csv || tail -3 /tmp/pmcf$i.log
done
python - "$OUT" <<'PY'
import csv, glob, os, collections, sys, json
R = os.environ["GRAFT_REPO_ROOT"]
res = {}
for f in sorted(glob.glob(R + f"/gpurun_out/r04/{sys.argv[1]}_*.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k] = round(sum(v) / len(v))
print(json.dumps(res, indent=1))
json.dump(res, open(R + f"/gpurun_out/r04/{sys.argv[1]}.json", "w"), indent=1)
PY
