#!/bin/bash
# usage (on the GPU box): bash scratch/pmc_attn2.sh -> MFMA / VALU busy counters of the fp16 attention kernels (scratch/bench_attn.py, BATCH=128)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r02; cd /tmp; export TMPDIR=/tmp; export BATCH=${BATCH:-128}
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmca$i -- python $R/scratch/bench_attn.py > /tmp/pmca$i.log 2>&1
  f=$(ls /tmp/pmca$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep "k_attention_h<3, 2, 40" $f | tail -20) > $R/gpurun_out/r02/pmc_attn2_$i.csv || tail -3 /tmp/pmca$i.log
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R + "/gpurun_out/r02/pmc_attn2_*.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(f), {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
