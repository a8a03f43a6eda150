#!/bin/bash
# usage (on the GPU box): bash scratch/pmc_slab.sh -> gpurun_out/r02/pmc_slab_<pass>.csv : LDS / MFMA / VALU counters of the 3x3 slab kernel
# on one SD shape (SHAPES index of scratch/bench_slab.py, default 1 = 128x64x64 640->320)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r02; cd /tmp; export TMPDIR=/tmp
export SHAPES=${SHAPES:-1}
i=0
for c in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmcs$i -- python $R/scratch/bench_slab.py > /tmp/pmcs$i.log 2>&1
  f=$(ls /tmp/pmcs$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep "k_conv3_slab" $f | tail -12) > $R/gpurun_out/r02/pmc_slab_$i.csv || tail -5 /tmp/pmcs$i.log
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R + "/gpurun_out/r02/pmc_slab_*.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(f), {k: sum(v) / len(v) for k, v in agg.items()})
PY
