#!/bin/bash
# usage (on the GPU box): ONLY=0 TILES=6 bash scratch/pmc_lin2.sh -> issue / busy counters of a pointwise kernel on one shape of scratch/bench_lin.py
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r02; cd /tmp; export TMPDIR=/tmp
export ONLY=${ONLY:-0} TILES=${TILES:-6}
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
         "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_MOPS_I8"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcm$i -- python $R/scratch/bench_lin.py > /tmp/pmcm$i.log 2>&1
  f=$(ls /tmp/pmcm$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep "k_lin_" $f | tail -24) > $R/gpurun_out/r02/pmc_lin2_$i.csv || tail -3 /tmp/pmcm$i.log
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R + "/gpurun_out/r02/pmc_lin2_*.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(f), {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
