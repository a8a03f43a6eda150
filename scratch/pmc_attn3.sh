#!/bin/bash
# usage (on the GPU box): bash scratch/pmc_attn3.sh -> issue / wait / MFMA / VALU / LDS counters of the d = 40 attention kernel
# (scratch/bench_attn.py, BATCH=128, ONLY40=1); KPAT selects the kernel rows (default k_attention_d40)
RR=${RR:-r03}; export RR; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$RR; cd /tmp; export TMPDIR=/tmp; export BATCH=${BATCH:-128} ONLY40=1
KPAT=${KPAT:-k_attention_d40}; TAG=${TAG:-d40}
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmca$i -- python $R/scratch/bench_attn.py > /tmp/pmca$i.log 2>&1
  f=$(ls /tmp/pmca$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep "$KPAT" $f | tail -20) > $R/gpurun_out/$RR/pmc_attn_${TAG}_$i.csv || tail -3 /tmp/pmca$i.log
done
python - <<PY
import csv, glob, os, collections, json
R = os.environ["GRAFT_REPO_ROOT"]
out = {}
for f in sorted(glob.glob(R + "/gpurun_out/" + os.environ["RR"] + "/pmc_attn_${TAG}_*.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.update({k: round(sum(v) / len(v)) for k, v in agg.items()})
print(json.dumps(out, indent=1))
json.dump(out, open(R + "/gpurun_out/" + os.environ["RR"] + "/pmc_attn_${TAG}.json", "w"), indent=1)
PY
