#!/bin/bash
# round 5 (GPU box): issue / busy / LDS / wait counters of k_gemm_bx3 on the SD conv-forward shape 32768 x 320 x 2880 (bf16x3 operands),
# separate --pmc passes (counters only with --kernel-trace).  -> gpurun_out/r05/r05_pmc_gemm_bx3.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  SHAPE=32768,320,2880 GEMM_PREC=bf16x3 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pgb$i -- python $R/scratch/pmc_gemm.py > /tmp/pgb$i.log 2>&1
  f=$(ls /tmp/pgb$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep "k_gemm_bx3" $f | tail -40) > $O/pmc_gemm_bx3_$i.csv || tail -3 /tmp/pgb$i.log
done
python - <<'PY'
import csv, glob, os, collections, json
R = os.environ["GRAFT_REPO_ROOT"]
res = {}
for f in sorted(glob.glob(R + "/gpurun_out/r05/pmc_gemm_bx3_*.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k] = round(sum(v) / len(v))
res["_note"] = ("k_gemm_bx3<1, 1> at M x N x K = 32768 x 320 x 2880 (bf16x3), per launch; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them "
                "(FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950: MI355X_MICROARCH.md)")
print(json.dumps(res, indent=1))
json.dump(res, open(R + "/gpurun_out/r05/r05_pmc_gemm_bx3.json", "w"), indent=1)
PY
