#!/bin/bash
# round 6: PMC counters of the ping-pong forms against the forms they were measured against (separate --pmc passes, kernel trace only):
#   slab kernel 128x64x64 640->320 3x3: TFMQ_SLAB_PP=0 / 1;  d = 40 attention (UNet batch 128): TFMQ_ATTN_PP=0 / 1
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PASSES=("SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
        "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
        "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM")
run() {   # tag, kernel pattern, command...
  tag=$1; pat=$2; shift 2
  i=0
  for c in "${PASSES[@]}"; do
    i=$((i+1)); rm -rf /tmp/pp_$tag$i
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pp_$tag$i -- "$@" > /tmp/pp_$tag$i.log 2>&1
    f=$(ls /tmp/pp_$tag$i/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && (head -1 $f; grep "$pat" $f | tail -40) > $O/pmc_pp_${tag}_$i.csv || tail -3 /tmp/pp_$tag$i.log
  done
}
export SHAPE=128,64,64,640,320,3
TFMQ_SLAB_PP=0 run slab0 k_conv3_slab python $R/scratch/pmc_conv.py
TFMQ_SLAB_PP=1 run slab1 k_conv3_slab python $R/scratch/pmc_conv.py
export BATCH=128 ONLY40=1
TFMQ_ATTN_PP=0 run attn0 k_attention_d40 python $R/scratch/bench_attn.py
TFMQ_ATTN_PP=1 run attn1 k_attention_d40 python $R/scratch/bench_attn.py
python - <<'PY'
import csv, glob, os, collections, json
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r06"
out = {}
for tag in ("slab0", "slab1", "attn0", "attn1"):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(f"{O}/pmc_pp_{tag}_*.csv")):
        for r in csv.DictReader(open(f)):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: round(sum(v) / len(v)) for k, v in agg.items()}
    d = {}
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES sums over the 32 SEs x ...; per-SIMD busy fraction as in profiles/r04_pmc_attention_d40.json: MFMA busy cycles / (GRBM_GUI_ACTIVE x 1024 SIMDs)
        d["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 1024 / 8), 4) if c.get("GRBM_GUI_ACTIVE") else None
    if c.get("SQ_WAVE_CYCLES"):
        d["wave_cycles_waiting_frac (s_waitcnt / barrier)"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 4)
        d["wave_cycles_issue_stalled_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 4)
    if c.get("SQ_INSTS_MFMA"):
        d["valu_instructions_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"], 2)
    out[tag] = {"counters": c, "derived": d}
json.dump(out, open(O + "/r06_pmc_pingpong.json", "w"), indent=1)
for t, v in out.items(): print(t, v["derived"], {k: v["counters"].get(k) for k in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_VALU_MFMA_BUSY_CYCLES")})
PY
