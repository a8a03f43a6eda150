#!/bin/bash
# usage (GPU box): bash scratch/pmc_calib.sh -> gpurun_out/$RR/pmc_calib.txt : FETCH_SIZE / WRITE_SIZE (KiB) per launch of scratch/pmc_calib.py's known-byte launches
RR=${RR:-r04}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$RR; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -- python $R/scratch/pmc_calib.py > /tmp/cal_$c.log 2>&1
done
python - <<'P' | tee $O/pmc_calib.txt
import csv, glob, collections, re
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/cal_{c}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").strip()[:70]
        agg[(n, r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][c] = sum(v[-3:]) / len(v[-3:])
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    if v.get("FETCH_SIZE", 0) * 1024 < 20e6 and v.get("WRITE_SIZE", 0) * 1024 < 20e6: continue
    print(f"{k[0]:70s} grid {k[1]:>10s}  FETCH_SIZE raw {v.get('FETCH_SIZE', 0) * 1024 / 1e6:9.1f} MB   WRITE_SIZE raw {v.get('WRITE_SIZE', 0) * 1024 / 1e6:9.1f} MB")
P
