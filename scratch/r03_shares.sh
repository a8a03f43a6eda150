#!/bin/bash
# per-shape conv / linear table of one SD forward at the bench batch + kernel-family shares of the timed sampling (kernel trace; the last 22000 dispatches = 47 of its 50 step graphs: the warm-up sampling ends with the fp16-stream probe, which must stay outside the window)
RR=${RR:-r03}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$RR; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SD_BATCH=64 SD_STEPS=50 python $R/scratch/sd_breakdown.py > $O/sd_breakdown_b64.txt 2> $O/sd_breakdown_b64.err
rm -rf /tmp/tr_sd
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_sd -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cali-leg > $O/trace_bench.log 2> $O/trace_bench.err
t=$(ls /tmp/tr_sd/*/*kernel_trace.csv | head -1)
python $R/scratch/trace_gaps.py $t 22000 > $O/${RR}_trace_family_shares_sd.txt 2>&1
python - "$t" > $O/${RR}_trace_kernel_shares_sd.txt <<'P'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"])); rows = rows[-22000:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
fam = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").strip()
    fam[n][0] += 1; fam[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:60s} n={v[0]:6d} {v[1]/1e6:9.1f} ms {v[1]/busy*100:5.1f} %  avg {v[1]/v[0]/1e3:8.1f} us")
P
head -30 $O/${RR}_trace_kernel_shares_sd.txt; head -12 $O/${RR}_trace_family_shares_sd.txt
