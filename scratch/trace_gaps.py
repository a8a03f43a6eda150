"""Kernel-trace analysis of one bench run (rocprofv3 --kernel-trace --output-format csv): for the last N dispatches (the timed
samplings), the sum of kernel durations vs the wall span, the idle gaps between consecutive kernels, and time per kernel family.
usage: trace_gaps.py <kernel_trace.csv> [n_last]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
rows = rows[-n_last:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
pos = [g for g in gaps if g > 0]
print(f"dispatches {len(rows)}  span {span/1e6:.1f} ms  busy {busy/1e6:.1f} ms ({busy/span*100:.1f} %)  idle gaps {sum(pos)/1e6:.1f} ms, mean {sum(pos)/max(len(pos),1)/1e3:.2f} us, >50us: {sum(1 for g in pos if g > 50000)}")
fam = collections.defaultdict(lambda: [0, 0])
for r in rows:
    name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").strip()
    name = re.sub(r"<.*", "", name)
    fam[name][0] += 1
    fam[name][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{k:34s} n={v[0]:6d} {v[1]/1e6:9.1f} ms {v[1]/busy*100:5.1f} %  avg {v[1]/v[0]/1e3:8.1f} us")
