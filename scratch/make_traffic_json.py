"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<round>_traffic_<workload>.json.
Only dispatches after the last k_upsample2x marker kernel (scratch/pmc_forward.py) are kept.
usage: make_traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <note>"""
import csv, json, sys, collections, re


def load(path, counter):
    rows = sorted((r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "k_upsample2x" in r["Kernel_Name"]]
    if not marks:
        raise SystemExit("marker kernel not found in " + path)
    rows = rows[marks[-1] + 1:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).strip()[:60]
        a = agg[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"note": sys.argv[4], "kernels": {}}
for k in f:
    n = f[k][0]
    fk, wk = f[k][1] / n, (w[k][1] / w[k][0] if k in w and w[k][0] else 0.0)
    out["kernels"][k] = {"launches": n, "fetch_KiB_per_launch_raw": fk, "write_KiB_per_launch": wk,
                         "hbm_bytes_per_launch": (2 * fk + wk) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k:60s} n={v['launches']:5d} {v['hbm_bytes_per_launch']/1e6:9.2f} MB/launch")
