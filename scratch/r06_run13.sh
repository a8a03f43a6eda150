#!/bin/bash
# round 6, call 13: same-box A/B of the sampling metric: round-5 forms (TFMQ_SLAB_PP=0 TFMQ_ATTN_LDS3=0) vs this round's defaults
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run13_step_ab.txt; : > $O
for arm in old new old new; do
  if [ $arm = old ]; then export TFMQ_SLAB_PP=0 TFMQ_ATTN_LDS3=0; else unset TFMQ_SLAB_PP TFMQ_ATTN_LDS3; fi
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cali-leg > /tmp/line.json 2> /tmp/line.err
  python - $arm >> $O <<'PY'
import json, sys
d = json.loads(open('/tmp/line.json').read().strip().splitlines()[-1])
print(sys.argv[1], "images/s", d["value"], "gelu_exact", d.get("value_gelu_exact"), "ms_per_step", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"])
PY
done
cat $O
