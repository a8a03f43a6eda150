#!/bin/bash
# round 5, final GPU run: smoke(), the whole -m gpu suite, the other BASELINE configs' bench lines, the CIFAR recipe, the driver's command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/run15_smoke.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/run15_suite.txt
for wl in cifar cin256 celeba; do
  timeout 900 python bench.py --workload $wl --steps 2 --warmup 1 2>$O/run15_$wl.err | grep '^{"metric"' | tail -1 > $O/r05_bench_line_$wl.json
  python -c "import json,sys; j=json.load(open('$O/r05_bench_line_$wl.json')); print('$wl', j['value'], j.get('value_gelu_exact'), j['roofline']['frac'])" | tee -a $O/run15_lines.txt
done
ITERS=20000 OUT=$O/r05_cifar_calibration_full.json timeout 1500 python scratch/cifar_cali_full.py 2>$O/run15_cifar_cali.err | cut -c1-400 | tee $O/run15_cifar_cali.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cali-leg 2>/dev/null | grep '^{"metric"' | tail -1 > $O/r05_bench_line_sd_steps20_warmup5.json
python -c "import json; j=json.load(open('$O/r05_bench_line_sd_steps20_warmup5.json')); print('sd 20/5', j['value'])" | tee -a $O/run15_lines.txt
