#!/bin/bash
# round 5, GPU run 12: captured reconstruction iterations -- tests, then same-box A/B of the SD calibration job (TFMQ_RECON_GRAPH=0 / 1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_recon_graph_gpu.py tests/test_recon_units_gpu.py tests/test_calibration_gpu.py tests/test_recon_precision_gpu.py tests/test_configs_r02_gpu.py tests/test_fisher_gpu.py tests/test_delta_learning_gpu.py tests/test_calibration_multi_gpu.py -x -q 2>&1 | tail -12 | tee $O/run12_tests.txt
echo "== cali A/B (2 x 32 samples, 1000 iterations per unit, all 74 units)" | tee $O/run12_cali.txt
for g in 1 0; do
  echo "-- TFMQ_RECON_GRAPH=$g" | tee -a $O/run12_cali.txt
  TFMQ_RECON_GRAPH=$g timeout 1500 python bench.py --workload cali --cali-iters 1000 --cali-samples 32 --cali-groups 2 2>$O/run12_cali_g$g.err | tee -a $O/run12_cali.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['calibration']['phases_s'], j['calibration']['adaround_iterations_per_s'], j['roofline']['achieved'] if j.get('roofline') else None)"
  tail -2 $O/run12_cali_g$g.err
done
