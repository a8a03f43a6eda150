#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_recon_graph_gpu.py tests/test_lossfunc_call_gpu.py -q 2>&1 | grep -v amdgpu.ids | tail -60 > $O/run14_tests.txt; tail -5 $O/run14_tests.txt
timeout 3300 python bench.py --workload cali --cali-generate --cali-groups 8 --cali-samples 128 --cali-iters 20000 2>$O/run14_cali_full.err | tee $O/r05_bench_line_cali_sd_full_20000.json | cut -c1-300
tail -3 $O/run14_cali_full.err
