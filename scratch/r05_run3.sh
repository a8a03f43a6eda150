#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
rm -f $O/gemm_abl.txt
timeout 600 python -m pytest tests/test_gemm_bx3_gpu.py -x -q 2>&1 | tail -3 | tee $O/run3_tests.txt
bash scratch/r05_gemm_abl.sh run
echo "== rocprofv3 of a calibration slice" | tee $O/run3_cali.txt
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cali -o cali -- python $R/bench.py --workload cali --cali-only model.input_blocks.1 --cali-iters 1000 --cali-samples 32 --cali-groups 2 2>$O/run3_prof.err | tee -a $O/run3_cali.txt
cd $R
find /tmp/prof_cali -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05_cali_sd_kernel_stats.csv
find /tmp/prof_cali -name "*agent_info.csv" | head -1 | xargs -I{} cp {} $O/r05_agent_info.csv
head -14 $O/r05_cali_sd_kernel_stats.csv | cut -c1-200
