#!/bin/bash
# round 5, GPU run 10: profiles of the round (PMC traffic, rocprofv3 stats, the plain default bench line, kernel shares, per-shape table), then
# the SD calibration job at the RECIPE's set size (50 steps x 256 samples generated inside the job) for three units at 20 000 iterations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
RR=r05 WLS=sd bash scratch/refresh_profiles.sh > $O/run10_refresh.log 2>&1
RR=r05 bash scratch/r03_shares.sh > $O/run10_shares.log 2>&1
cd $R
timeout 2400 python bench.py --workload cali --cali-generate --cali-groups 50 --cali-samples 256 --cali-iters 20000 \
  --cali-only model.input_blocks.4.0,model.input_blocks.4.1.transformer_blocks.0,model.middle_block.0 2>$O/run10_cali50x256.err | tee $O/r05_bench_line_cali_sd_50x256_three_units.json | cut -c1-400
tail -3 $O/run10_cali50x256.err
