#!/bin/bash
# round 3: SD calibration at the recipe's iteration count, one resolution level per gpurun call (a call is limited to 3600 s).
#   LEVEL=64|32|16|8  bash scratch/r03_cali_level.sh        -> gpurun_out/r03/bench_line_cali_sd_${LEVEL}level_20000.json
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r03; cd $R
case ${LEVEL:-64} in
  64) U="model.input_blocks.1.,model.input_blocks.2.,model.input_blocks.3.,model.output_blocks.9.,model.output_blocks.10.,model.output_blocks.11." ;;
  32) U="model.input_blocks.4.,model.input_blocks.5.,model.input_blocks.6.,model.output_blocks.6.,model.output_blocks.7.,model.output_blocks.8." ;;
  16) U="model.input_blocks.7.,model.input_blocks.8.,model.input_blocks.9.,model.output_blocks.3.,model.output_blocks.4.,model.output_blocks.5." ;;
  8)  U="model.input_blocks.10,model.input_blocks.11,model.middle_block,model.output_blocks.0.,model.output_blocks.1.,model.output_blocks.2.,tib" ;;
esac
timeout ${LIMIT:-3400} python bench.py --workload cali --cali-iters ${ITERS:-20000} --cali-samples 128 --cali-groups 8 --cali-only "$U" --no-cpu-baseline \
   2> gpurun_out/r03/cali_${LEVEL:-64}.err | grep '^{"metric"' | tail -1 > gpurun_out/r03/bench_line_cali_sd_${LEVEL:-64}level_${ITERS:-20000}.json
tail -3 gpurun_out/r03/cali_${LEVEL:-64}.err; head -c 1200 gpurun_out/r03/bench_line_cali_sd_${LEVEL:-64}level_${ITERS:-20000}.json
