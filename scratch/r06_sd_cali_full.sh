#!/bin/bash
# round 6: the SD v1-4 w4a8 calibration on the RECIPE'S OWN SET (txt2img.py:421-429,486: 50 DDIM steps x 256 samples = 12 800, generated inside
# the job by FP sampling), every reconstruction unit at 20 000 iterations, in parts that partition the units (each part repeats set generation,
# weight initialisation and the Finite-Set pass).  PART=a | b | c
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
export TFMQ_CACHE_F16=1      # (a cache that fits neither the device in fp32 nor the host cap: fp16 on the device, logged with its inexact count)
case "$PART" in
  a) ONLY="tib,model.input_blocks";;
  b) ONLY="model.output_blocks.0.,model.output_blocks.1.,model.output_blocks.2.,model.output_blocks.3.,model.output_blocks.4.,model.output_blocks.5.,model.output_blocks.6.,model.output_blocks.7.";;
  c) ONLY="model.middle_block,model.output_blocks.8.,model.output_blocks.9.,model.output_blocks.10.,model.output_blocks.11.,model.out.";;
  *) echo "PART=a|b|c"; exit 2;;
esac
ITERS=${ITERS:-20000}
timeout ${LIMIT:-3300} python bench.py --workload cali --cali-generate --cali-groups 50 --cali-samples 256 --cali-iters $ITERS --cali-only "$ONLY" \
  > $O/sd_cali_50x256_part_$PART.json 2> $O/sd_cali_50x256_part_$PART.err
echo "rc=$?"; (free -g; grep "\[cali\]" $O/sd_cali_50x256_part_$PART.err | tail -80) > $O/sd_cali_50x256_part_$PART.units.txt; tail -c 600 $O/sd_cali_50x256_part_$PART.json; grep -i "pinned host\|error\|Traceback" $O/sd_cali_50x256_part_$PART.err | tail -5
