#!/bin/bash
# round 5: one part of the full cin256 calibration recipe (a gpurun call is limited to one hour; the parts partition the reconstruction units,
# each repeats calibration-set generation, weight initialisation and the Finite-Set pass).  PART=a | b
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
if [ "$PART" = a ]; then ONLY="tib,model.input_blocks,model.middle_block"; else ONLY="model.output_blocks,model.out"; fi
FLOW=cin256 ONLY=$ONLY OUT=$O/r05_cin256_calibration_part_$PART.json timeout 3500 python scratch/ldm_cali_full.py 2>$O/cin256_$PART.err | tee $O/cin256_$PART.txt
tail -3 $O/cin256_$PART.err
