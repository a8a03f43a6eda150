#!/bin/bash
RR=${RR:-r03}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$RR; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SD_BATCH=1 SD_STEPS=4 TFMQ_TUNE_REPORT=2 python $R/scratch/sd_breakdown.py > $O/sd_breakdown_b1_ks.txt 2> $O/sd_breakdown_b1_ks.err
head -40 $O/sd_breakdown_b1_ks.txt; tail -1 $O/sd_breakdown_b1_ks.txt
BATCHES=1 bash $R/scratch/r03_small_batch.sh
