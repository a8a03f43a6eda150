#!/bin/bash
# round 5, GPU run 5: fixture F27 in the background (host cores), the full CelebA-HQ LDM calibration recipe on the GPU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 3600 python tests/golden/gen_golden_sd_traj.py --threads 128 --out $O/f27_sd_traj.npz > $O/run5_f27.log 2>&1; echo "F27 exit $?" >> $O/run5_f27.log ) &
F27=$!
sleep 120
FLOW=celeba OUT=$O/r05_celeba_calibration_full.json timeout 4000 python scratch/ldm_cali_full.py 2>$O/run5_celeba.err | tee $O/run5_celeba.txt
wait $F27
tail -4 $O/run5_f27.log
