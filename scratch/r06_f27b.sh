#!/bin/bash
# round 6: fixture F27b -- the oracle's DDIM-50 / CFG 7.5 trajectory of bench.py's SD workload for TWO more images in one batch (host cores of the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout ${LIMIT:-3000} python tests/golden/gen_golden_sd_traj.py --seeds 2026,2027 --threads 128 --out $O/f27b_sd_traj_multi.npz > $O/f27b_gen.log 2> $O/f27b_gen.err
echo "rc=$?"; tail -2 $O/f27b_gen.err; ls -la $O/f27b_sd_traj_multi.npz
if [ -f $O/f27b_sd_traj_multi.npz ]; then
  cp $O/f27b_sd_traj_multi.npz tests/golden/f27b_sd_traj_multi.npz
  timeout 600 python -m pytest tests/test_sd_trajectory_gpu.py -m gpu -q -s -p no:cacheprovider -k "more_images" 2>&1 | grep -a "F27b\|passed\|failed\|Error" > $O/f27b_test.txt
  cat $O/f27b_test.txt
fi
