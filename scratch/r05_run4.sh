#!/bin/bash
# round 5, GPU run 4: fixture F27 (the oracle's DDIM-50 SD trajectory, ~20 min of host time) in the background; meanwhile on the GPU: the whole
# -m gpu suite, and reduced trials of the CelebA / cin256 calibration recipes (rates for sizing the full runs).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
nproc | tee $O/run4_host.txt
( timeout 3000 python tests/golden/gen_golden_sd_traj.py --threads 96 --out $O/f27_sd_traj.npz > $O/run4_f27.log 2>&1; echo "F27 exit $?" >> $O/run4_f27.log ) &
F27=$!
sleep 90     # (its device set-up first: the box has one GPU, the suite below shares it)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/run4_pytest.txt
echo "== celeba trial" | tee $O/run4_ldm.txt
FLOW=celeba T=20 CALI_BATCH=64 ITERS=200 OUT=$O/celeba_trial.json timeout 900 python scratch/ldm_cali_full.py 2>$O/run4_celeba.err | tee -a $O/run4_ldm.txt
echo "== cin256 trial" | tee -a $O/run4_ldm.txt
FLOW=cin256 T=4 CALI_BATCH=2 ITERS=200 OUT=$O/cin256_trial.json timeout 900 python scratch/ldm_cali_full.py 2>$O/run4_cin256.err | tee -a $O/run4_ldm.txt
wait $F27
tail -5 $O/run4_f27.log
