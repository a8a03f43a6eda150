#!/bin/bash
# CPU side: build one library per ablation mask of k_ff_fused into scratch/ab/ (python tfmq-dm_amd/build.py with -DFF_ABLATE=<mask>):
#   bash scratch/r04_ff_abl.sh build "1 2 4 6 8 32 7"
# GPU side: time scratch/bench_ff.py with each:   bash scratch/r04_ff_abl.sh run
cd $(dirname $0)/..
if [ "$1" = build ]; then
  mkdir -p scratch/ab
  for m in $2; do
    rm -rf /tmp/abl_build && cp -r tfmq-dm_amd/build /tmp/abl_build_src 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DFF_ABLATE=$m -c tfmq-dm_amd/csrc/ff_fused.hip -o /tmp/ff_abl_$m.o || exit 1
    objs=$(ls tfmq-dm_amd/build/*.o | grep -v ff_fused.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/ab/libtfmq_ff_abl_$m.so $objs /tmp/ff_abl_$m.o -ldl || exit 1
    echo built mask $m
  done
else
  for f in scratch/ab/libtfmq_ff_abl_*.so; do
    echo "== $f"; TFMQ_LIB_PATH=$PWD/$f timeout 200 python scratch/bench_ff.py 2>&1 | grep -v amdgpu.ids
  done
fi
