#!/bin/bash
# CPU side: one library per ablation mask of the slab kernel's K loop into scratch/ab/:   bash scratch/r04_slab_abl.sh build "1 2 4 8 5 13"
# GPU side: time the 3x3 shapes with each:                                                 bash scratch/r04_slab_abl.sh run
cd $(dirname $0)/..
if [ "$1" = build ]; then
  mkdir -p scratch/ab
  for m in $2; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -DSLAB_ABLATE=$m -c tfmq-dm_amd/csrc/conv_slab.hip -o /tmp/slab_abl_$m.o || exit 1
    objs=$(ls tfmq-dm_amd/build/*.o | grep -v conv_slab.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/ab/libtfmq_slab_abl_$m.so $objs /tmp/slab_abl_$m.o -ldl || exit 1
    echo built mask $m
  done
else
  echo "== product"; STATS=0 TILES=5 SHAPES=${SHAPES:-0,3,6} timeout 200 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids
  for f in scratch/ab/libtfmq_slab_abl_*.so; do
    echo "== $f"; TFMQ_LIB_PATH=$PWD/$f STATS=0 TILES=5 SHAPES=${SHAPES:-0,3,6} timeout 200 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids
  done
fi
