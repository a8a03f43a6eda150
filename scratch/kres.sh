#!/bin/bash
# resource usage per kernel of one csrc file: scratch/kres.sh conv_slab.hip [extra flags]
f=$1; shift
cd $(dirname $0)/../tfmq-dm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include "$@" -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//; s/.*remark: *//' | \
 awk -F': ' '/Function Name/{n=$2} /^VGPRs:/{v=$2} /^TotalSGPRs/{sg=$2} /ScratchSize/{s=$2} /Occupancy/{o=$2} /SGPRs Spill/{ss=$2} /VGPRs Spill/{vs=$2} /LDS Size/{print n, "vgpr="v, "sgpr="sg, "scratch="s, "occ="o, "sspill="ss, "vspill="vs, "lds="$2}' | c++filt | cut -c1-220
