#!/bin/bash
# round 5, CPU side: libraries whose hot kernels are compiled under another LLVM machine-scheduler strategy (-mllvm -amdgpu-sched-strategy=...)
cd $(dirname $0)/..
mkdir -p scratch/ab
for st in max-ilp max-memory-clause; do
  objs=""
  for f in conv_slab conv_lin ff_fused row_chain attention_f16 conv_igemm; do
    extra=""; [ $f = attention_f16 ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $extra -mllvm -amdgpu-sched-strategy=$st -c tfmq-dm_amd/csrc/$f.hip -o /tmp/sched_${st}_$f.o 2>/dev/null || exit 1
    objs="$objs /tmp/sched_${st}_$f.o"
  done
  rest=$(ls tfmq-dm_amd/build/*.o | grep -v -E "/(conv_slab|conv_lin|ff_fused|row_chain|attention_f16|conv_igemm)\.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/ab/libtfmq_sched_$st.so $rest $objs -ldl || exit 1
  echo built $st
done
