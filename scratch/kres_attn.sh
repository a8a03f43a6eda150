#!/bin/bash
cd $(dirname $0)/..; scratch/kres.sh attention_f16.hip -mllvm -amdgpu-mfma-vgpr-form=1 "$@"
