#!/bin/bash
# the three trajectories of tools/slab_drift.py and their comparison -> gpurun_out/r04_slab_drift.txt
set -e
N=${1:-50}
mkdir -p gpurun_out
timeout -k 10 300 python tools/slab_drift.py run fp32 fp32 $N > gpurun_out/drift_fp32.log 2>&1
SISR_SLAB_BF16=1 timeout -k 10 300 python tools/slab_drift.py run bf16_slabs_bf16 bf16 $N > gpurun_out/drift_b1.log 2>&1
SISR_SLAB_BF16=0 timeout -k 10 300 python tools/slab_drift.py run bf16_slabs_f32 bf16 $N > gpurun_out/drift_b0.log 2>&1
python tools/slab_drift.py compare fp32 bf16_slabs_bf16 bf16_slabs_f32 > gpurun_out/r04_slab_drift.txt
cat gpurun_out/r04_slab_drift.txt
