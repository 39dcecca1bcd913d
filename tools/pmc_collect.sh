#!/bin/bash
# Collect PMC counters of the three trunk kernels (forward conv, data-gradient conv, weight gradient) with separate
# rocprofv3 --pmc passes (SQ: 8 slots per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass), then reduce them to
# per-launch averages with tools/pmc_reduce.py.  Usage (on the GPU box, from the repo root):
#   bash tools/pmc_collect.sh gpurun_out/pmc     ->  gpurun_out/pmc/<role>_<pass>/..._counter_collection.csv
set -u
OUT=${1:-gpurun_out/pmc}
ROOT=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp ITERS=15
PASSES=(
  "insts:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"
  "cycles:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES"
  "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
  "fetch:FETCH_SIZE"
  "write:WRITE_SIZE"
  "l2:TCC_HIT_sum TCC_MISS_sum"
)
for role in fwd dgrad wgrad; do
  for p in "${PASSES[@]}"; do
    name=${p%%:*}; ctrs=${p#*:}
    ( cd /tmp && ROLE=$role rocprofv3 --pmc $ctrs --output-format csv -d "$ROOT/$OUT/${role}_${name}" -o c -- python3 "$ROOT/tools/prof_conv.py" > "$ROOT/$OUT/${role}_${name}.log" 2>&1 ) || echo "pass $role/$name failed"
  done
done
python3 tools/pmc_reduce.py "$OUT"
