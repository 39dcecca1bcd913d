#!/bin/bash
# round-4 evidence in one GPU call: full GPU suite with durations, PMC traffic of the config iterations, a kernel + memory-copy
# trace of cfg2.  Each step runs only if the previous one succeeded.
set -u
mkdir -p gpurun_out
C=${SISR_COMMIT:-?}
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r4_gpu_tests.log 2>&1 || { echo "gpu tests failed"; tail -30 gpurun_out/r4_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r4_gpu_tests.log
SISR_COMMIT=$C bash tools/pmc_cfg.sh gpurun_out/pmc_cfg cfg2 cfg3 cfg4 cfg5 > gpurun_out/pmc_cfg.log 2>&1 || { echo "pmc failed"; tail -5 gpurun_out/pmc_cfg.log; exit 1; }
tail -4 gpurun_out/pmc_cfg.log
ROOT=$PWD
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/mc && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/mc -- python3 $ROOT/bench.py --precision bf16 --steps 1 --warmup 0 --no-cpu-baseline --configs cfg2 --config-iters 10 > $ROOT/gpurun_out/r4_mc.log 2>&1 ) || { echo "memcopy trace failed"; tail -5 gpurun_out/r4_mc.log; exit 1; }
for f in $(find /tmp/mc -name "*memory_copy*.csv" -o -name "*kernel_stats.csv"); do cp $f gpurun_out/r4_mc_$(basename $f); done
ls gpurun_out | grep r4_mc
