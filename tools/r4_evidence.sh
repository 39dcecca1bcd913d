#!/bin/bash
# round-4 evidence in one GPU call (each step only if the previous one succeeded): the full GPU suite with durations, the default
# bench line, per-layer probe, kernel stats / per-grid totals / launch order of the config iterations, PMC traffic of the iterations.
set -u
mkdir -p gpurun_out
C=${SISR_COMMIT:-?}
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r4_gpu_tests.log 2>&1 || { echo "gpu tests failed"; tail -30 gpurun_out/r4_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r4_gpu_tests.log
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_default.jsonl 2> gpurun_out/r4_bench_default.err || { echo "bench failed"; tail -5 gpurun_out/r4_bench_default.err; exit 1; }
tail -c 700 gpurun_out/r4_bench_default.jsonl; echo
timeout -k 10 300 python tools/probe_deep.py all > gpurun_out/r4_probe.log 2>&1 || { echo "probe failed"; exit 1; }
bash tools/cfg_profile.sh r4e cfg2 cfg3 cfg4 cfg5 || exit 1
SISR_COMMIT=$C bash tools/pmc_cfg.sh gpurun_out/pmc_cfg cfg2 cfg3 cfg4 cfg5 > gpurun_out/pmc_cfg.log 2>&1 || { echo "pmc failed"; tail -5 gpurun_out/pmc_cfg.log; exit 1; }
tail -4 gpurun_out/pmc_cfg.log
