#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate --pmc passes) of full SRGAN iterations of BASELINE configs; see tools/pmc_cfg.py
# usage (GPU box, repo root):  bash tools/pmc_cfg.sh gpurun_out/pmc_cfg cfg2 cfg3 ...
set -u
OUT=$1; shift
ROOT=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
for C in "$@"; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout -k 10 400 rocprofv3 --pmc $CTR --output-format csv -d "$ROOT/$OUT/${C}_${CTR}" -o c -- python3 "$ROOT/tools/pmc_cfg.py" run $C > "$ROOT/$OUT/${C}_${CTR}.log" 2>&1 ) || { echo "pass $C/$CTR failed"; tail -3 "$ROOT/$OUT/${C}_${CTR}.log"; exit 1; }
    echo "$C $CTR done"
  done
done
python3 tools/pmc_cfg.py reduce "$OUT" "$@"
