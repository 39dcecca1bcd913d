#!/bin/bash
# A/B of environment switches on the config iteration times: tools/ab_cfg.sh "cfg2,cfg4" "" "SISR_WGRAD_BATCH=0" ...
CFGS=$1; shift
for V in "$@"; do
  echo "=== ${V:-default}"
  env $V timeout -k 10 300 python bench.py --precision bf16 --steps 3 --warmup 1 --no-cpu-baseline --configs $CFGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['summary']['configs_ms_per_iteration'])"
done
