#!/bin/bash
# A/B of environment switches on the headline bf16 step: tools/ab_step.sh "" "SISR_X=1" ...
for V in "$@"; do
  echo "=== ${V:-default}"
  env $V timeout -k 10 300 python bench.py --precision bf16 --steps 40 --warmup 5 --no-cpu-baseline --configs none 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['summary']['ms_per_step'])"
done
