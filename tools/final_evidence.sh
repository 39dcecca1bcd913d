#!/bin/bash
# Round-end evidence in one GPU call: smoke, the default bench line, per-kernel stats + ordered kernel lists of the three builds,
# PMC counters of the fp32 trunk kernels.  Everything lands under gpurun_out/$1/.  (The -m gpu suite is run by its own call.)
# usage: bash tools/final_evidence.sh <name> <commit> [nobench]
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-final}
mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ "${3:-}" != nobench ]; then
    timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { echo "smoke failed"; tail -3 $out/smoke.log; exit 1; }
    tail -3 $out/smoke.log
    t0=$(date +%s); timeout -k 10 900 python bench.py > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -3 $out/bench.err; exit 1; }
    echo "bench wall $(( $(date +%s) - t0 )) s"
fi
bash tools/evidence_stats.sh ${1:-final} || exit 1
SISR_PRECISION=fp32 SISR_COMMIT=${2:-HEAD} timeout -k 10 600 bash tools/pmc_collect.sh gpurun_out/${1:-final}/pmc_fp32 > $out/pmc_fp32.log 2>&1
tail -3 $out/pmc_fp32.log
