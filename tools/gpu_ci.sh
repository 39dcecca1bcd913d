#!/bin/bash
# Runs GPU test steps in sequence; a later step runs only if the previous one exited 0 or 1
# (tests passed / some assertions failed), never after a crash, signal or timeout.
set -u
mkdir -p gpurun_out
run_step() {
    local name=$1; shift
    echo "=== $name: $*" | tee -a gpurun_out/ci.log
    timeout -k 10 "${STEP_TIMEOUT:-600}" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "=== $name exit $rc" | tee -a gpurun_out/ci.log
    tail -n "${TAIL:-40}" "gpurun_out/$name.log"
    if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name (rc=$rc)"; exit $rc; fi
    return 0
}
: > gpurun_out/ci.log
for step in "$@"; do
    name=$(echo "$step" | cut -d: -f1)
    cmd=$(echo "$step" | cut -d: -f2-)
    run_step "$name" bash -c "$cmd"
done
