#!/bin/bash
# developer builds of the library with parts of wgrad_deep.hip switched off (-DWD_DBG_NOK: no K loop, -DWD_DBG_NOXFORM: plain-copy
# prologues) next to the normal one, and the per-layer probe under each: where a tile's 3.7 us go.  Run from the repo root on the GPU box.
set -e
cd single-image-super-resolution_amd/csrc
mkdir -p dbg
for V in NOK NOXFORM; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -DWD_DBG_$V -c wgrad_deep.hip -o dbg/wgrad_deep_$V.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^wgrad_deep.o$') dbg/wgrad_deep_$V.o -o dbg/libsisr_$V.so
done
cd ../..
for L in "" single-image-super-resolution_amd/csrc/dbg/libsisr_NOK.so single-image-super-resolution_amd/csrc/dbg/libsisr_NOXFORM.so; do
  echo "=== ${L:-normal build}"
  if [ -n "$L" ]; then export SISR_LIB=$PWD/$L; else unset SISR_LIB; fi
  timeout -k 10 200 python tools/probe_deep.py all 2>/dev/null | grep "^D\|^G\|^---" | sed 's/fwd.*wgrad  /wgrad /' | cut -c1-150
done
