for V in "" "SISR_DEEP_TARGET=512" "SISR_DEEP_TARGET=512 SISR_DEEP_MINCPS=1" "SISR_DEEP_TARGET=1024 SISR_DEEP_MINCPS=1"; do
  echo "=== ${V:-default}"
  env $V timeout -k 10 200 python tools/probe_deep.py hr96 2>/dev/null | grep "^D\|^V\|^G" | awk '{printf "%s %s %s %s  fwd %s dgrad %s | %s\n",$1,$2,$3,$4,$6,$11,$0}' | cut -c1-60
done
