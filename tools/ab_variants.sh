#!/bin/bash
# A/B of library variants on the 64 x 4K step, overlapped and sequential: bash tools/ab_variants.sh variant [variant ...] [-- bench args]
cd "$GRAFT_REPO_ROOT"
vars=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vars+=("$1"); shift; done; [ "$1" = "--" ] && shift
show='import sys,json; d=json.loads(sys.stdin.read()); print("   ", d["value"], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})'
for rep in 1 2; do for v in "${vars[@]}"; do
  if [ "$v" != default ]; then export ADF_WLS_LIB=$GRAFT_REPO_ROOT/addingdisparityfiltering_amd/libadf_wls_$v.so; else unset ADF_WLS_LIB; fi
  for ov in 0 1; do
    echo "== variant $v ADF_NO_OVERLAP=$ov $*"
    ADF_NO_OVERLAP=$ov python bench.py --cpu-seconds 0 --matcher-pairs 0 --natural-pairs 0 --next-rows 0 --no-check --steps 10 "$@" 2>/dev/null | python -c "$show"
  done
done; done
