#!/bin/bash
# quick A/B of library variants on the 64 x 4K step only: bash tools/ab_quick.sh variant [variant ...]
cd "$GRAFT_REPO_ROOT"
show='import sys,json; d=json.loads(sys.stdin.read()); print("   ", d["value"], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})'
for v in "$@"; do
  if [ "$v" != default ]; then export ADF_WLS_LIB=$GRAFT_REPO_ROOT/addingdisparityfiltering_amd/libadf_wls_$v.so; else unset ADF_WLS_LIB; fi
  echo "== variant $v"
  for rep in 1 2 3; do
    ADF_NO_OVERLAP=1 python bench.py --cpu-seconds 0 --matcher-pairs 0 --natural-pairs 0 --next-rows 0 --no-check --steps 10 2>/dev/null | python -c "$show"
  done
done
