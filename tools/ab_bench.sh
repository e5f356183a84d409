#!/bin/bash
# A/B of library variants on the bench configurations (round 3).  usage: bash tools/ab_bench.sh variant [variant ...]
# ("default" = libadf_wls.so, anything else = libadf_wls_<name>.so built with build.build_variant)
cd "$GRAFT_REPO_ROOT"
show='import sys,json; d=json.loads(sys.stdin.read()); print("   ", d["value"], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})'
for v in "$@"; do
  if [ "$v" != default ]; then export ADF_WLS_LIB=$GRAFT_REPO_ROOT/addingdisparityfiltering_amd/libadf_wls_$v.so; else unset ADF_WLS_LIB; fi
  echo "== variant $v"
  for rep in 1 2; do
    echo "  cfg3 x64:";  python bench.py --cpu-seconds 0 --matcher-pairs 0 --natural-pairs 0 --next-rows 0 --no-check --steps 10 2>/dev/null | python -c "$show"
  done
  echo "  cfg5 x256:"; python bench.py --config 5 --pairs 256 --cpu-seconds 0 --matcher-pairs 0 --natural-pairs 0 --next-rows 0 --no-check --steps 20 2>/dev/null | python -c "$show"
  echo "  cfg2 x16:";  python bench.py --config 2 --pairs 16 --cpu-seconds 0 --matcher-pairs 0 --natural-pairs 0 --next-rows 0 --no-check --steps 20 2>/dev/null | python -c "$show"
  echo "  cfg1 x64:";  python bench.py --config 1 --pairs 64 --cpu-seconds 0 --matcher-pairs 0 --natural-pairs 0 --next-rows 0 --no-check --steps 20 2>/dev/null | python -c "$show"
done
