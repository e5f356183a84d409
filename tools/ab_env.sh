#!/bin/bash
# A/B of an environment knob on the 64 x 4K step: bash tools/ab_env.sh NAME value [value ...] [-- bench args]
cd "$GRAFT_REPO_ROOT"
name=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" = "--" ] && shift
show='import sys,json; d=json.loads(sys.stdin.read()); print("   ", d["value"], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})'
for rep in 1 2; do for v in "${vals[@]}"; do
  echo "== $name=$v"
  env $name=$v python bench.py --cpu-seconds 0 --matcher-pairs 0 --natural-pairs 0 --next-rows 0 --no-check --steps 10 "$@" 2>/dev/null | python -c "$show"
done; done
