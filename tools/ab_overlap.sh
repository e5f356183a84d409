#!/bin/bash
cd "$GRAFT_REPO_ROOT"
show='import sys,json; d=json.loads(sys.stdin.read()); print("   ", d["value"], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})'
for mode in 0 1 0 1; do
  echo "== ADF_NO_OVERLAP=$mode"
  ADF_NO_OVERLAP=$mode python bench.py --cpu-seconds 0 --matcher-pairs 0 --natural-pairs 0 --next-rows 0 --no-check --steps 10 "$@" 2>/dev/null | python -c "$show"
done
