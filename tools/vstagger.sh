#!/bin/bash
# Column-pass stagger experiment (round 3): builds happen on the CPU side (tools/vstagger_build.py), this runs them.
#   gpurun -- 'bash tools/vstagger.sh'
set -e
cd "$GRAFT_REPO_ROOT"
for v in "" vstag2 vstag4 vstag8 vstag4b; do
  if [ -n "$v" ]; then export ADF_WLS_LIB=$GRAFT_REPO_ROOT/addingdisparityfiltering_amd/libadf_wls_$v.so; else unset ADF_WLS_LIB; fi
  [ -z "$v" ] || [ -f "$ADF_WLS_LIB" ] || continue
  echo "== variant ${v:-default}"
  for rep in 1 2; do
  ADF_NO_OVERLAP=1 python bench.py --cpu-seconds 0 --matcher-pairs 0 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['checked'][0]['disparity_max_abs_lsb'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
  done
done
