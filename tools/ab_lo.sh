#!/bin/bash
# A/B of library variants on the down-scaled path (64 pairs, 4K views, 1080p maps): bash tools/ab_lo.sh variant [variant ...]
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in "$@"; do
  if [ "$v" != default ]; then export ADF_WLS_LIB=$GRAFT_REPO_ROOT/addingdisparityfiltering_amd/libadf_wls_$v.so; else unset ADF_WLS_LIB; fi
  echo "== variant $v"
  python3 tools/scaled_time.py 3840 2160 64 2 2>&1 | grep -v amdgpu.ids | grep -A1 "radius 2"
done; done
