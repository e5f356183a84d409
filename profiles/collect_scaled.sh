#!/bin/bash
# Down-scaled path (the sample's default pipeline): 64 pairs, 4K views, 1080p maps, with the first row pass
# interpolating the maps itself (default) and through the two resize kernels (ADF_SCALED_FUSE=0), same box, plus the
# rocprofv3 per-kernel summary of both.   gpurun --timeout 900 -- 'bash profiles/collect_scaled.sh'
set -e
round=${ADF_ROUND:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${round}_scaled_path.txt
: > $out
# (fuse half): 1 1 = default (half-width form of the fused prologue: the maps are exactly half the view), 1 0 = its general
# form (ADF_LO_HALF=0), 0 0 = round 3's resize kernels
for v in "1 1" "1 0" "0 0" "1 1" "1 0" "0 0"; do
  set -- $v; fuse=$1; half=$2
  echo "# ADF_SCALED_FUSE=$fuse ADF_LO_HALF=$half" >> $out
  ADF_SCALED_FUSE=$fuse ADF_LO_HALF=$half python3 tools/scaled_time.py 3840 2160 64 2 >> $out 2>&1
done
for fuse in 1 0; do
  rm -rf gpurun_out/${round}_scaled_stats_$fuse
  ADF_SCALED_FUSE=$fuse rocprofv3 --kernel-trace --stats -d gpurun_out/${round}_scaled_stats_$fuse -o stats --output-format csv -- python3 tools/scaled_time.py 3840 2160 64 2 > gpurun_out/${round}_scaled_stats_$fuse.log 2>&1
  echo "# rocprofv3 --kernel-trace --stats, ADF_SCALED_FUSE=$fuse (kernel, calls, total ns, average ns)" >> $out
  f=$(find gpurun_out/${round}_scaled_stats_$fuse -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("   %-110s %5s %12s %10.0f" % (r["Name"][:110], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"])))
PY
done
cat $out
