#!/bin/bash
# Per-kernel register / scratch / occupancy summary of one HIP source (cross-compiled, no GPU needed).
#   tools/isa_stats.sh addingdisparityfiltering_amd/csrc/fgs_wave_h.hip [extra hipcc flags]
src=$1; shift
out=/tmp/isa/$(basename "$src" .hip).s
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math --cuda-device-only -S "$src" -o "$out" "$@" 2>/dev/null || exit 1
awk '/^; Kernel info:/ {k=1} /\.amdhsa_kernel / {name=$2} /^; NumVgprs:/ {v=$3} /^; ScratchSize:/ {s=$3} /^; Occupancy:/ {o=$3; print v, s, o, name}' "$out" | while read v s o n; do echo "$v vgpr  scratch $s  occ $o  $(echo $n | c++filt | sed 's/adf::(anonymous namespace):://; s/(.*//')"; done
