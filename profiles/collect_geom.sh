#!/bin/bash
# Round 3: the 64 x 4K step on the geometries the reference's own factories produce:
#   config   ROI (256,0,3584,2160), radius 2     (SGBM factory, the headline)
#   bm       ROI (263,7,3570,2146), radius 5     (StereoBM factory, block 15: DF.cpp:401-402)
#   generic  ROI (256,0,3584,2160), radius 5     (createDisparityWLSFilterGeneric's default radius, DF.cpp:155)
# gpurun --timeout 1100 -- 'bash profiles/collect_geom.sh'
set -e
round=${ADF_ROUND:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/${round}
python3 bench.py --steps 10 --warmup 3 > ${o}_wave_bench_n1.json 2> ${o}_wave_bench_n1.err
python3 bench.py --steps 10 --warmup 3 --roi bm --cpu-seconds 0 --matcher-pairs 0 > ${o}_wave_bench_bmroi_n1.json 2> ${o}_wave_bench_bmroi_n1.err
python3 bench.py --steps 10 --warmup 3 --radius 5 --cpu-seconds 0 --matcher-pairs 0 > ${o}_wave_bench_radius5_n1.json 2> ${o}_wave_bench_radius5_n1.err
rm -rf ${o}_bmroi_stats ${o}_wave_stats
rocprofv3 --kernel-trace --stats -d ${o}_bmroi_stats -o stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --roi bm --cpu-seconds 0 --no-check --matcher-pairs 0 --natural-pairs 0 --next-rows 0 > ${o}_bmroi_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d ${o}_wave_stats -o stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-check --matcher-pairs 0 --natural-pairs 0 --next-rows 0 > ${o}_wave_stats.log 2>&1
find ${o}_bmroi_stats ${o}_wave_stats -name "*kernel_stats.csv"
python3 - <<'PY'
import json
for n in ("wave_bench_n1", "wave_bench_bmroi_n1", "wave_bench_radius5_n1"):
    d = json.loads(open("gpurun_out/r03_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["path"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
