#!/bin/bash
# Round 3, VERDICT r2 item 7: kernel timeline of ONE pair per call (configs 2 and 5): per-kernel durations and the gaps
# between consecutive kernels, from rocprofv3's kernel trace.  profiles/summarize_latency.py prints the table.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in 2 5; do
  rm -rf gpurun_out/r03_lat_cfg$c
  python3 tools/latency_trace.py $c 500 > gpurun_out/r03_lat_cfg$c.txt 2> gpurun_out/r03_lat_cfg$c.err
  rocprofv3 --kernel-trace -d gpurun_out/r03_lat_cfg$c -o t --output-format csv -- python3 tools/latency_trace.py $c 300 >> gpurun_out/r03_lat_cfg$c.txt 2>> gpurun_out/r03_lat_cfg$c.err
done
python3 tools/graph_latency.py 2 > gpurun_out/r03_graph_latency.txt 2>&1 || true
python3 tools/graph_latency.py 5 >> gpurun_out/r03_graph_latency.txt 2>&1 || true
python3 profiles/summarize_latency.py > gpurun_out/r03_latency_timeline.txt
cat gpurun_out/r03_latency_timeline.txt; tail -12 gpurun_out/r03_graph_latency.txt
