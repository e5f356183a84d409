#!/bin/bash
# Round 3, VERDICT r2 item 5: is the Infinity Cache worth scheduling for?  Per-kernel durations (rocprofv3 kernel trace)
# of the 4K filter call at 1, 2, 4, 8 and 64 pairs per call; profiles/summarize_mall.py turns them into a table.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 1 2 4 8 64; do
  rm -rf gpurun_out/r03_mall_p$n
  rocprofv3 --kernel-trace --stats -d gpurun_out/r03_mall_p$n -o stats --output-format csv -- python3 bench.py --pairs $n --steps 20 --warmup 3 --cpu-seconds 0 --no-check --matcher-pairs 0 > gpurun_out/r03_mall_p$n.json 2> gpurun_out/r03_mall_p$n.err
done
python3 profiles/summarize_mall.py > gpurun_out/r03_mall_kernels.txt
cat gpurun_out/r03_mall_kernels.txt
