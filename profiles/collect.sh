#!/bin/bash
# Collects the round's measurement artifacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash profiles/collect.sh wave'
# Outputs land in gpurun_out/ (scratch); profiles/summarize.py turns them into the committed summaries.
set -e
solver=${1:-wave}
round=${ADF_ROUND:-r04}
tag=gpurun_out/${round}_${solver}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf ${tag}_stats ${tag}_pmc_fetch ${tag}_pmc_write
# 1. the bench line itself (with the CPU baseline leg)
python3 bench.py --solver $solver --steps 10 --warmup 3 > ${tag}_bench_n1.json 2> ${tag}_bench_n1.err
# 2. per-kernel durations of the same command (shorter run, no CPU leg)
rocprofv3 --kernel-trace --stats -d ${tag}_stats -o stats --output-format csv -- python3 bench.py --solver $solver --steps 5 --warmup 2 --cpu-seconds 0 --no-check --matcher-pairs 0 --natural-pairs 0 --next-rows 0 > ${tag}_stats.log 2>&1
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (8 pairs, one step, kernel trace only)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${tag}_pmc_fetch -o pmc --output-format csv -- python3 bench.py --solver $solver --pairs 8 --steps 1 --warmup 0 --cpu-seconds 0 --no-check --matcher-pairs 0 --natural-pairs 0 --next-rows 0 > ${tag}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ${tag}_pmc_write -o pmc --output-format csv -- python3 bench.py --solver $solver --pairs 8 --steps 1 --warmup 0 --cpu-seconds 0 --no-check --matcher-pairs 0 --natural-pairs 0 --next-rows 0 > ${tag}_pmc_write.log 2>&1
# 4. the block matcher feeding the filter: its own kernel summary and timings
if [ "$solver" = wave ]; then
  rm -rf gpurun_out/${round}_matcher_stats gpurun_out/${round}_sgbm_stats
  rocprofv3 --kernel-trace --stats -d gpurun_out/${round}_matcher_stats -o stats --output-format csv -- python3 tools/bm_time.py 3840 2160 256 15 4 > gpurun_out/${round}_matcher_times.txt 2>&1
  python3 tools/bm_time.py 1920 1080 160 15 8 >> gpurun_out/${round}_matcher_times.txt 2>&1
  python3 tools/bm_time.py 1242 375 128 9 64 >> gpurun_out/${round}_matcher_times.txt 2>&1
  python3 tools/bm_time.py 1920 1080 160 15 1 >> gpurun_out/${round}_matcher_times.txt 2>&1
  python3 tools/bm_time.py 3840 2160 256 15 4 15 10 >> gpurun_out/${round}_matcher_times.txt 2>&1
  python3 tools/bm_cpu_time.py 1920 1080 160 15 >> gpurun_out/${round}_matcher_times.txt 2>&1
  # 5. the semi-global matcher (3-way): kernel summary and timings
  rocprofv3 --kernel-trace --stats -d gpurun_out/${round}_sgbm_stats -o stats --output-format csv -- python3 tools/sgbm_time.py 3840 2160 256 3 2 1 > gpurun_out/${round}_sgbm_times.txt 2>&1
  python3 tools/sgbm_time.py 1920 1080 160 3 4 1 >> gpurun_out/${round}_sgbm_times.txt 2>&1
  python3 tools/sgbm_time.py 1920 1080 160 3 4 3 >> gpurun_out/${round}_sgbm_times.txt 2>&1
  python3 tools/sgbm_time.py 1242 375 128 3 16 1 >> gpurun_out/${round}_sgbm_times.txt 2>&1
  python3 tools/sgbm_time.py 1920 1080 160 3 1 1 >> gpurun_out/${round}_sgbm_times.txt 2>&1
  python3 tools/sgbm_time.py 1920 1080 160 3 4 1 0 >> gpurun_out/${round}_sgbm_times.txt 2>&1
  python3 tools/sgbm_time.py 1920 1080 160 3 4 1 1 >> gpurun_out/${round}_sgbm_times.txt 2>&1
fi
find ${tag}_stats ${tag}_pmc_fetch ${tag}_pmc_write -name "*.csv" | sort
cat ${tag}_bench_n1.json
