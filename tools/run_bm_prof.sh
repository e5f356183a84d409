set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/bmprof -o bm -- python3 tools/bm_time.py 3840 2160 256 15 4 > gpurun_out/bm_time_4k.txt 2>&1
python3 tools/bm_time.py 1920 1080 160 15 8 >> gpurun_out/bm_time_4k.txt 2>&1
python3 tools/bm_time.py 1242 375 128 9 64 >> gpurun_out/bm_time_4k.txt 2>&1
python3 tools/bm_cpu_time.py 1920 1080 160 15 >> gpurun_out/bm_time_4k.txt 2>&1
./tools/micro/valu_rates2 > gpurun_out/valu_rates2.txt 2>&1
find gpurun_out/bmprof -name "*kernel_stats*" | head
