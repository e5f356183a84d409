set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmc_sq1 -o sq1 --output-format csv -- python3 bench.py --pairs 8 --steps 1 --warmup 0 --cpu-seconds 0 --no-check > gpurun_out/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_sq2 -o sq2 --output-format csv -- python3 bench.py --pairs 8 --steps 1 --warmup 0 --cpu-seconds 0 --no-check > gpurun_out/pmc_sq2.log 2>&1
find gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 -name "*.csv" | head
