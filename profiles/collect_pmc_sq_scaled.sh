#!/bin/bash
# Round 4: SQ counters of the down-scaled call's kernels (8 pairs, 4K views / 1080p maps, sequential: ADF_NO_OVERLAP=1),
# the fused low-resolution first row pass (half-width form wave_hpass_kernel<56,2,3,1>, general form <56,2,2,1>) beside round 3's form (ADF_SCALED_FUSE=0:
# resize kernels + wave_hpass_kernel<56,2,1,1>).  One rocprofv3 pass per counter group, as profiles/collect_pmc_sq.sh.
#   gpurun --timeout 900 -- 'bash profiles/collect_pmc_sq_scaled.sh'
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ADF_NO_OVERLAP=1
for fuse in 2 1 0; do       # 2: default (half-width form of the prologue), 1: its general form (ADF_LO_HALF=0), 0: the resize kernels
  if [ $fuse = 0 ]; then export ADF_SCALED_FUSE=0; else export ADF_SCALED_FUSE=1; fi
  if [ $fuse = 2 ]; then export ADF_LO_HALF=1; else export ADF_LO_HALF=0; fi
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
             "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAVES SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    rm -rf gpurun_out/r04_sqs${fuse}_$i
    rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/r04_sqs${fuse}_$i -o pmc --output-format csv -- python3 tools/scaled_time.py 3840 2160 8 2 > gpurun_out/r04_sqs${fuse}_$i.log 2>&1
  done
done
{ echo "# profiles/collect_pmc_sq_scaled.sh (round 4)"; echo "## default: fused prologue, half-width form"; python3 profiles/summarize_pmc_sq.py r04_sqs2 "down-scaled call"; echo; echo "## ADF_LO_HALF=0: fused prologue, general form"; python3 profiles/summarize_pmc_sq.py r04_sqs1 "down-scaled call"; echo; echo "## ADF_SCALED_FUSE=0: resize kernels"; python3 profiles/summarize_pmc_sq.py r04_sqs0 "down-scaled call"; } > gpurun_out/r04_pmc_sq_scaled.txt
grep -A 12 "wave_hpass_kernel<56, 2, [123]" gpurun_out/r04_pmc_sq_scaled.txt
