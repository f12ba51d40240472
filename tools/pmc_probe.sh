#!/bin/bash
# PMC passes on the dominant kernel in isolation (tools/dominant_kernel.py [frame group] with MSLAM_GEMM=<cfg> optional):
# one counter group per run, as the guide prescribes; summary JSON -> gpurun_out/pmc/<tag>.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
B=${1:-4}; TAG=${2:-dominant}
rm -rf /tmp/pmc_*
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$tag -o p -- python3 tools/dominant_kernel.py $B > gpurun_out/pmc/run_$tag.log 2>&1
done
python3 tools/pmc_dominant.py $B > gpurun_out/pmc/$TAG.json
cat gpurun_out/pmc/$TAG.json
