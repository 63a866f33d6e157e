#!/bin/bash
# conv decomposition sweep + PMC counters for the conv kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
echo "== sweep"; timeout 900 python scripts/conv_sweep.py --out gpurun_out/conv_sweep.json 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$R/gpurun_out/counters.txt" 2>&1
echo "== pmc pass 1"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$R/gpurun_out/pmc1" -o p1 -- python "$R/scripts/conv_sweep.py" --layers conv2_2 conv4_2 conv5_1 --cfgs -1 4 5 --iters 2 > "$R/gpurun_out/pmc1.log" 2>&1; echo "rc=$?"
echo "== pmc pass 2"
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d "$R/gpurun_out/pmc2" -o p2 -- python "$R/scripts/conv_sweep.py" --layers conv2_2 conv4_2 conv5_1 --cfgs -1 4 5 --iters 2 > "$R/gpurun_out/pmc2.log" 2>&1; echo "rc=$?"
cd "$R"; tail -3 gpurun_out/pmc1.log gpurun_out/pmc2.log; find gpurun_out/pmc1 gpurun_out/pmc2 -name "*.csv" | head
