#!/bin/bash
# PMC counters for the conv kernel on representative layers (separate --pmc passes, kernel-trace only)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
LAYERS="${LAYERS:-conv1_2 conv2_2 conv3_2 conv4_2 conv5_1}"
CFGS="${CFGS:--1}"
echo "== timing"; timeout 600 python scripts/conv_sweep.py --layers $LAYERS --cfgs $CFGS 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
echo "== pmc pass 1"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$R/gpurun_out/pmc1" -o p1 -- python "$R/scripts/conv_sweep.py" --layers $LAYERS --cfgs $CFGS --iters 2 > "$R/gpurun_out/pmc1.log" 2>&1; echo "rc=$?"
echo "== pmc pass 2"
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$R/gpurun_out/pmc2" -o p2 -- python "$R/scripts/conv_sweep.py" --layers $LAYERS --cfgs $CFGS --iters 2 > "$R/gpurun_out/pmc2.log" 2>&1; echo "rc=$?"
cd "$R"; tail -3 gpurun_out/pmc1.log gpurun_out/pmc2.log
python scripts/pmc_summary.py gpurun_out/pmc1 gpurun_out/pmc2 conv_mfma
