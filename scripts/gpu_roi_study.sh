#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d "$R/gpurun_out/rpmc1" -o p1 -- python "$R/scripts/roi_bench.py" > "$R/gpurun_out/rpmc1.log" 2>&1; echo "rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d "$R/gpurun_out/rpmc2" -o p2 -- python "$R/scripts/roi_bench.py" > "$R/gpurun_out/rpmc2.log" 2>&1; echo "rc=$?"
cd "$R"
python scripts/pmc_summary.py gpurun_out/rpmc1 gpurun_out/rpmc2 roi_pool_planes_kernel | awk 'NR<4 || /==/ {print} ' 
python - <<'PY'
import csv
for f in ['gpurun_out/rpmc1/p1_kernel_trace.csv']:
    d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'roi_pool_planes_kernel<false' in r['Kernel_Name'] or 'planes_kernelILb0' in r['Kernel_Name']]
    print('durations ns', sorted(d)[:5], len(d))
PY
