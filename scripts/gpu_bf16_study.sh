#!/bin/bash
# PMC counters for the bf16 conv kernel (separate --pmc passes, kernel-trace only).  MODE = FRCNN_BF16_DMA value.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
export FRCNN_BF16_DMA=${MODE:-23}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d "$R/gpurun_out/bpmc1" -o p1 -- python "$R/scripts/conv_bf16_sweep.py" > "$R/gpurun_out/bpmc1.log" 2>&1; echo "rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$R/gpurun_out/bpmc2" -o p2 -- python "$R/scripts/conv_bf16_sweep.py" > "$R/gpurun_out/bpmc2.log" 2>&1; echo "rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d "$R/gpurun_out/bpmc3" -o p3 -- python "$R/scripts/conv_bf16_sweep.py" > "$R/gpurun_out/bpmc3.log" 2>&1; echo "rc=$?"
cd "$R"; tail -2 gpurun_out/bpmc1.log gpurun_out/bpmc2.log gpurun_out/bpmc3.log
python - <<'PY'
import csv, glob, collections
for d in ("bpmc1", "bpmc2", "bpmc3"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "conv_" in r["Kernel_Name"] and "bf16" in r["Kernel_Name"]:
                acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for g in sorted(acc, key=int):
            print(d, "grid", g, " ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in acc[g].items()))
PY
