#!/bin/bash
# PMC comparison of the three bf16-MFMA convolution kernels on conv3_2 (256 -> 256, 150x250): plain bf16 (mode 141 / 231), the
# split-product forward kernel and the split-product weight gradient.  Separate --pmc passes, kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r02pmc; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cat > /tmp/one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["R"])
import numpy as np
import chainer_faster_rcnn_amd as pkg
rt = pkg.runtime.default_runtime()
rs = np.random.RandomState(0)
ci = co = 256; h, w = 150, 250
x = rt.mem.from_numpy(np.maximum(rs.randn(1, ci, h, w), 0).astype(np.float32))
dy = rt.mem.from_numpy((rs.randn(1, co, h, w) * 0.1).astype(np.float32))
wt = rt.mem.from_numpy((rs.randn(co, ci, 3, 3) * 0.03).astype(np.float32))
b = rt.mem.from_numpy(np.zeros(co, np.float32))
xb, wb = rt.bf16_from_nchw(x), rt.bf16_pack_conv_w(wt, 3)
xs, ws = rt.f32s_from_nchw(x), rt.f32s_pack_conv_w(wt)
out = rt.mem.empty((ci * 9, co), "f32")
for _ in range(12):
    rt.conv_bf16(xb, wb, b, ci, co, 3, relu=True)
    rt.conv3x3_f32s(xs, ws, b, ci, co, relu=True)
    rt.conv_wgrad_f32s(x, dy, out=out)
    rt.conv_wgrad(x, dy, 3, out=out)
rt.mem.synchronize()
PY
cd /tmp; export TMPDIR=/tmp R
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$R/$O/p1" -o p1 -- python /tmp/one.py > "$R/$O/p1.log" 2>&1; echo "rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$R/$O/p2" -o p2 -- python /tmp/one.py > "$R/$O/p2.log" 2>&1; echo "rc=$?"
cd "$R"
python - <<'PY'
import csv, glob, collections, json
out = {}
for d in ("p1", "p2"):
    for f in glob.glob("gpurun_out/r02pmc/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            key = "conv_dma_bf16" if "conv_dma_bf16" in n else "conv_f32s" if "conv_f32s_kernel" in n else "wgrad_f32s" if "conv_wgrad_f32s" in n else "wgrad_f32" if "conv_wgrad_dma" in n else None
            if key:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in acc:
            out.setdefault(k, {}).update({c: sum(v[2:]) / max(1, len(v[2:])) for c, v in acc[k].items()})
for k, v in out.items():
    busy = v.get("SQ_BUSY_CYCLES", 0) or 1
    print(k, {c: float("%.4g" % x) for c, x in sorted(v.items())})
json.dump(out, open("gpurun_out/r02pmc/summary.json", "w"), indent=1, sort_keys=True)
PY
