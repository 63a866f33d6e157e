#!/bin/bash
# Round 2: wave-parallel chunk resolve + strided second-stage mask launch -- NMS / proposal parity and timing.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02h}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest nms/proposals"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "nms or proposal or forward" --timeout 600 > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
echo "== prop bench"; timeout 300 python scripts/prop_bench.py > $O/prop.log 2>&1; grep -v amdgpu.ids $O/prop.log | tail -8
for w in 256 512 2048; do echo "tail wgs $w"; FRCNN_NMS_TAIL_WGS=$w timeout 300 python scripts/prop_bench.py 2>&1 | grep "SCAN=0\|train mode"; done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prop" -o prop -- python "$R/scripts/prop_bench.py" > "$R/$O/prop_prof.log" 2>&1; echo "prof rc=$?"
cd "$R"; head -12 $O/prop/prop_kernel_stats.csv | cut -c1-200
