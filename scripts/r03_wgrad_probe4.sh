cd ${GRAFT_REPO_ROOT:-/root/repo}; B=scripts/micro/_bin; O=gpurun_out/r03wg; mkdir -p $O
{ echo "== conv1_1 weight gradient: first-layer form"; $B/wgrad_micro conv1_1; echo "== generic"; FRCNN_WGRAD_CONV1=generic $B/wgrad_micro conv1_1; } > $O/wgrad_conv1.txt 2>&1; cat $O/wgrad_conv1.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "conv_wgrad_forms or conv_backward or train" --timeout 300 2>&1 | tail -3
timeout 300 python bench.py --mode train --steps 30 --warmup 3 > $O/train_conv1.json 2>/dev/null; cut -c1-200 $O/train_conv1.json
