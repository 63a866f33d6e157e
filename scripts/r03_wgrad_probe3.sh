cd ${GRAFT_REPO_ROOT:-/root/repo}; B=scripts/micro/_bin; O=gpurun_out/r03wg; mkdir -p $O
{
echo "== default"; $B/wgrad_micro conv1_2 conv3_2 conv4_2 conv5_1
echo "== FRCNN_WGRAD_DB=1 (DMAs ahead of the MFMAs)"; FRCNN_WGRAD_DB=1 $B/wgrad_micro conv1_2 conv3_2 conv4_2 conv5_1
echo "== FRCNN_WGRAD_DB=2 (DMAs inside the MFMA stream)"; FRCNN_WGRAD_DB=2 $B/wgrad_micro
} > $O/wgrad_micro3.txt 2>&1; cat $O/wgrad_micro3.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "conv_wgrad_forms or conv_backward" --timeout 300 2>&1 | tail -3
