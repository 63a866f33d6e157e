cd ${GRAFT_REPO_ROOT:-/root/repo}; B=scripts/micro/_bin; O=gpurun_out/r03wg; mkdir -p $O
{
for pm in 1 2 3; do for w in 2 3; do echo "== FRCNN_WGRAD_PRIO=$pm WPS=$w"; FRCNN_WGRAD_PRIO=$pm FRCNN_WGRAD_WPS=$w $B/wgrad_micro conv1_2 conv3_2 conv4_2 conv5_1; done; done
echo "== default"; $B/wgrad_micro conv1_2 conv3_2 conv4_2 conv5_1
} > $O/wgrad_micro2.txt 2>&1; cat $O/wgrad_micro2.txt
