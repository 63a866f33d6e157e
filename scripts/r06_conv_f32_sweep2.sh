cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for L in conv2_1 conv3_1 conv3_2 conv4_1 conv4_2; do
  for c in -1 36 46 30 34 135 234 230 136; do
    printf "$L rep $rep cfg %4s: " $c; CONV_MICRO_CFG=$c CONV_MICRO_BURST=8 timeout 60 ./scripts/micro/_bin/conv_f32_micro $L 2>&1 | grep "^$L" | sed 's/.*GFLOP *//'
  done
done
done
