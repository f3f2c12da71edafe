cd $GRAFT_REPO_ROOT
for m in dinov2 medsiglip; do
  echo "== base $m"; timeout -k 10 120 python tools/bench_linear_h2.py --model $m || exit 1
  for v in 1 2 3 4; do echo "== exp$v $m"; MIRX_LIB_PATH=$GRAFT_REPO_ROOT/exp/liblh2_exp$v.so timeout -k 10 120 python tools/bench_linear_h2.py --model $m || exit 1; done
done
