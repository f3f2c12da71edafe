# attribution runs of k_linear_t2: diagnostic builds exp/liblt2_exp<mask>.so (csrc/k_linear_t2.hip MIRX_LT2_EXP bit mask;
# build each with `make -C image-retrieval---thesis-2026_amd/csrc diag-lt2 MASK=<mask>`)
cd $GRAFT_REPO_ROOT
for m in ${MODELS:-dinov2}; do
  echo "== base $m"; timeout -k 10 120 python tools/bench_linear_t2.py --model $m --no-check 2>&1 | grep -v "amdgpu.ids\|Warning\|detach\|err = " | cut -c1-100 || exit 1
  for v in ${MASKS:-2 8 16 10 18 26}; do echo "== exp$v $m"; MIRX_LIB_PATH=$GRAFT_REPO_ROOT/exp/liblt2_exp$v.so timeout -k 10 120 python tools/bench_linear_t2.py --model $m --no-check 2>&1 | grep -v "amdgpu.ids\|Warning\|detach\|err = " | cut -c1-100 || exit 1; done
done
