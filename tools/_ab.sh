#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$GRAFT_REPO_ROOT/exp
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py tests/test_vit_gpu.py tests/test_siglip_gpu.py tests/test_linear_gpu.py -x -q > gpurun_out/t_att.log 2>&1; tail -3 gpurun_out/t_att.log
for m in "dinov2 32 518" "medsiglip 16 448"; do
  set -- $m
  echo "== $1 B=$2 new"; python tools/bench_embed.py --model $1 --batch $2 --size $3 --iters 6 --warmup 2 2>&1 | tail -1
  echo "== $1 B=$2 prev"; MIRX_LIB_PATH=$E/libprev.so python tools/bench_embed.py --model $1 --batch $2 --size $3 --iters 6 --warmup 2 2>&1 | tail -1
  echo "== $1 B=$2 new"; python tools/bench_embed.py --model $1 --batch $2 --size $3 --iters 6 --warmup 2 2>&1 | tail -1
done
