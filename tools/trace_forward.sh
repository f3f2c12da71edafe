#!/bin/bash
# kernel traces of one DenseNet forward at several batch sizes -> gpurun_out/kt_<B>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in "$@"; do
  rm -rf gpurun_out/kt_$b
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_$b -- python tools/bench_embed.py --batch $b --iters 3 --warmup 2 > gpurun_out/kt_$b.log 2>&1
  python tools/layer_times.py gpurun_out/kt_$b --all > gpurun_out/kt_$b.txt
  tail -1 gpurun_out/kt_$b.log
  rm -rf gpurun_out/kt_$b
done
