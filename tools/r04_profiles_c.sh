#!/bin/bash
# Round-4 evidence for the token-major Linear kernels (run on the GPU box): micro benchmark of mirx_linear_terms against
# mirx_linear_split2h on the DINOv2 / MedSigLIP layer shapes, the in-kernel cycle stamps of k_linear_t2 (diagnostic library
# exp/liblt2_exp32.so, csrc/k_linear_t2.hip MIRX_LT2_EXP=32), and the PMC split of both kernels.  Outputs: gpurun_out/r04_linear_*.
cd $GRAFT_REPO_ROOT
{
  for m in dinov2 medsiglip; do
    echo "== $m, bench batch"; timeout -k 10 200 python tools/bench_linear_t2.py --model $m 2>&1 | grep -v "amdgpu.ids\|Warning\|detach\|err = " || exit 1
  done
  echo "== dinov2, 2740 token rows (two images: every tile cut along K)"; timeout -k 10 200 python tools/bench_linear_t2.py --model dinov2 --tokens 2740 2>&1 | grep -v "amdgpu.ids\|Warning\|detach\|err = "
  if [ -f exp/liblt2_exp32.so ]; then
    echo "== dinov2, cycle stamps of the last launch of each timing loop (wave 0 of every workgroup; diagnostic build)"
    MIRX_LIB_PATH=$GRAFT_REPO_ROOT/exp/liblt2_exp32.so timeout -k 10 200 python tools/bench_linear_t2.py --model dinov2 --no-check --iters 20 --stamps 2>&1 | grep stamps
  fi
} > gpurun_out/r04_linear_bench.txt 2>&1
bash tools/pmc_linear.sh dinov2 > gpurun_out/r04_linear_pmc.txt 2>&1
tail -40 gpurun_out/r04_linear_bench.txt; cat gpurun_out/r04_linear_pmc.txt
