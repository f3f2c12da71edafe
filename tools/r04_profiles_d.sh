#!/bin/bash
# round-4 evidence, part D: in-kernel cycle stamps of the filter GEMM (diagnostic library exp/libgemm_cyc.so =
# make diag-gemm NAME=cyc FLAGS=-DMIRX_EXP_CYCLES; the kernel prints from a few workgroups) -> gpurun_out/r04_gemm_cycles.txt
cd $GRAFT_REPO_ROOT
{
  echo "# MIRX_LIB_PATH=exp/libgemm_cyc.so python tools/bench_search.py --q 4096 --iters 2   (D = 1024, 1M rows; s_memtime = shader-clock cycles)"
  MIRX_LIB_PATH=$GRAFT_REPO_ROOT/exp/libgemm_cyc.so timeout -k 10 300 python tools/bench_search.py --q 4096 --iters 2 2>&1 | grep -v "amdgpu.ids\|Warning"
} > gpurun_out/r04_gemm_cycles.txt 2>&1
tail -12 gpurun_out/r04_gemm_cycles.txt
