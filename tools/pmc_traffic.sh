#!/bin/bash
# Fabric-side traffic of the dominant kernels (run on the GPU box): separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
# never combined with a trace), summarised by tools/pmc_traffic.py into gpurun_out/${R:-r04}_pmc_traffic.json -- copy that to
# profiles/.  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE counts half the bytes of wide coalesced loads on gfx950.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # tag counter command...
  tag=$1; ctr=$2; shift 2
  rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmct_${tag}_${ctr} -- "$@" > gpurun_out/pmct_${tag}_${ctr}.log 2>&1
}
for ctr in FETCH_SIZE WRITE_SIZE; do
  run gemm $ctr python tools/bench_search.py --q 4096 --iters 1
  run densenet $ctr python tools/bench_embed.py --batch 2048 --iters 1 --warmup 1
  run convnextv2 $ctr python tools/bench_embed.py --model convnextv2 --batch 64 --size 384 --iters 1 --warmup 1
  run dinov2 $ctr python tools/bench_embed.py --model dinov2 --batch 32 --size 518 --iters 1 --warmup 1
  run medsiglip $ctr python tools/bench_embed.py --model medsiglip --batch 16 --size 448 --iters 1 --warmup 1
done
python tools/pmc_traffic.py > gpurun_out/${R:-r04}_pmc_traffic.json
cat gpurun_out/${R:-r04}_pmc_traffic.json
