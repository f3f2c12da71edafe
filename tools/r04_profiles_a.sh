#!/bin/bash
# round-4 evidence, part A: the contract line, the bench's kernel split, the search sweep -> gpurun_out/r04_* (copy to profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; echo "bench rc=$?"
rm -rf gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --embed-streams 1 > gpurun_out/prof_bench.log 2>&1
cp "$(find gpurun_out/prof_bench -name '*kernel_stats.csv' | head -1)" gpurun_out/r04_bench_kernel_stats.csv
rm -rf gpurun_out/prof_bench
python tools/bench_search.py --sweep 1,64,1024,8192 --json gpurun_out/r04_search_sweep.json > gpurun_out/r04_search_sweep.log 2>&1
tail -5 gpurun_out/r04_search_sweep.log
# the small-launch kernels: kernel split of the DenseNet forward at the reference's batch sizes
for b in 1 64; do
  rm -rf gpurun_out/prof_b$b
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$b -- python tools/bench_embed.py --batch $b --iters 20 --warmup 3 > gpurun_out/prof_b$b.log 2>&1
  cp "$(find gpurun_out/prof_b$b -name '*kernel_stats.csv' | head -1)" gpurun_out/r04_densenet_b${b}_kernel_stats.csv
  rm -rf gpurun_out/prof_b$b
done
