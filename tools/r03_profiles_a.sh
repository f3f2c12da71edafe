#!/bin/bash
# round-3 evidence, part A: the contract line, the bench's kernel split, the search sweep -> gpurun_out/r03_* (copy to profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err; echo "bench rc=$?"
rm -rf gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --embed-streams 1 > gpurun_out/prof_bench.log 2>&1
cp "$(find gpurun_out/prof_bench -name '*kernel_stats.csv' | head -1)" gpurun_out/r03_bench_kernel_stats.csv
rm -rf gpurun_out/prof_bench
python tools/bench_search.py --sweep 1,64,1024,8192 --json gpurun_out/r03_search_sweep.json > gpurun_out/r03_search_sweep.log 2>&1
tail -5 gpurun_out/r03_search_sweep.log
