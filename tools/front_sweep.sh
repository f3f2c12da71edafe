#!/bin/bash
# MALL experiment: stem + first dense block(s) on sub-batches whose buffers stay in the 256 MiB Infinity Cache
set -e
out=gpurun_out/front_sweep.txt
: > $out
for cfg in "0 1 1" "0 1 2" "16 1 1" "32 1 1" "48 1 1" "64 1 1" "128 1 1" "32 1 2" "64 1 2" "32 2 1" "64 2 1" "64 2 2" "256 1 2"; do
  set -- $cfg
  echo "== front_subbatch=$1 front_blocks=$2 streams=$3" >> $out
  python tools/bench_embed.py --batch 4096 --iters 6 --warmup 2 --streams $3 --front-subbatch $1 --front-blocks $2 >> $out 2>&1
done
cat $out
