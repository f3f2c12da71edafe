#!/bin/bash
# phase split of k_stem_h2 from its diagnostic builds (make diag-src SRC=k_stem_h2 NAME=stem_expM FLAGS=-DMIRX_STEM_EXP=M;
# 1 no K loop, 2 no conv-tile epilogue, 4 no pooling; results wrong, timing only)
cd $GRAFT_REPO_ROOT
python tools/bench_stem.py 4096 2>&1 | grep "stem B"
for m in "$@"; do echo "-- MIRX_STEM_EXP=$m"; MIRX_LIB_PATH=$GRAFT_REPO_ROOT/exp/libstem_exp$m.so python tools/bench_stem.py 4096 2>&1 | grep "stem B"; done
