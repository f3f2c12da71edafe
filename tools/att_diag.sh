#!/bin/bash
# phase attribution of k_attention_h2 from its diagnostic builds (make diag-src SRC=k_attention_h2 NAME=att_expM FLAGS=-DMIRX_ATT_EXP=M;
# bits: 1 no softmax arithmetic, 2 no tile staging, 4 no P V MFMAs, 8 no Q K MFMAs, 16 no barrier; results wrong, timing only)
cd $GRAFT_REPO_ROOT
echo "shipped:"; python tools/bench_attention.py --iters 20 2>&1 | grep "split-2"
for m in "$@"; do echo "MIRX_ATT_EXP=$m:"; MIRX_LIB_PATH=$GRAFT_REPO_ROOT/exp/libatt_exp$m.so python tools/bench_attention.py --iters 20 2>&1 | grep "split-2"; done
