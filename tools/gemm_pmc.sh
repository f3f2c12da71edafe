#!/bin/bash
# usage: pmc.sh <tag> [env assignments...]   -- three PMC passes of one search-only launch set
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "$@"; do export "$a"; done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcx_${tag}_a -- python tools/bench_search.py --q 4096 --iters 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmcx_${tag}_b -- python tools/bench_search.py --q 4096 --iters 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmcx_${tag}_c -- python tools/bench_search.py --q 4096 --iters 1 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcx_${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gemm" in r["Kernel_Name"] and (", 0, false" in r["Kernel_Name"] or "gemm16<0, false" in r["Kernel_Name"]):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("${tag}", {k: round(sum(v)/len(v)) for k, v in sorted(acc.items())})
PY
