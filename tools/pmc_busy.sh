#!/bin/bash
# Matrix-pipe utilisation of the hot kernels (run on the GPU box): one rocprofv3 --pmc pass (no trace) per workload.
# SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs, so the fraction of SIMD-cycles
# with the matrix pipe busy is MFMA_BUSY / (128 x GUI_ACTIVE) -- at the clock the kernel actually holds.
# Summary -> gpurun_out/${R:-r04}_mfma_busy.txt (copy to profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/busy_$tag -- "$@" > gpurun_out/busy_$tag.log 2>&1; }
run gemm python tools/bench_search.py --q 4096 --iters 1
run embed python tools/bench_embed.py --batch 1024 --iters 1 --warmup 1
python - <<'PY' > gpurun_out/${R:-r04}_mfma_busy.txt
import csv, glob, collections, re
print("# rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES (one pass, no trace); per kernel, summed over launches")
print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE): MFMA_BUSY sums 1024 SIMDs, GUI_ACTIVE sums 8 XCDs; the fraction of SIMD-cycles with the matrix pipe busy, at the clock the kernel actually holds")
for tag in ("gemm", "embed"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/busy_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("mirx::(anonymous namespace)::", "mirx::"))[:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:30]:
        g, m, b = v.get("GRBM_GUI_ACTIVE", 0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("SQ_BUSY_CU_CYCLES", 0)
        if g <= 0:
            continue
        if m <= 0:
            continue
        print(f"{tag:6s} {k:60s} GUI_ACTIVE {g:14.0f}  MFMA_BUSY {m:16.0f}  mfma_busy {m / (128 * g):6.3f}")
PY
cat gpurun_out/${R:-r04}_mfma_busy.txt
