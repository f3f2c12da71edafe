#!/bin/bash
# Where the cycles of the token-major Linear kernels go (run on the GPU box): three rocprofv3 --pmc passes (no trace) over
# tools/bench_linear_t2.py.  Per kernel: matrix-pipe busy fraction, wave-cycle split (parked / issue-stalled / issuing), LDS.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
model=${1:-dinov2}
run() { tag=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcl_$tag -- python tools/bench_linear_t2.py --model $model --no-check --iters 2 > gpurun_out/pmcl_$tag.log 2>&1; }
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES || exit 1
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS || exit 1
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS || exit 1
python - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for tag in "abc":
    for f in glob.glob(f"gpurun_out/pmcl_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("mirx::(anonymous namespace)::", "mirx::"))[:48]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:12]:
    g = v.get("GRBM_GUI_ACTIVE", 0)
    if g <= 0 or v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print(f"{k:48s} mfma_busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (128 * g):5.3f} | wave-cycles: parked {v.get('SQ_WAIT_ANY', 0) / wc:5.3f} "
          f"issue-stall {v.get('SQ_WAIT_INST_ANY', 0) / wc:5.3f} (lds {v.get('SQ_WAIT_INST_LDS', 0) / wc:5.3f}) issuing {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.3f} | "
          f"lds: idx_active/gui {v.get('SQ_LDS_IDX_ACTIVE', 0) / (32 * g):5.3f} conflict/idx {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):5.3f} "
          f"waves/simd {wc / (128 * g):4.2f}")
PY
