#!/bin/bash
# Where the cycles of the kernels of ANY command go (run on the GPU box): rocprofv3 --pmc passes (no trace), per kernel the
# matrix-pipe busy fraction, the wave-cycle split (parked / issue-stalled / issuing), vector-unit and LDS activity, bank conflicts.
# usage: tools/pmc_kernels.sh <tag> <command...>     -> gpurun_out/pmck_<tag>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
run() { p=$1; shift; rm -rf gpurun_out/pmck_${tag}_$p; rocprofv3 --pmc $CTRS --output-format csv -d gpurun_out/pmck_${tag}_$p -- "$@" > gpurun_out/pmck_${tag}_$p.log 2>&1; }
CTRS="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" run a "$@"
CTRS="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" run b "$@"
CTRS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" run c "$@"
CTRS="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" run d "$@"
python - "$tag" <<'PY' > gpurun_out/pmck_$tag.txt
import csv, glob, collections, re, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for p in "abcd":
    for f in glob.glob(f"gpurun_out/pmck_{tag}_{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("mirx::(anonymous namespace)::", "mirx::"))[:48]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:10]:
    g = v.get("GRBM_GUI_ACTIVE", 0)
    if g <= 0:
        continue
    wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{k:48s} mfma_busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (128 * g):5.3f} | wave-cycles: parked {v.get('SQ_WAIT_ANY', 0) / wc:5.3f} "
          f"issue-stall {v.get('SQ_WAIT_INST_ANY', 0) / wc:5.3f} (lds {v.get('SQ_WAIT_INST_LDS', 0) / wc:5.3f}) issuing {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.3f} "
          f"(valu {v.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.3f} scalar {v.get('SQ_ACTIVE_INST_SCA', 0) / wc:5.3f}) | valu insts/simd-cycle {v.get('SQ_INSTS_VALU', 0) / (128 * g):5.3f} | "
          f"lds: idx_active/gui {v.get('SQ_LDS_IDX_ACTIVE', 0) / (32 * g):5.3f} conflict/idx {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):5.3f} "
          f"waves/simd {wc / (128 * g):4.2f}")
PY
for p in a b c d; do rm -rf gpurun_out/pmck_${tag}_$p; done
cat gpurun_out/pmck_$tag.txt
