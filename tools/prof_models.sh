#!/bin/bash
# Steady-state kernel split of the token-major backbones (development + profiles/): rocprofv3 --kernel-trace --stats of
# tools/bench_embed.py per model; the stats CSVs are copied to gpurun_out/ for inspection.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "convnextv2 64 384" "dinov2 32 518" "medsiglip 16 448"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python tools/bench_embed.py --model $1 --batch $2 --size $3 --iters 5 > gpurun_out/prof_$1.log 2>&1
  f=$(find gpurun_out/prof_$1 -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${R:-r04}_$1_kernel_stats.csv
  tail -1 gpurun_out/prof_$1.log
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
mirx = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith("mirx::") or "mirx::" in r["Name"])
print(f"  total {tot/1e6:.1f} ms, mirx share {100*mirx/tot:.1f} %")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f"  {float(r['TotalDurationNs'])/1e6:9.2f} ms {r['Calls']:>6} {r['Name'][:110]}")
PY
done
