#!/bin/bash
# per-kernel totals of one embed run, two trees side by side (same box): tools/ab_kernels.sh <other tree> [batch]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
other=$1; b=${2:-4096}
for tag in new old; do
  dir=$GRAFT_REPO_ROOT; [ $tag = old ] && dir=$GRAFT_REPO_ROOT/$other
  rm -rf gpurun_out/ab_$tag
  (cd $dir && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab_$tag -- python tools/bench_embed.py --batch $b --iters 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/ab_$tag.log 2>&1)
  tail -1 gpurun_out/ab_$tag.log
done
python - <<'PY'
import csv, glob, re
def load(tag):
    f = sorted(glob.glob(f"gpurun_out/ab_{tag}/**/*kernel_stats.csv", recursive=True))[-1]
    out = {}
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("mirx::(anonymous namespace)::", ""))[:60]
        out[k] = out.get(k, 0.0) + float(r["TotalDurationNs"]) / 1e6
    return out
a, b = load("new"), load("old")
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, 0) + b.get(k, 0)))[:16]
print(f"{'kernel':60s} {'new ms':>9s} {'old ms':>9s}")
for k in keys:
    print(f"{k:60s} {a.get(k, 0):9.2f} {b.get(k, 0):9.2f}")
print(f"{'total':60s} {sum(a.values()):9.2f} {sum(b.values()):9.2f}")
PY
rm -rf gpurun_out/ab_new gpurun_out/ab_old
