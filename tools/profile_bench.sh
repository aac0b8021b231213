#!/bin/bash
# usage (GPU box): tools/profile_bench.sh <config> <tag> [bench.py args...]
#   rocprofv3 --kernel-trace --stats of `python3 bench.py --config <config> ...`; prints the top kernels and leaves the
#   summary in gpurun_out/<tag>_kernel_stats.csv (copy the ones to keep into profiles/).
cfg=${1:-c2}; tag=${2:-prof_$cfg}; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
rm -rf gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o r -- python3 bench.py --config $cfg --no-roofline --no-cpu-baseline --no-optimizer-line --no-native-line "$@" > gpurun_out/$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, shutil, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv; log tail:"); print(open(f"gpurun_out/{tag}.log").read()[-2000:]); sys.exit(1)
shutil.copy(f[0], f"gpurun_out/{tag}_kernel_stats.csv")
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms")
for r in rows[:22]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us avg", r["Percentage"].rjust(6), "%")
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/$tag.log | head -1 || true
rm -rf gpurun_out/$tag        # the raw trace (tens of MB); the summary CSV is kept
