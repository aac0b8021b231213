#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile.sh <tag> <bench.py args...>
# Writes gpurun_out/prof_<tag>/ (rocprofv3 --kernel-trace --stats, CSV) and prints the top kernels.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o r -- python3 bench.py "$@" > gpurun_out/prof_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_{tag}/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv; log tail:"); print(open(f"gpurun_out/prof_{tag}.log").read()[-2000:]); sys.exit(1)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:24]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(6), r["AverageNs"].rjust(12), r["Percentage"])
PY
