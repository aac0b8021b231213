#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_traffic.sh <dtype>
# Two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), eager launches so
# that every kernel is its own dispatch. Writes gpurun_out/pmc_<dtype>.json = per-kernel average HBM traffic per launch,
# corrected as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB; FETCH_SIZE counts
# 128-B read requests as 64 B on gfx950 (doubled here); WRITE_SIZE is exact.
dt=${1:-f32}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${dt}_$c -o r -- python3 bench.py --dtype $dt --no-graph --no-cpu-baseline --no-roofline --no-optimizer-line --no-native-line --steps 20 --warmup 5 > gpurun_out/pmc_${dt}_$c.log 2>&1
done
python3 - "$dt" <<'PY'
import csv, glob, json, sys, collections
dt = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc_{dt}_{c}/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no counter file for", c); print(open(f"gpurun_out/pmc_{dt}_{c}.log").read()[-1500:]); sys.exit(1)
    acc = collections.defaultdict(lambda: [0.0, 0])
    seen = set()
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != c:
            continue
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[name][0] += float(row["Counter_Value"])
        key = (row["Dispatch_Id"], name)
        if key not in seen:
            seen.add(key); acc[name][1] += 1
    for name, (tot, n) in acc.items():
        out.setdefault(name, {})[c + "_KiB_per_launch_raw"] = tot / max(n, 1)
        out[name]["launches"] = n
res = {}
for name, d in out.items():
    f = d.get("FETCH_SIZE_KiB_per_launch_raw", 0.0) * 1024 * 2      # gfx950: half-counted
    w = d.get("WRITE_SIZE_KiB_per_launch_raw", 0.0) * 1024
    res[name] = {"read_bytes": f, "write_bytes": w, "traffic_bytes": f + w, "launches": d["launches"], **d}
json.dump({"dtype": dt, "workload": {"config": "c2", "batch": 256, "frames": 15, "layers": 1}, "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half count) + WRITE_SIZE KiB x 1024", "kernels": res},
          open(f"gpurun_out/pmc_{dt}.json", "w"), indent=1)
for name, d in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes"])[:12]:
    print(f"{name[:60]:60s} n={d['launches']:4d} read {d['read_bytes']/1e6:9.2f} MB  write {d['write_bytes']/1e6:9.2f} MB")
PY
