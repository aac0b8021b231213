#!/usr/bin/env python3
"""Hardware counters of one bench.py workload, per kernel (GPU box, repo root):

    python3 tools/pmc_collect.py <tag> [--passes traffic,sq] [-- bench.py args...]

Runs `rocprofv3 --pmc ... --kernel-trace -- python3 bench.py <args> --no-graph ...` once per counter group (FETCH_SIZE and
WRITE_SIZE do not fit one pass on gfx950; the SQ counters need two passes of <= 8) as CHILD processes — this script never
touches the GPU itself — with eager launches so that every kernel is its own dispatch, and writes
gpurun_out/pmc_<tag>.json:

    {"csrc_sha": sha256 of egot2_amd/csrc + include (bench.csrc_sha(): bench.py drops the numbers when the kernels changed),
     "workload": {...bench args...}, "steps_profiled": W + K,
     "kernels": {name: {"launches", "read_bytes", "write_bytes", "traffic_bytes" (per launch), SQ_* (per launch)}}}

HBM bytes are corrected as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB; FETCH_SIZE
counts 128-B read requests as 64 B on gfx950 (doubled here); WRITE_SIZE is exact. Copy the files to keep into profiles/.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SQ1 = "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2 = "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"
IF1 = "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
GROUPS = {"ifetch": [("if1", IF1)], "traffic": [("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")], "sq": [("sq1", SQ1), ("sq2", SQ2)]}


def kname(raw):
    return raw.split("(")[0].replace("void ", "").split("<")[0]


def main():
    argv = sys.argv[1:]
    bench_args = []
    if "--" in argv:
        i = argv.index("--")
        argv, bench_args = argv[:i], argv[i + 1:]
    tag = argv[0]
    passes = "traffic,sq"
    if "--passes" in argv:
        passes = argv[argv.index("--passes") + 1]
    steps, warmup = 10, 3
    import bench
    a = bench.parse_args(bench_args)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(dict)
    for grp in passes.split(","):
        for pname, counters in GROUPS[grp]:
            d = os.path.join(out_dir, f"pmc_{tag}_{pname}")
            subprocess.run(["rm", "-rf", d])
            cmd = ["rocprofv3", "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "r", "--",
                   "python3", os.path.join(ROOT, "bench.py")] + bench_args + ["--no-graph", "--no-cpu-baseline", "--no-roofline", "--no-optimizer-line",
                   "--no-native-line", "--steps", str(steps), "--warmup", str(warmup), "--trials", "1"]
            with open(d + ".log", "w") as log:
                rc = subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT, env=env, cwd="/tmp").returncode
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                print(f"pass {pname}: rc={rc}, no counters; log tail:\n" + open(d + ".log").read()[-1500:])
                continue
            seen = collections.defaultdict(set)
            for row in csv.DictReader(open(files[0])):
                name = kname(row["Kernel_Name"])
                per[name][row["Counter_Name"]] += float(row["Counter_Value"])
                seen[name].add(row["Dispatch_Id"])
            for name, s in seen.items():
                launches[name][pname] = len(s)
            if "--keep-raw" not in sys.argv:       # the raw counter CSVs are tens of MB per pass: gpurun merges back <= 64 MiB
                subprocess.run(["rm", "-rf", d])
    kernels = {}
    for name, d in per.items():
        o = {}
        for pname, counters in GROUPS["traffic"] + GROUPS["sq"] + GROUPS["ifetch"]:
            n = launches[name].get(pname)
            if not n:
                continue
            o["launches"] = n
            for c in counters.split():
                if c in d:
                    o[c] = d[c] / n
        if "FETCH_SIZE" in o or "WRITE_SIZE" in o:
            o["read_bytes"] = o.pop("FETCH_SIZE", 0.0) * 1024 * 2          # gfx950: half-counted
            o["write_bytes"] = o.pop("WRITE_SIZE", 0.0) * 1024
            o["traffic_bytes"] = o["read_bytes"] + o["write_bytes"]
        kernels[name] = o
    res = {"csrc_sha": bench.csrc_sha(),
           "workload": {"config": a.config, "batch": bench.resolve_batch(a), "frames": a.frames, "layers": a.layers, "dtype": a.dtype,
                        "encoder_only": bool(a.encoder_only), "bench_args": bench_args},
           "steps_profiled": steps + warmup + 1,      # + the eager step bench.py runs to count the library's launches
           "correction": "read_bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 half count); write_bytes = WRITE_SIZE KiB x 1024; all values per launch",
           "kernels": kernels}
    path = os.path.join(out_dir, f"pmc_{tag}.json")
    json.dump(res, open(path, "w"), indent=1)
    print("wrote", path)
    tot = 0.0
    for name, o in sorted(kernels.items(), key=lambda kv: -kv[1].get("traffic_bytes", 0) * kv[1].get("launches", 0))[:14]:
        if not name.startswith("egx::"):
            continue
        per_step = o.get("traffic_bytes", 0) * o.get("launches", 0) / (steps + warmup + 1)
        tot += per_step
        line = f"{name[:48]:48s} n={o.get('launches', 0):5d} read {o.get('read_bytes', 0) / 1e6:8.2f} MB write {o.get('write_bytes', 0) / 1e6:8.2f} MB /launch, {per_step / 1e6:8.1f} MB/step"
        wc = o.get("SQ_WAVE_CYCLES")
        if wc:
            busy = o.get("SQ_BUSY_CYCLES", 0) / 32.0
            line += ("  | mfma %.3g valu %.3g (%.1f/mfma) mfma_busy %.0f%% wait_any %.0f%% wait_inst %.0f%% lds_conf %.3g" % (
                o.get("SQ_INSTS_MFMA", 0), o.get("SQ_INSTS_VALU", 0), o.get("SQ_INSTS_VALU", 0) / max(o.get("SQ_INSTS_MFMA", 0), 1),
                100 * o.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(busy, 1), 100 * o.get("SQ_WAIT_ANY", 0) / wc,
                100 * o.get("SQ_WAIT_INST_ANY", 0) / wc, o.get("SQ_LDS_BANK_CONFLICT", 0)))
        print(line)
    print(f"egx kernels: {tot / 1e6:.1f} MB of HBM traffic per step")


if __name__ == "__main__":
    main()
