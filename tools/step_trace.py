#!/usr/bin/env python3
"""The kernels of ONE step of a bench.py workload in launch order, with durations and gaps (GPU box, repo root):

    python3 tools/step_trace.py <tag> [-- bench.py args...]

Runs `rocprofv3 --kernel-trace -- python3 bench.py <args> --no-graph --steps 3 --warmup 2 --trials 1` as a CHILD process (this
script never touches the GPU), finds the period of the dispatch list (every step launches the same sequence)
and writes the last complete step to gpurun_out/steptrace_<tag>.txt: start offset, duration, gap to the previous kernel's
end, grid, workgroup, kernel name. The raw trace is deleted.
"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(raw):
    s = raw.replace("void ", "").replace("egx::", "").replace("(anonymous namespace)::", "")
    return s.split("(")[0]


def main():
    argv = sys.argv[1:]
    bench_args = []
    if "--" in argv:
        i = argv.index("--")
        argv, bench_args = argv[:i], argv[i + 1:]
    tag = argv[0]
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    d = os.path.join(out_dir, f"steptrace_{tag}")
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "r", "--", "python3", os.path.join(ROOT, "bench.py")] + bench_args + [
        "--no-graph", "--no-cpu-baseline", "--no-roofline", "--no-optimizer-line", "--no-native-line", "--steps", "3", "--warmup", "2", "--trials", "1"]
    with open(d + ".log", "w") as log:
        rc = subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp").returncode
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if rc != 0 or not files:
        print(f"rc={rc}, no trace; log tail:\n" + open(d + ".log").read()[-1500:])
        sys.exit(1)
    rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    # the last COMPLETE step: the shortest period P with names[-P:] == names[-2P:-P] (the steps launch identical sequences)
    n = len(names)
    per = next((P for P in range(4, n // 2 + 1) if names[n - P:] == names[n - 2 * P:n - P]), n)
    lo, hi = n - per, n
    # rotate so that the step starts at the library's first kernel (weight pack / cast) when there is one
    first = next((i for i in range(lo, hi) if names[i].startswith(("wide_cast_batch", "pack_weights"))), lo)
    lo, hi = first - per, first
    if lo < 0:
        lo, hi = n - per, n
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = t0
    lines = []
    busy = 0
    for r, n in zip(rows[lo:hi], names[lo:hi]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        grid = "x".join(str(int(r[f"Grid_Size_{a}"]) // max(int(r[f"Workgroup_Size_{a}"]), 1)) for a in "XYZ")
        lines.append("%9.1f %8.1f %7.1f  %-12s wg %-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, grid, r["Workgroup_Size_X"], n))
        busy += e - s
        prev_end = e
    head = "# %s: %d kernels, %.1f us busy, %.1f us first start to last end\n#  start_us   dur_us  gap_us  grid(wgs)    wg   kernel\n" % (
        " ".join(bench_args), hi - lo, busy / 1e3, (prev_end - t0) / 1e3)
    path = os.path.join(out_dir, f"steptrace_{tag}.txt")
    open(path, "w").write(head + "\n".join(lines) + "\n")
    print(head + "\n".join(lines[:400]))
    subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
