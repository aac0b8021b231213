#!/usr/bin/env python3
"""The numbers DESIGN.md section 5 / README quote, straight from the committed bench lines and kernel stats of a round:

    python3 tools/profile_table.py [rNN]        (default r04; reads profiles/<rNN>_bench_*.json and *_kernel_stats.csv)
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    r = sys.argv[1] if len(sys.argv) > 1 else "r04"
    base = os.path.join(ROOT, "profiles", f"{r}_bench_")
    lines = {}
    for f in sorted(glob.glob(base + "*.json")):
        name = f[len(base):-5]
        try:
            lines[name] = json.load(open(f))
        except Exception as e:      # a failed bench leaves no JSON; a truncated one is reported, not fatal
            print(f"{name}: unreadable ({e})")
    ref = lines.get("f32s", {}).get("ms_per_step")
    print(f"{'line':28s} {'ms/step':>9s} {'clips/s':>10s}  roofline (kernel, frac, HBM MB/step)")
    for name, d in lines.items():
        ro = d.get("roofline") or {}
        extra = ""
        if ro:
            mb = ro.get("hbm_bytes_per_step")
            extra = f"{ro.get('kernel', '').replace('egx::', '')} {ro.get('frac', 0):.3f}" + (f" {mb / 1e6:.0f} MB" if mb else "")
        print(f"{name:28s} {d['ms_per_step']:9.4f} {d['value']:10.0f}  {extra}")
    if ref:
        print("\nper-token cost of the long-sequence lines relative to T = 15 (3 840 frames per task at B = 256):")
        for name, frames in (("c2_t30", 30), ("c2_t60", 60), ("c2_t150", 150)):
            if name in lines:
                d = lines[name]
                b = d["config"]["global_batch"]
                print(f"  {name}: {d['ms_per_step'] / ref * (3840 / (b * frames)):.2f}x")
    cb = lines.get("f32s", {}).get("cpu_baseline")
    if cb:
        print(f"\ncpu_baseline: {cb['value']:.0f} clips/s at {cb['cores']} threads ({cb.get('cpu_model', '?')}); sweep {cb.get('thread_sweep')}; "
              f"p=0 {cb.get('train_p0', {}).get('value', 0):.0f}; eval {cb.get('eval_forward', {}).get('value', 0):.0f}")
    for f in sorted(glob.glob(base + "*_kernel_stats.csv")):
        name = f[len(base):-len("_kernel_stats.csv")]
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(x["TotalDurationNs"]) for x in rows) or 1.0
        top = ", ".join(f"{x['Name'].replace('void ', '').replace('egx::', '').split('(')[0][:34]} {float(x['AverageNs']) / 1e3:.0f}us x{int(x['Calls'])} ({100 * float(x['TotalDurationNs']) / tot:.0f}%)"
                        for x in rows[:5])
        print(f"\n{name}: {top}")


if __name__ == "__main__":
    main()
