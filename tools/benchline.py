"""One-line summary of bench.py JSON lines (development aid): python tools/benchline.py [file] [label]  (or stdin)."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
label = sys.argv[2] if len(sys.argv) > 2 else ""
for line in src:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline") or {}
    ks = {}
    if r.get("kernel") and r.get("avg_launch_us"):
        ks[r["kernel"].split("::")[-1]] = (r["avg_launch_us"], r["frac"])
        ks.update({k: (v.get("avg_launch_us") or v.get("us_per_step"), v["frac"]) for k, v in (r.get("other_kernels") or {}).items()})
    print(label, d.get("arithmetic", d["dtype"])[:6], round(d["value"]), "clips/s", round(d["ms_per_step"] * 1000, 1), "us/step;",
          " ".join(f"{k.replace('_kernel', '')}={round(t)}us({f:.3f})" for k, (t, f) in ks.items()))
