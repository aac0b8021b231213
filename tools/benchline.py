import sys,json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line); r=d["roofline"]
    print(d["dtype"], round(d["value"]), "clips/s", round(d["ms_per_step"]*1000), "us/step; bwd", round(r["avg_launch_us"]), "us frac", round(r["frac"],3), {k: round(v["avg_launch_us"]) for k,v in r["other_kernels"].items()})
