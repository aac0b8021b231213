import sys,json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line); r=d["roofline"]
    ks={r["kernel"].split("::")[-1]: (r["avg_launch_us"], r["frac"])}
    ks.update({k:(v["avg_launch_us"], v["frac"]) for k,v in r["other_kernels"].items()})
    print(d["dtype"], round(d["value"]), "clips/s", round(d["ms_per_step"]*1000), "us/step;", " ".join(f"{k.replace('_kernel','')}={round(t)}us({f:.2f})" for k,(t,f) in ks.items()))
