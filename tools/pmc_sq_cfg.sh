#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sq_cfg.sh <config> [bench args]   — SQ counters per kernel for any bench config
cfg=${1:-c4}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/sqc_${cfg}_$i -o r -- python3 bench.py --config $cfg --no-graph --no-cpu-baseline --no-roofline --no-optimizer-line --no-native-line --steps 2 --warmup 1 --trials 1 "$@" > gpurun_out/sqc_${cfg}_$i.log 2>&1
done
python3 - "$cfg" <<'PY'
import csv, glob, sys, collections
cfg = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for i in (1, 2):
    files = glob.glob(f"gpurun_out/sqc_{cfg}_{i}/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no counters for pass", i); print(open(f"gpurun_out/sqc_{cfg}_{i}.log").read()[-1500:]); continue
    for row in csv.DictReader(open(files[0])):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("egx::"): continue
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(name, i)].add(row["Dispatch_Id"])
P1 = "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS".split()
for name, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    n1 = max(len(cnt[(name, 1)]), 1); n2 = max(len(cnt[(name, 2)]), 1)
    o = {k: v / (n1 if k in P1 else n2) for k, v in d.items()}
    wc = o.get("SQ_WAVE_CYCLES", 1)
    print(name, "launches", n1)
    print("   wave_cycles %.3g busy_cycles %.3g wait_any %.0f%% wait_inst_any %.0f%% (lds %.0f%%) active_any %.0f%% valu %.0f%% lds %.0f%%" % (
        wc, o.get("SQ_BUSY_CYCLES", 0), 100 * o.get("SQ_WAIT_ANY", 0) / wc, 100 * o.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * o.get("SQ_WAIT_INST_LDS", 0) / wc,
        100 * o.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * o.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * o.get("SQ_ACTIVE_INST_LDS", 0) / wc))
    print("   mfma_busy %.3g insts: mfma %.3g valu %.3g lds %.3g vmem_rd %.3g vmem_wr %.3g lds_conflict %.3g" % (
        o.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), o.get("SQ_INSTS_MFMA", 0), o.get("SQ_INSTS_VALU", 0), o.get("SQ_INSTS_LDS", 0), o.get("SQ_INSTS_VMEM_RD", 0), o.get("SQ_INSTS_VMEM_WR", 0), o.get("SQ_LDS_BANK_CONFLICT", 0)))
PY
