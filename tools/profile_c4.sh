#!/bin/bash
# usage (GPU box): tools/profile_c4.sh <dtype>   -> kernel stats of the C4 configuration (HOI LTA 4-task, generic kernels)
dt=${1:-bf16}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
cat > /tmp/c4.py <<PY
import os, sys
sys.path.insert(0, "$root")
import torch
from types import SimpleNamespace as NS
from egot2_amd import hoi_lta
dev = torch.device("cuda:0"); B = 256
cfg = NS(FORECASTING=NS(NUM_INPUT_CLIPS=32, NUM_ACTIONS_TO_PREDICT=20),
         MODEL=NS(TRANSLATION_HEADS=8, TRANSLATION_LAYERS=4, TRANSLATION_INPUT_FEATURES=768, TRANSLATION_DROPOUT=0.1,
                  NUM_CLASSES=[115, 478], DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False))
m = hoi_lta.TaskFusionMFTransformerLTA4Task(cfg).to(dev).set_compute("$dt").train()
feats = [torch.randn(B, 32, 8192, device=dev), torch.randn(B, 32, 8192, device=dev), torch.randn(B, 32, 768, device=dev), torch.randn(B, 32, 2048, device=dev)]
for it in range(4):
    for p in m.parameters(): p.grad = None
    y = m.forward_features(*feats)
    sum(t.float().sum() for t in y).backward()
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4_$dt -o r -- python3 /tmp/c4.py > gpurun_out/prof_c4_$dt.log 2>&1
python3 - "$dt" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_c4_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), ("%.1f" % (float(r["TotalDurationNs"]) / 4 / 1e3)).rjust(10), "us/step", r["Percentage"])
PY
