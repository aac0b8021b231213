mkdir -p gpurun_out/r5t
B="--no-cpu-baseline --no-roofline --steps 5 --warmup 2 --min-seconds 1.2"
for rep in 1 2; do for cfg in c5hhi c5hoi c4; do for dfr in 0 1; do
  EGX_ROW_DEFER=$dfr python bench.py --config $cfg $B 2>/dev/null | tail -1 > gpurun_out/r5t/${cfg}_d$dfr.json
  python - <<PY
import json; d=json.load(open("gpurun_out/r5t/${cfg}_d$dfr.json")); print("$cfg row_defer=$dfr", round(d["ms_per_step"],4), "ms")
PY
done; done; done
