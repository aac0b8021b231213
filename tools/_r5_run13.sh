mkdir -p gpurun_out/r5o
timeout 1200 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_golden.py -x -q > gpurun_out/r5o/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5o/tests.log; grep -E "passed|failed|rc=|Error" gpurun_out/r5o/tests.log | tail -4
B="--no-cpu-baseline --no-roofline --steps 5 --warmup 2 --min-seconds 1.2"
for rep in 1 2; do for cfg in c5hhi c5hoi; do for grp in 0 1; do
  EGX_DEC_GROUP=$grp python bench.py --config $cfg $B 2>/dev/null | tail -1 > gpurun_out/r5o/${cfg}_grp$grp.json
  python - <<PY
import json; d=json.load(open("gpurun_out/r5o/${cfg}_grp$grp.json")); print("$cfg group=$grp", round(d["ms_per_step"],4), "ms", d.get("library_launches_per_step"), "launches", d["config"].get("launch"))
PY
done; done; done
