mkdir -p gpurun_out/r5t
timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_golden.py tests/test_gpu_wide.py tests/test_gpu_train.py -x -q > gpurun_out/r5t/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5t/tests.log; grep -E "passed|failed|rc=|Error" gpurun_out/r5t/tests.log | tail -4
B="--no-cpu-baseline --no-roofline --steps 5 --warmup 2 --min-seconds 1.2"
for rep in 1 2; do for cfg in c5hhi c5hoi c4; do
  python bench.py --config $cfg $B 2>/dev/null | tail -1 > gpurun_out/r5t/$cfg.json
  python - <<PY
import json; d=json.load(open("gpurun_out/r5t/$cfg.json")); print("$cfg", round(d["ms_per_step"],4), "ms", d.get("library_launches_per_step"), "launches")
PY
done; done
python bench.py --config c5hhi --encoder-only $B 2>/dev/null | tail -1 > gpurun_out/r5t/c5hhi_enc.json; python tools/benchline.py gpurun_out/r5t/c5hhi_enc.json c5hhi_enc
