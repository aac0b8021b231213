mkdir -p gpurun_out/r5k
python -m pytest tests -m gpu -x -q > gpurun_out/r5k/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5k/tests.log; tail -4 gpurun_out/r5k/tests.log
EGX_LITMUS_STRICT=1 python -m pytest tests/test_gpu_sliced.py -q -s -k litmus 2>&1 | grep -E "variant|passed|failed" > gpurun_out/r5k/litmus.txt; cat gpurun_out/r5k/litmus.txt
python bench.py > gpurun_out/r5k/bench.log 2>&1; tail -1 gpurun_out/r5k/bench.log > gpurun_out/r5k/bench_f32s.json; python tools/benchline.py gpurun_out/r5k/bench_f32s.json f32s
python bench.py --force-dist --no-cpu-baseline --no-roofline --no-native-line > gpurun_out/r5k/bench_fd.log 2>&1; tail -1 gpurun_out/r5k/bench_fd.log > gpurun_out/r5k/bench_forcedist.json; python tools/benchline.py gpurun_out/r5k/bench_forcedist.json forcedist
python bench.py --force-dist --graph-collectives --no-cpu-baseline --no-roofline --no-native-line > gpurun_out/r5k/bench_fdg.log 2>&1; tail -1 gpurun_out/r5k/bench_fdg.log > gpurun_out/r5k/bench_forcedist_graph.json; python tools/benchline.py gpurun_out/r5k/bench_forcedist_graph.json forcedist-graph
python - <<'PY'
import json
for f in ("bench_f32s","bench_forcedist","bench_forcedist_graph"):
    d=json.load(open(f"gpurun_out/r5k/{f}.json"))
    print(f, {k:d.get(k) for k in ("ms_per_step","rccl_ranks","exposed_collective_us","ms_per_step_without_exchange","library_launches_per_step")}, (d.get("dropout0") or {}).get("ms_per_step"), (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"))
PY
