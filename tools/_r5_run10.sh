mkdir -p gpurun_out/r5l
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_golden.py tests/test_gpu_train.py tests/test_gpu_feature_sink.py -x -q -k "bench_batch or golden or egx_allreduce or rccl or sink or head" > gpurun_out/r5l/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5l/tests.log; grep -E "passed|failed|rc=|Error" gpurun_out/r5l/tests.log | tail -5
EGX_LITMUS_STRICT=1 python -m pytest tests/test_gpu_sliced.py -q -s -k litmus 2>&1 | grep -E "variant|passed|failed" > gpurun_out/r5l/litmus.txt; cat gpurun_out/r5l/litmus.txt
for rides in 1 0; do
  EGX_REDUCE_RIDES=$rides bash tools/profile_bench.sh c2 r5l/prof_rides$rides --min-seconds 0.6 > gpurun_out/r5l/prof_rides$rides.txt 2>&1
  echo "== rides=$rides"; grep -E "kernel|ms_per_step" gpurun_out/r5l/prof_rides$rides.txt | head -11 | sed 's/(egx::[A-Za-z]*Params[^)]*)//' 
done
