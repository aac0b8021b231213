set -x
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_cut.py tests/test_gpu_feature_sink.py tests/test_gpu_sliced.py -x -q -s > gpurun_out/r5b/tests_new.log 2>&1; echo "rc=$?" >> gpurun_out/r5b/tests_new.log
tail -15 gpurun_out/r5b/tests_new.log
for i in 1 2 3; do tools/micro/slice_litmus 128 100 5 8; tools/micro/slice_litmus_nowait 128 100 5 8; tools/micro/slice_litmus_nowait 128 100 5 64; done > gpurun_out/r5b/litmus.txt 2>&1
cat gpurun_out/r5b/litmus.txt
B="--no-cpu-baseline --no-optimizer-line --no-native-line --no-roofline --min-seconds 1.5"
for rep in 1 2; do
for cd in c2:f32s c2:bf16 c3:bf16 c2:f32; do
  cfg=${cd%%:*}; dt=${cd##*:}
  for cut in 0 1; do
    EGX_FFN_CUT=$cut python bench.py --config $cfg --dtype $dt $B 2>/dev/null | tail -1 > gpurun_out/r5b/${cfg}_${dt}_cut$cut.json
    python tools/benchline.py gpurun_out/r5b/${cfg}_${dt}_cut$cut.json "$cfg $dt cut=$cut"
  done
done
done
