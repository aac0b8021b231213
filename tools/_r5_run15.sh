mkdir -p gpurun_out/r5s
timeout 900 python -m pytest tests/test_gpu_cut.py tests/test_gpu_sliced.py -x -q -k "not litmus" > gpurun_out/r5s/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5s/tests.log; grep -E "passed|failed|rc=|Error|assert" gpurun_out/r5s/tests.log | tail -6
B="--no-cpu-baseline --no-optimizer-line --no-native-line --no-roofline --min-seconds 1.2"
for rep in 1 2; do
for cd in c2:f32s c2:bf16 c3:bf16 c2:f32; do
  cfg=${cd%%:*}; dt=${cd##*:}
  for a8 in 0 1; do
    EGX_FFN_CUT=1 EGX_ATTN8=$a8 python bench.py --config $cfg --dtype $dt $B 2>/dev/null | tail -1 > gpurun_out/r5s/${cfg}_${dt}_a8$a8.json
    python tools/benchline.py gpurun_out/r5s/${cfg}_${dt}_a8$a8.json "$cfg $dt cut=1 attn8=$a8"
  done
  EGX_FFN_CUT=0 python bench.py --config $cfg --dtype $dt $B 2>/dev/null | tail -1 > gpurun_out/r5s/${cfg}_${dt}_cut0.json
  python tools/benchline.py gpurun_out/r5s/${cfg}_${dt}_cut0.json "$cfg $dt cut=0"
done
done
EGX_FFN_CUT=1 bash tools/profile_bench.sh c2 r5s/prof_f32s --min-seconds 0.6 > gpurun_out/r5s/prof_f32s.txt 2>&1; grep -E "kernel|ms_per_step" gpurun_out/r5s/prof_f32s.txt | head -8 | cut -c1-150
