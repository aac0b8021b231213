mkdir -p gpurun_out/r5h
timeout 900 python -m pytest tests/test_gpu_cut.py tests/test_gpu_fused_hoi.py tests/test_gpu_golden.py -x -q > gpurun_out/r5h/tests.log 2>&1; tail -4 gpurun_out/r5h/tests.log
B="--no-cpu-baseline --no-optimizer-line --no-native-line --no-roofline --min-seconds 1.2"
for cd in c2:f32s c2:bf16 c3:bf16 pnr:f32s c2:f32; do
  cfg=${cd%%:*}; dt=${cd##*:}
  for mode in "0 0" "0 1" "1 0" "1 1"; do set -- $mode
    EGX_FFN_CUT=$1 EGX_TOKPREP=$2 python bench.py --config $cfg --dtype $dt $B 2>/dev/null | tail -1 > gpurun_out/r5h/${cfg}_${dt}_cut$1_prep$2.json
    python tools/benchline.py gpurun_out/r5h/${cfg}_${dt}_cut$1_prep$2.json "$cfg $dt cut=$1 prep=$2"
  done
done
for dt in f32s bf16; do
  EGX_FFN_CUT=1 bash tools/profile_bench.sh c2 r5h/prof_$dt --dtype $dt --min-seconds 0.6 > gpurun_out/r5h/prof_$dt.txt 2>&1
  echo "== $dt cut"; grep -E "kernel|ms_per_step" gpurun_out/r5h/prof_$dt.txt | head -9 | sed 's/(egx::[A-Za-z]*Params[^)]*)//' 
done
