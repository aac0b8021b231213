mkdir -p gpurun_out/r5i
timeout 600 python -m pytest tests/test_gpu_cut.py -x -q -k "token_prep" > gpurun_out/r5i/tests.log 2>&1; tail -2 gpurun_out/r5i/tests.log
for dt in f32s bf16; do for cut in 0 1; do
  EGX_FFN_CUT=$cut bash tools/profile_bench.sh c2 r5i/prof_${dt}_$cut --dtype $dt --min-seconds 0.6 > gpurun_out/r5i/prof_${dt}_$cut.txt 2>&1
  echo "== $dt cut=$cut"; grep -E "kernel|ms_per_step" gpurun_out/r5i/prof_${dt}_$cut.txt | head -10 | sed 's/(egx::[A-Za-z]*Params[^)]*)//' 
done; done
EGX_TOKPREP=0 EGX_FFN_CUT=0 python bench.py --no-cpu-baseline --no-optimizer-line --no-native-line --no-roofline --min-seconds 1.2 2>/dev/null | tail -1 > gpurun_out/r5i/base.json; python tools/benchline.py gpurun_out/r5i/base.json "c2 f32s base"
EGX_TOKPREP=0 EGX_FFN_CUT=0 python bench.py --dtype bf16 --no-cpu-baseline --no-optimizer-line --no-native-line --no-roofline --min-seconds 1.2 2>/dev/null | tail -1 > gpurun_out/r5i/base16.json; python tools/benchline.py gpurun_out/r5i/base16.json "c2 bf16 base"
