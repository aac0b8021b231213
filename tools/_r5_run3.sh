mkdir -p gpurun_out/r5c
for dt in f32s bf16; do
  for cut in 0 1; do
    export EGX_FFN_CUT=$cut
    bash tools/profile_bench.sh c2 r5c/prof_${dt}_cut$cut --dtype $dt --min-seconds 1.0 > gpurun_out/r5c/prof_${dt}_cut$cut.txt 2>&1
    head -14 gpurun_out/r5c/prof_${dt}_cut$cut.txt
  done
done
