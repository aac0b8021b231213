mkdir -p gpurun_out/r5n
bash tools/profile_bench.sh c2 r5n/prof_det --deterministic --min-seconds 0.6 > gpurun_out/r5n/prof_det.txt 2>&1
grep -E "kernel|ms_per_step" gpurun_out/r5n/prof_det.txt | head -12 | sed 's/(egx::[A-Za-z]*Params[^)]*)//'
