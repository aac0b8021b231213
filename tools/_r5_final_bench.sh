bash tools/final_profiles.sh r05 bench,prof,trace > gpurun_out/final_bench.log 2>&1; tail -60 gpurun_out/final_bench.log
