set -e
bash tools/final_profiles.sh r05 pmc > gpurun_out/final_pmc.log 2>&1 || { tail -20 gpurun_out/final_pmc.log; exit 1; }
mkdir -p profiles
cp gpurun_out/final/r05_pmc_*.json profiles/
bash tools/final_profiles.sh r05 bench,prof,trace > gpurun_out/final_bench.log 2>&1 || { tail -20 gpurun_out/final_bench.log; exit 1; }
tail -50 gpurun_out/final_bench.log
