set -x
mkdir -p gpurun_out/p1
EGX_LIB=$PWD/egot2_amd/_variants/lib_stamps.so python tools/stamps.py > gpurun_out/p1/stamps_fwd.txt 2>&1
EGX_LIB=$PWD/egot2_amd/_variants/lib_stamps.so python tools/stamps_bwd.py > gpurun_out/p1/stamps_bwd.txt 2>&1
tools/pmc_sq_cfg.sh c2 --dtype bf16 > gpurun_out/p1/sq_c2_bf16.txt 2>&1
tools/pmc_sq_cfg.sh c3 > gpurun_out/p1/sq_c3.txt 2>&1
tools/profile_bench.sh c5hoi p1/prof_c5hoi --steps 5 --warmup 2 --trials 2 > gpurun_out/p1/prof_c5hoi.txt 2>&1
tools/profile_bench.sh c5hhi p1/prof_c5hhi --steps 5 --warmup 2 --trials 2 > gpurun_out/p1/prof_c5hhi.txt 2>&1
python tools/gemm_bench.py 20 > gpurun_out/p1/gemm_bench.txt 2>&1
python bench.py --config c3 --no-cpu-baseline --min-seconds 0.5 > gpurun_out/p1/bench_c3.log 2>&1
python bench.py --dtype bf16 --no-cpu-baseline --min-seconds 0.5 > gpurun_out/p1/bench_bf16.log 2>&1
cat gpurun_out/p1/stamps_fwd.txt gpurun_out/p1/stamps_bwd.txt
