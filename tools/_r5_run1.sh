set -x
mkdir -p gpurun_out/r5a
python -m pytest tests -m gpu -x -q > gpurun_out/r5a/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5a/tests.log
tail -3 gpurun_out/r5a/tests.log
python bench.py > gpurun_out/r5a/bench_f32s.log 2>&1; tail -1 gpurun_out/r5a/bench_f32s.log > gpurun_out/r5a/bench_f32s.json
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r5a/bench_bf16.log 2>&1; tail -1 gpurun_out/r5a/bench_bf16.log > gpurun_out/r5a/bench_bf16.json
python bench.py --config c3 --no-cpu-baseline > gpurun_out/r5a/bench_c3.log 2>&1; tail -1 gpurun_out/r5a/bench_c3.log > gpurun_out/r5a/bench_c3.json
EGX_LIB=$PWD/egot2_amd/_variants/lib_stamps.so python tools/stamps.py > gpurun_out/r5a/stamps_fwd.txt 2>&1
EGX_LIB=$PWD/egot2_amd/_variants/lib_stamps.so python tools/stamps_bwd.py > gpurun_out/r5a/stamps_bwd.txt 2>&1
cat gpurun_out/r5a/stamps_fwd.txt gpurun_out/r5a/stamps_bwd.txt
for f in f32s bf16 c3; do python tools/benchline.py gpurun_out/r5a/bench_$f.json $f; done
