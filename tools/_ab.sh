# usage (GPU box): bash tools/_ab.sh  — tests + A/B of the product library against _variants/lib_base.so (development aid)
set -x
mkdir -p gpurun_out/ab
python -m pytest tests/test_gpu_translator.py tests/test_gpu_golden.py tests/test_gpu_train.py tests/test_gpu_dropout_graph.py tests/test_gpu_fused_vs_generic.py tests/test_gpu_scale_edges.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/ab/tests.log
cat gpurun_out/ab/tests.log
B="--no-cpu-baseline --no-optimizer-line --no-native-line --min-seconds 1.0"
for dt in f32s bf16 f32; do
  for lib in base new; do
    if [ $lib = base ]; then export EGX_LIB=$PWD/egot2_amd/_variants/lib_base.so; else unset EGX_LIB; fi
    python bench.py --dtype $dt $B 2>/dev/null | tail -1 > gpurun_out/ab/${dt}_$lib.json
    python tools/benchline.py gpurun_out/ab/${dt}_$lib.json $lib
  done
done
for lib in base new; do
  if [ $lib = base ]; then export EGX_LIB=$PWD/egot2_amd/_variants/lib_base.so; else unset EGX_LIB; fi
  python bench.py --config c3 $B 2>/dev/null | tail -1 > gpurun_out/ab/c3_$lib.json
  python tools/benchline.py gpurun_out/ab/c3_$lib.json $lib
done
unset EGX_LIB
EGX_LIB=$PWD/egot2_amd/_variants/lib_stamps.so python tools/stamps.py 2>&1 | grep -v Warn | tail -6
EGX_LIB=$PWD/egot2_amd/_variants/lib_stamps.so python tools/stamps_bwd.py 2>&1 | grep "p=0.5"
