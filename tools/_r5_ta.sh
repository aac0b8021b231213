mkdir -p gpurun_out/ta
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_scale_edges.py -x -q -m gpu 2>&1 | tail -4
B="--no-cpu-baseline --no-optimizer-line --no-native-line --no-roofline --min-seconds 1.0"
for rep in 1 2; do
for cd in "150:f32s" "150:bf16" "60:f32s" "30:f32s"; do
 fr=${cd%%:*}; dt=${cd##*:}
 for v in 0 1; do
  EGX_TILED_PLANES=$v python bench.py --config c2 --frames $fr --dtype $dt $B 2>/dev/null | tail -1 > gpurun_out/ta/t${fr}_${dt}_$v.json
  python tools/benchline.py gpurun_out/ta/t${fr}_${dt}_$v.json "c2 T=$fr $dt planes=$v"
 done
done
done
