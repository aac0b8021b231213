mkdir -p gpurun_out/r5g
B="--no-cpu-baseline --no-optimizer-line --no-native-line --no-roofline --min-seconds 1.5"
for rep in 1 2; do
for cd in c2:f32s c2:f32 c2:bf16 c3:bf16 c3:f32s pnr:f32s; do
  cfg=${cd%%:*}; dt=${cd##*:}
  for cut in 0 1; do
    EGX_FFN_CUT=$cut python bench.py --config $cfg --dtype $dt $B 2>/dev/null | tail -1 > gpurun_out/r5g/${cfg}_${dt}_cut$cut.json
    python tools/benchline.py gpurun_out/r5g/${cfg}_${dt}_cut$cut.json "$cfg $dt cut=$cut"
  done
done
done
