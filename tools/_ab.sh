# usage (GPU box): bash tools/_ab.sh [base-variant] [configs...] — same-box A/B of the product library against
# egot2_amd/_variants/lib_<base-variant>.so (default: head), interleaved twice (development aid)
base=${1:-head}; shift
cfgs=${@:-"c2:f32s c2:bf16 c3:bf16"}
mkdir -p gpurun_out/ab
B="--no-cpu-baseline --no-optimizer-line --no-native-line --no-roofline --min-seconds 1.5"
for rep in 1 2; do
for cd in $cfgs; do
  cfg=${cd%%:*}; dt=${cd##*:}
  for lib in $base new; do
    if [ $lib = new ]; then unset EGX_LIB; else export EGX_LIB=$PWD/egot2_amd/_variants/lib_$lib.so; fi
    python bench.py --config $cfg --dtype $dt $B 2>/dev/null | tail -1 > gpurun_out/ab/${cfg}_${dt}_$lib.json
    python tools/benchline.py gpurun_out/ab/${cfg}_${dt}_$lib.json "$cfg $dt $lib"
  done
done
done
unset EGX_LIB
