# usage (GPU box): bash tools/_abmany.sh "<variant names, 'new' = the product library>" "<cfg:dtype[:extra bench args] ...>" [reps] — interleaved same-box runs
vars=$1; cfgs=$2; reps=${3:-3}
B="--no-cpu-baseline --no-optimizer-line --no-native-line --no-roofline --min-seconds 1.2"
for rep in $(seq $reps); do for cd in $cfgs; do
  cfg=$(echo $cd | cut -d: -f1); dt=$(echo $cd | cut -d: -f2); extra=$(echo $cd | cut -d: -f3- | tr ':' ' ')
  for v in $vars; do
    if [ $v = new ]; then unset EGX_LIB; else export EGX_LIB=$PWD/egot2_amd/_variants/lib_$v.so; fi
    python bench.py --config $cfg --dtype $dt $extra $B 2>/dev/null | tail -1 | python tools/benchline.py /dev/stdin "$cfg $dt $extra $v"
  done
done; done
unset EGX_LIB
