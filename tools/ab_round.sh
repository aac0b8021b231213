# usage (GPU box): bash tools/ab_round.sh — same-box, interleaved: the round-5 library (egot2_amd/_variants/lib_r5.so, built from commit af434af's csrc;
# bench.py --no-fused-ce --no-weight-cache = the round-5 step: packing launch and weighted_ce launch inside the step) against the product library, and the cut
# mode against the one-launch kernels on the final tree (development aid; results: profiles/r06_ab_round.txt)
B="--no-cpu-baseline --no-optimizer-line --no-native-line --no-roofline"
for rep in 1 2 3; do
  for cd in c2:f32s c2:bf16 c3:bf16; do
    cfg=${cd%%:*}; dt=${cd##*:}
    echo -n "$cfg $dt r5 : "; EGX_LIB=$PWD/egot2_amd/_variants/lib_r5.so EGX_LIB_UNSAFE=1 python bench.py --config $cfg --dtype $dt $B --no-fused-ce --no-weight-cache 2>/dev/null | tail -1 | python tools/benchline.py
    echo -n "$cfg $dt r6 : "; python bench.py --config $cfg --dtype $dt $B 2>/dev/null | tail -1 | python tools/benchline.py
  done
  echo -n "c2 f32s r6 one-launch (EGX_FFN_CUT=0): "; EGX_FFN_CUT=0 python bench.py $B 2>/dev/null | tail -1 | python tools/benchline.py
  echo -n "c2 f32s r6 cut        (EGX_FFN_CUT=1): "; EGX_FFN_CUT=1 python bench.py $B 2>/dev/null | tail -1 | python tools/benchline.py
done
