#!/bin/bash
# small-batch step times, one workgroup per clip (EGX_FFN_SLICES=1) vs the sliced default (GPU box, repo root) -> stdout
B="--no-cpu-baseline --no-roofline --no-optimizer-line --no-native-line --min-seconds 1.0"
for cfg in "c2 f32s" "c2 bf16" "c3 bf16"; do
  set -- $cfg
  for b in 26 32 64 128; do
    for s in 1 auto; do
      if [ $s = auto ]; then unset EGX_FFN_SLICES; else export EGX_FFN_SLICES=$s; fi
      line=$(timeout 120 python bench.py --config $1 --dtype $2 --batch $b $B 2>/dev/null < /dev/null | tail -1)
      echo "$1 $2 B=$b slices=$s $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us/step %.0f clips/s' % (d['ms_per_step']*1e3, d['value']))")"
    done
  done
done
unset EGX_FFN_SLICES
