#!/bin/bash
# usage (GPU box): tools/bench_variants.sh "<bench args>" name1 name2 ...   — times egot2_amd/_variants/lib_<name>.so in turn
args=$1; shift
cp egot2_amd/libegot2x.so /tmp/lib_keep.so
for n in "$@"; do
  cp egot2_amd/_variants/lib_$n.so egot2_amd/libegot2x.so
  python bench.py $args --no-cpu-baseline --no-optimizer-line --no-native-line 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print('$n', d['dtype'][:5], round(d['ms_per_step'], 4), round(r.get('avg_launch_us') or 0, 1), {k: round(v['avg_launch_us'], 1) for k, v in (r.get('other_kernels') or {}).items()})
"
done
cp /tmp/lib_keep.so egot2_amd/libegot2x.so
