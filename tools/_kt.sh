for lib in head new head new; do
  if [ $lib = new ]; then unset EGX_LIB; else export EGX_LIB=$PWD/egot2_amd/_variants/lib_$lib.so; fi
  for dt in f32s bf16; do
  python bench.py --config c2 --dtype $dt --no-cpu-baseline --no-optimizer-line --no-native-line --min-seconds 1 2>/dev/null | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$lib $dt', round(j['ms_per_step']*1e3,1), 'bwd', round(r['avg_launch_us'],1), {k:round(v['avg_launch_us'],1) for k,v in r['other_kernels'].items()})"
  done
done
