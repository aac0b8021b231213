set -x
mkdir -p gpurun_out/final
# bench lines (default = f32s headline incl. native_f32 and cpu baseline), other dtypes / configs
python bench.py > gpurun_out/final/bench_f32s.log 2>&1; tail -1 gpurun_out/final/bench_f32s.log > gpurun_out/final/r02_bench_f32s.json
python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/final/bench_f32.log 2>&1; tail -1 gpurun_out/final/bench_f32.log > gpurun_out/final/r02_bench_f32.json
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/final/bench_bf16.log 2>&1; tail -1 gpurun_out/final/bench_bf16.log > gpurun_out/final/r02_bench_bf16.json
python bench.py --config c3 --no-cpu-baseline > gpurun_out/final/bench_c3.log 2>&1; tail -1 gpurun_out/final/bench_c3.log > gpurun_out/final/r02_bench_c3.json
python bench.py --config c4 --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/final/bench_c4.log 2>&1; tail -1 gpurun_out/final/bench_c4.log > gpurun_out/final/r02_bench_c4.json
python bench.py --config c5hoi --encoder-only --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/final/bench_c5hoi.log 2>&1; tail -1 gpurun_out/final/bench_c5hoi.log > gpurun_out/final/r02_bench_c5hoi_enc.json
python bench.py --config c5hhi --encoder-only --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/final/bench_c5hhi.log 2>&1; tail -1 gpurun_out/final/bench_c5hhi.log > gpurun_out/final/r02_bench_c5hhi_enc.json
python bench.py --deterministic --no-cpu-baseline --no-roofline --no-native-line > gpurun_out/final/bench_det.log 2>&1; tail -1 gpurun_out/final/bench_det.log > gpurun_out/final/r02_bench_f32s_deterministic.json
# rocprof kernel stats
tools/profile_bench.sh c2 final/prof_f32s > gpurun_out/final/prof_f32s.txt 2>&1
tools/profile_bench.sh c2 final/prof_f32 --dtype f32 > gpurun_out/final/prof_f32.txt 2>&1
tools/profile_bench.sh c2 final/prof_bf16 --dtype bf16 > gpurun_out/final/prof_bf16.txt 2>&1
tools/profile_bench.sh c3 final/prof_c3 > gpurun_out/final/prof_c3.txt 2>&1
tools/profile_bench.sh c4 final/prof_c4 --steps 5 --warmup 2 > gpurun_out/final/prof_c4.txt 2>&1
tools/profile_bench.sh c5hoi final/prof_c5hoi --encoder-only --steps 5 --warmup 2 > gpurun_out/final/prof_c5hoi.txt 2>&1
tools/profile_bench.sh c5hhi final/prof_c5hhi --encoder-only --steps 5 --warmup 2 > gpurun_out/final/prof_c5hhi.txt 2>&1
for f in gpurun_out/final/r02_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['dtype'][:12], round(d['ms_per_step'],4), round(d['value']), (d.get('roofline') or {}).get('frac'))"; done
