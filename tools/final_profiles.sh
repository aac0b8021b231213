#!/bin/bash
# usage (GPU box, repo root): tools/final_profiles.sh [round-tag, default r04]
# Bench lines, rocprofv3 kernel stats and PMC counters of every BASELINE configuration -> gpurun_out/final/ (copy what is to
# be kept into profiles/). Every step checks its exit code: a failed bench never leaves a half-written JSON behind.
set -euo pipefail
R=${1:-r06}
STAGES=${2:-bench,prof,trace,pmc}      # which parts to run (run pmc first and copy its files into profiles/ when the bench lines are to carry the counters)
OUT=gpurun_out/final
mkdir -p $OUT
bench() {   # bench <name> <bench.py args...>
  local name=$1; shift
  # (no retry: round 5 retried a failed line once because the captured collectives could abort in PyTorch's RCCL watchdog thread; since round 6 the
  # captured exchange runs on egx_allreduce, which has no such thread — a failure here is a real one)
  if python bench.py "$@" > $OUT/bench_$name.log 2>&1; then
    tail -1 $OUT/bench_$name.log > $OUT/${R}_bench_$name.json
    python tools/benchline.py $OUT/${R}_bench_$name.json $name
  else
    echo "bench $name FAILED"; tail -5 $OUT/bench_$name.log; return 1
  fi
}
prof() {    # prof <config> <name> <bench.py args...>
  local cfg=$1 name=$2; shift 2
  tools/profile_bench.sh $cfg final/prof_$name "$@" > $OUT/prof_$name.txt 2>&1 || { echo "profile $name FAILED"; tail -5 $OUT/prof_$name.txt; return 1; }
  cp gpurun_out/final/prof_${name}_kernel_stats.csv $OUT/${R}_bench_${name}_kernel_stats.csv
}
if [[ $STAGES == *bench* ]]; then
bench f32s
bench f32 --dtype f32 --no-cpu-baseline
bench bf16 --dtype bf16 --no-cpu-baseline
bench c3 --config c3 --no-cpu-baseline
bench c3_unfused --config c3 --no-fused-ce --no-cpu-baseline --no-roofline      # lossAV as its own two launches (egx_linear_ce_*)
bench c4 --config c4 --no-cpu-baseline --steps 5 --warmup 2
bench c4_sink --config c4 --feat-source sink --feat-dtype bf16 --no-cpu-baseline --steps 5 --warmup 2
bench pnr --config pnr --no-cpu-baseline --no-native-line
bench c2_t30 --frames 30 --no-cpu-baseline --no-native-line
bench c2_t60 --frames 60 --no-cpu-baseline --no-native-line
bench c2_t150 --frames 150 --no-cpu-baseline --no-native-line
bench c2_t150_bf16 --frames 150 --dtype bf16 --no-cpu-baseline
bench c3_t60 --config c3 --frames 60 --no-cpu-baseline
bench c5hoi --config c5hoi --no-cpu-baseline --steps 5 --warmup 2
bench c5hoi_enc --config c5hoi --encoder-only --no-cpu-baseline --steps 5 --warmup 2
bench c5hhi --config c5hhi --no-cpu-baseline --steps 5 --warmup 2
bench c5hhi_enc --config c5hhi --encoder-only --no-cpu-baseline --steps 5 --warmup 2
bench f32s_deterministic --deterministic --no-cpu-baseline --no-roofline --no-native-line
bench f32s_forcedist --force-dist --no-cpu-baseline --no-roofline --no-native-line
bench f32s_graph_forcedist --force-dist --graph-collectives --no-cpu-baseline --no-roofline --no-native-line
bench c4_forcedist --config c4 --force-dist --no-cpu-baseline --no-roofline --steps 5 --warmup 2
bench c5hoi_forcedist --config c5hoi --force-dist --no-cpu-baseline --no-roofline --steps 5 --warmup 2
bench c5hoi_eager --config c5hoi --no-graph --no-cpu-baseline --no-roofline --steps 5 --warmup 2
bench c5hoi_graph_forcedist --config c5hoi --force-dist --graph-collectives --no-cpu-baseline --no-roofline --steps 5 --warmup 2
# cut mode (ffn_cut.hip) against the one-launch kernels, both ways round (default: cut for c2 f32s, one launch elsewhere)
EGX_FFN_CUT=0 bench f32s_onelaunch --no-cpu-baseline --no-native-line
EGX_FFN_CUT=1 bench bf16_cut --dtype bf16 --no-cpu-baseline
EGX_FFN_CUT=1 bench c3_cut --config c3 --no-cpu-baseline
EGX_DEC_GROUP=0 bench c5hhi_ungrouped --config c5hhi --no-cpu-baseline --no-roofline --steps 5 --warmup 2
bench c4_graph_forcedist --config c4 --force-dist --graph-collectives --no-cpu-baseline --no-roofline --steps 5 --warmup 2
# small batches (sliced mode of the per-clip kernels): the reference sampler's own batch, and the per-GPU share of a strong-scaled B = 256 on 8 GPUs
bench c2_b26 --batch 26 --no-cpu-baseline --no-roofline --no-native-line
bench c2_b32 --batch 32 --no-cpu-baseline --no-roofline --no-native-line
bench c2_b32_forcedist --batch 32 --force-dist --no-cpu-baseline --no-roofline --no-native-line
EGX_FFN_SLICES=1 bench c2_b32_unsliced --batch 32 --no-cpu-baseline --no-roofline --no-native-line
# EgoT2-g HHI on long sequences (wide path with the online-softmax attention, decoder onto a 450-token memory)
bench c5hhi_t60 --config c5hhi --frames 60 --batch 64 --no-cpu-baseline --no-roofline --steps 5 --warmup 2
bench c5hhi_t150 --config c5hhi --frames 150 --batch 25 --no-cpu-baseline --no-roofline --steps 5 --warmup 2
bench c5hhi_t150_enc --config c5hhi --frames 150 --batch 25 --encoder-only --no-cpu-baseline --no-roofline --steps 5 --warmup 2
fi
if [[ $STAGES == *prof* ]]; then
prof c2 f32s
prof c2 f32 --dtype f32
prof c2 bf16 --dtype bf16
prof c3 c3
prof c4 c4 --steps 5 --warmup 2
prof c5hoi c5hoi --steps 5 --warmup 2
prof c5hhi c5hhi --steps 5 --warmup 2
prof c2 c2_t150 --frames 150
prof pnr pnr
fi
if [[ $STAGES == *trace* ]]; then      # one step's kernels in launch order (eager launches)
for c in c2 c3 c4 c5hoi c5hhi; do
  python3 tools/step_trace.py $c -- --config $c > /dev/null 2>&1 && cp gpurun_out/steptrace_$c.txt $OUT/${R}_steptrace_$c.txt
done
python3 tools/step_trace.py c2_t150 -- --config c2 --frames 150 > /dev/null 2>&1 && cp gpurun_out/steptrace_c2_t150.txt $OUT/${R}_steptrace_c2_t150.txt
fi
if [[ $STAGES == *pmc* ]]; then
python3 tools/pmc_collect.py f32s > $OUT/pmc_f32s.txt 2>&1 && cp gpurun_out/pmc_f32s.json $OUT/${R}_pmc_c2_f32s.json
python3 tools/pmc_collect.py bf16 -- --dtype bf16 > $OUT/pmc_bf16.txt 2>&1 && cp gpurun_out/pmc_bf16.json $OUT/${R}_pmc_c2_bf16.json
python3 tools/pmc_collect.py c3 -- --config c3 > $OUT/pmc_c3.txt 2>&1 && cp gpurun_out/pmc_c3.json $OUT/${R}_pmc_c3_bf16.json
python3 tools/pmc_collect.py c4 -- --config c4 > $OUT/pmc_c4.txt 2>&1 && cp gpurun_out/pmc_c4.json $OUT/${R}_pmc_c4_bf16.json
python3 tools/pmc_collect.py c5hoi -- --config c5hoi > $OUT/pmc_c5hoi.txt 2>&1 && cp gpurun_out/pmc_c5hoi.json $OUT/${R}_pmc_c5hoi_bf16.json
python3 tools/pmc_collect.py c2_t150 -- --frames 150 > $OUT/pmc_c2_t150.txt 2>&1 && cp gpurun_out/pmc_c2_t150.json $OUT/${R}_pmc_c2_t150_f32s.json
python tools/gemm_bench.py 20 > $OUT/${R}_gemm_bench.txt 2>&1
tail -n 4 $OUT/pmc_*.txt
fi
