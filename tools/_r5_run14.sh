mkdir -p gpurun_out/r5p
for grp in 0 1; do
EGX_DEC_GROUP=$grp bash tools/profile_bench.sh c5hhi r5p/prof_c5hhi_$grp --steps 5 --warmup 2 --min-seconds 0.6 > gpurun_out/r5p/prof_c5hhi_$grp.txt 2>&1
echo "== c5hhi group=$grp"; grep -E "kernel|ms_per_step" gpurun_out/r5p/prof_c5hhi_$grp.txt | head -16 | cut -c1-160
done
B="--no-cpu-baseline --no-roofline --steps 5 --warmup 2 --min-seconds 1.2"
for cfg in c5hhi c5hoi; do python bench.py --config $cfg $B 2>/dev/null | tail -1 > gpurun_out/r5p/$cfg.json; python tools/benchline.py gpurun_out/r5p/$cfg.json $cfg; done
python bench.py --config c5hoi --graph $B 2>/dev/null | tail -1 > gpurun_out/r5p/c5hoi_graph.json; python tools/benchline.py gpurun_out/r5p/c5hoi_graph.json c5hoi-graph
