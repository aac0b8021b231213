mkdir -p gpurun_out/r5e
export EGX_FFN_CUT=1
timeout 600 python -m pytest tests/test_gpu_cut.py -x -q > gpurun_out/r5e/tests_cut.log 2>&1; tail -2 gpurun_out/r5e/tests_cut.log
for dt in f32s bf16; do
  bash tools/profile_bench.sh c2 r5e/prof_$dt --dtype $dt --min-seconds 0.6 > gpurun_out/r5e/prof_$dt.txt 2>&1
  echo "== $dt"; grep -E "ffn_fwd_kernel|ffn_bwd_kernel|fused_fwd_kernel|fused_bwd_kernel|ms_per_step" gpurun_out/r5e/prof_$dt.txt | sed 's/(egx::Fused[A-Za-z]*Params[, int]*)//' 
done
export EGX_LIB=$PWD/egot2_amd/_variants/lib_stamps.so
python tools/stamps_cut.py 2>&1 | grep -v Warn | tail -13
