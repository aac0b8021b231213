mkdir -p gpurun_out/r5d
export EGX_FFN_CUT=1
run() { # name lib
  for dt in f32s bf16; do
    bash tools/profile_bench.sh c2 r5d/prof_$1_$dt --dtype $dt --min-seconds 0.6 > gpurun_out/r5d/prof_$1_$dt.txt 2>&1
    echo "== $1 $dt"; grep -E "ffn_fwd_kernel|ffn_bwd_kernel|fused_fwd_kernel|fused_bwd_kernel|ms_per_step" gpurun_out/r5d/prof_$1_$dt.txt | sed 's/(egx::Fused[A-Za-z]*Params[, int]*)//' 
  done
}
unset EGX_LIB; run product
for v in noload nostore ring8 ring2; do export EGX_LIB=$PWD/egot2_amd/_variants/lib_$v.so; run $v; done
export EGX_LIB=$PWD/egot2_amd/_variants/lib_stamps.so
python tools/stamps_cut.py 2>&1 | grep -v Warn | tail -8
unset EGX_LIB
for n in 0 256 512; do tools/micro/slice_litmus 64 100 3 8 $n 2; tools/micro/slice_litmus_nowait 64 100 3 8 $n 2; done 2>&1 | tee gpurun_out/r5d/litmus.txt
