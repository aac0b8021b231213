for v in stamps stamps200; do
  echo "== $v"; EGX_LIB=$PWD/egot2_amd/_variants/lib_$v.so python tools/stamps_small_dw.py 2>&1 | grep -E "^f32s|^bf16"
  echo "== $v rides=0"; EGX_REDUCE_RIDES=0 EGX_LIB=$PWD/egot2_amd/_variants/lib_$v.so python tools/stamps_small_dw.py 2>&1 | grep -E "^f32s|^bf16"
done
