"""Timing of the wide-path bf16 GEMMs at the BASELINE configs[3] shapes through the C ABI (egx_wide_gemm).
usage: python tools/gemm_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [(0, 32768, 768, 768), (0, 32768, 2304, 768), (0, 32768, 2048, 768), (0, 32768, 768, 2048), (0, 32768, 768, 2304),
          (0, 8192, 768, 8192), (2, 768, 768, 32768), (2, 2304, 768, 32768), (2, 2048, 768, 32768), (2, 768, 2048, 32768),
          (2, 768, 8192, 8192), (0, 11520, 256, 256), (0, 12288, 512, 2048), (0, 12288, 2048, 512), (0, 12288, 512, 512),
          (0, 12288, 1536, 512), (0, 4096, 512, 8192), (0, 11520, 768, 256), (0, 11520, 2048, 256), (0, 11520, 256, 2048)]
for layout, M, N, K in shapes:
    if layout == 0:
        A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    else:
        A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
    Cf = torch.empty(M, N, device=dev)
    Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    sc = torch.empty(lib.egx_wide_gemm_scratch(layout, M, N, K), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = lib.egx_wide_gemm(layout, A.data_ptr(), B.data_ptr(), Cf.data_ptr() if layout else None, None if layout else Cb.data_ptr(),
                               M, N, K, None, 0, None, sc.data_ptr(), st)
        assert rc == 0, lib.egx_last_error()
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    # the same product through torch.matmul (hipBLASLt / rocBLAS bf16, fp32 accumulate): a yardstick, not a code path
    def lib_run():
        return (A @ B.t()) if layout == 0 else (A.t() @ B)
    for _ in range(3):
        lib_run()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        lib_run()
    e1.record(); torch.cuda.synchronize()
    us_lib = e0.elapsed_time(e1) / reps * 1e3
    print(f"   torch.matmul {us_lib:8.1f} us  {2.0 * M * N * K / us_lib / 1e6:7.1f} TFLOP/s", end="  | ")
    print(f"{'NT' if layout == 0 else 'TN'} M={M:6d} N={N:5d} K={K:6d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s (incl. memset/reduce launches)")
