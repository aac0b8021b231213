"""Phase stamps of the ping-pong NT GEMM (workgroup 0, wave 0; needs a build with -DEGX_STAMPS): prologue, K loop, drain, epilogue issue, store drain."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
lib.egx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for M, N, K in [(32768, 2048, 768), (32768, 768, 768), (32768, 768, 2048), (32768, 2304, 768)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    sc = torch.empty(max(lib.egx_wide_gemm_scratch(0, M, N, K), 256), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        assert lib.egx_wide_gemm(0, A.data_ptr(), B.data_ptr(), None, Cb.data_ptr(), M, N, K, None, 0, None, sc.data_ptr(), st) == 0
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.egx_debug_stamps(buf, -1000)
    t = list(buf)
    names = ["prologue", "K loop", "drain", "epilogue issue", "store drain"]
    print(f"M={M} N={N} K={K} ({K // 64} K tiles): total {t[5] - t[0]} ticks; " + " ".join(f"{n}={t[i + 1] - t[i]}" for i, n in enumerate(names)))
