"""Correctness of egx_wide_gemm (NT) against torch.matmul on random bf16 operands, several shapes incl. ragged M / N, repeated to
catch staging races; the tile variant follows EGX_WIDE_TILE (1024 = the ping-pong 256 x 256 kernel). usage: python tools/gemm_check.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.manual_seed(0)
bad = 0
for M, N, K in [(256, 256, 64), (256, 256, 128), (512, 256, 192), (300, 260, 256), (1024, 768, 768), (4096, 2304, 768), (32768, 768, 2048),
                (1000, 1000, 1024), (8192, 768, 8192), (777, 516, 320),
                (32768, 768, 768), (32768, 768, 2304), (32700, 768, 768), (16384, 1536, 512), (65536, 384, 256),       # 256 x 192 tiles
                (32768, 2048, 256), (24576 - 100, 1024, 320)]:     # persistent 256 x 256 launches (bf16 output)
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev)
    ref = (A.float() @ B.float().t()) + bias
    sc = torch.empty(lib.egx_wide_gemm_scratch(0, M, N, K), dtype=torch.uint8, device=dev)
    worst = 0.0
    for rep in range(reps):
        Cf = torch.full((M, N), float("nan"), device=dev)
        rc = lib.egx_wide_gemm(0, A.data_ptr(), B.data_ptr(), Cf.data_ptr(), None, M, N, K, bias.data_ptr(), 0, None, sc.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.egx_last_error()
        torch.cuda.synchronize()
        err = (Cf - ref).abs().max().item() / ref.abs().max().item()
        worst = max(worst, err if err == err else 1e9)
    # bf16 output only (ReLU), and fp32 output + fp32 residual + bf16 copy: the other store paths of the epilogue
    Cb = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    rc = lib.egx_wide_gemm(0, A.data_ptr(), B.data_ptr(), None, Cb.data_ptr(), M, N, K, bias.data_ptr(), 1, None, sc.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.egx_last_error()
    e_b = ((Cb.float() - ref.clamp_min(0).bfloat16().float()).abs().max() / ref.abs().max()).item()
    res = torch.randn(M, N, device=dev)
    Cf = torch.full((M, N), float("nan"), device=dev)
    Cb.fill_(float("nan"))
    rc = lib.egx_wide_gemm(0, A.data_ptr(), B.data_ptr(), Cf.data_ptr(), Cb.data_ptr(), M, N, K, bias.data_ptr(), 0, res.data_ptr(), sc.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.egx_last_error()
    e_r = ((Cf - (ref + res)).abs().max() / ref.abs().max()).item()
    e_rb = ((Cb.float() - Cf.bfloat16().float()).abs().max() / ref.abs().max()).item()
    worst2 = max(v if v == v else 1e9 for v in (e_b, e_r, e_rb))
    ok = worst < 2e-3 and worst2 < 1e-2
    bad += not ok
    print(f"M={M:6d} N={N:5d} K={K:5d}  max rel err {worst:.2e}  bf16-out {e_b:.1e} residual {e_r:.1e} / {e_rb:.1e}  {'ok' if ok else 'FAIL'}")
sys.exit(1 if bad else 0)
