"""Phase breakdown of the hidden loop of ffn_fwd_kernel (cut mode; needs a build with -DEGX_STAMPS): cycle sums over the loop for waves 0
and 4 (the two waves of SIMD 0) of workgroup 0."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import hhi_ttm, _lib
from egot2_amd.synth import hhi_args
lib = _lib.load()
dev = torch.device("cuda:0")
p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=p_drop)).to(dev).train()
feats = [torch.randn(256, 15, 256, device=dev) for _ in range(3)]
names = ["loop-top", "wait W1[0]", "gemm1", "epilogue", "wait W2[0]", "gemm2"]
for comp in ("f32", "f32s", "bf16"):
    m.set_compute(comp, "fused")
    with torch.no_grad():
        for _ in range(3):
            m.forward_features(*feats)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    lib.egx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.egx_debug_stamps(buf, -3000)
    b = list(buf)
    for w in (0, 1):
        v = b[8 * w:8 * w + 6]
        print(comp, f"fwd loop wave {4 * w}: total {sum(v)}  gemm2(all but last)={v[0]} waitW1={v[1]} gemm1={v[2]} epilogue={v[3]} waitW2={v[4]} gemm2(last)={v[5]}")
    a = b[16:24]
    print(comp, "ffn_fwd_kernel", a[7] - a[0], " ".join(f"{n}={a[i + 1] - a[i]}" for i, n in enumerate(["staging", "loop", "wait-others", "reduce", "sum", "LN2", "out+head"])))
if len(sys.argv) > 2:
    sys.exit(0)
for comp in ("f32s", "bf16"):
    m.set_compute(comp, "fused")
    for _ in range(2):
        for q_ in m.parameters():
            q_.grad = None
        m.forward_features(*feats).sum().backward()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    lib.egx_debug_stamps(buf, -3000)
    a = list(buf)[24:32]
    print(comp, "ffn_bwd_kernel", a[7] - a[0], " ".join(f"{n}={a[i + 1] - a[i]}" for i, n in enumerate(["staging", "head", "LN2bwd+planes", "loop", "wait-others", "reduce", "dy1"])))
