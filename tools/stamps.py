"""Phase breakdown of the fused forward kernel (needs a build with EGX_CXXFLAGS=-DEGX_STAMPS)."""
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
names = ["zero", "proj", "ln0", "qkv", "attn", "outproj", "ln1", "ffn", "part-store", "ln2"]
for comp in ("f32", "f32s", "bf16"):
    m.set_compute(comp, "fused")
    with torch.no_grad():
        for _ in range(3):
            m.forward_features(*feats)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    lib.egx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.egx_debug_stamps(buf, 32)
    t = list(buf)[:10]
    print(comp, "total ticks", t[9] - t[0], " ".join(f"{n}={t[i + 1] - t[i]}" for i, n in enumerate(names[:9])))
    print("   ffn loop (wave 0): wait_w1 %d gemm1 %d wait_w2 %d epilogue %d gemm2 %d" % tuple(list(buf)[16:21]))
    b = list(buf)
    print("   token prep (wave 0): lds zero + barrier %d | steps: setup %d (of it before the first step: %d) first-fragment wait %d mfma + refill %d" % (b[10] - b[0], b[21], b[11], b[22], b[23]))
