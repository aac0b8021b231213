"""Phase stamps of small_dw_kernel for one workgroup (needs -DEGX_STAMPS [-DEGX_SSTAMP_BLOCK=n])."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import hhi_ttm, _lib
from egot2_amd.synth import hhi_args
lib = _lib.load()
dev = torch.device("cuda:0")
m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.5)).to(dev).train()
feats = [torch.randn(256, 15, 256, device=dev) for _ in range(3)]
for comp in ("f32s", "bf16"):
    m.set_compute(comp, "fused")
    for _ in range(3):
        for q in m.parameters():
            q.grad = None
        m.forward_features(*feats).sum().backward()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    lib.egx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.egx_debug_stamps(buf, -4000)
    a = list(buf)[:7]
    names = ["tail (slab + partial-row reduction share)", "setup + first loads issued", "first K-block", "remaining K-blocks", "atomics issued", "atomics drained"]
    print(comp, "total", a[6] - a[0], " | ".join(f"{n}={a[i + 1] - a[i]}" for i, n in enumerate(names)))
