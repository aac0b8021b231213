"""Phase stamps of the forward clip kernel (block 0) for a small batch, one workgroup per clip vs sliced
(GPU box; EGX_LIB=egot2_amd/_variants/lib_stamps.so built with -DEGX_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egot2_amd import hhi_ttm, _lib
import egot2_amd.functional as _F_tuning; _F_tuning.reload_tuning_each_call = True   # the switches below are flipped inside this process
from egot2_amd.synth import hhi_args
lib = _lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.5)).to(dev).train()
feats = [torch.randn(B, 15, 256, device=dev) for _ in range(3)]
names = ["zero", "proj", "ln0", "qkv", "attn", "outproj", "ln1", "ffn", "part-store", "ln2"]
lib.egx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for comp in ("f32s", "bf16"):
    for s in (1, 2, 4, 8):
        os.environ["EGX_FFN_SLICES"] = str(s)
        m.set_compute(comp, "fused")
        with torch.no_grad():
            for _ in range(3):
                m.forward_features(*feats)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 32)()
        lib.egx_debug_stamps(buf, 32)
        t = list(buf)[:10]
        print(comp, "slices", s, "total", t[9] - t[0], " ".join(f"{n}={t[i + 1] - t[i]}" for i, n in enumerate(names[:9])))
