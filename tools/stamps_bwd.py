"""Phase breakdown of fused_bwd_kernel (needs a build with EGX_CXXFLAGS=-DEGX_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import hhi_ttm, _lib
from egot2_amd.synth import hhi_args
lib = _lib.load()
dev = torch.device("cuda:0")
names = ["load+LN2bwd", "colsum+LN1fwd", "FFN dX", "xwave-sum", "LN1bwd", "colsum+outproj", "x_in", "QKV", "attn", "inb+inproj", "tokprep"]
for comp in ("f32", "f32s", "bf16"):
    for p in (0.0, 0.5):
        m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=p)).to(dev).set_compute(comp, "fused").train()
        if p == 0.0:
            m.pos_embed.dropout.p = 0.0
        feats = [torch.randn(256, 15, 256, device=dev) for _ in range(3)]
        for _ in range(3):
            for q in m.parameters():
                q.grad = None
            m.forward_features(*feats).sum().backward()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 32)()
        lib.egx_debug_stamps(buf, -32)
        t = list(buf)[:12]
        print(comp, "p=%.1f" % p, "total", t[11] - t[0], " ".join(f"{n}={t[i + 1] - t[i]}" for i, n in enumerate(names)))
