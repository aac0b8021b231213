import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from egot2_amd import hhi_asd, _lib
from egot2_amd.synth import hhi_args
lib = _lib.load()
dev = torch.device("cuda:0")
m = hhi_asd.TaskFusionMFTransformer3Task(hhi_args(hidden_dim=128, num_heads=4, dropout=0.1, num_layers=2)).to(dev).set_compute("bf16").train()
m.enable_weight_cache()
head = hhi_asd.lossAV(128).to(dev)
feats = [torch.randn(256, 15, 256, device=dev) for _ in range(3)]
y = torch.randint(0, 2, (256 * 15,), device=dev)
lib.egx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for fused in (False, True, False, True):
    with torch.no_grad():
        for _ in range(3):
            if fused: m.forward_features(*feats, lossav=head, labels=y)
            else: m.forward_features(*feats)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    lib.egx_debug_stamps(buf, 32)
    t = list(buf)[:12]
    print("fused" if fused else "plain", "t9-t0", t[9] - t[0], "deltas", [t[i + 1] - t[i] for i in range(9)], "t10-t0", t[10] - t[0], "tail: 9->11", list(buf)[11] - t[9], "11->12", list(buf)[12] - list(buf)[11], "12->13", list(buf)[13] - list(buf)[12], "13->14", list(buf)[14] - list(buf)[13])
