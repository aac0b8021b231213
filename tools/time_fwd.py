"""Device-side timing of the forward paths (eval) — development aid, not the bench contract."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import hhi_ttm
from egot2_amd.synth import hhi_args

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args()).to(dev).eval()
feats = [torch.randn(B, 15, 256, device=dev) for _ in range(3)]
for comp in ("f32", "bf16"):
    for impl in ("generic", "fused"):
        m.set_compute(comp, impl)
        with torch.no_grad():
            for _ in range(5):
                m.forward_features(*feats)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                m.forward_features(*feats)
            e1.record()
            torch.cuda.synchronize()
        print(f"{comp:5s} {impl:8s} fwd {e0.elapsed_time(e1) / 50 * 1e3:9.1f} us/batch  (B={B})", flush=True)
