import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import hhi_ttm
from egot2_amd.synth import hhi_args
from egot2_amd.train import CrossEntropyLoss, FusedAdam
dev = torch.device("cuda:0")
for mt in (True, False):
    torch.autograd.set_multithreading_enabled(mt)
    for compute in ("f32s", "bf16"):
        m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.5)).to(dev).set_compute(compute).train()
        crit = CrossEntropyLoss(torch.FloatTensor([0.266, 0.734])).to(dev)
        feats = [torch.randn(256, 15, 256, device=dev) for _ in range(3)]
        target = torch.randint(0, 2, (256,), device=dev)
        opt = FusedAdam(m.parameters(), lr=1e-4)
        def eager():
            opt.zero_grad(set_to_none=True)
            loss = crit(m.forward_features(*feats), target)
            loss.backward(); opt.step()
        for _ in range(30): eager()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(400): eager()
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 400
        print(f"autograd multithreading {mt}: {compute} eager loop {te * 1e6:.0f} us/step", flush=True)
