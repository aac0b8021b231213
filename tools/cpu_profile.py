"""Host-side cost of one training step (cProfile) — development aid."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import hhi_ttm
from egot2_amd.synth import hhi_args
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)      # the backward's python (EncoderFn.backward) runs in this thread: cProfile sees it
m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.5)).to(dev).set_compute(sys.argv[1] if len(sys.argv) > 1 else "bf16").train()
feats = [torch.randn(256, 15, 256, device=dev) for _ in range(3)]
target = torch.randint(0, 2, (256,), device=dev)
w = torch.tensor([0.266, 0.734], device=dev)
params = [p for p in m.parameters()]
def step():
    for p in params:
        p.grad = None
    loss = torch.nn.functional.cross_entropy(m.forward_features(*feats), target, weight=w)
    loss.backward()
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / 200:.0f} us/step, drained after {1e6 * (t2 - t0) / 200:.0f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
