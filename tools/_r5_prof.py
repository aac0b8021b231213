import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import hhi_ttm
from egot2_amd.synth import hhi_args
from egot2_amd.train import CrossEntropyLoss, FusedAdam
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.5)).to(dev).set_compute("f32s").train()
crit = CrossEntropyLoss(torch.FloatTensor([0.266, 0.734])).to(dev)
feats = [torch.randn(256, 15, 256, device=dev) for _ in range(3)]
target = torch.randint(0, 2, (256,), device=dev)
opt = FusedAdam(m.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(m.forward_features(*feats), target)
    loss.backward(); opt.step()
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"single-threaded autograd: enqueue {1e6*(t1-t0)/200:.0f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
