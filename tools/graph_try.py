"""Does the whole training step capture into a HIP graph, and what does replay cost? (development aid)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egot2_amd import hhi_ttm
from tests.util import hhi_args
dev = torch.device("cuda:0")
for comp in ("bf16", "f32"):
    m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.0)).to(dev).set_compute(comp).train()
    m.pos_embed.dropout.p = 0.0
    feats = [torch.randn(256, 15, 256, device=dev) for _ in range(3)]
    target = torch.randint(0, 2, (256,), device=dev)
    w = torch.tensor([0.266, 0.734], device=dev)
    params = [p for p in m.parameters()]
    def step():
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(m.forward_features(*feats), target, weight=w)
        loss.backward()
        return loss
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step()
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    err = max((p.grad - ref[k]).abs().max().item() for k, p in m.named_parameters())
    print(comp, f"graph replay {dt * 1e6:.0f} us/step -> {256 / dt:.0f} clips/s; max grad diff vs eager {err:.2e}; loss {loss.item():.4f}")
