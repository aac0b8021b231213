"""Development probe: does a wide-path fwd+bwd capture into a hipGraph? usage: python tools/micro/wide_capture_probe.py p devseed B n"""
import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
import torch
from types import SimpleNamespace as NS
from egot2_amd import hoi_lta
from tests.util import seeded_feats, seeded_state_dict
p, dev, B, n = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cuda = torch.device("cuda:0")
cfg = NS(FORECASTING=NS(NUM_INPUT_CLIPS=n, NUM_ACTIONS_TO_PREDICT=3),
         MODEL=NS(TRANSLATION_HEADS=8, TRANSLATION_LAYERS=2, TRANSLATION_INPUT_FEATURES=256, TRANSLATION_DROPOUT=p,
                  NUM_CLASSES=[5, 7], DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False))
m = hoi_lta.TaskFusionMFTransformerLTA4Task(cfg)
m.load_state_dict(seeded_state_dict(m, 9))
m = m.to(cuda).set_compute("bf16", "wide").train()
if dev:
    m.enable_device_seed()
feats = [f.to(cuda) for f in seeded_feats(10, [(B, n, 8192), (B, n, 8192), (B, n, 256), (B, n, 2048)])]
params = [q for q in m.parameters() if q.requires_grad]
mode = sys.argv[5] if len(sys.argv) > 5 else "full"
def step():
    for q in params:
        q.grad = None
    o = m.forward_features(*feats)
    if mode == "fwd":
        return o[0].sum()
    loss = o[0].square().mean() + o[1].square().mean()
    loss.backward()
    return loss
if mode == "eagerfirst":
    m._egx_seed_dev.fill_(777)
    e = step(); torch.cuda.synchronize(); print("eager", e.item())
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    loss = step()
g.replay(); torch.cuda.synchronize()
print("OK", sys.argv[1:], loss.item())
