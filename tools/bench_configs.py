"""Forward+backward timings of the non-headline configs of BASELINE.json (C3 ASD bf16 L=2, C4 HOI LTA 4-task d=768, C5
EgoT2-g encoders HHI d=256 / HOI d=512) on one GPU: ms per step and achieved algorithmic TFLOP/s (SURVEY.md §8d formula).
These are parity-test configurations, not bench.py lines; the numbers document where the shape-generic kernels stand.
usage: python tools/bench_configs.py [--steps 20]"""
import argparse
import os
import sys
import time
from types import SimpleNamespace as NS

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from egot2_amd.synth import hhi_args  # noqa: E402


def flops(B, segs, d, dff, L, proj):
    """fwd + bwd algorithmic FLOPs: segs = [(T, d_in, projected)]."""
    S = sum(t for t, _, _ in segs)
    N = B * S
    pf = sum(2.0 * B * t * k * d for t, k, pj in segs if pj)
    layer = 2.0 * N * d * 3 * d + 4.0 * B * S * S * d + 2.0 * N * d * d + 4.0 * N * d * dff
    fwd = pf + L * layer
    return 3 * fwd - pf - pf      # bwd = 2 fwd - proj dX (frozen features); projections: fwd + dW only


def timeit(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []

    def run(name, model, feats, out_fn, segs, d, dff, L, B, compute):
        print("running", name, compute, "B =", B, flush=True)
        model = model.to(dev).set_compute(compute).train()
        params = [p for p in model.parameters() if p.requires_grad]

        def step():
            for p in params:
                p.grad = None
            y = out_fn(model, feats)
            ys = y if isinstance(y, (list, tuple)) else [y]
            sum(t.float().sum() for t in ys).backward()
        dt = timeit(step, args.steps)
        f = flops(B, segs, d, dff, L, True)
        rows.append((name, compute, B, dt * 1e3, f / dt / 1e12, B / dt))

    from egot2_amd import hhi_asd, hhi_multitask, hoi_lta, hoi_multitask
    # C3: ASD 3-task translator, 2 layers d=128, bf16 (fused path)
    B = 256
    m = hhi_asd.TaskFusionMFTransformer3Task(hhi_args(num_layers=2, dropout=0.1))
    feats = [torch.randn(B, 15, 256, device=dev) for _ in range(3)]
    for comp in ("bf16", "f32"):
        run("C3 ASD 3-task L=2 d=128 S=45", m, feats, lambda mm, f: mm.forward_features(*f), [(15, 256, True)] * 3, 128, 2048, 2, B, comp)
    # C4: HOI LTA 4-task, n=32 clips per task, d=768, 8 heads, 4 layers (generic path)
    for B in (64, 256):
        cfg = NS(FORECASTING=NS(NUM_INPUT_CLIPS=32, NUM_ACTIONS_TO_PREDICT=20),
                 MODEL=NS(TRANSLATION_HEADS=8, TRANSLATION_LAYERS=4, TRANSLATION_INPUT_FEATURES=768, TRANSLATION_DROPOUT=0.1,
                          NUM_CLASSES=[115, 478], DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False))
        m = hoi_lta.TaskFusionMFTransformerLTA4Task(cfg)
        feats = [torch.randn(B, 32, 8192, device=dev), torch.randn(B, 32, 8192, device=dev), torch.randn(B, 32, 768, device=dev),
                 torch.randn(B, 32, 2048, device=dev)]
        segs = [(32, 8192, True), (32, 8192, True), (32, 768, False), (32, 2048, True)]
        for comp in ("bf16", "f32"):
            run(f"C4 HOI LTA-4task L=4 d=768 S=128", m, feats, lambda mm, f: mm.forward_features(*f), segs, 768, 2048, 4, B, comp)
        del m, feats
        torch.cuda.empty_cache()
    # C5: EgoT2-g encoders
    B = 256
    vocab = {'</s>': 0, '<unk>': 1, 'ttm': 2, 'lam': 3, 'asd': 4, '0': 5, '1': 6}
    m = hhi_multitask.TaskTranslationPromptTransformer(hhi_args(hidden_dim=256, num_heads=4, num_layers=3, dropout=0.1), vocab)
    feats = [torch.randn(B, 15, 256, device=dev) for _ in range(3)]
    for comp in ("bf16", "f32"):
        run("C5 EgoT2-g HHI encoder L=3 d=256 S=45", m, feats, lambda mm, f: mm.encode_features("ttm", *f), [(15, 256, True)] * 3, 256, 2048, 3, B, comp)
    HOI_G_VOCAB = {'</s>': 0, '<unk>': 1, 'pnr': 2, 'oscc': 3, 'action_verb': 4, 'action_noun': 5, 'lta_verb': 6, 'lta_noun': 7,
                   '0': 8, '1': 9, '2': 10, '3': 11}
    m = hoi_multitask.TaskTranslationPromptTransformer6Task(NS(hidden_dim=512, num_heads=8, num_layers=3, dropout=0.1), HOI_G_VOCAB)
    feats = [torch.randn(B, 16, 8192, device=dev), torch.randn(B, 16, 8192, device=dev), torch.randn(B, 8, 2048, device=dev),
             torch.randn(B, 8, 256, device=dev)]
    segs = [(16, 8192, True), (16, 8192, True), (8, 2048, True), (8, 256, True)]
    for comp in ("bf16", "f32"):
        run("C5 EgoT2-g HOI encoder L=3 d=512 S=48", m, feats, lambda mm, f: mm.encode_features("pnr", *f), segs, 512, 2048, 3, B, comp)

    print(f"{'config':44s} {'dtype':5s} {'B':>4s} {'ms/step':>9s} {'TFLOP/s':>9s} {'clips/s':>10s}")
    for name, comp, B, ms, tf, cps in rows:
        print(f"{name:44s} {comp:5s} {B:4d} {ms:9.3f} {tf:9.1f} {cps:10.0f}")


if __name__ == "__main__":
    main()
