#!/usr/bin/env python3
"""BASELINE.json configs[0]: the `run_ttm.py` plumbing entry on synthetic feature tensors.

Mirrors HHI/scripts/run_ttm.py:15-36 over HHI/tasks/ttm/video_task_2loader.py:16-36,62-64 without Lightning, datasets or
backbone checkpoints: argparse flags with the reference's names (HHI/configs/ttm/config.py:11-55) -> `build_model(args)`
through the MODEL registry -> per step `model.forward_features(...)` -> `CrossEntropyLoss(weight=[0.266, 0.734])` ->
`Adam(lr, weight_decay)` (FusedAdam over the flat gradient buffer). Random (B, T, 256) tensors stand in for the frozen
backbones' `middle=True` features. Needs a GPU (the translator has no CPU path).

  python tools/run_ttm_synth.py --model TaskFusionMFTransformer2Task --num_layers 1 --hidden_dim 128 --dropout 0.5
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parser():
    p = argparse.ArgumentParser()
    # the flags the reference's TTM recipe passes (README.md:81-84)
    p.add_argument("--model", default="TaskFusionMFTransformer2Task")
    p.add_argument("--num_layers", type=int, default=1)
    p.add_argument("--hidden_dim", type=int, default=128)
    p.add_argument("--num_heads", type=int, default=4)
    p.add_argument("--dropout", type=float, default=0.5)
    p.add_argument("--lr", type=float, default=5e-4)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--nofreeze", action="store_true", default=True)
    p.add_argument("--lam_checkpoint", default=None)
    p.add_argument("--ttm_checkpoint", default=None)
    p.add_argument("--asd_checkpoint", default=None)
    p.add_argument("--two_loader", action="store_true")
    # synthetic-run knobs
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--frames", type=int, default=15)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f32s"])
    p.add_argument("--seed", type=int, default=0)
    return p


def main(argv=None):
    args = parser().parse_args(argv)
    import torch
    from egot2_amd import hhi_ttm
    from egot2_amd.train import CrossEntropyLoss, FusedAdam
    if not torch.cuda.is_available():
        raise SystemExit("run_ttm_synth.py needs a GPU: the translator has no CPU path")
    dev = torch.device("cuda:0")
    torch.manual_seed(args.seed)
    args.hidden_dim2 = 512
    model = hhi_ttm.build_model(args).to(dev).set_compute(args.dtype).train()     # TalkingToMe2Loader.__init__ :20
    criterion = CrossEntropyLoss(weight=torch.FloatTensor([0.266, 0.734])).to(dev)    # :21-22
    optimizer = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)   # configure_optimizers :62-64
    K = 2 if args.model.endswith("2Task") else 3
    g = torch.Generator().manual_seed(1234)
    # a fixed synthetic "dataset" of 4 batches whose label depends on the features, so that the loss can go down
    batches = []
    for _ in range(4):
        feats = [torch.randn(args.batch_size, args.frames, 256, generator=g) for _ in range(K)]
        label = (feats[0].mean(dim=(1, 2)) > 0).long()
        batches.append(([f.to(dev) for f in feats], label.to(dev)))
    losses = []
    for step in range(args.steps):                                                # training_step :29-36
        feats, label = batches[step % len(batches)]
        optimizer.zero_grad(set_to_none=True)
        loss = criterion(model.forward_features(*feats), label)
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
        print(f"step {step:3d}  train_loss {losses[-1]:.4f}", flush=True)
    model.eval()
    with torch.no_grad():
        feats, label = batches[0]
        acc = (model.forward_features(*feats).argmax(1) == label).float().mean().item()
    print(f"final: first-epoch mean loss {sum(losses[:4]) / 4:.4f} -> last-epoch mean loss {sum(losses[-4:]) / 4:.4f}; "
          f"train accuracy on batch 0: {acc:.3f}")
    return losses


if __name__ == "__main__":
    main()
