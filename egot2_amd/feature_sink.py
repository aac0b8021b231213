"""Producer side of the feature hand-off between the frozen task backbones and the translator.

The reference materialises, per step and per task, the backbone's `middle=True` output as an fp32 tensor, reshapes /
permutes / averages it in separate torch kernels and hands the result to `nn.Linear(8192, d)`
(HHI/models/ttm/model.py:32-37, HHI/models/lam/model.py:27-32, HHI/models/asd/talkNetModel.py:60-64,
HOI/models/pnr/head_helper.py:353-373, HOI/models/lta/lta_models_lta_transfer.py:335-345). Here:

* `FeatureSink` owns ONE packed buffer per task stream, (B, T, d_in) in the dtype the translator's projection consumes in place
  (bf16 for the wide path, fp32 for the d = 128 kernels). Producers write straight into it — a clip's token row, a block of
  frames — so no torch.stack / cat / permute copies sit between the backbone and the projection GEMM.
* `PooledFeatureHead` is the drop-in for the PNR / OSCC head (`ResNetKeyframeLocalizationHead`): with `middle=True` it runs
  libegot2x's `egx_pool_pack` — AvgPool3d + permute + (optional) per-clip temporal mean + bf16 cast in one pass over the res5
  map — into a sink row; with `middle=False` it is the reference's pooled projection (+ activation in eval mode).
* `FeatureCache` keeps a Stage-II feature cache on disk: the frozen backbones never change, so their packed features can be
  computed once per clip and replayed (`forward_features` / `forward_frame_features` consume what it returns unchanged).

Nothing here falls back to torch arithmetic for the pooled path: tensors must be on the GPU, the library must be built.
"""
from __future__ import annotations

import hashlib
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from . import functional as F_egx
from ._lib import check, ptr


class FeatureSink:
    """Packed (B, T, d_in) hand-off buffers, one per named stream."""

    def __init__(self, device, dtype: torch.dtype = torch.bfloat16):
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("FeatureSink holds bf16 or fp32 features")
        self.device, self.dtype = torch.device(device), dtype
        self._buf: Dict[str, torch.Tensor] = {}

    def alloc(self, name: str, B: int, T: int, d_in: int, dtype: Optional[torch.dtype] = None, fresh: bool = False) -> torch.Tensor:
        """(Re)uses the stream's buffer when the shape matches: an inference loop allocates once. `fresh=True` hands out a NEW
        tensor: the producers fill the stream through raw pointers (no autograd version bump), so a stream that a pending
        backward may still read — the translator's projections save their input for dW — must not be refilled in place."""
        dt = dtype or self.dtype
        t = None if fresh else self._buf.get(name)
        if t is None or tuple(t.shape) != (B, T, d_in) or t.dtype != dt:
            t = torch.empty((B, T, d_in), dtype=dt, device=self.device)
            self._buf[name] = t
        return t

    def get(self, name: str) -> torch.Tensor:
        return self._buf[name]

    def names(self) -> List[str]:
        return list(self._buf)

    def put(self, name: str, feats: torch.Tensor, t0: int = 0) -> torch.Tensor:
        """feats (B, t, d_in) -> rows [t0, t0 + t) of the stream (one fused cast-copy into place)."""
        buf = self._buf[name]
        if feats.dim() != 3 or feats.shape[0] != buf.shape[0] or feats.shape[2] != buf.shape[2] or t0 + feats.shape[1] > buf.shape[1]:
            raise _lib.EgxError(f"FeatureSink.put({name}): block {tuple(feats.shape)} at row {t0} does not fit {tuple(buf.shape)}")
        buf[:, t0:t0 + feats.shape[1]].copy_(feats)
        return buf

    def put_pooled_map(self, name: str, fmap: torch.Tensor, pool: Sequence[int], *, token: Optional[int] = None,
                       frames_mean: bool = False) -> torch.Tensor:
        """fmap (N, C, T, H, W) res5 feature map of N = B samples (fp32 or bf16, on the GPU) -> the head's `middle=True` rows.

        token = i: AvgPool3d + permute + temporal mean of the frames -> row i of every sample of the (B, n, H'W'C) stream
                   (encode_clips_pnr's `.mean(dim=1)` for input clip i);
        token = None: the (B, T', H'W'C) rows of all frames (kt = 1) or the single pooled row (kt = T) fill the stream."""
        buf = self._buf[name]
        if not fmap.is_cuda:
            raise _lib.EgxError("put_pooled_map: the feature map must be on the GPU (libegot2x has no CPU path)")
        if fmap.dim() != 5 or fmap.dtype not in (torch.float32, torch.bfloat16):
            raise _lib.EgxError(f"put_pooled_map: feature map {tuple(fmap.shape)} {fmap.dtype}; expected (N, C, T, H, W) fp32 / bf16")
        fmap = fmap.contiguous()
        N, Cc, T, H, W = fmap.shape
        kt, kh, kw = (int(v) for v in pool)
        row_len = (H - kh + 1) * (W - kw + 1) * Cc
        reduce_t = frames_mean or kt > 1
        rows = 1 if reduce_t else T - kt + 1
        if buf.shape[0] != N or buf.shape[2] != row_len:
            raise _lib.EgxError(f"put_pooled_map({name}): stream {tuple(buf.shape)} vs {N} maps of {row_len}-wide rows")
        if token is None:
            if buf.shape[1] != rows:
                raise _lib.EgxError(f"put_pooled_map({name}): stream holds {buf.shape[1]} rows per sample, the map yields {rows}")
            off = 0
        else:
            if not reduce_t:
                raise _lib.EgxError("put_pooled_map: token=i writes ONE row per sample: pass frames_mean=True (or kt = T)")
            if not 0 <= token < buf.shape[1]:
                raise _lib.EgxError(f"put_pooled_map({name}): token {token} outside the {buf.shape[1]} rows of the stream")
            off = token * row_len
        lib = _lib.load()
        dst = buf.data_ptr() + off * buf.element_size()
        check(lib.egx_pool_pack(ptr(fmap), int(fmap.dtype == torch.bfloat16), N, Cc, T, H, W, kt, kh, kw, int(frames_mean),
                                dst, int(buf.dtype == torch.bfloat16), buf.shape[1] * row_len, F_egx._stream()))
        return buf


class PooledFeatureHead(nn.Module):
    """Drop-in for `ResNetKeyframeLocalizationHead` (HOI/models/pnr/head_helper.py:300-390): same constructor, same parameter
    names (`projection.weight / .bias`), same `forward(inputs, middle=False)` contract for a single pathway.

    middle=True : the packed `middle` features (N, T', H'W'C) — through `egx_pool_pack`; into the attached sink stream when
                  `attach(sink, name)` was called (then `token` selects the row and the per-clip temporal mean is fused), else
                  into a fresh tensor of `out_dtype`.
    middle=False: Linear(8192, num_classes) on the pooled rows (libegot2x GEMM), the activation in eval mode, the reference's
                  final permute.
    """

    def __init__(self, dim_in, num_classes, pool_size, dropout_rate=0.0, act_func="softmax", out_dtype=torch.float32):
        super().__init__()
        assert len({len(pool_size), len(dim_in)}) == 1 and len(dim_in) == 1, "one pathway (the PNR / OSCC backbones are single-pathway I3D)"
        self.pool_size = [int(v) for v in pool_size[0]]
        self.dropout_rate = float(dropout_rate)
        self.projection = nn.Linear(8192, num_classes, bias=True)
        if act_func == "softmax_2":
            self.act = nn.Softmax(dim=2)
        elif act_func == "softmax_1":
            self.act = nn.Softmax(dim=1)
        elif act_func == "none":
            self.act = nn.Identity()
        else:
            raise NotImplementedError("{} is not supported as an activation function.".format(act_func))
        self.out_dtype = out_dtype
        self._sink: Optional[FeatureSink] = None
        self._stream_name: Optional[str] = None
        self.token: Optional[int] = None

    def attach(self, sink: FeatureSink, name: str):
        self._sink, self._stream_name = sink, name
        return self

    def forward(self, inputs, middle=False):
        assert len(inputs) == 1, "Input tensor does not contain 1 pathway"
        fmap = inputs[0]
        if self.dropout_rate > 0.0 and self.training:
            raise _lib.EgxError("PooledFeatureHead: the frozen PNR / OSCC backbones run with DROPOUT_RATE = 0 (the reference config); "
                                "feature dropout belongs to the translator (egx_config.p_feat)")
        N, Cc, T, H, W = fmap.shape
        kt, kh, kw = self.pool_size
        row_len = (H - kh + 1) * (W - kw + 1) * Cc
        if middle and self._sink is not None and self.token is not None:
            return self._sink.put_pooled_map(self._stream_name, fmap, self.pool_size, token=self.token, frames_mean=True)
        rows = T - kt + 1
        tmp = FeatureSink(fmap.device, self.out_dtype if middle else torch.float32)
        tmp.alloc("x", N, rows, row_len)
        x = tmp.put_pooled_map("x", fmap, self.pool_size)
        if middle:
            return x
        x = F_egx.linear(x.reshape(N * rows, row_len), self.projection.weight, self.projection.bias, "f32").reshape(N, rows, -1)
        if not self.training:
            x = self.act(x)
        return x.permute(0, 2, 1)


def attach_sink(backbone: nn.Module, head_attr: str, sink: FeatureSink, name: str) -> PooledFeatureHead:
    """Swap `backbone.<head_attr>` (a ResNetKeyframeLocalizationHead-like module with `projection` and `pathway0_avgpool`) for a
    PooledFeatureHead that shares its projection parameters and writes `middle=True` features into `sink[name]`."""
    old = getattr(backbone, head_attr)
    pool = getattr(old, "pathway0_avgpool")
    ks = pool.kernel_size if isinstance(pool.kernel_size, (tuple, list)) else (pool.kernel_size,) * 3
    new = PooledFeatureHead([old.projection.in_features // 4], old.projection.out_features, [list(ks)],
                            dropout_rate=0.0, act_func="none", out_dtype=sink.dtype)
    new.projection = old.projection
    if hasattr(old, "act"):
        new.act = old.act
    new.attach(sink, name)
    setattr(backbone, head_attr, new)
    return new


class FeatureCache:
    """On-disk cache of packed Stage-II features: one file per (clip key, stream set), holding exactly the tensors
    `forward_features` takes. Keys are caller-defined strings (the reference's clip uid + frame window)."""

    def __init__(self, root: str):
        self.root = root
        os.makedirs(root, exist_ok=True)

    def _path(self, key: str) -> str:
        h = hashlib.sha1(key.encode()).hexdigest()
        return os.path.join(self.root, h[:2], h + ".pt")

    def has(self, key: str) -> bool:
        return os.path.exists(self._path(key))

    def save(self, key: str, feats: Dict[str, torch.Tensor]) -> None:
        path = self._path(key)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + f".tmp{os.getpid()}"
        torch.save({"key": key, "feats": {k: v.detach().cpu() for k, v in feats.items()}}, tmp)
        os.replace(tmp, path)       # atomic: data-loader workers may race on the same clip

    def load(self, key: str, device=None) -> Dict[str, torch.Tensor]:
        blob = torch.load(self._path(key), map_location="cpu")
        if blob.get("key") != key:
            raise KeyError(f"feature cache: hash collision or foreign file for key {key!r}")
        return {k: (v.to(device, non_blocking=True) if device is not None else v) for k, v in blob["feats"].items()}

    def load_batch(self, keys: Iterable[str], sink: FeatureSink) -> Dict[str, torch.Tensor]:
        """Clips `keys` (each cached with a leading batch dimension of 1) -> the sink's streams (B, T, d_in), in place."""
        keys = list(keys)
        first = self.load(keys[0])
        for name, t in first.items():
            sink.alloc(name, len(keys), t.shape[1], t.shape[2], dtype=t.dtype)
        for b, key in enumerate(keys):
            for name, t in (first if b == 0 else self.load(key)).items():
                sink.get(name)[b:b + 1].copy_(t, non_blocking=True)
        return {name: sink.get(name) for name in first}
