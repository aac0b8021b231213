"""Single-kernel parity (-m gpu): every HIP op against the fp64 torch restatement of the same op."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(lib, rc):
    assert rc == 0, lib.egx_last_error().decode()


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("compute,tol", [(0, 2e-5), (1, 2e-2)])
@pytest.mark.parametrize("M,N,K", [(96, 64, 32), (3840, 128, 256), (720, 384, 128), (1000, 2048, 128), (257, 132, 68),
                                    (45, 593, 96), (130, 7, 33)])
def test_gemm(egx_lib, cuda, layout, compute, tol, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + layout)
    if layout == 0:
        A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        ref = A.double() @ B.double().T
    elif layout == 1:
        A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
        ref = A.double() @ B.double()
    else:
        A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
        ref = A.double().T @ B.double()
    bias = torch.randn(N, generator=g) if layout != 2 else None
    if bias is not None:
        ref = torch.relu(ref + bias.double())
    Ad, Bd = A.to(cuda), B.to(cuda)
    Cd = torch.full((M, N), float("nan"), device=cuda)
    bd = bias.to(cuda) if bias is not None else None
    scratch = torch.empty(64 * M * N * 4 + 256, dtype=torch.uint8, device=cuda)
    _check(egx_lib, egx_lib.egx_gemm(layout, _ptr(Ad), _ptr(Bd), _ptr(Cd), M, N, K, _ptr(bd), 1 if bias is not None else 0,
                                     compute, _ptr(scratch), scratch.numel(), _stream()))
    torch.cuda.synchronize()
    err = (Cd.cpu().double() - ref).abs().max().item()
    scale = math.sqrt(K)
    assert err <= tol * scale, f"max err {err} (tol {tol * scale})"


@pytest.mark.parametrize("rows,d", [(45, 128), (1000, 256), (333, 768), (64, 1024), (17, 100)])
def test_layernorm_fwd_bwd(egx_lib, cuda, rows, d):
    g = torch.Generator().manual_seed(rows + d)
    x, r = torch.randn(rows, d, generator=g), torch.randn(rows, d, generator=g)
    w, b = torch.randn(d, generator=g), torch.randn(d, generator=g)
    dy = torch.randn(rows, d, generator=g)
    xd, rd, wd, bd, dyd = [t.to(cuda) for t in (x, r, w, b, dy)]
    pre = torch.empty(rows, d, device=cuda)
    stats = torch.empty(rows, 2, device=cuda)
    y = torch.empty(rows, d, device=cuda)
    _check(egx_lib, egx_lib.egx_layernorm_fwd(_ptr(xd), _ptr(rd), _ptr(wd), _ptr(bd), 1e-5, _ptr(pre), _ptr(stats), _ptr(y),
                                              rows, d, _stream()))
    xr = (x + r).double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-5)
    yr.backward(dy.double())
    assert (y.cpu().double() - yr.detach()).abs().max() < 1e-5
    assert (pre.cpu() - (x + r)).abs().max() < 1e-6
    dx = torch.empty(rows, d, device=cuda)
    dw = torch.zeros(d, device=cuda)
    db = torch.zeros(d, device=cuda)
    _check(egx_lib, egx_lib.egx_layernorm_bwd(_ptr(dyd), _ptr(pre), _ptr(stats), _ptr(wd), _ptr(dx), _ptr(dw), _ptr(db),
                                              rows, d, _stream()))
    torch.cuda.synchronize()
    assert (dx.cpu().double() - xr.grad).abs().max() < 2e-5
    assert (dw.cpu().double() - wr.grad).abs().max() < 1e-4 * math.sqrt(rows)
    assert (db.cpu().double() - br.grad).abs().max() < 1e-4 * math.sqrt(rows)


@pytest.mark.parametrize("B,S,H,d", [(3, 45, 4, 128), (2, 30, 4, 128), (2, 128, 8, 768), (1, 450, 4, 128), (2, 48, 8, 512),
                                      (2, 45, 4, 256), (1, 200, 8, 768)])
def test_attention_fwd_bwd(egx_lib, cuda, B, S, H, d):
    g = torch.Generator().manual_seed(B * 1000 + S + d)
    qkv = torch.randn(B, S, 3 * d, generator=g)
    do = torch.randn(B, S, d, generator=g)
    qr = qkv.double().requires_grad_(True)
    dh = d // H
    q, k, v = [t.reshape(B, S, H, dh).permute(0, 2, 1, 3) for t in (qr[..., :d], qr[..., d:2 * d], qr[..., 2 * d:])]
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, d)
    ref.backward(do.double())
    qd, dod = qkv.to(cuda), do.to(cuda)
    out = torch.empty(B, S, d, device=cuda)
    lse = torch.empty(B, H, S, device=cuda)
    _check(egx_lib, egx_lib.egx_attention_fwd(_ptr(qd), _ptr(out), _ptr(lse), B, S, H, d, 0.0, 0, _stream()))
    dq = torch.full((B, S, 3 * d), float("nan"), device=cuda)
    _check(egx_lib, egx_lib.egx_attention_bwd(_ptr(qd), _ptr(out), _ptr(lse), _ptr(dod), _ptr(dq), B, S, H, d, 0.0, 0, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref.detach()).abs().max() < 2e-5
    assert (dq.cpu().double() - qr.grad).abs().max() < 1e-4


@pytest.mark.parametrize("compute,tol", [(0, 2e-4), (1, 8e-2)])  # bf16: ReLU-mask sign flips dominate (sqrt of flipped fraction)
@pytest.mark.parametrize("N,S,d_ff", [(45 * 8, 45, 2048), (45 * 256, 45, 2048), (100, 30, 256)])
def test_fused_ffn_weight_grads(egx_lib, cuda, compute, tol, N, S, d_ff):
    """ffn_dw_kernel (H and dH recomputed on chip) against autograd of the same two linears in fp64."""
    g = torch.Generator().manual_seed(N + d_ff)
    x1 = torch.randn(N, 128, generator=g)
    gy = torch.randn(N, 128, generator=g) * 0.1
    W1 = torch.randn(d_ff, 128, generator=g) / math.sqrt(128)
    b1 = torch.randn(d_ff, generator=g) * 0.1
    W2 = torch.randn(128, d_ff, generator=g) / math.sqrt(d_ff)
    W1r, b1r, W2r = [t.double().requires_grad_(True) for t in (W1, b1, W2)]
    y = torch.relu(x1.double() @ W1r.T + b1r) @ W2r.T
    y.backward(gy.double())
    d = [t.to(cuda) for t in (x1, gy, W1, b1, W2)]
    dW1 = torch.zeros(d_ff, 128, device=cuda)
    db1 = torch.zeros(d_ff, device=cuda)
    dW2 = torch.zeros(128, d_ff, device=cuda)
    scratch = torch.empty(egx_lib.egx_ffn_dw_scratch(N, d_ff, compute) + 256, dtype=torch.uint8, device=cuda)
    _check(egx_lib, egx_lib.egx_ffn_dw(_ptr(d[0]), _ptr(d[1]), _ptr(d[2]), _ptr(d[3]), _ptr(d[4]), N, S, d_ff, 0.0, 0,
                                       _ptr(dW1), _ptr(db1), _ptr(dW2), compute, _ptr(scratch), _stream()))
    torch.cuda.synchronize()
    errs = {name: ((got.cpu().double() - ref).norm() / ref.norm()).item()
            for got, ref, name in ((dW1, W1r.grad, "dW1"), (db1, b1r.grad, "db1"), (dW2, W2r.grad, "dW2"))}
    assert all(e < tol for e in errs.values()), f"rel errs {errs}"
