"""nn.Linear-shaped products on f16-pair plane images (csrc/planes.hip, round 5: the input projection of an upper encoder layer -
models/encoder.py:25-30 - and its two gradients at BASELINE.json configs[4] sizes).  Through the C ABI against f64 products of the SAME
f32 inputs: f32-grade bars (the products are 22-bit f16 pairs with f32 accumulation), also with gradient magnitudes spread over many
orders between row blocks and unit groups, an all-zero row block and an all-zero group.  The model-level parity tests at configs[4]
dimensions (tests/test_gpu_tiles.py::test_config_c_*, tests/test_gpu_round5.py) run through cpg.ops.Linear2PlanesFn."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X: no CUDA/HIP device visible")


@pytest.mark.parametrize("R,H,G,K1,K2,wild", [(8192, 256, 3, 256, 256, False), (8192, 128, 4, 128, 384, True), (12800, 1024, 3, 1024, 1024, True)])
def test_plane_products_vs_f64(R, H, G, K1, K2, wild):
    from cpg import ops
    from cpg.ops import _p, _stream, call, query
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(R + H + G)
    K, N = K1 + K2, G * H
    assert ops.planes_ok(R, K, N)
    x1 = (torch.rand(R, K1, generator=g) * 2 - 1).to(dev)                   # states: |x| <= 1
    x2 = (torch.rand(R, K2, generator=g) * 2 - 1).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    dy = (torch.randn(R, N, generator=g) * 1e-3)
    if wild:
        rowexp = torch.randint(-8, 5, (R // 32,), generator=g).repeat_interleave(32).float()
        colexp = torch.randint(-9, 4, (H // 32,), generator=g).repeat_interleave(32).float().repeat(G)
        sc = (10.0 ** rowexp)[:, None] * (10.0 ** colexp)[None, :]
        sc[32:64, :] = 0.0
        sc[:, 32:64] = 0.0           # unit group 1 of block 0
        dy = dy * sc
    dy = dy.to(dev)
    X = torch.cat([x1, x2], 1).double()
    # ---- forward
    ximg = ops.pair_rows(x1, x2)
    y = torch.full((R, N), float("nan"), device=dev)
    sc_ = torch.empty(int(query("cpg_weight_image_bytes", N, K)), device=dev, dtype=torch.uint8)
    call("cpg_linear_fwd_planes", _p(ximg), R, K, _p(w), K, _p(b), _p(y), N, N, 0, _p(sc_), sc_.numel(), _stream())
    want = X @ w.double().T + b.double()
    bound = (X.abs() @ w.double().abs().T).amax()
    assert (y.double() - want).abs().max().item() <= 3e-6 * bound.item()
    y2 = y.clone()
    call("cpg_linear_fwd_planes", _p(ximg), R, K, _p(w), K, None, _p(y2), N, N, 1, _p(sc_), sc_.numel(), _stream())   # accumulate, no bias
    assert (y2.double() - (2 * want - b.double())).abs().max().item() <= 6e-6 * bound.item()
    # ---- gradient image + dX + dW
    off = (ctypes.c_int * 4)(*[q * H for q in range(G)], *([0] * (4 - G)))
    gp = torch.empty(int(query("cpg_grad_planes_bytes", R, H, G)), device=dev, dtype=torch.uint8)
    gp.fill_(0xFF)
    call("cpg_grad_planes", _p(dy), N, R, H, G, off, _p(gp), _stream())
    dx = torch.full((R, K), float("nan"), device=dev)
    sc2 = torch.empty(int(query("cpg_weight_image_bytes", K, N)), device=dev, dtype=torch.uint8)
    call("cpg_linear_bwd_input_planes", _p(gp), R, H, G, _p(w), K, _p(dx), K, K, 0, _p(sc2), sc2.numel(), _stream())
    want_dx = dy.double() @ w.double()
    # per 32-row block: relative to what the block's largest |dy| row could contribute (f32 accumulate + 22-bit operands)
    blk = (dy.double().abs() @ w.double().abs()).view(R // 32, 32, K).amax(dim=(1, 2), keepdim=True).expand(R // 32, 32, K).reshape(R, K)
    assert torch.isfinite(dx).all()
    assert ((dx.double() - want_dx).abs() <= 4e-6 * blk + 1e-300).all(), float(((dx.double() - want_dx).abs() / (blk + 1e-300)).max())
    if wild:
        assert torch.equal(dx[32:64], torch.zeros_like(dx[32:64]))            # the all-zero row block: exact zeros
    dw = torch.full((N, K), float("nan"), device=dev)
    ws = torch.empty(int(query("cpg_linear_bwd_weight_planes_workspace", R, H, G, K)), device=dev, dtype=torch.uint8)
    call("cpg_linear_bwd_weight_planes", _p(gp), R, H, G, _p(ximg), K, _p(dw), K, 0, _p(ws), ws.numel(), _stream())
    torch.cuda.synchronize()
    want_dw = dy.double().T @ X
    grp = (dy.double().abs().T @ X.abs()).view(G, H // 32, 32, K).amax(dim=(2, 3), keepdim=True).expand(G, H // 32, 32, K).reshape(N, K)
    assert torch.isfinite(dw).all()
    assert ((dw.double() - want_dw).abs() <= 4e-6 * grp + 1e-300).all(), float(((dw.double() - want_dw).abs() / (grp + 1e-300)).max())
    if wild:
        assert torch.equal(dw[32:64], torch.zeros_like(dw[32:64]))            # the all-zero unit group of block 0


def test_linear2_planes_function_matches_linear2():
    """cpg.ops.Linear2PlanesFn (the encoder's upper-layer input projection where cpg_planes_ok) against cpg.ops.Linear2Fn (the round-4
    engines) through autograd: outputs and all four gradients within f32-grade bars of each other."""
    from cpg import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    R, H, K1 = 8192, 256, 256
    x1 = (torch.rand(R, K1, generator=g) * 2 - 1).to(dev).requires_grad_()
    x2 = (torch.rand(R, K1, generator=g) * 2 - 1).to(dev).requires_grad_()
    w = (torch.randn(3 * H, 2 * K1, generator=g) / 20).to(dev).requires_grad_()
    b = (torch.randn(3 * H, generator=g) * 0.1).to(dev).requires_grad_()
    gy = (torch.randn(R, 3 * H, generator=g) * 1e-2).to(dev)
    outs = []
    for planes in (True, False):
        for t in (x1, x2, w, b):
            t.grad = None
        if planes:
            y = ops.Linear2PlanesFn.apply(x1, x2, ops.pair_rows(x1.detach(), x2.detach()), w, b, 3)
        else:
            y = ops.Linear2Fn.apply(x1, x2, w, b)
        y.backward(gy)
        outs.append([y.detach().clone()] + [t.grad.clone() for t in (x1, x2, w, b)])
    for a, r, name in zip(outs[0], outs[1], ("y", "dx1", "dx2", "dw", "db")):
        assert (a - r).abs().max().item() <= 2e-5 * r.abs().max().item(), name
