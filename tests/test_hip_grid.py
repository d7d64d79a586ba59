"""GPU parity of the grid-side fusions (SURVEY 8f-1 first step) against the plain torch composition they replace
(conv + bias + ReLU, cat + 1x1 conv, F.interpolate) on the same device, and against the C oracle for the upsample."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _cl(t):
    return t.to(_dev()).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("cin,cout,k,hw,relu", [(32, 64, 3, 64, True), (64, 32, 1, 33, False), (8, 16, 3, 17, True)])
def test_conv_bias_act(cin, cout, k, hw, relu):
    from tomosar2height_amd import grid
    import tomosar2height_amd as t2h
    t2h.allow_library_fallback(k != 3).set()           # conv_bias_act's generic (non-3x3) branch is MIOpen + fused bias/ReLU
    g = torch.Generator().manual_seed(cin + cout)
    torch.manual_seed(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2)
    with torch.no_grad():
        conv.bias.copy_(torch.randn(cout, generator=g))
    conv = conv.to(_dev()).to(memory_format=torch.channels_last)
    x = _cl(torch.randn(2, cin, hw, hw, generator=g)).requires_grad_(True)
    gout = _cl(torch.randn(2, cout, hw, hw, generator=g))
    y = grid.conv_bias_act(x, conv, relu=relu)
    y.backward(gout)
    got = (y.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    x.grad = None
    conv.zero_grad()
    yr = conv(x)
    yr = F.relu(yr) if relu else yr
    yr.backward(gout)
    want = (yr.detach(), x.grad, conv.weight.grad, conv.bias.grad)
    # with a ReLU, a pre-activation within rounding of zero may take different sides in the two fp32 summation orders and
    # move single gradient entries: judge those gradients in L2 (exact per-op parity: tests/test_hip_conv.py)
    for i, (a, b, tol) in enumerate(zip(got, want, (1e-6, 1e-5, 1e-4, 1e-5))):
        if relu and i > 0:
            assert (a - b).norm().item() <= 2e-3 * b.norm().item()
        else:
            scale = b.abs().max().item() + 1e-12
            assert (a - b).abs().max().item() <= tol * scale + 1e-7


def test_head1x1_matches_cat_conv():
    from tomosar2height_amd import grid
    g = torch.Generator().manual_seed(4)
    conv = torch.nn.Conv2d(288, 1, 1).to(_dev())
    xs = [_cl(torch.randn(1, c, 40, 40, generator=g)).requires_grad_(True) for c in (32, 64, 128, 64)]
    gout = torch.randn(1, 1, 40, 40, generator=g).to(_dev())
    out = grid.head1x1(xs, conv)
    out.backward(gout)
    got = [out.detach().clone()] + [x.grad.clone() for x in xs] + [conv.weight.grad.clone(), conv.bias.grad.clone()]
    for x in xs:
        x.grad = None
    conv.zero_grad()
    ref = conv(torch.cat(xs, dim=1))
    ref.backward(gout)
    want = [ref.detach()] + [x.grad for x in xs] + [conv.weight.grad, conv.bias.grad]
    for a, b in zip(got, want):
        scale = b.abs().max().item() + 1e-12
        assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-5 * scale


def test_head1x1_weight_gradient_accumulates_into_existing_grads():
    """r05: under ``mlp.direct_grad_accumulation`` the head's weight / bias gradient is ADDED to the ``.grad`` buffers that exist
    (the trainer's bucket; ``t2h_head1x1_bwd`` dx_flags bit 1) and autograd receives ``None`` for them: exactly the gradient
    of the plain path on top of what the buffers held, bit for bit (the same fixed summation tree, one more addition)."""
    from tomosar2height_amd import grid, mlp
    g = torch.Generator().manual_seed(5)
    conv = torch.nn.Conv2d(288, 1, 1).to(_dev())
    xs = [_cl(torch.randn(1, c, 40, 40, generator=g)).requires_grad_(True) for c in (32, 64, 128, 64)]
    gout = torch.randn(1, 1, 40, 40, generator=g).to(_dev())
    grid.head1x1(xs, conv).backward(gout)
    plain_w, plain_b, plain_x = conv.weight.grad.clone(), conv.bias.grad.clone(), [x.grad.clone() for x in xs]
    held_w, held_b = torch.randn_like(conv.weight), torch.randn_like(conv.bias)
    conv.weight.grad, conv.bias.grad = held_w.clone(), held_b.clone()
    for x in xs:
        x.grad = None
    seen = []
    hooks = [conv.weight.register_hook(lambda gr: seen.append(gr)), conv.bias.register_hook(lambda gr: seen.append(gr))]
    with mlp.direct_grad_accumulation(True):
        grid.head1x1(xs, conv).backward(gout)
    for h in hooks:
        h.remove()
    assert seen == [None, None]                                        # nothing went through autograd's accumulation
    assert torch.equal(conv.weight.grad, held_w + plain_w) and torch.equal(conv.bias.grad, held_b + plain_b)
    assert all(torch.equal(x.grad, p) for x, p in zip(xs, plain_x))


@pytest.mark.parametrize("b,c,h,size", [(1, 32, 256, 512), (2, 8, 16, 32), (1, 4, 7, 19)])
def test_upsample_cl_vs_oracle(b, c, h, size):
    from tomosar2height_amd import grid
    from oracle import c_oracle
    g = torch.Generator().manual_seed(h)
    x = torch.randn(b, c, h, h, generator=g)
    add = torch.randn(b, c, size, size, generator=g)
    gout = torch.randn(b, c, size, size, generator=g)
    xd = _cl(x).requires_grad_(True)
    y = grid.upsample_bilinear_cl(xd, size)
    assert y.permute(0, 2, 3, 1).is_contiguous()
    np.testing.assert_allclose(y.detach().cpu().numpy(), c_oracle.upsample_bilinear_fwd(x.numpy(), size), rtol=1e-6, atol=1e-6)
    y.backward(_cl(gout))
    np.testing.assert_allclose(xd.grad.cpu().numpy(), c_oracle.upsample_bilinear_bwd(gout.numpy(), h, h), rtol=1e-5, atol=1e-5)
    y2 = grid.upsample_bilinear_cl(xd.detach(), size, _cl(add))
    np.testing.assert_allclose(y2.cpu().numpy(), y.detach().cpu().numpy() + add.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("b,c,h,w", [(1, 32, 64, 64), (2, 64, 6, 10), (1, 512, 2, 2), (1, 4, 256, 256)])
def test_maxpool2x2_matches_aten_including_ties(b, c, h, w):
    """2x2 max-pool forward / backward on NHWC planes against ATen on the CPU; the planes hold many exact ties (zeros,
    repeated values) whose gradient routing (first maximum in scan order) must match."""
    from tomosar2height_amd import grid
    g = torch.Generator().manual_seed(b * c + h)
    x = torch.randint(-2, 3, (b, c, h, w), generator=g).float()          # few distinct values: ties in most windows
    x[:, :, : h // 2] *= (torch.rand(b, c, h // 2, w, generator=g) < 0.5)  # and blocks of zeros
    gout = torch.randn(b, c, h // 2, w // 2, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    yr.backward(gout)
    xg = _cl(x).requires_grad_(True)
    y = grid.maxpool2x2(xg, torch.nn.MaxPool2d(2, 2))
    assert type(y.grad_fn).__name__ == "_MaxPool2x2Backward"
    y.backward(_cl(gout))
    assert torch.equal(y.detach().cpu(), yr.detach())
    assert torch.equal(xg.grad.cpu(), xr.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("b,c,h,w", [(1, 8, 4, 4), (2, 32, 16, 8), (1, 64, 1, 1), (1, 16, 33, 5), (1, 256, 32, 32)])
def test_upsample2x_matches_nn_upsample(b, c, h, w):
    """grid.upsample2x == nn.Upsample(mode='bilinear', scale_factor=2) (align_corners=False), forward and backward, on CPU
    torch as the reference (ATen's own arithmetic); odd sizes and 1 x 1 planes included."""
    from tomosar2height_amd import grid
    g = torch.Generator().manual_seed(b + c + h)
    x = torch.randn(b, c, h, w, generator=g)
    gy = torch.randn(b, c, 2 * h, 2 * w, generator=g)
    xr = x.clone().requires_grad_(True)
    want = torch.nn.Upsample(mode="bilinear", scale_factor=2)(xr)
    want.backward(gy)
    xd = x.to("cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    got = grid.upsample2x(xd)
    assert got.is_contiguous(memory_format=torch.channels_last) or min(got.shape[1:]) == 1
    got.backward(gy.to("cuda"))
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-5, atol=1e-5)
