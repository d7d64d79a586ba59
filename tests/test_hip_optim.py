"""FlatAdamW (one HIP launch over the flat gradient bucket) against torch.optim.AdamW -- the optimizer the reference
builds at train.py:97 -- and the CyclicLR schedule of conf/model/tomosar2height.yaml:46-55 driving both."""
import copy

import numpy as np
import pytest
import torch

from tomosar2height_amd.config import berlin_config


def _net(dev):
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3, padding=1), torch.nn.Conv2d(16, 1, 1), torch.nn.Linear(37, 5),
                              torch.nn.Linear(5, 3, bias=False)).to(dev)
    net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)     # dense, non-contiguous
    return net


def _grads(net, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    for p in net.parameters():
        gr = torch.randn(p.shape, generator=g).to(p.device)
        p.grad = gr.contiguous(memory_format=torch.channels_last) if p.dim() == 4 and p.stride() != p.contiguous().stride() else gr


def test_cyclic_lr_schedule_is_the_reference_one():
    from tomosar2height_amd.optim import cyclic_lr
    cfg = berlin_config()
    opt = torch.optim.AdamW(torch.nn.Linear(2, 2).parameters(), lr=cfg.training.learning_rate)
    sched = cyclic_lr(opt, cfg)
    lrs = []
    for _ in range(2001):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    # triangular2, 500 up / 500 down, 1e-4 .. 5e-4, amplitude halves every cycle (train.py:98-104)
    assert lrs[0] == pytest.approx(1e-4) and lrs[500] == pytest.approx(5e-4) and lrs[1000] == pytest.approx(1e-4)
    assert lrs[250] == pytest.approx(3e-4) and lrs[1500] == pytest.approx(3e-4) and lrs[2000] == pytest.approx(1e-4)
    assert opt.param_groups[0]["betas"] == (0.9, 0.999)                 # cycle_momentum: false


@pytest.mark.gpu
def test_flat_adamw_matches_torch_adamw_under_cyclic_lr():
    from tomosar2height_amd.optim import FlatAdamW, cyclic_lr
    dev = torch.device("cuda:0")
    cfg = berlin_config()
    a, b = _net(dev), _net(dev)
    oa = FlatAdamW(a.parameters(), lr=cfg.training.learning_rate)
    ob = torch.optim.AdamW(b.parameters(), lr=cfg.training.learning_rate)
    sa, sb = cyclic_lr(oa, cfg), cyclic_lr(ob, cfg)
    for step in range(6):
        _grads(a, step); _grads(b, step)
        oa.step(); ob.step()
        sa.step(); sb.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(oa.state[pa]["exp_avg_sq"].cpu().numpy(), ob.state[pb]["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=1e-12)
        assert float(oa.state[pa]["step"]) == 6.0


@pytest.mark.gpu
def test_flat_adamw_state_dict_round_trip_and_zero_grad():
    from tomosar2height_amd.optim import FlatAdamW
    dev = torch.device("cuda:0")
    a = _net(dev)
    oa = FlatAdamW(a.parameters(), lr=1e-3)
    for step in range(2):
        _grads(a, step)
        oa.step()
    sd = copy.deepcopy(oa.state_dict())
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}       # torch.optim.AdamW's checkpoint format
    b = _net(dev)
    b.load_state_dict(a.state_dict())
    ob = FlatAdamW(b.parameters(), lr=1e-3)
    ob.load_state_dict(sd)
    tb = torch.optim.AdamW(_net(dev).parameters(), lr=1e-3)
    tb.load_state_dict(copy.deepcopy(sd))                                  # and torch's own optimizer accepts it
    _grads(a, 9); _grads(b, 9)
    oa.step(); ob.step(zero_grad=True)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
        assert not pb.grad.any()
