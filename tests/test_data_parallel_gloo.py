"""CPU, world_size = 2 over gloo: the tile-level data-parallel path of tomosar2height_amd.trainer.Trainer
(one flat all-reduce(SUM) per optimizer step, never-used parameters excluded) reproduces the single-process
accumulated step.  The network inside is the oracle's CPU restatement (tests may use it); on the GPU box the
same Trainer drives the HIP model over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_TILES = 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    from oracle import torch_ref
    from detinit import det_init_
    from ref_import import make_cfg
    return det_init_(torch_ref.TomoSAR2Height(make_cfg(depth=3, reso=16, hidden=32, start_filts=8)), seed=4)


def _tile(i):
    from detinit import synth_cloud
    g = torch.Generator().manual_seed(100 + i)
    dsm = (torch.rand(64, 64, generator=g) * 30).repeat_interleave(8, 0).repeat_interleave(8, 1)
    return {"inputs": synth_cloud(150 + 10 * i, seed=200 + i), "dsm": dsm[None]}


def _run_single():
    from tomosar2height_amd.trainer import Trainer
    model = _model()
    tr = Trainer(model, torch.optim.AdamW(model.parameters(), lr=1e-3), device=torch.device("cpu"),
                 optimize_every=N_TILES, use_cloud=True)
    stepped = [tr.train_step(_tile(i)) for i in range(N_TILES)]
    assert stepped == [False] * (N_TILES - 1) + [True]
    return {k: v.detach().clone() for k, v in model.named_parameters()}, float(tr.last_avg_loss)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tomosar2height_amd.trainer import Trainer, broadcast_parameters
        model = _model()
        if rank == 1:                       # replicas must not depend on every rank building identical weights
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.123)
        broadcast_parameters(model, dist.group.WORLD)
        tr = Trainer(model, torch.optim.AdamW(model.parameters(), lr=1e-3), device=torch.device("cpu"),
                     optimize_every=N_TILES, use_cloud=True, process_group=dist.group.WORLD)
        assert tr.local_every == N_TILES // world
        stepped = [tr.train_step(_tile(i)) for i in range(rank, N_TILES, world)]     # rank r: tiles r, r+W, ...
        assert stepped[-1] is True and not any(stepped[:-1])
        none_grad = sorted(k for k, p in model.named_parameters() if p.grad is None)
        bucket_elems = tr.bucket.flat.numel()
        if rank == 0:
            torch.save({"params": {k: v.detach() for k, v in model.named_parameters()},
                        "loss": float(tr.last_avg_loss), "none_grad": none_grad, "bucket": bucket_elems}, out)
        # replicas stay identical after the step
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        other = flat.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(flat, other)
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_single_process(tmp_path):
    want, want_loss = _run_single()
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    np.testing.assert_allclose(got["loss"], want_loss, rtol=1e-6)
    worst = 0.0
    for k, v in want.items():
        worst = max(worst, (got["params"][k] - v).abs().max().item())
        # AdamW (lr 1e-3) turns the fp32 rounding of a near-zero gradient sum (rank-wise vs sequential order, host-dependent
        # thread counts) into update differences of a few 1e-6; a lost or doubled rank contribution would move whole tensors
        # by ~1e-3
        np.testing.assert_allclose(got["params"][k].numpy(), v.numpy(), rtol=1e-5, atol=2e-5, err_msg=k)
    # the 8 never-used tensors of up_convs[depth-2] (alto.py:241-242) are outside the bucket on every rank
    assert len(got["none_grad"]) == 8 and all("up_convs.1." in k for k in got["none_grad"])
    # every view starts on a 16-byte boundary: sizes are rounded up to 4 floats inside the bucket
    n_live = sum(-(-v.numel() // 4) * 4 for k, v in want.items() if k not in got["none_grad"])
    assert got["bucket"] == n_live


def test_optimize_every_must_divide_by_world():
    from tomosar2height_amd.trainer import Trainer

    class _FakeGroup:
        pass
    model = torch.nn.Linear(2, 2)
    orig = dist.get_world_size
    dist.get_world_size = lambda group=None: 3
    try:
        with pytest.raises(ValueError, match="multiple of the world size"):
            Trainer(model, torch.optim.SGD(model.parameters(), lr=0.1), optimize_every=64, process_group=_FakeGroup())
    finally:
        dist.get_world_size = orig
