"""Worker of tests/test_data_parallel_gpu.py: one rank of a multi-process run of the HIP path (started as a fresh child
process by torch.distributed.run, before anything touched the GPU).  All ranks share cuda:0 (one-GPU box), so the
collective backend is gloo for world > 1; `rccl1` runs the same Trainer over a 1-rank RCCL group ("nccl" backend), which
exercises RCCL initialisation and the all-reduce launch on the hardware.

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/dp_worker.py dp OUT.json
"""
import os as _os
# all ranks share cuda:0 here: one stream per process (see bench.py --share-gpu); `rccl1_full` (one rank owns the GPU) switches the
# Trainer's defaults back on itself
_os.environ.setdefault("T2H_OVERLAP_WGRAD", "0")
_os.environ.setdefault("T2H_OVERLAP_CONV_WGRAD", "0")
_os.environ.setdefault("T2H_PIPELINE_TILES", "0")
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_TILES = 4
N_POINTS = 6000


def _tiles(dev):
    from tomosar2height_amd.synthetic import berlin_tile
    out = []
    for i in range(N_TILES):
        t = berlin_tile(300 + i, n_points=N_POINTS + 64 * i)        # ragged N, as real tiles
        out.append({k: t[k].to(dev) for k in ("inputs", "dsm")})
    return out


def _model(dev, perturb=False):
    from tomosar2height_amd import TomoSAR2Height
    from tomosar2height_amd.config import berlin_config
    torch.manual_seed(11)
    m = TomoSAR2Height(berlin_config()).to(dev)
    if perturb:
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.0625)
    return m


def _step(model, dev, tiles, group, every):
    """One optimizer step over `tiles`; returns (flat reduced gradient, flat post-step parameters, avg loss)."""
    from tomosar2height_amd.trainer import Trainer
    grabbed = {}
    tr = Trainer(model, torch.optim.AdamW(model.parameters(), lr=1e-3), device=dev, optimize_every=every, use_cloud=True,
                 process_group=group)
    def grab(flat):
        grabbed["g"] = flat.clone()
        # the same gradient in PARAMETER order (zeros for the parameters that never receive one): lines up with `params` below
        grabbed["gp"] = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()])
    tr.on_reduced = grab
    stepped = [tr.train_step(t) for t in tiles]
    assert stepped[-1] is True and not any(stepped[:-1]), stepped
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    tr.grad_in_param_order = grabbed["gp"]
    return grabbed["g"], params, float(tr.last_avg_loss), tr


def _per_element(p_dp, p_1, g_dp, g_1):
    clear = (g_1.abs() > 1e2 * (g_dp - g_1).abs()) & (g_1.abs() > 1e-6)
    return {"clear_fraction": float(clear.float().mean().item()),
            "param_max_abs_clear": float(((p_dp - p_1).abs() * clear).max().item())}


def run_dp(out_path):
    """W ranks, HIP model, gloo over tensors on the shared GPU: bucket == single-process bucket up to fp32 re-association,
    replicas bit-identical after the step, never-used parameters outside the bucket."""
    from tomosar2height_amd.trainer import broadcast_parameters
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    tiles = _tiles(dev)
    model = _model(dev, perturb=rank == 1)            # rank 1 starts different: the broadcast must fix it
    broadcast_parameters(model, dist.group.WORLD)
    start = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    g_dp, p_dp, loss_dp, tr = _step(model, dev, tiles[rank::world], dist.group.WORLD, N_TILES)
    none_grad = sorted(k for k, p in model.named_parameters() if p.grad is None)
    live = sum(p.numel() for p in model.parameters() if p.grad is not None)
    other = p_dp.clone()
    dist.broadcast(other, src=0)
    identical = bool(torch.equal(other, p_dp))
    # the same four tiles by this rank alone, from the same start
    single = _model(dev)
    off = 0
    with torch.no_grad():
        for p in single.parameters():
            p.copy_(start[off:off + p.numel()].view_as(p))
            off += p.numel()
    g_1, p_1, loss_1, tr1 = _step(single, dev, tiles, None, N_TILES)
    torch.cuda.synchronize()
    res = {"rank": rank, "world": world, "identical_replicas": identical,
           "grad_max_rel": float(((g_dp - g_1).abs().max() / g_1.abs().max()).item()),
           "param_max_abs": float((p_dp - p_1).abs().max().item()),
           # AdamW's first step moves a weight by lr * g / (|g| + eps): where |g| is within the re-association noise of the two
           # summation orders the step is noise too (up to 2 lr apart).  Where the gradient stands clear of that noise (100 x)
           # the two runs must agree tightly:
           "param_max_abs_significant": float(((p_dp - p_1).abs() * (tr1.grad_in_param_order.abs() >
                                                1e2 * (tr.grad_in_param_order - tr1.grad_in_param_order).abs().max())).max().item()),
           "significant_fraction": float((tr1.grad_in_param_order.abs() >
                                          1e2 * (tr.grad_in_param_order - tr1.grad_in_param_order).abs().max()).float().mean().item()),
           # r05: the same PER ELEMENT -- a weight whose own gradient stands 100 x clear of its own re-association
           # difference (and of AdamW's eps) must take the same step in both runs; most weights qualify
           **_per_element(p_dp, p_1, tr.grad_in_param_order, tr1.grad_in_param_order),
           "loss_dp": loss_dp, "loss_single": loss_1, "none_grad": none_grad, "live": live,
           "bucket": int(tr.bucket.flat.numel()), "bucket_single": int(tr1.bucket.flat.numel()),
           "bucket_views_aligned": all(p.grad.data_ptr() % 16 == 0 for p in model.parameters() if p.grad is not None)}
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f)


def run_rccl1(out_path):
    """A 1-rank RCCL group: all-reduce(SUM) over one rank is the identity, so the step must be BIT-identical to the
    step without a process group -- and RCCL itself (communicator, all-reduce kernel on the flat bucket) has run."""
    dev = torch.device("cuda:0")
    tiles = _tiles(dev)[:2]
    a = _model(dev)
    g_a, p_a, loss_a, tr = _step(a, dev, tiles, dist.group.WORLD, 2)
    b = _model(dev)
    g_b, p_b, loss_b, _ = _step(b, dev, tiles, None, 2)
    t = torch.arange(8, dtype=torch.float64, device=dev)
    dist.all_reduce(t)                                  # the float64 all-reduce of the multi-rank mosaic
    torch.cuda.synchronize()
    with open(out_path, "w") as f:
        json.dump({"backend": dist.get_backend(), "grad_equal": bool(torch.equal(g_a, g_b)),
                   "param_equal": bool(torch.equal(p_a, p_b)), "loss_equal": loss_a == loss_b,
                   "f64_ok": bool(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float64)))}, f)


def run_rccl1_full(out_path):
    """r06: the composition eight real ranks run, with one rank owning the GPU -- the Trainer's DEFAULTS
    (tile pipeline on two streams, weight gradients on the side streams, single tiles coalesced into micro-batches) + the RCCL
    all-reduce on the process group's stream + the optimizer boundary -- over TWO optimizer steps of ragged tiles, bit-identical
    to the same loop without a process group (all-reduce over one rank is the identity), and the second step must start from
    the first step's weights in both."""
    from tomosar2height_amd.synthetic import berlin_tile
    from tomosar2height_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    tiles = []
    for i in range(10):
        t = berlin_tile(500 + i, n_points=36000 + 1500 * (i % 4))
        tiles.append({k: t[k].to(dev) for k in ("inputs", "dsm")})
    out = {}
    for name, group in (("rccl", dist.group.WORLD), ("alone", None)):
        model = _model(dev)
        tr = Trainer(model, torch.optim.AdamW(model.parameters(), lr=1e-3), device=dev, optimize_every=5, use_cloud=True,
                     process_group=group)
        tr.pipeline_tiles = tr.overlap_wgrad = tr.overlap_conv_wgrad = True
        assert tr.coalesce_tiles == 4 and tr.pipeline_micro_batches
        grads, losses = [], []
        tr.on_reduced = lambda flat, grads=grads: grads.append(flat.clone())
        stepped = []
        for t in tiles:
            stepped.append(tr.train_step(t))
            if stepped[-1]:
                losses.append(float(tr.last_avg_loss))
        assert stepped == [False] * 4 + [True] + [False] * 4 + [True], stepped
        torch.cuda.synchronize()
        out[name] = (grads, losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()]),
                     (tr._tile_streams is not None, tr._side is not None, tr._conv_side is not None))
    (ga, la, pa, used_a), (gb, lb, pb, used_b) = out["rccl"], out["alone"]
    with open(out_path, "w") as f:
        json.dump({"backend": dist.get_backend(), "boundaries": len(ga),
                   "grad_equal": [bool(torch.equal(x, y)) for x, y in zip(ga, gb)],
                   "grads_differ_between_steps": not torch.equal(ga[0], ga[1]),
                   "loss_equal": la == lb, "param_equal": bool(torch.equal(pa, pb)),
                   "pipeline_and_side_streams_used": [list(used_a), list(used_b)]}, f)


def run_mosaic(out_path):
    """Multi-rank DSMGenerator (tiles i % W == rank, one float64 all-reduce of the dsm/weight pair) == one rank alone."""
    from tomosar2height_amd.generator import DSMGenerator
    from tomosar2height_amd.synthetic import berlin_tile
    rank = dist.get_rank()
    dev = torch.device("cuda:0")
    model = _model(dev)
    tiles = []
    for i, (x0, y0) in enumerate(((0.0, 0.0), (256.0, 0.0), (0.0, 256.0), (256.0, 256.0), (128.0, 128.0))):
        t = berlin_tile(40 + i, n_points=3000)
        t["min_bound"] = torch.tensor([[x0, y0, 0.0]])
        t["max_bound"] = torch.tensor([[x0 + 512.0, y0 + 512.0, 100.0]])
        tiles.append(t)
    tiles.insert(2, {"is_valid": torch.tensor([False])})
    multi = DSMGenerator(model, dev, tiles, bounds=(0.0, 0.0, 768.0, 768.0), process_group=dist.group.WORLD).generate_dsm()
    alone = DSMGenerator(model, dev, tiles, bounds=(0.0, 0.0, 768.0, 768.0)).generate_dsm()
    torch.cuda.synchronize()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"max_abs": float((multi - alone).abs().max().item()), "scale": float(alone.abs().max().item()),
                       "nan": bool(torch.isnan(multi).any().item())}, f)


if __name__ == "__main__":
    mode, out = sys.argv[1], sys.argv[2]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if mode in ("rccl1", "rccl1_full"):
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo")
    try:
        {"dp": run_dp, "rccl1": run_rccl1, "rccl1_full": run_rccl1_full, "mosaic": run_mosaic}[mode](out)
    finally:
        dist.barrier()
        dist.destroy_process_group()
