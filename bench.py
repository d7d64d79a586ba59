#!/usr/bin/env python3
"""Headline benchmark: training tile-steps/s on Berlin-shaped synthetic tiles, cloud-only, fp32
(BASELINE.json configs[1]: N = 131072 points / tile, R = 256, ALTO depth 5, 512^2 target).

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one tile through ``Trainer.train_step``: forward, L1 loss, backward, and -- every
``optimize_every``/world tiles per rank -- one RCCL all-reduce(SUM) of the flat gradient bucket + AdamW step
(reference: trainer.py:47-89, optimize_every = 64).  Tiles are resident in HBM before the timed region.
Weak scaling: every rank runs K tiles; value = world * K / max-over-ranks(time).

Rank 0 prints ONE JSON line (contract fields + `roofline` + `cpu_baseline` + a per-kernel table).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--points", type=int, default=131072, help="points per tile (SURVEY.md 8d default)")
    ap.add_argument("--optimize-every", type=int, default=64, help="tiles per optimizer step (reference: 64)")
    ap.add_argument("--tile-pool", type=int, default=4, help="distinct resident tiles cycled per rank")
    ap.add_argument("--channels-last", type=int, default=1, help="grid side in NHWC (same numerics, no layout copies)")
    ap.add_argument("--miopen-find", type=int, default=0, help="torch.backends.cudnn.benchmark")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-every", type=int, default=16, help="record per-kernel HIP events on every n-th timed step")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores")
    ap.add_argument("--mlp-precision", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="fp32 = native fp32 MFMA, BASELINE configs[1] (the headline); bf16 = bf16-operand MFMA "
                         "(configs[2]); bf16x3 = fp32-grade products from 3-way bf16 splitting (opt-in experiment)")
    ap.add_argument("--hip-graph", type=int, default=0,
                    help="1 = capture the fixed-N tile step (fwd+loss+bwd) into a hipGraph and replay it")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train = the headline (configs[1]); infer = configs[4]: Munich cloud+image forward only")
    ap.add_argument("--batch", type=int, default=1, help="tiles per forward in --mode infer")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real multi-GPU run); gloo only to smoke-test the N>1 code path "
                         "with several ranks sharing one GPU")
    ap.add_argument("--use-image", action="store_true",
                    help="train the cloud+image network (BASELINE configs[2] with --mlp-precision bf16): image U-Net encoder on")
    ap.add_argument("--uniform-xy", action="store_true", help="no-skew control: all points uniform in the tile (SURVEY 8d)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (gloo smoke test only)")
    return ap.parse_args()


def pmc_traffic():
    """HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), committed under
    profiles/ by profiles/collect_pmc.py for the N = 131072 workload; {} if not collected."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get("bytes_per_launch", {})
    except (OSError, ValueError):
        return {}


def cpu_baseline(points: int, threads: int):
    """The oracle's torch restatement of the reference model ("port"), one tile-step (fwd + bwd) of the SAME
    workload on the host cores.  This is the only place bench.py touches oracle/."""
    from oracle import torch_ref
    from tomosar2height_amd.config import berlin_config
    from tomosar2height_amd.synthetic import berlin_tile
    if threads > 0:
        torch.set_num_threads(threads)
    cores = torch.get_num_threads()
    torch.manual_seed(0)
    model = torch_ref.TomoSAR2Height(berlin_config())
    warm = berlin_tile(1, n_points=2048)
    torch_ref.train_loss(model, warm["inputs"], None, warm["dsm"]).backward()
    tile = berlin_tile(0, n_points=points)
    t0 = time.perf_counter()
    torch_ref.train_loss(model, tile["inputs"], None, tile["dsm"]).backward()
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"1 tile-step (fwd+bwd, N={points}, fp32) of the oracle torch restatement, {dt:.1f} s"}


def infer_bench(args, world, rank, dev, group):
    """BASELINE.json configs[4]: Munich (ALTO depth 6, footprint head, cloud + image), forward only as in
    generator.py:142-147 (`model.eval(); no_grad`), `--batch` equal-N tiles per forward, tiles round-robin over ranks
    (no collective on the data path).  One step = one forward of `--batch` tiles; value = tiles/s over all ranks."""
    from tomosar2height_amd import TomoSAR2Height
    from tomosar2height_amd.config import munich_config
    from tomosar2height_amd.synthetic import berlin_tile
    torch.manual_seed(0)
    model = TomoSAR2Height(munich_config(use_image=True)).to(dev).eval()
    model.set_channels_last(bool(args.channels_last))
    model.set_mlp_precision(args.mlp_precision)
    batches = []
    for i in range(args.tile_pool):
        ts = [berlin_tile(seed=1000 * rank + 10 * i + j, n_points=args.points, with_image=True) for j in range(args.batch)]
        batches.append((torch.cat([t["inputs"] for t in ts], 0).to(dev), torch.cat([t["image"] for t in ts], 0).to(dev)))

    graph = None
    if args.hip_graph:
        # BASELINE configs[4] "hipGraph-captured forward": one capture for the fixed tile shape, later tiles are copied
        # into the static inputs and the ~230 launches replayed with one host call
        static_cloud, static_image = torch.empty_like(batches[0][0]), torch.empty_like(batches[0][1])
        static_cloud.copy_(batches[0][0]); static_image.copy_(batches[0][1])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                model(input_cloud=static_cloud, input_image=static_image)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            static_out = model(input_cloud=static_cloud, input_image=static_image)

    def run(n, off=0):
        with torch.no_grad():
            for s in range(n):
                cloud, image = batches[(off + s) % len(batches)]
                if graph is not None:
                    static_cloud.copy_(cloud, non_blocking=True)
                    static_image.copy_(image, non_blocking=True)
                    graph.replay()
                else:
                    model(input_cloud=cloud, input_image=image)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        print(json.dumps({
            "metric": "inference tiles/sec (Munich cloud+image+footprint, forward only)",
            "value": round(world * args.steps * args.batch / elapsed, 4), "unit": "tiles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.mlp_precision == "fp32" else args.mlp_precision, "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[4]: Munich cloud+image, ALTO depth 6, footprint head, "
                                   f"N={args.points} points/tile, {args.batch} tile(s) per forward",
                       "parallelism": f"dp{world}", "params": sum(p.numel() for p in model.parameters()),
                       "hip_graph": bool(args.hip_graph)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU path to benchmark)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
        group = dist.group.WORLD
    torch.backends.cudnn.benchmark = bool(args.miopen_find)

    from tomosar2height_amd import TomoSAR2Height, _lib, grid
    from tomosar2height_amd.config import berlin_config, munich_config
    from tomosar2height_amd.synthetic import berlin_tile
    from tomosar2height_amd.trainer import Trainer, broadcast_parameters

    if args.mode == "infer":
        return infer_bench(args, world, rank, dev, group)

    cfg = berlin_config(use_image=args.use_image)
    torch.manual_seed(0)
    model = TomoSAR2Height(cfg).to(dev)
    model.set_mlp_precision(args.mlp_precision)
    model.set_channels_last(bool(args.channels_last))
    if world > 1:
        broadcast_parameters(model, group)
    opt = torch.optim.AdamW(model.parameters(), lr=cfg.training.learning_rate)     # train.py:97
    trainer = Trainer(model, opt, device=dev, optimize_every=args.optimize_every, use_cloud=True, use_image=args.use_image,
                      process_group=group)

    tiles = []
    for i in range(args.tile_pool):
        t = berlin_tile(seed=1000 * rank + i, n_points=args.points, clustered=not args.uniform_xy, with_image=args.use_image)
        tiles.append({k: t[k].to(dev) for k in (("inputs", "dsm", "image") if args.use_image else ("inputs", "dsm"))})

    def run(n_steps, offset=0, timeline=None, every=8):
        """`timeline`: record per-launch HIP events on every `every`-th tile-step only -- recording two events around
        each of the ~370 t2h launches of a step costs ~5 ms of host time per step, so instrumenting all K steps would
        distort the headline number by ~15 %."""
        for s in range(n_steps):
            if timeline is not None and (s % every == every - 1 or (n_steps < every and s == n_steps - 1)):
                with timeline:
                    trainer.train_step(tiles[(offset + s) % len(tiles)])
            else:
                trainer.train_step(tiles[(offset + s) % len(tiles)])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    if args.hip_graph:
        trainer.capture_graph(tiles[0])
        run(2)
    fence()
    timeline = None if (args.no_kernel_timing or args.hip_graph) else _lib.KernelTimeline()
    t0 = time.perf_counter()
    run(args.steps, args.warmup, timeline, args.timing_every)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * args.steps / elapsed
        out = {
            "metric": "training tiles/sec (Berlin crop, cloud+image)" if args.use_image else "training tiles/sec (Berlin crop, cloud-only)",
            "value": round(value, 4), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "f32 tensors, bf16-operand MFMA in the per-point GEMMs",
                      "bf16x3": "f32 tensors, per-point GEMM products via exact 3-way bf16 split (6 bf16 MFMAs)"}[args.mlp_precision],
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: Berlin cloud-only, fp32, B=1 tile, "
                                   f"N={args.points} points/tile, R=256, ALTO depth 5, 512x512 target, "
                                   f"optimize_every={args.optimize_every} (AdamW + grad all-reduce amortised)",
                       "points_per_tile": args.points, "optimize_every": args.optimize_every,
                       "parallelism": f"dp{world}", "channels_last": bool(args.channels_last),
                       "point_distribution": "uniform (no-skew control)" if args.uniform_xy else "70 % in 160 buildings + 30 % uniform",
                       "grid_convs": "t2h implicit-GEMM (csrc/conv.hip)" if (grid.USE_HIP_CONV and args.channels_last) else "MIOpen",
                       "miopen_find": bool(args.miopen_find), "hip_graph": bool(args.hip_graph)},
        }
        if timeline is not None:
            timed_steps = max(1, len([i for i in range(args.steps) if i % args.timing_every == args.timing_every - 1]))   # >= 1: see run()
            out["config"]["kernel_timing"] = f"HIP events on {timed_steps} of the {args.steps} timed steps"
            kernels = []
            for name, d in sorted(timeline.summary().items(), key=lambda kv: -kv[1]["ms"]):
                avg_us = 1e3 * d["ms"] / d["calls"]
                per_launch_b, per_launch_f = d["bytes"] / d["calls"], d["flops"] / d["calls"]
                gbs = per_launch_b / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
                tfs = per_launch_f / (avg_us * 1e-6) / 1e12 if avg_us > 0 else 0.0
                # roofline that bounds the launch: arithmetic intensity vs the machine balance (157.3 TF / 8 TB/s)
                mfma_bound = per_launch_f > 0 and per_launch_f / max(per_launch_b, 1) > MFMA_F32_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS
                kernels.append({"kernel": name, "launches_per_step": round(d["calls"] / timed_steps, 2),
                                "avg_us": round(avg_us, 2), "bytes_per_launch": int(per_launch_b),
                                "flops_per_launch": int(per_launch_f), "GBps": round(gbs, 1), "TFLOPs": round(tfs, 2),
                                "bound": "mfma" if mfma_bound else "hbm",
                                "frac": round(tfs / MFMA_F32_PEAK_TFLOPS if mfma_bound else gbs / HBM_PEAK_GBS, 4),
                                "ms_per_step": round(d["ms"] / timed_steps, 4)})

            def roof(k):
                traffic = pmc_traffic().get(k["kernel"]) if args.points == 131072 else None
                if k["bound"] == "mfma":
                    return {"kernel": k["kernel"], "bound": "mfma", "achieved": k["TFLOPs"], "peak": MFMA_F32_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": k["frac"], "traffic": traffic, "avg_us": k["avg_us"]}
                return {"kernel": k["kernel"], "bound": "hbm", "achieved": k["GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": k["frac"], "traffic": traffic, "avg_us": k["avg_us"]}

            if kernels:
                out["roofline"] = roof(kernels[0])                       # the launch with the largest time share
                named = {k["kernel"]: k for k in kernels}
                # the scatter-reduce kernels north_star names (pool_local and the largest mean)
                out["roofline_scatter_reduce"] = [roof(named[n]) for n in
                                                  ("t2h_segmean_fwd[C=512,r=32]", "t2h_pool_max_fwd") if n in named]
                # the largest grid convolutions: implicit-GEMM kernels of csrc/conv.hip
                out["roofline_grid_conv"] = [roof(named[n]) for n in
                                             ("t2h_conv3x3_fwd[64->128,512x512]", "t2h_conv3x3_dgrad[128->64,512x512]",
                                              "t2h_conv3x3_wgrad[64->128,512x512]") if n in named]
                out["t2h_kernels_ms_per_step"] = round(sum(k["ms_per_step"] for k in kernels), 3)
                out["kernels"] = kernels
        if world == 1 and not args.skip_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.points, args.cpu_threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
