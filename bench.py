#!/usr/bin/env python3
"""Headline benchmark: training tile-steps/s on Berlin-shaped synthetic tiles, cloud-only, fp32
(BASELINE.json configs[1]: N = 131072 points / tile, R = 256, ALTO depth 5, 512^2 target).

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one tile through ``Trainer.train_step``: forward, L1 loss, backward, and -- every
``optimize_every``/world tiles per rank -- one RCCL all-reduce(SUM) of the flat gradient bucket + AdamW step
(reference: trainer.py:47-89, optimize_every = 64).  Tiles are resident in HBM before the timed region
(``--from-producer``: the CHUNK is resident and every tile is cropped / normalised / augmented on the device in the loop).
Weak scaling: every rank runs K tiles; value = world * K / max-over-ranks(time).

Legs, in this order (rank 0 prints ONE compact JSON line at the very end, nothing after it):
  1. warm-up (W tiles), then the TIMED region: exactly K tile-steps, no per-launch instrumentation.  The accumulation
     phase is aligned so that the K-th tile ends an optimizer step: the region holds ceil(K / (optimize_every/world))
     all-reduce + AdamW steps -- never fewer per tile than the reference's one per 64 tiles.
  1b. sustained leg (untimed for `value`): ``--sustain-s`` seconds of back-to-back tile-steps, one HIP event per step
     -> ``sustained`` (ms/step over seconds, first vs last quartile of the per-step GPU times).
  2. profile leg (untimed): ``--profile-steps`` more tile-steps with two HIP events around every C-ABI launch, on the
     stream the kernels run on -> per-kernel table (written to ``--kernel-table``, not printed) and the ``roofline``
     objects: launches are aggregated per DEVICE KERNEL SYMBOL (t2h_last_kernel_name), the way
     ``rocprofv3 --kernel-trace --stats`` aggregates, so the two can be compared directly.
     The trainer's overlaps (weight gradients on a side stream beside the data-gradient chain; tile i + 1's forward beside tile
     i's backward -- both on in the timed region) are OFF in this leg: a kernel's two events then bracket that kernel alone
     (profiles/run_profiles.sh runs the profiler passes the same way).
  3. ``--check-dp`` (optional): the data-parallel equivalence check of SURVEY.md 8e.
  4. cpu_baseline (rank 0, N = 1 only): the oracle torch restatement on the host cores, SURVEY.md 8d protocol.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 matrix peak
# csrc/conv_bx3.hip forms an fp32 product from six bf16 MFMAs: its ceiling in ALGORITHMIC (fp32-equivalent) flops is the dense
# bf16 peak / 6 -- the same fraction as executed bf16 flops / 2.5 PF
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_BX3_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6
# the scatter-reduce kernels north_star names: the largest scatter_mean, and pool_local -- which since r02 has no kernel of
# its own: the segmented max / its backward run in the loaders of the fused trunk block kernels (csrc/trunk.hip)
# r03: with the deferred point update (deferred.py) the wide scatter_means run as per-cell SUMS of the hidden activations
# (t2h_segsum_fwd at the finest resolution + 2x2 pooling) and their joint backward
# (since r03y the coarse levels' sums are formed on chip by t2h_sample_relu_cellsums, backward t2h_sample_bwd_from_sums)
SCATTER_REDUCE_TAGS = ("t2h_sample_relu_cellsums[C=1024,r=32]", "t2h_sample_bwd_from_sums[C=1024,r=32]",
                       "t2h_segsum_fwd[C=128,r=256]",       # the per-cell sums that still run as a kernel of their own (last level)
                       "t2h_segsum_fwd[C=1024,r=256]", "t2h_segsum_fwd[C=512,r=256]", "t2h_segmean_fwd[C=512,r=32]",
                       # r06: with the micro-batches of the default Trainer the trunk forward (fc_pos, 5 blocks, the 4 pool_locals, fc_c) is ONE
                       # launch, t2h_trunk_fused_fwd; tile by tile it is the five t2h_trunk_block_fwd launches
                       "t2h_trunk_fused_fwd", "t2h_trunk_block_fwd[mid]", "t2h_trunk_block_bwd[mid]", "t2h_pool_max_fwd", "t2h_pool_max_bwd")


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
    ap.add_argument("--profile-steps", type=int, default=16,
                    help="untimed tile-steps with per-launch HIP events after the timed region (0 = no kernel table)")
    ap.add_argument("--kernel-table", default=os.path.join("gpurun_out", "bench_kernels.json"),
                    help="where rank 0 writes the full per-kernel table (never printed: the stdout line stays compact)")
    ap.add_argument("--tile-prefetch", type=int, default=0,
                    help="build the next tile's point index on a side stream during the current step (Trainer.prepare); "
                         "measured slower (12.20 vs 11.97 ms: like every two-stream overlap tried on this stack), so off")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=8, help="intra-op threads of the headline CPU baseline "
                                                                "(reference default: conf/config.yaml:20-21, train.py:77-78)")
    ap.add_argument("--cpu-max-threads", type=int, default=64, help="thread cap of the all-cores CPU leg")
    ap.add_argument("--cpu-warmup", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-budget-s", type=float, default=150.0,
                    help="stop adding timed CPU repetitions once the whole CPU leg has taken this long (>= 1 timed step "
                         "per thread setting is always taken)")
    ap.add_argument("--mlp-precision", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="fp32 = native fp32 MFMA, BASELINE configs[1] (the headline); bf16 = bf16-operand MFMA "
                         "(configs[2]); bf16x3 = fp32-grade products from 3-way bf16 splitting (opt-in experiment)")
    ap.add_argument("--hip-graph", type=int, default=0,
                    help="1 = capture the fixed-N tile step (fwd+loss+bwd) into a hipGraph and replay it")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train = the headline (configs[1]); infer = configs[4]: Munich cloud+image forward only")
    ap.add_argument("--batch", type=int, default=1, help="tiles per forward in --mode infer")
    ap.add_argument("--train-batch", type=int, default=1,
                    help="tiles of the accumulation window per forward / backward (micro-batch; a SECOND line beside the B = 1 "
                         "headline: the reference feeds tiles one at a time only because N varies, tomosar2height.yaml:40).  With "
                         "> 1 the resident tiles get DIFFERENT point counts (mean = --points) and run as ragged batches")
    ap.add_argument("--coalesce", type=int, default=-1,
                    help="tiles Trainer.train_step(tile) holds back and issues as one ragged micro-batch (Trainer.coalesce_tiles; "
                         "-1 = the Trainer's default, T2H_COALESCE_TILES or 4; 1 = the strict B = 1 step, also reported as `strict_b1`)")
    ap.add_argument("--strict-b1-steps", type=int, default=16,
                    help="tile-steps of a short timed leg with coalescing off (every train_step call issues its own tile: the B = 1 "
                         "step of rounds 1-5); reported as `strict_b1`, never part of `value` (0: skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real multi-GPU run); gloo only to smoke-test the N>1 code path "
                         "with several ranks sharing one GPU")
    ap.add_argument("--use-image", action="store_true",
                    help="train the cloud+image network (BASELINE configs[2] with --mlp-precision bf16): image U-Net encoder on")
    ap.add_argument("--uniform-xy", action="store_true", help="no-skew control: all points uniform in the tile (SURVEY 8d)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (gloo smoke test only)")
    ap.add_argument("--check-dp", type=int, default=-1,
                    help="SURVEY 8e equivalence check: the W-rank accumulated + all-reduced gradient of 2W fixed tiles vs "
                         "the same tiles accumulated by one rank alone; reports max_rel_diff, allreduce_ms, rccl_ranks.  "
                         "-1 (default): on whenever N > 1 (it runs after the timed region and costs < 1 s), 0 / 1: off / on")
    ap.add_argument("--fused-optimizer", type=int, default=1, help="1 = t2h flat-bucket AdamW kernel, 0 = torch.optim.AdamW")
    ap.add_argument("--sustain-s", type=float, default=6.0,
                    help="after the timed region: this many seconds of back-to-back tile-steps (one HIP event per step, nothing "
                         "else) -> sustained_ms_per_step and first / last quartile step time; never part of `value`.  0 = off")
    ap.add_argument("--pin-cores", type=int, default=1,
                    help="N > 1: bind every rank to its own CPU cores (NUMA-local to its GPU where sysfs tells) before the GPU is "
                         "initialised; 0 = leave the affinity alone")
    ap.add_argument("--micro-batch-steps", type=int, default=32,
                    help="tile-steps of a THIRD short timed leg with the accumulation window micro-batched: 4 tiles of different "
                         "point counts per forward / backward (Trainer.train_step([tiles]): the same accumulated gradient to fp32 "
                         "re-association); reported as `micro_batched`, never part of `value` (0: skip; only with --train-batch 1)")
    ap.add_argument("--exact-split-steps", type=int, default=16,
                    help="tile-steps of a SECOND short timed leg with the 3x3 / transposed convolutions on the exact three-way bf16 "
                         "split (six MFMAs per product: fp32 arithmetic bit for bit up to summation order) instead of the default "
                         "two-way fp16 split with block scales; reported as `exact_split`, never part of `value` (0: skip)")
    ap.add_argument("--from-producer", action="store_true",
                    help="every tile of every leg is cropped / normalised / augmented / raster-patched on the device by "
                         "producer.TileSource from a synthetic chunk resident in HBM (dataset.py:201-330) inside the loop, "
                         "instead of cycling --tile-pool prepared tiles; N varies per tile around --points")
    ap.add_argument("--chunk-tiles", type=int, default=3, help="--from-producer: the chunk is this many 512 m tiles per side")
    ap.add_argument("--producer-prefetch", type=int, default=1,
                    help="--from-producer: 1 = tile i + 1 is produced on a side stream while step i runs (the overlap the "
                         "reference gets from DataLoader workers), 0 = produce, then train, in sequence on one stream")
    return ap.parse_args()


def pmc_traffic():
    """HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), committed under
    profiles/ by profiles/collect_pmc.py for the N = 131072 workload.  NOT measured in this run (PMC collection needs
    rocprofv3 around the process): returns (bytes per launch by kernel, provenance string); ({}, None) if not collected."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return d.get("bytes_per_launch", {}), f"profiles/pmc_traffic.json ({d.get('tag', '?')})"
    except (OSError, ValueError):
        return {}, None


# entry points that are several device kernels and note no symbol of their own: HBM traffic per CALL = the launch-weighted sum of
# their kernels' PMC traffic over the entry point's calls per step (bench-pass figures of profiles/pmc_traffic.json)
ENTRY_KERNELS = {"t2h_sample_bwd_from_sums": ("sample_bwd_walk_kernel<0>", "sample_bwd_walk_kernel<1>", "sample_bwd_gather4_kernel",
                                              "sample_bwd_gather9_kernel")}


def entry_traffic(symbol, calls_per_step):
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        per, det = d["bytes_per_launch"], d["detail"]
        parts = [k for k in ENTRY_KERNELS.get(symbol, ()) if k in per and "launches_per_step" in det.get(k, {})]
        if not parts or calls_per_step <= 0:
            return None
        return int(sum(per[k] * det[k]["launches_per_step"] for k in parts) / calls_per_step)
    except (OSError, ValueError, KeyError):
        return None


def rocprof_durations():
    """Per-kernel-symbol durations of a `rocprofv3 --kernel-trace` run of THIS command (steady state, last 6 tile-steps), committed
    by profiles/run_profiles.sh as profiles/rocprof_kernels.json.  NOT measured in this run: HIP-event pairs around single launches
    over-read short and MFMA-dense kernels by 2-18 % (non-uniformly), so the line carries the profiler's figure beside its own."""
    try:
        with open(os.path.join(ROOT, "profiles", "rocprof_kernels.json")) as f:
            d = json.load(f)
        return {k: v["avg_us"] for k, v in d.get("kernels", {}).items()}, f"profiles/rocprof_kernels.json ({d.get('tag', '?')})"
    except (OSError, ValueError, KeyError):
        return {}, None


def bx3_peak(symbol: str):
    """Matrix-core ceiling of a csrc/conv_bx3.hip kernel in ALGORITHMIC flops: dense bf16 peak / 6 with the bf16 three-way split
    (NPL = 3), dense fp16 peak (= the bf16 one) / 3 with the fp16 two-way split (NPL = 2), the bf16 peak itself in the bf16 mode
    (NPL = 1); None for any other symbol."""
    if not symbol.startswith("bx3_"):
        return None
    args = symbol[symbol.find("<") + 1:symbol.rfind(">")].split(",")
    npl = args[5] if symbol.startswith("bx3_rows") and len(args) > 5 else (args[1] if len(args) > 1 else "3")
    npl = npl.strip()
    return MFMA_BF16_PEAK_TFLOPS if npl == "1" else (MFMA_BF16_PEAK_TFLOPS / 3 if npl == "2" else MFMA_BX3_PEAK_TFLOPS)


def point_update_note():
    """Which association of the ALTO point update runs (same function as alto.py:121-130, see mlp.py / deferred.py)."""
    from tomosar2height_amd import deferred, mlp
    parts = []
    if mlp.GRID_FIRST_MIN_RATIO > 0:
        parts.append(f"fc_comm.0 on pixels where >= {mlp.GRID_FIRST_MIN_RATIO:g} points/pixel")
    if deferred.DEFER_MIN_CHANNELS > 0:
        parts.append(f"fc_comm.2 / fc_c on per-cell sums from >= {deferred.DEFER_MIN_CHANNELS} channels on")
    return "; ".join(parts) if parts else "point-wise as the reference (alto.py:121-130)"


def cpu_baseline(args):
    """The oracle's torch restatement of the reference model ("port"), fwd + bwd tile-steps of the SAME workload on the
    host cores: ``--cpu-warmup`` + ``--cpu-steps`` timed repetitions, median, at the reference's 8 intra-op threads
    (the headline `value`) and at all host cores (`all_cores`).  This is the only place bench.py touches oracle/."""
    from oracle import torch_ref
    from tomosar2height_amd.config import berlin_config
    from tomosar2height_amd.synthetic import berlin_tile
    t_leg = time.perf_counter()
    host_cores = os.cpu_count() or 1
    torch.manual_seed(0)
    model = torch_ref.TomoSAR2Height(berlin_config())
    tile = berlin_tile(0, n_points=args.points)

    def one():
        model.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        torch_ref.train_loss(model, tile["inputs"], None, tile["dsm"]).backward()
        return time.perf_counter() - t0

    def leg(threads, warm, reps):
        torch.set_num_threads(threads)
        for _ in range(warm):
            one()
        times = [one()]
        while len(times) < reps and time.perf_counter() - t_leg < args.cpu_budget_s:
            times.append(one())
        med = statistics.median(times)
        return {"value": round(1.0 / med, 5), "cores": torch.get_num_threads(), "median_s": round(med, 2),
                "timed_steps": len(times), "warmup_steps": warm}

    small = berlin_tile(1, n_points=2048)      # first-touch allocations / lazy initialisation, not a measurement
    torch_ref.train_loss(model, small["inputs"], None, small["dsm"]).backward()
    main = leg(min(args.cpu_threads, host_cores), args.cpu_warmup, args.cpu_steps)
    out = {"value": main["value"], "unit": "tiles/s", "cores": main["cores"], "kind": "port",
           "sample": f"median of {main['timed_steps']} tile-steps (fwd+bwd, N={args.points}) after {main['warmup_steps']} warm-up, "
                     f"oracle torch restatement, {main['median_s']} s/step, {main['cores']} threads (reference default)"}
    if host_cores > main["cores"]:
        # "all cores", bounded: median of 3 timed steps after one warm-up at that thread count (the thread pool is
        # re-created), at most 64 threads -- with every hardware thread of a 256-thread host torch's CPU path needed 164 s
        # per tile-step (r02a), 19x slower than with 8 threads; `cores` states what was used.  --cpu-budget-s still bounds it
        allc = leg(min(host_cores, args.cpu_max_threads), 1, 3)
        out["all_cores"] = {k: allc[k] for k in ("value", "cores", "median_s", "timed_steps")}
        out["all_cores"]["host_cores"] = host_cores
    out["leg_s"] = round(time.perf_counter() - t_leg, 1)
    return out


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
            static_out = model(input_cloud=static_cloud, input_image=static_image)  # noqa: F841 (kept alive by the graph)

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


# ---------------------------------------------------------------------------------------------- kernel table
def kernel_tables(timeline, n_steps):
    """(per-tag rows, per-symbol rows) from a KernelTimeline over ``n_steps`` tile-steps, both sorted by time."""
    def row(name, d):
        peak_tf = bx3_peak(str(d.get("symbol", name))) or MFMA_F32_PEAK_TFLOPS
        avg_us = 1e3 * d["ms"] / d["calls"]
        per_b, per_f = d["bytes"] / d["calls"], d["flops"] / d["calls"]
        gbs = per_b / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
        tfs = per_f / (avg_us * 1e-6) / 1e12 if avg_us > 0 else 0.0
        # the roofline that bounds the launch: arithmetic intensity against the machine balance (157.3 TF / 8 TB/s)
        mfma = per_f > 0 and per_f / max(per_b, 1) > peak_tf * 1e3 / HBM_PEAK_GBS
        return {"kernel": name, "launches_per_step": round(d["calls"] / n_steps, 2), "avg_us": round(avg_us, 2),
                "bytes_per_launch": int(per_b), "flops_per_launch": int(per_f), "GBps": round(gbs, 1),
                "TFLOPs": round(tfs, 2), "bound": "mfma" if mfma else "hbm", "peak_tflops": round(peak_tf, 1),
                "frac": round(tfs / peak_tf if mfma else gbs / HBM_PEAK_GBS, 4),
                "ms_per_step": round(d["ms"] / n_steps, 4)}

    per_tag = timeline.summary()
    per_symbol = {}
    for name, d in per_tag.items():
        s = per_symbol.setdefault(d["symbol"], {"calls": 0, "ms": 0.0, "bytes": 0, "flops": 0, "entry_points": set(),
                                                "symbol": d["symbol"]})
        for k in ("calls", "ms", "bytes", "flops"):
            s[k] += d[k]
        s["entry_points"].add(name.split("[")[0])
    tags = sorted((dict(row(n, d), symbol=d["symbol"]) for n, d in per_tag.items()), key=lambda r: -r["ms_per_step"])
    syms = sorted((dict(row(n, d), entry_points=sorted(d["entry_points"])) for n, d in per_symbol.items()),
                  key=lambda r: -r["ms_per_step"])
    return tags, syms


def roof(k, traffic=None):
    mfma = k["bound"] == "mfma"
    return {"kernel": k["kernel"], "bound": k["bound"], "achieved": k["TFLOPs"] if mfma else k["GBps"],
            "peak": k.get("peak_tflops", MFMA_F32_PEAK_TFLOPS) if mfma else HBM_PEAK_GBS, "unit": "TFLOP/s" if mfma else "GB/s",
            "frac": k["frac"], "traffic": traffic, "avg_us": k["avg_us"], "launches_per_step": k["launches_per_step"]}


# ---------------------------------------------------------------------------------------------- DP equivalence
def check_dp(args, world, rank, dev, group, model, make_trainer):
    """W ranks x 2 fixed-seed tiles each, accumulated + all-reduced (SUM), against the same 2W tiles
    accumulated by this rank alone, in fp32 (difference = re-association only).  Also times the all-reduce of the flat
    bucket and checks that the replicas' parameters are still bit-identical."""
    from tomosar2height_amd.synthetic import berlin_tile
    n_pts = min(args.points, 32768)
    tiles = [berlin_tile(seed=7000 + i, n_points=n_pts) for i in range(2 * world)]
    tiles = [{k: t[k].to(dev) for k in ("inputs", "dsm")} for t in tiles]
    grabbed = {}

    def grab(key):
        def hook(flat):
            grabbed[key] = flat.clone()
        return hook

    null_opt = torch.optim.SGD(model.parameters(), lr=0.0)
    dp = make_trainer(null_opt, 2 * world, group)
    dp.coalesce_tiles = 1           # tile by tile on both sides: the difference is then the ranks' summation order alone (a rank's
    dp.on_reduced = grab("dp")      # 2 tiles and the single run's 2W tiles would otherwise be cut into different micro-batches)
    for i in range(2):
        dp.train_step(tiles[rank + i * world])           # rank r: tiles r, r + W  (i mod W == r)
    single = make_trainer(null_opt, 2 * world, None)
    single.coalesce_tiles = 1
    single.on_reduced = grab("single")
    for t in tiles:
        single.train_step(t)
    torch.cuda.synchronize()
    ref = grabbed["single"]
    rel = float(((grabbed["dp"] - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item())
    out = {"tiles": 2 * world, "points_per_tile": n_pts, "max_rel_diff": rel, "bucket_floats": int(ref.numel())}
    if world > 1:
        flat = dp.bucket.flat
        for _ in range(2):
            dist.all_reduce(flat, group=group)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(flat, group=group)
        torch.cuda.synchronize()
        out["allreduce_ms"] = round(1e3 * (time.perf_counter() - t0) / 5, 3)
        flat.zero_()
        psum = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        both = torch.cat([psum, -psum])
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)      # max(x) == -max(-x) on every rank <=> identical
        out["replicas_identical"] = bool((both[0] + both[1]).item() == 0.0)
    out["rccl_ranks"] = world if (world > 1 and dist.get_backend(group) == "nccl") else 0
    return out


def self_launch(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as a CHILD ``torch.distributed.run`` (never
    exec: nothing here has touched the GPU yet, and nothing will in this process), pass the command line through, stream
    the children's stderr, and re-print rank 0's JSON line as the last thing on stdout.  Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] no launcher in the environment: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)       # stderr: inherited (streams)
    line = None
    for out_line in child.stdout:
        if out_line.startswith("{"):
            line = out_line.rstrip("\n")            # the ranks print one JSON line (rank 0); keep the last one seen
        else:
            sys.stderr.write(out_line)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 or line is not None else 1


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus():
    """CPUs local to each GPU, in HIP device order, WITHOUT touching the GPU (sysfs only): the KFD topology lists the GPUs
    (nodes with SIMDs) in the order the runtime enumerates them and gives each one's PCI address, whose sysfs entry names the
    NUMA-local CPUs.  None when the topology is not readable (containers without /sys/class/kfd)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        out = []
        for node in sorted(os.listdir(base), key=int):
            props = dict(line.split()[:2] for line in open(os.path.join(base, node, "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) == 0:
                continue
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
            out.append(_parse_cpulist(open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read()))
        return out or None
    except (OSError, ValueError, KeyError):
        return None


def pin_rank(local_rank: int, local_world: int, shared_gpu: bool = False):
    """Bind this rank to its own CPU cores BEFORE anything initialises the GPU (the HIP runtime's helper threads inherit the
    mask).  A rank of the B = 1 step needs about two cores (Python issue thread + autograd's device thread + HIP runtime:
    `sustained.host_cpu_ms_per_step` ~ 13 ms per 6.7 ms step), and a rank whose threads migrate or share a core with a neighbour
    becomes the straggler every other rank waits for at the all-reduce: the known risk of the 8-GPU line.
    Cores: those NUMA-local to the rank's GPU (sysfs), shared out among the ranks whose GPUs have the same local set; without a
    readable topology an equal slice of the process's current mask.  Returns the sorted core list (also left in T2H_RANK_CPUS)."""
    avail = sorted(os.sched_getaffinity(0))
    if local_world <= 1 or len(avail) < local_world:
        return avail
    topo = None if shared_gpu else gpu_local_cpus()
    mine = None
    if topo is not None and local_rank < len(topo) and local_world <= len(topo):
        local = sorted(topo[local_rank] & set(avail))
        peers = [r for r in range(local_world) if topo[r] == topo[local_rank]]
        per = len(local) // max(len(peers), 1)
        if per >= 1:
            i = peers.index(local_rank)
            mine = local[i * per:(i + 1) * per]
    if not mine:
        per = len(avail) // local_world
        mine = avail[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    os.environ["T2H_RANK_CPUS"] = ",".join(map(str, mine))
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(len(mine), int(os.environ.get("OMP_NUM_THREADS", len(mine))))))
    return mine


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    # first thing, before any GPU call (torch.cuda.is_available() below initialises the runtime): this rank's own cores
    rank_cpus = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), shared_gpu=args.share_gpu) if args.pin_cores else None
    if rank_cpus is not None and world > 1:
        torch.set_num_threads(max(1, min(len(rank_cpus), torch.get_num_threads())))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU path to benchmark)")
    if args.share_gpu:
        local_rank = 0
        # several processes time-slicing ONE GPU (smoke test of the multi-rank code path): every extra hardware queue multiplies
        # the context switches (per-launch durations of 100 ms were measured with the side streams on) -- keep one stream per rank
        os.environ["T2H_OVERLAP_WGRAD"] = "0"
        os.environ["T2H_OVERLAP_CONV_WGRAD"] = "0"
        os.environ["T2H_PIPELINE_TILES"] = "0"
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
    from tomosar2height_amd.config import berlin_config
    from tomosar2height_amd.synthetic import berlin_tile
    from tomosar2height_amd.trainer import Trainer, broadcast_parameters

    if not args.channels_last:
        _lib.allow_library_fallback(True).set()       # the NCHW grid side is MIOpen's by definition (A/B runs only)
    if args.mode == "infer":
        return infer_bench(args, world, rank, dev, group)

    cfg = berlin_config(use_image=args.use_image)
    torch.manual_seed(0)
    model = TomoSAR2Height(cfg).to(dev)
    model.set_mlp_precision(args.mlp_precision)
    model.set_channels_last(bool(args.channels_last))
    if world > 1:
        broadcast_parameters(model, group)

    def make_optimizer():
        if args.fused_optimizer:
            try:
                from tomosar2height_amd.optim import FlatAdamW
                return FlatAdamW(model.parameters(), lr=cfg.training.learning_rate), "t2h FlatAdamW (one kernel over the flat bucket)"
            except ImportError:
                pass
        return torch.optim.AdamW(model.parameters(), lr=cfg.training.learning_rate), "torch.optim.AdamW"     # train.py:97

    def make_trainer(opt, every, grp):
        return Trainer(model, opt, device=dev, optimize_every=every, use_cloud=True, use_image=args.use_image,
                       process_group=grp)

    opt, opt_name = make_optimizer()
    trainer = make_trainer(opt, args.optimize_every, group)
    if args.coalesce > 0:
        trainer.coalesce_tiles = args.coalesce
    if args.train_batch > 1 or args.hip_graph or args.tile_prefetch or args.share_gpu:
        trainer.coalesce_tiles = 1                        # (explicit micro-batches / graph replays / prepared tiles are issued as given)
    co = trainer.coalesce_tiles

    tiles, source, anchors = [], None, None
    if args.from_producer:
        # the step right before the path: a chunk cloud (float64 world coordinates) + DSM raster (+ image)
        # resident in HBM; every tile of every leg is cropped / normalised / augmented / patched on the device in the loop
        import numpy as np
        from tomosar2height_amd.producer import RasterPatcher, TileProducer, TileSource
        from tomosar2height_amd.synthetic import berlin_chunk
        ch = berlin_chunk(seed=100 + rank, tiles_per_side=args.chunk_tiles, n_points=args.points,
                          clustered=not args.uniform_xy, with_image=args.use_image)
        source = TileSource(TileProducer(ch["points"].to(dev), z_bound=ch["z_bound"]),
                            RasterPatcher(ch["dsm"].to(dev), ch["left"], ch["top"]),
                            RasterPatcher(ch["image"].to(dev), ch["left"], ch["top"]) if args.use_image else None,
                            flip_augm=True, rotate_augm=True, rng=np.random.RandomState(7 + rank),
                            stream=torch.cuda.Stream() if args.producer_prefetch else None)
        span = 512.0 * (args.chunk_tiles - 1)
        anchors = np.floor(np.random.RandomState(11 + rank).uniform(0, span, (4096, 2))) + np.array([ch["left"], ch["bottom"]])
    else:
        ragged = (-0.10, 0.06, -0.04, 0.08)               # --train-batch > 1: point counts around --points, mean = --points
        for i in range(args.tile_pool if args.train_batch == 1 else max(args.tile_pool, 2 * args.train_batch) // 4 * 4):
            n_i = args.points if args.train_batch == 1 else int(round(args.points * (1.0 + ragged[i % 4])))
            t = berlin_tile(seed=1000 * rank + i, n_points=n_i, clustered=not args.uniform_xy, with_image=args.use_image)
            tiles.append({k: t[k].to(dev) for k in (("inputs", "dsm", "image") if args.use_image else ("inputs", "dsm"))})

    state = {"i": 0, "optimizer_steps": 0, "points": 0}

    tile_stream = torch.cuda.Stream() if (args.tile_prefetch and args.mode == "train" and not args.hip_graph) else None

    def next_tile():
        if tile_stream is None:
            return raw_tile()
        # the index of tile i (cell sort, sampling adjoint, counts) was built on a side stream while step i - 1 ran; build
        # tile i + 1's now, before step i is issued (Trainer.prepare): every tile's index is still built once per step
        ready = state.pop("indexed", None)
        if ready is None:
            ready = trainer.prepare(raw_tile(), tile_stream)
        state["indexed"] = trainer.prepare(raw_tile(), tile_stream)
        return ready

    def raw_tile():
        j = state["j"] = state.get("j", -1) + 1                       # raw tiles handed out so far
        if source is None:
            return tiles[j % len(tiles)]
        def produce(defer):
            # empty windows are skipped, as the reference's loop does (train.py:150-151); the anchors are drawn inside the chunk,
            # so with the synthetic chunk this never triggers
            for _ in range(len(anchors)):
                k = state["a"] = state.get("a", -1) + 1
                cand = source.get(anchors[k % len(anchors)], defer_wait=defer) if defer else source.get(anchors[k % len(anchors)])
                if bool(cand["is_valid"][0]):
                    return cand
            raise RuntimeError("--from-producer: no valid tile in the chunk")
        if not args.producer_prefetch:
            t = produce(False)
        else:
            # tile j was produced (on the producer's side stream) while the previous step ran; produce tile j + 1 now, before
            # step j is issued, so that its crop runs beside the steps in flight and its host read does not wait.  The main
            # stream waits for a tile only when it is USED (TileSource.wait), not when it is produced
            t = state.pop("pending", None) or produce(True)
            state["pending"] = produce(True)
            source.wait(t)
        state["points"] += t["inputs"].shape[1]
        return t

    def run(n_steps, events=None):
        """``n_steps`` TILE-steps; with --train-batch B they are issued B at a time (never across an optimizer boundary)."""
        done = 0
        while done < n_steps:
            k = min(args.train_batch, n_steps - done, trainer.local_every - trainer.accumulated_steps)
            batch = [next_tile() for _ in range(k)]
            if trainer.train_step(batch if k > 1 else batch[0]):
                state["optimizer_steps"] += 1
            state["i"] += k
            done += k
            if events is not None:
                events.append(trainer.step_event())          # (end of the last tile whose backward has been issued)
                state.setdefault("event_tiles", []).append(k)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    if args.warmup > 0:
        # one optimizer boundary belongs to the warm-up: the optimizer's lazy state (moment buffers, FlatAdamW's chunk table)
        # is created by its first step, exactly like the kernels' first launches above
        trainer.optimizer_boundary()
    if args.hip_graph:
        if trainer.pipeline_tiles:
            trainer.capture_pipeline_graphs(tiles[0])      # forward / backward graphs of the two tile streams
        else:
            trainer.capture_graph(tiles[0])
        run(2)
    # phase-align: the K-th timed tile must end an optimizer step, so the timed region contains the all-reduce + AdamW
    # work of ceil(K / local_every) optimizer steps (the driver's K = 20 would otherwise never reach the 64th tile)
    le = trainer.local_every
    trainer.accumulated_steps = (le - args.steps % le) % le
    fence()
    steps_before, points_before = state["optimizer_steps"], state["points"]
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    timed_optimizer_steps = state["optimizer_steps"] - steps_before
    timed_points = state["points"] - points_before
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- leg 1b: sustained run (never part of `value`): does the rate of the short timed region hold over seconds?
    sustained = None
    if args.sustain_s > 0 and not args.hip_graph:
        n_sus = max(8, int(round(args.sustain_s / (elapsed / args.steps))))       # same count on every rank (elapsed is the max)
        evs = [torch.cuda.Event(enable_timing=True)]
        state["event_tiles"] = []
        fence()
        evs[0].record()
        ts0 = time.perf_counter()
        cpu0 = time.process_time()
        run(n_sus, evs)
        issue_s, issue_cpu_s = time.perf_counter() - ts0, time.process_time() - cpu0
        fence()
        sus_elapsed = time.perf_counter() - ts0
        if world > 1:
            tmax = torch.tensor([sus_elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            sus_elapsed = float(tmax.item())
        calls = state.pop("event_tiles")
        # GPU time per tile between call ends; with coalescing only every co-th call issues work: groups of co calls
        g = max(1, trainer.coalesce_tiles)
        per = [evs[i].elapsed_time(evs[min(i + g, len(calls))]) / sum(calls[i:i + g]) for i in range(0, len(calls), g)]
        q = max(1, len(per) // 4)
        state.pop("event_tiles", None)
        sustained = {"steps": n_sus, "seconds": round(sus_elapsed, 2), "ms_per_step": round(1e3 * sus_elapsed / n_sus, 3),
                     "tiles_per_s": round(world * n_sus / sus_elapsed, 3),
                     "first_quartile_ms": round(sum(per[:q]) / q, 3), "last_quartile_ms": round(sum(per[-q:]) / q, 3),
                     "median_ms": round(statistics.median(per), 3), "max_ms": round(max(per), 3),
                     "optimizer_steps": n_sus // trainer.local_every,
                     # host side: wall time until the last launch was issued (== the whole leg when the host or the HIP queue's
                     # depth limit paces the run) and CPU time of this process per step (Python + HIP runtime, all threads)
                     "host_issue_ms_per_step": round(1e3 * issue_s / n_sus, 3),
                     "host_cpu_ms_per_step": round(1e3 * issue_cpu_s / n_sus, 3)}
        # what the host alone needs per tile-step: issue time of single steps onto an EMPTY queue (no back-pressure); well
        # below ms_per_step = the run is GPU-bound and a rank needs that fraction of one core
        lone = []
        g = max(1, trainer.coalesce_tiles)
        trainer.flush_pipeline()
        trainer.accumulated_steps = trainer.accumulated_steps // g * g       # (whole groups: a group is one issue)
        lone_cpu = []
        for _ in range(8):
            fence()
            th, tc = time.perf_counter(), time.process_time()
            run(g)
            lone.append((time.perf_counter() - th) / g)
            lone_cpu.append((time.process_time() - tc) / g)
        fence()
        sustained["host_issue_ms_empty_queue"] = round(1e3 * statistics.median(lone), 3)
        # CPU time of this process (all threads) while issuing onto an empty queue: what the step costs the host when nothing
        # blocks.  `host_cpu_ms_per_step` above is taken while the queue is full: it then mostly counts the HIP runtime's
        # spin-waits for queue space (about two threads' worth of the wall time), not work
        sustained["host_cpu_ms_empty_queue"] = round(1e3 * statistics.median(lone_cpu), 3)

    # ---- leg 1c: the same step with the convolutions in the EXACT split arithmetic (never part of `value`): what `dtype: f32`
    # costs when no block-floating-point caveat is accepted (DESIGN 4.1a vs 4.1b)
    exact_split = None
    if args.exact_split_steps > 0 and args.mode == "train" and not args.hip_graph and grid.CONV_PRECISION == "f16x2":
        trainer.flush_pipeline()
        grid.set_conv_precision("bf16x3")
        try:
            run(6)                                            # (first use prepares the three-way split weights)
            trainer.flush_pipeline()                          # (tiles the Trainer still holds for its next micro-batch are issued on
            fence()                                           #  both sides: the timed region runs exactly its own tiles)
            te = time.perf_counter()
            run(args.exact_split_steps)
            trainer.flush_pipeline()
            fence()
            es = time.perf_counter() - te
            if world > 1:
                tmax = torch.tensor([es], device=dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                es = float(tmax.item())
            exact_split = {"value": round(world * args.exact_split_steps / es, 3), "ms_per_step": round(1e3 * es / args.exact_split_steps, 3),
                           "steps": args.exact_split_steps,
                           "arithmetic": "bf16x3: convolutions + wide grid-side products on the exact 3-way bf16 split, 6 MFMAs per product"}
        finally:
            trainer.flush_pipeline()
            grid.set_conv_precision(None)
            run(2)                                            # back on the default arithmetic before the next leg
            fence()

    # ---- leg 1c': coalescing off (never part of `value`): every train_step(tile) issues its own forward / backward -- the B = 1 step
    # that was the headline of rounds 1-5
    strict_b1 = None
    if args.strict_b1_steps > 0 and args.mode == "train" and not args.hip_graph and trainer.coalesce_tiles > 1:
        trainer.flush_pipeline()
        keep_co, trainer.coalesce_tiles = trainer.coalesce_tiles, 1
        try:
            run(6)
            fence()
            tb = time.perf_counter()
            run(args.strict_b1_steps)
            fence()
            eb1 = time.perf_counter() - tb
            if world > 1:
                tmax = torch.tensor([eb1], device=dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                eb1 = float(tmax.item())
            strict_b1 = {"value": round(world * args.strict_b1_steps / eb1, 3), "ms_per_step": round(1e3 * eb1 / args.strict_b1_steps, 3),
                         "steps": args.strict_b1_steps, "what": "T2H_COALESCE_TILES=1: one forward / backward per train_step call"}
        finally:
            trainer.flush_pipeline()
            trainer.coalesce_tiles = keep_co
            fence()

    # ---- leg 1d: the same accumulation window micro-batched (never part of `value`): 4 ragged tiles per forward / backward.  The
    # reference runs its 64 tiles one at a time only because their point counts differ (tomosar2height.yaml:40); they are
    # independent and their gradients are summed (trainer.py:69-89), so this is the same training step with a quarter of the
    # launches per tile -- what takes the host out of a B = 1 step
    micro_batched = None
    if (args.micro_batch_steps > 0 and args.mode == "train" and args.train_batch == 1 and not args.hip_graph and source is None
            and tile_stream is None and args.optimize_every % (4 * world) == 0 and trainer.coalesce_tiles == 1):
        trainer.flush_pipeline()
        keep_tiles, keep_j = list(tiles), state.get("j", -1)
        rag = (-0.10, 0.06, -0.04, 0.08)
        pool = []
        for i in range(8):
            t = berlin_tile(seed=1000 * rank + 500 + i, n_points=int(round(args.points * (1.0 + rag[i % 4]))),
                            clustered=not args.uniform_xy, with_image=args.use_image)
            pool.append({k: t[k].to(dev) for k in (("inputs", "dsm", "image") if args.use_image else ("inputs", "dsm"))})
        tiles[:] = pool
        args.train_batch = 4
        try:
            n_mb = max(4, args.micro_batch_steps // 4 * 4)
            trainer.accumulated_steps = trainer.accumulated_steps // 4 * 4        # (micro-batches never cross an optimizer boundary)
            run(16)
            fence()
            tm = time.perf_counter()
            run(n_mb)
            fence()
            em = time.perf_counter() - tm
            if world > 1:
                tmax = torch.tensor([em], device=dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                em = float(tmax.item())
            micro_batched = {"value": round(world * n_mb / em, 3), "ms_per_tile": round(1e3 * em / n_mb, 3), "tiles": n_mb,
                             "tiles_per_forward": 4, "points": "ragged, -10..+8 % around --points"}
        finally:
            args.train_batch = 1
            tiles[:] = keep_tiles
            state["j"] = keep_j
            run(2)
            fence()

    # ---- leg 2: per-launch HIP events (every rank runs it: the optimizer boundaries inside are collective)
    timeline = None
    if args.profile_steps > 0 and not args.hip_graph:
        timeline = _lib.KernelTimeline()
        # per-kernel durations are those of kernels running ALONE: the overlaps of the timed step (weight gradients on a side
        # stream, the next tile's forward beside this tile's backward) are switched off for this untimed leg, so a kernel's events
        # do not span another kernel's time on shared CUs.  (The timed region above keeps them on; the sum of these durations
        # therefore exceeds ms_per_step.)
        saved_overlap = (trainer.overlap_wgrad, trainer.overlap_conv_wgrad, trainer.pipeline_tiles)
        trainer.flush_pipeline()
        trainer.overlap_wgrad = trainer.overlap_conv_wgrad = trainer.pipeline_tiles = False
        try:
            with timeline:
                run(args.profile_steps)
            fence()
        finally:
            trainer.overlap_wgrad, trainer.overlap_conv_wgrad, trainer.pipeline_tiles = saved_overlap
    # one optimizer boundary on its own (all-reduce + AdamW + bucket zero), between two events on the compute stream
    fence()
    eb, ee = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eb.record()
    trainer.optimizer_boundary()
    ee.record()
    fence()
    boundary_ms = eb.elapsed_time(ee)

    dp = None
    if args.check_dp == 1 or (args.check_dp < 0 and world > 1):
        try:
            dp = check_dp(args, world, rank, dev, group, model, make_trainer)
            if args.backend == "nccl" and world > 1 and dp.get("rccl_ranks") != world:
                raise AssertionError(f"--backend nccl with {world} ranks, but the gradient all-reduce ran on {dp.get('rccl_ranks')} RCCL ranks")
        except Exception as e:          # the headline line must survive a failing diagnostic (every rank fails alike or the
            dp = {"error": f"{type(e).__name__}: {e}"[:300]}      # collective inside raises on all of them)
    per_rank = None
    if world > 1:
        # what every rank's HOST did (the scaling risk of the line: DESIGN section 6): its cores and its CPU time per tile-step
        mine = {"rank": rank, "cpus": os.environ.get("T2H_RANK_CPUS", ""),
                "host_cpu_ms_per_step": None if sustained is None else sustained["host_cpu_ms_per_step"],
                "host_issue_ms_per_step": None if sustained is None else sustained["host_issue_ms_per_step"]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine, group=group)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * args.steps / elapsed
        out = {
            "metric": "training tiles/sec (Berlin crop, cloud+image)" if args.use_image else "training tiles/sec (Berlin crop, cloud-only)",
            "value": round(value, 4), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "f32 tensors, bf16-operand MFMA in the per-point GEMMs and the 3x3 convolutions",
                      "bf16x3": "f32 tensors, per-point GEMM products via exact 3-way bf16 split (6 bf16 MFMAs)"}[args.mlp_precision],
            "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[2]: Berlin cloud+image" if args.use_image
                                    else "BASELINE.json configs[1]: Berlin cloud-only")
                                   + f", {args.mlp_precision} per-point GEMMs, "
                                   + ((f"train_step(tile) per tile, {co} tiles coalesced per forward/backward (Trainer default), "
                                       if co > 1 else "B=1 tile, ") if args.train_batch == 1 else
                                      f"micro-batches of {args.train_batch} tiles with different point counts (ragged; SECOND line, not the "
                                      "B=1 headline), mean ")
                                   + f"N={args.points} points/tile, R=256, ALTO depth 5, 512x512 target, "
                                   f"optimize_every={args.optimize_every}",
                       "points_per_tile": args.points, "optimize_every": args.optimize_every, "train_batch": args.train_batch,
                       "coalesce_tiles": co,
                       "optimizer_steps_in_timed_region": timed_optimizer_steps, "optimizer": opt_name,
                       "optimizer_boundary_ms": round(boundary_ms, 3),
                       "parallelism": f"dp{world}", "collective": (dist.get_backend(group) if world > 1 else None),
                       "rccl_ranks": world if (world > 1 and dist.get_backend(group) == "nccl") else 0,
                       "channels_last": bool(args.channels_last),
                       "point_distribution": "uniform (no-skew control)" if args.uniform_xy else "70 % in 160 buildings + 30 % uniform",
                       "grid_convs": ({"bf16x3": "t2h csrc/conv_bx3.hip: every fp32 product from six bf16 MFMAs (exact 3-way operand "
                                                 "split, fp32 accumulate; error vs float64 = the fp32 MFMA kernels'), planes >= 32 wide; "
                                                 "csrc/conv.hip (fp32 MFMA) for the rest",
                                       "f16x2": "conv_bx3.hip: fp32 products from 3 fp16 MFMAs (2-way split, block scales)",
                                       "bf16": "t2h csrc/conv_bx3.hip, operands rounded to bf16 (one MFMA per product, fp32 accumulate)",
                                       "fp32": "t2h implicit-GEMM on fp32 MFMA (csrc/conv.hip)"}[grid.CONV_PRECISION]
                                      if (grid.USE_HIP_CONV and args.channels_last) else "MIOpen"),
                       "point_update": point_update_note(),
                       "library_fallbacks": getattr(grid, "fallback_count", lambda: None)(),
                       "hip_graph": bool(args.hip_graph)},
        }
        if args.mlp_precision != "fp32":
            from tomosar2height_amd import mlp
            out["config"]["trunk_precision"] = mlp.trunk_precision()
        if args.mlp_precision == "bf16":
            # BASELINE configs[2] ("bf16 MLP GEMMs on MFMA") is kept as an ARITHMETIC, retired as a speed mode: the default line
            # already runs its convolutions and wide grid-side products on the 16-bit matrix cores (fp16 two-way split), what this
            # mode still changes are the per-point GEMMs of three levels, whose A operand is produced per tile in fp32 -- measured
            # 99.0 vs 101.0 tiles/s (r04) against the fp32-result line of the same configuration: no gain to report
            out["config"]["mode_note"] = ("configs[2] arithmetic (operands rounded to bf16, fp32 accumulate); not a speed mode: the "
                                          "default fp32-result line already runs the 16-bit matrix cores (DESIGN.md section 5)")
        if timeline is not None:
            tags, syms = kernel_tables(timeline, args.profile_steps)
            traffic, traffic_src = pmc_traffic() if (args.points == 131072 and not args.from_producer) else ({}, None)
            named = {k["kernel"]: k for k in tags}
            if syms:
                top_traffic = traffic.get(syms[0]["kernel"])
                if top_traffic is None and traffic:
                    top_traffic = entry_traffic(syms[0]["kernel"], syms[0]["launches_per_step"])
                out["roofline"] = roof(syms[0], top_traffic)                          # the kernel symbol with the largest time share
                prof_us, prof_src = rocprof_durations()
                if syms[0]["kernel"] in prof_us and args.points == 131072 and args.train_batch == 1 and not args.use_image:
                    # the profiler's duration of the same symbol (committed trace of the same command): `frac` stays on the
                    # conservative HIP-event figure, `frac_rocprof` is what rocprofv3 --stats readers will recompute
                    us = prof_us[syms[0]["kernel"]]
                    per = syms[0]["flops_per_launch"] / 1e12 if syms[0]["bound"] == "mfma" else syms[0]["bytes_per_launch"] / 1e9
                    out["roofline"]["avg_us_rocprof"] = us
                    out["roofline"]["frac_rocprof"] = round(per / (us * 1e-6) / out["roofline"]["peak"], 4)
                    out["roofline"]["rocprof_source"] = prof_src
                out["roofline"]["entry_points"] = syms[0]["entry_points"]
                out["roofline"]["traffic_source"] = traffic_src if out["roofline"]["traffic"] is not None else None
                out["roofline"]["how"] = f"HIP events per launch, {args.profile_steps} untimed steps"
                # the scatter-reduce kernels north_star names (pool_local and the largest mean)
                # (compact: peak and unit are those of `bound` -- 8000 GB/s for hbm, 157.3 TFLOP/s fp32 MFMA for mfma)
                out["roofline_scatter_reduce"] = [{k: v for k, v in roof(named[n], traffic.get(n, traffic.get(named[n]["symbol"]))).items()
                                                   if k not in ("peak", "unit", "launches_per_step")}
                                                  for n in SCATTER_REDUCE_TAGS if n in named]
                out["roofline_top_symbols"] = [{"kernel": s["kernel"][:60], "ms_per_step": s["ms_per_step"], "frac": s["frac"],
                                                "bound": s["bound"]} for s in syms[:3]]          # (the stdout line stays < 4 KB: the full table is in the file)
                out["t2h_kernels_ms_per_step"] = round(sum(k["ms_per_step"] for k in tags), 3)
                # sum of the kernels' durations ALONE over the time of a timed step: > 1 is what the step's overlaps (side stream,
                # tile pipeline) hide, plus whatever the per-launch event pairs of the profile leg over-read (2-3 %)
                out["event_inflation"] = round(out["t2h_kernels_ms_per_step"] / ms_per_step, 4)
                out["t2h_launches_per_step"] = round(sum(k["launches_per_step"] for k in tags), 1)
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.kernel_table)), exist_ok=True)
                with open(args.kernel_table, "w") as f:
                    json.dump({"workload": out["config"]["workload"], "profile_steps": args.profile_steps,
                               "by_entry_point": tags, "by_kernel_symbol": syms}, f, indent=1)
                out["config"]["kernel_table"] = args.kernel_table
            except OSError as e:
                out["config"]["kernel_table"] = f"not written: {e}"
        if exact_split is not None:
            out["exact_split"] = exact_split
        if strict_b1 is not None:
            out["strict_b1"] = strict_b1
        if micro_batched is not None:
            out["micro_batched"] = micro_batched
        if sustained is not None:
            out["sustained"] = sustained
            out["sustained_ms_per_step"] = sustained["ms_per_step"]
        if args.from_producer:
            out["config"]["tile_source"] = (f"producer.TileSource in the loop: crop + normalise + rot/flip augmentation + DSM patch "
                                            f"per tile from a {args.chunk_tiles}x{args.chunk_tiles}-tile chunk resident in HBM"
                                            f"{', next tile produced on a side stream during the step' if args.producer_prefetch else ''}; "
                                            f"mean N = {timed_points / max(args.steps, 1):.0f} points/tile in the timed region")
            out["config"]["workload"] += ", tiles from the device tile producer (dataset.py:201-330)"
        if dp is not None:
            out["check_dp"] = dp
        if per_rank is not None:
            sets = [set(r["cpus"].split(",")) - {""} for r in per_rank]
            out["ranks"] = {"host_cpu_ms_per_step": [r["host_cpu_ms_per_step"] for r in per_rank],
                            "host_issue_ms_per_step": [r["host_issue_ms_per_step"] for r in per_rank],
                            "cores_per_rank": [len(c) for c in sets],
                            "affinity_disjoint": bool(all(sets)) and all(not (sets[i] & sets[j]) for i in range(world)
                                                                         for j in range(i + 1, world)),
                            "first_cpu": [min((int(c) for c in st), default=-1) for st in sets]}
        if world == 1 and not args.skip_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)          # last: the GPU legs above run back to back
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
