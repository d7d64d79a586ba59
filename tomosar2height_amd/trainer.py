"""Training / evaluation step (reference: trainer.py:8-146) plus tile-level data parallelism.

``Trainer.train_step(data)`` keeps the reference contract: L1(mean) on heights (+ ``weight_ce`` x
BCE-with-logits on ``dsm > 1e-4`` with a footprint head), ``loss.backward()``, gradients SUMMED over
``optimize_every`` tiles (no 1/k scaling, trainer.py:69-89), then one optimizer step.

Data parallel (an addition: the reference is single-device): with ``process_group`` set, each of
the W ranks runs ``optimize_every / W`` of the tiles of one optimizer step, and one RCCL all-reduce(SUM) of a
single flat fp32 gradient bucket precedes ``optimizer.step()``.  SUM, not mean: it reproduces the
single-device accumulated gradient.  Parameters that never receive a gradient
(``up_convs[depth-2].{upconv,fc_comm,fc_c}``, alto.py:241-242) are a static set and stay out of the bucket, so
AdamW skips them exactly as it does in the reference.
"""
import contextlib
import os
from collections import defaultdict

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, mlp


def _is_dense(t: torch.Tensor) -> bool:
    """True if t's elements tile a contiguous block of memory exactly once (some permutation of a contiguous tensor)."""
    expected = 1
    for size, stride in sorted(zip(t.shape, t.stride()), key=lambda ss: ss[1]):
        if size == 1:
            continue
        if stride != expected:
            return False
        expected *= size
    return True


class GradBucket:
    """One contiguous fp32 buffer holding every live gradient; ``p.grad`` are views into it."""

    def __init__(self, params):
        self.params = [p for p in params if p.grad is not None]
        # every view starts on a 16-byte boundary (4 floats): the accumulating reductions of the weight-gradient
        # kernels then always take their float4 path, whatever odd-sized parameter (conv4.bias has 1 element) precedes
        total = sum(-(-p.numel() // 4) * 4 for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.grad.dtype, device=ref.grad.device)
        off = 0
        for p in self.params:
            n = p.numel()
            g = p.grad
            # keep the gradient's own dense layout (channels_last conv weights get channels_last grads: autograd's
            # "gradient layout contract"), so later accumulations are plain contiguous adds
            if g.is_contiguous() or not _is_dense(g):
                view = self.flat[off:off + n].view_as(p)
            else:
                view = torch.as_strided(self.flat, g.size(), g.stride(), storage_offset=off)
            view.copy_(g)
            p.grad = view
            off += -(-n // 4) * 4

    def all_reduce(self, group):
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)

    def zero_(self):
        self.flat.zero_()


class Trainer:
    def __init__(self, model: nn.Module, optimizer, device=None, optimize_every=1, use_cloud=False, use_image=False,
                 use_footprint=False, weight_ce=10., process_group=None, check_domain=True, scheduler=None):
        """``check_domain``: at every optimizer step (one device read per ``optimize_every`` tiles) raise if any input
        point was outside [0, 1)^2 -- the reference's scatter would index out of range for such a point."""
        self.model = model
        self.optimizer = optimizer
        self.device = device
        self.loss_ce = nn.BCEWithLogitsLoss(reduction="mean")
        self.loss_l1 = nn.L1Loss(reduction="mean")
        self.weight_ce = weight_ce
        self.optimizer.zero_grad()

        self.check_domain = check_domain
        self.scheduler = scheduler        # stepped once per optimizer step, as train.py:188-190 does with CyclicLR
        self.on_reduced = None
        self.group = process_group
        self.world = dist.get_world_size(process_group) if process_group is not None else 1
        if optimize_every % self.world:
            raise ValueError(f"optimize_every={optimize_every} must be a multiple of the world size {self.world}")
        self.optimize_every = optimize_every
        self.local_every = optimize_every // self.world      # tiles per rank per optimizer step
        self.bucket = None
        self._graph = None
        self.direct_accumulation = os.environ.get("T2H_DIRECT_ACCUM", "1") != "0"
        # weight-gradient GEMMs on a side stream (see mlp.direct_grad_accumulation), joined at the end of every train_step's
        # backward (T2H_OVERLAP_WGRAD=0: A/B).  On by default since r04: with the fp16-split convolutions most launches of a
        # B = 1 step leave CUs idle (8.94 -> 8.71 ms with both overlaps).  bench.py switches them off for its untimed per-kernel leg
        # (--profile-steps), so the durations of its kernel table stay those of kernels running alone
        self.overlap_wgrad = os.environ.get("T2H_OVERLAP_WGRAD", "1") == "1"
        self._side = None
        # the weight gradients of the convolutions on planes up to 128 x 128 (latency-bound launches that leave most CUs idle)
        # beside the following layers' data gradients: T2H_OVERLAP_CONV_WGRAD=1
        self.overlap_conv_wgrad = os.environ.get("T2H_OVERLAP_CONV_WGRAD", "1") == "1"
        self._conv_side = None
        # the tiles of an accumulation window are independent: tile i + 1's forward runs beside tile i's backward, each tile on
        # its own stream of a ping-pong pair (see ``_train_step_pipelined``); T2H_PIPELINE_TILES=0: one tile after the other
        self.pipeline_tiles = os.environ.get("T2H_PIPELINE_TILES", "1") == "1"
        # ... and so do the micro-batches of a window (T2H_PIPELINE_MICRO_BATCHES=0: a micro-batch runs on the caller's stream alone)
        self.pipeline_micro_batches = os.environ.get("T2H_PIPELINE_MICRO_BATCHES", "1") == "1"
        self._tile_streams = None
        self._pending = None            # (loss, l1, ce, stream, weights version, graph set) of the tile whose backward is held back
        self._pipe_graphs = None        # capture_pipeline_graphs: forward / backward hipGraphs of the two tile streams
        self._bwd_done = None           # event: end of the last issued backward (the next one accumulates into the same buffers)
        self._tile_parity = 0
        self._held = []                 # (event at the end of a tile's backward, that tile's ``data``): see _hold_inputs
        # Single tiles handed to ``train_step`` one by one are held back and issued ``coalesce_tiles`` at a time as ONE ragged
        # micro-batch (``_losses_micro_batch``): the tiles of an accumulation window are independent and their gradients are summed
        # (trainer.py:69-89), the reference feeds them one at a time only because their point counts differ
        # (tomosar2height.yaml:40), and a B = 1 step is ~340 launches most of which leave the chip idle (r05: 144 -> 174 tiles/s
        # with the same kernels).  Never across an optimizer boundary; ``flush_gradients()`` / ``optimizer_boundary()`` issue what
        # is held.  T2H_COALESCE_TILES=1: every tile issued by its own call (the strict B = 1 step).
        self.coalesce_tiles = max(1, int(os.environ.get("T2H_COALESCE_TILES", "4")))
        self._coalesced = []            # tiles accepted by train_step, not yet issued
        self._coalesced_version = None  # the weights' version counter when the first of them was accepted

        # The composed weight maps of the deferred ALTO levels depend on the weights only: computed once per optimizer step,
        # their gradient accumulated over the step's tiles and back-propagated once (deferred.ComposeCache).  Parameter
        # gradients are therefore complete after ``flush_gradients()`` / at ``optimizer_boundary()``, not after every tile.
        # Ownership: the cache belongs to THIS trainer and is visible to the network only while one of its own steps runs
        # (``_own_cache``): a forward / backward issued by anybody else -- a second Trainer on the same model, a plain
        # ``model(...).backward()`` -- never routes gradients into it and sees complete ``.grad`` (plain autograd).
        self.compose_cache = None
        self._unet = getattr(getattr(model, "point_encoder", None), "unet", None)
        if os.environ.get("T2H_COMPOSE_CACHE", "1") != "0" and self._unet is not None and hasattr(self._unet, "forward_sorted"):
            from . import deferred
            self.compose_cache = deferred.ComposeCache()

        enc = getattr(model, "point_encoder", None)
        if check_domain and enc is not None and hasattr(enc, "snapshot_domain"):
            enc.snapshot_domain = True          # the boundary's out-of-domain read waits for the last tile's index only (pointnet.py)
        self._flag_stream = None
        self.accumulated_steps = 0
        self.accumulated_loss = 0.0
        self.accumulated_loss_dict = {"loss_ce": 0.0, "loss_l1": 0.0}
        self.last_avg_loss = 0.0
        self.last_avg_loss_dict = {"loss_ce": 0.0, "loss_l1": 0.0}
        self.use_cloud, self.use_image, self.use_footprint = use_cloud, use_image, use_footprint

    # ------------------------------------------------------------------------------------------ compose cache
    @contextlib.contextmanager
    def _own_cache(self):
        """Make this trainer's ComposeCache the one the ALTO U-Net uses for the forward passes issued inside (their backward
        keeps the cache the forward saw).  Refuses to run inside another trainer's step on the same network."""
        if self.compose_cache is None:
            yield
            return
        prev = getattr(self._unet, "compose_cache", None)
        if prev is not None and prev is not self.compose_cache:
            raise RuntimeError("Trainer: another Trainer's step is active on this network (nested train_step)")
        self._unet.compose_cache = self.compose_cache
        try:
            yield
        finally:
            self._unet.compose_cache = prev

    # ------------------------------------------------------------------------------------------ loss
    def _losses(self, data, mask_threshold):
        device = self.device
        if isinstance(data, (list, tuple)):
            return self._losses_micro_batch(data, mask_threshold)
        input_cloud = data.get("inputs").to(device) if self.use_cloud else None
        input_image = data.get("image").to(device) if self.use_image else None
        dsm_gt = data.get("dsm").to(device)[None, ...]
        pa, pb = self.model(input_cloud=input_cloud, input_image=input_image)
        loss_l1 = self.loss_l1(pa.squeeze(), dsm_gt.squeeze().float())
        if self.use_footprint:
            loss_ce = self.weight_ce * self.loss_ce(pb.squeeze(), (dsm_gt.squeeze() > mask_threshold).float())
        else:
            loss_ce = torch.zeros((), device=device)      # (a fill kernel: capturable, unlike torch.tensor(0.0))
        return loss_l1, loss_ce

    def _losses_micro_batch(self, tiles, mask_threshold):
        """Several tiles of one accumulation window in ONE forward / backward.  The reference runs them one by one only because
        their point counts differ (tomosar2height.yaml:40); they are independent, and it SUMS their gradients (trainer.py:69-89),
        so the loss of the micro-batch is the SUM of the per-tile mean losses: every tile has the same number of pixels, hence
        sum_b mean_b = B x (mean over the whole batch).  The clouds go to the model as a list -> one ragged ``TileIndex``."""
        device, nb = self.device, len(tiles)
        clouds = [t.get("inputs").to(device) for t in tiles] if self.use_cloud else None
        if clouds is not None and len({c.shape[1] for c in clouds}) == 1:
            clouds = torch.cat(clouds, 0)                  # equal point counts: the plain [B, N, 3] batch
        image = torch.cat([t.get("image").to(device) for t in tiles], 0) if self.use_image else None
        dsm_gt = torch.cat([t.get("dsm").to(device).reshape(1, *t.get("dsm").shape[-2:]) for t in tiles], 0).float()
        pa, pb = self.model(input_cloud=clouds, input_image=image)
        loss_l1 = nb * self.loss_l1(pa.reshape(dsm_gt.shape), dsm_gt)
        if self.use_footprint:
            loss_ce = nb * self.weight_ce * self.loss_ce(pb.reshape(dsm_gt.shape), (dsm_gt > mask_threshold).float())
        else:
            loss_ce = torch.zeros((), device=device)
        return loss_l1, loss_ce

    # ------------------------------------------------------------------------------------------ hipGraph
    def capture_graph(self, example):
        """Capture forward + loss + backward of one tile into a hipGraph (``torch.cuda.CUDAGraph``) for tiles of
        ``example``'s shapes.  Later ``train_step`` calls with the same shapes copy the tile into static buffers and
        replay the ~470 launches with one host call; other shapes (real tiles have varying N) run eagerly.  Needs the
        gradient bucket, i.e. at least one eager ``train_step`` first.  In channels_last mode the replay is bit-identical to
        eager execution (with the NCHW / MIOpen grid side: up to MIOpen's run-to-run conv-wgrad rounding).  At N = 131072
        the step is GPU-bound (measured: no gain); it pays for small tiles."""
        if self.bucket is None:
            raise RuntimeError("capture_graph: run one eager train_step first (the gradient bucket must exist)")
        self.flush_pipeline()
        self._snapshots_off()
        dev = self.device
        static = {k: example[k].to(dev).clone() for k in ("inputs", "image", "dsm") if example.get(k) is not None}
        self.model.train()
        saved = self.bucket.flat.clone()                   # warm-up / capture passes must not pollute the accumulators
        saved_cache = self.compose_cache.snapshot() if self.compose_cache is not None else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with self._own_cache():
            with torch.cuda.stream(side):                  # warm-up on a side stream, as torch.cuda.graphs prescribes
                for _ in range(2):
                    l1, ce = self._losses(static, 0.0001)
                    with mlp.direct_grad_accumulation(self.direct_accumulation):
                        (l1 + ce).backward()
            del l1, ce                                     # drop the warm-up autograd graph (and its AccumulateGrad nodes)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):     # capture on the stream the warm-up ran on
                l1, ce = self._losses(static, 0.0001)
                with mlp.direct_grad_accumulation(self.direct_accumulation), _lib.reduce_capture(self.direct_accumulation):
                    (l1 + ce).backward()
        self.bucket.flat.copy_(saved)
        if saved_cache is not None:
            self.compose_cache.restore(saved_cache)
        self._graph = {"graph": graph, "static": static, "l1": l1.detach(), "ce": ce.detach(),
                       "shapes": {k: tuple(v.shape) for k, v in static.items()}}
        return graph

    def capture_pipeline_graphs(self, example):
        """The tile pipeline (``_train_step_pipelined``) as hipGraphs: for each stream of the ping-pong pair ONE graph of the
        forward (+ loss) and ONE of the backward of a tile of ``example``'s shapes, captured on that stream and sharing a memory
        pool (the backward graph reads the activations the forward graph leaves in its static buffers).  ``train_step`` then
        replays forward(i) before backward(i - 1) exactly as the eager pipeline issues them -- same kernels, same order per
        stream, the same events between the streams -- with four host calls per tile instead of ~330: the GPU side of the
        pipeline (6.7 ms per tile) no longer depends on how fast the host issues.  Tiles of other shapes (real tiles have
        varying N) run eagerly.  Needs the gradient bucket, i.e. at least one eager ``train_step`` first; the weight gradients
        stay on the tile's stream inside the graphs (graph branches measured slower than a linear graph on this stack)."""
        if self.bucket is None:
            raise RuntimeError("capture_pipeline_graphs: run one eager train_step first (the gradient bucket must exist)")
        self._snapshots_off()
        self.flush_pipeline()
        dev = self.device
        main = torch.cuda.current_stream(dev)
        if self._tile_streams is None:
            self._tile_streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        self.model.train()
        saved = self.bucket.flat.clone()                   # warm-up / capture passes must not pollute the accumulators
        saved_cache = self.compose_cache.snapshot() if self.compose_cache is not None else None
        sets = []
        for st in self._tile_streams:
            static = {k: example[k].to(dev).clone() for k in ("inputs", "image", "dsm") if example.get(k) is not None}
            st.wait_stream(main)
            with self._own_cache():
                with torch.cuda.stream(st):                # warm-up on the capture stream, as torch.cuda.graphs prescribes
                    for _ in range(2):
                        l1, ce = self._losses(static, 0.0001)
                        with mlp.direct_grad_accumulation(self.direct_accumulation):
                            (l1 + ce).backward()
                del l1, ce
                main.wait_stream(st)
                gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(gf, stream=st):
                    l1, ce = self._losses(static, 0.0001)
                    loss = l1 + ce
                with torch.cuda.graph(gb, stream=st, pool=gf.pool()):
                    with mlp.direct_grad_accumulation(self.direct_accumulation), _lib.reduce_capture(self.direct_accumulation):
                        loss.backward()
            main.wait_stream(st)
            sets.append({"fwd": gf, "bwd": gb, "static": static, "l1": l1.detach(), "ce": ce.detach(), "loss": loss.detach()})
        self.bucket.flat.copy_(saved)
        if saved_cache is not None:
            self.compose_cache.restore(saved_cache)
        self._pipe_graphs = {"sets": sets, "shapes": {k: tuple(v.shape) for k, v in sets[0]["static"].items()}}
        return self._pipe_graphs

    def _snapshots_off(self):
        """hipGraph replays run no Python per tile: the out-of-domain totals are then read from the device at the boundary (the
        snapshot's event record inside a capture could not be waited for from the host)."""
        enc = getattr(self.model, "point_encoder", None)
        if enc is not None and hasattr(enc, "snapshot_domain"):
            enc.snapshot_domain = False

    def _pipe_graphs_match(self, data) -> bool:
        g = getattr(self, "_pipe_graphs", None)
        if g is None:
            return False
        return all(data.get(k) is not None and tuple(getattr(data[k], "shape", ())) == shp for k, shp in g["shapes"].items())

    def _graph_matches(self, data) -> bool:
        g = getattr(self, "_graph", None)
        if g is None:
            return False
        return all(data.get(k) is not None and tuple(getattr(data[k], "shape", ())) == shp for k, shp in g["shapes"].items())

    # ------------------------------------------------------------------------------------------ input pipeline
    def prepare(self, data, stream=None):
        """Build the point index of the NEXT tile (cell sort, sampling adjoint, cell counts) ahead of its ``train_step``, on
        ``stream`` -- a side stream, so these small latency-bound kernels run beside the current tile's step instead of alone
        at the head of the next one.  Returns ``data`` with the cloud replaced by the prebuilt ``TileIndex``; results are the
        same as for the raw cloud.  (HIP model with a point encoder only; anything else is returned unchanged.)"""
        enc = getattr(self.model, "point_encoder", None)
        cloud = data.get("inputs") if self.use_cloud else None
        if enc is None or not hasattr(enc, "prepare") or not torch.is_tensor(cloud):
            return data
        out = dict(data)
        out["inputs"] = enc.prepare(cloud.to(self.device), stream=stream)
        return out

    # ------------------------------------------------------------------------------------------ train
    def train_step(self, data) -> bool:
        """One tile -- or a LIST of tiles of the same accumulation window as one micro-batch (``_losses_micro_batch``: same
        accumulated gradient as feeding them one by one, to fp32 re-association) --: forward, loss, backward.  Returns True
        when this call ended with an optimizer step.

        Single tiles are coalesced (``coalesce_tiles``, default 4): the call accepts the tile and returns; the k-th call, the
        call that completes the accumulation window, ``flush_gradients()`` or ``optimizer_boundary()`` issue the held tiles as
        one ragged micro-batch.  The window's accumulated gradient, its optimizer step and ``last_avg_loss`` are those of the
        tile-by-tile loop (to fp32 re-association: ``test_coalesced_single_tile_api_*``).  The trainer keeps a reference to the
        tile's tensors until they are issued: do not overwrite them in place before the window's flush."""
        if (self.coalesce_tiles > 1 and self.bucket is not None and self._graph is None and self._pipe_graphs is None
                and isinstance(data, dict) and self._coalescible(data)):
            # accepted, issued later: with the (k - 1) tiles before or after it, at the optimizer boundary, or by a flush
            if not self._coalesced:
                self._coalesced_version = self._weights_version()
            self._coalesced.append(data)
            due = self.accumulated_steps + len(self._coalesced) >= self.local_every
            if len(self._coalesced) < self.coalesce_tiles and not due:
                return False
            return self._flush_coalesced()
        if self._coalesced:
            self._flush_coalesced()                 # (a list, a graph-shaped tile, ...: the order of the tiles is kept)
        return self._train_step_now(data)

    def _coalescible(self, data) -> bool:
        """Raw device tensors only: a cloud [1, N, 3] (+ image) + target, as the reference's loader hands them over.  A tile whose
        index was built ahead (``prepare``) is issued on its own -- its index is per tile."""
        cloud = data.get("inputs") if self.use_cloud else None
        if self.use_cloud and not (torch.is_tensor(cloud) and cloud.dim() == 3 and cloud.shape[0] == 1):
            return False
        return torch.is_tensor(data.get("dsm")) and next(self.model.parameters()).is_cuda

    def _flush_coalesced(self) -> bool:
        """Issue the tiles ``train_step`` holds back as one micro-batch.  True if that ended with an optimizer step."""
        tiles, self._coalesced = self._coalesced, []
        if not tiles:
            return False
        if self._coalesced_version != self._weights_version():
            # in the tile-by-tile loop these tiles' forwards would have run on the weights of their train_step call
            self._reset_accumulators()
            raise RuntimeError("the weights changed (an optimizer step outside the Trainer?) while tiles accepted by train_step were "
                               "still unflushed (coalescing): call Trainer.flush_gradients() before stepping an optimizer yourself "
                               "-- the accumulated gradients of this window have been dropped")
        return self._train_step_now(tiles if len(tiles) > 1 else tiles[0])

    def _train_step_now(self, data) -> bool:
        # trainer.py:59 calls model.train() every step.  Module.train() walks every submodule (0.4 ms per call here), so the
        # root flag is looked at per tile and the whole tree once per accumulation window: a caller that put a SUBMODULE into
        # eval mode while the root stayed in training mode is brought back at the next window, like the reference does
        if not self.model.training or (self.accumulated_steps == 0 and self._any_submodule_in_eval()):
            self.model.train()
        n_tiles = len(data) if isinstance(data, (list, tuple)) else 1
        if n_tiles > 1 and (self.accumulated_steps + n_tiles > self.local_every):
            raise ValueError(f"a micro-batch of {n_tiles} tiles would cross the optimizer step boundary "
                             f"({self.accumulated_steps} of {self.local_every} tiles accumulated)")
        if n_tiles == 1 and isinstance(data, (list, tuple)):
            data = data[0]
        first = data[0] if n_tiles > 1 else data
        if (self.pipeline_tiles and (n_tiles == 1 or self.pipeline_micro_batches) and self.bucket is not None and self._graph is None
                and isinstance(first, dict) and torch.is_tensor(first.get("dsm")) and next(self.model.parameters()).is_cuda):
            return self._train_step_pipelined(data, n_tiles)
        self.flush_pipeline()
        if n_tiles == 1 and self._graph_matches(data):
            g = self._graph
            for k, buf in g["static"].items():
                buf.copy_(data[k], non_blocking=True)
            g["graph"].replay()
            loss_l1, loss_ce = g["l1"], g["ce"]
            loss = loss_l1 + loss_ce
        else:
            with self._own_cache():
                loss_l1, loss_ce = self._losses(data, 0.0001)             # trainer.py:63-69
            loss = loss_l1 + loss_ce
            self._backward(loss)
        if self.bucket is None:
            self.flush_gradients()          # (the parameters behind the composed maps must have their gradient before the bucket is laid out)
            # first tile: the set of parameters that receive gradients is now known (it is static); from here on
            # their .grad are views into one flat buffer that the wgrad kernels accumulate into directly
            self.bucket = GradBucket(list(self.model.parameters()))

        self.accumulated_steps += n_tiles
        self.accumulated_loss += loss.detach()
        self.accumulated_loss_dict["loss_ce"] += loss_ce.detach()
        self.accumulated_loss_dict["loss_l1"] += loss_l1.detach()
        if self.accumulated_steps < self.local_every:
            return False

        self.optimizer_boundary()
        return True

    def _any_submodule_in_eval(self) -> bool:
        mods = getattr(self, "_modules_flat", None)
        if mods is None:
            mods = self._modules_flat = list(self.model.modules())
        return any(not m.training for m in mods)

    def _backward(self, loss):
        """``loss.backward()`` of one tile (or micro-batch) with the weight gradients on the side streams."""
        side = None
        if self.overlap_wgrad and self.bucket is not None and loss.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=loss.device)
            side = self._side
        conv_side = None
        if self.overlap_conv_wgrad and self.bucket is not None and loss.is_cuda:
            if self._conv_side is None:
                self._conv_side = torch.cuda.Stream(device=loss.device)
            conv_side = self._conv_side
        direct = self.bucket is not None and self.direct_accumulation
        # the weight gradients accumulate straight into the bucket and nothing reads it before the pass ends: their slab
        # reductions run as ONE batched launch at the end of the pass instead of one launch per layer (T2H_BATCH_REDUCE=0: A/B)
        # (not with the side streams: there each reduction runs at once beside other kernels, which measures faster than one
        # batched launch on an otherwise idle chip -- 8.71 against 8.95 ms per step)
        batch = (direct and loss.is_cuda and side is None and conv_side is None
                 and os.environ.get("T2H_BATCH_REDUCE", "1") != "0")
        with mlp.direct_grad_accumulation(direct, side, conv_side), _lib.reduce_capture(batch):
            loss.backward()
            for st in (side, conv_side):
                if st is not None:
                    # join: the overlap is with this tile's own backward; afterwards the gradients are visible in
                    # stream order on the current stream like any other result -- and the batched reduction, which runs
                    # when the capture block is left, reads slabs that were written on these streams
                    torch.cuda.current_stream().wait_stream(st)

    # ------------------------------------------------------------------------------------------ tiles in a two-stage pipeline
    def _train_step_pipelined(self, data, n_tiles=1) -> bool:
        """One tile of a window whose tiles overlap: the forward of tile i is issued -- on stream S[i % 2] -- BEFORE the backward
        of tile i - 1, which runs on S[(i - 1) % 2] (autograd runs a node's backward on the stream of its forward).  A tile
        lives on one stream from its first kernel to its last, so its activations are allocated, used and recycled in that
        stream's order; what the two streams share is ordered by events: the weights (read-only inside a window), the gradient
        bucket and the loss accumulators (tile i's backward waits for the end of tile i - 1's), the side stream of the weight
        gradients (forks from and joins the backward's stream).  Same kernels, same summation order per buffer as one tile
        after the other -- bit-identical gradients (``test_pipelined_tiles_give_identical_gradients``) -- with the chip's idle
        CUs of a B = 1 step filled by the neighbouring tile."""
        dev = next(self.model.parameters()).device
        main = torch.cuda.current_stream(dev)
        if self._tile_streams is None:
            self._tile_streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        parity = self._tile_parity
        st = self._tile_streams[parity]
        self._tile_parity ^= 1
        st.wait_stream(main)                                  # the tile's tensors, the weights of the last optimizer step
        gset = self._pipe_graphs["sets"][parity] if (n_tiles == 1 and self._pipe_graphs_match(data)) else None
        with torch.cuda.stream(st):
            if gset is not None:                              # (this stream's previous tile has finished with the static buffers:
                for k, buf in gset["static"].items():         #  its backward graph precedes these copies in stream order)
                    buf.copy_(data[k], non_blocking=True)
                gset["fwd"].replay()
                loss, loss_l1, loss_ce = gset["loss"], gset["l1"], gset["ce"]
            else:
                with self._own_cache():
                    loss_l1, loss_ce = self._losses(data, 0.0001)
                loss = loss_l1 + loss_ce
        self._issue_pending_backward()
        self._pending = (loss, loss_l1, loss_ce, st, self._weights_version(), gset, data)
        self.accumulated_steps += n_tiles
        if self.accumulated_steps < self.local_every:
            return False
        self.optimizer_boundary()
        return True

    def _issue_pending_backward(self):
        if self._pending is None:
            return
        loss, loss_l1, loss_ce, st, version, gset, data = self._pending
        self._pending = None
        if version != self._weights_version():
            self._reset_accumulators()
            raise RuntimeError("the weights changed (an optimizer step outside the Trainer?) while the backward of the previous tile "
                               "was still unflushed in the tile pipeline: call Trainer.flush_gradients() before stepping an "
                               "optimizer yourself -- the accumulated gradients of this window have been dropped")
        if self._bwd_done is not None:
            st.wait_event(self._bwd_done)                     # the previous tile's backward has the same accumulators
        with torch.cuda.stream(st):
            if gset is not None:
                gset["bwd"].replay()
                if self.compose_cache is not None:
                    self.compose_cache.pending = True         # (what the eager backward notes in Python)
            else:
                self._backward(loss)
            self.accumulated_loss += loss.detach()
            self.accumulated_loss_dict["loss_ce"] += loss_ce.detach()
            self.accumulated_loss_dict["loss_l1"] += loss_l1.detach()
            self._bwd_done = st.record_event(torch.cuda.Event(enable_timing=True))
        self._hold_inputs(self._bwd_done, data)

    def _hold_inputs(self, done, data):
        """The caller's tensors of a pipelined tile -- cloud or prebuilt TileIndex, target, image -- were allocated on the
        CALLER's stream (or a producer's), but are read by the tile's stream and the weight-gradient side streams until the
        end of its backward, one ``train_step`` call after the caller may have dropped them.  The caching allocator would hand
        such a block back to its own stream at once (that stream has no pending use of it) and the next tile's producer would
        overwrite it under the queued kernels.  So the trainer keeps ``data`` referenced until the event at the end of that
        tile's backward -- recorded after the side streams have joined -- HAS COMPLETED (host-side query: valid for whatever
        stream owns the allocation).  A few tiles' inputs (MBs) at most."""
        held = self._held
        while held and held[0][0].query():
            held.pop(0)
        held.append((done, data))

    def _weights_version(self) -> int:
        """Version counter of one trained parameter (every optimizer, FlatAdamW included, bumps all of them together)."""
        p = getattr(self, "_sentinel", None)
        if p is None:
            p = self._sentinel = next(q for q in self.model.parameters() if q.requires_grad)
        return p._version

    def step_event(self):
        """A timing-enabled event at the end of the GPU work of the most recent tile whose backward has been issued (the tile
        before the last ``train_step`` when the tile pipeline is on): consecutive ones are one tile-step apart."""
        if self._bwd_done is not None:
            return self._bwd_done
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def flush_pipeline(self):
        """Issue the backward that ``_train_step_pipelined`` still holds back and make the calling stream wait for the tile
        streams: afterwards gradients and loss accumulators are complete in the caller's stream order."""
        if self._coalesced:
            self._flush_coalesced()
        if self._pending is None and self._bwd_done is None:
            return
        self._issue_pending_backward()
        main = torch.cuda.current_stream()
        for st in self._tile_streams or ():
            main.wait_stream(st)
        self._bwd_done = None

    def flush_gradients(self):
        """Make every parameter's ``.grad`` complete for the tiles seen so far (issues the backward that the tile pipeline still
        holds back, back-propagates what the ComposeCache holds)."""
        self.flush_pipeline()
        if self.compose_cache is not None:
            self.compose_cache.flush()

    def optimizer_boundary(self):
        """End of an optimizer step (trainer.py:78-89): [all-reduce(SUM) of the flat gradient bucket over the ranks,]
        ``optimizer.step()``, loss averaging, gradients to zero.  ``on_reduced(flat_grad)``, if set, sees the complete
        accumulated (and reduced) gradient just before the optimizer consumes it (tests, ``bench.py --check-dp``)."""
        if self._coalesced and self._flush_coalesced():
            return                                                        # (the held tiles completed the window: its boundary has run)
        self.flush_pipeline()
        self.flush_gradients()
        if self.world > 1:
            self.bucket.all_reduce(self.group)                            # one SUM all-reduce per step
        if self.on_reduced is not None:
            self.on_reduced(self.bucket.flat)
        if self.check_domain and hasattr(self.model, "out_of_domain_total"):
            # before the optimizer consumes anything: a caller that catches the error continues from the weights, moments
            # and learning rate of the last good step, with empty accumulators
            bad = self.model.out_of_domain_total()
            if self.world > 1:
                # every rank must take the same branch (the all-reduce above has already happened on all of them).  On a stream
                # of its own: the .item() then does not wait for the tile streams' queued work (the collective itself still runs
                # behind the gradient all-reduce on the process group's internal stream -- one bucket's worth of waiting)
                dev = self.bucket.flat.device
                if dev.type == "cuda":
                    if self._flag_stream is None:
                        self._flag_stream = torch.cuda.Stream(device=dev)
                    with torch.cuda.stream(self._flag_stream):
                        flag = torch.tensor([float(bad)], device=dev)
                        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=self.group)
                        bad = int(flag.item())
                else:
                    flag = torch.tensor([float(bad)], device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=self.group)
                    bad = int(flag.item())
            if bad:
                self._reset_accumulators()
                raise ValueError(f"{bad} input point(s) of the last {self.optimize_every} tile(s) had x or y outside [0, 1) "
                                 "(or NaN): normalise / crop the tiles as dataset.py:270-278 does")
        self.optimizer.step()
        if self.compose_cache is not None:
            # new weights -> new composed maps, in place.  (A forward would notice by itself -- every optimizer, FlatAdamW
            # included, bumps the parameters' version counters, which ComposeCache.get compares -- but a replayed hipGraph
            # runs no Python per tile.)
            self.compose_cache.refresh()
        # the split copies of the convolution weights (and of the matrices of the grid-side products) follow the new weights here,
        # all of them in a few launches (grid.SplitWeightCache.refresh): a replayed hipGraph runs no Python per tile, and eagerly
        # the first tile after the step would otherwise prepare ~110 buffers one by one (0.85 ms of host time)
        from . import grid
        grid.split_weights.refresh(stale_only=self._graph is None and self._pipe_graphs is None)
        if self.scheduler is not None:
            self.scheduler.step()                                         # train.py:188-190: once per iteration
        with torch.no_grad():
            denom = self.optimize_every
            acc = self.accumulated_loss
            acc_d = dict(self.accumulated_loss_dict)
            if self.world > 1:
                packed = torch.stack([torch.as_tensor(acc, device=self.device, dtype=torch.float32),
                                      torch.as_tensor(acc_d["loss_ce"], device=self.device, dtype=torch.float32),
                                      torch.as_tensor(acc_d["loss_l1"], device=self.device, dtype=torch.float32)])
                dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
                acc, acc_d = packed[0], {"loss_ce": packed[1], "loss_l1": packed[2]}
            self.last_avg_loss = acc / denom
            self.last_avg_loss_dict = {k: v / denom for k, v in acc_d.items()}
        self._reset_accumulators()

    def _reset_accumulators(self):
        self._coalesced = []                               # (error path: tiles accepted but never issued are dropped with the window)
        if self._pending is not None:                      # (error path: a forward whose backward will never be issued)
            st, data = self._pending[3], self._pending[6]
            self._hold_inputs(st.record_event(torch.cuda.Event()), data)
        self._pending = None
        if self._tile_streams is not None:                 # (nothing of the tile pipeline may still be accumulating)
            cur = torch.cuda.current_stream()
            for st in self._tile_streams:
                cur.wait_stream(st)
        self._bwd_done = None
        if self.compose_cache is not None and self.compose_cache.pending:    # (error path: drop what was accumulated)
            for e in self.compose_cache.levels:
                e["ga"].zero_()
                e["gconst"].zero_()
            self.compose_cache.pending = False
        self.accumulated_loss = 0.0
        self.accumulated_steps = 0
        self.accumulated_loss_dict = {k: 0.0 for k in self.accumulated_loss_dict}
        self.bucket.zero_()     # == optimizer.zero_grad() for every parameter that has a gradient

    # ------------------------------------------------------------------------------------------ eval
    def eval_step(self, data):
        if self._coalesced:
            self._flush_coalesced()
        self.model.eval()
        with torch.no_grad():
            loss_l1, loss_ce = self._losses(data, 0.00001)                # trainer.py:136
            loss = loss_l1 + loss_ce
        return {"loss": loss.item(), "loss_l1": loss_l1.item(), "loss_ce": loss_ce.item()}

    def evaluate(self, val_loader):
        metrics = defaultdict(list)
        for data in val_loader:
            for k, v in self.eval_step(data).items():
                metrics[k].append(v)
        return {k: torch.tensor(v).mean().item() for k, v in metrics.items()}


def broadcast_parameters(model: nn.Module, group=None, src: int = 0):
    """Identical replicas before the first step (rank `src`'s init wins)."""
    flat = torch.cat([p.data.reshape(-1) for p in model.parameters()])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for p in model.parameters():
        n = p.numel()
        p.data.copy_(flat[off:off + n].view_as(p))
        off += n
