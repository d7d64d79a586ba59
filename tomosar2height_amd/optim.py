"""Optimizer and LR schedule of the training loop (reference: train.py:97-104 ``optim.AdamW(model.parameters(), lr)`` +
``CyclicLR``; stepped at trainer.py:78-79 and train.py:188-190) on the MI355X path.

``FlatAdamW`` IS a ``torch.optim.AdamW`` (same constructor defaults, ``param_groups``, ``state`` keys ``step`` /
``exp_avg`` / ``exp_avg_sq``, ``state_dict()`` format -- so reference checkpoints' ``'optimizer'`` entry loads and any
``torch.optim.lr_scheduler`` drives it), but ``step()`` is ONE HIP launch (``t2h_adamw_flat_step``) over all parameters:
the gradients already sit in the Trainer's flat bucket, the two moments live in flat buffers here, and a device table
maps 4096-element chunks to tensors.  torch's multi-tensor step is ~10 launches over ~150 tensors.

``cyclic_lr(optimizer, cfg)`` builds the reference's scheduler from ``cfg.training.scheduler`` (conf/model/
tomosar2height.yaml:46-55: CyclicLR 1e-4..5e-4, triangular2, 500/500, cycle_momentum false)."""
import torch

from . import _lib


def _same_dense_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.shape == b.shape and a.stride() == b.stride()


class FlatAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False,
                         fused=False)
        self._plans = {}          # group index -> plan

    # ---------------------------------------------------------------------------------------- planning
    def _plan(self, gi, group):
        live = [p for p in group["params"] if p.grad is not None]
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in live)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        if not live:
            plan = {"key": key, "live": [], "n_chunks": 0}
            self._plans[gi] = plan
            return plan
        dev = live[0].device
        for p in live:
            if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                raise RuntimeError("FlatAdamW: fp32 parameters on the MI355X only (tomosar2height_amd has no CPU path)")
            if p.grad.is_sparse or not _same_dense_layout(p, p.grad):
                raise RuntimeError("FlatAdamW: a gradient must share its parameter's dense memory layout")
        total = sum(-(-p.numel() // 4) * 4 for p in live)
        old = plan
        m_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        v_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        chunk = _lib.load().t2h_adamw_chunk_elems()
        rows, chunks, off = [], [], 0
        for ti, p in enumerate(live):
            n = p.numel()
            m = torch.as_strided(m_flat, p.size(), p.stride(), storage_offset=off)
            v = torch.as_strided(v_flat, p.size(), p.stride(), storage_offset=off)
            st = self.state[p]
            if "exp_avg" in st:                           # state loaded from a checkpoint / carried over a re-plan
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
            else:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"], st["exp_avg_sq"] = m, v
            rows.append([p.data_ptr(), p.grad.data_ptr(), m_flat.data_ptr() + 4 * off, v_flat.data_ptr() + 4 * off, n])
            chunks += [[ti, s] for s in range(0, n, chunk)]
            off += -(-n // 4) * 4
        del old
        plan = {"key": key, "live": live, "m": m_flat, "v": v_flat,
                "table": torch.tensor(rows, dtype=torch.int64).to(dev),
                "chunks": torch.tensor(chunks, dtype=torch.int32).to(dev), "n_chunks": len(chunks),
                "bytes": 28 * sum(p.numel() for p in live)}
        self._plans[gi] = plan
        return plan

    # ---------------------------------------------------------------------------------------- step
    @torch.no_grad()
    def step(self, closure=None, zero_grad: bool = False):
        """One AdamW step; ``zero_grad=True`` also clears the gradients in the same pass."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                raise RuntimeError("FlatAdamW: amsgrad / maximize are not built (the reference uses neither)")
            plan = self._plan(gi, group)
            if not plan["live"]:
                continue
            steps = {float(self.state[p]["step"]) for p in plan["live"]}
            if len(steps) != 1:
                raise RuntimeError("FlatAdamW: parameters of one group must share their step count")
            step = int(steps.pop()) + 1
            beta1, beta2 = group["betas"]
            _lib.call("t2h_adamw_flat_step", _lib.ptr(plan["table"]), _lib.ptr(plan["chunks"]), plan["n_chunks"],
                      float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]),
                      step, 1 if zero_grad else 0, _lib.stream(), nbytes=plan["bytes"])
            for p in plan["live"]:
                self.state[p]["step"] += 1
            # the kernel wrote the weights through raw pointers: tell autograd, as an in-place torch op would (anything
            # that caches functions of the weights keys on ``_version`` -- deferred.ComposeCache -- and saved-tensor checks)
            torch.autograd.graph.increment_version(plan["live"])
        return loss

    def load_state_dict(self, state_dict):
        """torch's loader replaces the state tensors; the next ``step()`` re-plans and copies them into the flat buffers."""
        super().load_state_dict(state_dict)
        self._plans = {}
        for st in self.state.values():
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].detach().to("cpu", torch.float32).reshape(())


def cyclic_lr(optimizer, cfg):
    """The scheduler ``train.py:98-104`` builds from ``cfg.training.scheduler`` (only the type the reference's YAML selects
    and its siblings in torch: the schedule itself is host arithmetic on ``param_groups[...]['lr']``)."""
    import torch.optim.lr_scheduler as sched
    spec = cfg["training"]["scheduler"]
    kinds = {"CyclicLR": sched.CyclicLR, "ReduceLROnPlateau": sched.ReduceLROnPlateau,
             "CosineAnnealingLR": sched.CosineAnnealingLR, "CosineAnnealingWarmRestarts": sched.CosineAnnealingWarmRestarts}
    return kinds[spec["type"]](optimizer, **dict(spec["kwargs"]))
