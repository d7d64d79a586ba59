"""TEST INFRASTRUCTURE (oracle) -- restatement of ``torch_scatter.scatter_max`` /
``scatter_mean`` for the call shapes the reference uses.

pytorch-scatter (pinned by nothing in the reference: ``environment.yml:13-15``
lists ``pytorch-scatter`` without a version; with pytorch 2.3.0 it resolves to
2.1.2) is not vendored under /root/reference and cannot be installed here, so
its published algorithm is restated:

* ``scatter_mean(src, index, out=zeros)`` (reference call sites
  ``tomosar2height/encoder/pointnet.py:109``, ``encoder/alto.py:85,194``) is a
  python composite in pytorch-scatter 2.1.x::

      out.scatter_add_(dim, index, src)
      count = zeros.scatter_add_(dim, index, ones); count[count < 1] = 1
      out.true_divide_(count)

  i.e. an fp32 sum in source order followed by one division.

* ``scatter_max(src, index, dim, dim_size=)`` (``pointnet.py:95``) returns
  ``(out, arg)``; CPU kernel: running value initialised to the lowest float,
  updated on strict ``>`` (first occurrence wins ties), cells never written
  get value 0 and ``arg = src.size(dim)``.  Backward routes the gradient to
  ``arg`` only.  PARITY UNPINNED for the tie-break (no reference test holds a
  vector for it).

Shapes handled: ``src [B, C, N]`` (any strides), ``index [B, 1, N]`` broadcast
over C, reduction over the last dim -- exactly what the reference passes.
"""
import torch


def _expand_index(index: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    if index.dim() != src.dim():
        raise ValueError("index must have the same number of dims as src")
    return index.expand_as(src)


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out: torch.Tensor = None,
                 dim_size: int = None) -> torch.Tensor:
    if dim not in (-1, src.dim() - 1):
        raise NotImplementedError("oracle restates the last-dim call only")
    idx = _expand_index(index, src)
    if out is None:
        size = list(src.shape)
        size[-1] = int(dim_size) if dim_size is not None else int(index.max()) + 1
        out = src.new_zeros(size)
    # out-of-place on purpose (autograd friendly); `out` is all-zero at every
    # reference call site so this equals the in-place composite.
    summed = out.scatter_add(-1, idx, src)
    count = torch.zeros_like(out).scatter_add(-1, idx, torch.ones_like(src, dtype=out.dtype))
    count = count.clamp(min=1)
    return summed / count


class _ScatterMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx, dim_size):
        n = src.shape[-1]
        flat_src = src.reshape(-1, n)
        flat_idx = idx.reshape(-1, n)
        rows = flat_src.shape[0]
        # value: plain amax per cell
        lowest = torch.finfo(src.dtype).min
        val = flat_src.new_full((rows, dim_size), lowest)
        val = val.scatter_reduce(1, flat_idx, flat_src, reduce="amax", include_self=True)
        # arg: first position whose value equals the cell max
        pos = torch.arange(n, device=src.device).expand(rows, n)
        is_max = flat_src == val.gather(1, flat_idx)
        cand = torch.where(is_max, pos, torch.full_like(pos, n))
        arg = torch.full((rows, dim_size), n, dtype=torch.long, device=src.device)
        arg = arg.scatter_reduce(1, flat_idx, cand, reduce="amin", include_self=True)
        touched = arg < n
        val = torch.where(touched, val, torch.zeros_like(val))
        out_shape = list(src.shape[:-1]) + [dim_size]
        ctx.save_for_backward(arg)
        ctx.n = n
        ctx.src_shape = src.shape
        ctx.mark_non_differentiable(arg)
        return val.reshape(out_shape), arg.reshape(out_shape)

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        n = ctx.n
        g = grad_out.reshape(arg.shape)
        grad_src = g.new_zeros(arg.shape[0], n + 1)
        grad_src.scatter_(1, arg, g)  # arg is injective among touched cells
        return grad_src[:, :n].reshape(ctx.src_shape), None, None


def scatter_max(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out=None, dim_size: int = None):
    if dim not in (-1, src.dim() - 1):
        raise NotImplementedError("oracle restates the last-dim call only")
    if out is not None:
        raise NotImplementedError("reference never passes out= to scatter_max")
    idx = _expand_index(index, src)
    if dim_size is None:
        dim_size = int(index.max()) + 1
    return _ScatterMax.apply(src, idx, int(dim_size))
