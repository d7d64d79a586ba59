"""Per-point MLP arithmetic (the dense GEMM side of the hot path: pointnet.py:36-40, resnet.py:26-31,
alto.py:63-69,164-170).  Rows are points in cell-sorted order.

Round-1 state: these are plain library GEMMs through PyTorch-ROCm (rocBLAS / hipBLASLt, fp32) on the
device; the hand-written MFMA kernels of SURVEY.md section 7 step 6 replace them behind this same
interface.  Nothing here runs on the CPU in the product path: every input comes out of a t2h HIP op.
"""
from typing import Optional

import torch
import torch.nn.functional as F


def _need_gpu(x: torch.Tensor):
    if not x.is_cuda:
        raise RuntimeError("tomosar2height_amd.mlp: expected device tensors; there is no CPU path "
                           "(the CPU restatement lives in oracle/ for tests only)")


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], relu_in: bool = False) -> torch.Tensor:
    _need_gpu(x)
    return F.linear(F.relu(x) if relu_in else x, weight, bias)


def resblock(xa: torch.Tensor, xb: Optional[torch.Tensor], w0, b0, w1, b1, ws) -> torch.Tensor:
    """block/resnet.py:36-54 on ``x = [xa | xb]``: ``shortcut(x) + fc_1(relu(fc_0(relu(x))))``."""
    _need_gpu(xa)
    if xb is None:
        h = F.linear(F.relu(xa), w0, b0)
        xs = xa if ws is None else F.linear(xa, ws)
    else:
        ca = xa.shape[-1]
        h = F.linear(F.relu(xa), w0[:, :ca], b0) + F.linear(F.relu(xb), w0[:, ca:])
        xs = F.linear(xa, ws[:, :ca]) + F.linear(xb, ws[:, ca:])
    return xs + F.linear(F.relu(h), w1, b1)


def comm_mlp(sampled: torch.Tensor, w_a, b_a, w_b, b_b, c_last: Optional[torch.Tensor], w_c, b_c) -> torch.Tensor:
    """ALTO point update (alto.py:121-128, 245-253): ``fc_comm(sampled) + fc_c(c_last)`` with
    ``fc_comm = Linear(C,2C) -> ReLU -> Linear(2C,C)``."""
    _need_gpu(sampled)
    c = F.linear(F.relu(F.linear(sampled, w_a, b_a)), w_b, b_b)
    if c_last is not None:
        c = c + F.linear(c_last, w_c, b_c)
    return c
