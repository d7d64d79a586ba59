"""ResnetBlockFC with the reference's parameters (block/resnet.py:4-54): ``fc_0``, ``fc_1`` and a bias-free
``shortcut`` when the width changes.  The arithmetic is ``mlp.resblock`` (fp32 MFMA kernels with the ReLUs, biases
and the residual add fused); inside the PointNet trunk the blocks are driven by ``mlp.point_trunk`` instead, which
also removes the reference's ``torch.cat([net, pooled], dim=2)`` (pointnet.py:78)."""
import torch
import torch.nn as nn

from .. import mlp


class ResnetBlockFC(nn.Module):
    def __init__(self, size_in, size_out=None, size_h=None):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        self.actvn = nn.ReLU()
        self.shortcut = nn.Linear(size_in, size_out, bias=False) if size_in != size_out else None
        nn.init.zeros_(self.fc_1.weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ws = None if self.shortcut is None else self.shortcut.weight
        return mlp.resblock(x, self.fc_0.weight, self.fc_0.bias, self.fc_1.weight, self.fc_1.bias, ws)
