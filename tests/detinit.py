"""Name-keyed deterministic parameter fill, shared by the golden generator and the tests.

Independent of module construction order and of the global RNG, so the reference
model (built in the build container) and the HIP model (built on the GPU box) get
bit-identical weights without shipping a 44 MB state_dict.
"""
import math
import zlib

import torch


def det_init_(model: torch.nn.Module, seed: int = 0, bias_scale: float = 0.05) -> torch.nn.Module:
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
            if p.dim() >= 2:
                rf = 1
                for s_ in p.shape[2:]:
                    rf *= s_
                # symmetric in fan_in/fan_out, so Linear/Conv2d/ConvTranspose2d need no special case
                bound = math.sqrt(6.0 / ((p.shape[0] + p.shape[1]) * rf))
            else:
                bound = bias_scale
            vals = (torch.rand(p.shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
            p.copy_(vals)
    return model


def synth_cloud(n: int, seed: int = 0, batch: int = 1, clustered: bool = True) -> torch.Tensor:
    """Small Berlin-shaped cloud [B,N,3]: xy strictly inside (0,1), z >= 0 (dataset.py:278-289)."""
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(batch, n, 2, generator=g)
    if clustered:
        n_c = max(1, n * 7 // 10)
        centers = torch.rand(batch, 12, 2, generator=g)
        pick = torch.randint(0, 12, (batch, n_c), generator=g)
        side = 0.02 + 0.1 * torch.rand(batch, 12, 1, generator=g)
        off = (torch.rand(batch, n_c, 2, generator=g) - 0.5)
        xy[:, :n_c] = torch.gather(centers, 1, pick[..., None].expand(-1, -1, 2)) + \
            off * torch.gather(side, 1, pick[..., None])
        perm = torch.randperm(n, generator=g)
        xy = xy[:, perm]
    eps = 2.0 ** -20
    xy = xy.clamp(eps, 1 - eps)
    z = torch.rand(batch, n, 1, generator=g) * 0.3
    return torch.cat([xy, z], dim=2).float().contiguous()
