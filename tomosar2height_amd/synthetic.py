"""Berlin-shaped synthetic tiles: the tile contract of
``TomoSARDataset.__getitem__`` (reference dataset.py:280-289,310,328) without the unpublished data.

``inputs [N,3]`` fp32: 70 % of the points inside 160 axis-aligned "buildings" (centres U(0,1)^2, sides
U(10,60) m of a 512 m tile), 30 % U(0,1)^2, clamped to [2^-20, 1-2^-20]; z = U(0,h_b)/190.2 on buildings
(h_b ~ U(5,60) m), |N(0,1.5 m)|/190.2 on the ground; order shuffled (real tiles are not cell-sorted).
``dsm [512,512]``: U(0,30) m inside the building rectangles, 0 elsewhere.  numpy RandomState(seed): the same tile
on every host.
"""
import numpy as np
import torch

Z_SPAN_BERLIN = 190.2
DEFAULT_POINTS = 131072


def berlin_tile(seed: int, n_points: int = DEFAULT_POINTS, n_buildings: int = 160, clustered: bool = True,
                with_image: bool = False):
    rng = np.random.RandomState(seed)
    n_b = int(round(0.7 * n_points)) if clustered else 0
    centres = rng.uniform(0, 1, (n_buildings, 2))
    sides = rng.uniform(10, 60, (n_buildings, 2)) / 512.0
    heights = rng.uniform(5, 60, n_buildings)
    which = rng.randint(0, n_buildings, n_b)
    xy_b = centres[which] + (rng.uniform(0, 1, (n_b, 2)) - 0.5) * sides[which]
    z_b = rng.uniform(0, 1, n_b) * heights[which] / Z_SPAN_BERLIN
    n_g = n_points - n_b
    xy_g = rng.uniform(0, 1, (n_g, 2))
    z_g = np.abs(rng.normal(0, 1.5, n_g)) / Z_SPAN_BERLIN
    pts = np.concatenate([np.concatenate([xy_b, z_b[:, None]], 1), np.concatenate([xy_g, z_g[:, None]], 1)], 0)
    eps = 2.0 ** -20
    pts[:, :2] = np.clip(pts[:, :2], eps, 1 - eps)
    pts = pts[rng.permutation(n_points)].astype(np.float32)

    dsm = np.zeros((512, 512), np.float32)
    roof = rng.uniform(0, 30, n_buildings).astype(np.float32)
    for c, s, h in zip(centres, sides, roof):
        x0, x1 = np.clip(((c[0] - s[0] / 2) * 512, (c[0] + s[0] / 2) * 512), 0, 512).astype(int)
        y0, y1 = np.clip(((c[1] - s[1] / 2) * 512, (c[1] + s[1] / 2) * 512), 0, 512).astype(int)
        dsm[y0:y1, x0:x1] = h
    tile = {"inputs": torch.from_numpy(pts)[None], "dsm": torch.from_numpy(dsm)[None],
            "is_valid": torch.tensor([True])}
    if with_image:
        tile["image"] = torch.from_numpy(rng.normal(0, 1, (1, 3, 512, 512)).astype(np.float32))
    return tile
