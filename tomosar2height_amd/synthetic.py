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


def berlin_chunk(seed: int, tiles_per_side: int = 3, n_points: int = DEFAULT_POINTS, clustered: bool = True,
                 with_image: bool = False, left: float = 390000.0, bottom: float = 5810000.0, z0: float = 30.0):
    """A synthetic CHUNK in the form ``TomoSARDataset`` holds one in RAM (reference dataset.py:95-140): the point cloud
    in float64 world coordinates (metres, UTM-sized offsets -- the reason the reference normalises in float64,
    dataset.py:232), a north-up DSM raster at 1 m / pixel and optionally a normalised image raster.  It is
    ``tiles_per_side``^2 Berlin-shaped tiles side by side, so any 512 m window holds about ``n_points`` points.
    Returns the pieces ``producer.TileProducer`` / ``RasterPatcher`` take."""
    k = int(tiles_per_side)
    rng = np.random.RandomState(seed)
    pts, side = [], 512 * k
    dsm = np.zeros((side, side), np.float32)                       # built south row first, flipped to north-up below
    img = np.zeros((3, side, side), np.float64) if with_image else None
    for j in range(k):
        for i in range(k):
            t = berlin_tile(seed * 1000 + j * k + i, n_points=n_points, clustered=clustered, with_image=with_image)
            p = t["inputs"][0].numpy().astype(np.float64)
            pts.append(np.stack([left + 512.0 * (i + p[:, 0]), bottom + 512.0 * (j + p[:, 1]),
                                 z0 + Z_SPAN_BERLIN * p[:, 2]], 1))
            dsm[512 * j:512 * (j + 1), 512 * i:512 * (i + 1)] = t["dsm"][0].numpy()
            if with_image:
                img[:, 512 * j:512 * (j + 1), 512 * i:512 * (i + 1)] = t["image"][0].numpy()
    pts = np.concatenate(pts, 0)
    pts = pts[rng.permutation(pts.shape[0])]
    out = {"points": torch.from_numpy(pts), "dsm": torch.from_numpy(np.ascontiguousarray(dsm[::-1])),
           "left": left, "bottom": bottom, "top": bottom + float(side), "z_bound": (-33.7, 156.5)}
    if with_image:
        out["image"] = torch.from_numpy(np.ascontiguousarray(img[:, ::-1]))
    return out
