"""Tile producer (reference dataset.py:229-278): oracle vs the fixture built from the reference's own
utility functions (CPU) and the device kernels vs both (GPU).  Selected indices bit exact; coordinates within one
float32 ulp (the reference multiplies by an inverted 4x4 matrix, the restatements use the closed form)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import producer_ref

Z_SPAN = 156.5 - (-33.7)


def _ulp_close(got, want):
    """one float32 ulp, or 1e-7 absolute near zero (the reference's matrix form leaves ~1e-17 instead of an exact 0
    at the z-min point), and bit-equal for > 99 % of the values."""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1.2e-7, atol=1e-7)
    assert (got == want).mean() > 0.99


def test_oracle_matches_reference_fixture():
    g = load_golden("tile_producer")
    assert int(g["n_3"]) == 0 and int(g["n_0"]) > 100
    for i in range(4):
        idx, pts, z_shift = producer_ref.produce_tile(g["chunk"], g["anchors"][i], z_span=Z_SPAN)
        if int(g[f"n_{i}"]) == 0:
            assert idx.size == 0
            continue
        assert np.array_equal(idx, g[f"index_{i}"])
        if i == 0:
            assert 0 not in idx and 1 not in idx and 2 in idx                  # points ON the window edge are excluded
        _ulp_close(pts, g[f"inputs_{i}"])
        assert z_shift == float(g[f"zshift_{i}"][0])


@pytest.mark.gpu
def test_device_producer_matches_fixture_and_feeds_the_model():
    from tomosar2height_amd.producer import TileProducer
    g = load_golden("tile_producer")
    dev = torch.device("cuda:0")
    prod = TileProducer(torch.from_numpy(g["chunk"]).to(dev))
    for i in range(4):
        tile = prod.crop(g["anchors"][i], with_index=True)
        if int(g[f"n_{i}"]) == 0:
            assert not bool(tile["is_valid"][0]) and "inputs" not in tile
            continue
        assert bool(tile["is_valid"][0])
        assert np.array_equal(tile["index"].cpu().numpy(), g[f"index_{i}"])
        _ulp_close(tile["inputs"][0].cpu().numpy(), g[f"inputs_{i}"])
        assert tile["z_shift"].item() == float(g[f"zshift_{i}"][0])
        assert tile["inputs"].shape[0] == 1 and tile["inputs"].dtype == torch.float32
    # a large random chunk against the oracle, then straight into the tile index
    rng = np.random.RandomState(0)
    chunk = np.stack([rng.uniform(1000, 3000, 300000), rng.uniform(5000, 7000, 300000), rng.uniform(0, 80, 300000)], 1)
    prod = TileProducer(torch.from_numpy(chunk).to(dev))
    tile = prod.crop((1500.0, 5600.0), with_index=True)
    idx, pts, _ = producer_ref.produce_tile(chunk, (1500.0, 5600.0), z_span=Z_SPAN)
    assert np.array_equal(tile["index"].cpu().numpy(), idx)
    _ulp_close(tile["inputs"][0].cpu().numpy(), pts)
    from tomosar2height_amd.tile import TileIndex
    t = TileIndex(tile["inputs"], 256)
    assert t.out_of_domain() == 0 and t.n_points == idx.size


AUG = [(r, f) for r in range(4) for f in (-1, 0, 1)]


def test_oracle_augmentation_matches_reference_fixture():
    """flip_mat @ rot_mat on the points and rot90 / flip on the rasters (dataset.py:253-328) against the fixture composed
    from the reference's own utilities."""
    g = load_golden("tile_producer_aug")
    row, col, ph, pw = (int(v) for v in g["row_col_shape"])
    for rot, flip in AUG:
        tag = f"r{rot}_f{flip + 1}"
        idx, pts, _ = producer_ref.produce_tile(g["chunk"], g["anchor"], z_span=Z_SPAN, rot_times=rot, flip_dim=flip)
        assert np.array_equal(idx, g[f"index_{tag}"]), tag
        _ulp_close(pts, g[f"inputs_{tag}"])
        assert np.array_equal(producer_ref.raster_patch(g["dsm_data"][None], row, col, (ph, pw), rot, flip), g[f"dsm_{tag}"]), tag
        assert np.array_equal(producer_ref.raster_patch(g["image"], row, col, (ph, pw), rot, flip), g[f"image_{tag}"]), tag


@pytest.mark.gpu
def test_device_augmented_tiles_match_fixture():
    from tomosar2height_amd.producer import RasterPatcher, TileProducer, TileSource
    g = load_golden("tile_producer_aug")
    dev = torch.device("cuda:0")
    row, col, ph, pw = (int(v) for v in g["row_col_shape"])
    prod = TileProducer(torch.from_numpy(g["chunk"]).to(dev))
    # a raster whose pixel (row, col) holds the window's bottom-left corner: 32 m pixels, 16 x 16 pixel patches
    px = 32.0
    left, top = float(g["anchor"][0]) - col * px, float(g["anchor"][1]) + (row + 1) * px
    dsm = RasterPatcher(torch.from_numpy(g["dsm_data"]).to(dev), left, top, (px, px))
    img = RasterPatcher(torch.from_numpy(g["image"]).to(dev), left, top, (px, px))
    assert dsm.patch_shape == (ph, pw) and dsm.query_col_row(float(g["anchor"][0]) + px / 2, float(g["anchor"][1]) + px / 2) == (col, row)
    for rot, flip in AUG:
        tag = f"r{rot}_f{flip + 1}"
        tile = prod.crop(g["anchor"], with_index=True, rot_times=rot, flip_dim=flip)
        assert np.array_equal(tile["index"].cpu().numpy(), g[f"index_{tag}"]), tag
        _ulp_close(tile["inputs"][0].cpu().numpy(), g[f"inputs_{tag}"])
        assert np.array_equal(dsm.patch(g["anchor"], rot, flip).cpu().numpy(), g[f"dsm_{tag}"]), tag
        assert np.array_equal(img.patch(g["anchor"], rot, flip).cpu().numpy(), g[f"image_{tag}"]), tag

    class _Fixed:                         # the augmentation draw of dataset.py:253-263, pinned
        def __init__(self, seq): self.seq = list(seq)
        def choice(self, n): return self.seq.pop(0)
    src = TileSource(prod, dsm, img, flip_augm=True, rotate_augm=True, rng=_Fixed([3, 1]))     # rot 3, flip key index 1 -> 0
    t = src.get(g["anchor"])
    assert (t["rotate"], t["flip"]) == (3, 0) and t["dsm"].shape == (1, ph, pw) and t["image"].shape == (1, 3, ph, pw)
    assert np.array_equal(t["dsm"].cpu().numpy(), g["dsm_r3_f1"]) and np.array_equal(t["image"][0].cpu().numpy(), g["image_r3_f1"])
    # the same tile produced on a side stream (the prefetching form: the host's point-count read waits for the crop only,
    # not for a training step running on the main stream) -- while the main stream is kept busy -- is bit-identical
    busy = torch.randn(4096, 4096, device=dev)
    for _ in range(4):
        busy = busy @ busy * 1e-4
    src2 = TileSource(prod, dsm, img, flip_augm=True, rotate_augm=True, rng=_Fixed([3, 1]), stream=torch.cuda.Stream())
    t2 = src2.get(g["anchor"])
    for k in ("inputs", "dsm", "image"):
        assert torch.equal(t2[k], t[k]), k
    assert (t2["rotate"], t2["flip"]) == (3, 0)
    with pytest.raises(RuntimeError, match="leave the"):
        dsm.patch((left - 10 * px, float(g["anchor"][1])))
