"""DSM mosaic (reference generator.py:85-157): blend weights pinned by the fixture produced from the
reference's own static method; device accumulate / finalize against the numpy oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import mosaic_ref


def test_blend_weight_oracle_and_product_match_reference_fixture():
    from tomosar2height_amd.generator import DSMGenerator
    g = load_golden("mosaic_blend_weight")
    for k in g.files:
        if k.startswith("w512"):
            continue
        _, shape, a, b = k.split("_")
        r, c = (int(v) for v in shape.split("x"))
        pct = (float(a), float(b))
        np.testing.assert_allclose(mosaic_ref.linear_blend_patch_weight((r, c), pct), g[k], rtol=0, atol=1e-15)
        np.testing.assert_array_equal(DSMGenerator._linear_blend_patch_weight((r, c), list(pct)).numpy(), g[k])
    w = DSMGenerator._linear_blend_patch_weight((512, 512), [0.5, 0.5]).numpy()          # the shipped configuration
    np.testing.assert_array_equal(w[0, :], g["w512_row0"])
    np.testing.assert_array_equal(w[:, 0], g["w512_col0"])
    np.testing.assert_array_equal(w[255:257, 255:257], g["w512_centre"])
    assert w.min() == pytest.approx(1e-6) and w.max() == 1.0


def test_cal_shape_and_col_row():
    from tomosar2height_amd.generator import DSMGenerator
    assert DSMGenerator.cal_dsm_shape((10.0, 20.0), (1034.5, 788.0), (1.0, 1.0)) == (768, 1024)
    gen = DSMGenerator.__new__(DSMGenerator)
    gen.l_bound, gen.t_bound, gen.pixel_size = 10.0, 788.0, [1.0, 1.0]
    assert gen.query_col_row(10.5, 787.5) == (0, 0) == mosaic_ref.col_row(10.5, 787.5, 10.0, 788.0, (1.0, 1.0))
    assert gen.query_col_row(522.49, 276.5) == (512, 511)


@pytest.mark.gpu
def test_mosaic_accumulate_finalize_vs_oracle():
    from tomosar2height_amd.generator import DSMGenerator
    dev = torch.device("cuda:0")
    gen = DSMGenerator(model=None, device=dev, tiles=[], bounds=(0.0, 0.0, 160.0, 130.0), patch_size=(64.0, 64.0))
    assert gen.dsm_shape == (130, 160)
    g = torch.Generator().manual_seed(0)
    tiles = [(torch.randn(1, 64, 64, 1, generator=g) * 20, t, l) for t, l in ((0, 0), (0, 32), (32, 0), (32, 32), (60, 96), (66, 40))]
    dsm = torch.zeros(gen.dsm_shape, dtype=torch.float64, device=dev)
    weight = torch.zeros_like(dsm)
    for h, t, l in tiles:
        gen.accumulate(dsm, weight, h.to(dev), t, l)
    from tomosar2height_amd import _lib
    _lib.call("t2h_mosaic_finalize", _lib.ptr(dsm), _lib.ptr(weight), dsm.numel(), _lib.stream())
    want = mosaic_ref.mosaic([(h[0, :, :, 0].numpy(), t, l) for h, t, l in tiles], gen.dsm_shape,
                             mosaic_ref.linear_blend_patch_weight((64, 64), (0.5, 0.5)))
    got = dsm.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want).any()       # uncovered pixels stay NaN
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=1e-12, atol=1e-12)
    assert (np.nan_to_num(got) >= 0).all()


@pytest.mark.gpu
def test_generate_dsm_end_to_end():
    """2 x 2 sliding tiles (stride 256 m, as conf/dataset/base.yaml:29) through the HIP model and the mosaic kernels
    against the oracle model + numpy mosaic."""
    from detinit import det_init_
    from oracle import torch_ref
    from tomosar2height_amd import TomoSAR2Height
    from tomosar2height_amd.config import berlin_config
    from tomosar2height_amd.generator import DSMGenerator
    from tomosar2height_amd.synthetic import berlin_tile
    dev = torch.device("cuda:0")
    cfg = berlin_config()
    ref = det_init_(torch_ref.TomoSAR2Height(cfg), seed=17)
    model = TomoSAR2Height(cfg)
    model.load_state_dict(ref.state_dict())
    model.to(dev)
    tiles = []
    for i, (x0, y0) in enumerate(((0.0, 0.0), (256.0, 0.0), (0.0, 256.0), (256.0, 256.0))):
        t = berlin_tile(40 + i, n_points=3000)
        t["min_bound"] = torch.tensor([[x0, y0, 0.0]])
        t["max_bound"] = torch.tensor([[x0 + 512.0, y0 + 512.0, 100.0]])
        tiles.append(t)
    tiles.append({"is_valid": torch.tensor([False])})                       # skipped like generator.py:133-134
    gen = DSMGenerator(model, dev, tiles, bounds=(0.0, 0.0, 768.0, 768.0))
    got = gen.generate_dsm().cpu().numpy()
    ref.eval()
    ref_tiles = []
    with torch.no_grad():
        for t in tiles[:4]:
            h = ref(input_cloud=t["inputs"])[0][0, :, :, 0].numpy()
            l, _ = mosaic_ref.col_row(t["min_bound"][0, 0].item() + 0.5, t["min_bound"][0, 1].item() + 0.5, 0.0, 768.0, (1, 1))
            _, top = mosaic_ref.col_row(t["max_bound"][0, 0].item() - 0.5, t["max_bound"][0, 1].item() - 0.5, 0.0, 768.0, (1, 1))
            ref_tiles.append((h, top, l))
    want = mosaic_ref.mosaic(ref_tiles, (768, 768), mosaic_ref.linear_blend_patch_weight((512, 512), (0.5, 0.5)))
    assert got.shape == (768, 768) and not np.isnan(got).any()
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 1e-4 * scale
