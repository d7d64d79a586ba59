"""-m gpu: micro-batched accumulation window.  The reference feeds the 64 tiles of an optimizer step one at
a time only because their point counts differ (tomosar2height.yaml:40, trainer.py:72-89); they are independent and their
gradients are SUMMED.  Here several tiles share one ragged ``TileIndex`` (``t2h_tile_build_ragged``) and one set of launches;
these tests pin that to the one-tile-at-a-time path: the index bit for bit, heights to 2e-5, the accumulated gradient bucket to
fp32 re-association (2e-5 of its max-norm), and the reference Trainer's post-AdamW weights (``trainer_accumulation`` fixture)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from detinit import det_init_, synth_cloud

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("counts", [(5000, 1, 7777, 2048), (3, 131072), (40000,)])
def test_ragged_tile_index_equals_the_single_tile_indices(counts):
    from tomosar2height_amd.tile import TileIndex
    clouds = [synth_cloud(n, seed=300 + i).to(_dev()) for i, n in enumerate(counts)]
    rag = TileIndex(clouds, 256)
    assert rag.ragged and rag.B == len(counts) and rag.n_points == sum(counts) and rag.N == -sum(counts) and rag.dim == 4
    m0, row = 256 * 256, 0
    for b, c in enumerate(clouds):
        one = TileIndex(c, 256)
        n = counts[b]
        assert torch.equal(rag.pts[row:row + n, :3], one.pts)
        assert torch.equal(rag.pts[row:row + n, 3].view(torch.int32), torch.full((n,), b, dtype=torch.int32, device=_dev()))
        assert torch.equal(rag.perm[row:row + n], one.perm)
        assert torch.equal(rag.cell[row:row + n], one.cell + b * m0)
        assert torch.equal(rag.off0[b * m0:(b + 1) * m0 + 1], one.off0 + row)
        row += n
    assert rag.out_of_domain() == 0


def test_ragged_batch_rejects_empty_tiles_and_mixed_dims():
    from tomosar2height_amd.tile import TileIndex
    a = synth_cloud(100, seed=1).to(_dev())
    with pytest.raises(ValueError, match="at least one point"):
        TileIndex([a, a[:, :0]], 256)
    with pytest.raises(TypeError):
        TileIndex([a, a[..., :2]], 256)


def _berlin_model(seed=21):
    from tomosar2height_amd import TomoSAR2Height
    from tomosar2height_amd.config import berlin_config
    model = det_init_(TomoSAR2Height(berlin_config()), seed=seed).to(_dev())
    model.set_channels_last(True)
    return model


def test_ragged_forward_equals_single_tile_forwards():
    """Training-mode forward of a ragged batch (dense tiles: the grid-first / deferred / on-chip kernels engage) against the
    same tiles one at a time: per-tile sums are formed in the same order, so heights agree to fp32 noise of the batched
    grid-side products (2e-5 of the height scale)."""
    from tomosar2height_amd.synthetic import berlin_tile
    model = _berlin_model()
    counts = (131072, 98304, 120000)
    clouds = [berlin_tile(seed=40 + i, n_points=n)["inputs"].to(_dev()) for i, n in enumerate(counts)]
    with torch.no_grad():
        model.train()
        pa, _ = model(input_cloud=clouds)
        assert pa.shape == (3, 512, 512, 1)
        for i, c in enumerate(clouds):
            one, _ = model(input_cloud=c)
            err = ((pa[i] - one[0]).abs().max() / one.abs().max()).item()
            assert err <= 2e-5, (i, err)


def test_micro_batched_window_gives_the_same_accumulated_gradient():
    """Trainer.train_step([4 ragged tiles]) against four one-tile train_steps on the same weights: the flat gradient bucket the
    optimizer consumes (``on_reduced``) to 2e-5 of its max-norm, the summed loss to 1e-6 relative; and the micro-batch must not
    straddle an optimizer boundary."""
    from tomosar2height_amd.synthetic import berlin_tile
    from tomosar2height_amd.trainer import Trainer
    counts = (131072, 110000, 90000, 125000)
    tiles = []
    for i, n in enumerate(counts):
        t = berlin_tile(seed=60 + i, n_points=n)
        tiles.append({"inputs": t["inputs"].to(_dev()), "dsm": t["dsm"].to(_dev())})
    res = {}
    for mode in ("single", "micro"):
        model = _berlin_model(seed=23)
        tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=_dev(), optimize_every=4, use_cloud=True)
        seen = []
        tr.on_reduced = lambda flat: seen.append(flat.clone())
        if mode == "single":
            stepped = [tr.train_step(t) for t in tiles]
            assert stepped == [False, False, False, True]
        else:
            assert tr.train_step(tiles) is True
        res[mode] = (seen[0], float(tr.last_avg_loss))
    (g1, l1), (g4, l4) = res["single"], res["micro"]
    assert abs(l1 - l4) <= 1e-6 * abs(l1), (l1, l4)
    rel = ((g1 - g4).abs().max() / g1.abs().max()).item()
    print(f"[micro-batch] bucket max rel diff {rel:.2e}, avg loss {l1:.6f} / {l4:.6f}")
    assert rel <= 2e-5, rel
    model = _berlin_model(seed=23)
    tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=_dev(), optimize_every=4, use_cloud=True)
    tr.train_step(tiles[0])
    with pytest.raises(ValueError, match="boundary"):
        tr.train_step(tiles)


def test_coalesced_single_tile_api_gives_the_window_of_the_tile_by_tile_loop():
    """``Trainer.train_step(tile)`` called tile by tile, as the reference's loop does (train.py:147-152): by default the trainer
    holds single tiles back and issues them four at a time as ragged micro-batches (``coalesce_tiles``), never across the
    optimizer boundary.  Against the same calls with coalescing off: the same calls end with an optimizer step, the flat gradient
    the optimizer consumes agrees to 2e-5 of its max-norm, the window's average loss to 1e-6; ``flush_gradients()`` in the
    middle of a window issues what is held; a tile handed over as a one-element list or with a prebuilt index keeps its place
    in the order."""
    from tomosar2height_amd.synthetic import berlin_tile
    from tomosar2height_amd.trainer import Trainer
    counts = (60000, 48000, 52000, 40000, 56000, 44000, 50000)
    tiles = []
    for i, n in enumerate(counts):
        t = berlin_tile(seed=80 + i, n_points=n)
        tiles.append({"inputs": t["inputs"].to(_dev()), "dsm": t["dsm"].to(_dev())})
    res = {}
    for co in (1, 4):
        model = _berlin_model(seed=29)
        tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=_dev(), optimize_every=len(tiles), use_cloud=True)
        assert tr.coalesce_tiles == 4, "coalescing is the default"
        tr.coalesce_tiles = co
        seen, batches, inner = [], [], tr._losses
        tr.on_reduced = lambda flat: seen.append(flat.clone())

        def rec(data, thr, inner=inner, batches=batches):
            batches.append(len(data) if isinstance(data, (list, tuple)) else 1)
            return inner(data, thr)
        tr._losses = rec
        stepped = [tr.train_step(t) for t in tiles]
        assert stepped == [False] * (len(tiles) - 1) + [True], stepped
        # tile 0 lays the gradient bucket out (alone); then 4 held + the 2 that complete the window
        assert batches == ([1] * 7 if co == 1 else [1, 4, 2]), batches
        res[co] = (seen[0], float(tr.last_avg_loss))
        assert tr.accumulated_steps == 0 and not tr._coalesced
        if co == 4:
            # a flush in the middle of the window issues the held tiles; lists and prepared tiles keep the order of the calls
            del batches[:]
            tr.train_step(tiles[0])
            tr.train_step(tiles[1])
            assert batches == [] and len(tr._coalesced) == 2
            tr.flush_gradients()
            assert batches == [2] and tr.accumulated_steps == 2 and not tr._coalesced
            tr.train_step(tiles[2])
            tr.train_step([tiles[3]])                       # (a list is issued as given: the held tile goes first)
            assert batches == [2, 1, 1], batches
            tr.train_step(tiles[4])
            tr.train_step(tr.prepare(tiles[5]))             # (prebuilt index: issued on its own, after the held tile)
            assert batches == [2, 1, 1, 1, 1], batches
            assert tr.train_step(tiles[6]) is True
            g2 = seen[1]
            rel2 = ((g2 - res[1][0]).abs().max() / res[1][0].abs().max()).item()
            assert rel2 <= 2e-5, rel2
    (g1, l1), (g4, l4) = res[1], res[4]
    assert abs(l1 - l4) <= 1e-6 * abs(l1), (l1, l4)
    rel = ((g1 - g4).abs().max() / g1.abs().max()).item()
    print(f"[coalesced] bucket max rel diff {rel:.2e}, avg loss {l1:.6f} / {l4:.6f}")
    assert rel <= 2e-5, rel


def _dsm(lo):
    return torch.from_numpy(lo).repeat_interleave(8, -2).repeat_interleave(8, -1)     # (as tests/test_hip_model.py)


def test_trainer_accumulation_golden_as_one_micro_batch():
    """The reference Trainer's own post-AdamW weights after 3 accumulated tiles (trainer.py:47-89; fixture generated by running
    it) reproduced with the three tiles as ONE micro-batch."""
    import tomosar2height_amd as t2h
    from tomosar2height_amd import TomoSAR2Height
    from tomosar2height_amd.config import berlin_config
    from tomosar2height_amd.trainer import Trainer
    g = load_golden("trainer_accumulation")
    cfg = berlin_config()
    t2h.allow_library_fallback(True).set()             # reduced widths (start_filts = 8): below the kernels' 16-channel slabs
    cfg.model.encoder_kwargs.plane_resolution = 16
    cfg.model.encoder_kwargs.unet_kwargs.depth = 3
    cfg.model.encoder_kwargs.unet_kwargs.start_filts = 8
    model = det_init_(TomoSAR2Height(cfg), seed=9).to(_dev())
    tr = Trainer(model, torch.optim.AdamW(model.parameters(), lr=1e-4), device=_dev(), optimize_every=3, use_cloud=True)
    batch = [{"inputs": torch.from_numpy(g[f"cloud_{t}"]), "dsm": _dsm(g[f"dsm_lo_{t}"])[None]} for t in range(3)]
    assert tr.train_step(batch) is True
    np.testing.assert_allclose(float(tr.last_avg_loss), float(g["last_avg_loss"]), rtol=1e-5)
    params = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("after."):
            np.testing.assert_allclose(params[k[len("after."):]].detach().cpu().numpy(), g[k], rtol=1e-4, atol=2e-6)
