"""-m gpu: the configuration bench.py TIMES (BASELINE.json configs[1]: `berlin_tile`, N = 131072 points, channels_last,
fp32) -- and its cloud+image sibling (configs[2]'s fp32 baseline) -- against the CPU torch oracle with the same weights:
every height and every parameter gradient (reference: model.py:54-67 through trainer.py:61-70).

Kernel selection on the HIP path is size-keyed (split plans, 512 / 768 / 1024-workgroup grids, the transposed-matrix
sample backward below 4 rows per pixel, `segmean_cells` at >= 16 points per cell, the persistent trunk), so the smaller
oracle comparisons of test_hip_model.py do not exercise the kernels the benchmark runs; this one does.  The oracle's
forward + backward costs ~10 s (cloud-only) on the host at this size."""
import numpy as np
import pytest
import torch

from detinit import det_init_

pytestmark = pytest.mark.gpu


def _rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() / (np.abs(want).max() + 1e-30)


@pytest.mark.parametrize("use_image", [False, True], ids=["cloud_only", "cloud_image"])
def test_benchmarked_configuration_matches_the_oracle_at_full_size(use_image):
    from oracle import torch_ref
    from tomosar2height_amd import TomoSAR2Height, fallback_counts
    from tomosar2height_amd.config import berlin_config
    from tomosar2height_amd.synthetic import DEFAULT_POINTS, berlin_tile
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = berlin_config(use_image=use_image)
    ref = det_init_(torch_ref.TomoSAR2Height(cfg), seed=31)
    model = TomoSAR2Height(cfg)
    model.load_state_dict(ref.state_dict(), strict=True)
    model.to(dev)
    model.set_channels_last(True)                                   # what bench.py runs
    tile = berlin_tile(seed=1000, n_points=DEFAULT_POINTS, with_image=use_image)      # bench.py's rank-0 tile 0
    assert tile["inputs"].shape == (1, 131072, 3)
    cloud, image = tile["inputs"], tile.get("image")
    # smooth (linear) loss with fixed random weights: the gradient does not depend on sign(pa - dsm), see
    # test_hip_model.py::test_model_vs_torch_oracle_all_grads
    w = torch.randn(512, 512, generator=torch.Generator().manual_seed(1))

    pa_ref, _ = ref(input_cloud=cloud, input_image=image)
    (pa_ref.squeeze() * w).mean().backward()
    before = sum(fallback_counts().values())
    pa, pb = model(input_cloud=cloud.to(dev), input_image=None if image is None else image.to(dev))
    loss = (pa.squeeze() * w.to(dev)).mean()
    loss.backward()
    torch.cuda.synchronize()
    assert sum(fallback_counts().values()) == before, "a vendor-library fallback ran in the benchmarked configuration"
    assert pb is None and pa.shape == (1, 512, 512, 1)

    err = _rel(pa.detach().cpu().numpy(), pa_ref.detach().numpy())
    assert err <= 1e-4, f"heights: max rel err {err:.3e} > 1e-4 (north_star tolerance)"

    # gradient resolution: see test_model_vs_torch_oracle_all_grads (ReLU / max-pool / arg-max masks flip under 1e-7
    # activation differences; the oracle itself moved to the device deviates from its CPU run by up to 6e-3 max-normalised)
    rows = []
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert (p.grad is None) == (q.grad is None), k
        if p.grad is None:
            continue
        got, want = p.grad.cpu().double(), q.grad.double()
        rows.append((_rel(got.numpy(), want.numpy()), ((got - want).norm() / (want.norm() + 1e-30)).item(), k))
    rows.sort(reverse=True)
    print(f"[full size, image={use_image}] heights {err:.2e}; gradients (max-normalised, L2, name), worst first:")
    for mx, l2, k in rows[:8]:
        print(f"    {mx:.2e} {l2:.2e} {k}")
    # Resolution of the max-normalised criterion on SMALL planes: a ReLU unit of a 32 x 32 (16 x 16) plane whose mask flips
    # under a 1e-6 activation difference switches one of only 1024 (256) terms of ONE output-channel row of that layer's
    # weight gradient, i.e. moves that row by ~1/sqrt(1024) = 3 % (measured: image_encoder.up_convs.0.conv2.weight 1.1e-2 with
    # L2 2.3e-3, everything else < 6.1e-3).  The image U-Net's three deepest levels therefore get 3e-2 in the max norm; the L2
    # criterion (3e-3) holds everywhere, and the mask-pinned checks of test_hip_masks.py / test_hip_conv.py pin the arithmetic.
    small_planes = ("image_encoder.down_convs.4.", "image_encoder.down_convs.5.", "image_encoder.up_convs.0.")
    for mx, l2, k in rows:
        lim_mx = 3e-2 if k.startswith(small_planes) else 1e-2
        assert mx <= lim_mx, f"{k}: max-normalised gradient error {mx:.2e} > {lim_mx:g}"
        assert l2 <= 3e-3, f"{k}: L2 relative gradient error {l2:.2e}"
