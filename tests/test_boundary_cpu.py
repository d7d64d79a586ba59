"""CPU (no GPU needed): the C-ABI library builds, loads, exports every symbol include/t2h.h declares, and the
product refuses to run without a device instead of falling back."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "t2h.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(t2h_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tomosar2height_amd import _lib
    from tomosar2height_amd.csrc import build
    build.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 36
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/t2h.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.t2h_abi_version() == _lib.ABI_VERSION


def test_host_side_helpers_need_no_gpu():
    from tomosar2height_amd import _lib
    lib = _lib.load()
    assert lib.t2h_pool_winner_stride(32) == 8 and lib.t2h_pool_winner_stride(6) == 6
    assert lib.t2h_tile_workspace_bytes(1, 131072, 8) >= 4 * 131072 * 4
    assert lib.t2h_tile_workspace_bytes(1, 10, 11) == 0            # nbits out of range
    # argument validation happens before any launch: null pointers are an error code, not a crash
    assert lib.t2h_segmean_fwd(None, None, 1, 100, 8, 0, 32, None, None, 0, None) == -1
    assert b"null pointer" in lib.t2h_last_error_string()


def test_grid_conv_dispatch_host_logic():
    """grid.py's routing (host logic, no launches): which convolutions the implicit-GEMM kernels accept, workspace
    plans of the C side, and that geometry outside them is left to the stock module."""
    from tomosar2height_amd import _lib, grid
    lib = _lib.load()
    x = torch.rand(1, 32, 16, 16)
    assert not grid.conv3x3_supported(x, torch.nn.Conv2d(32, 32, 3, padding=1))                 # host tensor
    assert not grid.upconv2x2_supported(x, torch.nn.ConvTranspose2d(32, 16, 2, stride=2))
    meta = torch.empty(1, 32, 16, 16, device="meta")
    for conv, ok in ((torch.nn.Conv2d(32, 64, 3, padding=1), None), (torch.nn.Conv2d(32, 64, 3, padding=0), False),
                     (torch.nn.Conv2d(32, 64, 3, stride=2, padding=1), False), (torch.nn.Conv2d(24, 64, 3, padding=1), False),
                     (torch.nn.Conv2d(32, 64, 5, padding=2), False), (torch.nn.Conv2d(32, 64, 3, padding=1, groups=2), False)):
        if ok is False:
            assert not grid.conv3x3_supported(meta, conv)
    # split-reduction plans: small planes with many channels need slabs, large planes do not; weight gradients always do
    assert lib.t2h_conv3x3_fwd_workspace_bytes(1, 32, 32, 512, 512) % (1024 * 512 * 4) == 0
    assert lib.t2h_conv3x3_fwd_workspace_bytes(1, 32, 32, 512, 512) > 0
    assert lib.t2h_conv3x3_dgrad_workspace_bytes(1, 512, 512, 64, 128) == 0
    assert lib.t2h_conv3x3_wgrad_workspace_bytes(1, 512, 512, 64, 128) >= (128 * 9 * 64 + 128) * 4
    assert lib.t2h_upconv2x2_wgrad_workspace_bytes(1, 32, 32, 512, 256) >= 512 * 4 * 256 * 4
    assert lib.t2h_conv3x3_fwd_workspace_bytes(0, 32, 32, 512, 512) == 0
    # geometry errors come back as codes with a message, never as a launch
    assert lib.t2h_conv3x3_fwd(1, 1, None, 1, 1, 12, 16, 32, 32, 0, None, 0, None) < 0
    assert b"powers of two" in lib.t2h_last_error_string()
    assert lib.t2h_upconv2x2_fwd(1, 1, None, 1, 1, 16, 16, 24, 32, 0, None) < 0
    assert b"multiples of 16" in lib.t2h_last_error_string()


def test_no_cpu_fallback():
    from tomosar2height_amd.tile import TileIndex
    from tomosar2height_amd import ops
    cloud = torch.rand(1, 16, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        TileIndex(cloud, 16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.coordinate2index(cloud, 16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.upsample_bilinear(torch.rand(1, 2, 4, 4), 8)
    with pytest.raises(ValueError, match="power of two"):
        TileIndex(cloud, 12)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tomosar2height_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "libt2h_oracle" not in src


def test_every_entry_point_rejects_bad_arguments_without_a_gpu():
    """The C side validates before launching: null pointers / impossible sizes come back as T2H_ERR_ARG (-1) or
    T2H_ERR_WORKSPACE (-3) with a message, never a crash -- checked here without any device."""
    from tomosar2height_amd import _lib
    lib = _lib.load()
    n = None
    cases = {
        "t2h_coordinate2index": (n, 3, 10, 16, n, n),
        "t2h_tile_build": (n, 3, 1, 10, 8, n, n, n, n, n, n, 0, n),
        "t2h_tile_build_ragged": (n, 3, 2, n, 8, n, 4, n, n, n, n, n, 0, n),
        "t2h_pool_max_fwd": (n, 32, n, 1, 8, 32, n, 32, n, n),
        "t2h_pool_max_bwd": (n, 32, n, n, 1, 8, 32, 0, n, 32, n),
        "t2h_pool_mean": (n, 32, n, 1, 8, 32, 0, n, 32, n),
        "t2h_pool_rows_fwd": (n, 32, n, n, 100, 32, n, 32, n, n),
        "t2h_pool_rows_bwd": (n, 32, n, n, n, 100, 32, 0, n, 32, n),
        "t2h_segmean_fwd": (n, n, 1, 10, 8, 0, 32, n, n, 0, n),
        "t2h_segmean_bwd": (n, n, n, 1, 10, 8, 0, 32, n, n),
        "t2h_segmean_bwd_add": (n, n, n, 1, 10, 8, 0, 32, n, n, n),
        "t2h_segsum_fwd": (n, n, 1, 10, 8, 0, 32, n, 32, n, 0, n),
        "t2h_plane_sumpool2x2": (n, 32, 1, 32, 32, n, 32, n),
        "t2h_segsum_bwd_multi": (n, n, n, 1, n, 1, 10, 8, 32, n, n, n, n),
        "t2h_cell_counts": (n, 1, 8, 0, n, n),
        "t2h_mean_bias_fwd": (n, n, n, 64, 32, n, n),
        "t2h_mean_bias_bwd": (n, n, 64, 32, n, n, n, 0, n),
        "t2h_sample_relu_cellsums": (n, n, 3, n, 1, 10, 8, 2, 0, 256, n, 256, n, n),
        "t2h_sample_relu_cellsums2": (n, n, 3, n, 1, 10, 8, 2, 0, 256, n, 256, n, 256, n, n),
        "t2h_sample_bwd_from_sums": (n, n, n, 1, n, n, 0, n, 3, n, 1, 10, 8, 0, 32, n, n, 0, n),
        "t2h_sample_bwd_from_sums_ordered": (n, n, n, 1, n, n, 0, n, 3, n, 1, 10, 8, 0, 32, n, n, 0, n, n),
        "t2h_sample_relu_cellsums_ordered": (n, n, 3, n, 1, 10, 8, 2, 0, 256, n, 256, n, 256, n, n, n),
        "t2h_cell_order_build": (n, 1, 8, 2, n, n),
        "t2h_cell_order_build_range": (n, 1, 8, 1, 3, n, n),
        "t2h_sample_fwd": (n, n, 3, 1, 10, 32, 32, n, n),
        "t2h_sample_fwd_relu": (n, n, 3, 1, 10, 32, 32, n, n, n),
        "t2h_sample_bwd": (n, n, 3, n, 1, 10, 8, 0, 32, n, n, 0, n),
        "t2h_sample_bwd_add": (n, n, 3, n, 1, 10, 8, 0, 32, n, n, n, 0, n),
        "t2h_sample_bwd_atomic": (n, n, 3, 1, 10, 32, 32, n, n),
        "t2h_sample_adjoint_build": (n, 3, n, 1, 10, 8, 0, n, n, n),
        "t2h_sample_bwd_adjoint": (n, n, n, 1, 8, 0, 32, n, n, n),
        "t2h_linear_fwd": (n, 32, n, n, n, 32, 10, 32, 32, 0, n),
        "t2h_linear_fwd_add": (n, 32, n, n, n, 32, n, 32, 10, 32, 32, 0, n),
        "t2h_linear_dgrad": (n, 32, n, n, 32, 10, 32, 32, n, 0, 0, n),
        "t2h_linear_wgrad": (n, 32, n, 32, 10, 32, 32, 0, n, n, n, 0, n),
        "t2h_upsample_bilinear_fwd": (n, n, 1, 32, 16, 16, 32, 32, n, n),
        "t2h_upsample_bilinear_bwd": (n, 1, 32, 16, 16, 32, 32, n, n),
        "t2h_upsample_bilinear_nhwc_fwd": (n, n, 1, 32, 16, 16, 32, 32, n, n),
        "t2h_upsample_bilinear_nhwc_bwd": (n, 1, 32, 16, 16, 32, 32, n, n),
        "t2h_upsample2x_nhwc_fwd": (n, 1, 32, 16, 16, n, n),
        "t2h_upsample2x_nhwc_bwd": (n, 1, 32, 16, 16, n, n),
        "t2h_bias_relu_fwd": (n, n, 100, 32, 1, n),
        "t2h_bias_relu_bwd": (n, n, n, 100, 32, 1, 0, n, n, 0, n),
        "t2h_upsample_bicubic_fwd": (n, n, 1, 32, 16, 16, 32, 32, 0, n, n),
        "t2h_upsample_bicubic_bwd": (n, 1, 32, 16, 16, 32, 32, 0, n, n),
        "t2h_sample_bicubic_fwd": (n, n, 3, 1, 10, 16, 32, 0, n, n),
        "t2h_sample_bicubic_bwd": (n, n, 3, 1, 10, 16, 32, 0, n, n),
        "t2h_sample_nearest_fwd": (n, n, 3, 1, 10, 16, 32, 0, n, n),
        "t2h_sample_nearest_bwd": (n, n, 3, 1, 10, 16, 32, 0, n, n),
        "t2h_trunk_fused_fwd": (n, 3, n, n, n, 5, n, n, n, n, 100, n, n, n, n, n, 0, n, n),
        "t2h_trunk_units_build": (n, n, 100, n, n),
        "t2h_head1x1_fwd": (n, n, 4, n, n, 100, n, n),
        "t2h_head1x1_bwd": (n, n, n, 4, n, n, 100, 0, n, n, n, 0, n),
        "t2h_relu_mask": (n, n, n, 128, n),
        "t2h_conv3x3_fwd": (n, n, n, n, 1, 32, 32, 32, 32, 0, n, 0, n),
        "t2h_conv3x3_dgrad": (n, n, n, n, 1, 32, 32, 32, 32, 0, n, 0, n),
        "t2h_conv3x3_wgrad": (n, n, n, n, 1, 32, 32, 32, 32, 0, n, 0, n),
        "t2h_conv3x3_bx3_prepare": (n, 32, 32, 0, n, n),
        "t2h_conv3x3_bx3_fwd": (n, n, n, n, 1, 32, 32, 32, 32, 0, n, 0, n),
        "t2h_conv3x3_bx3_dgrad": (n, n, n, n, 1, 32, 32, 32, 32, 0, n, 0, n),
        "t2h_conv3x3_bx3_wgrad": (n, n, n, n, 1, 32, 32, 32, 32, 0, n, 0, n),
        "t2h_gemm_bx3_prepare": (n, 64, 64, 32, 0, n, n),
        "t2h_upconv2x2_bx3_fwd": (n, n, n, n, n, 1, 16, 16, 64, 64, 0, n),
        "t2h_upconv2x2_bx3_dgrad": (n, 64, n, n, 1, 16, 16, 64, 64, 0, n, 0, n),
        "t2h_upconv2x2_bx3_wgrad": (n, 64, n, n, n, 1, 16, 32, 64, 64, 0, n, 0, n),
        "t2h_conv3x3_f16x2_prepare": (n, 32, 32, 0, n, n),
        "t2h_gemm_f16x2_prepare": (n, 64, 64, 32, 0, n, n),
        "t2h_split_weights_batch": (n, 1, n),
        "t2h_conv3x3_bx3_dgrad_rank1": (n, n, n, n, n, n, 1, 512, 512, 64, 128, 64, n),
        "t2h_gemm_bx3_wgrad": (n, 2752, n, 64, 128, 64, 2752, n, n, 64, n, 0, n),
        "t2h_gemm_bx3": (n, 64, n, n, n, 0, n, 32, 128, 64, 32, 0, n, 0, n),
        "t2h_upconv2x2_fwd": (n, n, n, n, 1, 32, 32, 32, 32, 0, n),
        "t2h_upconv2x2_fwd_add": (n, n, n, n, n, 1, 32, 32, 32, 32, 0, n),
        "t2h_upconv2x2_dgrad": (n, n, n, 1, 32, 32, 32, 32, 0, n, 0, n),
        "t2h_upconv2x2_wgrad": (n, n, n, 1, 32, 32, 32, 32, 0, n, 0, n),
        "t2h_upconv2x2_wgrad_bias": (n, n, n, n, 1, 32, 32, 32, 32, 0, n, 0, n),
        "t2h_maxpool2x2_nhwc_fwd": (n, 1, 64, 64, 32, n, n, n),
        "t2h_maxpool2x2_nhwc_bwd": (n, n, 1, 64, 64, 32, n, n),
        "t2h_maxpool2x2_nhwc_bwd_add": (n, n, 1, 64, 64, 32, n, 32, n, n),
        "t2h_mosaic_accumulate": (n, 64, 64, n, n, n, 100, 100, 0, 0, 1, n),
        "t2h_mosaic_finalize": (n, n, 100, n),
        "t2h_tile_crop_normalise": (n, 100, 0.0, 0.0, 1.0, 1.0, 512.0, 512.0, 190.2, n, n, n, n, n, 0, n),
        "t2h_tile_crop_finish": (n, n),
        "t2h_tile_crop_normalise_aug": (n, 100, 0.0, 0.0, 1.0, 1.0, 512.0, 512.0, 190.2, 0, -1, n, n, n, n, n, 0, n),
        "t2h_raster_patch": (n, 0, 1, 64, 64, 0, 0, 16, 16, 0, -1, n, n),
        "t2h_adamw_flat_step": (n, n, 4, 1e-4, 0.9, 0.999, 1e-8, 0.01, 1, 0, n),
        "t2h_conv3x3_smallcin_fwd": (n, n, n, n, 1, 32, 32, 3, 32, 0, n),
        "t2h_conv3x3_smallcin_dgrad": (n, n, n, 1, 32, 32, 3, 32, 0, n),
        "t2h_conv3x3_smallcin_wgrad": (n, n, n, n, 1, 32, 32, 3, 32, 0, n, 0, n),
        "t2h_trunk_block_fwd": (n, 3, n, n, n, 32, n, n, n, n, n, n, n, n, n, 100, n, n, n, n, 32, n, n, n),
        "t2h_trunk_block_bwd": (n, 32, n, 32, n, n, n, n, n, n, n, n, 32, n, 32, n, 3, n, n, n, n, n, 100, n, n, 0, n),
        "t2h_trunk_block_reduce": (n, 100, 0, 0, n, n, n, n, n, n, n, 0, n),
        "t2h_scatter_max_fwd": (n, 32, n, n, 1, 100, 4, 32, n, n, n),
        "t2h_scatter_max_bwd": (n, n, 1, 32, 100, 256, n, n),
        "t2h_nchw_to_nhwc": (n, 1, 32, 64, n, n),
        "t2h_nhwc_to_nchw": (n, 1, 32, 64, n, n),
    }
    launching = [k for k, (res, _a) in _lib.SIGNATURES.items() if res is _lib._i and k not in ("t2h_abi_version", "t2h_pool_winner_stride", "t2h_adamw_chunk_elems", "t2h_conv3x3_bx3_supported", "t2h_gemm_bx3_supported", "t2h_gemm_bx3_wgrad_supported", "t2h_conv3x3_bx3_dgrad_rank1_supported", "t2h_upconv2x2_bx3_supported",
                                                                                    "t2h_reduce_capture_begin", "t2h_reduce_capture_pending", "t2h_reduce_capture_end",
                                                                                    "t2h_debug_poison_lds")]          # (test instrumentation: no arguments to reject)
    assert sorted(cases) == sorted(launching), set(launching) ^ set(cases)
    for name, args in cases.items():
        rc = getattr(lib, name)(*args)
        assert rc in (-1, -3), f"{name} returned {rc}"
        assert len(lib.t2h_last_error_string()) > 8


def test_in_place_gradient_joins_only_into_solely_owned_buffers():
    """mlp._join_plane_grad / deferred._DeferredLevel accumulate into the gradient tensor autograd hands them only
    when nobody else still reads it (mlp.sole_owner).  torch's add backward hands the SAME tensor to both producers: the node
    that runs first must not write into it."""
    import torch
    from tomosar2height_amd import mlp
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            seen.append(mlp.sole_owner(g))
            return g * 2

    class ChannelsLastGrad(torch.autograd.Function):        # returns a permuted view of a fresh NHWC buffer, as grid.py does
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            d = torch.empty(1, 3, 3, 4)
            d.copy_(g.permute(0, 2, 3, 1))
            return d.permute(0, 3, 1, 2)

    x = torch.randn(1, 4, 3, 3, requires_grad=True)
    (Probe.apply(x) * 3).sum().backward()
    assert seen == [True]
    seen.clear()
    z = Probe.apply(x) + Probe.apply(x * 1.5)              # one gradient buffer queued for two nodes
    (z * z).sum().backward()
    assert seen == [False, True]
    seen.clear()
    ChannelsLastGrad.apply(Probe.apply(x)).sum().backward()
    assert seen == [True]
