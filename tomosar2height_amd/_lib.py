"""ctypes binding of libt2h_hip.so (include/t2h.h).  This is the ONLY way the package reaches the
device: there is no torch/CPU fallback -- a missing library or a non-GPU tensor raises.

torch is imported first on purpose: it loads its bundled libamdhip64.so.7, and the dynamic linker then
resolves our DT_NEEDED entry of the same SONAME to that already-loaded runtime, so device pointers,
streams and the caching allocator are shared with PyTorch.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# T2H_LIBRARY: load another build of the same ABI (A/B runs of a kernel change; a site-specific install path)
LIB_PATH = os.environ.get("T2H_LIBRARY") or os.path.join(_HERE, "libt2h_hip.so")
ABI_VERSION = 19

_vp, _i, _i64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/t2h.h one to one
SIGNATURES = {
    "t2h_abi_version": (_i, []),
    "t2h_last_error_string": (ctypes.c_char_p, []),
    "t2h_last_kernel_name": (ctypes.c_char_p, []),
    "t2h_clear_kernel_name": (None, []),
    "t2h_debug_poison_lds": (_i, [_vp]),
    "t2h_coordinate2index": (_i, [_vp, _i, _i64, _i, _vp, _vp]),
    "t2h_tile_workspace_bytes": (_sz, [_i, _i, _i]),
    "t2h_tile_build": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "t2h_tile_ragged_workspace_bytes": (_sz, [_i, _i64, _i, _i]),
    "t2h_tile_build_ragged": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "t2h_pool_winner_stride": (_i, [_i]),
    "t2h_pool_max_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "t2h_pool_max_bwd": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "t2h_pool_mean": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "t2h_scatter_max_fwd": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "t2h_scatter_max_bwd": (_i, [_vp, _vp, _i, _i, _i, _i64, _vp, _vp]),
    "t2h_pool_rows_fwd": (_i, [_vp, _i, _vp, _vp, _i64, _i, _vp, _i, _vp, _vp]),
    "t2h_pool_rows_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _i64, _i, _i, _vp, _i, _vp]),
    "t2h_segmean_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "t2h_segmean_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "t2h_segmean_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2h_segmean_bwd_add": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "t2h_segsum_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "t2h_plane_sumpool2x2": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "t2h_segsum_bwd_multi": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "t2h_cell_counts": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "t2h_mean_bias_fwd": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "t2h_mean_bias_bwd_workspace_bytes": (_sz, [_i64, _i]),
    "t2h_mean_bias_bwd": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _sz, _vp]),
    "t2h_sample_relu_cellsums": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "t2h_sample_relu_cellsums2": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "t2h_cell_order_len": (_sz, [_i, _i, _i]),
    "t2h_cell_order_build": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "t2h_cell_order_build_range": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "t2h_sample_relu_cellsums_ordered": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "t2h_sample_bwd_from_sums": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "t2h_sample_bwd_from_sums_ordered": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "t2h_sample_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2h_sample_fwd_relu": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "t2h_sample_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "t2h_sample_bwd": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "t2h_sample_adjoint_offsets_len": (_sz, [_i, _i, _i]),
    "t2h_sample_adjoint_build": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "t2h_sample_bwd_adjoint": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "t2h_sample_bwd_add": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "t2h_sample_bwd_atomic": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2h_linear_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "t2h_linear_fwd_add": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "t2h_linear_dgrad": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "t2h_linear_wgrad_workspace_bytes": (_sz, [_i, _i, _i]),
    "t2h_linear_wgrad": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "t2h_upsample_bilinear_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2h_upsample_bilinear_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2h_bias_relu_fwd": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "t2h_bias_relu_bwd_workspace_bytes": (_sz, [_i64, _i]),
    "t2h_bias_relu_bwd": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "t2h_head1x1_fwd": (_i, [_vp, _vp, _i, _vp, _vp, _i64, _vp, _vp]),
    "t2h_head1x1_bwd_workspace_bytes": (_sz, [_i64, _i]),
    "t2h_head1x1_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i64, _i, _vp, _vp, _vp, _sz, _vp]),
    "t2h_upsample_bilinear_nhwc_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2h_upsample_bilinear_nhwc_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2h_upsample2x_nhwc_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "t2h_upsample2x_nhwc_bwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "t2h_relu_mask": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "t2h_conv3x3_fwd_workspace_bytes": (_sz, [_i] * 5),
    "t2h_conv3x3_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_conv3x3_dgrad_workspace_bytes": (_sz, [_i] * 5),
    "t2h_conv3x3_dgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_conv3x3_wgrad_workspace_bytes": (_sz, [_i] * 5),
    "t2h_conv3x3_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_conv3x3_bx3_supported": (_i, [_i] * 5),
    "t2h_conv3x3_bx3_weights_bytes": (_sz, [_i, _i]),
    "t2h_conv3x3_bx3_prepare": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "t2h_conv3x3_bx3_fwd_workspace_bytes": (_sz, [_i] * 5),
    "t2h_conv3x3_bx3_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_conv3x3_bx3_dgrad_workspace_bytes": (_sz, [_i] * 5),
    "t2h_conv3x3_bx3_dgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_conv3x3_bx3_wgrad_workspace_bytes": (_sz, [_i] * 5),
    "t2h_conv3x3_bx3_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_gemm_bx3_supported": (_i, [_i64, _i, _i]),
    "t2h_gemm_bx3_weights_bytes": (_sz, [_i, _i]),
    "t2h_gemm_bx3_prepare": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "t2h_gemm_bx3_workspace_bytes": (_sz, [_i64, _i, _i]),
    "t2h_upconv2x2_bx3_supported": (_i, [_i, _i, _i, _i, _i]),
    "t2h_upconv2x2_bx3_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "t2h_upconv2x2_bx3_dgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "t2h_upconv2x2_bx3_dgrad": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_upconv2x2_bx3_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "t2h_upconv2x2_bx3_wgrad": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_conv3x3_f16x2_weights_bytes": (_sz, [_i, _i]),
    "t2h_conv3x3_f16x2_prepare": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "t2h_gemm_f16x2_weights_bytes": (_sz, [_i, _i]),
    "t2h_gemm_f16x2_prepare": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "t2h_split_weights_batch": (_i, [_vp, _i, _vp]),
    "t2h_conv3x3_bx3_dgrad_rank1_supported": (_i, [_i, _i, _i, _i, _i]),
    "t2h_conv3x3_bx3_dgrad_rank1": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "t2h_gemm_bx3_wgrad_supported": (_i, [_i64, _i, _i]),
    "t2h_gemm_bx3_wgrad_workspace_bytes": (_sz, [_i64, _i, _i]),
    "t2h_gemm_bx3_wgrad": (_i, [_vp, _i, _vp, _i, _i64, _i, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "t2h_gemm_bx3": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i64, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_reduce_capture_begin": (_i, []),
    "t2h_reduce_capture_pending": (_i, []),
    "t2h_reduce_capture_end": (_i, [_vp]),
    "t2h_upconv2x2_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "t2h_upconv2x2_fwd_add": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "t2h_upconv2x2_dgrad_workspace_bytes": (_sz, [_i] * 5),
    "t2h_upconv2x2_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_upconv2x2_wgrad_workspace_bytes": (_sz, [_i] * 5),
    "t2h_upconv2x2_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_upconv2x2_wgrad_bias": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_maxpool2x2_nhwc_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "t2h_maxpool2x2_nhwc_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "t2h_maxpool2x2_nhwc_bwd_add": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "t2h_mosaic_accumulate": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "t2h_mosaic_finalize": (_i, [_vp, _vp, _i64, _vp]),
    "t2h_tile_crop_workspace_bytes": (_sz, [_i64]),
    "t2h_tile_crop_normalise": (_i, [_vp, _i64] + [ctypes.c_double] * 7 + [_vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "t2h_tile_crop_finish": (_i, [_vp, _vp]),
    "t2h_tile_crop_normalise_aug": (_i, [_vp, _i64] + [ctypes.c_double] * 7 + [_i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "t2h_raster_patch": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2h_conv3x3_smallcin_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "t2h_conv3x3_smallcin_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "t2h_conv3x3_smallcin_wgrad_workspace_bytes": (_sz, [_i, _i]),
    "t2h_conv3x3_smallcin_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "t2h_trunk_block_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp] + [_vp] * 7 + [_i64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "t2h_upsample_bicubic_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2h_upsample_bicubic_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2h_sample_bicubic_fwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _i, _vp, _vp]),
    "t2h_sample_bicubic_bwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _i, _vp, _vp]),
    "t2h_sample_nearest_fwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _i, _vp, _vp]),
    "t2h_sample_nearest_bwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _i, _vp, _vp]),
    "t2h_trunk_units_words": (_i64, [_i64]),
    "t2h_trunk_units_build": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "t2h_trunk_fused_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "t2h_trunk_block_bwd_workspace_bytes": (_sz, [_i64]),
    "t2h_trunk_block_bwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp,
                                 _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "t2h_trunk_block_reduce": (_i, [_vp, _i64, _i, _i] + [_vp] * 7 + [_i, _vp]),
    "t2h_adamw_chunk_elems": (_i, []),
    "t2h_adamw_flat_step": (_i, [_vp, _vp, _i] + [ctypes.c_double] * 5 + [_i64, _i, _vp]),
    "t2h_nchw_to_nhwc": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "t2h_nhwc_to_nchw": (_i, [_vp, _i, _i, _i, _vp, _vp]),
}

RELU_IN, RELU_OUT, ACCUM, BF16, BF16X3, DEFER_REDUCE, F16X2 = 1, 2, 4, 8, 16, 32, 64


class PrepDesc(ctypes.Structure):
    """``t2h_prep_desc`` of include/t2h.h (one prepared weight buffer of ``t2h_split_weights_batch``)."""
    _fields_ = [("w", ctypes.c_void_p), ("wf", ctypes.c_void_p), ("kind", ctypes.c_int), ("h2", ctypes.c_int), ("a", ctypes.c_int),
                ("b", ctypes.c_int), ("ldw", ctypes.c_int), ("maxslot", ctypes.c_int), ("trailer_word", ctypes.c_uint),
                ("reserved", ctypes.c_uint)]


class reduce_capture:
    """``with _lib.reduce_capture():`` around a backward pass: weight-gradient calls that opt in (``defer_reduce()`` -> flag
    T2H_DEFER_REDUCE) record their slab reduction instead of launching it; leaving the block runs them all in one launch per 24
    (include/t2h.h, t2h_reduce_capture_begin).  The workspaces of recorded calls are kept alive here until then."""
    active = None

    def __init__(self, enabled: bool = True):
        self.enabled = enabled and reduce_capture.active is None
        self.keep = []
        self.outputs = set()

    def __enter__(self):
        if self.enabled:
            check(load().t2h_reduce_capture_begin(), "t2h_reduce_capture_begin")
            reduce_capture.active = self
        return self

    def __exit__(self, *exc):
        if self.enabled:
            reduce_capture.active = None
            rc = load().t2h_reduce_capture_end(stream())
            cur = torch.cuda.current_stream() if self.keep and self.keep[0].is_cuda else None
            for ws in self.keep:                 # (slabs written on a side stream are read by this launch on the current one)
                ws.record_stream(cur)
            self.keep.clear()
            self.outputs.clear()
            if exc[0] is None:
                check(rc, "t2h_reduce_capture_end")


def defer_reduce(ws=None, out=None) -> int:
    """Flag for a weight-gradient call whose outputs nobody reads before the current backward pass ends (0 when no capture is
    active); ``ws``: the call's workspace, kept alive until the batched reduction has run.  ``out``: the gradient tensor -- a second
    call into the same tensor within one capture is NOT deferred (it reduces at once, on its own stream), so the library never has
    to flush the recorded ones early, possibly on a stream that has not been joined with their producers'."""
    cap = reduce_capture.active
    if cap is None:
        return 0
    if out is not None:
        key = out.data_ptr()
        if key in cap.outputs:
            return 0
        cap.outputs.add(key)
    if ws is not None:
        cap.keep.append(ws)
    return DEFER_REDUCE

_lib = None


class T2HLibraryError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------------- library fallbacks
# A few shapes off the reference's default configurations have no t2h kernel (odd Linear widths of the per-pixel FC
# decoder, convolutions other than 3x3/s1/p1, 1x1 and 2x2/s2-transposed, the NCHW grid side).  They used to run on
# MIOpen / rocBLAS / ATen silently; now such a call RAISES unless fallbacks are allowed, and every fallback taken is
# counted, so a regression that flips a layer of the default path onto a vendor library cannot pass the tests.
_fallback_allowed = os.environ.get("T2H_ALLOW_LIBRARY_FALLBACK", "0") == "1"
_fallback_counts = {}


def library_fallback(what: str):
    """Call right before computing with a vendor-library / ATen op in place of a t2h kernel."""
    if not _fallback_allowed:
        raise T2HLibraryError(
            f"{what}: no t2h kernel covers this shape, and library fallbacks are off.  Wrap the call in "
            "`with tomosar2height_amd.allow_library_fallback():` (or set T2H_ALLOW_LIBRARY_FALLBACK=1) to run it on "
            "MIOpen / rocBLAS / ATen instead.")
    _fallback_counts[what] = _fallback_counts.get(what, 0) + 1


class allow_library_fallback:
    """Context manager (or ``allow_library_fallback(True).set()`` for good): vendor-library fallbacks may be taken."""

    def __init__(self, allowed: bool = True):
        self.allowed = allowed

    def set(self):
        global _fallback_allowed
        _fallback_allowed = self.allowed
        return self

    def __enter__(self):
        global _fallback_allowed
        self.prev, _fallback_allowed = _fallback_allowed, self.allowed
        return self

    def __exit__(self, *exc):
        global _fallback_allowed
        _fallback_allowed = self.prev


def fallback_counts(reset: bool = False) -> dict:
    out = dict(_fallback_counts)
    if reset:
        _fallback_counts.clear()
    return out


def load():
    """Load (once) and type the library.  Raises T2HLibraryError if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise T2HLibraryError(
            f"{LIB_PATH} not found: build it with `python -m tomosar2height_amd.csrc.build` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise T2HLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    if lib.t2h_abi_version() != ABI_VERSION:
        raise T2HLibraryError(f"ABI version mismatch: library {lib.t2h_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().t2h_last_error_string().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


class KernelTimeline:
    """Opt-in per-call HIP-event timing of the C-ABI launches (used by bench.py for the roofline figures).

    Events are recorded on torch's current stream -- the stream every t2h launch goes to -- right before and
    after the call.  ``summary()`` must be called after a device synchronise."""

    def __init__(self):
        self.records = []          # (name, algorithmic_bytes, flops, start_event, end_event, kernel symbol)

    def __enter__(self):
        global _timeline
        self._prev, _timeline = _timeline, self
        return self

    def __exit__(self, *exc):
        global _timeline
        _timeline = self._prev

    def summary(self):
        out = {}
        for name, nbytes, flops, s, e, symbol in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "bytes": 0, "flops": 0, "symbol": symbol})
            d["calls"] += 1
            d["ms"] += s.elapsed_time(e)
            d["bytes"] += nbytes
            d["flops"] += flops
        return out


_timeline = None


def timing() -> bool:
    """True while a KernelTimeline records: call sites build their (f-string) tags only then."""
    return _timeline is not None


_fn_cache = {}


# test instrumentation: NaN patterns into every free CU's LDS before every entry point (t2h_debug_poison_lds, include/t2h.h)
_POISON_LDS = os.environ.get("T2H_POISON_LDS", "0") == "1"


def call(name: str, *args, nbytes: int = 0, flops: int = 0, tag: str = None):
    """Invoke ``name`` from the library, raising on a non-zero return code.  ``nbytes`` / ``flops`` = the
    ALGORITHMIC HBM bytes / floating-point operations of this launch, only used when a
    KernelTimeline is active."""
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    if _POISON_LDS:
        load().t2h_debug_poison_lds(stream())
    tl = _timeline
    if tl is None:
        rc = fn(*args)
        if rc != 0:
            check(rc, name)
        return
    else:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib = load()
        lib.t2h_clear_kernel_name()
        s.record()
        rc = fn(*args)
        e.record()
        # the device kernel behind the entry point (as rocprofv3 names it); entry points that do not note one are
        # single-kernel and keep their own name
        symbol = lib.t2h_last_kernel_name().decode() or name
        tl.records.append((tag or name, nbytes, flops, s, e, symbol))
    check(rc, name)


def ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


_ws_cache = {}


def ws_bytes(name: str, *args) -> int:
    """``name(*args)`` of the library's ``*_workspace_bytes`` queries, memoised: they are pure functions of the shape, and a
    tile-step asks the same ~60 questions every time (a ctypes round trip each)."""
    key = (name, args)
    hit = _ws_cache.get(key)
    if hit is None:
        hit = _ws_cache[key] = int(getattr(load(), name)(*args))
    return hit


def workspace(nbytes: int, device) -> torch.Tensor:
    """Scratch buffer from PyTorch's caching allocator (the library itself never allocates)."""
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


_on_stream = None       # raw hipStream_t while an ``on_stream`` block is active
_set_stream = getattr(torch._C, "_cuda_setStream", None)


class on_stream:
    """``with _lib.on_stream(side, back):`` -- what ``with torch.cuda.stream(side):`` does for a block that was entered with
    ``back`` current (launches AND allocations of the block belong to ``side``: its workspaces are recycled in the side stream's
    own order), without the context manager's bookkeeping: ``torch.cuda.stream(...)`` / ``torch.cuda.current_stream()`` each cost
    a device-count query (13 us; ~110 of them per tile-step with the weight gradients on a side stream = 1.4 ms of host time).
    Tensors the block reads must be ``record_stream``-ed by the caller as with any side stream."""

    def __init__(self, st, back):
        self.st, self.back = st, back

    def __enter__(self):
        global _on_stream
        st = self.st
        self.prev, _on_stream = _on_stream, st.cuda_stream
        if _set_stream is not None:
            _set_stream(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)
        else:
            torch.cuda.set_stream(st)
        return self

    def __exit__(self, *exc):
        global _on_stream
        _on_stream = self.prev
        b = self.back
        if _set_stream is not None:
            _set_stream(stream_id=b.stream_id, device_index=b.device_index, device_type=b.device_type)
        else:
            torch.cuda.set_stream(b)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> int:
    """The current torch stream of the current device as a raw hipStream_t.  Called once per launch (~470 times per
    tile-step): the raw getter avoids constructing a ``torch.cuda.Stream`` object each time (~3 us -> ~0.3 us)."""
    o = _on_stream
    if o is not None:
        return o
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_CACHE_READY = os.environ.get("T2H_CACHE_READY", "1") != "0"      # 0: the r04-r06 behaviour, to show that the tests catch it


class Ready:
    """Who may read a device buffer that some stream filled lazily (a cache entry: split weights, composed maps): the stream that
    filled it, and any other stream AFTER it has waited for the event recorded behind the fill.  r06: the tile pipeline runs the
    forwards of consecutive micro-batches on two streams that are ordered by nothing but the weights -- an entry created by the
    first four-tile forward (shapes a single tile does not use) was read by the next forward, on the other stream, before its
    fill had run: garbage weights in one product of one forward, 1 window in 150 (profiles/r06_coresidency.txt section 7)."""
    __slots__ = ("event", "seen")

    def __init__(self):
        self.event, self.seen = None, None

    def mark(self):
        """The buffer's fill has just been issued on the current stream."""
        self.event = torch.cuda.Event()
        self.event.record()
        self.seen = {stream()}

    def wait(self):
        """Called before every use: one dictionary look-up on the stream that already knows the buffer."""
        cur = stream()
        if self.seen is None or cur in self.seen or not _CACHE_READY:
            return
        if not torch.cuda.is_current_stream_capturing():       # (a capture is preceded by warm-up passes and a device-wide wait)
            torch.cuda.current_stream().wait_event(self.event)
        self.seen.add(cur)


def require_device(*tensors, what="t2h op"):
    """Argument validation that the C side cannot do: device, dtype, contiguity."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                f"{what}: expected a tensor on the MI355X (cuda device), got {t.device}. "
                "tomosar2height_amd has no CPU path; the CPU restatement lives in oracle/ for tests only.")
        if not t.is_contiguous():
            raise RuntimeError(f"{what}: tensor must be contiguous")
